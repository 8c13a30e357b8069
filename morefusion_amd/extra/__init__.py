# flake8: noqa
from . import _torch as torch_
from ._torch import median
