# flake8: noqa
from . import models
