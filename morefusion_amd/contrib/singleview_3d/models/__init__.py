# flake8: noqa
from .model import Model, PitchTableModels
