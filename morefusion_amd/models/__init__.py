# flake8: noqa
from .backbone2d import PSPNetExtractor, ResNet18
