# flake8: noqa
from .backbone2d import PSPNetExtractor, ResNet18, ResNet18Extractor
