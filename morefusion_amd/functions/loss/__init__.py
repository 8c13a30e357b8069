# flake8: noqa
from .average_distance import average_distance
