# flake8: noqa
from .average_distance import average_distance, average_distance_batch  # noqa: F401
from .confidence_loss import confidence_loss  # noqa: F401
