# flake8: noqa
from .auc_for_errors import auc_for_errors
from .average_distance import average_distance
from .ycb_video_add_auc import ycb_video_add_auc
