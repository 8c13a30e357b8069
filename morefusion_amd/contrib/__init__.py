# flake8: noqa
# the hot-path subset of morefusion/contrib/__init__.py:3-11
from .icc_batch import IccScenes
from .iterative_closest_point_link import IterativeClosestPointLink, icp_refine
from .iterative_collision_check_link import IterativeCollisionCheckLink
from .occupancy_registration import OccupancyRegistration, OccupancyRegistrationLink
from . import singleview_3d
