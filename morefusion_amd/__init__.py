"""morefusion_amd -- MI355X-native volumetric pose hot path of MoreFusion.

Same call surface as the reference's ``morefusion.functions`` / ``morefusion.geometry`` /
``morefusion.contrib`` for the voxelize -> 3D-CNN -> ICC/ICP path, hosted on PyTorch-ROCm
and backed by hand-written gfx950 kernels in ``libmfhip.so`` (include/mfhip.h).
"""
# flake8: noqa
__version__ = "0.1.0"

from . import _lib
from . import functions
from . import geometry
from . import metrics
from . import extra
from . import optimizers
from . import contrib
from . import synthetic
from . import data_formats
from . import serializers
from . import training
