import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    from morefusion_amd import miopen_cache
    miopen_cache.enable()  # stock-backbone solver choices shipped with the package: no MIOpen search per test process


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def fixtures3():
    """The three real ICC instances the reference ships
    (examples/ycb_video/pose_refinement/data/0000000{0,1,2}.npz)."""
    return [golden(f"fixture_pose_refinement_0000000{i}.npz") for i in range(3)]
