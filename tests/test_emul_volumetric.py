"""The channels-last inference path of the pose network (volumetric_cl.py: point prep, point MLP, occupancy convs,
conv3 dense + sparse, conv4 implicit GEMM, channels-last samplers, points-major heads, pose epilogue) END TO END ON
THE CPU, with EVERY buffer any kernel is handed fenced by an inaccessible page: the kernel text of conv3d.hip /
sparseconv.hip / interp.hip / linear.hip / pointops.hip (and psp_tail.hip / preprocess.hip for the front end) runs
behind the fiber emulator with torch CPU tensors as "device" memory; ``emul.GuardedTensors`` mirrors the storage
behind every pointer into a mapping that ends (run 1) or starts (run 2) at a PROT_NONE page.  The (rot, trans, conf)
of the path are compared with the channels-first dense formulation of contrib/singleview_3d/models/model.py:93-164,
232-275 (``Model._extract`` with dense Conv3d, oracle voxel ops).  An out-of-bounds access of one byte is a SIGSEGV:
each run is a child process (``python -X faulthandler`` names the library call).  One object: conv4 alone is 2 M
emulated MFMA wave-instructions (~30 s per run)."""
import os
import subprocess
import sys

import pytest

from host_emul import emul

pytestmark = pytest.mark.skipif(not emul.available(), reason="g++ not available")
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(case, side, tmp_path):
    log = tmp_path / f"{case}_{side}.calls"
    env = dict(os.environ, MF_GUARD_LOG=str(log))
    p = subprocess.run([sys.executable, "-X", "faulthandler", os.path.join(HERE, "host_emul", "guard_case.py"), case, side],
                       capture_output=True, text=True, timeout=900, env=env)
    last = log.read_text().strip().splitlines()[-1:] if log.exists() else []
    assert p.returncode == 0 and f"GUARD_OK {case} {side}" in p.stdout, (
        f"exit {p.returncode}; last library call: {last}\n" + p.stdout[-1500:] + p.stderr[-3000:])


@pytest.mark.parametrize("side", ["after", "before"])
def test_channels_last_volumetric_path_matches_dense_formulation_behind_guard_pages(side, tmp_path):
    _run("volumetric", side, tmp_path)


@pytest.mark.parametrize("side", ["after", "before"])
def test_front_end_kernels_behind_guard_pages(side, tmp_path):
    """k_psp_tail through PSPNetExtractor.forward_sampled_rows (image corners / edges, NCHW and channels-last maps)
    and k_valid_order through Model._select_points (ragged counts, an exact-chunk and a ragged-chunk image size)."""
    _run("frontend", side, tmp_path)


@pytest.mark.parametrize("side", ["after", "before"])
def test_bf16_training_operators_behind_guard_pages(side, tmp_path):
    """csrc/gemm_bf16.hip (NT / TN engines, packs, finish passes) and the channels-last bf16 voxelization / sampling
    kernels, forward and backward, with every tensor against an inaccessible page: their masked operand chunks are
    buffer loads at an out-of-range offset over a 2 GB span -- nothing in hardware stops a wrong UNMASKED offset at
    the end of the tensor, so the bound is checked here."""
    _run("training", side, tmp_path)
