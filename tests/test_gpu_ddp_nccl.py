"""BASELINE config 5 readiness on ONE GPU: the bf16 training step inside ``DistributedDataParallel`` over the
``nccl`` backend (= RCCL on ROCm) at world size 1 -- RCCL communicator creation, DDP's parameter broadcast at
construction and the bucketed gradient all-reduce hooks all execute -- must give the same losses as the bare
module.  Reference: examples/ycb_video/singleview_3d/train.py:342-369 (ChainerMN multi-node optimizer over
``pure_nccl``)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train(tmp_path, tag, *flags):
    out = tmp_path / f"{tag}.json"
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "examples/singleview_3d_train.py", "--steps", "3", "--global-batch", "2",
                        "--json", str(out), *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(open(out).read())


def test_ddp_over_rccl_world_size_one_matches_bare_module(tmp_path):
    ddp = _train(tmp_path, "ddp", "--ddp")
    bare = _train(tmp_path, "bare")
    assert ddp["ddp"] and ddp["backend"].startswith("nccl") and not bare["ddp"]
    assert np.isfinite(ddp["loss_per_step"]).all()
    # same seeds, same data; bf16 autocast + atomics in the backward ops leave run-to-run noise
    np.testing.assert_allclose(ddp["loss_per_step"], bare["loss_per_step"], rtol=0.03)
    assert ddp["objects_per_s_steady_mean"] > 0
