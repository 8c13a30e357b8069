"""`python bench.py --gpus N` and `python examples/singleview_3d_train.py --gpus N` launch their own N
ranks (no external torchrun): proven here without a GPU through the `--dry-run-cpu` legs -- N gloo
ranks, a stub step, the real launcher, the real pose all-gather / DDP plumbing, ONE JSON line from
rank 0.  The reference's analogue is `mpirun -n N ... train.py --multi-node` (README.md:130-148,
examples/ycb_video/singleview_3d/train.py:228-233,312-318)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}:\n{p.stdout}"
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    out = _run(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    c = out["collective"]
    assert c["ranks_seen"] == [0, 1]
    assert c["gathered_rows"] == 2 * c["rows_per_rank"] == 16   # world x n_local rows of [*,7] poses
    assert c["rank_order_ok"]
    assert out["value"] is None and "DRY RUN" in out["data"]    # never mistaken for a measurement


def test_bench_single_rank_needs_no_launcher():
    out = _run(["bench.py", "--steps", "2", "--warmup", "0", "--dry-run-cpu"])
    assert out["n_gpus"] == 1 and out["collective"]["gathered_rows"] == 8


def test_bench_respects_an_external_launcher():
    """The driver's `python -m torch.distributed.run ... bench.py --gpus N` must not re-launch."""
    from morefusion_amd import parallel
    port = parallel.free_port()
    out = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
                "--dry-run-cpu"])
    assert out["n_gpus"] == 2 and out["collective"]["ranks_seen"] == [0, 1]


def test_training_example_self_launches_two_ranks():
    out = _run(["examples/singleview_3d_train.py", "--gpus", "2", "--dry-run-cpu"])
    assert out["n_ranks"] == 2 and out["ranks_seen"] == [0, 1] and out["ddp_gradient_is_rank_mean"]
