"""world_size-2 gloo run of the multi-GPU layout on CPU: scene sharding + the single
pose all-gather (equal and ragged shards).  The same code path runs over RCCL on GPUs."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from morefusion_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_scenes, objs_per_scene, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = parallel.shard_range(n_scenes, rank, world)
        # a rank's "refined poses": deterministic function of the global object id
        ids = np.arange(a * objs_per_scene, b * objs_per_scene)
        local = torch.from_numpy(np.stack([ids * 10.0 + k for k in range(7)], 1).astype(np.float32))
        gathered = parallel.all_gather_poses(local)
        np.save(os.path.join(out_dir, f"ragged_{rank}.npy"), gathered.numpy())
        if n_scenes % world == 0:
            g2 = parallel.all_gather_poses_equal(local)
            np.save(os.path.join(out_dir, f"equal_{rank}.npy"), g2.numpy())
    finally:
        dist.destroy_process_group()


def _run(n_scenes, objs, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_scenes, objs, str(tmp_path)), nprocs=2, join=True)
    ids = np.arange(n_scenes * objs)
    expect = np.stack([ids * 10.0 + k for k in range(7)], 1).astype(np.float32)
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"ragged_{r}.npy"), expect)
        if n_scenes % 2 == 0:
            np.testing.assert_array_equal(np.load(tmp_path / f"equal_{r}.npy"), expect)


def test_pose_all_gather_equal_shards(tmp_path):
    _run(4, 8, tmp_path)  # config 4 in miniature: scenes x 8 objects


def test_pose_all_gather_ragged_shards(tmp_path):
    _run(3, 5, tmp_path)  # 2 + 1 scenes


def test_single_process_is_identity():
    x = torch.arange(21.0).reshape(3, 7)
    assert parallel.all_gather_poses(x) is x
    assert parallel.all_gather_poses_equal(x) is x


# ---- evaluator aggregation across ranks (pose_estimation_evaluator.py:82-90) -------------
def _eval_batches():
    # 4 batches; per-instance ADD keys as Model.evaluate(per_instance=True) reports them
    return [
        {"loss": 1.0, "add/0002/a": 0.010, "add_s/0002/a": 0.005, "add_or_add_s/0002/a": 0.010},
        {"loss": 3.0, "add/0002/b": 0.030, "add_s/0002/b": 0.015, "add_or_add_s/0002/b": 0.030,
         "add/0013/c": 0.200, "add_s/0013/c": 0.004, "add_or_add_s/0013/c": 0.004},
        {"loss": 2.0, "add/0013/d": 0.050, "add_s/0013/d": 0.008, "add_or_add_s/0013/d": 0.008},
        {"loss": 6.0, "add/0002/e": 0.012, "add/0002/f": 0.018, "add_s/0002/e": 0.002,
         "add_s/0002/f": 0.003, "add_or_add_s/0002/e": 0.012, "add_or_add_s/0002/f": 0.018},
    ]


def _eval_worker(rank, world, port, out_dir):
    import json
    from morefusion_amd.training import PoseEstimationEvaluator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batches = _eval_batches()
        a, b = parallel.shard_range(len(batches), rank, world)
        result = PoseEstimationEvaluator(batches[a:b], lambda **kw: kw)()
        with open(os.path.join(out_dir, f"eval_{rank}.json"), "w") as f:
            json.dump(result, f)
    finally:
        dist.destroy_process_group()


def test_evaluator_summary_single_and_two_ranks(tmp_path):
    import json
    from morefusion_amd import metrics
    from morefusion_amd.training import PoseEstimationEvaluator, summarize_observations
    single = PoseEstimationEvaluator(_eval_batches(), lambda **kw: kw)()
    P = "validation/main/"
    assert single[P + "loss"] == 3.0
    # class 0002: batches 1, 2, 4 report it; batch 4 holds two instances -> the last one enters the mean
    np.testing.assert_allclose(single[P + "add/0002"], (0.010 + 0.030 + 0.018) / 3)
    np.testing.assert_allclose(single[P + "add/0013"], (0.200 + 0.050) / 2)
    np.testing.assert_allclose(single[P + "add"], (single[P + "add/0002"] + single[P + "add/0013"]) / 2)
    # AUC / <2cm use every instance
    v2 = [0.010, 0.030, 0.012, 0.018]
    np.testing.assert_allclose(single[P + "auc/add/0002"], metrics.ycb_video_add_auc(v2, max_value=0.1))
    assert single[P + "<2cm/add/0002"] == 0.75 and single[P + "<2cm/add/0013"] == 0.0
    assert single[P + "<2cm/add_s/0013"] == 1.0
    np.testing.assert_allclose(single[P + "auc/add"],
                               (single[P + "auc/add/0002"] + single[P + "auc/add/0013"]) / 2)
    np.testing.assert_allclose(single[P + "<2cm/add_or_add_s"], (0.75 + 1.0) / 2)
    # NaN entries are dropped like DataFrame.dropna()
    assert summarize_observations([{P + "loss": float("nan")}, {P + "loss": 2.0}])[P + "loss"] == 2.0

    port = _free_port()
    mp.spawn(_eval_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "eval_0.json"))
    r1 = json.load(open(tmp_path / "eval_1.json"))
    assert r1 == {}  # only rank 0 reports
    assert r0.keys() == single.keys()
    for k in single:
        np.testing.assert_allclose(r0[k], single[k], rtol=1e-12)


# ---- the benchmark's timing contract (bench.py) under two ranks ---------------------------
def _timed_worker(rank, world, port, out_dir):
    import json
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []
        local = torch.full((2, 7), float(rank))

        def step():  # rank 1 is the slow one; every step ends with the pose all-gather
            time.sleep(0.01 * (1 + 2 * rank))
            calls.append(parallel.all_gather_poses_equal(local).shape[0])

        elapsed = parallel.timed_steps(step, steps=5, warmup=2)
        with open(os.path.join(out_dir, f"timed_{rank}.json"), "w") as f:
            json.dump(dict(elapsed=elapsed, calls=calls), f)
    finally:
        dist.destroy_process_group()


def test_timed_steps_reports_the_slowest_rank(tmp_path):
    import json
    port = _free_port()
    mp.spawn(_timed_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "timed_0.json"))
    r1 = json.load(open(tmp_path / "timed_1.json"))
    assert r0["calls"] == r1["calls"] == [4] * 7  # 2 warm-up + 5 timed steps, gathered [2*2, 7]
    assert r0["elapsed"] == r1["elapsed"]          # MAX over ranks, identical everywhere
    assert 0.15 <= r0["elapsed"] < 1.0             # 5 x 30 ms of the slow rank, not 5 x 10 ms
    # single process: no collectives, plain wall clock
    n = []
    assert parallel.timed_steps(lambda: n.append(1), steps=3, warmup=1) >= 0.0 and len(n) == 4
