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
