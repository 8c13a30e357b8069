#!/usr/bin/env python
"""singleview_3d training step loop -- counterpart of the reference's
examples/ycb_video/singleview_3d/train.py:143-493 for BASELINE config 5
(`--with-occupancy`, bf16, data-parallel): synthetic example dicts (no dataset is reachable
offline), Adam(lr 1e-4) (train.py:342), batch 16 // n_gpu per rank (train.py:361).

Single GPU:   python examples/singleview_3d_train.py --steps 5
N GPUs (DP):  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
                  examples/singleview_3d_train.py --steps 5
The reference all-reduces gradients with ChainerMN `pure_nccl` (train.py:231,344); here it is
torch DDP over RCCL (bucketed all-reduce overlapped with backward).  Convolutions / GEMMs run
under bf16 autocast; the voxel ops and the loss stay fp32 (the HIP kernels are fp32).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as morefusion  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model, PitchTableModels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=16)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-bf16", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    torch.manual_seed(0)  # identical initial weights on every rank
    rs = np.random.RandomState(0)
    pcds = {c: rs.uniform(-0.05, 0.05, (2000, 3)).astype(np.float32) for c in morefusion.synthetic.CLASS_PITCH}
    model = Model(n_fg_class=21, with_occupancy=True, models=PitchTableModels(pcds)).to(device).train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank]) if world > 1 else model
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)

    per_rank = max(1, args.global_batch // world)
    np.random.seed(1234 + rank)  # per-rank point / CAD subsampling streams
    for step in range(args.steps):
        b = morefusion.synthetic.make_singleview_batch(per_rank, seed=1000 * rank + step)
        inputs = {k: torch.as_tensor(b[k]).to(device) for k in
                  ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty",
                   "quaternion_true", "translation_true")}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.no_bf16):
            loss = net(**inputs)
        loss.backward()
        optimizer.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            lt = loss.detach().clone()
            dist.all_reduce(lt)
            loss_avg = float(lt) / world
        else:
            loss_avg = float(loss.detach())
        if rank == 0:
            print(f"step {step}: loss {loss_avg:.5f}  {per_rank * world / dt:.1f} objects/s "
                  f"(global batch {per_rank * world}, {world} GPU(s))", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
