#!/usr/bin/env python
"""singleview_3d training step loop -- counterpart of the reference's
examples/ycb_video/singleview_3d/train.py:143-493 for BASELINE config 5
(`--with-occupancy`, bf16, data-parallel): synthetic example dicts (no dataset is reachable
offline), Adam(lr 1e-4) (train.py:342), batch 16 // n_gpu per rank (train.py:361).

Single GPU:   python examples/singleview_3d_train.py --steps 5
N GPUs (DP):  python examples/singleview_3d_train.py --gpus N --steps 5      (launches its own N ranks;
              an external `python -m torch.distributed.run --nproc-per-node N ...` works too)
The reference all-reduces gradients with ChainerMN `pure_nccl` (train.py:231,344); here it is
torch DDP over RCCL (bucketed all-reduce overlapped with backward).  Under bf16 autocast the 3-D CNN, the 1x1
convolutions, voxelization and sampling run on the hand-written bf16 kernels (models/bf16_ops.py), the 2-D backbone
on MIOpen, the loss on its fp32 HIP op.
``--graph``: the step's device work is captured into TWO hipGraphs and replayed -- forward + backward (+ the copy of
every gradient into one flat bucket), then Adam; between them, eager, the all-reduce of the bucket over RCCL (in
``--exchange-chunks`` pieces) when the run is data-parallel (``--ddp`` / ``--gpus N``: parallel.DataParallelStep --
torch's DDP cannot be captured on this stack).  The capture comes FIRST: three eager warm-up steps without the
exchange, their effect on parameters / BatchNorm statistics / Adam's state undone, the capture, and only then the
process's first collective (no ProcessGroupNCCL watchdog polls events during the capture); every step is a replay.  The host keeps what the reference does on the host (the NumPy-RNG point
selection and CAD subsample).  The eager step is launch-bound for a fifth of its time (~1100 launches, 22.1 ms for
17.7 ms of kernels): 742 -> 841 objects/s on one MI355X (profiles/r04_train_1gpu_bf16_hipgraph_step.json).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as morefusion  # noqa: E402
from morefusion_amd import miopen_cache, parallel  # noqa: E402

miopen_cache.enable()  # shipped MIOpen solver choices for the stock 2-D backbone's shapes (no search on a fresh box)
from morefusion_amd.contrib.singleview_3d.models import Model, PitchTableModels  # noqa: E402


def dry_run_cpu(args, world, rank):
    """Launcher + DDP plumbing without a GPU: N gloo ranks wrap a stub module in DistributedDataParallel,
    take one optimiser step on rank-dependent data and check that every rank holds the rank-averaged
    gradient (the all-reduce train.py:344 gets from ChainerMN).  No kernel of the pose network runs."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    stub = torch.nn.Linear(7, 3)
    net = torch.nn.parallel.DistributedDataParallel(stub) if world > 1 else stub
    x = torch.full((4, 7), float(rank + 1))
    net(x).sum().backward()
    # d/dW sum(Wx + b) = sum_rows x  -> rank r contributes 4 (r + 1); DDP leaves the mean over ranks
    expect = 4.0 * sum(r + 1 for r in range(world)) / world
    ok = bool(torch.allclose(stub.weight.grad, torch.full_like(stub.weight.grad, expect)))
    census = parallel.rank_census(torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"dry_run": "cpu/gloo stub module, no pose-network kernel ran", "n_ranks": world,
                          "ranks_seen": census, "ddp_gradient_is_rank_mean": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: the launcher's WORLD_SIZE or 1")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=16)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-bf16", action="store_true")
    ap.add_argument("--ddp", action="store_true",
                    help="wrap the model in DistributedDataParallel over RCCL even at world size 1")
    ap.add_argument("--graph", action="store_true",
                    help="capture forward + backward + Adam of one step into a hipGraph after --graph-warmup eager "
                         "steps and replay it (single process; the host keeps the point selection and the CAD subsample)")
    ap.add_argument("--graph-warmup", type=int, default=3)
    ap.add_argument("--exchange-chunks", type=int, default=4,
                    help="--graph --ddp: the flat gradient bucket is all-reduced in this many chunks (async, awaited "
                         "in front of the update)")
    ap.add_argument("--no-dropout", action="store_true",
                    help="PSPNet's dropouts off (pspnet.py:24-30): makes an eager and a --graph run comparable step by step")
    ap.add_argument("--json", default=None, help="write a one-line JSON record of the run to this path")
    ap.add_argument("--dry-run-cpu", action="store_true", help="launcher + DDP plumbing on CPU/gloo, stub module")
    args = ap.parse_args()

    if args.gpus and args.gpus > 1 and not parallel.launched():
        raise SystemExit(parallel.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus in (None, world), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_run_cpu:
        return dry_run_cpu(args, world, rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_ddp = world > 1 or args.ddp
    # --graph with data parallelism: NOT through DistributedDataParallel (its all-reduces run inside the backward
    # pass: captured, ProcessGroupNCCL's watchdog aborts the process with hipErrorCapturedEvent on this stack, round 4)
    # but through parallel.DataParallelStep -- forward + backward and the optimiser step are two graphs, ONE flat
    # gradient all-reduce over RCCL sits between them, eager.
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(parallel.free_port()))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.no_dropout:
        torch.nn.functional.dropout = lambda x, p=0.5, training=True, inplace=False: x
    torch.manual_seed(0)  # identical initial weights on every rank
    rs = np.random.RandomState(0)
    pcds = {c: rs.uniform(-0.05, 0.05, (2000, 3)).astype(np.float32) for c in morefusion.synthetic.CLASS_PITCH}
    model = Model(n_fg_class=21, with_occupancy=True, models=PitchTableModels(pcds)).to(device).train()
    if use_ddp:
        parallel.dense_grad_strides(model)  # (MIOpen's 1x1-convolution weight gradients: see there)
    side = torch.cuda.Stream(device=device) if args.graph else None
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank]) if use_ddp and not args.graph else model
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, capturable=bool(args.graph))

    def autocast_loss(**kw):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.no_bf16):
            return net(**kw)

    dp = parallel.DataParallelStep(model.parameters(), optimizer, autocast_loss, exchange=use_ddp,
                                   chunks=args.exchange_chunks) if args.graph else None

    per_rank = max(1, args.global_batch // world)
    np.random.seed(1234 + rank)  # per-rank point / CAD subsampling streams
    rates, losses = [], []
    # profiling aid: MF_TRAIN_MARK=1 launches a recognisable kernel (erfinv) at program start and in front of the
    # last two steps, so that `MF_MARK=erfinv tools/kernel_stats.py` keeps exactly one steady-state step
    mark = (lambda: torch.zeros(1, device=device).erfinv_()) if os.environ.get("MF_TRAIN_MARK") == "1" else (lambda: None)
    mark()
    KEYS = ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty", "quaternion_true", "translation_true")

    def device_inputs(inp):
        """The uploaded batch -> everything the device side of the step reads: the network inputs plus what the host
        decides (point selection: one synchronisation; CAD subsample + ADD-S flags: host RNG, model.py:207-220,411-414)."""
        pix = model._select_points(inp["pcd"])  # first: the eager step draws the point subsample before the CAD one
        cad, sym = model.loss_prepare(inp["class_id"], device)
        assert torch.is_tensor(cad), "--graph needs CAD clouds of one size"
        return dict(class_id=inp["class_id"], rgb=inp["rgb"], pcd=inp["pcd"], pix=pix,
                    pitch=inp["pitch"].float(), origin=inp["origin"].float(),
                    grid_nontarget_empty=inp["grid_nontarget_empty"], quaternion_true=inp["quaternion_true"],
                    translation_true=inp["translation_true"], cad=cad, symmetric=sym)

    def eager_step(inputs):
        optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.no_bf16):
            loss = net(**inputs)
        loss.backward()
        optimizer.step()
        return loss

    for step in range(args.steps):
        if step >= args.steps - 2:
            mark()
        b = morefusion.synthetic.make_singleview_batch(per_rank, seed=1000 * rank + step)
        inputs = {k: torch.as_tensor(b[k]).to(device) for k in KEYS}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if not args.graph:
            loss = eager_step(inputs)
        else:
            new = device_inputs(inputs)
            if step == 0:
                # Capture FIRST, exchange afterwards (parallel.DataParallelStep.capture_before_exchange): the eager
                # warm-up steps of torch's capture recipe run on the side stream WITHOUT the gradient exchange --
                # they only bring the gradient accumulators, Adam's state and MIOpen's solver choices into being --,
                # parameters / BatchNorm statistics / optimiser state go back to their initial values, the two
                # graphs are captured, and only then does this process issue its first collective: no
                # ProcessGroupNCCL watchdog has an event to poll while the capture runs.  A failed capture leaves
                # the step eager (dp.capture_error).
                captured = dp.capture_before_exchange(new, side, warmup=args.graph_warmup, buffers=model.buffers())
                if rank == 0:
                    print("hipGraph capture:", "ok" if captured else f"FAILED, eager steps ({dp.capture_error})", flush=True)
            loss = dp.replay(new)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            lt = loss.detach().clone()
            dist.all_reduce(lt)
            loss_avg = float(lt) / world
        else:
            loss_avg = float(loss.detach())
        rates.append(per_rank * world / dt)
        losses.append(loss_avg)
        if rank == 0:
            print(f"step {step}: loss {loss_avg:.5f}  {per_rank * world / dt:.1f} objects/s "
                  f"(global batch {per_rank * world}, {world} GPU(s))"
                  + ("  [hipGraph replay]" if args.graph and dp.graph_fb is not None and step > 0 else
                     "  [warm-up + capture + replay]" if args.graph and step == 0 else ""), flush=True)
    if rank == 0 and args.json:
        skip = 2  # the first steps carry MIOpen's algorithm search / the warm-up and the capture
        steady = rates[skip:] or rates
        rec = {"what": "singleview_3d training step (BASELINE config 5 on this many GPUs), synthetic batch",
               "n_gpus": world, "global_batch": per_rank * world, "dtype": "f32" if args.no_bf16 else "bf16 autocast",
               "ddp": bool(use_ddp), "backend": "nccl (RCCL)" if use_ddp else None, "steps": args.steps,
               "hipgraph_step": bool(args.graph and dp.graph_fb is not None),
               "hipgraph_capture_error": dp.capture_error if args.graph else None,
               "exchange_chunks": len(dp.chunk_flat) if args.graph else None,
               "objects_per_s_steady_mean": round(float(np.mean(steady)), 2),
               "objects_per_s_per_step": [round(r, 2) for r in rates], "loss_per_step": [round(x, 5) for x in losses]}
        with open(args.json, "w") as f:
            f.write(json.dumps(rec) + "\n")
    if use_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
