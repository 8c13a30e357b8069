"""Multi-GPU layout of the path: one process per GPU, scenes sharded
across ranks, no data-path collective until the refined poses are all-gathered once.

ICC couples the objects of ONE scene (every object's no-entry grid is max-ed with the
others' occupancy, contrib/iterative_collision_check_link.py:67-85) but scenes are
independent, and so are the per-object voxelize / 3D-CNN stages: the natural shard unit
is the scene.  The reference has no multi-GPU inference path at all (its only collective
is ChainerMN's gradient all-reduce in train.py:344); the pose all-gather below is the
exchange BASELINE config 4 names: [n_local, 7] float32 per rank (1.8 KB for 8 x 8 objects)
-> one latency-bound ``all_gather_into_tensor`` over RCCL/xGMI at the END of the step,
never per iteration.
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_procs, script, argv, env=None):
    """``python script --gpus N`` with no launcher around it: re-run ``script`` as N ranks of ONE
    node under ``torch.distributed.run`` (one process per GPU, rendezvous on 127.0.0.1 and a free
    port) and return its exit code.  The counterpart of the reference's ``mpirun -n N ... train.py
    --multi-node`` (README.md:130-148, train.py:228-233), without the external launcher.  Callers
    use it only when ``WORLD_SIZE`` is absent from the environment, i.e. not already launched."""
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_procs)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // int(n_procs))))
    return subprocess.call(cmd, env=e)


def launched():
    """True inside a rank started by torch.distributed.run / torchrun / the driver's launcher."""
    return "WORLD_SIZE" in os.environ


def rank_census(device, group=None):
    """Which ranks took part in a collective over ``group``: every rank contributes its own id to
    one all-gather; returns the list every rank sees (``[0..world-1]`` when the job is whole)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [0]
    world = dist.get_world_size(group)
    mine = torch.tensor([dist.get_rank(group)], dtype=torch.int64, device=device)
    out = torch.empty((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().tolist()


def shard_range(n_items, rank, world_size):
    """Contiguous block partition of ``range(n_items)``; the first ``n_items % world``
    ranks get one extra item.  Returns (start, stop)."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_poses(local_poses, group=None):
    """local_poses [n_local, 7] (q wxyz, t) -> [sum n_local, 7] on every rank, rank order.

    Equal shard sizes use a single ``all_gather_into_tensor``; ragged shards are padded
    to the maximum (the counts travel in the same call as an extra column)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_poses
    world = dist.get_world_size(group)
    n_local = torch.tensor([local_poses.shape[0]], device=local_poses.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    padded = local_poses.new_zeros((n_max, local_poses.shape[1]))
    padded[: local_poses.shape[0]] = local_poses
    out = local_poses.new_empty((world * n_max, local_poses.shape[1]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == n_max for c in counts):
        return out
    return torch.cat([out[r * n_max: r * n_max + c] for r, c in enumerate(counts)], dim=0)


def all_gather_poses_equal(local_poses, out=None, group=None, always=False):
    """Fast path for equal shards (the benchmark's weak-scaling layout): one collective,
    no host synchronisation, optional preallocated output.  ``always=True`` issues the collective even in a
    group of one rank (the single-GPU test of the RCCL path)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_poses
    if dist.get_world_size(group) == 1 and not always:
        return local_poses
    world = dist.get_world_size(group)
    if out is None:
        out = local_poses.new_empty((world * local_poses.shape[0], local_poses.shape[1]))
    dist.all_gather_into_tensor(out, local_poses.contiguous(), group=group)
    return out


def timed_steps(step, steps, warmup, device=None, group=None):
    """The benchmark's timing contract: ``warmup`` untimed calls of ``step()``, then exactly
    ``steps`` calls bracketed by (device sync + barrier) on both sides; returns the elapsed
    seconds of the SLOWEST rank (all-reduce MAX), identical on every rank."""
    cuda = device is not None and torch.device(device).type == "cuda"
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def fence():
        if cuda:
            torch.cuda.synchronize(device)
        if multi:
            dist.barrier(group=group)
        if cuda:
            torch.cuda.synchronize(device)

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if cuda else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(tt.item())
    return elapsed


def dense_grad_strides(module):
    """Give every parameter gradient the parameter's own strides.  MIOpen returns the weight gradient of a 1 x 1
    convolution with channels-last strides ([2560, 1, 2560, 2560] for a [1024, 2560, 1, 1] weight: the same bytes as
    the contiguous [2560, 1, 1, 1], the size-1 dimensions carry no information); DistributedDataParallel compares
    strides literally, warns "Grad strides do not match bucket view strides" and copies through a slower path.  A
    hook re-views such a gradient with the parameter's strides (no copy, no launch)."""
    def hook_for(p):
        def hook(g):
            if g.stride() != p.stride() and g.is_contiguous() and p.is_contiguous():
                return g.as_strided(g.shape, p.stride())
            return g
        return hook
    for p in module.parameters():
        if p.requires_grad and p.dim() == 4 and p.shape[2] == 1 and p.shape[3] == 1:
            p.register_hook(hook_for(p))
    return module


class DataParallelStep:
    """Data-parallel optimiser step whose device work can be replayed from hipGraphs (BASELINE config 5:
    ``train.py:228-233,342-344`` -- ChainerMN's multi-node optimizer all-reduces the gradients of every rank's
    share of the batch before the update).

    torch's ``DistributedDataParallel`` issues its bucketed all-reduces from autograd hooks, i.e. INSIDE the backward
    pass: captured into a hipGraph on this stack (torch 2.10 + ROCm 7 RCCL) ProcessGroupNCCL's watchdog queries an
    event of the capturing stream and the process aborts (``hipErrorCapturedEvent``,
    profiles/r04_train_hipgraph_under_ddp_abort.log).  Here the exchange sits BETWEEN two graphs instead:

    * phase 1 (capturable): forward, backward, then every gradient is copied into ONE flat float32 bucket
      (a few multi-tensor copies);
    * exchange (always eager): one ``all_reduce`` of the bucket over the process group -- RCCL over xGMI on the GPUs
      (124 MB fp32 for the pose network: ~1.4 ms ring lower bound over one xGMI link pair), gloo in the
      CPU tests -- followed by the division by the world size inside phase 2;
    * phase 2 (capturable): the optimiser step on the bucket's views (``param.grad`` aliases its slice).

    What is lost against DDP is the overlap of the exchange with the backward pass; what is gained is the replay of
    the ~1100 launches of the step (22.1 -> 19.0 ms on one MI355X).  ``loss_fn(**inputs)`` runs the module's forward
    and returns the scalar loss; ``group=None`` with an initialised default group reduces over it, without one the
    exchange is skipped (single process)."""

    def __init__(self, params, optimizer, loss_fn, group=None, exchange=None):
        self.params = [p for p in params if p.requires_grad]
        self.optimizer = optimizer
        self.loss_fn = loss_fn
        self.group = group
        self.exchange = (dist.is_available() and dist.is_initialized()) if exchange is None else exchange
        self.world = dist.get_world_size(group) if self.exchange else 1
        dev, n = self.params[0].device, sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.graph_fb = self.graph_opt = self.static = self.static_loss = None

    # -- the three parts of a step ------------------------------------------------------------------
    def _forward_backward(self, inputs):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(**inputs)
        loss.backward()
        self._pack()
        return loss

    def _pack(self):
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(self.views, self.params):
            if p.grad is None:  # a parameter the loss does not reach contributes zeros (DDP's find_unused_parameters)
                v.zero_()

    def _all_reduce(self):
        if self.exchange:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def _update(self):
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        for v, p in zip(self.views, self.params):
            p.grad = v
        self.optimizer.step()

    # -- eager ----------------------------------------------------------------------------------------
    def step(self, inputs):
        loss = self._forward_backward(inputs)
        self._all_reduce()
        self._update()
        return loss

    # -- captured ---------------------------------------------------------------------------------------
    def capture(self, inputs, stream):
        """Capture phase 1 and phase 2 on ``stream`` (the side stream the eager warm-up steps ran on, torch's capture
        recipe; the optimiser must be ``capturable``).  ``inputs``: dict of device tensors; their clones become the
        graphs' static inputs."""
        self.static = {k: v.clone() for k, v in inputs.items()}
        self.optimizer.zero_grad(set_to_none=True)
        # ProcessGroupNCCL's watchdog THREAD polls the events of the eager all-reduces (hipEventQuery): under the
        # default global capture mode any such call from any thread while this thread captures is an error and the
        # process aborts (seen at world size 1 when a warm-up step's all-reduce was still being reaped).  The exchange
        # is never part of a capture here.
        # Thread-local mode was not enough on every run (one abort in six processes: the watchdog's query landed inside
        # the capture and was still refused), so: relaxed mode -- no call of any thread is checked; nothing this thread
        # does while capturing is unsafe (allocations come from the graph's pool) -- and the watchdog gets time to
        # reap the finished all-reduces (its poll period is 100 ms) before the capture begins.
        torch.cuda.synchronize()
        if self.exchange:
            import time
            time.sleep(0.3)
        mode = dict(capture_error_mode="relaxed")
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb, stream=stream, **mode):
            self.static_loss = self._forward_backward(self.static)
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt, stream=stream, pool=self.graph_fb.pool(), **mode):
            self._update()

    def replay(self, inputs):
        for k, v in inputs.items():
            self.static[k].copy_(v)
        self.graph_fb.replay()
        self._all_reduce()
        self.graph_opt.replay()
        return self.static_loss
