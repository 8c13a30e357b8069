"""BASELINE config 5 on the CPU: two-rank data-parallel training step of the pose network over gloo.

The reference trains with ChainerMN's multi-node optimizer (gradient all-reduce,
examples/ycb_video/singleview_3d/train.py:228-233,342-344); here the model is wrapped in torch's
``DistributedDataParallel`` (RCCL on the GPUs, gloo in this test).  The two HIP ops of the
training graph (dense average_voxelization_3d, interpolate_voxel_grid) and the valid-pixel
compaction are replaced by autograd stand-ins built on the oracle (test-only), everything else is
the product's host code: ``Model.forward`` -> ``predict`` -> ``loss`` (ADD; non-symmetric classes).
Checked: after ``backward`` every rank holds the MEAN of the two ranks' local gradients, equal to
what two single-process runs on the same data and seeds average to."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WATCH = ("conv1_rot.weight", "conv3.weight", "conv4.weight", "resnet_extractor.res5.1.conv2.weight",
         "pspnet_extractor.up3.conv.weight", "conv1_occ.weight")


def _install_standins():
    from oracle import oracle_c as OC
    from oracle import oracle_np as O
    import morefusion_amd.contrib.singleview_3d.models.model as model_mod
    from morefusion_amd.contrib.singleview_3d.models import Model

    class AvgVox(torch.autograd.Function):
        @staticmethod
        def forward(ctx, values, points, bi, B, origin, pitch, dims):
            m, c = OC.average_voxelization_3d(values.detach().numpy(), points.numpy(), bi.numpy(), batch_size=B,
                                              origin=origin, pitch=pitch, dimensions=dims)
            ctx.aux = (points.numpy(), bi.numpy(), c, origin, pitch, dims)
            return torch.from_numpy(m), torch.from_numpy(c)

        @staticmethod
        def backward(ctx, gm, gc):
            points, bi, c, origin, pitch, dims = ctx.aux
            gv = O.average_voxelization_3d_backward(gm.numpy(), points, bi, c, origin=origin, pitch=pitch,
                                                    dimensions=dims, mode="gpu")
            return torch.from_numpy(gv), None, None, None, None, None, None

    def avg_cpu(values, points, batch_indices, *, batch_size, origin, pitch, dimensions, return_counts=False, **kw):
        m, c = AvgVox.apply(values, points, batch_indices, batch_size, origin, pitch, dimensions)
        return (m, c) if return_counts else m

    class Interp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, vox, points, bi):
            ctx.aux = (points.numpy(), bi.numpy(), tuple(vox.shape))
            return torch.from_numpy(OC.interpolate_voxel_grid(vox.detach().numpy(), points.numpy(), bi.numpy()))

        @staticmethod
        def backward(ctx, g):
            points, bi, shape = ctx.aux
            return torch.from_numpy(O.interpolate_voxel_grid_backward(np.ascontiguousarray(g.numpy()), points, bi,
                                                                      shape, mode="gpu")), None, None

    def interp_cpu(vox, points, batch_indices, channels_first=False, batch_start=None):
        out = Interp.apply(vox, points, batch_indices)
        return out.t().contiguous() if channels_first else out

    def select_cpu(self, pcd):
        order, counts = O.valid_pixel_order(pcd.numpy())
        return self._subsample(torch.from_numpy(order), counts)

    model_mod.functions_module.average_voxelization_3d = avg_cpu
    model_mod.functions_module.interpolate_voxel_grid = interp_cpu
    Model._select_points = select_cpu
    return Model


def _batch(rank):
    import morefusion_amd as mf
    b = mf.synthetic.make_singleview_batch(1, seed=50 + rank)
    b["class_id"] = np.array([2], np.int32)  # a non-symmetric class: ADD (the CPU composite of the loss)
    return {k: torch.as_tensor(b[k]) for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty",
                                               "quaternion_true", "translation_true")}


def _build(Model):
    import morefusion_amd as mf
    from morefusion_amd.contrib.singleview_3d.models import PitchTableModels
    torch.manual_seed(0)
    rs = np.random.RandomState(0)
    pcds = {c: rs.uniform(-0.05, 0.05, (600, 3)).astype(np.float32) for c in mf.synthetic.CLASS_PITCH}
    return Model(n_fg_class=21, with_occupancy=True, models=PitchTableModels(pcds)).train()


def _local_step(net, rank):
    np.random.seed(1234 + rank)
    torch.manual_seed(77 + rank)  # dropout masks
    loss = net(**_batch(rank))
    loss.backward()
    return float(loss.detach())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _build(_install_standins())
        extra = {}
        if mode == "ddp":
            net = torch.nn.parallel.DistributedDataParallel(model)
            loss = _local_step(net, rank)
        else:
            # parallel.DataParallelStep (round 5): forward + backward, the flat gradient bucket all-reduced (whole, or
            # -- round 6 -- in chunks issued from autograd hooks while the backward pass still runs), update -- the
            # form whose two device phases are captured into hipGraphs on the GPU (`--ddp --graph`)
            from morefusion_amd.parallel import DataParallelStep
            w0 = {n: p.detach().clone() for n, p in model.named_parameters() if n in WATCH}
            opt = torch.optim.SGD(model.parameters(), lr=0.5)
            chunks = 5 if mode == "flat_bucket_chunked" else 1
            dp = DataParallelStep(model.parameters(), opt, lambda **kw: model(**kw), chunks=chunks)
            assert dp.world == 2 and dp.flat.numel() == sum(p.numel() for p in model.parameters())
            # the chunks tile the bucket: whole parameters each, in order, nothing lost
            assert len(dp.chunk_flat) == chunks and sum(c.numel() for c in dp.chunk_flat) == dp.flat.numel()
            assert dp.chunk_first[0] == 0 and dp.chunk_first[-1] == len(dp.params)
            np.random.seed(1234 + rank)
            torch.manual_seed(77 + rank)  # dropout masks
            loss = float(dp.step(_batch(rank)).detach())
            # chunked: every chunk went out on its own (issued from the backward pass' hooks, the last layers' first)
            assert dp.exchanges == chunks
            extra["update"] = {n: (w0[n] - p.detach()) / 0.5 for n, p in model.named_parameters() if n in WATCH}
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if n in WATCH}
        torch.save(dict({"loss": loss, "grads": grads}, **extra), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ddp", "flat_bucket", "flat_bucket_chunked"])
def test_ddp_training_step_averages_the_two_ranks_gradients(tmp_path, mode):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    # single-process references on the same data / seeds
    import morefusion_amd.contrib.singleview_3d.models.model as model_mod
    from morefusion_amd.contrib.singleview_3d.models import Model as RealModel
    saved = (model_mod.functions_module.average_voxelization_3d, model_mod.functions_module.interpolate_voxel_grid,
             RealModel._select_points, torch.get_num_threads())
    torch.set_num_threads(4)
    try:
        Model = _install_standins()
        ref_grads, ref_loss = [], []
        for r in range(2):
            model = _build(Model)
            ref_loss.append(_local_step(model, r))
            ref_grads.append({n: p.grad.clone() for n, p in model.named_parameters() if n in WATCH})
    finally:
        (model_mod.functions_module.average_voxelization_3d, model_mod.functions_module.interpolate_voxel_grid,
         RealModel._select_points) = saved[:3]
        torch.set_num_threads(saved[3])
    for r in range(2):
        assert abs(got[r]["loss"] - ref_loss[r]) < 1e-5 * max(1.0, abs(ref_loss[r]))
    assert set(got[0]["grads"]) == set(WATCH)
    for n in WATCH:
        mean = 0.5 * (ref_grads[0][n] + ref_grads[1][n])
        assert float(mean.abs().max()) > 0, n
        for r in range(2):
            torch.testing.assert_close(got[r]["grads"][n], mean, rtol=2e-4, atol=1e-7 + 2e-4 * float(mean.abs().max()))
        torch.testing.assert_close(got[0]["grads"][n], got[1]["grads"][n], rtol=0, atol=0)  # identical on both ranks
        if mode.startswith("flat_bucket"):  # ... and the SGD update it drove is that mean, on both ranks
            for r in range(2):
                torch.testing.assert_close(got[r]["update"][n], mean, rtol=2e-3, atol=1e-6 + 2e-3 * float(mean.abs().max()))


def test_dense_grad_strides_reviews_a_channels_last_strided_1x1_weight_gradient():
    """parallel.dense_grad_strides: the hook hands DDP a gradient with the parameter's strides (MIOpen returns the
    1 x 1 convolutions' weight gradients with channels-last strides; same bytes), without copying."""
    import torch
    from morefusion_amd import parallel
    conv = parallel.dense_grad_strides(torch.nn.Conv2d(8, 4, 1))
    g = torch.randn(4, 8, 1, 1).as_strided((4, 8, 1, 1), (8, 1, 8, 8))
    (hook,) = conv.weight._backward_hooks.values()
    out = hook(g)
    assert out.stride() == conv.weight.stride() and out.data_ptr() == g.data_ptr() and torch.equal(out, g)
    assert hook(out) is out
    assert not getattr(torch.nn.Conv2d(8, 4, 3).weight, "_backward_hooks", None)  # (only 1 x 1 kernels get one)
    conv(torch.randn(2, 8, 3, 3)).sum().backward()
    assert conv.weight.grad.stride() == conv.weight.stride()


def test_data_parallel_step_guards():
    """parallel.DataParallelStep refuses what its flat fp32 bucket cannot hold, and a capture after an eager exchange
    (the watchdog race) unless the caller asks for the timing-based form."""
    import torch
    from morefusion_amd.parallel import DataParallelStep
    lin = torch.nn.Linear(4, 3)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    dp = DataParallelStep(lin.parameters(), opt, lambda x: lin(x).sum(), exchange=False, chunks=8)
    assert len(dp.chunk_flat) == 2 and dp.exchanges == 0  # (never more chunks than parameters)
    w0 = lin.weight.detach().clone()
    dp.step(dict(x=torch.ones(2, 4)))
    torch.testing.assert_close(lin.weight, w0 - 0.1 * 2.0 * torch.ones(3, 4))
    assert dp.replay(dict(x=torch.ones(2, 4))) is not None  # not captured: replay is the eager step
    half = torch.nn.Linear(4, 3).to(torch.bfloat16)
    with pytest.raises(TypeError):
        DataParallelStep(half.parameters(), opt, lambda x: half(x).sum(), exchange=False)
    dp.exchanges = 1
    with pytest.raises(RuntimeError):
        dp.capture_before_exchange(dict(x=torch.ones(2, 4)), None)
    with pytest.raises(RuntimeError):
        dp.capture(dict(x=torch.ones(2, 4)), None)
