"""SURVEY 8(f4) on the MI355X: a Chainer-layout ``.npz`` checkpoint -> ``serializers.load_npz`` into a
FRESH model -> ``Model.predict`` on the GPU reproduces the golden outputs of the reference's own network
code (tests/golden/ref_predict.npz), and the evaluation extension's summary from GPU poses.

Reference flow: examples/ycb_video/singleview_3d/evaluate.py:32-44 (build the model from the logged
arguments, ``chainer.serializers.load_npz(args.model, model)``, ``model.to_gpu()``), train.py:440-461
(snapshot of ``model`` every evaluation), training/extensions/pose_estimation_evaluator.py:112-142."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402
from morefusion_amd import serializers  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model, PitchTableModels  # noqa: E402

KEYS = ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")


def _golden_batch():
    g = golden("ref_predict.npz")
    b = mf.synthetic.make_singleview_batch(int(g["batch_size"]), seed=int(g["seed"]))
    return g, {k: torch.as_tensor(b[k]).cuda() for k in KEYS}


def _check(outs, g, tag, tol):
    q, t, c = (x.detach().cpu().numpy() for x in outs)
    np.testing.assert_allclose(q, g[f"{tag}__quaternion"], rtol=0, atol=tol)
    np.testing.assert_allclose(c, g[f"{tag}__confidence"], rtol=0, atol=tol)
    np.testing.assert_allclose(t, g[f"{tag}__translation"], rtol=0, atol=tol * 0.01)
    # the north-star tolerance itself: ADD between the GPU pose and the reference network's pose of the same point
    # (the reference's most confident one per object), over a 0.1 m model cloud -- within 1e-4 m
    from oracle import oracle_np as O
    cloud = np.random.RandomState(0).uniform(-0.05, 0.05, (500, 3)).astype(np.float32)
    worst = 0.0
    for b in range(q.shape[0]):
        i = int(np.argmax(g[f"{tag}__confidence"][b]))
        Tg = O.transformation_matrix(q[b, i].astype(np.float64), t[b, i].astype(np.float64))
        Tr = O.transformation_matrix(g[f"{tag}__quaternion"][b, i].astype(np.float64),
                                     g[f"{tag}__translation"][b, i].astype(np.float64))
        worst = max(worst, float(O.metrics_average_distance(cloud, Tg, Tr)[0]))
    print(f"ADD GPU vs reference network ({tag}): {worst:.3e} m")
    assert worst <= 1e-4


def test_chainer_checkpoint_round_trip_reproduces_reference_predict(tmp_path):
    g, inp = _golden_batch()
    torch.manual_seed(int(g["weight_seed"]))
    trained = Model(n_fg_class=21, with_occupancy=True)   # the weights the golden was generated with
    path = tmp_path / "snapshot_model_best_auc_add.npz"
    serializers.save_npz(path, trained)                   # Chainer key layout (77 arrays)
    with np.load(path) as z:
        assert "conv3/W" in z.files and "resnet_extractor/res2/a/conv1/W" in z.files
        assert z["pspnet_extractor/up1/prelu/W"].shape == ()   # Chainer's scalar PReLU slope

    torch.manual_seed(12345)                              # a fresh model: different random init
    fresh = Model(n_fg_class=21, with_occupancy=True)
    assert not torch.equal(fresh.conv3.weight, trained.conv3.weight)
    unused = serializers.load_npz(path, fresh)            # evaluate.py:42
    assert unused == []
    fresh = fresh.cuda().eval()                           # evaluate.py:43-44
    with torch.no_grad():
        fresh.predict(**inp)  # MIOpen solver choice settles on the first call of a shape
        _check(fresh.predict(**inp), g, "given", 1e-3)
        inp2 = dict(inp, origin=None)
        _check(fresh.predict(**inp2), g, "median", 1e-3)


def test_trainer_snapshot_prefix_and_pretrained_resnet18_variant(tmp_path):
    """``pretrained_resnet18=True`` (train.py:50-56: chainercv2 ResNet-18 with BatchNorm persistents) and a
    trainer snapshot's ``updater/model:main/`` prefix: save -> load into a fresh model -> identical GPU outputs.
    (No reference golden exists for this variant: chainercv2 is absent offline; the round trip is exact.)"""
    _, inp = _golden_batch()
    torch.manual_seed(3)
    a = Model(n_fg_class=21, with_occupancy=True, pretrained_resnet18=True)
    for m in a.modules():  # non-trivial BatchNorm statistics, so that losing a persistent would show
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    path = tmp_path / "model.npz"
    serializers.save_npz(path, a)
    with np.load(path) as z:
        arrays = {k: z[k] for k in z.files}
    assert any(k.endswith("/bn/avg_mean") for k in arrays) and any(k.endswith("/bn/gamma") for k in arrays)
    snap = tmp_path / "snapshot_iter_100.npz"
    np.savez(snap, **{"updater/model:main/" + k: v for k, v in arrays.items()},
             **{"updater/optimizer:main/t": np.int32(100)})
    torch.manual_seed(4)
    b = Model(n_fg_class=21, with_occupancy=True, pretrained_resnet18=True)
    unused = serializers.load_npz(snap, b, path="updater/model:main/")
    assert unused == []
    for (na, ta), (nb, tb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert na == nb and torch.equal(ta, tb), na
    a, b = a.cuda().eval(), b.cuda().eval()
    with torch.no_grad():
        a.predict(**inp)
        b.predict(**inp)  # (MIOpen's solver choice settles on the first call of a shape, per module)
        for x, y in zip(a.predict(**inp), b.predict(**inp)):
            np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=0, atol=1e-5)
    with pytest.raises(KeyError):  # strict: the plain-ResNet18 checkpoint layout does not fit this variant
        g = golden("ref_predict.npz")
        torch.manual_seed(int(g["weight_seed"]))
        plain = tmp_path / "plain.npz"
        serializers.save_npz(plain, Model(n_fg_class=21, with_occupancy=True))
        serializers.load_npz(plain, Model(n_fg_class=21, with_occupancy=True, pretrained_resnet18=True))


def test_evaluator_summary_from_gpu_poses(tmp_path):
    """pose_estimation_evaluator.py:112-142 over ``Model.evaluate(per_instance=True)`` rows computed from
    poses that ``Model.predict`` produced on the GPU (random weights: the values are large, the bookkeeping
    is what is checked -- per-class regrouping, AUC, <2cm, parent means)."""
    from morefusion_amd import metrics
    from morefusion_amd.training import PoseEstimationEvaluator
    rs = np.random.RandomState(0)
    pcds = {c: rs.uniform(-0.05, 0.05, (300, 3)).astype(np.float32) for c in mf.synthetic.CLASS_PITCH}
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True, models=PitchTableModels(pcds)).cuda().eval()
    batches = []
    for seed in (31, 32, 33):
        b = mf.synthetic.make_singleview_batch(2, seed=seed)
        batches.append({k: torch.as_tensor(b[k]).cuda() for k in KEYS + ("quaternion_true", "translation_true")})
    rows = []

    def eval_func(**batch):
        true_q, true_t = batch.pop("quaternion_true"), batch.pop("translation_true")
        q, t, conf = model.predict(**batch)
        idx = conf.argmax(dim=1)
        ar = torch.arange(q.shape[0], device=q.device)
        if len(rows) == 0:   # first batch: perfect poses, so that AUC / <2cm are not all zero
            qp, tp = true_q.float(), true_t.float()
        else:
            qp, tp = q[ar, idx], t[ar, idx]
        rep = model.evaluate(class_id=batch["class_id"], quaternion_true=true_q, translation_true=true_t,
                             quaternion_pred=qp, translation_pred=tp, per_instance=True)
        rows.append(rep)
        return rep

    result = PoseEstimationEvaluator(batches, eval_func)()
    P = "validation/main/"
    per_class = {}
    for rep in rows:
        for k, v in rep.items():
            typ, cid, _ = k.split("/")
            per_class.setdefault((typ, cid), []).append(v)
    assert len(per_class) >= 3
    for (typ, cid), vals in per_class.items():
        np.testing.assert_allclose(result[f"{P}auc/{typ}/{cid}"], metrics.ycb_video_add_auc(vals, max_value=0.1))
        np.testing.assert_allclose(result[f"{P}<2cm/{typ}/{cid}"], float(np.mean(np.asarray(vals) < 0.02)))
    for typ in ("add", "add_s", "add_or_add_s"):
        cls = [result[f"{P}auc/{typ}/{cid}"] for (t_, cid) in per_class if t_ == typ]
        np.testing.assert_allclose(result[f"{P}auc/{typ}"], np.mean(cls))
        assert 0.0 <= result[f"{P}auc/{typ}"] <= 1.0
    assert result[f"{P}auc/add_s"] > 0.0   # the perfect-pose batch contributes
