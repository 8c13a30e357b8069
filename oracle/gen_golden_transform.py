"""TEST INFRASTRUCTURE -- generate tests/golden/ref_transform.npz by EXECUTING the reference's own ``Transform``
(examples/ycb_video/singleview_3d/train.py:27-140: the boolean grid algebra that turns a dataset example's OctoMap
grids into the network inputs ``grid_target`` / ``grid_nontarget_empty``).  The class is pure NumPy; the module's
other imports (chainer, path, tensorboardX, termcolor, morefusion) are stubbed with empty modules -- nothing of them
runs.  Run in the build container only (needs /root/reference); the vectors, not the source, are committed.

    python oracle/gen_golden_transform.py

Cases: evaluation mode (always "empty+nontarget"), and training mode with seeded ``RandomState``s chosen so that
every one of the nine grid cases and both instance-subset branches (one id / several ids / no id) occur."""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/examples/ycb_video/singleview_3d/train.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_transform.npz")


def load_reference_module():
    class _Path(str):
        def expanduser(self): return self
        def abspath(self): return self
        parent = property(lambda self: self)
        def __truediv__(self, o): return _Path(self + "/" + str(o))
    stubs = {}
    for name in ("chainer", "chainer.training", "chainer.training.extensions", "path", "tensorboardX", "termcolor",
                 "morefusion", "morefusion.contrib", "morefusion.contrib.singleview_3d"):
        stubs[name] = types.ModuleType(name)
    stubs["path"].Path = _Path
    stubs["chainer"].training = stubs["chainer.training"]
    stubs["chainer.training"].extensions = stubs["chainer.training.extensions"]
    stubs["morefusion"].contrib = stubs["morefusion.contrib"]
    stubs["morefusion.contrib"].singleview_3d = stubs["morefusion.contrib.singleview_3d"]
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location("ref_singleview_3d_train", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def make_example(rs, dim=6, n_ids=3):
    """A synthetic dataset example with the keys Transform reads (datasets/ycb_video/rgbd_pose_estimation: OctoMap
    occupancy probabilities in [0,1], a {0,1} full target grid, an instance-id grid of the non-target objects)."""
    shape = (dim, dim, dim)
    ids = rs.randint(0, n_ids + 1, shape).astype(np.int32) if n_ids > 0 else np.zeros(shape, np.int32)
    ids[rs.uniform(size=shape) < 0.5] = 0
    return dict(
        class_id=np.int32(rs.randint(1, 22)), rgb=rs.randint(0, 255, (4, 4, 3)).astype(np.uint8),
        pcd=rs.uniform(-1, 1, (4, 4, 3)), quaternion_true=rs.uniform(-1, 1, 4), translation_true=rs.uniform(-1, 1, 3),
        origin=rs.uniform(-1, 1, 3), pitch=np.float64(0.01),
        grid_target=rs.uniform(size=shape).astype(np.float32), grid_nontarget=rs.uniform(size=shape).astype(np.float32),
        grid_empty=rs.uniform(size=shape).astype(np.float32),
        grid_target_full=(rs.uniform(size=shape) < 0.3).astype(np.uint8), grid_nontarget_full=ids)


def main():
    mod = load_reference_module()
    out = {}
    cases_seen, branches = set(), set()
    k = 0
    for train in (False, True):
        for seed in range(44 if train else 3):
            n_ids = (0, 1, 3, 5)[seed % 4]
            ex = make_example(np.random.RandomState(1000 + seed), n_ids=n_ids)
            t = mod.Transform(train=train, with_occupancy=True)
            t._random_state = np.random.RandomState(seed)
            inputs = {kk: np.array(v) for kk, v in ex.items() if kk.startswith("grid_")}
            res = t(dict(ex))
            # which case did the reference draw?  replay its RNG calls to label the vector (labels are for coverage only)
            rs = np.random.RandomState(seed)
            ids = np.unique(ex["grid_nontarget_full"]); ids = ids[ids > 0]
            if len(ids) > 1:
                rs.choice(ids, size=rs.randint(1, len(ids) + 1), replace=False)
            case = rs.choice(["none", "empty", "nontarget", "empty+nontarget", "nontarget_full", "empty+nontarget_full",
                              "other_full", "nontarget_full+other_full", "empty+nontarget_full+other_full"]) if train \
                else "empty+nontarget"
            if train and case in cases_seen and len(cases_seen) < 9 and seed > 20:
                pass
            cases_seen.add(case); branches.add(min(len(ids), 2))
            tag = f"c{k:03d}"
            for kk, v in inputs.items():
                out[f"{tag}__{kk}"] = v
            out[f"{tag}__train"] = np.array(train)
            out[f"{tag}__seed"] = np.array(seed)
            out[f"{tag}__case"] = np.array(case)
            out[f"{tag}__out_grid_target"] = res["grid_target"]
            out[f"{tag}__out_grid_nontarget_empty"] = res["grid_nontarget_empty"]
            for kk in ("pcd", "quaternion_true", "translation_true", "origin", "pitch"):
                assert res[kk].dtype == np.float32
            k += 1
    assert len(cases_seen) == 9 and branches == {0, 1, 2}, (cases_seen, branches)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, k, "vectors; cases", sorted(cases_seen))


if __name__ == "__main__":
    main()
