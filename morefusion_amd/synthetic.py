"""Synthetic inputs of the reference's shapes.

No dataset, CAD model or pretrained weight is reachable offline, so tests,
``bench.py`` and ``__graft_entry__.smoke()`` synthesise their inputs here:

* the example-dict schema of ``datasets/rgbd_pose_estimation/base.py:164-175``
  (consumed by ``examples/ycb_video/singleview_3d/train.py:35-140``), and
* ICC scenes in the argument layout of
  ``examples/ycb_video/pose_refinement/check_iterative_collision_check_link.py:22-46``
  built from the three real fixture instances the reference ships (when given)
  plus procedurally generated solid primitives.

NumPy/SciPy only (host side, runs before the timed region).
"""
import numpy as np

# per-class voxel pitch = bbox diagonal / 32
# (ros/src/morefusion_ros/include/morefusion_ros/utils/data.h:12-32)
CLASS_PITCH = {
    1: 0.006296589104319322, 2: 0.008705823111730123, 3: 0.006425726070431774,
    4: 0.004375644727606043, 5: 0.007023497839423789, 6: 0.003923674166124662,
    7: 0.006018916012848706, 8: 0.004320481778555272, 9: 0.004535342826373148,
    10: 0.006631487204390293, 11: 0.009982031658204186, 12: 0.008721623259758258,
    13: 0.007331656585392745, 14: 0.005318687227615036, 15: 0.008406278399464109,
    16: 0.0079006960844688, 17: 0.00699458097945295, 18: 0.0038783371057780278,
    19: 0.006648125743278138, 20: 0.008405508709996566, 21: 0.0033429720217908734,
}
# morefusion/datasets/ycb_video/class_names.py:32-46 (bowl, wood block, clamps, brick)
CLASS_IDS_SYMMETRIC = (13, 16, 19, 20, 21)


def random_rotation(rs, max_angle=np.pi):
    axis = rs.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = rs.uniform(-max_angle, max_angle)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def synthetic_sdf(points):
    """Stand-in for ``YCBVideoModels.get_sdf`` (datasets/ycb_video/models.py:56-79,
    positive inside) for a solid point set whose true SDF is not shipped: distance
    to the nearest boundary sample, boundary = points with an incomplete
    26-neighbourhood on the set's own lattice."""
    from scipy.spatial import cKDTree

    pts = np.asarray(points, dtype=np.float64)
    tree = cKDTree(pts)
    d, _ = tree.query(pts, k=2)
    spacing = np.median(d[:, 1])
    cnt = np.array([len(x) for x in tree.query_ball_point(pts, spacing * 1.8)])
    boundary = cnt < 0.85 * np.percentile(cnt, 90)
    if not boundary.any():
        boundary[:] = True
    db, _ = cKDTree(pts[boundary]).query(pts)
    return db.astype(np.float32)


def make_primitive(kind, pitch, rs):
    """Solid primitive sampled on a lattice of spacing ``pitch`` (like the
    down-sampled solid voxel centres ``get_sdf`` returns), with its analytic
    signed distance (positive inside).  Bounding-box diagonal ~ 32*pitch."""
    diag = 32 * pitch
    if kind == "box":
        half = np.array([0.30, 0.22, 0.16]) * diag * rs.uniform(0.9, 1.1, 3)
    elif kind == "cylinder":
        half = np.array([0.2, 0.2, 0.33]) * diag * rs.uniform(0.9, 1.1)
    else:  # sphere
        half = np.full(3, 0.26 * diag * rs.uniform(0.9, 1.1))
    n = np.ceil(half / pitch).astype(int) + 1
    ax = [np.arange(-k, k + 1) * pitch for k in n]
    g = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    if kind == "box":
        q = np.abs(g) - half
        sd = -(np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0))
    elif kind == "cylinder":
        dr = np.hypot(g[:, 0], g[:, 1]) - half[0]
        dz = np.abs(g[:, 2]) - half[2]
        q = np.stack([dr, dz], 1)
        sd = -(np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0))
    else:
        sd = half[0] - np.linalg.norm(g, axis=1)
    keep = sd >= -0.25 * pitch
    return g[keep].astype(np.float32), sd[keep].astype(np.float32)


def voxelize_bool(points_world, origin, pitch, dim=32):
    idx = np.round((points_world - origin) / pitch).astype(int)
    ok = ((idx >= 0) & (idx < dim)).all(axis=1)
    g = np.zeros((dim,) * 3, dtype=bool)
    g[tuple(idx[ok].T)] = True
    return g


def make_icc_scene(n_objects=8, seed=0, fixtures=None, dim=32):
    """ICC scene (BASELINE config 3): the given real fixture instances first, then
    synthetic primitives, each with a perturbed initial pose."""
    from scipy import ndimage

    rs = np.random.RandomState(seed)
    fixtures = list(fixtures or [])[:n_objects]
    objs = []
    for f in fixtures:
        objs.append(dict(
            class_id=int(f["class_id"]), points=f["pcd_cad"].astype(np.float32),
            sdf=synthetic_sdf(f["pcd_cad"]), pitch=np.float32(f["pitch"]),
            origin=f["origin"].astype(np.float32), grid_target=f["grid_target"].astype(np.float32),
            grid_nontarget_empty=f["grid_nontarget_empty"].astype(np.float32),
            transform_init=f["transform_init"].astype(np.float32), transform_gt=None))
    kinds = ["box", "cylinder", "sphere"]
    classes = [c for c in CLASS_PITCH if 0.0055 < CLASS_PITCH[c] < 0.0095]
    n_syn = n_objects - len(objs)
    # synthetic objects sit on a ring in front of the camera (z ~ 0.6 m), close
    # enough that neighbouring grids overlap -> the collision term is exercised
    syn = []
    for k in range(n_syn):
        cid = classes[rs.randint(len(classes))]
        pitch = np.float32(CLASS_PITCH[cid])
        pts, sdf = make_primitive(kinds[k % 3], float(pitch), rs)
        ang = 2 * np.pi * k / max(n_syn, 1) + rs.uniform(-0.2, 0.2)
        rad = 0.16 if n_syn > 1 else 0.0
        center = np.array([0.25 + rad * np.cos(ang), 0.05 + rad * np.sin(ang),
                           0.62 + rs.uniform(-0.03, 0.03)])
        T_gt = np.eye(4)
        T_gt[:3, :3] = random_rotation(rs)
        T_gt[:3, 3] = center
        syn.append(dict(class_id=cid, pitch=pitch, points=pts, sdf=sdf, T_gt=T_gt))
    occ_world = [(s["points"] @ s["T_gt"][:3, :3].T + s["T_gt"][:3, 3]) for s in syn]
    for k, s in enumerate(syn):
        pitch = s["pitch"]
        pw = occ_world[k]
        center = np.median(pw, axis=0)
        origin = (center - pitch * (dim / 2.0 - 0.5)).astype(np.float32)
        full = voxelize_bool(pw, origin, pitch, dim)
        # visible half shell: surface voxels on the camera side of the centre
        surf = pw[s["sdf"] < 1.2 * pitch]
        surf = surf[surf[:, 2] < center[2] + 2 * pitch]
        target = voxelize_bool(surf, origin, pitch, dim)
        others = np.zeros_like(full)
        for j in range(n_syn):
            if j != k:
                others |= voxelize_bool(occ_world[j], origin, pitch, dim)
        ne = ~ndimage.binary_dilation(full, iterations=2) | others
        ne &= ~target
        dT = np.eye(4)
        dT[:3, :3] = random_rotation(rs, np.deg2rad(15))
        dT[:3, 3] = rs.uniform(-0.015, 0.015, 3)
        # perturb about the object's centre so the offset stays <= 1.5 cm
        C = np.eye(4)
        C[:3, 3] = s["T_gt"][:3, 3]
        T_init = C @ dT @ np.linalg.inv(C) @ s["T_gt"]
        objs.append(dict(
            class_id=s["class_id"], points=s["points"], sdf=s["sdf"], pitch=pitch, origin=origin,
            grid_target=target.astype(np.float32), grid_nontarget_empty=ne.astype(np.float32),
            transform_init=T_init.astype(np.float32), transform_gt=s["T_gt"].astype(np.float32)))
    return dict(
        class_id=np.array([o["class_id"] for o in objs], dtype=np.int32),
        points=[o["points"] for o in objs],
        sdf=[o["sdf"] for o in objs],
        pitch=np.array([o["pitch"] for o in objs], dtype=np.float32),
        origin=np.stack([o["origin"] for o in objs]).astype(np.float32),
        grid_target=np.stack([o["grid_target"] for o in objs]),
        grid_nontarget_empty=np.stack([o["grid_nontarget_empty"] for o in objs]),
        transform_init=np.stack([o["transform_init"] for o in objs]),
        transform_gt=[o["transform_gt"] for o in objs],
    )


def make_singleview_batch(batch_size=1, seed=0, image_size=256, dim=32):
    """Example dicts for ``Model.predict`` (BASELINE config 2):
    rgb u8 [B,H,W,3] inside a centred disc mask, pcd f32 [B,H,W,3] on a noisy sphere
    cap at z ~ 0.6 m with NaN outside the mask, class ids, pitch, origin
    (median - 15.5*pitch, model.py:202-205) and a no-entry grid."""
    rs = np.random.RandomState(seed)
    H = W = image_size
    yy, xx = np.mgrid[0:H, 0:W]
    out = dict(class_id=[], rgb=[], pcd=[], pitch=[], origin=[], grid_nontarget_empty=[],
               quaternion_true=[], translation_true=[])
    classes = sorted(CLASS_PITCH)
    for b in range(batch_size):
        cid = classes[(1 + 3 * b + seed) % len(classes)]
        pitch = np.float32(CLASS_PITCH[cid])
        r_px = 50 + rs.randint(0, 6)  # ~8000 px
        mask = (yy - H / 2) ** 2 + (xx - W / 2) ** 2 < r_px ** 2
        rgb = np.zeros((H, W, 3), np.uint8)
        rgb[mask] = rs.randint(0, 256, (int(mask.sum()), 3))
        R = 0.35 * 32 * pitch
        u = (xx - W / 2) / r_px
        v = (yy - H / 2) / r_px
        rr = np.clip(1 - u ** 2 - v ** 2, 0, None)
        cz = 0.6 + 0.02 * b
        pcd = np.stack([u * R + 0.01 * b, v * R, cz - R * np.sqrt(rr)], -1)
        pcd += rs.normal(0, 0.003, pcd.shape) * 0.3
        pcd = pcd.astype(np.float32)
        pcd[~mask] = np.nan
        center = np.median(pcd[mask], axis=0)
        origin = (center - pitch * (dim / 2.0 - 0.5)).astype(np.float32)
        g = rs.uniform(size=(dim,) * 3) < 0.6
        zz = np.mgrid[0:dim, 0:dim, 0:dim]
        g &= ((zz - dim / 2 + 0.5) ** 2).sum(axis=0) > (0.38 * dim) ** 2
        out["class_id"].append(cid)
        out["rgb"].append(rgb)
        out["pcd"].append(pcd)
        out["pitch"].append(pitch)
        out["origin"].append(origin)
        out["grid_nontarget_empty"].append(g)
        q = rs.normal(size=4)
        out["quaternion_true"].append((q / np.linalg.norm(q)).astype(np.float32))
        out["translation_true"].append(np.array([0.01 * b, 0, cz], np.float32))
    return dict(
        class_id=np.array(out["class_id"], np.int32), rgb=np.stack(out["rgb"]),
        pcd=np.stack(out["pcd"]), pitch=np.array(out["pitch"], np.float32),
        origin=np.stack(out["origin"]), grid_nontarget_empty=np.stack(out["grid_nontarget_empty"]),
        quaternion_true=np.stack(out["quaternion_true"]),
        translation_true=np.stack(out["translation_true"]),
    )


def make_singleview_examples(n_examples=1, seed=0, image_size=256, dim=32):
    """Raw per-object example dicts in the dataset's schema (SURVEY A0:
    datasets/rgbd_pose_estimation/base.py:164-175 plus the ``*_full`` grids of the training
    set): class_id i32, rgb u8 [H,W,3], pcd f64 [H,W,3] (NaN outside the mask -- the dataset
    hands out float64, ``Transform`` casts), quaternion_true / translation_true f64, pitch,
    origin f64, grid_target / grid_nontarget / grid_empty f32 OctoMap PROBABILITIES in [0,1],
    grid_target_full {0,1}, grid_nontarget_full i32 instance ids.
    ``transform_example`` (examples/ycb_video/singleview_3d/train.py:27-140) turns one into the
    network's inputs."""
    batch = make_singleview_batch(n_examples, seed=seed, image_size=image_size, dim=dim)
    rs = np.random.RandomState(seed + 7919)
    zz = np.stack(np.mgrid[0:dim, 0:dim, 0:dim], -1) - (dim / 2 - 0.5)
    r = np.linalg.norm(zz, axis=-1)
    examples = []
    for b in range(n_examples):
        noise = lambda lo, hi: rs.uniform(lo, hi, (dim,) * 3).astype(np.float32)  # noqa: E731
        full = r < 0.30 * dim                                # the whole object
        seen = full & (zz[..., 2] < 0)                       # its camera-facing half, as fused so far
        other = (np.linalg.norm(zz - np.array([0.55 * dim, 0, 0]), axis=-1) < 0.25 * dim)
        other2 = (np.linalg.norm(zz + np.array([0, 0.6 * dim, 0]), axis=-1) < 0.2 * dim)
        ne = batch["grid_nontarget_empty"][b]
        ids = np.zeros((dim,) * 3, np.int32)
        ids[other] = 4
        ids[other2 & ~other] = 9
        examples.append(dict(
            class_id=np.int32(batch["class_id"][b]), rgb=batch["rgb"][b],
            pcd=batch["pcd"][b].astype(np.float64),
            quaternion_true=batch["quaternion_true"][b].astype(np.float64),
            translation_true=batch["translation_true"][b].astype(np.float64),
            pitch=np.float64(batch["pitch"][b]), origin=batch["origin"][b].astype(np.float64),
            grid_target=np.where(seen, noise(0.55, 0.97), noise(0.0, 0.45)),
            # (the mapping's non-target / empty grids are > 0.5 on the target's own voxels too:
            #  Transform removes them again with ``^ grid_target``, train.py:50-54)
            grid_nontarget=np.where((other & ~full) | seen, noise(0.6, 0.95), noise(0.0, 0.4)),
            grid_empty=np.where((ne & ~full & ~other) | seen, noise(0.55, 0.99), noise(0.0, 0.45)),
            grid_target_full=full.astype(np.uint8), grid_nontarget_full=ids))
    return examples


def transform_example(example, train=False, with_occupancy=True, random_state=None):
    """The reference's per-example ``Transform`` (train.py:27-140): dtype casts and the boolean
    grid algebra (data_formats.grids_for_network) -> the keys ``Model.predict`` / ``forward`` take."""
    from .data_formats import grids_for_network

    d = dict(example)
    assert d["class_id"].dtype == np.int32 and d["rgb"].dtype == np.uint8
    for k in ("pcd", "quaternion_true", "translation_true"):
        d[k] = np.asarray(d[k], np.float32)
    raw = {k: d.pop(k) for k in ("grid_target", "grid_nontarget", "grid_empty", "grid_target_full",
                                 "grid_nontarget_full")}
    if not with_occupancy:
        d.pop("pitch")
        d.pop("origin")
        return d
    d["origin"] = np.asarray(d["origin"], np.float32)
    d["pitch"] = np.asarray(d["pitch"], np.float32)
    d["grid_target"], d["grid_nontarget_empty"] = grids_for_network(
        raw["grid_target"], raw["grid_nontarget"], raw["grid_empty"], raw["grid_target_full"],
        raw["grid_nontarget_full"], train=train, random_state=random_state)
    return d


def make_rgbd_frame(seed=0, height=480, width=640):
    """A synthetic RGB-D frame + instance image for the pre-processing row: several instances
    whose bounding boxes exercise every branch of the crop/centerize geometry (wide, tall,
    small -> up-scaling, exactly 256 x 256, exact 2:1 reduction, too few valid points, absent).
    Returns dict(rgb u8 [H,W,3], depth f32 [H,W] with NaN holes, K [3,3] float64,
    label i32 [H,W], instance_ids i32 [n])."""
    rs = np.random.RandomState(seed)
    rgb = rs.randint(0, 256, (height, width, 3)).astype(np.uint8)
    depth = rs.uniform(0.5, 1.5, (height, width)).astype(np.float32)
    depth[rs.uniform(size=depth.shape) < 0.1] = np.nan
    label = np.zeros((height, width), np.int32)
    yy, xx = np.mgrid[:height, :width]
    label[(yy >= 10) & (yy < 110) & (xx >= 20) & (xx < 330)] = 3                      # wide box
    label[((yy - 200) / 37.0) ** 2 + ((xx - 60) / 21.0) ** 2 <= 1.0] = 5            # small ellipse
    label[(yy >= 120) & (yy < 376) & (xx >= 100) & (xx < 356)] = 7                   # 256 x 256
    label[(yy >= 170) & (yy < 470) & (xx >= 360) & (xx < 400) & ((yy + xx) % 3 > 0)] = 11  # tall, holey
    label[(yy >= 2) & (yy < 8) & (xx >= 400) & (xx < 406)] = 13                       # 36 px: skipped
    label[(yy >= 380) & (yy < 470) & (xx >= 410) & (xx < 630)] = 17
    depth[label == 17] = np.nan                                                       # no valid depth
    if height >= 480 and width >= 640:
        label[:] = np.where((yy >= 100) & (yy < 400) & (xx >= 64) & (xx < 576) & (label == 0)
                            & ((yy // 50 + xx // 64) % 2 == 0), 19, label)           # 512 x 300 -> 2:1
    K = np.array([[619.4, 0, width / 2 - 0.3], [0, 618.9, height / 2 + 0.7], [0, 0, 1]])
    ids = np.array([3, 5, 7, 11, 13, 17, 19, 23], np.int32)                            # 23 is absent
    return dict(rgb=rgb, depth=depth, K=K, label=label, instance_ids=ids)
