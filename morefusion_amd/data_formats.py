"""Data formats on either side of the path.

* Sparse voxel-grid wire format of the ROS pipeline (``VoxelGrid.msg``: flat ``indices`` =
  ``i*Y*Z + j*Z + k``, ``values``, ``dims``, ``pitch``, ``origin``), decoded at
  ros/src/morefusion_ros/nodes/collision_based_pose_refinement.py:86-98 and
  singleview_3d_pose_estimation.py:131-148.  ``decode_voxel_grid`` scatters on whatever device
  the index tensor lives on (so recorded grids can go straight to HBM); ``encode_voxel_grid``
  is the inverse.
* The boolean grid algebra that turns a dataset example's OctoMap probability grids into the
  network inputs ``grid_target`` / ``grid_nontarget_empty``
  (examples/ycb_video/singleview_3d/train.py:46-129, class ``Transform``).
"""
import numpy as np
import torch

GRID_CASES = (
    "none", "empty", "nontarget", "empty+nontarget", "nontarget_full", "empty+nontarget_full",
    "other_full", "nontarget_full+other_full", "empty+nontarget_full+other_full",
)


def decode_voxel_grid(indices, values, dims, dtype=torch.float32):
    """flat indices [n] (+ values [n] or None for a boolean grid) -> dense [X,Y,Z]."""
    indices = torch.as_tensor(indices).long()
    X, Y, Z = (int(d) for d in dims)
    if indices.numel() and (int(indices.min()) < 0 or int(indices.max()) >= X * Y * Z):
        raise ValueError("voxel index outside the grid")
    if values is None:
        grid = torch.zeros(X * Y * Z, dtype=torch.bool, device=indices.device)
        grid[indices] = True
    else:
        grid = torch.zeros(X * Y * Z, dtype=dtype, device=indices.device)
        grid[indices] = torch.as_tensor(values, dtype=dtype, device=indices.device)
    return grid.reshape(X, Y, Z)


def encode_voxel_grid(matrix):
    """dense [X,Y,Z] -> (flat indices of the non-zero voxels, their values, dims)."""
    matrix = torch.as_tensor(matrix)
    flat = matrix.reshape(-1)
    indices = torch.nonzero(flat, as_tuple=False)[:, 0]
    return indices, flat[indices], tuple(matrix.shape)


def grids_for_network(grid_target, grid_nontarget, grid_empty, grid_target_full=None,
                      grid_nontarget_full=None, *, train=False, random_state=None):
    """OctoMap probability grids -> (grid_target, grid_nontarget_empty) booleans.

    Evaluation (``train=False``) always uses the "empty+nontarget" case; training draws one of
    ``GRID_CASES`` and a random subset of the non-target instance ids, with the same sequence
    of ``random_state`` calls as the reference (NumPy global state by default)."""
    rs = np.random.mtrand._rand if random_state is None else random_state
    target = np.asarray(grid_target) > 0.5
    nontarget = (np.asarray(grid_nontarget) > 0.5) ^ target
    empty = (np.asarray(grid_empty) > 0.5) ^ target
    if grid_target_full is None:
        if train:
            raise ValueError("training needs grid_target_full / grid_nontarget_full")
        return target, nontarget | empty
    target_full = np.asarray(grid_target_full)
    if not np.isin(target_full, [0, 1]).all():
        raise ValueError("grid_target_full must be a {0,1} grid")
    target_full = target_full.astype(bool)
    ids_grid = np.asarray(grid_nontarget_full)
    ids = np.unique(ids_grid)
    ids = ids[ids > 0]
    if len(ids) > 0:
        if len(ids) > 1:
            ids = rs.choice(ids, size=rs.randint(1, len(ids) + 1), replace=False)
        nontarget_full = np.isin(ids_grid, ids)
    else:
        nontarget_full = np.zeros_like(target)
    nontarget_full = nontarget_full ^ target_full
    case = rs.choice(GRID_CASES) if train else "empty+nontarget"
    if case == "none":
        nte = np.zeros_like(target)
    elif case == "empty+nontarget_full+other_full":
        nte = ~target_full
    elif case == "empty":
        nte = empty
    elif case == "nontarget":
        nte = nontarget
    elif case == "empty+nontarget":
        nte = nontarget | empty
    elif case == "nontarget_full":
        nte = nontarget_full
    elif case == "empty+nontarget_full":
        nte = empty | nontarget_full
    else:
        other_full = ~target_full & ~nontarget_full & ~empty & ~target & ~nontarget
        nte = other_full if case == "other_full" else (nontarget_full | other_full)
    return target, nte
