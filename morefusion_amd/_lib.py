"""ctypes binding of libmfhip.so (include/mfhip.h).

The HIP library is THE implementation of the voxel / refinement ops: there is no
Python or CPU fallback.  If the shared object is missing (or a tensor is not on a
HIP device) the ops raise -- loudly -- instead of computing something else.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MF_LIBMFHIP=<file name in this directory>: a differently built copy of the same library (tools/stamps_*.py use
# libmfhip_dbg.so = `make ICC_DEBUG=1 OUT=../libmfhip_dbg.so OBJDIR=_obj_dbg`: per-phase time stamps in the ICC kernels)
SO_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("MF_LIBMFHIP", "libmfhip.so")))
_lib = None

_p = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_d = ctypes.c_double


class IccBatch(ctypes.Structure):
    """mfIccBatch (include/mfhip.h)."""

    _fields_ = [
        ("pts4", _p), ("obj_off", _p), ("scene_off", _p), ("obj_scene", _p),
        ("pitch", _p), ("origin", _p), ("grid_target", _p), ("grid_ne", _p),
        ("n_objects", ctypes.c_int32), ("n_scenes", ctypes.c_int32),
        ("n_points", ctypes.c_int32), ("dim", ctypes.c_int32),
        ("max_scene_objects", ctypes.c_int32),
        ("voxel_threshold", _f), ("sdf_offset", _f), ("grid_ne_binary", ctypes.c_int32), ("flags", ctypes.c_int32),
    ]


_SIGNATURES = {
    "mf_version": ([], _i),
    "mf_last_error_string": ([], ctypes.c_char_p),
    "mf_average_voxelization_3d_fwd": ([_p, _p, _p, _i64, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p], _i),
    "mf_average_voxelization_3d_bwd": ([_p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p, _p], _i),
    "mf_max_voxelization_3d_fwd": ([_p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p], _i),
    "mf_max_voxelization_3d_bwd": ([_p, _p, _i64, _i, _i, _i, _i, _i, _p, _p], _i),
    "mf_interpolate_voxel_grid_fwd": ([_p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _i, _p], _i),
    "mf_interpolate_voxel_grid_bwd": ([_p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _i, _p], _i),
    "mf_occupancy_grid_3d_fwd": ([_p, _i64, _f, _f, _f, _f, _i, _i, _i, _f, _p, _p, _p], _i),
    "mf_occupancy_grid_3d_bwd": ([_p, _p, _i64, _f, _f, _f, _f, _i, _i, _i, _f, _p, _p, _p], _i),
    "mf_truncated_distance_function_fwd": ([_p, _i64, _f, _f, _f, _f, _i, _i, _i, _f, _p, _p, _p], _i),
    "mf_truncated_distance_function_bwd": ([_p, _p, _p, _i64, _f, _f, _f, _f, _i, _i, _i, _f, _p, _p], _i),
    "mf_pseudo_occupancy_weights": ([_p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p, _p, _p, _p], _i),
    "mf_nn": ([_p, _i64, _p, _i64, _p, _p, _p], _i),
    "mf_icp_loss_grad": ([_p, _i64, _p, _i64, _p, _f, _p, _p], _i),
    "mf_icp_refine": ([_p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, _f, _p, _p, _p, _p, ctypes.c_int32,
                       ctypes.c_int32, _f, _f, _p, _p, _p], _i),
    "mf_icc_workspace_bytes": ([ctypes.POINTER(IccBatch)], _i64),
    "mf_icc_iteration_launches": ([ctypes.POINTER(IccBatch)], _i),
    "mf_icc_launch_stage": ([ctypes.POINTER(IccBatch), _p, _p, _p, ctypes.c_int32, _p], _i),
    "mf_icc_prepare": ([ctypes.POINTER(IccBatch), _p, _p], _i),
    "mf_icc_loss_grad": ([ctypes.POINTER(IccBatch), _p, _p, _p, _p, _p, _p, _p], _i),
    "mf_icc_refine": ([ctypes.POINTER(IccBatch), _p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, _f, _f, _p, _p, _p, _p], _i),
    "mf_icc_debug_stamps": ([_p, _i], _i),
    "mf_sparse_conv3d_workspace_bytes": ([ctypes.c_int32] * 5 + [_i64], _i64),
    "mf_sparse_conv3d_k4s2_points_fwd": ([_p, _p, _p, _i64, _f, _f, _f, _f, _p, _p, _p, _p, _p]
                                         + [ctypes.c_int32] * 6 + [_p], _i),
    "mf_sparse_conv3d_pack_weights": ([_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_sparse_conv3d_k4s2_fwd": ([_p, _p, _p, _p, _p, _p, _p] + [ctypes.c_int32] * 6 + [_p], _i),
    "mf_pack_points_sdf": ([_p, _p, _i64, _p, _p], _i),
    "mf_average_distance_fwd": ([_p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p, _p], _i),
    "mf_average_distance_bwd": ([_p, _p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p, _p], _i),
    "mf_conv3d_k4s2_pack_weights": ([_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_conv3d_k4s2_default_split": ([ctypes.c_int32] * 4, ctypes.c_int32),
    "mf_conv3d_k4s2_workspace_bytes": ([ctypes.c_int32] * 4, _i64),
    "mf_conv3d_k4s2_fwd": ([_p, _p, _p, _p, _p, _p] + [ctypes.c_int32] * 6 + [_p], _i),
    "mf_to_channels_last": ([_p, _p, ctypes.c_int32, ctypes.c_int32, _i64, _p], _i),
    "mf_sparse_conv3d_k4s2_points_cl_fwd": ([_p, _i64, _p, _p, _i64, _f, _f, _f, _f, _p, _p, _p, _p, _p] + [ctypes.c_int32] * 6 + [_p], _i),
    "mf_interpolate_voxel_grid_cl_fwd": ([_p, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _i64, _p], _i),
    "mf_occupancy_convs_fwd": ([_p] * 7 + [ctypes.c_int32] * 2 + [_p], _i),
    "mf_linear_fwd": ([_p, _i64, ctypes.c_int32, _p, _i64, ctypes.c_int32, _p, _i64, _p, _i64] + [ctypes.c_int32] * 7 + [_p], _i),
    "mf_cast_rows_bf16": ([_p, _i64, _p, _i64, _i64, ctypes.c_int32, _p], _i),
    "mf_relu_mask_bf16": ([_p, _p, _p, _p, _i64, _p], _i),
    "mf_linear_bf16": ([_p, _i64, ctypes.c_int32, _p, _i64, ctypes.c_int32, _p, _i64, _p, _i64] + [ctypes.c_int32] * 8 + [_p], _i),
    "mf_linear_wgrad_bf16": ([_p, _i64, ctypes.c_int32, _p, _i64, ctypes.c_int32, _p, _i64, ctypes.c_int32, _p]
                             + [ctypes.c_int32] * 5 + [_p], _i),
    "mf_conv3d_k4s2_pack_bf16": ([_p] + [ctypes.c_int32] * 4 + [_p, _p, _p], _i),
    "mf_conv3d_k4s2_bf16_fwd": ([_p, _p, _p, _p] + [ctypes.c_int32] * 6 + [_p], _i),
    "mf_conv3d_k4s2_bf16_dgrad": ([_p, _p, _p] + [ctypes.c_int32] * 6 + [_p], _i),
    "mf_conv3d_k4s2_bf16_wgrad_workspace_bytes": ([ctypes.c_int32] * 3, _i64),
    "mf_conv3d_k4s2_bf16_wgrad_default_split": ([ctypes.c_int32] * 4, ctypes.c_int32),
    "mf_conv3d_k4s2_bf16_wgrad": ([_p, _p, _p, _p] + [ctypes.c_int32] * 7 + [_p], _i),
    "mf_conv3d_bf16_pack": ([_p] + [ctypes.c_int32] * 5 + [_p, _p, _p, _p], _i),
    "mf_conv3d_bf16_fwd": ([_p, _p, _p, _p] + [ctypes.c_int32] * 11 + [_p], _i),
    "mf_conv3d_bf16_fwd_workspace_bytes": ([ctypes.c_int32] * 8, _i64),
    "mf_conv3d_bf16_fwd_ws": ([_p, _p, _p, _p, _p, _i64] + [ctypes.c_int32] * 11 + [_p], _i),
    "mf_conv3d_k3_narrow_bf16_pack_elems": ([ctypes.c_int32], _i64),
    "mf_conv3d_k3_narrow_bf16_pack": ([_p] + [ctypes.c_int32] * 5 + [_p, _p], _i),
    "mf_conv3d_k3_narrow_bf16": ([_p, _p, _p, _p] + [ctypes.c_int32] * 6 + [_p], _i),
    "mf_conv3d_bf16_wgrad_workspace_bytes": ([ctypes.c_int32] * 4, _i64),
    "mf_wgrad_split": ([_i64, _i64, _i64], _i),
    "mf_linear_wgrad_bf16_default_split": ([_i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32], ctypes.c_int32),
    "mf_conv3d_bf16_wgrad_default_split": ([ctypes.c_int32] * 5, ctypes.c_int32),
    "mf_conv3d_bf16_wgrad": ([_p, _p, _p, _p] + [ctypes.c_int32] * 11 + [_p], _i),
    "mf_sparse_conv3_bf16_max_rows": ([_i64], _i64),
    "mf_sparse_conv3_bf16_workspace_bytes": ([_i64, ctypes.c_int32, ctypes.c_int32], _i64),
    "mf_sparse_conv3_bf16_tables": ([_p, _i64, ctypes.c_int32, ctypes.c_int32, _p], _i),
    "mf_sparse_conv3_bf16_index": ([_p, _p, _i64, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_sparse_conv3_bf16_pack": ([_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p, _p], _i),
    "mf_sparse_conv3_bf16_unpack_dw": ([_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_sparse_conv3_bf16_reduce": ([_p, _p, _p, _p, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_sparse_conv3_bf16_gather_dy": ([_p, _p, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_linear_bf16_tiles": ([_p, ctypes.c_int32, _p, _i64, ctypes.c_int32, _p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p], _i),
    "mf_linear_wgrad_bf16_ranges": ([_p, ctypes.c_int32, _p, ctypes.c_int32, _p, _i64, ctypes.c_int32, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p], _i),
    "mf_conv3d_k4s2_bf16_pack_cols": ([_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_conv3d_k4s2_bf16_col2im": ([_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_average_voxelization_rows_bf16_fwd": ([_p, _i64, _p, _p, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p, _p, _p, _p, _i64, _p], _i),
    "mf_average_voxelization_rows_bf16_bwd": ([_p, _i64, _p, _p, _p, _p, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _i64, _p], _i),
    "mf_average_voxelization_cl_bf16_fwd": ([_p, _i64, _p, _p, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _i64,
                                             _p, _p, _p, _p], _i),
    "mf_average_voxelization_cl_bf16_bwd": ([_p, _i64, _p, _p, _p, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p,
                                             _i64, _p], _i),
    "mf_interpolate_voxel_grid_cl_bf16_fwd": ([_p, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _i64, _p], _i),
    "mf_interpolate_voxel_grid_cl_bf16_bwd": ([_p, _i64, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _i, _p], _i),
    "mf_upsample_bilinear_cl_fwd": ([_p, _p] + [ctypes.c_int32] * 7 + [_p], _i),
    "mf_upsample_bilinear_cl_bwd": ([_p, _p] + [ctypes.c_int32] * 7 + [_p], _i),
    "mf_upsample_bilinear_cf_fwd": ([_p, _p, _i64] + [ctypes.c_int32] * 5 + [_p], _i),
    "mf_upsample_bilinear_cf_bwd": ([_p, _p, _i64] + [ctypes.c_int32] * 5 + [_p], _i),
    "mf_prelu_fwd": ([_p, _p, _p, _i64, ctypes.c_int32, _p], _i),
    "mf_rgb_normalize": ([_p, ctypes.c_int32, _p, _p, _p, _i64, _p], _i),
    "mf_bn_act_fwd": ([_p, _p, _p, _p, _p, _p, ctypes.c_float, _p, _i64, ctypes.c_int32, _i64, ctypes.c_int32,
                       ctypes.c_int32, ctypes.c_int32, _p], _i),
    "mf_prelu_bwd_workspace_floats": ([_i64], _i64),
    "mf_prelu_bwd": ([_p, _p, _p, _p, _p, _p, _i64, ctypes.c_int32, _p], _i),
    "mf_gemm_bf16_last_tile": ([], _i),
    "mf_psp_tail_rows_bf16_fwd": ([_p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p], _i),
    "mf_psp_tail_rows_bf16_bwd": ([_p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p, _p], _i),
    "mf_confidence_loss_fwd": ([_p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _p, _p, _p], _i),
    "mf_confidence_loss_bwd": ([_p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _p, _p, _p], _i),
    "mf_pose_epilogue_train_fwd": ([_p, _p, _p, _p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p, _p, _p], _i),
    "mf_pose_epilogue_train_bwd": ([_p, _p, _p, _p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p, _p, _p], _i),
    "mf_transformation_matrix_fwd": ([_p, _p, _i64, _p, _p], _i),
    "mf_transformation_matrix_bwd": ([_p, _p, _i64, _p, _p, _p], _i),
    "mf_point_prep": ([_p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _f, _p, _p, _p, _p, _p], _i),
    "mf_pose_epilogue": ([_p, _i64, ctypes.c_int32, _p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p, _p, _p, _p], _i),
    "mf_psp_tail_fwd": ([_p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p] + [ctypes.c_int32] * 4 + [_p, _p], _i),
    "mf_valid_pixel_order": ([_p, ctypes.c_int32, ctypes.c_int32, _p, _p, _p], _i),
    "mf_instance_stats": ([_p, _p, _i, _i, _p, _i, _p, _p], _i),
    "mf_instance_crops": ([_p, _p, _p, _i, _i, _d, _d, _d, _d, _p, _p, _i, _i, _i, _p, _p, _p, _p], _i),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """Load libmfhip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "morefusion_amd/csrc`). morefusion_amd has no CPU fallback."
            )
        handle = ctypes.CDLL(SO_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is absent
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().mf_last_error_string().decode()
        raise RuntimeError(f"{what} failed ({code}): {msg}")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    """Every tensor on the GPU, and on the CURRENT device: the kernels are launched on the current
    device's stream (``stream_ptr``), so a tensor of another GPU would be dereferenced by the wrong
    device (one process may drive several GPUs: select the device with ``torch.cuda.device``)."""
    cur = None
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected torch.Tensor, got {type(t)}")
        if not t.is_cuda:
            raise RuntimeError(
                "morefusion_amd voxel/refinement ops run on the MI355X only: got a "
                f"{t.device} tensor (there is no CPU fallback; move inputs to 'cuda')."
            )
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(
                f"tensor on {t.device} but the current device is cuda:{cur}: run the op under "
                f"`with torch.cuda.device({t.device.index}):` (kernels launch on the current device's stream)")


def ptr(t):
    return None if t is None else t.data_ptr()


def f32c(t):
    return t.detach().to(torch.float32).contiguous()


def i32c(t):
    return t.detach().to(torch.int32).contiguous()


def as_float3(origin):
    if isinstance(origin, torch.Tensor):
        origin = origin.detach().cpu().tolist()
    o = [float(x) for x in origin]
    if len(o) != 3:
        raise ValueError("origin must have 3 elements")
    return o
