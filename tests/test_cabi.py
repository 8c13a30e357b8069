"""The C-ABI library loads here (no GPU) and exports exactly what include/mfhip.h
declares; ops refuse CPU tensors instead of falling back.  No compute calls."""
import ctypes
import os
import re

import pytest
import torch

import morefusion_amd as mf
from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "mfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = mf._lib.lib()  # dlopen works without a GPU
    declared = _declared()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mfhip.h but not exported"
    assert sorted(mf._lib.EXPORTED_SYMBOLS) == declared  # python binding covers the whole ABI
    assert lib.mf_version() >= 100
    assert isinstance(lib.mf_last_error_string(), bytes)


def test_icc_batch_struct_matches_header():
    text = open(os.path.join(ROOT, "include", "mfhip.h")).read()
    body = text[text.index("typedef struct {"):text.index("} mfIccBatch;")]
    fields = re.findall(r"\b(\w+);", body)
    assert fields == [f[0] for f in mf._lib.IccBatch._fields_]
    assert ctypes.sizeof(mf._lib.IccBatch) == 8 * 8 + 5 * 4 + 2 * 4 + 4 + 4 + 4  # (+ flags, + tail padding to 8)


def test_workspace_size_is_host_only_arithmetic():
    def desc(n_objects, n_scenes, n_points, max_ns, thr=2.0):
        return mf._lib.IccBatch(None, None, None, None, None, None, None, None, n_objects, n_scenes,
                                n_points, 32, max_ns, thr, 0.0, 1)
    ws = mf._lib.lib().mf_icc_workspace_bytes
    n = ws(ctypes.byref(desc(8, 1, 28000, 8)))
    one = desc(8, 1, 28000, 8)
    one.flags = 1  # reserved since round 6 (round 5's opt-in one-launch iteration is gone): a set bit is refused
    assert ws(ctypes.byref(one)) < 0
    # winners of 16 grids + compact bins: per grid 68 bins of max(64, P_g / 8) records + an overflow list of 2 P_g,
    # summed over the grids (sum_g P_g = Ns * P_scene) -- O(N * sum P), not nbins x that (round 2: 68 x 8 x 28000 x 16 B)
    sumP = 8 * 28000
    assert n >= 2 * 8 * 32 ** 3 * 8 + (68 * sumP // 8 + 2 * sumP) * 16
    assert n < 2 * 8 * 32 ** 3 * 8 + 0.25 * 68 * sumP * 16
    assert ws(ctypes.byref(desc(64, 8, 8 * 28000, 8))) > n
    assert ws(ctypes.byref(desc(8, 1, 28000, 8, thr=4.0))) > n       # kernel size 5: two more planes
    assert ws(ctypes.byref(desc(40, 1, 28000, 40))) > n              # 33..64 objects in a scene: allowed since round 3
    assert ws(ctypes.byref(desc(96, 1, 28000, 96))) > n              # 65..128 objects: the single-pass path (round 6)
    assert ws(ctypes.byref(desc(130, 1, 28000, 130))) < 0            # > 128 objects in a scene
    general = desc(70, 1, 28000, 70)
    general.grid_ne_binary = 0                                       # any no-entry values: the two-kernel path stays at 64
    assert ws(ctypes.byref(general)) < 0
    assert ws(ctypes.byref(desc(8, 1, 28000, 8, thr=9.0))) < 0       # kernel size > 7


def test_ops_refuse_cpu_tensors_loudly():
    v = torch.zeros(4, 2)
    p = torch.zeros(4, 3)
    b = torch.zeros(4, dtype=torch.int32)
    # (round 5: average_voxelization_3d and occupancy_grid_3d dispatch NumPy arrays / CPU tensors to the product's own
    # CPU path, like the reference's get_array_module -- tests/test_cpu_path.py; the GPU-only ops still refuse)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mf.functions.interpolate_voxel_grid(torch.zeros(1, 1, 2, 2, 2), p, b)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mf.functions.truncated_distance_function(p, pitch=1.0, origin=(0, 0, 0), dims=(4, 4, 4), truncation=2.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mf.geometry.nn(p, p)
    assert not mf.functions.occupancy_grid_3d(p, pitch=1.0, origin=(0, 0, 0), dims=(4, 4, 4)).is_cuda


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(mf._lib, "_lib", None)
    monkeypatch.setattr(mf._lib, "SO_PATH", str(tmp_path / "libmfhip.so"))
    with pytest.raises(RuntimeError, match="not built"):
        mf._lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "morefusion_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("CPU oracle", "").replace("oracle order", "").replace("oracle)", ""), os.path.join(dp, f)


def test_wgrad_split_cost_model_is_a_host_function_with_sane_answers():
    """mf_wgrad_split (csrc/gemm_bf16.hip): no device call -- the slab count of a weight-gradient GEMM from the number
    of 128 x 128 output tiles, 64-row K-tiles and the slab size.  conv3 at the training batch (160 tiles, 1024
    K-tiles) fills exactly one round of the chip's 512 workgroup slots with 3 slabs (4 would spill a quarter-full
    second round); conv4 (512 tiles) needs none; tiny layers split deep but keep >= 8 K-tiles per slab."""
    import ctypes
    from morefusion_amd import _lib
    L = _lib.lib()
    L.mf_wgrad_split.argtypes = [ctypes.c_int64] * 3
    L.mf_wgrad_split.restype = ctypes.c_int32
    assert L.mf_wgrad_split(160, 1024, 256 * 10240 * 4) == 3
    assert L.mf_wgrad_split(512, 128, 512 * 16384 * 4) == 1
    assert L.mf_wgrad_split(120, 250, 1920 * 984 * 4) == 4
    for tiles, ktiles, slab in ((1, 250, 8192), (2, 8192, 6912), (7, 3, 100), (0, 0, 0), (4096, 1, 1 << 20)):
        s = L.mf_wgrad_split(tiles, ktiles, slab)
        assert 1 <= s <= 512 and (s == 1 or ktiles // s >= 8), (tiles, ktiles, s)


def test_bf16_engines_reject_operands_beyond_the_32_bit_byte_offsets():
    """the bf16 GEMM engines address operands with 32-bit BYTE offsets from a 2^31-byte buffer
    resource (an offset >= 2^31 is the masked value and reads zeros), so a tensor of 2^30 .. 2^31 ELEMENTS must be
    refused -- the old guard let it through and convolved the tail of the batch as zeros.  Host-side argument checks
    only (no launch)."""
    from morefusion_amd import _lib
    L = _lib.lib()
    # conv3 at D = 32, Cin = 160: B = 205 is the first batch beyond 2^30 input elements (204 fits)
    assert 205 * 32 ** 3 * 160 >= 1 << 30 > 204 * 32 ** 3 * 160
    assert L.mf_conv3d_bf16_fwd(None, None, None, None, 205, 160, 256, 32, 4, 2, 1, 1, 1, 0, 256, None) < 0
    assert b"2^30" in L.mf_last_error_string()
    assert L.mf_linear_bf16(None, 0, 1024, None, 0, 1024, None, 0, None, 0, 1024, 1 << 20, 1024, 1024, 1, 0, 0, 0, None) < 0
    assert b"2^31 bytes" in L.mf_last_error_string()
    assert L.mf_linear_wgrad_bf16(None, 0, 1024, None, 0, 1024, None, 0, 1024, None, 1 << 20, 1024, 1024, 1, 1, None) < 0
