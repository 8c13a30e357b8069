#!/usr/bin/env python
"""The FMA fork of the CUDA-text goldens -- TEST INFRASTRUCTURE ONLY (build container).

``tests/golden/ref_cuda_*.npz`` come from the reference's CUDA kernel text compiled WITHOUT fused
multiply-add contraction (``oracle/cuda_text.py``, ``-ffp-contract=off``): the C meaning of the text.
nvcc contracts ``a*b+c`` by default (``-fmad=true``); which expressions it fuses is the compiler's choice,
so "bit-exact against the reference's kernels" is only well defined if the outputs that matter -- voxel
indices and counts, max-voxel winners, TDF arg-min ids, nearest-neighbour indices -- do not depend on it.

This script re-runs the first pass of ``oracle/gen_golden_cuda.py`` (K1-K9 + the links' forward) with the
kernel text compiled ``-ffp-contract=fast -mfma`` into a scratch directory and compares every array with the
committed golden: integer arrays element by element, float arrays by bit pattern and magnitude.  The verdict
is written to ``tests/golden/ref_cuda_fma_fork.json`` (committed; ``tests/test_oracle_golden.py`` asserts on
it).  The HIP kernels are built with ``-ffp-contract=off`` (csrc/Makefile), i.e. they implement the
un-contracted fork by construction.

Usage:  python oracle/gen_golden_cuda_fma.py      (needs /root/reference and an x86 host with FMA)
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import cuda_text  # noqa: E402

FILES = ["ref_cuda_tdf.npz", "ref_cuda_pseudo_occupancy.npz", "ref_cuda_interpolate.npz", "ref_cuda_nn.npz",
         "ref_cuda_voxelization.npz", "ref_cuda_links.npz"]


def main():
    cuda_text.FP_CONTRACT = "fast"
    from oracle import gen_golden_cuda as GC
    golden = GC.OUT
    scratch = tempfile.mkdtemp(prefix="mf_fma_fork_")
    for f in os.listdir(golden):
        if f.startswith("fixture_pose_refinement_"):
            shutil.copy(os.path.join(golden, f), scratch)
    GC.OUT = scratch
    GC.main()
    report = {"how": "oracle/gen_golden_cuda_fma.py: kernel text compiled g++ -ffp-contract=fast -mfma vs the committed "
                     "-ffp-contract=off goldens", "files": {}}
    int_flips = 0
    for f in FILES:
        a, b = np.load(os.path.join(golden, f)), np.load(os.path.join(scratch, f))
        assert sorted(a.files) == sorted(b.files), f
        rec = {}
        for k in a.files:
            x, y = a[k], b[k]
            if x.dtype.kind in "iub":
                n = int((x != y).sum())
                if n:
                    rec[k] = {"kind": "integer", "elements": int(x.size), "differ": n}
                    int_flips += n
            elif x.dtype.kind == "f":
                same = x.view(np.uint32 if x.dtype == np.float32 else np.uint64) == \
                    y.view(np.uint32 if y.dtype == np.float32 else np.uint64)
                n = int((~same).sum())
                if n:
                    d = np.abs(x.astype(np.float64) - y.astype(np.float64))
                    scale = max(float(np.abs(x).max()), 1e-30)
                    rec[k] = {"kind": "float", "elements": int(x.size), "differ_in_bits": n,
                              "max_abs_diff": float(d.max()), "max_abs_diff_over_max_abs": float(d.max() / scale)}
        report["files"][f] = rec if rec else "identical"
    report["integer_outputs_identical"] = int_flips == 0
    report["integer_elements_that_differ"] = int_flips
    with open(os.path.join(golden, "ref_cuda_fma_fork.json"), "w") as fh:
        json.dump(report, fh, indent=1, sort_keys=True)
    shutil.rmtree(scratch, ignore_errors=True)
    print(json.dumps(report, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
