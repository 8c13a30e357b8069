"""Run the reference's CUDA kernel TEXT on the CPU -- TEST INFRASTRUCTURE ONLY.

The reference never ships a binary for its GPU-only ops: the CUDA C lives in Python strings
handed to ``cupy.ElementwiseKernel`` (through ``chainer.backends.cuda.elementwise``) and in one
``.cu`` file handed to ``cupy.RawKernel``.  Neither cupy nor a CUDA device exists here, but the
TEXT is plain C: this module stands in for the two cupy entry points by generating a C++
translation unit around the unmodified text, compiling it with g++ and calling it on NumPy
arrays.  Nothing of the reference is copied into the repository: the text is taken from the
reference module at run time (``oracle/gen_golden_cuda.py``, build container only) and only the
resulting input/output vectors are committed under ``tests/golden/``.

Semantics reproduced
* ``ElementwiseKernel(in_params, out_params, operation, name, preamble=...)(*args, size=None)``:
  the loop ``for i in range(size)`` over the broadcast shape of the non-``raw`` array arguments;
  ``raw`` arguments are indexable views, the others are references to element ``i`` (0-d arrays
  and Python scalars broadcast); ``T`` is bound to the dtype of the first ``T`` argument.  The
  body runs SEQUENTIALLY in increasing ``i`` -- one legal schedule of the parallel kernel.  For
  the kernels pinned here the result is schedule-independent except for arg-min ties, where the
  sequential schedule keeps the lowest ``i`` (``atomicMin`` returns the old value, ``if
  (distance < old) atomicExch(...)``): the deterministic rule the oracle and the HIP kernels use.
* ``RawKernel(code, name)(grid, block, args=..., shared_mem=...)``: the ``__global__`` function is
  compiled against ``tests/host_emul/mf_common.h`` (the fiber-based HIP emulator of the test
  suite: every thread of a block is a fiber, ``__syncthreads`` parks it) and launched block by
  block.
* Arithmetic is the C meaning of the text: IEEE float32 operations, NO fused multiply-add
  (``-ffp-contract=off``), ``round`` = half away from zero, float -> int conversion truncates.
  (nvcc contracts a*b+c into FMA by default; which expressions it fuses is a compiler decision,
  not part of the source text.)
"""
import ctypes
import hashlib
import os
import re
import subprocess

import numpy as np

_CACHE = os.environ.get("MF_CUDA_TEXT_CACHE", "/tmp/mf_cuda_text")
# "off" (default): the C meaning of the text, one rounding per operation -- what the goldens are made with.
# "fast": the OTHER legal compilation -- a*b+c contracted into fused multiply-adds wherever the compiler sees
# one (g++ -ffp-contract=fast -mfma), as nvcc does by default (-fmad=true).  oracle/gen_golden_cuda_fma.py
# regenerates every CUDA-text golden in this mode and records which outputs move.
FP_CONTRACT = os.environ.get("MF_CUDA_TEXT_CONTRACT", "off")
_HERE = os.path.dirname(os.path.abspath(__file__))
_EMUL = os.path.join(os.path.dirname(_HERE), "tests", "host_emul")

_CTYPE = {"float32": "float", "float64": "double", "int8": "signed char", "uint8": "unsigned char",
          "int16": "short", "int32": "int", "uint32": "unsigned int", "int64": "long long",
          "uint64": "unsigned long long", "bool": "bool"}

_PRELUDE = r"""
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <algorithm>
using namespace std;
#define __device__
#define __host__
#define __forceinline__ inline
template <class T> struct RawArg {
  T *p; long long n;
  T &operator[](long long i) const { return p[i]; }
  long long size() const { return n; }
};
template <class T, class V> inline T atomicMin(T *p, V v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class V> inline T atomicMax(T *p, V v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class V> inline T atomicAdd(T *p, V v) { T o = *p; *p = o + (T)v; return o; }
template <class T, class V> inline T atomicExch(T *p, V v) { T o = *p; *p = (T)v; return o; }
template <class T, class V> inline T atomicCAS(T *p, V c, V v) { T o = *p; if (o == (T)c) *p = (T)v; return o; }
struct IndexerStub { long long n; long long size() const { return n; } };
"""


def _compile(src, tag, extra=()):
    os.makedirs(_CACHE, exist_ok=True)
    # (g++ forms FMAs in its widening_mul pass, which -O1 does not run: the contracted fork is built at -O2)
    fp = ["-ffp-contract=off"] if FP_CONTRACT == "off" else [f"-ffp-contract={FP_CONTRACT}", "-mfma", "-O2"]
    h = hashlib.sha1((src + "|".join(extra) + "|".join(fp)).encode()).hexdigest()[:16]
    so = os.path.join(_CACHE, f"{tag}_{h}.so")
    if not os.path.exists(so):
        cpp = so[:-3] + ".cpp"
        with open(cpp, "w") as f:
            f.write(src)
        cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", *fp, "-w", *extra, "-o", so, cpp]
        subprocess.run(cmd, check=True)
    return ctypes.CDLL(so)


def _parse_params(text):
    out = []
    for item in [s.strip() for s in text.replace("\n", " ").split(",") if s.strip()]:
        toks = item.split()
        raw = toks[0] == "raw"
        if raw:
            toks = toks[1:]
        assert len(toks) == 2, item
        out.append((raw, toks[0], toks[1]))
    return out


class ElementwiseKernel:
    def __init__(self, in_params, out_params, operation, name="kernel", preamble="", **kwargs):
        self.params = _parse_params(in_params) + _parse_params(out_params)
        self.n_in = len(_parse_params(in_params))
        self.operation = operation
        self.name = re.sub(r"\W", "_", name)
        self.preamble = preamble

    def __call__(self, *args, size=None):
        assert len(args) == len(self.params), (len(args), len(self.params))
        arrs = []
        tmpl = {}
        for (raw, ty, nm), a in zip(self.params, args):
            a = np.asarray(a)
            if ty not in _CTYPE and ty not in tmpl:  # template type letter: first use binds it
                tmpl[ty] = a.dtype.name
            want = np.dtype(tmpl.get(ty, ty))
            if a.dtype != want:  # Python scalars / dtype-less values take the declared type
                assert a.ndim == 0, f"{nm}: {a.dtype} passed for {ty}"
                a = a.astype(want)
            arrs.append(a)
        # loop shape = broadcast of the non-raw array arguments (or size=)
        shapes = [a.shape for (raw, _, _), a in zip(self.params, arrs) if not raw and a.ndim > 0]
        if size is None:
            size = int(np.prod(np.broadcast_shapes(*shapes))) if shapes else 1
        loop_shape = np.broadcast_shapes(*shapes) if shapes else ()
        keep = []  # contiguous buffers handed to C; outputs are copied back
        decl, ptrs = [], []
        for k, ((raw, ty, nm), a) in enumerate(zip(self.params, arrs)):
            ct = _CTYPE[tmpl.get(ty, ty)]
            is_out = k >= self.n_in
            if raw:
                buf = np.ascontiguousarray(a)
                decl.append(f"  RawArg<{ct}> {nm}{{({ct} *)a[{k}], {buf.size}LL}};")
            else:
                if a.ndim == 0 or a.size == 1:
                    buf = np.ascontiguousarray(a).reshape(-1)
                    stride = 0
                else:
                    assert a.shape == tuple(loop_shape), f"{nm}: shape {a.shape} vs loop {loop_shape}"
                    buf = np.ascontiguousarray(a)
                    stride = 1
                const = "" if is_out else "const "
                decl.append(f"  {const}{ct} *{nm}__p = ({ct} *)a[{k}]; const long long {nm}__s = {stride};")
            keep.append((buf, a, is_out))
            ptrs.append(buf.ctypes.data)
        refs = []
        for k, (raw, ty, nm) in enumerate(self.params):
            if raw:
                continue
            ct = _CTYPE[tmpl.get(ty, ty)]
            const = "" if k >= self.n_in else "const "
            refs.append(f"      {const}{ct} &{nm} = {nm}__p[i * {nm}__s];")
        typedefs = "\n".join(f"typedef {_CTYPE[v]} {k};" for k, v in tmpl.items())
        src = (_PRELUDE + typedefs + "\n" + self.preamble + "\n"
               + f'extern "C" void run_{self.name}(void **a, long long n) {{\n'
               + "\n".join(decl) + "\n  IndexerStub _ind{n};\n"
               + "  for (long long i = 0; i < n; ++i) {\n    [&]() {\n"
               + "\n".join(refs) + "\n" + self.operation + "\n    }();\n  }\n}\n")
        lib = _compile(src, self.name)
        fn = getattr(lib, f"run_{self.name}")
        fn.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_longlong]
        fn.restype = None
        fn((ctypes.c_void_p * len(ptrs))(*ptrs), size)
        for buf, orig, is_out in keep:  # a non-contiguous output was computed in a copy: write it back
            if is_out and orig.size and not np.shares_memory(buf, orig):
                np.copyto(orig, buf.reshape(orig.shape))


def elementwise(in_params, out_params, operation, name, **kwargs):
    """chainer.backends.cuda.elementwise (a memoised ElementwiseKernel constructor)."""
    return ElementwiseKernel(in_params, out_params, operation, name, **kwargs)


class RawKernel:
    """cupy.RawKernel(code, name): ``kernel(grid, block, args=(...), shared_mem=...)``."""

    def __init__(self, code, name, **kwargs):
        self.code = code
        self.name = name

    def __call__(self, grid, block, args=(), shared_mem=0, **kwargs):
        # argument list of the __global__ function, from its text
        m = re.search(r"void\s+" + re.escape(self.name) + r"\s*\(([^)]*)\)", self.code, re.S)
        assert m, "kernel signature not found"
        sig = [s.strip() for s in m.group(1).replace("\n", " ").split(",")]
        unpack, call, ptrs, keep = [], [], [], []
        for k, (s, a) in enumerate(zip(sig, args)):
            ty = s.rsplit(None, 1)[0] if not s.rsplit(None, 1)[0].endswith("*") else s.rsplit(None, 1)[0]
            is_ptr = "*" in s
            if is_ptr:
                base = s.split("*")[0].strip()
                arr = np.asarray(a)
                assert arr.flags.c_contiguous
                keep.append(arr)
                ptrs.append(arr.ctypes.data)
                unpack.append(f"  {base} *p{k} = ({base} *)a[{k}];")
            else:
                base = s.rsplit(None, 1)[0]
                val = np.asarray(a).astype({"int": np.int32, "float": np.float32, "double": np.float64,
                                            "long long": np.int64}[base]).reshape(1)
                keep.append(val)
                ptrs.append(val.ctypes.data)
                unpack.append(f"  {base} p{k} = *({base} *)a[{k}];")
            call.append(f"p{k}")
        g = tuple(grid) + (1,) * (3 - len(grid))
        b = tuple(block) + (1,) * (3 - len(block))
        src = ('#include "mf_common.h"\n' + self.code + "\n"
               + f'extern "C" void run_{self.name}(void **a) {{\n' + "\n".join(unpack) + "\n"
               + f"  hipLaunchKernelGGL({self.name}, dim3({g[0]}, {g[1]}, {g[2]}), dim3({b[0]}, {b[1]}, {b[2]}), 0, 0, "
               + ", ".join(call) + ");\n}\n")
        lib = _compile(src, self.name, extra=("-I", _EMUL, "-I", os.path.join(os.path.dirname(_HERE), "include")))
        fn = getattr(lib, f"run_{self.name}")
        fn.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        fn.restype = None
        fn((ctypes.c_void_p * len(ptrs))(*ptrs))
