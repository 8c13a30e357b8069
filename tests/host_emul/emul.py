"""TEST INFRASTRUCTURE ONLY: build csrc/*.hip for the host behind tests/host_emul/mf_common.h
(fiber emulator) and drive the C ABI with NumPy arrays as "device" memory.  The kernel TEXT
that ships in libmfhip.so is what runs here; nothing of this is imported by the product."""
import ctypes
import hashlib
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "morefusion_amd", "csrc")
HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")

_p, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class IccBatch(ctypes.Structure):
    """mfIccBatch (include/mfhip.h) -- same layout as morefusion_amd/_lib.py."""

    _fields_ = [
        ("pts4", _p), ("obj_off", _p), ("scene_off", _p), ("obj_scene", _p),
        ("pitch", _p), ("origin", _p), ("grid_target", _p), ("grid_ne", _p),
        ("n_objects", ctypes.c_int32), ("n_scenes", ctypes.c_int32),
        ("n_points", ctypes.c_int32), ("dim", ctypes.c_int32),
        ("max_scene_objects", ctypes.c_int32),
        ("voxel_threshold", _f), ("sdf_offset", _f), ("grid_ne_binary", ctypes.c_int32), ("flags", ctypes.c_int32),
    ]


def available():
    return shutil.which("g++") is not None


def build(sources, extra_flags=()):
    """Compile the given csrc/*.hip files with g++ behind the emulator shim -> ctypes library."""
    os.makedirs(BUILD, exist_ok=True)
    h = hashlib.sha1()
    for name in list(sources) + ["mf_common.h"]:
        path = os.path.join(HERE if name == "mf_common.h" else CSRC, name)
        h.update(open(path, "rb").read())
    for hname in sorted(os.listdir(CSRC)):
        if hname.endswith(".h"):
            h.update(open(os.path.join(CSRC, hname), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "mfhip.h"), "rb").read())
    h.update(" ".join(extra_flags).encode())
    tag = h.hexdigest()[:16]
    so = os.path.join(BUILD, f"libmfemul_{tag}.so")
    if not os.path.exists(so):
        work = os.path.join(BUILD, tag)
        os.makedirs(work, exist_ok=True)
        for hname in os.listdir(CSRC):  # shared device headers (the emulator's mf_common.h shadows csrc's)
            if hname.endswith(".h") and hname != "mf_common.h":
                shutil.copy(os.path.join(CSRC, hname), os.path.join(work, hname))
        shutil.copy(os.path.join(HERE, "mf_common.h"), os.path.join(work, "mf_common.h"))
        cpps = []
        for name in sources:
            dst = os.path.join(work, name.replace(".hip", ".cpp"))
            shutil.copy(os.path.join(CSRC, name), dst)
            cpps.append(dst)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w",
                        "-I", os.path.join(ROOT, "include"), *extra_flags, "-o", so, *cpps], check=True)
        shutil.rmtree(work, ignore_errors=True)
    return ctypes.CDLL(so)


def ptr(a):
    return None if a is None else a.ctypes.data


class EmulIccScenes:
    """NumPy twin of morefusion_amd.contrib.IccScenes over the emulated library."""

    def __init__(self, lib, scenes, voxel_dim=32, voxel_threshold=2, sdf_offset=0.0, single_pass=None):
        self.lib = lib
        lib.mf_icc_workspace_bytes.restype = _i64
        P = ctypes.POINTER(IccBatch)
        lib.mf_icc_workspace_bytes.argtypes = [P]
        lib.mf_icc_prepare.argtypes = [P, _p, _p]
        lib.mf_icc_loss_grad.argtypes = [P, _p, _p, _p, _p, _p, _p, _p]
        lib.mf_icc_refine.argtypes = [P, _p, _p, _p, _p, ctypes.c_int32, ctypes.c_int32, _f, _f, _p, _p, _p, _p]
        pts, sdf, obj_off, scene_off, obj_scene = [], [], [0], [0], []
        pitch, origin, gt, gne = [], [], [], []
        for s, sc in enumerate(scenes):
            n = len(sc["points"])
            for i in range(n):
                pts.append(np.asarray(sc["points"][i], np.float32))
                sdf.append(np.asarray(sc["sdf"][i], np.float32))
                obj_off.append(obj_off[-1] + pts[-1].shape[0])
                obj_scene.append(s)
            scene_off.append(scene_off[-1] + n)
            pitch.append(np.asarray(sc["pitch"], np.float32).reshape(n))
            origin.append(np.asarray(sc["origin"], np.float32).reshape(n, 3))
            gt.append(np.asarray(sc["grid_target"], np.float32).reshape(n, voxel_dim, voxel_dim, voxel_dim))
            gne.append(np.asarray(sc["grid_nontarget_empty"], np.float32).reshape(n, voxel_dim, voxel_dim, voxel_dim))
        self.n_objects, self.n_scenes, self.n_points = len(obj_scene), len(scenes), obj_off[-1]
        self.pts4 = np.ascontiguousarray(np.concatenate(
            [np.concatenate([p, s[:, None]], 1) for p, s in zip(pts, sdf)]).astype(np.float32))
        self.obj_off = np.asarray(obj_off, np.int32)
        self.scene_off = np.asarray(scene_off, np.int32)
        self.obj_scene = np.asarray(obj_scene, np.int32)
        self.pitch = np.ascontiguousarray(np.concatenate(pitch))
        self.origin = np.ascontiguousarray(np.concatenate(origin))
        self.grid_target = np.ascontiguousarray(np.concatenate(gt))
        self.grid_ne = np.ascontiguousarray(np.concatenate(gne))
        self.desc = IccBatch(
            ptr(self.pts4), ptr(self.obj_off), ptr(self.scene_off), ptr(self.obj_scene), ptr(self.pitch),
            ptr(self.origin), ptr(self.grid_target), ptr(self.grid_ne), self.n_objects, self.n_scenes,
            self.n_points, voxel_dim, max(scene_off[i + 1] - scene_off[i] for i in range(len(scenes))),
            float(voxel_threshold), float(sdf_offset),
            int(bool(np.isin(self.grid_ne, (0, 1)).all()) if single_pass is None else single_pass))
        self.desc.flags = 0
        nbytes = lib.mf_icc_workspace_bytes(ctypes.byref(self.desc))
        assert nbytes > 0
        self.ws = np.zeros(nbytes + 256, np.uint8)
        self.ws_ptr = (self.ws.ctypes.data + 255) & ~255
        rc = lib.mf_icc_prepare(ctypes.byref(self.desc), self.ws_ptr, None)
        assert rc == 0, rc

    def loss_grad(self, q, t):
        q = np.ascontiguousarray(q, np.float32)
        t = np.ascontiguousarray(t, np.float32)
        loss = np.zeros(self.n_scenes, np.float32)
        gq = np.zeros((self.n_objects, 4), np.float32)
        gt = np.zeros((self.n_objects, 3), np.float32)
        rc = self.lib.mf_icc_loss_grad(ctypes.byref(self.desc), ptr(q), ptr(t), ptr(loss), ptr(gq), ptr(gt),
                                       self.ws_ptr, None)
        assert rc == 0, rc
        return loss, gq, gt

    def refine(self, q, t, m, v, n_iter, step0=0, alpha_q=0.01, alpha_t=0.001, losses=None, traj=None):
        for x in (q, t, m, v):
            assert x.dtype == np.float32 and x.flags.c_contiguous
        rc = self.lib.mf_icc_refine(ctypes.byref(self.desc), ptr(q), ptr(t), ptr(m), ptr(v), int(n_iter),
                                    int(step0), float(alpha_q), float(alpha_t), ptr(losses), ptr(traj),
                                    self.ws_ptr, None)
        assert rc == 0, rc


def guarded(a):
    """Copy of ``a`` whose last byte sits right in front of an inaccessible page: a kernel that reads or writes
    past the end of the buffer segfaults in the emulator instead of passing by luck (what the GPU does when the
    allocation happens to end at a mapping boundary).  The array must be 16-byte sized for aligned kernels."""
    import mmap
    a = np.ascontiguousarray(a)
    page = mmap.PAGESIZE
    n = a.nbytes
    npages = (n + page - 1) // page
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mmap.restype = ctypes.c_void_p
    libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    base = libc.mmap(None, (npages + 1) * page, mmap.PROT_READ | mmap.PROT_WRITE,
                     mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    assert base not in (None, ctypes.c_void_p(-1).value)
    assert libc.mprotect(base + npages * page, page, 0) == 0  # PROT_NONE
    start = base + npages * page - n
    buf = (ctypes.c_char * n).from_address(start)
    out = np.frombuffer(buf, dtype=a.dtype).reshape(a.shape)
    out[...] = a
    return out


class GuardedTensors:
    """Every torch (CPU) tensor whose ``data_ptr()`` is handed to the emulated library lives, for the duration
    of the call, in a mapping fenced by an inaccessible page -- ``side="after"``: the storage's last byte sits
    (up to 16-byte alignment) right in front of it, ``side="before"``: its first byte right behind it.  A kernel
    that touches one byte outside ANY of its buffers -- inputs, weights, scratch, outputs, whatever the host code
    allocated -- segfaults deterministically instead of reading a neighbour allocation by luck, which is what the
    GPU's caching allocator lets it do until an allocation ends at a mapping boundary.

    How: ``torch.Tensor.data_ptr`` is patched to mirror the tensor's whole storage into such a mapping (copy in)
    and to return the address inside the mirror; the library proxy copies every mirror back after each call.
    Mirrors are cached per (storage address, size), so addresses are stable (weight-pack cache keys)."""

    def __init__(self, lib, side="after", log=None):
        import mmap
        assert side in ("after", "before")
        self.lib, self.side, self.log = lib, side, log
        self.page = mmap.PAGESIZE
        self._prot_rw = mmap.PROT_READ | mmap.PROT_WRITE
        self._flags = mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS
        self.libc = ctypes.CDLL(None, use_errno=True)
        self.libc.mmap.restype = ctypes.c_void_p
        self.libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_long]
        self.libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.mirrors, self.pending = {}, {}
        self.calls = 0

    def _mirror(self, base, n):
        key = (base, n)
        addr = self.mirrors.get(key)
        if addr is None:
            page = self.page
            npages = max((n + page - 1) // page, 1)
            m = self.libc.mmap(None, (npages + 1) * page, self._prot_rw, self._flags, -1, 0)
            assert m not in (None, ctypes.c_void_p(-1).value)
            if self.side == "after":
                assert self.libc.mprotect(m + npages * page, page, 0) == 0
                addr = (m + npages * page - n) & ~15
            else:
                assert self.libc.mprotect(m, page, 0) == 0
                addr = m + page
            self.mirrors[key] = addr
        return addr

    def __enter__(self):
        import torch
        self._orig = torch.Tensor.data_ptr
        orig, me = self._orig, self

        def data_ptr(t):
            st = t.untyped_storage()
            base, n = st.data_ptr(), st.nbytes()
            if n == 0:
                return orig(t)
            addr = me._mirror(base, n)
            ctypes.memmove(addr, base, n)
            me.pending[(base, n)] = addr
            return addr + (orig(t) - base)

        torch.Tensor.data_ptr = data_ptr
        return self

    def __exit__(self, *exc):
        import torch
        torch.Tensor.data_ptr = self._orig
        self._flush()

    def _flush(self):
        for (base, n), addr in self.pending.items():
            ctypes.memmove(base, addr, n)
        self.pending.clear()

    def __getattr__(self, name):  # library proxy
        fn = getattr(self.lib, name)

        def call(*args):
            if self.log is not None:
                self.log.write(f"{name}\n")
                self.log.flush()
            try:
                return fn(*args)
            finally:
                self.calls += 1
                self._flush()
        return call
