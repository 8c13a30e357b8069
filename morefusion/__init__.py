"""``import morefusion`` -- the reference package's name, resolved to ``morefusion_amd``.

A reference script keeps its ``import morefusion`` line: ``morefusion.functions``,
``morefusion.geometry``, ``morefusion.contrib``, ``morefusion.metrics``, ``morefusion.extra`` are the
MI355X implementations (same names, keyword-only markers and error messages; CUDA tensors of
PyTorch-ROCm in place of CuPy arrays -- see INTEGRATION.md 5 and ``morefusion_amd.chainer_compat`` for
the few Chainer names the drivers themselves use).  Every already-importable submodule is aliased, so
``import morefusion.contrib.singleview_3d`` and ``from morefusion.functions import ...`` resolve to
the same module objects as their ``morefusion_amd`` spellings.
"""
import importlib
import pkgutil
import sys

import morefusion_amd as _impl

for _m in pkgutil.walk_packages(_impl.__path__, prefix="morefusion_amd."):
    if "csrc" not in _m.name and "libmfhip" not in _m.name.rsplit(".", 1)[-1]:  # (the C-ABI library is not a Python module)
        importlib.import_module(_m.name)
for _name, _mod in list(sys.modules.items()):
    if _name == "morefusion_amd" or _name.startswith("morefusion_amd."):
        sys.modules["morefusion" + _name[len("morefusion_amd"):]] = _mod
