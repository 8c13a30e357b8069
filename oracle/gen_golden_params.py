#!/usr/bin/env python
"""Golden list of the reference pose network's parameter paths and shapes.

TEST INFRASTRUCTURE ONLY; runs only in the build container (``/root/reference`` mounted).
The reference's own link definitions are EXECUTED -- models/dense_fusion/resnet.py,
models/dense_fusion/pspnet.py and the constructor of contrib/singleview_3d/models/model.py:48-91
-- under a stand-in ``chainer`` that implements nothing but the link tree (``Chain.init_scope``,
child registration through ``__setattr__``, ``L.Convolution{1,2,3}D`` / ``L.PReLU`` parameter
shapes).  ``Chain.namedparams()`` then yields exactly the keys ``chainer.serializers.save_npz``
writes (path without the leading '/').  The result is committed as
tests/golden/ref_chainer_param_paths.json; tests/test_host_logic.py checks that
``morefusion_amd.serializers.chainer_key`` maps the torch model onto exactly this set.

Usage:  python oracle/gen_golden_params.py
"""
import contextlib
import importlib.util
import json
import os
import sys
import types

sys.dont_write_bytecode = True  # /root/reference is read-only
REF = "/root/reference/morefusion"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "ref_chainer_param_paths.json")


class Link:
    def __init__(self):
        object.__setattr__(self, "_params", {})
        object.__setattr__(self, "_children", {})
        object.__setattr__(self, "_in_scope", False)

    @contextlib.contextmanager
    def init_scope(self):
        object.__setattr__(self, "_in_scope", True)
        try:
            yield
        finally:
            object.__setattr__(self, "_in_scope", False)

    def __setattr__(self, name, value):
        if getattr(self, "_in_scope", False) and isinstance(value, Link):
            self._children[name] = value
        object.__setattr__(self, name, value)

    def namedparams(self, prefix=""):
        for k, shape in self._params.items():
            yield prefix + "/" + k, shape
        for name, child in self._children.items():
            yield from child.namedparams(prefix + "/" + name)


class Chain(Link):
    pass


def _conv(nd):
    class Conv(Link):
        def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, nobias=False,
                     dilate=1, **kw):
            super().__init__()
            k = (ksize,) * nd if isinstance(ksize, int) else tuple(ksize)
            self._params["W"] = [out_channels, in_channels, *k]  # in_channels None = lazily shaped
            if not nobias:
                self._params["b"] = [out_channels]
    return Conv


class PReLU(Link):
    def __init__(self, shape=(), init=0.25):
        super().__init__()
        self._params["W"] = list(shape)


def install_stub():
    chainer = types.ModuleType("chainer")
    chainer.Chain = Chain
    chainer.Link = Link
    L = types.ModuleType("chainer.links")
    L.Convolution1D, L.Convolution2D, L.Convolution3D = _conv(1), _conv(2), _conv(3)
    L.PReLU = PReLU
    F = types.ModuleType("chainer.functions")
    chainer.links, chainer.functions = L, F
    chainer.__path__ = []  # a package: ``from chainer.backends import cuda`` resolves
    backends = types.ModuleType("chainer.backends")
    backends.cuda = types.ModuleType("chainer.backends.cuda")
    chainer.backends = backends
    chainercv = types.ModuleType("chainercv")
    chainercv.links = types.ModuleType("chainercv.links")
    chainercv.links.PickableSequentialChain = Chain
    for name, mod in (("chainer", chainer), ("chainer.links", L), ("chainer.functions", F),
                      ("chainer.backends", backends), ("chainer.backends.cuda", backends.cuda),
                      ("chainercv", chainercv), ("chainercv.links", chainercv.links)):
        sys.modules[name] = mod


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    install_stub()
    resnet = load(os.path.join(REF, "models/dense_fusion/resnet.py"), "ref_resnet")
    pspnet = load(os.path.join(REF, "models/dense_fusion/pspnet.py"), "ref_pspnet")
    # contrib/singleview_3d/models/model.py does ``import morefusion`` and reaches the two
    # extractors through it: give it a namespace that holds exactly those two reference classes
    mf = types.ModuleType("morefusion")
    mf.models = types.SimpleNamespace(
        dense_fusion=types.SimpleNamespace(ResNet18=resnet.ResNet18, PSPNetExtractor=pspnet.PSPNetExtractor),
        ResNet18Extractor=None)
    mf.datasets = types.SimpleNamespace(YCBVideoModels=lambda: None)
    for name in ("morefusion", "trimesh", "trimesh.transformations", "numpy.random"):
        sys.modules.setdefault(name, mf if name == "morefusion" else types.ModuleType(name))
    sys.modules["morefusion"] = mf
    model_mod = load(os.path.join(REF, "contrib/singleview_3d/models/model.py"), "ref_model")
    model = model_mod.Model(n_fg_class=21, pretrained_resnet18=False, with_occupancy=True)
    params = {k.lstrip("/"): shape for k, shape in model.namedparams()}
    # lazily shaped inputs (Convolution(None, ...)): the sizes follow from the forward pass
    # (model.py:118-141): conv3 sees 128+16 voxelized + 16 occupancy channels, the heads 1024
    lazy = {"conv3/W": 160, "conv1_rot/W": 984, "conv1_trans/W": 984, "conv1_conf/W": 984}
    for k, cin in lazy.items():
        assert params[k][1] is None, k
        params[k][1] = cin
    assert all(None not in v for v in params.values())
    json.dump(dict(source="executed: models/dense_fusion/resnet.py, pspnet.py, "
                          "contrib/singleview_3d/models/model.py:48-91 (n_fg_class=21, with_occupancy=True)",
                   params=params), open(OUT, "w"), indent=1, sort_keys=True)
    print(len(params), "parameters ->", OUT)


if __name__ == "__main__":
    main()
