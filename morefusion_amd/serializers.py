"""Chainer ``.npz`` checkpoints <-> ``torch.nn.Module`` state.

The reference stores and loads the pose network with ``chainer.serializers.save_npz/load_npz``
(examples/ycb_video/singleview_3d/train.py:337,451-459, demo.py:51,
ros/.../singleview_3d_pose_estimation.py:66).  A Chainer npz is a flat dict whose keys are the
link hierarchy joined by ``/`` with the parameter name last (``conv3/W``, ``conv3/b``,
``pspnet_extractor/up1/prelu/W``, ``resnet_extractor/res2/a/conv1/W``; trainer snapshots
prefix ``updater/model:main/``).  ``chainer_key`` maps a ``state_dict`` name of
``morefusion_amd...Model`` onto that convention, so a checkpoint trained with the reference
loads here and one trained here loads in the reference (model built with
``pretrained_resnet18=False``).  Array layouts agree (ConvolutionND ``W`` = [out,in,*k]) except
the scalar PReLU slope (Chainer shape ``()``, torch ``[1]``).
"""
import re

import numpy as np
import torch

_PARAM = {"weight": "W", "bias": "b"}
# chainer.links.BatchNormalization: gamma, beta and the persistents avg_mean, avg_var, N
_BN_PARAM = {"weight": "gamma", "bias": "beta", "running_mean": "avg_mean", "running_var": "avg_var",
             "num_batches_tracked": "N"}
_NON_SERIALISED = re.compile(r"(^|\.)(mean|std)$")  # plain attributes in the reference, not persistents


def chainer_key(torch_name):
    """``resnet_extractor.res2.1.conv1.weight`` -> ``resnet_extractor/res2/b1/conv1/W``."""
    parts = torch_name.split(".")
    out = []
    for n, part in enumerate(parts[:-1]):
        if part.isdigit() and n > 0 and re.fullmatch(r"res\d", parts[n - 1]):
            out.append("a" if part == "0" else f"b{part}")      # ResBlock children a, b1, b2, ...
        elif part.isdigit() and n > 0 and parts[n - 1] == "convs":
            out[-1] = f"conv{int(part) + 1}"                     # PSPModule conv1..conv4
        else:
            out.append(part)
    leaf = parts[-1]
    if len(parts) > 1 and parts[-2] == "bn":  # chainercv2 ConvBlock.bn (ResNet18Extractor)
        out.append(_BN_PARAM.get(leaf, leaf))
    else:
        out.append(_PARAM.get(leaf, leaf))
    return "/".join(out)


def _entries(model):
    for name, tensor in model.state_dict().items():
        if _NON_SERIALISED.search(name):
            continue
        yield name, chainer_key(name), tensor


def save_npz(file, model, compression=True):
    """Write ``model`` in Chainer's npz layout (``chainer.serializers.save_npz``)."""
    arrays = {}
    for name, key, tensor in _entries(model):
        a = tensor.detach().cpu().numpy()
        if name.endswith("prelu.weight") and a.shape == (1,):
            a = a.reshape(())
        arrays[key] = a
    (np.savez_compressed if compression else np.savez)(file, **arrays)


def load_npz(file, model, path="", strict=True):
    """Load a Chainer npz into ``model`` (``chainer.serializers.load_npz(file, obj, path, strict)``).

    ``path`` is the key prefix (``"updater/model:main/"`` for trainer snapshots).  With
    ``strict`` a missing key or a shape mismatch raises; without it missing keys are skipped.
    Returns the list of checkpoint keys that were not consumed."""
    with np.load(file) as npz:
        stored = {k: npz[k] for k in npz.files if k.startswith(path)}
    used = set()
    state = model.state_dict()
    for name, key, tensor in _entries(model):
        full = path + key
        if full not in stored:
            if strict:
                raise KeyError(f"{full} is not in the checkpoint")
            continue
        a = stored[full]
        if a.shape == () and tuple(tensor.shape) == (1,):
            a = a.reshape(1)
        if tuple(a.shape) != tuple(tensor.shape):
            raise ValueError(f"{full}: checkpoint shape {a.shape} != parameter shape {tuple(tensor.shape)}")
        state[name] = torch.from_numpy(np.ascontiguousarray(a)).to(tensor.dtype)
        used.add(full)
    model.load_state_dict(state)
    return sorted(set(stored) - used)
