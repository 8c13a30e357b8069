"""Needed-set PSPNet decoder (opt-in) on the MI355X: same network outputs
as the dense decoder.  Kept in its own, last-collected file: the feature is not yet part of the
measured default path."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model  # noqa: E402


def test_sparse_decoder_predict_matches_dense_decoder():
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
    batch = mf.synthetic.make_singleview_batch(4, seed=3)
    inputs = {k: torch.as_tensor(batch[k]).cuda()
              for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
    with torch.no_grad():
        model.sparse_pspnet_decoder = False
        model.predict(**inputs)  # first call of these shapes: MIOpen settles its solver choice
        q0, t0, c0 = model.predict(**inputs)
        model.sparse_pspnet_decoder = True
        q1, t1, c1 = model.predict(**inputs)
    assert torch.isfinite(q1).all() and torch.isfinite(t1).all() and torch.isfinite(c1).all()
    # identical mathematics, different summation order (GEMM on gathered windows vs MIOpen conv)
    torch.testing.assert_close(q1, q0, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(t1, t0, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(c1, c0, rtol=2e-3, atol=2e-3)
