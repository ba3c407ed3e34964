"""GPU: the layers the dense product (csrc/so3_dense.hip) takes BY DEFAULT, against the reference's own numbers
(tests/golden/dense_*.npz, made by running the reference: tests/golden/make_golden_dense.py).  Closes the one place where the
default kernel of the deep layers was pinned through oracle slabs alone: here out / dF / dW are the reference's autograd and the
test asserts the regime the call took.  Bars as the other golden layers (tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_golden import DENSE_CASES, dense_grad_out  # noqa: E402
from test_gpu_parity import _run_layer, rel_err, dev, vg  # noqa: E402,F401


@pytest.mark.parametrize('name', DENSE_CASES)
def test_dense_layer_golden(dev, vg, golden, name):
    _, _, _, L = vg
    assert L.DENSE_MODE == 'auto'                     # the default decision, nothing forced
    g = golden(name + '.npz')
    L.BACKWARD_LOG, L.FORWARD_LOG = [], []
    try:
        conv, feats, (inter_idx, inter_w, sample_idx, y) = _run_layer(vg, dev, g)
        gf, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], dense_grad_out(g).to(dev))
        log, flog = L.BACKWARD_LOG, L.FORWARD_LOG
    finally:
        L.BACKWARD_LOG = L.FORWARD_LOG = None
    o = g['W'].shape[0]
    assert [r['regime'] for r in log] == ['dense rows'], log
    if name == 'dense_parts_o256':
        assert log[0].get('parts') == 2, log
    # the forward takes the product too (at 128-row widths since round 6: L.DENSE_FWD_NARROW)
    assert [r['dense'] for r in flog] == [o % 256 == 0 or L.DENSE_FWD_NARROW], flog
    out = y.feats.detach().cpu().numpy()
    assert rel_err(out[:, ::16], g['out_channels16']) < 1e-5
    assert rel_err(out[:, :, ::32], g['out_points32']) < 1e-5
    assert rel_err(gf.cpu().numpy(), g['grad_feats']) < 1e-5
    assert rel_err(gW.cpu().numpy(), g['grad_W']) < 2e-5


def test_dense_layer_golden_matches_the_list_kernels_bit_for_bit_in_structure(dev, vg, golden, monkeypatch):
    """The same fixture with the dense product switched off: the list kernels meet the same bars (the fixture is not tuned to one
    kernel), and the neighbour lists are the reference's."""
    _, _, _, L = vg
    import vgtk.cuda.grouping as G
    g = golden('dense_identity_o256.npz')
    xyz = torch.from_numpy(g['xyz']).to(dev)
    assert np.array_equal(G.ball_query(xyz, xyz, float(g['radius']), int(g['nn'])).cpu().numpy(), g['ball_idx'])
    monkeypatch.setattr(L, 'DENSE_MODE', 'off')
    L.BACKWARD_LOG = []
    try:
        conv, feats, (_, _, _, y) = _run_layer(vg, dev, g)
        gf, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], dense_grad_out(g).to(dev))
        log = L.BACKWARD_LOG
    finally:
        L.BACKWARD_LOG = None
    assert log[0]['regime'] != 'dense rows'
    assert rel_err(y.feats.detach().cpu().numpy()[:, ::16], g['out_channels16']) < 1e-5
    assert rel_err(gf.cpu().numpy(), g['grad_feats']) < 1e-5 and rel_err(gW.cpu().numpy(), g['grad_W']) < 2e-5
