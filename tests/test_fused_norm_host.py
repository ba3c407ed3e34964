"""CPU: the host logic and the arithmetic of the conv + BatchNorm + leaky_relu node (vgtk/so3conv/functional.py TrainEpilogue; kernels
csrc/bn_act.hip bn_act_bwd_reduce_fromy_kernel, csrc/so3_dense.hip dense_split_kernel<true> / dense_untranspose_kernel) against torch's own
BatchNorm2d + leaky_relu and its autograd (`feat = self.norm(x.feats); feat = self.relu(feat)`, SPConvNets/utils/base_so3poseconv.py:L214-221):
  * TrainEpilogue.moments: scale / shift from pivoted sums, running statistics as nn.BatchNorm2d updates them;
  * the backward FROM THE OUTPUT: pre-activation and xhat recovered from y' (also with a negative gamma), gx = k1 g - k2 - k3 xhat, the row
    bound |k1| max|g| + |k2| + |k3| max|xhat| really bounds |gx|.
The GPU tests compare the kernels with the separate modules; this file pins the formulas those kernels implement."""
import pytest
import torch
import torch.nn.functional as F


def _case(seed=3, b=3, c=5, p=7, a=4):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(b, c, p, a, generator=gen, dtype=torch.float64) * 2.0 + torch.randn(1, c, 1, 1, generator=gen, dtype=torch.float64)
    gamma = torch.rand(c, generator=gen, dtype=torch.float64) + 0.5
    gamma[1] = -0.8
    beta = torch.randn(c, generator=gen, dtype=torch.float64) * 0.3
    return x, gamma, beta


def test_moments_give_the_batchnorm_affine_map_and_running_statistics():
    import vgtk.so3conv as sptk
    from vgtk.so3conv.functional import TrainEpilogue
    x, gamma, beta = _case()
    b, c = x.shape[:2]
    norm = sptk.BatchNormLeakyReLU(c, negative_slope=0.01)
    ref = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        for m in (norm, ref):
            m.weight.copy_(gamma.float()); m.bias.copy_(beta.float())
            m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
        ref.running_mean.copy_(norm.running_mean); ref.running_var.copy_(norm.running_var)
    pivot = x[0, :, 0, 0].clone()
    d = x - pivot[None, :, None, None]
    ep = TrainEpilogue(norm)
    scale, shift, slope = ep.moments(d.sum((0, 2, 3)), (d * d).sum((0, 2, 3)), pivot, b * x.shape[2] * x.shape[3])
    ref.train()
    want = F.leaky_relu(ref(x.float()), 0.01)
    got = F.leaky_relu(x.float() * scale[None, :, None, None] + shift[None, :, None, None], slope)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    assert torch.allclose(norm.running_mean, ref.running_mean, rtol=1e-6, atol=1e-6) and torch.allclose(norm.running_var, ref.running_var, rtol=1e-6, atol=1e-6)
    k1, beta_s, inv_gamma, total = ep.saved
    assert total == b * x.shape[2] * x.shape[3] and torch.allclose(beta_s.double(), beta, atol=1e-6)
    assert torch.allclose(inv_gamma.double() * gamma, torch.ones(c, dtype=torch.float64), atol=1e-6)


@pytest.mark.parametrize('slope', [0.01, 0.2])
def test_backward_from_the_output_equals_autograd(slope):
    x, gamma, beta = _case(seed=9)
    x = x.requires_grad_(True)
    g_ = gamma.clone().requires_grad_(True)
    b_ = beta.clone().requires_grad_(True)
    y = F.leaky_relu(F.batch_norm(x, None, None, g_, b_, True, 0.0, 1e-5), slope)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    gx_ref, gg_ref, gb_ref = torch.autograd.grad(y, [x, g_, b_], gy)
    # what the node does, from y' alone
    with torch.no_grad():
        yv = y.detach()
        n = x.numel() // x.shape[1]
        mean = x.mean((0, 2, 3)); var = x.var((0, 2, 3), unbiased=False)
        invstd = torch.rsqrt(var + 1e-5)
        k1 = gamma * invstd
        bc = lambda t: t[None, :, None, None]
        pos = yv > 0
        pre = torch.where(pos, yv, yv / slope)
        xhat = (pre - bc(beta)) / bc(gamma)
        g = torch.where(pos, gy, gy * slope)
        sg, sgx = g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))
        k2, k3 = k1 * sg / n, k1 * sgx / n
        gx = bc(k1) * g - bc(k2) - bc(k3) * xhat
        assert torch.allclose(xhat, (x.detach() - bc(mean)) * bc(invstd), atol=1e-9)           # the pre-activation IS recoverable (negative gamma too)
        assert torch.allclose(gx, gx_ref, atol=1e-10) and torch.allclose(sgx, gg_ref, atol=1e-9) and torch.allclose(sg, gb_ref, atol=1e-9)
        # the row bound of the stored-operand split (per cloud, channel, anchor over the points)
        gmax, xmax = g.abs().amax(2), xhat.abs().amax(2)                                   # [b,c,a]
        bound = k1.abs()[None, :, None] * gmax + k2.abs()[None, :, None] + k3.abs()[None, :, None] * xmax
        assert bool((gx.abs().amax(2) <= bound * (1 + 1e-12)).all())
