"""Fused MFMA field of the light-conditioned configuration (csrc/fused_mlp_rf.hip: grid_mlp 32-64-64-16, view_mlp
47-80-80-3 over [features, SH16(view), SH16(light)], optional BARF level window) against plain PyTorch on the GPU.

Forward: fp32 torch field (f16 operand rounding through three + three layers: tolerances at each check).
Backward: (a) the kernels' arithmetic restated in torch (f16 operands, f32 accumulation) with all ReLUs active -- every
fragment / transposition / tile-to-weight map must agree to product rounding; (b) random weights against the restatement
and against fp32 autograd in the L2 sense; the gradient with respect to the un-normalised view direction against
autograd through normalise -> SH -> field."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 32), (64, 64), (16, 64), (80, 47), (80, 80), (3, 80)]


def make_weights(seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(o, i, generator=g) * scale * (2.0 / i) ** 0.5).cuda().contiguous() for o, i in SHAPES]


def sh16(d):
    """Degree-4 SH of unit vectors, differentiable (the polynomials of shencoder.cu:49-120 written out in torch)."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2)], -1)


def test_torch_sh_matches_the_kernel():
    from raw_ngp_amd import _lib
    d = torch.nn.functional.normalize(torch.randn(4000, 3, device="cuda"), dim=-1)
    out = torch.empty(4000, 16, device="cuda")
    _lib.shencoder_backend.sh_encode_forward(d.contiguous(), out, 4000, 3, 4, None)
    np.testing.assert_allclose(sh16(d).cpu().numpy(), out.cpu().numpy(), rtol=1e-5, atol=2e-6)


def torch_field(enc_bf, dirs, ldirs, W, level_w=None):
    """enc_bf [M,32] (feature 2 l + c) -> sigma, rgb, density pre-activations, colour pre-activations (fp32)."""
    f = enc_bf if level_w is None else enc_bf * level_w.repeat_interleave(2)
    h = torch.relu(f @ W[0].t())
    h = torch.relu(h @ W[1].t())
    h = h @ W[2].t()
    sigma = torch.exp(h[:, 0])
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    ld = ldirs / ldirs.norm(dim=-1, keepdim=True)
    x = torch.cat([h[:, 1:], sh16(d), sh16(ld)], -1)
    c = torch.relu(x @ W[3].t())
    c = torch.relu(c @ W[4].t())
    c = c @ W[5].t()
    return sigma, torch.clamp(torch.exp(c - 5.0), max=5.0), h, c


def window(alpha=6.3):
    """BARF weights for 16 levels (network.py:99-109) at some annealing state: ones, a cosine ramp, zeros."""
    k = torch.arange(16, dtype=torch.float32)
    w = (1 - torch.cos(np.pi * (alpha - k).clamp(0, 1))) / 2
    w[0] = 1.0
    return w.cuda()


@pytest.mark.parametrize("M", [1, 31, 32, 33, 1000, 40000])
@pytest.mark.parametrize("win", [False, True], ids=["plain", "window"])
def test_forward_matches_torch(M, win):
    from raw_ngp_amd import _lib
    mb = _lib.mlp_rf_backend
    W = make_weights()
    g = torch.Generator(device="cuda").manual_seed(M)
    stride = M + 7
    enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
    dirs = torch.randn(M, 3, device="cuda", generator=g) * 1.7
    ldirs = torch.randn(M, 3, device="cuda", generator=g) * 0.6
    lw = window() if win else None
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    sigma = torch.full((M,), -1.0, device="cuda")
    rgb = torch.full((M, 3), -1.0, device="cuda")
    mb.forward(enc, stride, dirs, ldirs, lw, None, M, image, sigma, rgb)
    enc_bf = enc[:, :M].permute(1, 0, 2).reshape(M, 32)
    rs, rc, h, c = torch_field(enc_bf, dirs, ldirs, W, lw)
    # f16 operands: ~1e-3 relative on pre-activations; exp() turns absolute pre-activation error into relative output error
    np.testing.assert_allclose(torch.log(sigma).cpu().numpy(), h[:, 0].cpu().numpy(), rtol=0, atol=2e-2)
    np.testing.assert_allclose(sigma.cpu().numpy(), rs.cpu().numpy(), rtol=3e-2, atol=1e-4)
    np.testing.assert_allclose(rgb.cpu().numpy(), rc.cpu().numpy(), rtol=4e-2, atol=1e-4)
    # density-only query (the density-grid refresh): same sigma bits, no direction tensors
    s2 = torch.full((M,), -1.0, device="cuda")
    mb.forward(enc, stride, None, None, lw, None, M, image, s2, None)
    assert torch.equal(s2, sigma)


def test_forward_reads_count_from_device():
    from raw_ngp_amd import _lib
    mb = _lib.mlp_rf_backend
    W = make_weights(1)
    M_cap, M = 5000, 1234
    enc = torch.randn(16, M_cap, 2, device="cuda") * 0.5
    dirs, ldirs = torch.randn(M_cap, 3, device="cuda"), torch.randn(M_cap, 3, device="cuda")
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    sigma = torch.full((M_cap,), -1.0, device="cuda")
    rgb = torch.full((M_cap, 3), -1.0, device="cuda")
    cnt = torch.tensor([M, 99999], dtype=torch.int32, device="cuda")
    mb.forward(enc, M_cap, dirs, ldirs, None, cnt, M_cap, image, sigma, rgb)
    assert torch.all(sigma[:M] > 0) and torch.all(sigma[M:] == -1.0) and torch.all(rgb[M:] == -1.0)


def q16(t):
    return t.half().float()


def emulated_backward(enc_bf, dirs, ldirs, W, dsigma, drgb, S, level_w=None):
    """The kernels' arithmetic in torch: every MFMA operand rounded to f16, fp32 accumulation, ReLU masks from the rounded
    activations, deltas carried with the loss scale S.  Returns d(enc) [M,32], the gradient at the SH(view) inputs
    [M,16] (unscaled), and the six weight gradients."""
    Wq = [q16(w) for w in W]
    f = enc_bf if level_w is None else enc_bf * level_w.repeat_interleave(2)
    x0 = q16(f)
    h1 = q16(torch.relu(x0 @ Wq[0].t()))
    h2 = q16(torch.relu(h1 @ Wq[1].t()))
    o = h2 @ Wq[2].t()
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    ld = ldirs / ldirs.norm(dim=-1, keepdim=True)
    x3 = q16(torch.cat([o[:, 1:], sh16(d), sh16(ld)], -1))
    h3 = q16(torch.relu(x3 @ Wq[3].t()))
    h4 = q16(torch.relu(h3 @ Wq[4].t()))
    c = h4 @ Wq[5].t()
    e = torch.exp(c - 5.0)
    d6 = q16(torch.where(e <= 5.0, drgb * e * S, torch.zeros_like(e)))
    dW6 = d6.t() @ h4
    d5 = q16((d6 @ Wq[5]) * (h4 > 0))
    dW5 = d5.t() @ h3
    d4 = q16((d5 @ Wq[4]) * (h3 > 0))
    dW4 = d4.t() @ x3
    dx3 = d4 @ Wq[3]
    d3 = q16(torch.cat([(dsigma * torch.exp(o[:, 0].clamp(-80, 80)) * S)[:, None], dx3[:, :15]], -1))
    dW3 = d3.t() @ h2
    d2 = q16((d3 @ Wq[2]) * (h2 > 0))
    dW2 = d2.t() @ h1
    d1 = q16((d2 @ Wq[1]) * (h1 > 0))
    dW1 = d1.t() @ x0
    dx0 = d1 @ Wq[0]
    if level_w is not None:
        dx0 = dx0 * level_w.repeat_interleave(2)
    return dx0 / S, dx3[:, 15:31] / S, [g / S for g in (dW1, dW2, dW3, dW4, dW5, dW6)]


def dirs_gradient(dirs, g_sh):
    """d / d dirs of <g_sh, SH16(dirs / |dirs|)> by autograd."""
    d = dirs.clone().requires_grad_(True)
    (sh16(d / d.norm(dim=-1, keepdim=True)) * g_sh).sum().backward()
    return d.grad


def always_active_weights(seed=5):
    """Weights for which no hidden pre-activation is near 0 (see test_gpu_fused_mlp.py): positive encoder inputs and
    positive W1 / W2 / W5, W4 dominated by its (positive) SH_0 column of the view direction."""
    g = torch.Generator().manual_seed(seed)
    w1 = torch.rand(64, 32, generator=g) * 0.1 + 0.01
    w2 = torch.rand(64, 64, generator=g) * 0.05 + 0.005
    w3 = torch.randn(16, 64, generator=g) * 0.1
    w4 = torch.randn(80, 47, generator=g) * 0.02
    w4[:, 15] = 20.0 + torch.rand(80, generator=g)
    w5 = torch.rand(80, 80, generator=g) * 0.02 + 0.002
    w6 = torch.randn(3, 80, generator=g) * 0.05
    return [w.cuda().contiguous() for w in (w1, w2, w3, w4, w5, w6)]


def run_backward(W, enc, stride, dirs, ldirs, lw, dsigma, drgb, M, S=1024.0, want_ddirs=True):
    from raw_ngp_amd import _lib
    mb = _lib.mlp_rf_backend
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    denc = torch.full((16, stride, 2), 7.0, device="cuda")
    ddirs = torch.full((M, 3), 7.0, device="cuda") if want_ddirs else None
    dws = [torch.full_like(w, 7.0) for w in W]
    mb.backward(enc, stride, dirs, ldirs, lw, dsigma, drgb, None, M, image, S, denc, ddirs, dws)
    assert torch.all(denc[:, M:] == 7.0)
    return denc[:, :M].permute(1, 0, 2).reshape(M, 32), ddirs, dws


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("M", [1, 33, 1000, 30000])
@pytest.mark.parametrize("win", [False, True], ids=["plain", "window"])
def test_backward_index_maps_exact(M, win):
    W = always_active_weights()
    g = torch.Generator(device="cuda").manual_seed(300 + M)
    stride = M + 5
    enc = torch.rand(16, stride, 2, device="cuda", generator=g) * 0.5 + 0.05
    dirs = torch.randn(M, 3, device="cuda", generator=g)
    ldirs = torch.randn(M, 3, device="cuda", generator=g)
    dsigma = torch.randn(M, device="cuda", generator=g) * 1e-3
    drgb = torch.randn(M, 3, device="cuda", generator=g) * 1e-3
    lw = window(9.4) if win else None
    got_denc, got_dd, dws = run_backward(W, enc, stride, dirs, ldirs, lw, dsigma, drgb, M)
    enc_bf = enc[:, :M].permute(1, 0, 2).reshape(M, 32).contiguous()
    e_denc, e_gsh, e_dws = emulated_backward(enc_bf, dirs, ldirs, W, dsigma, drgb, 1024.0, lw)
    e_dd = dirs_gradient(dirs, e_gsh)
    checks = [("denc", got_denc, e_denc), ("ddirs", got_dd, e_dd)] + [(f"dW{k + 1}", dws[k], e_dws[k]) for k in range(6)]
    for name, a, b in checks:
        l2, mx = rel(a, b)
        assert l2 < 2e-3 and mx < 1e-2, (name, l2, mx)
    if win:     # levels whose window weight is 0 receive no gradient at all
        assert torch.all(got_denc.view(M, 16, 2)[:, lw == 0] == 0)


@pytest.mark.parametrize("M", [33, 1000, 30000])
def test_backward_random_weights(M):
    W = [w.requires_grad_(True) for w in make_weights(2)]
    g = torch.Generator(device="cuda").manual_seed(100 + M)
    stride = M + 5
    enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
    dirs = torch.randn(M, 3, device="cuda", generator=g)
    ldirs = torch.randn(M, 3, device="cuda", generator=g)
    dsigma = torch.randn(M, device="cuda", generator=g) * 1e-3
    drgb = torch.randn(M, 3, device="cuda", generator=g) * 1e-3
    dsigma[::7] = 0.0
    if M > 64:                                  # a whole tile behind the compositor's early stop: skipped in all kernels
        dsigma[32:64] = 0.0
        drgb[32:64] = 0.0
    lw = window(11.7)
    Wd = [w.detach() for w in W]
    got_denc, got_dd, dws = run_backward(Wd, enc, stride, dirs, ldirs, lw, dsigma, drgb, M)
    enc_bf = enc[:, :M].permute(1, 0, 2).reshape(M, 32).contiguous()
    with torch.no_grad():
        e_denc, e_gsh, e_dws = emulated_backward(enc_bf, dirs, ldirs, Wd, dsigma, drgb, 1024.0, lw)
    e_dd = dirs_gradient(dirs, e_gsh)
    for name, a, b in [("denc", got_denc, e_denc), ("ddirs", got_dd, e_dd)] + [(f"dW{k + 1}", dws[k], e_dws[k])
                                                                                for k in range(6)]:
        l2, mx = rel(a, b)
        assert l2 < 3e-2, (name, l2, mx)
    if M > 64:
        assert torch.all(got_denc[32:64] == 0) and torch.all(got_dd[32:64] == 0)
    # plain fp32 autograd of the same field
    enc_ref = enc_bf.clone().requires_grad_(True)
    d_ref = dirs.clone().requires_grad_(True)
    rs, rc, _, _ = torch_field(enc_ref, d_ref, ldirs, W, lw)
    ((rs * dsigma).sum() + (rc * drgb).sum()).backward()
    # (every f16 rounding upstream can flip a ReLU whose pre-activation is within rounding of 0; d dirs sits behind the
    # longest chain of them and carries the largest share of such flips)
    for name, a, b in [("denc", got_denc, enc_ref.grad), ("ddirs", got_dd, d_ref.grad)] + [(f"dW{k + 1}", dws[k], W[k].grad)
                                                                                          for k in range(6)]:
        l2, mx = rel(a, b)
        assert l2 < (1e-1 if name == "ddirs" else 6e-2), (name, l2, mx)


def test_backward_saturates_instead_of_overflowing():
    """A delta beyond the f16 range is clipped to +-65504 (an inf would turn into NaN weights): gradients stay finite."""
    W = make_weights(4)
    M = 257
    enc = torch.randn(16, M, 2, device="cuda") * 0.5
    dirs, ldirs = torch.randn(M, 3, device="cuda"), torch.randn(M, 3, device="cuda")
    dsigma = torch.full((M,), 1e3, device="cuda")         # x 1024 x exp(raw): far beyond 65504
    drgb = torch.full((M, 3), 1e3, device="cuda")
    denc, dd, dws = run_backward(W, enc, M, dirs, ldirs, None, dsigma, drgb, M)
    for t in [denc, dd] + dws:
        assert torch.isfinite(t).all()


def test_backward_is_deterministic():
    W = make_weights(3)
    M = 5000
    enc = torch.randn(16, M, 2, device="cuda") * 0.5
    dirs, ldirs = torch.randn(M, 3, device="cuda"), torch.randn(M, 3, device="cuda")
    dsigma = torch.randn(M, device="cuda") * 1e-3
    drgb = torch.randn(M, 3, device="cuda") * 1e-3
    a = run_backward(W, enc, M, dirs, ldirs, None, dsigma, drgb, M)
    b = run_backward(W, enc, M, dirs, ldirs, None, dsigma, drgb, M)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
