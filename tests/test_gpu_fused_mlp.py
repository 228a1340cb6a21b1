"""Fused MFMA tiny-MLP against a plain PyTorch fp32 reference of the same field (GPU).

The fused path computes with f16 operands / f32 accumulation (the reference's `--fp16` autocast
precision), so outputs agree with the fp32 reference to ~1e-2 relative; tolerances are written at
each check."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_weights(seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32), (64, 64), (16, 64), (64, 31), (64, 64), (3, 64)]
    return [(torch.randn(o, i, generator=g) * scale * (2.0 / i) ** 0.5).cuda().contiguous() for o, i in shapes]


def torch_field(enc_bf, dirs, W):
    """enc_bf [M,32] (feature 2l+c), dirs [M,3] -> sigma_raw-activated outputs, fp32."""
    from raw_ngp_amd import _lib
    h = torch.relu(enc_bf @ W[0].t())
    h = torch.relu(h @ W[1].t())
    h = h @ W[2].t()
    sigma = torch.exp(h[:, 0])
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    sh = torch.empty(d.shape[0], 16, device=d.device)
    _lib.shencoder_backend.sh_encode_forward(d.contiguous(), sh, d.shape[0], 3, 4, None)
    x = torch.cat([h[:, 1:], sh], -1)
    c = torch.relu(x @ W[3].t())
    c = torch.relu(c @ W[4].t())
    c = c @ W[5].t()
    return sigma, torch.clamp(torch.exp(c - 5.0), max=5.0), h, c


@pytest.mark.parametrize("M", [1, 31, 32, 33, 1000, 40000])
def test_forward_matches_torch(M):
    from raw_ngp_amd import _lib
    mb = _lib.mlp_backend
    W = make_weights()
    g = torch.Generator(device="cuda").manual_seed(M)
    stride = M + 7
    enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
    dirs = torch.randn(M, 3, device="cuda", generator=g) * 1.7
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    sigma = torch.full((M,), -1.0, device="cuda")
    rgb = torch.full((M, 3), -1.0, device="cuda")
    mb.forward(enc, stride, dirs, None, M, image, sigma, rgb)
    enc_bf = enc[:, :M].permute(1, 0, 2).reshape(M, 32)
    rs, rc, h, c = torch_field(enc_bf, dirs, W)
    # f16 operands through three layers: ~1e-3 relative on pre-activations, exp() turns absolute
    # pre-activation error into relative output error
    np.testing.assert_allclose(torch.log(sigma).cpu().numpy(), h[:, 0].cpu().numpy(), rtol=0, atol=2e-2)
    np.testing.assert_allclose(sigma.cpu().numpy(), rs.cpu().numpy(), rtol=3e-2, atol=1e-4)
    np.testing.assert_allclose(rgb.cpu().numpy(), rc.cpu().numpy(), rtol=3e-2, atol=1e-4)


@pytest.mark.parametrize("M,act", [(1, None), (33, None), (50000, None), (4097, (0, 1, 2.0, 1))])
def test_density_scatter_is_the_density_query_followed_by_the_scatter(M, act):
    """ngp_x_mlp_density_scatter == ngp_x_mlp_forward(rgb = NULL) + ngp_x_density_grid_scatter, bit for bit (the maximum per
    cell does not depend on the order of the atomics); negative cells are dropped, untouched cells keep their -1."""
    from raw_ngp_amd import _lib
    mb, eb = _lib.mlp_backend, _lib.engine_backend
    W = make_weights(3)
    g = torch.Generator(device="cuda").manual_seed(M)
    stride, cells = M + 5, 4096
    enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
    idx = torch.randint(-1, cells // 2, (M,), device="cuda", generator=g, dtype=torch.int32)   # duplicates + dropped draws
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    sigma = torch.empty(M, device="cuda")
    mb.forward(enc, stride, None, None, M, image, sigma, None, act=act)
    want = torch.full((cells,), -1.0, device="cuda")
    eb.density_grid_scatter(idx, sigma, M, want)
    got = torch.full((cells,), -1.0, device="cuda")
    mb.density_scatter(enc, stride, M, image, idx, got, act=act)
    assert torch.equal(got, want)
    assert torch.all(got[cells // 2:] == -1.0) and (M < 100 or torch.any(got[:cells // 2] >= 0))


def test_forward_reads_count_from_device():
    from raw_ngp_amd import _lib
    mb = _lib.mlp_backend
    W = make_weights(1)
    M_cap, M = 5000, 1234
    enc = torch.randn(16, M_cap, 2, device="cuda") * 0.5
    dirs = torch.randn(M_cap, 3, device="cuda")
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    sigma = torch.full((M_cap,), -1.0, device="cuda")
    rgb = torch.full((M_cap, 3), -1.0, device="cuda")
    cnt = torch.tensor([M, 99999], dtype=torch.int32, device="cuda")
    mb.forward(enc, M_cap, dirs, cnt, M_cap, image, sigma, rgb)
    assert torch.all(sigma[:M] > 0) and torch.all(sigma[M:] == -1.0) and torch.all(rgb[M:] == -1.0)
    full = torch.empty(M_cap, device="cuda")
    full_rgb = torch.empty(M_cap, 3, device="cuda")
    mb.forward(enc, M_cap, dirs, None, M_cap, image, full, full_rgb)
    assert torch.equal(full[:M], sigma[:M]) and torch.equal(full_rgb[:M], rgb[:M])


def q16(t):
    """Round to f16 and back: the precision of an MFMA operand."""
    return t.half().float()


def emulated_backward(enc_bf, dirs, W, dsigma, drgb, S):
    """The fused kernels' arithmetic restated in PyTorch: fp32 accumulation, every MFMA operand
    (weights, activations, deltas) rounded to f16, ReLU masks taken from the rounded activations,
    deltas carried with the loss scale S.  Returns d(enc) [M,32] and the six weight gradients."""
    from raw_ngp_amd import _lib
    Wq = [q16(w) for w in W]
    x0 = q16(enc_bf)
    h1 = q16(torch.relu(x0 @ Wq[0].t()))
    h2 = q16(torch.relu(h1 @ Wq[1].t()))
    o = h2 @ Wq[2].t()                                   # fp32 accumulator
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    sh = torch.empty(d.shape[0], 16, device=d.device)
    _lib.shencoder_backend.sh_encode_forward(d.contiguous(), sh, d.shape[0], 3, 4, None)
    x3 = q16(torch.cat([o[:, 1:], sh], -1))
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
    return dx0 / S, [g / S for g in (dW1, dW2, dW3, dW4, dW5, dW6)]


def always_active_weights(seed=5):
    """Weights for which no hidden pre-activation is ever near 0 (all ReLUs stay on), so that the
    comparison is not sensitive to a unit flipping under f16 rounding: positive encoder inputs and positive
    W1/W2/W5, and W4 dominated by its (positive) SH_0 = 0.282 column."""
    g = torch.Generator().manual_seed(seed)
    w1 = torch.rand(64, 32, generator=g) * 0.1 + 0.01
    w2 = torch.rand(64, 64, generator=g) * 0.05 + 0.005
    w3 = torch.randn(16, 64, generator=g) * 0.1
    w4 = torch.randn(64, 31, generator=g) * 0.02
    w4[:, 15] = 20.0 + torch.rand(64, generator=g)
    w5 = torch.rand(64, 64, generator=g) * 0.02 + 0.002
    w6 = torch.randn(3, 64, generator=g) * 0.05
    return [w.cuda().contiguous() for w in (w1, w2, w3, w4, w5, w6)]


def run_backward(W, enc, stride, dirs, dsigma, drgb, M, S=1024.0):
    from raw_ngp_amd import _lib
    mb = _lib.mlp_backend
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    denc = torch.full((16, stride, 2), 7.0, device="cuda")
    dws = [torch.full_like(w, 7.0) for w in W]
    mb.backward(enc, stride, dirs, dsigma, drgb, None, M, image, S, denc, dws)
    assert torch.all(denc[:, M:] == 7.0)
    return denc[:, :M].permute(1, 0, 2).reshape(M, 32), dws


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("M", [1, 33, 1000, 30000])
def test_backward_index_maps_exact(M):
    """All ReLUs active: every operand fragment, transposition and accumulator-to-weight map is exercised and
    must agree with the restated arithmetic to f16-product rounding (2e-3 in L2, 1e-2 worst element)."""
    W = always_active_weights()
    g = torch.Generator(device="cuda").manual_seed(300 + M)
    stride = M + 5
    enc = torch.rand(16, stride, 2, device="cuda", generator=g) * 0.5 + 0.05
    dirs = torch.randn(M, 3, device="cuda", generator=g)
    dsigma = torch.randn(M, device="cuda", generator=g) * 1e-3
    drgb = torch.randn(M, 3, device="cuda", generator=g) * 1e-3
    got_denc, dws = run_backward(W, enc, stride, dirs, dsigma, drgb, M)
    enc_bf = enc[:, :M].permute(1, 0, 2).reshape(M, 32).contiguous()
    e_denc, e_dws = emulated_backward(enc_bf, dirs, W, dsigma, drgb, 1024.0)
    for name, a, b in [("denc", got_denc, e_denc)] + [(f"dW{k + 1}", dws[k], e_dws[k]) for k in range(6)]:
        l2, mx = rel(a, b)
        assert l2 < 2e-3 and mx < 1e-2, (name, l2, mx)


@pytest.mark.parametrize("M", [33, 1000, 30000])
def test_backward_random_weights(M):
    """Random weights (ReLUs switch on and off).  f16 rounding can flip a unit whose pre-activation is within
    rounding of 0, which changes that sample's delta for the unit by O(1); tensors are therefore bounded in
    the L2 sense: 3e-2 against the restated f16 arithmetic, 5e-2 against plain fp32 autograd."""
    W = [w.requires_grad_(True) for w in make_weights(2)]
    g = torch.Generator(device="cuda").manual_seed(100 + M)
    stride = M + 5
    enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
    dirs = torch.randn(M, 3, device="cuda", generator=g)
    dsigma = torch.randn(M, device="cuda", generator=g) * 1e-3
    drgb = torch.randn(M, 3, device="cuda", generator=g) * 1e-3
    dsigma[::7] = 0.0          # early-terminated samples carry no gradient
    Wd = [w.detach() for w in W]
    got_denc, dws = run_backward(Wd, enc, stride, dirs, dsigma, drgb, M)
    enc_bf = enc[:, :M].permute(1, 0, 2).reshape(M, 32).contiguous()
    with torch.no_grad():
        e_denc, e_dws = emulated_backward(enc_bf, dirs, Wd, dsigma, drgb, 1024.0)
    for name, a, b in [("denc", got_denc, e_denc)] + [(f"dW{k + 1}", dws[k], e_dws[k]) for k in range(6)]:
        l2, mx = rel(a, b)
        assert l2 < 3e-2, (name, l2, mx)
    enc_ref = enc_bf.clone().requires_grad_(True)
    rs, rc, _, _ = torch_field(enc_ref, dirs, W)
    ((rs * dsigma).sum() + (rc * drgb).sum()).backward()
    for name, a, b in [("denc", got_denc, enc_ref.grad)] + [(f"dW{k + 1}", dws[k], W[k].grad) for k in range(6)]:
        l2, mx = rel(a, b)
        assert l2 < 5e-2, (name, l2, mx)


@pytest.mark.parametrize("M", [33, 5000])
def test_backward_direction_gradient_matches_autograd(M):
    """d loss / d (un-normalised view direction) out of the plain view kernel (ngp_x_mlp_backward_dirs: pose refinement
    without the light-conditioned field) against autograd through the SH op (its own Jacobian kernel) and the torch field,
    normalisation included; the other outputs must not change when it is requested.  Same 5e-2 L2 bound as the weight
    gradients against fp32 autograd; the gradient is tangent to the direction."""
    from raw_ngp_amd import _lib
    from raw_ngp_amd.shencoder import SHEncoder
    mb = _lib.mlp_backend
    W = make_weights(3)
    g = torch.Generator(device="cuda").manual_seed(700 + M)
    stride = M + 5
    enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
    dirs = torch.randn(M, 3, device="cuda", generator=g) * 1.5
    dsigma = torch.randn(M, device="cuda", generator=g) * 1e-3
    drgb = torch.randn(M, 3, device="cuda", generator=g) * 1e-3
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    denc, denc2 = torch.empty(16, stride, 2, device="cuda"), torch.empty(16, stride, 2, device="cuda")
    dws, dws2 = [torch.empty_like(w) for w in W], [torch.empty_like(w) for w in W]
    ddirs = torch.full((M + 3, 3), 7.0, device="cuda")
    mb.backward(enc, stride, dirs, dsigma, drgb, None, M, image, 1024.0, denc, dws, ddirs=ddirs)
    mb.backward(enc, stride, dirs, dsigma, drgb, None, M, image, 1024.0, denc2, dws2)
    assert torch.equal(denc[:, :M], denc2[:, :M]) and all(torch.equal(a, b) for a, b in zip(dws, dws2))
    assert torch.all(ddirs[M:] == 7.0)
    # reference: fp32 torch field over the SH autograd op
    enc_bf = enc[:, :M].permute(1, 0, 2).reshape(M, 32)
    d = dirs.clone().requires_grad_(True)
    h = torch.relu(enc_bf @ W[0].t())
    h = torch.relu(h @ W[1].t())
    h = h @ W[2].t()
    sh = SHEncoder(degree=4)(d)                                  # normalises, like renderer.py:541 + the module
    c = torch.relu(torch.cat([h[:, 1:], sh], -1) @ W[3].t())
    c = torch.relu(c @ W[4].t())
    c = c @ W[5].t()
    (torch.clamp(torch.exp(c - 5.0), max=5.0) * drgb).sum().backward()
    l2, mx = rel(ddirs[:M], d.grad)
    assert float(d.grad.abs().max()) > 0 and l2 < 5e-2, (l2, mx)
    radial = (ddirs[:M] * dirs).sum(-1).abs() / (ddirs[:M].norm(dim=-1) * dirs.norm(dim=-1) + 1e-30)
    assert float(radial.max()) < 1e-3                            # tangent: scaling a direction does not change the colour


def test_backward_is_deterministic_and_independent_of_loss_scale():
    from raw_ngp_amd import _lib
    mb = _lib.mlp_backend
    W = make_weights(3)
    M = 5000
    enc = torch.randn(16, M, 2, device="cuda") * 0.5
    dirs = torch.randn(M, 3, device="cuda")
    dsigma = torch.randn(M, device="cuda") * 1e-3
    drgb = torch.randn(M, 3, device="cuda") * 1e-3
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    outs = []
    for scale in (1024.0, 1024.0, 128.0):
        denc = torch.zeros(16, M, 2, device="cuda")
        dws = [torch.zeros_like(w) for w in W]
        mb.backward(enc, M, dirs, dsigma, drgb, None, M, image, scale, denc, dws)
        outs.append([denc] + dws)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                      # fixed reduction order: bitwise reproducible
    for a, b in zip(outs[0], outs[2]):
        assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max())


def test_backward_saturates_instead_of_overflowing():
    """An output gradient that would overflow f16 after the loss scale (dsigma * exp(raw) * 1024 > 65504) is clipped at the
    f16 maximum instead of becoming inf: every gradient stays finite, so one bad sample cannot put NaN into the weights
    (the reference's GradScaler would skip such a step; a static scale has to survive it)."""
    W = make_weights(4)
    M = 257
    enc = torch.randn(16, M, 2, device="cuda") * 0.5
    dirs = torch.randn(M, 3, device="cuda")
    dsigma = torch.full((M,), 1e3, device="cuda")
    drgb = torch.full((M, 3), 1e3, device="cuda")
    denc, dws = run_backward(W, enc, M, dirs, dsigma, drgb, M)
    assert torch.isfinite(denc).all() and all(torch.isfinite(g).all() for g in dws)
    assert float(denc.abs().max()) > 0


@pytest.mark.parametrize("M,dead", [(1000, 0.35), (30000, 0.33), (4097, 0.0), (500, 1.0)])
def test_backward_over_a_sample_list_equals_the_backward_over_all_samples(M, dead):
    """ngp_x_mlp_backward_list: the backward over the list of samples whose output gradients can be non-zero (the fused step:
    the samples in front of the compositor's early stop) against the backward over all samples with zeros at the others.
    Per-sample results do not depend on which tile a sample sits in: d enc of the listed samples is the same BITS, in list
    order; the weight gradients are the same sums in a different order."""
    from raw_ngp_amd import _lib
    mb = _lib.mlp_backend
    W = make_weights(seed=3)
    g = torch.Generator(device="cuda").manual_seed(M)
    stride = M + 9
    enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
    dirs = torch.randn(M, 3, device="cuda", generator=g)
    dsigma = torch.randn(M, device="cuda", generator=g) * 1e-3
    drgb = torch.randn(M, 3, device="cuda", generator=g) * 1e-3
    # "rays" of 7..60 samples whose tails are dead (zero output gradients)
    live = torch.ones(M, dtype=torch.bool, device="cuda")
    at = 0
    rng = np.random.default_rng(M)
    while at < M:
        n = int(rng.integers(7, 60))
        k = int(round(n * (1.0 - dead)))
        live[at + k:at + n] = False
        at += n
    dsigma[~live] = 0.0
    drgb[~live] = 0.0
    idx = torch.nonzero(live).flatten().to(torch.int32)
    n_live = int(idx.numel())
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)

    def run(sample_index, count):
        denc = torch.full((16, stride, 2), 7.0, device="cuda")
        dws = [torch.empty_like(w) for w in W]
        mb.backward(enc, stride, dirs, dsigma, drgb, count, M, image, 1024.0, denc, dws, sample_index=sample_index)
        return denc, dws
    full, dw_full = run(None, None)
    pad = torch.zeros(max(M, 1), dtype=torch.int32, device="cuda")
    pad[:n_live] = idx
    listed, dw_list = run(pad, torch.tensor([n_live], dtype=torch.int32, device="cuda"))
    assert torch.all(full[:, :M][:, ~live] == 0.0)
    assert torch.equal(listed[:, :n_live], full[:, idx.long()])            # bit for bit, in list order
    assert torch.all(listed[:, n_live:] == 7.0)                             # nothing is written behind the list
    for a, b in zip(dw_list, dw_full):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-12, (float((a - b).abs().max()), scale)   # (measured: 3e-7)
