"""Kernels of the fused training step (csrc/engine_kernels.hip) and the step itself (nerf/engine.py).

Each kernel is checked against the CPU oracle (or the reference-generated fixture / torch for the pieces the
reference does in torch); the whole step is checked against the per-op autograd path on identical rays."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_raymarching import synth_samples  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def lib():
    from raw_ngp_amd import _lib
    _lib.load()
    return _lib


def test_near_far_v2_matches_reference_fixture(lib, golden_dir):
    g = np.load(os.path.join(golden_dir, "near_far_torch.npz"))
    N = g["rays_o"].shape[0]
    nears, fars = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    lib.engine_backend.near_far_from_aabb_v2(dev(g["rays_o"]), dev(g["rays_d"]), dev(g["aabb"]), N, float(g["min_near"]),
                                             nears, fars)
    # same IEEE operations as the torch function the reference calls (renderer.py:139-158)
    np.testing.assert_array_equal(host(nears), g["nears"][:, 0])
    np.testing.assert_array_equal(host(fars), g["fars"][:, 0])


def test_grid_forward_slab_matches_oracle(lib, orc):
    rng = np.random.default_rng(0)
    B, cap, L, H = 5000, 6000, 16, 16
    bound = 2.0
    offsets, scale = orc.grid_offsets(desired_resolution=2048 * bound)
    S = float(np.log2(scale))
    table = rng.uniform(-1, 1, (offsets[-1], 2)).astype(np.float32)
    xyz = rng.uniform(-bound * 1.02, bound * 1.02, (cap, 3)).astype(np.float32)
    x01 = ((xyz + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
    ref, _ = orc.grid_encode_forward(x01[:B], table, offsets, B, 3, 2, L, L, S, H)
    out = torch.full((L, cap, 2), 7.0, device="cuda")
    inp = torch.full((cap, 3), 7.0, device="cuda")
    cnt = torch.tensor([B, 0], dtype=torch.int32, device="cuda")
    lib.engine_backend.grid_encode_forward_slab(dev(xyz), bound, dev(table), dev(offsets), out, inp, cnt, cap, cap, L, L, S, H)
    np.testing.assert_allclose(host(out)[:, :B], ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(host(inp)[:B], x01[:B])
    assert torch.all(out[:, B:] == 7.0) and torch.all(inp[B:] == 7.0)


@pytest.mark.parametrize("costs", ["flat", "growing", "random", "one heavy level"])
@pytest.mark.parametrize("B", [5000, 6500, 1])
def test_grid_forward_slab_placement_by_cost_changes_no_bit(lib, orc, costs, B):
    """level_cost only moves (level, tile) work items between XCDs: every item still runs exactly once."""
    rng = np.random.default_rng(3)
    cap, L, H, bound = 6500, 16, 16, 1.0           # 26 tiles: 3 per slot and 2 left over; B = 1: a single live tile
    offsets, scale = orc.grid_offsets(desired_resolution=2048 * bound)
    S = float(np.log2(scale))
    table = dev(rng.uniform(-1, 1, (offsets[-1], 2)).astype(np.float32))
    xyz = dev(rng.uniform(-bound, bound, (cap, 3)).astype(np.float32))
    cnt = torch.tensor([B, 0], dtype=torch.int32, device="cuda")
    cost = {"flat": [1.0] * L, "growing": [1.0 + 0.3 * l for l in range(L)],
            "random": list(rng.uniform(0.05, 20.0, L)), "one heavy level": [1.0] * 9 + [1000.0] + [1.0] * 6}[costs]
    outs = []
    for c in (None, cost):
        out, inp = torch.full((L, cap, 2), 7.0, device="cuda"), torch.full((cap, 3), 7.0, device="cuda")
        lib.engine_backend.grid_encode_forward_slab(xyz, bound, table, dev(offsets), out, inp, cnt, cap, cap, L, L, S, H,
                                                    level_cost=c)
        outs.append((out, inp))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    with pytest.raises(RuntimeError, match="level_cost"):
        lib.engine_backend.grid_encode_forward_slab(xyz, bound, table, dev(offsets), out, inp, cnt, cap, cap, L, L, S, H,
                                                    level_cost=[1.0] * 15 + [0.0])


@pytest.mark.parametrize("T_thresh", [0.0, 1e-8, 1e-4])
def test_wave_compositing_matches_oracle(lib, orc, T_thresh):
    rng = np.random.default_rng(11)
    N = 3000
    sig, rgb, ts, rays, M = synth_samples(rng, N, max_cnt=300)
    rays[5] = [M - 3, 10]          # overflowing ray -> ignored
    rw, rws, rdep, rimg = orc.composite_rays_train_forward(sig, rgb, ts, rays, M, N, T_thresh)
    w = torch.full((M,), 7.0, device="cuda")
    ws, dep, img = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
    e = lib.engine_backend
    e.composite_rays_train_forward(dev(sig), dev(rgb), dev(ts), dev(rays), M, N, T_thresh, w, ws, dep, img)
    tol = dict(rtol=3e-4, atol=3e-6 + 2 * T_thresh)      # prefix products instead of the serial walk, __expf
    live = np.ones(M, bool)
    live[M - 3:] = rays[5, 0] + np.arange(3) >= M          # samples only the overflowing ray refers to
    covered = np.zeros(M, bool)
    for n in range(N):
        if rays[n, 1] and rays[n, 0] + rays[n, 1] <= M:
            covered[rays[n, 0]:rays[n, 0] + rays[n, 1]] = True
    np.testing.assert_allclose(host(w)[covered], rw[covered], **tol)
    np.testing.assert_allclose(host(ws), rws, **tol)
    np.testing.assert_allclose(host(dep), rdep, rtol=3e-4, atol=2e-5 + 8 * T_thresh)
    np.testing.assert_allclose(host(img), rimg, **tol)

    gw = rng.normal(size=M).astype(np.float32)
    gws = rng.normal(size=N).astype(np.float32)
    gdep = rng.normal(size=N).astype(np.float32)
    gimg = rng.normal(size=(N, 3)).astype(np.float32)
    rgs, rgc = orc.composite_rays_train_backward(gw, gws, gdep, gimg, sig, rgb, ts, rays, rws, rdep, rimg, M, N, T_thresh)
    gs, gc = torch.full((M,), 7.0, device="cuda"), torch.full((M, 3), 7.0, device="cuda")
    e.composite_rays_train_backward(dev(gw), dev(gws), dev(gdep), dev(gimg), dev(sig), dev(rgb), dev(ts), dev(rays),
                                    dev(rws), dev(rdep), dev(rimg), M, N, T_thresh, gs, gc)
    np.testing.assert_allclose(host(gc)[covered], rgc[covered], rtol=3e-4, atol=1e-5 + 8 * T_thresh)
    err = np.abs(host(gs) - rgs)[covered]
    assert np.all(err <= 3e-3 * np.abs(rgs[covered]) + 1e-3 * np.abs(ts[covered, 1]) * 50 + 1e-6)


def test_mse_backward_matches_autograd(lib, orc):
    rng = np.random.default_rng(12)
    N = 1500
    sig, rgb, ts, rays, M = synth_samples(rng, N)
    gt = rng.uniform(0, 1, (N, 4)).astype(np.float32)
    bg = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    e = lib.engine_backend
    from raw_ngp_amd import raymarching
    tsig, trgb = dev(sig).requires_grad_(True), dev(rgb).requires_grad_(True)
    w, ws, dep, img = raymarching.composite_rays_train(tsig, trgb, dev(ts), dev(rays), 1e-4)
    pred = img + (1 - ws).unsqueeze(-1) * dev(bg)
    tgt = dev(gt)[:, :3] * dev(gt)[:, 3:] + dev(bg) * (1 - dev(gt)[:, 3:])
    loss = ((pred - tgt) ** 2).mean(-1).mean()
    loss.backward()
    gs, gc = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
    lo = torch.zeros(1, device="cuda")
    e.composite_mse_backward(dev(gt), dev(bg), 0.0, tsig.detach(), trgb.detach(), dev(ts), dev(rays), ws.detach(),
                             dep.detach(), img.detach(), M, N, 1e-4, gs, gc, lo)
    np.testing.assert_allclose(float(lo), float(loss.detach()), rtol=1e-5)
    covered = np.zeros(M, bool)
    for n in range(N):
        covered[rays[n, 0]:rays[n, 0] + rays[n, 1]] = True
    np.testing.assert_allclose(host(gc)[covered], host(trgb.grad)[covered], rtol=1e-3, atol=1e-8)
    ref = host(tsig.grad)[covered]
    assert np.all(np.abs(host(gs)[covered] - ref) <= 3e-3 * np.abs(ref) + 1e-7)


@pytest.mark.parametrize("lam", [1e-2, 0.5])
def test_entropy_term_of_the_fused_compositor_step_matches_autograd(lib, lam):
    """lambda_entropy * mean_rays(H(clamp(weights_sum, 1e-5, 1 - 1e-5))) (train_utils.py:554-557) inside the one-launch
    compositor step: loss value and d sigma / d rgb against torch autograd of the same expression over the per-op compositor.
    Rays with weights_sum pinned at the clamp (opaque rays) get no entropy gradient, as with torch.clamp."""
    rng = np.random.default_rng(13)
    N = 1200
    sig, rgb, ts, rays, M = synth_samples(rng, N)
    sig[rays[7, 0]:rays[7, 0] + rays[7, 1]] = 1e4          # an opaque ray: weights_sum == 1 -> clamped
    gt = rng.uniform(0, 1, (N, 4)).astype(np.float32)
    e = lib.engine_backend
    from raw_ngp_amd import raymarching
    tsig, trgb = dev(sig).requires_grad_(True), dev(rgb).requires_grad_(True)
    w, ws, dep, img = raymarching.composite_rays_train(tsig, trgb, dev(ts), dev(rays), 1e-4)
    tgt = dev(gt)[:, :3] * dev(gt)[:, 3:]
    wc = ws.clamp(1e-5, 1 - 1e-5)
    ent = (-wc * torch.log2(wc) - (1 - wc) * torch.log2(1 - wc)).mean()
    loss = ((img - tgt) ** 2).mean(-1).mean() + lam * ent
    loss.backward()
    gs, gc = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
    lo = torch.zeros(1, device="cuda")
    ws2, dep2, img2 = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
    e.composite_train_live(dev(gt), None, 0.0, None, None, 1.0 / (3 * N), None, tsig.detach(), trgb.detach(), dev(ts), dev(rays),
                           M, N, 1e-4, ws2, dep2, img2, gs, gc, lo, lambda_entropy=lam)
    np.testing.assert_allclose(float(lo), float(loss.detach()), rtol=2e-5)
    assert float(ent.detach()) > 0.05
    covered = np.zeros(M, bool)
    for n in range(N):
        covered[rays[n, 0]:rays[n, 0] + rays[n, 1]] = True
    np.testing.assert_allclose(host(gc)[covered], host(trgb.grad)[covered], rtol=1e-3, atol=1e-8)
    ref = host(tsig.grad)[covered]
    assert np.all(np.abs(host(gs)[covered] - ref) <= 3e-3 * np.abs(ref) + 1e-7 + 1e-6 * lam)
    # the entropy term really contributes to d sigma
    gs0 = torch.empty(M, device="cuda")
    e.composite_train_live(dev(gt), None, 0.0, None, None, 1.0 / (3 * N), None, tsig.detach(), trgb.detach(), dev(ts), dev(rays),
                           M, N, 1e-4, ws2, dep2, img2, gs0, gc, torch.zeros(1, device="cuda"))
    assert float((gs - gs0).abs().max()) > 1e-5 * lam


def test_adam_matches_torch(lib):
    torch.manual_seed(0)
    n = 100003                                   # exercises the scalar tail
    p0 = torch.randn(n, device="cuda")
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-15)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 6):
        g = torch.randn(n, device="cuda") * (10.0 ** -step)
        g[::3] = 0.0
        lr = 1e-2 * 0.9 ** step
        for grp in opt.param_groups:
            grp["lr"] = lr
        ref.grad = g.clone()
        opt.step()
        gg = g.clone()
        lib.engine_backend.adam_step(p, gg, m, v, lr, 0.9, 0.999, 1e-15, step, zero_grad=(step % 2 == 0))
        assert torch.all(gg == 0) if step % 2 == 0 else torch.equal(gg, g)
        np.testing.assert_allclose(host(p), host(ref), rtol=2e-6, atol=3e-7)


def test_fused_step_matches_autograd_path(lib):
    """Same rays, no jitter: loss and gradients of the fused step vs the per-op autograd path (both with the
    fused MLP; differences: wave vs serial compositing, fused loss)."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=100, fused_mlp=True)
    data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=4, H=64, W=64)
    model = NeRFNetwork(opt).cuda()
    with torch.no_grad():                                     # make the field non-trivial
        model.grid_encoder.embeddings.uniform_(-0.5, 0.5)
    eng = FusedTrainer(opt, model, data, device="cuda", capacity=1024 * 256)
    model.train()
    model.update_extra_state()
    batch = data.sample_rays(opt.num_rays, torch.Generator(device="cuda").manual_seed(1))
    gt = batch["images"]
    eng.forward_backward(batch["rays_o"].contiguous(), batch["rays_d"].contiguous(), gt.contiguous(),
                         torch.zeros(opt.num_rays, device="cuda"))
    M = int(eng.arena.counter[0])
    assert 0 < M <= eng.cap and int(eng.arena.counter[1]) == M

    model.zero_grad()
    out = model.render(batch["rays_o"], batch["rays_d"], bg_color=0, perturb=False)
    assert out["num_points"] == M
    tgt = gt[:, :3] * gt[:, 3:]
    loss = ((out["image"] - tgt) ** 2).mean(-1).mean()
    loss.backward()
    np.testing.assert_allclose(float(eng.loss), float(loss.detach()), rtol=2e-4)
    ref_t = model.grid_encoder.embeddings.grad
    ref_w = torch.cat([l.weight.grad.reshape(-1) for l in list(model.grid_mlp.net) + list(model.view_mlp.net)])

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    assert rel(eng.table_grad, ref_t) < 2e-3
    assert rel(eng.w_grad, ref_w) < 2e-3
    # and the step itself moves the parameters like torch.optim.Adam does
    p_before = eng.table.clone()
    eng.optimizer_step()
    ref_p = torch.nn.Parameter(p_before.clone())
    ref_p.grad = ref_t.clone()
    torch.optim.Adam([ref_p], lr=eng.lr(), eps=1e-15).step()
    changed = ref_t != 0
    np.testing.assert_allclose(host(eng.table[changed]), host(ref_p.detach()[changed]), rtol=0, atol=2.1 * eng.lr())
    assert float((eng.table - ref_p.detach()).abs().mean()) < 0.05 * eng.lr()
    assert torch.all(eng.table_grad == 0)


def test_sample_weight_term_of_the_fused_compositor_step_matches_the_per_op_backward(lib):
    """loss += lambda * sum_i weights[i] * term[i] (the shape of the orientation term, renderer.py:571) inside the one-launch
    compositor step: loss value and d sigma / d rgb against torch autograd over the per-op compositor, whose backward gives
    grad_weights the reference's treatment (raymarching.cu:694)."""
    rng = np.random.default_rng(17)
    N, lam = 1200, 0.3
    sig, rgb, ts, rays, M = synth_samples(rng, N)
    term = (rng.uniform(0, 1, M) ** 2).astype(np.float32)
    gt = rng.uniform(0, 1, (N, 4)).astype(np.float32)
    e = lib.engine_backend
    from raw_ngp_amd import raymarching
    tsig, trgb = dev(sig).requires_grad_(True), dev(rgb).requires_grad_(True)
    w, ws, dep, img = raymarching.composite_rays_train(tsig, trgb, dev(ts), dev(rays), 1e-4)
    tgt = dev(gt)[:, :3] * dev(gt)[:, 3:]
    extra = torch.mean((w * dev(term)).sum(dim=-1))
    loss = ((img - tgt) ** 2).mean(-1).mean() + lam * extra
    loss.backward()
    gs, gc = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
    lo = torch.zeros(1, device="cuda")
    ws2, dep2, img2 = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
    live = (torch.empty(N, dtype=torch.int32, device="cuda"), torch.empty(M, dtype=torch.int32, device="cuda"),
            torch.empty(1, dtype=torch.int32, device="cuda"), None)
    for lv in (None, live):
        lo.zero_()
        e.composite_train_live(dev(gt), None, 0.0, None, None, 1.0 / (3 * N), None, tsig.detach(), trgb.detach(), dev(ts),
                               dev(rays), M, N, 1e-4, ws2, dep2, img2, gs, gc, lo, live=lv, sample_term=dev(term),
                               lambda_sample=lam)
        np.testing.assert_allclose(float(lo), float(loss.detach()), rtol=2e-5)
        assert float(extra.detach()) > 1.0                                  # a sum over the samples, not a mean over the rays
        covered = np.zeros(M, bool)
        for n in range(N):
            covered[rays[n, 0]:rays[n, 0] + rays[n, 1]] = True
        np.testing.assert_allclose(host(gc)[covered], host(trgb.grad)[covered], rtol=1e-3, atol=1e-8)
        ref = host(tsig.grad)[covered]
        assert np.all(np.abs(host(gs)[covered] - ref) <= 3e-3 * np.abs(ref) + 1e-6)
    gs0 = torch.empty(M, device="cuda")
    e.composite_train_live(dev(gt), None, 0.0, None, None, 1.0 / (3 * N), None, tsig.detach(), trgb.detach(), dev(ts), dev(rays),
                           M, N, 1e-4, ws2, dep2, img2, gs0, gc, torch.zeros(1, device="cuda"))
    assert float((gs - gs0).abs().max()) > 1e-3
    with pytest.raises(RuntimeError, match="lambda_sample"):
        e.composite_train_live(dev(gt), None, 0.0, None, None, 1.0 / (3 * N), None, tsig.detach(), trgb.detach(), dev(ts),
                               dev(rays), M, N, 1e-4, ws2, dep2, img2, gs0, gc, lo, sample_term=dev(term), lambda_sample=-1.0)


def test_fused_step_with_the_orientation_term_matches_the_per_op_path(lib):
    """lambda_orientation > 0 (renderer.py:558-571, train_utils.py:546-548): the per-sample term, the loss and the gradients
    of the fused step against the per-op path over the same rays (fused MLP on both sides), and the term itself against torch
    autograd through fp32 nn.Linear MLPs."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    torch.manual_seed(0)
    lam = 1e-2
    opt = Options(bound=1.0, num_rays=1024, iters=100, fused_mlp=True, lambda_orientation=lam, lambda_distort=1e-3)
    data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=4, H=64, W=64)
    model = NeRFNetwork(opt).cuda()
    with torch.no_grad():                                     # make the field non-trivial
        model.grid_encoder.embeddings.uniform_(-0.5, 0.5)
    eng = FusedTrainer(opt, model, data, device="cuda", capacity=1024 * 256)     # (lambda_distort: no term on this path)
    assert eng.orient
    model.train()
    model.update_extra_state()
    batch = data.sample_rays(opt.num_rays, torch.Generator(device="cuda").manual_seed(1))
    gt = batch["images"]
    eng.forward_backward(batch["rays_o"].contiguous(), batch["rays_d"].contiguous(), gt.contiguous(),
                         torch.zeros(opt.num_rays, device="cuda"))
    M = int(eng.arena.counter[0])
    assert 0 < M <= eng.cap
    term = eng.orient_term[:M].clone()

    model.zero_grad()
    out = model.render(batch["rays_o"], batch["rays_d"], bg_color=0, perturb=False)
    assert out["num_points"] == M
    tgt = gt[:, :3] * gt[:, 3:]
    mse = ((out["image"] - tgt) ** 2).mean(-1).mean()
    loss = mse + lam * out["orientation_loss"]
    assert float(lam * out["orientation_loss"].detach()) > 0.05 * float(mse.detach())     # the term matters here
    loss.backward()
    np.testing.assert_allclose(float(eng.loss), float(loss.detach()), rtol=1e-3)
    ref_t = model.grid_encoder.embeddings.grad
    ref_w = torch.cat([l.weight.grad.reshape(-1) for l in list(model.grid_mlp.net) + list(model.view_mlp.net)])

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    assert rel(eng.table_grad, ref_t) < 5e-3
    assert rel(eng.w_grad, ref_w) < 5e-3

    # the per-sample term: torch autograd through the same positions, fused MLP and fp32 nn.Linear
    ar = eng.arena
    xyzs, dirs = ar.xyzs[:M].clone(), ar.dirs[:M] / ar.dirs[:M].norm(dim=-1, keepdim=True)

    def torch_term():
        pos = xyzs.clone().requires_grad_(True)
        sigma = model(pos, dirs, None)["sigma"]
        n = torch.autograd.grad(sigma, pos, grad_outputs=torch.ones_like(sigma))[0]
        n = (-torch.nn.functional.normalize(n, dim=-1) + 1) / 2
        return torch.clamp((n * -dirs).sum(-1), max=0.0) ** 2
    same = torch_term()
    assert float(same.mean()) > 1e-3 and float((same > 0).float().mean()) > 0.03
    assert float((term - same).abs().max()) < 2e-2 and float((term - same).abs().mean()) < 1e-3
    opt.fused_mlp = False
    exact = torch_term()
    opt.fused_mlp = True
    err = (term - exact).abs()
    assert float(err.mean()) < 5e-3 and float(err.quantile(0.99)) < 5e-2, (float(err.mean()), float(err.quantile(0.99)))


def test_sample_rays_matches_oracle(lib, orc):
    rng = np.random.default_rng(3)
    V, H, W, N = 7, 40, 52, 5000
    images = rng.integers(0, 256, (V, H, W, 4), dtype=np.uint8)
    poses = np.tile(np.eye(4, dtype=np.float32), (V, 1, 1))
    poses[:, :3, :4] = rng.normal(size=(V, 3, 4)).astype(np.float32)
    intr = np.array([55.0, 54.0, W / 2, H / 2], np.float32)
    e = lib.engine_backend
    f = lambda *s: torch.empty(*s, device="cuda")
    for C in (4, 3):
        img = np.ascontiguousarray(images[..., :C])
        want = orc.sample_rays(img, poses, intr, N, seed=(9 << 32) | 77, draw=12)
        ro, rd, gt, nz, bg = f(N, 3), f(N, 3), f(N, 4), f(N), f(N, 3)
        idx = torch.empty(N, 2, dtype=torch.int32, device="cuda")
        draw = torch.tensor([12], dtype=torch.int32, device="cuda")
        e.sample_rays(dev(img), dev(poses), intr, N, (9 << 32) | 77, draw, ro, rd, gt, nz, bg, idx)
        np.testing.assert_array_equal(host(idx), want["index"])            # integer draws: bit exact
        np.testing.assert_array_equal(host(gt), want["gt"])
        np.testing.assert_array_equal(host(nz), want["noises"])
        np.testing.assert_array_equal(host(bg), want["bg"])
        np.testing.assert_array_equal(host(ro), want["rays_o"])
        np.testing.assert_allclose(host(rd), want["rays_d"], rtol=1e-6, atol=1e-6)
        # host-supplied draw number, optional outputs omitted
        ro2, rd2, gt2 = f(N, 3), f(N, 3), f(N, 4)
        e.sample_rays(dev(img), dev(poses), intr, N, (9 << 32) | 77, 12, ro2, rd2, gt2)
        assert torch.equal(rd2, rd) and torch.equal(gt2, gt)
        e.counter_add(draw, 1)
        assert int(draw) == 13


def test_device_schedule_and_adam_match_host_versions(lib, orc):
    e = lib.engine_backend
    torch.manual_seed(1)
    n = 4099
    ctr = torch.tensor([0], dtype=torch.int32, device="cuda")
    hyper = torch.zeros(4, device="cuda")
    p0, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    pa, ma, va = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb, mb_, vb = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 40):
        e.schedule_step(ctr, hyper, 1e-2, 30.0, 0.9, 0.999)
        lr, bc1, rs = orc.schedule(step - 1, 1e-2, 30.0, 0.9, 0.999)
        np.testing.assert_allclose(host(hyper)[:3], [lr, bc1, rs], rtol=2e-7)
        assert int(ctr) == step
        e.adam_step_dev(pa, g.clone(), ma, va, hyper, 0.9, 0.999, 1e-15)
        e.adam_step(pb, g.clone(), mb_, vb, float(lr), 0.9, 0.999, 1e-15, step)
        np.testing.assert_allclose(host(pa), host(pb), rtol=1e-6, atol=1e-8)


def test_graph_replay_matches_eager_steps(lib):
    """Same seed, same device-side ray draws: 40 steps replayed from captured graphs vs launched one by one."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    res = []
    for graph in (True, False):
        torch.manual_seed(0)
        opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True, capture_graph=graph)
        data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
        eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
        assert eng.use_graph == graph and eng.device_sampler
        losses = []
        for _ in range(40):
            losses.append(eng.train_step().clone())
        torch.cuda.synchronize()
        assert (eng.last_graph_key is not None) == graph
        assert int(eng.step_ctr) == 40 and int(eng.draw_ctr) == 41      # step 40 is already drawn and marched
        res.append((torch.cat(losses).cpu().numpy(), int(eng.samples_seen), eng.table.clone()))
    (la, sa, ta), (lb, sb, tb) = res
    np.testing.assert_allclose(la[:6], lb[:6], rtol=2e-3)            # same draws, same kernels: only atomics order differs
    np.testing.assert_allclose(la[:16], lb[:16], rtol=5e-2)          # ... and that difference grows step by step
    np.testing.assert_allclose(la, lb, rtol=0.1)
    assert abs(sa - sb) <= 0.02 * sb


@pytest.mark.parametrize("loss_weight,lam", [("none", 0.0), ("planck", 0.0), ("gaussian", 0.0), ("hanning", 0.0), ("gaussian", 0.05)])
def test_fused_step_with_the_hdr_loss_matches_the_per_op_path(lib, loss_weight, lam):
    """image_mode HDR (train_utils.py:512-536) with each loss weight of raw_utils.py:30-53 (and with the entropy term on top):
    loss and gradients of the fused step against torch autograd over the per-op renderer and nerf.utils.hdr_loss."""
    from raw_ngp_amd.nerf import utils
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=100, fused_mlp=True, image_mode="HDR", loss_weight=loss_weight,
                  lambda_entropy=lam)
    data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=4, H=64, W=64)
    data.exposures = torch.tensor([50.0, 200.0, 400.0, 800.0], device="cuda")    # (the untrained field renders ~ 0.003)
    model = NeRFNetwork(opt).cuda()
    with torch.no_grad():
        model.grid_encoder.embeddings.uniform_(-0.5, 0.5)
    eng = FusedTrainer(opt, model, data, device="cuda", capacity=1024 * 256)
    model.train()
    model.update_extra_state()
    batch = data.sample_rays(opt.num_rays, torch.Generator(device="cuda").manual_seed(1))
    gt = batch["images"]
    exposure = data.exposures[batch["index"]]
    eng.slots[0].exposure.copy_(exposure)
    eng.forward_backward(batch["rays_o"].contiguous(), batch["rays_d"].contiguous(), gt.contiguous(),
                         torch.zeros(opt.num_rays, device="cuda"))
    model.zero_grad()
    out = model.render(batch["rays_o"], batch["rays_d"], bg_color=0, perturb=False)
    tgt = gt[:, :3] * gt[:, 3:]
    loss = utils.hdr_loss(out["image"], tgt, exposure, loss_weight)
    if lam > 0:
        w = out["weights_sum"].clamp(1e-5, 1 - 1e-5)
        loss = loss + lam * (-w * torch.log2(w) - (1 - w) * torch.log2(1 - w)).mean()
    loss.backward()
    np.testing.assert_allclose(float(eng.loss), float(loss.detach()), rtol=1e-3)
    clipped = float((out["image"].detach() * exposure[:, None] > 1).float().mean())
    assert 0.02 < clipped < 0.98, clipped                       # predictions on both sides of the clip at white
    ref_t = model.grid_encoder.embeddings.grad
    ref_w = torch.cat([l.weight.grad.reshape(-1) for l in list(model.grid_mlp.net) + list(model.view_mlp.net)])

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    assert rel(eng.table_grad, ref_t) < 5e-3, rel(eng.table_grad, ref_t)
    assert rel(eng.w_grad, ref_w) < 5e-3, rel(eng.w_grad, ref_w)


@pytest.mark.parametrize("acts", [{}, dict(density_activation="softplus", beta=2.0, color_activation="sigmoid")],
                         ids=["", "softplus-density"])
@pytest.mark.parametrize("rfield", [False, True], ids=["plain", "light-conditioned"])
def test_orientation_term_of_a_train_step_matches_autograd(lib, rfield, acts):
    """The per-sample term a train_step leaves behind (graphs, march ahead on the side stream) against torch autograd through
    the model at the same samples -- learning rate 0, so the weights are the ones the step saw; both fields; also with a softplus
    density (ngp_x_orientation_term_act) and sigmoid colour."""
    from raw_ngp_amd.nerf import pose as P
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True, lambda_orientation=1e-2, rfield=rfield, lr=0.0, **acts)
    data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
    if rfield:
        data.ldirs = torch.from_numpy(P.synthetic_light_dirs(6)).cuda()
    model = NeRFNetwork(opt).cuda()
    with torch.no_grad():
        model.grid_encoder.embeddings.uniform_(-0.5, 0.5)
    eng = FusedTrainer(opt, model, data, device="cuda", capacity=1024 * 256)
    table0 = eng.table.clone()
    for _ in range(5):
        eng.train_step()
    torch.cuda.synchronize()
    assert torch.equal(eng.table, table0)
    slot = eng.slots[(eng.global_step - 1) % len(eng.slots)]
    M = int(slot.arena.counter[0])
    assert 0 < M <= eng.cap
    term = eng.orient_term[:M].clone()
    xyzs = slot.arena.xyzs[:M].clone()
    dirs = slot.arena.dirs[:M] / slot.arena.dirs[:M].norm(dim=-1, keepdim=True)
    ldirs = slot.arena.ldirs[:M].clone() if rfield else None
    model.train()
    pos = xyzs.clone().requires_grad_(True)
    sigma = model(pos, dirs, ldirs)["sigma"]
    n = torch.autograd.grad(sigma, pos, grad_outputs=torch.ones_like(sigma))[0]
    n = (-torch.nn.functional.normalize(n, dim=-1) + 1) / 2
    want = torch.clamp((n * -dirs).sum(-1), max=0.0) ** 2
    assert float((want > 0).float().mean()) > 0.03
    err = (term - want).abs()
    # (a nearly flat density leaves the direction of its gradient to the f16 rounding of either chain: a few samples may differ)
    assert float((err > 2e-2).float().mean()) < 1e-3 and float(err.mean()) < 1e-3, (float(err.max()), float(err.mean()))


def test_orientation_term_in_the_replayed_step(lib):
    """The step path with lambda_orientation > 0 (the term's two launches sit between the MLP forward and the compositor
    step): replayed from graphs vs launched one by one, and it is not the step without the term."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    res = {}
    for graph, lam in ((True, 1e-2), (False, 1e-2), (True, 0.0)):
        torch.manual_seed(0)
        opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True, capture_graph=graph, lambda_orientation=lam)
        data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
        eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
        assert eng.use_graph == graph and eng.orient == (lam > 0)
        losses = [eng.train_step().clone() for _ in range(40)]
        torch.cuda.synchronize()
        res[(graph, lam)] = torch.cat(losses).cpu().numpy()
    a, b, plain = res[(True, 1e-2)], res[(False, 1e-2)], res[(True, 0.0)]
    assert np.all(np.isfinite(a)) and a[-1] < a[0]
    np.testing.assert_allclose(a[:6], b[:6], rtol=2e-3)
    np.testing.assert_allclose(a, b, rtol=0.1)
    assert np.max(np.abs(a[:6] - plain[:6]) / plain[:6]) > 1e-3, (a[:6], plain[:6])


@pytest.mark.parametrize("full", [False, True], ids=["partial", "full-sweep"])
def test_density_grid_refresh_kernels_match_oracle(lib, orc, full):
    """Cell draws (bit-exact indices), jittered positions, scatter/EMA-max/mean and the device-thresholded packbits."""
    e = lib.engine_backend
    rng = np.random.default_rng(21)
    H, bound = 32, 2.0
    cells = H ** 3
    grid = rng.uniform(-0.5, 3.0, (1, cells)).astype(np.float32)
    grid[0, rng.random(cells) < 0.3] = -1.0                      # untrained cells
    grid[0, rng.random(cells) < 0.3] = 0.0
    half = bound / H
    n_uni, n_occ = (cells, 0) if full else (cells // 4, cells // 4)
    n = n_uni + n_occ
    seed, draw = (3 << 32) | 12345, 7
    ws = torch.empty(e.density_grid_workspace_bytes(H), dtype=torch.uint8, device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    xyz = torch.empty(n, 3, device="cuda")
    dgrid = dev(grid)
    e.density_grid_sample(dgrid[0], H, bound - half, half, n_uni, n_occ, full, seed, draw, ws, idx, xyz)
    # (random draws leave bin by bin, in Morton order of their cells -- the restatement's binned=True; a full sweep is the cells
    # in order)
    ridx, rxyz = orc.density_grid_sample(grid[0], H, bound - half, half, n_uni, n_occ, full, seed, draw, binned=True)
    gidx, gxyz = host(idx), host(xyz)
    np.testing.assert_array_equal(gidx, ridx)                    # slot by slot
    np.testing.assert_allclose(gxyz, rxyz, rtol=0, atol=2e-7)
    if not full:
        # Morton order up to the width of a bin (4096 bins per half: blocks of cells for the uniform half, ranges of picks for
        # the occupied one); the occupied half only picks occupied cells, every one of them about equally often
        shift = max(int(np.ceil(np.log2(cells))) - 12, 0)
        assert np.all(np.diff(gidx[:n_uni] >> shift) >= 0)
        occ = np.flatnonzero(grid[0] > 0)
        rank = np.searchsorted(occ, gidx[n_uni:])
        assert np.all(occ[rank] == gidx[n_uni:]) and np.all(np.diff(rank) >= -(len(occ) // 4096 + 2))
        hist = np.bincount(gidx[n_uni:], minlength=cells)[grid[0] > 0]
        assert hist.max() <= 12 and abs(hist.mean() - n_occ / (grid[0] > 0).sum()) < 1e-9
        # the same law as independent draws (the restatement's binned=False: another sample of it): every cell of the uniform
        # half is hit Binomial(n, 1 / cells) often -- compare the histograms of the hit counts
        iidx, _ = orc.density_grid_sample(grid[0], H, bound - half, half, n_uni, n_occ, full, seed, draw)
        hb, hi_ = (np.bincount(np.bincount(v[:n_uni], minlength=cells), minlength=8)[:8] for v in (gidx, iidx))
        expect = cells * np.array([np.exp(-0.25) * 0.25 ** k / np.prod(np.arange(1, k + 1)) for k in range(8)])
        for h in (hb, hi_):     # (Poisson(1/4) to within 4 sigma of the bin's count)
            assert np.all(np.abs(h - expect) <= 4 * np.sqrt(expect) + 2), (h, expect)
    # device counter as the draw number gives the same cells
    idx2 = torch.empty_like(idx)
    e.density_grid_sample(dgrid[0], H, bound - half, half, n_uni, n_occ, full, seed,
                          torch.tensor([draw], dtype=torch.int32, device="cuda"), ws, idx2, torch.empty_like(xyz))
    assert torch.equal(idx, idx2)
    # scatter + update + packbits
    sig = rng.uniform(0, 4, n).astype(np.float32)
    tmp = torch.full((1, cells), -1.0, device="cuda")
    e.density_grid_scatter(idx, dev(sig), n, tmp[0])
    rtmp = np.full(cells, -1.0, np.float32)
    np.maximum.at(rtmp, ridx, sig)
    np.testing.assert_array_equal(host(tmp)[0], rtmp)
    stats = torch.zeros(4 + 1024, device="cuda")
    e.density_grid_update(dgrid, tmp, 0.95, stats)
    rgrid, rmean = orc.density_grid_update(grid[0], rtmp, 0.95)
    np.testing.assert_array_equal(host(dgrid)[0], rgrid)
    assert torch.all(tmp == -1.0)
    bits = torch.zeros(cells // 8, dtype=torch.uint8, device="cuda")
    e.packbits_mean(dgrid, stats, 0.9, bits)
    np.testing.assert_allclose(float(stats[1]), rmean, rtol=1e-5)
    thresh = float(stats[2])
    assert thresh == min(float(stats[1]), np.float32(0.9))
    np.testing.assert_array_equal(host(bits), orc.packbits(rgrid, thresh))


def test_density_grid_sample_with_nothing_occupied(lib):
    e = lib.engine_backend
    H = 16
    grid = torch.full((H ** 3,), -1.0, device="cuda")
    ws = torch.empty(e.density_grid_workspace_bytes(H), dtype=torch.uint8, device="cuda")
    idx = torch.empty(2048, dtype=torch.int32, device="cuda")
    xyz = torch.empty(2048, 3, device="cuda")
    e.density_grid_sample(grid, H, 0.9, 0.1, 1024, 1024, False, 1, 0, ws, idx, xyz)
    assert torch.all(idx[1024:] == -1) and torch.all(idx[:1024] >= 0) and torch.all(xyz[1024:] == 0)


def test_fused_adam_matches_separate_adam(lib):
    """Adam applied inside the table-gradient reduction (single GPU default) vs gradient to HBM + Adam kernel."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    tabs = []
    for fuse in (True, False):
        torch.manual_seed(0)
        opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True, capture_graph=False, fuse_adam=fuse)
        data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
        eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
        assert eng.fuse_adam == fuse
        t0 = eng.table.clone()
        eng.train_step()         # ONE step: later steps amplify rounding differences (Adam moves rows with a
        torch.cuda.synchronize()  # near-zero gradient by +-lr whichever sign the rounding gives it)
        n_t = eng.table.numel()         # (separate Adam: the moments cover the flat parameter -- table, MLP weights, padding)
        tabs.append((eng.table.clone() - t0, eng.t_m.reshape(-1)[:n_t].clone(), eng.t_v.reshape(-1)[:n_t].clone(),
                     float(eng.loss)))
    (da, ma, va, la), (db, mb_, vb, lb) = tabs
    assert float(da.abs().max()) > 5e-3                       # the table moved (lr 1e-2)
    # The very first occupancy grid thresholds an untrained, almost constant density field at its own mean (float
    # atomics: the last bit of that mean depends on the order), so a few cells -- hence a few samples -- may differ
    # between two runs.  Everything else is the same arithmetic.
    np.testing.assert_allclose(la, lb, rtol=2e-3)
    assert float((ma - mb_).norm()) <= 2e-2 * float(mb_.norm())
    assert float((da - db).abs().mean()) < 0.05 * float(db.abs().mean())


def test_slab_forward_counts_like_the_count_kernel(lib, orc):
    """Bin sizes of the binned backward: counted by the encoder's forward pass (stage 1 + forward + stage 2) vs by the
    stand-alone count kernel (stage 0) -- identical workspace headers, with and without run merging."""
    rng = np.random.default_rng(4)
    B, L, H, bound = 20000, 16, 16, 1.0
    offsets, scale = orc.grid_offsets(desired_resolution=2048)
    S, rows = float(np.log2(scale)), int(offsets[-1])
    # ray-like runs: 400 rays x 50 steps of 0.0034, plus a few points outside the box
    o = rng.uniform(-0.8, 0.8, (400, 1, 3))
    d = rng.normal(size=(400, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    xyz = (o + d * (np.arange(50)[None, :, None] * 0.0034)).reshape(-1, 3).astype(np.float32)
    xyz[::997] = 1.5
    gb, eb = lib.gridencoder_backend, lib.engine_backend
    table = dev(rng.uniform(-1, 1, (rows, 2)).astype(np.float32))
    n_chunks_max = rows // gb.binned_geometry()[0] + L + 1
    head_words = 64 + 4 + 4 * n_chunks_max + 2
    for merge in (0, 300):
        ws_a = torch.zeros(gb.backward_workspace_bytes(B, L, rows), dtype=torch.uint8, device="cuda")
        ws_b = torch.zeros_like(ws_a)
        cnt = torch.tensor([B, B, 0, 0], dtype=torch.int32, device="cuda")
        gb.grid_backward_binned_prepare(dev(xyz), bound, dev(offsets), rows, cnt, B, L, L, S, H, ws_a, merge_max_res=merge)
        gb.grid_backward_binned_prepare(None, 0.0, dev(offsets), rows, cnt, B, L, L, S, H, ws_b, merge_max_res=merge, stage=1)
        enc, x01 = torch.empty(L, B, 2, device="cuda"), torch.empty(B, 3, device="cuda")
        eb.grid_encode_forward_slab(dev(xyz), bound, table, dev(offsets), enc, x01, cnt, B, B, L, L, S, H,
                                    binned_workspace=ws_b)
        gb.grid_backward_binned_prepare(None, 0.0, dev(offsets), rows, cnt, B, L, L, S, H, ws_b, stage=2)
        a = host(ws_a[:head_words * 4]).view(np.uint32)
        b = host(ws_b[:head_words * 4]).view(np.uint32)
        np.testing.assert_array_equal(a, b)
        counts = a[68:68 + n_chunks_max]
        assert counts.sum() > 0 and (merge == 0 or counts.sum() < 0.8 * B * L * 8)
        # and the forward result itself is unchanged by the counting
        enc2 = torch.empty_like(enc)
        eb.grid_encode_forward_slab(dev(xyz), bound, table, dev(offsets), enc2, x01, cnt, B, B, L, L, S, H)
        assert torch.equal(enc, enc2)


def test_binned_backward_with_zero_gradient_tails(lib, orc):
    """Samples behind the compositor's early stop carry exactly zero gradient: the binned backward drops their
    records (and the MLP backward their tiles); the table gradient must equal the oracle's all the same."""
    rng = np.random.default_rng(8)
    L, H, bound, per_ray, n_rays = 16, 16, 1.0, 48, 300
    B = per_ray * n_rays
    offsets, scale = orc.grid_offsets(desired_resolution=2048)
    S, rows = float(np.log2(scale)), int(offsets[-1])
    o = rng.uniform(-0.7, 0.7, (n_rays, 1, 3))
    d = rng.normal(size=(n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x01 = ((o + d * (np.arange(per_ray)[None, :, None] * 0.0034)).reshape(-1, 3) * 0.5 + 0.5).astype(np.float32)
    grad = rng.normal(size=(L, B, 2)).astype(np.float32)
    dead = (np.arange(B) % per_ray) >= rng.integers(5, per_ray, n_rays).repeat(per_ray)      # a suffix of every ray
    grad[:, dead] = 0.0
    assert 0.2 < dead.mean() < 0.8
    ref = orc.grid_encode_backward(np.ascontiguousarray(grad), x01, np.zeros((rows, 2), np.float32), offsets, B, 3, 2, L,
                                   L, S, H)[0]
    gb = lib.gridencoder_backend
    out = torch.zeros(rows, 2, device="cuda")
    ws = torch.zeros(gb.backward_workspace_bytes(B, L, rows), dtype=torch.uint8, device="cuda")
    cnt = torch.tensor([B, B, 0, 0], dtype=torch.int32, device="cuda")
    gb.grid_backward_binned(dev(grad), dev(x01), dev(offsets), out, cnt, B, B, L, L, S, H, ws)
    np.testing.assert_allclose(host(out), ref, rtol=2e-5, atol=2e-6)
    # fewer records were written than the samples would have emitted with gradients everywhere
    chunk_rows, fill_tile, region, _ = gb.binned_geometry()
    n_chunks_max = rows // chunk_rows + L + 1
    if gb.backward_needs_counts(B, L, dev(offsets)):     # global-bins layout: per-chunk counts (reserved) and cursors (written)
        head = host(ws[:(68 + 2 * n_chunks_max) * 4]).view(np.uint32)
        reserved, written = head[68:68 + n_chunks_max].sum(), head[68 + n_chunks_max:68 + 2 * n_chunks_max].sum()
    else:                                                # tile-local layout: the directory words carry the run lengths
        def records(g):
            out_ = torch.zeros(rows, 2, device="cuda")
            ws_ = torch.zeros(gb.backward_workspace_bytes(B, L, rows), dtype=torch.uint8, device="cuda")
            gb.grid_backward_binned(dev(g), dev(x01), dev(offsets), out_, cnt, B, B, L, L, S, H, ws_)
            tiles = (B + fill_tile - 1) // fill_tile
            head_bytes = (68 + 4 * n_chunks_max + 2) * 4
            head_bytes += (4 - (head_bytes // 4) % 4) % 4 * 4
            at = head_bytes + tiles * L * region * 10
            d = host(ws_[at:at + n_chunks_max * tiles * 4]).view(np.uint32)
            return int((d >> 16).sum())
        written, reserved = records(grad), records(np.ones_like(grad))
    assert 0 < written < 0.85 * reserved


def test_fused_training_is_bitwise_reproducible(lib):
    """Same seed, two runs of the default single-GPU engine (graphs, side/aux streams, fused Adam): identical parameters
    bit for bit.  No float atomics feed back into the state: table sums are 64-bit integers, weight gradients are
    reduced in a fixed order, the density mean is added up in block order."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    out = []
    for _ in range(2):
        torch.manual_seed(0)
        opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True)
        data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
        eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
        assert eng.fuse_adam and eng.use_graph
        for _ in range(40):
            eng.train_step()
        torch.cuda.synchronize()
        out.append((eng.table.clone(), eng.w_flat.clone(), eng.model.density_bitfield.clone(), int(eng.samples_seen)))
    assert out[0][3] == out[1][3]
    assert torch.equal(out[0][2], out[1][2])
    assert torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][0], out[1][0])


def test_step_groups_of_any_length_train_the_same_bits(lib):
    """train() replays consecutive regular steps from multi-step graphs: groups of 2 / 4 / 8 captured on demand, or, after
    precapture_groups(), of any length.  Either way the parameters after 40 steps are the bits of 40 single steps."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    out = []
    for mode in ("single", "groups", "precaptured"):
        torch.manual_seed(0)
        opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True, group_steps=15)
        data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
        eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
        if mode == "single":
            for _ in range(40):
                eng.train_step()
        else:
            eng.train(4)
            if mode == "precaptured":
                assert eng.precapture_groups() > 0
            eng.train(36)
        torch.cuda.synchronize()
        assert eng.global_step == 40
        out.append((eng.table.clone(), eng.w_flat.clone(), int(eng.samples_seen)))
    for other in out[1:]:
        assert other[2] == out[0][2]
        assert torch.equal(other[1], out[0][1]) and torch.equal(other[0], out[0][0])


def test_work_done_ahead_of_a_grid_refresh_trains_the_same_bits(lib, monkeypatch):
    """In front of a density-grid refresh the side stream draws the next batch, runs the grid-independent part of its march
    and draws the refresh's cells (steady state).  None of it may change a bit of the result."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    out = []
    for ahead in ("0", "1"):
        monkeypatch.setenv("NGP_SPLIT_MARCH", ahead)
        torch.manual_seed(0)
        opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True, update_extra_interval=4)
        data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
        eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
        eng.train(60)                   # 15 refreshes, then steady state
        eng.train(41)                   # groups of 3 between refreshes, one refresh step in between each
        for _ in range(9):
            eng.train_step()            # and the single-step path
        torch.cuda.synchronize()
        assert (eng._refresh_head_step > 100) == (ahead == "1")
        out.append((eng.table.clone(), eng.w_flat.clone(), eng.model.density_bitfield.clone(), int(eng.samples_seen)))
    assert out[0][3] == out[1][3]
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][0], out[1][0])


def test_passenger_launches_equal_their_separate_forms(lib, orc):
    """The fused step's two launch fusions against the separate kernels, bit for bit:
    ngp_x_mlp_forward_step_begin == ngp_x_step_begin (scalars + record-offset scan) followed by ngp_x_mlp_forward, and
    ngp_x_grid_backward_binned_apply_mlp == ngp_x_mlp_reduce_dw (+ Adam + operand image) and ngp_x_grid_backward_binned_apply."""
    rng = np.random.default_rng(31)
    gb, e, mb = lib.gridencoder_backend, lib.engine_backend, lib.mlp_backend
    L, H, B = 16, 16, 20000
    offsets, scale = orc.grid_offsets(desired_resolution=2048)
    S, rows = float(np.log2(scale)), int(offsets[-1])
    off = dev(offsets)
    xyz = dev(rng.uniform(-0.9, 0.9, (B, 3)).astype(np.float32))
    table0 = dev(rng.uniform(-1e-2, 1e-2, (rows, 2)).astype(np.float32))
    dirs = dev(rng.normal(size=(B, 3)).astype(np.float32))
    dsigma, drgb = dev((rng.normal(size=B) * 1e-3).astype(np.float32)), dev((rng.normal(size=(B, 3)) * 1e-3).astype(np.float32))
    cnt = torch.tensor([B, B, 0, 0], dtype=torch.int32, device="cuda")
    shapes = [(64, 32), (64, 64), (16, 64), (64, 31), (64, 64), (3, 64)]
    g = torch.Generator().manual_seed(5)
    flat0 = torch.cat([(torch.randn(o, i, generator=g) * (2.0 / i) ** 0.5).reshape(-1) for o, i in shapes]).cuda()

    def views(flat):
        out, at = [], 0
        for o, i in shapes:
            out.append(flat[at:at + o * i].view(o, i))
            at += o * i
        return out
    results = []
    for fused in (False, True):
        table, t_m, t_v = table0.clone(), torch.zeros_like(table0), torch.zeros_like(table0)
        w_flat, w_grad = flat0.clone(), torch.zeros_like(flat0)
        w_m, w_v = torch.zeros_like(flat0), torch.zeros_like(flat0)
        W, dws = views(w_flat), views(w_grad)
        image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
        mb.prepare(W, image)
        enc, x01 = torch.empty(L, B, 2, device="cuda"), torch.empty(B, 3, device="cuda")
        denc = torch.empty(L, B, 2, device="cuda")
        sigma, rgb = torch.empty(B, device="cuda"), torch.empty(B, 3, device="cuda")
        ws = torch.zeros(gb.backward_workspace_bytes(B, L, rows), dtype=torch.uint8, device="cuda")
        ws_mlp = torch.empty(mb.backward_workspace_bytes(B), dtype=torch.uint8, device="cuda")
        ctr = torch.tensor([7], dtype=torch.int32, device="cuda")
        hyper, loss = torch.zeros(4, device="cuda"), torch.full((1,), 3.0, device="cuda")
        seen = torch.tensor([100], dtype=torch.int64, device="cuda")
        gb.grid_backward_binned_prepare(None, 0.0, off, rows, cnt, B, L, L, S, H, ws, merge_max_res=414, stage=1)
        e.grid_encode_forward_slab(xyz, 1.0, table, off, enc, x01, cnt, B, B, L, L, S, H, binned_workspace=ws)
        begin = (ctr, hyper, 1e-2, 300.0, 0.9, 0.999, loss, seen, cnt, ws, L, rows, True)
        if fused:
            mb.forward(enc, B, dirs, cnt, B, image, sigma, rgb, step_begin=begin)
        else:
            e.step_begin(*begin[:9], binned_workspace=ws, L=L, n_rows_total=rows, single_segment=True)
            mb.forward(enc, B, dirs, cnt, B, image, sigma, rgb)
        mb.backward(enc, B, dirs, dsigma, drgb, cnt, B, image, 1024.0, denc, None, ws_mlp)
        table_adam = (table, t_m, t_v, hyper, 0.9, 0.999, 1e-15)
        mlp_adam = (w_flat, w_grad, w_m, w_v, hyper, 0.9, 0.999, 1e-15)
        if fused:
            gb.grid_backward_binned_apply(denc, x01, off, None, cnt, B, B, L, L, S, H, ws, adam=table_adam,
                                          mlp_tail=(B, 1024.0, dws, ws_mlp, mlp_adam, image))
        else:
            mb.reduce_dw(B, 1024.0, dws, ws_mlp, adam=mlp_adam, image=image)
            gb.grid_backward_binned_apply(denc, x01, off, None, cnt, B, B, L, L, S, H, ws, adam=table_adam)
        torch.cuda.synchronize()
        results.append(dict(sigma=sigma, rgb=rgb, hyper=hyper, loss=loss, seen=seen, ctr=ctr, table=table, t_m=t_m, t_v=t_v,
                            w_flat=w_flat, w_grad=w_grad, w_m=w_m, w_v=w_v, image=image))
    a, b = results
    assert int(a["ctr"]) == 8 and float(a["loss"]) == 0.0 and int(a["seen"]) == 100 + B and float(a["hyper"][0]) > 0
    assert float(a["w_grad"].abs().max()) > 0 and not torch.equal(a["table"], table0)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_bf16_wire_gradient_store_and_adam(lib, orc):
    """Data-parallel wire format: the overwrite-mode reduction can store the table gradient as bfloat16 (round to
    nearest even == torch's conversion of the f32 result, bit for bit), and Adam reads it as if it were widened."""
    rng = np.random.default_rng(21)
    L, H, B = 16, 16, 20000
    offsets, scale = orc.grid_offsets(desired_resolution=2048)
    S, rows = float(np.log2(scale)), int(offsets[-1])
    x01 = dev(rng.uniform(0, 1, (B, 3)).astype(np.float32))
    grad = dev(rng.normal(size=(L, B, 2)).astype(np.float32))
    gb, e = lib.gridencoder_backend, lib.engine_backend
    cnt = torch.tensor([B, B, 0, 0], dtype=torch.int32, device="cuda")
    outs, off = [], dev(offsets)
    for dtype in (torch.float32, torch.bfloat16):
        ws = torch.zeros(gb.backward_workspace_bytes(B, L, rows), dtype=torch.uint8, device="cuda")
        out = torch.full((rows, 2), 7.0, dtype=dtype, device="cuda")          # every row must be overwritten
        gb.grid_backward_binned_prepare(x01, 0.0, off, rows, cnt, B, L, L, S, H, ws, single_segment=True)
        gb.grid_backward_binned_apply(grad, x01, off, out, cnt, B, B, L, L, S, H, ws, overwrite=True)
        outs.append(out)
    g32, g16 = outs
    assert float(g32.abs().max()) > 0 and float((g32 == 0).float().mean()) > 0.01
    assert torch.equal(g16.view(torch.int16), g32.to(torch.bfloat16).view(torch.int16))
    with pytest.raises(RuntimeError, match="overwrite"):
        gb.grid_backward_binned_apply(grad, x01, off, g16, cnt, B, B, L, L, S, H, ws)

    torch.manual_seed(3)
    n, nb = rows * 2, 1027
    hyper = torch.tensor([1e-2, 1 - 0.9, 1 / np.sqrt(1 - 0.999), 0.0], dtype=torch.float32, device="cuda")
    res = []
    for g in (g16, g16.float()):
        p, m, v = torch.randn(n, device="cuda", generator=torch.Generator("cuda").manual_seed(5)).view(rows, 2), \
            torch.zeros(rows, 2, device="cuda"), torch.zeros(rows, 2, device="cuda")
        pb, gb_, mb_, vb = torch.ones(nb, device="cuda"), torch.full((nb,), 0.5, device="cuda"), \
            torch.zeros(nb, device="cuda"), torch.zeros(nb, device="cuda")
        e.adam_step_dev2((p, g, m, v, False), (pb, gb_, mb_, vb, True), hyper, 0.9, 0.999, 1e-15)
        assert float(gb_.abs().max()) == 0.0                                   # b's gradient zeroed as asked
        res.append((p, m, v, pb))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.equal(g16.view(torch.int16), g32.to(torch.bfloat16).view(torch.int16))     # a's gradient untouched


@pytest.mark.parametrize("graph", [False, True])
def test_operand_image_follows_the_weights_without_a_prepare_pass(lib, graph):
    """The fused step's weight tail (dW reduction + Adam) also rewrites each weight's two entries of the f16 MFMA operand
    image; after any number of steps the image must be what ngp_x_mlp_prepare builds from the current weights."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True, capture_graph=graph)
    data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
    eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
    assert eng.fuse_adam
    w0 = eng.w_flat.clone()
    for _ in range(7):          # crosses no grid refresh after the first step, so no prepare call in between
        eng.train_step()
    torch.cuda.synchronize()
    assert float((eng.w_flat - w0).abs().max()) > 1e-3
    fresh = torch.zeros_like(eng.mlp_image)
    lib.mlp_backend.prepare(eng.weights, fresh)
    assert torch.equal(fresh, eng.mlp_image)


@pytest.mark.parametrize("T_thresh,random_bg", [(1e-4, True), (1e-8, False), (0.3, True)])
def test_fused_compositor_step_is_the_two_calls(lib, T_thresh, random_bg):
    """ngp_x_composite_mse_train = forward + MSE backward in one launch: ray totals, loss and both gradients must be the
    bits the two separate entry points produce (early stops included: T_thresh 0.3 stops most rays early)."""
    rng = np.random.default_rng(31)
    N = 1700
    sig, rgb, ts, rays, M = synth_samples(rng, N)
    rays[5, 1] = 0                                  # an empty ray
    gt = dev(rng.uniform(0, 1, (N, 4)).astype(np.float32))
    bg = dev(rng.uniform(0, 1, (N, 3)).astype(np.float32)) if random_bg else None
    e = lib.engine_backend
    dsig, drgb, dts, drays = dev(sig), dev(rgb), dev(ts), dev(rays)

    def buffers():
        return (torch.full((N,), 9.0, device="cuda"), torch.full((N,), 9.0, device="cuda"),
                torch.full((N, 3), 9.0, device="cuda"), torch.full((M,), 9.0, device="cuda"),
                torch.full((M, 3), 9.0, device="cuda"), torch.zeros(1, device="cuda"))
    ws_a, dep_a, img_a, gs_a, gc_a, lo_a = buffers()
    w = torch.empty(M, device="cuda")
    e.composite_rays_train_forward(dsig, drgb, dts, drays, M, N, T_thresh, w, ws_a, dep_a, img_a)
    e.composite_mse_backward(gt, bg, 1.0, dsig, drgb, dts, drays, ws_a, dep_a, img_a, M, N, T_thresh, gs_a, gc_a, lo_a)
    ws_b, dep_b, img_b, gs_b, gc_b, lo_b = buffers()
    e.composite_mse_train(gt, bg, 1.0, dsig, drgb, dts, drays, M, N, T_thresh, ws_b, dep_b, img_b, gs_b, gc_b, lo_b)
    for a, b in ((ws_a, ws_b), (dep_a, dep_b), (img_a, img_b), (gs_a, gs_b), (gc_a, gc_b)):
        assert torch.equal(a, b)
    # (the loss value is a float atomic per workgroup: same terms, arrival order not fixed)
    np.testing.assert_allclose(float(lo_a), float(lo_b), rtol=1e-6)
    live = gs_a[gs_a != 9.0]                       # rows of some ray (the kernels write every sample of a live ray)
    assert float(lo_a) > 0 and live.numel() > 0 and float(live.abs().sum()) > 0


def test_fast_evaluation_renders_what_the_inference_loop_renders(lib):
    """FusedTrainer.render_rays (training kernels, forward only, adaptive block size for coherent image rays) against the
    reference-shaped alive-ray loop on the same model: same image to 1e-3, same PSNR; training is undisturbed."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=500, fused_mlp=True)
    dev_ = torch.device("cuda")
    data = SyntheticDataset(opt, dev_, "train", n_views=8, H=128, W=128)
    val = SyntheticDataset(opt, dev_, "val", n_views=2, H=160, W=160)
    eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 64)   # small arena: blocks shrink
    eng.train(300)
    d = val.view(0)
    eng.model.eval()
    with torch.no_grad():
        ref = eng.model.render(d["rays_o"], d["rays_d"], bg_color=0, perturb=False)["image"]
    img, overflow = eng.render_rays(d["rays_o"].contiguous(), d["rays_d"].contiguous(), 0.0)
    assert not overflow
    assert float((img - ref).abs().max()) < 2e-3 and float(ref.max()) > 0.2
    assert abs(eng.evaluate(val, fast=True) - eng.evaluate(val, fast=False)) < 0.02
    before = eng.global_step
    eng.train(5)
    assert eng.global_step == before + 5 and torch.isfinite(eng.loss).all()


@pytest.mark.parametrize("prefetch", [True, False])
def test_adaptive_ray_batches_follow_the_reference_rule(lib, prefetch):
    """`--adaptive_num_rays` on the device (train_utils.py:563-564): every batch gets round(num_points / samples * rays) rays
    of the previous batch's counts, nothing is read back by the step; samples per step settle at num_points.
    prefetch off = ONE ray slot: the previous batch's ray count and the new one are the same device word (the sampler's
    workgroups must all see the same count: adaptive_live_kernel)."""
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    from raw_ngp_amd.nerf.engine import FusedTrainer
    dev = torch.device("cuda")
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=400, adaptive_num_rays=True, num_points=2 ** 15, max_ray_batch=8192,
                  prefetch_march=prefetch)
    data = SyntheticDataset(opt, dev, "train", n_views=6, H=96, W=96)
    ft = FusedTrainer(opt, NeRFNetwork(opt), data, device=dev)
    assert ft.N == 8192 and len(ft.slots) == (2 if prefetch else 1)
    hist = []
    for it in range(300):
        ft.train_step()
        slot = ft.slots[it % len(ft.slots)]
        hist.append((int(slot.live), int(slot.arena.counter[0])))         # (rays, samples) of the batch just trained on
    assert hist[0][0] == 1024                                             # the first batch: opt.num_rays
    for (r0, s0), (r1, _) in zip(hist[:-1], hist[1:]):                    # the rule, batch after batch
        assert r1 == min(max(int(round(opt.num_points / max(s0, 1) * r0)), 1), ft.N), (r0, s0, r1)
    late = np.array([s for _, s in hist[-60:]], dtype=np.float64)
    assert abs(late.mean() / opt.num_points - 1) < 0.1, late.mean()       # samples per step ~ num_points
    assert len({r for r, _ in hist[-60:]}) > 1 and all(r != 1024 for r, _ in hist[-60:])   # the batch size moved and keeps adjusting
    assert int(ft.rays_seen) == sum(r for r, _ in hist)
    assert np.isfinite(float(ft.loss)) and torch.isfinite(ft.table).all()
    # parked ray slots carry nothing: no samples, no image -- and exactly the slots behind live[0] are parked (every
    # workgroup of the sampler used the same count)
    n_live = hist[-1][0]
    last = ft.slots[299 % len(ft.slots)]
    assert int(last.arena.rays[n_live:, 1].sum()) == 0
    parked = (last.rays_o[:, 2] == 1e6)
    assert int(parked.sum()) == ft.N - n_live and not bool(parked[:n_live].any())


def test_adaptive_ray_batches_with_the_hdr_loss(lib):
    """HDR loss + adaptive ray batches: parked ray slots carry view index -1; their exposure is written by the sampler
    (1.0), no gather with a negative index happens, the loss stays finite and the rule still holds."""
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    from raw_ngp_amd.nerf.engine import FusedTrainer
    dev = torch.device("cuda")
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=400, adaptive_num_rays=True, num_points=2 ** 15, max_ray_batch=4096,
                  image_mode="HDR")
    data = SyntheticDataset(opt, dev, "train", n_views=6, H=96, W=96)
    data.exposures = torch.tensor([0.5, 1.0, 2.0, 0.5, 1.0, 2.0], device=dev)
    ft = FusedTrainer(opt, NeRFNetwork(opt), data, device=dev)
    for it in range(40):
        ft.train_step()
    torch.cuda.synchronize()
    slot = ft.slots[39 % len(ft.slots)]
    n_live = int(slot.live)
    assert 1 <= n_live < ft.N
    view = slot.index[:, 0]
    assert bool((view[n_live:] == -1).all()) and bool((view[:n_live] >= 0).all())
    assert torch.equal(slot.exposure[:n_live], data.exposures[view[:n_live].long()])
    assert bool((slot.exposure[n_live:] == 1.0).all())
    assert np.isfinite(float(ft.loss)) and torch.isfinite(ft.table).all()


def test_grid_kernels_at_their_boundaries_stay_inside_their_buffers(lib, orc):
    """The arena exactly full (B == capacity), samples on the faces, edges and corners of [0,1]^3 (first and last cell of
    every level, including the last row of the last level's table) and a few outside: forward (+ Jacobian, + counting),
    binned fill + reduce.  Every output lives in one allocation with guard words between the tensors; the guards must come
    back untouched and the results must match the oracle."""
    rng = np.random.default_rng(14)
    e, gb = lib.engine_backend, lib.gridencoder_backend
    L, H, bound = 16, 16, 1.0
    offsets, scale = orc.grid_offsets(desired_resolution=2048 * bound)
    S, rows = float(np.log2(scale)), int(offsets[-1])
    cap = B = 4096 + 512                                         # not a multiple of the 512-sample fill tile's double
    x01 = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    corners = np.array([[a, b, c] for a in (0.0, 1.0) for b in (0.0, 1.0) for c in (0.0, 1.0)], np.float32)
    x01[:8] = corners
    x01[8:16] = np.nextafter(corners, 0.5).astype(np.float32)     # one ulp inside
    x01[16:24] = corners * np.float32(1 - 2.0 ** -12)             # inside the last cell of the finest level
    x01[24] = [1.0, 0.5, 0.25]
    x01[25] = [np.nextafter(np.float32(1), np.float32(2)), 0.5, 0.5]   # just outside -> zeros
    x01[26] = [-1e-7, 0.5, 0.5]
    x01[-1] = [1.0, 1.0, 1.0]                                     # the last slot of the arena
    xyz = (x01 * np.float32(2 * bound) - np.float32(bound)).astype(np.float32)
    x01 = ((xyz + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)      # what the kernel will compute
    table = rng.uniform(-1, 1, (rows, 2)).astype(np.float32)
    denc = rng.normal(size=(L, cap, 2)).astype(np.float32)
    ws_bytes = gb.backward_workspace_bytes(cap, L, rows)
    G = 1024                                                      # guard floats
    sizes = dict(enc=L * cap * 2, x01=cap * 3, jac=L * cap * 6, grad=rows * 2)
    flat = torch.full((sum(sizes.values()) + G * (len(sizes) + 1),), 123.0, device="cuda")
    views, at = {}, G
    for k, n in sizes.items():
        views[k] = flat[at:at + n]
        at += n + G
    guard = torch.ones_like(flat, dtype=torch.bool)
    at = G
    for n in sizes.values():
        guard[at:at + n] = False
        at += n + G
    ws_flat = torch.full((ws_bytes + 8192,), 0x5A, dtype=torch.uint8, device="cuda")
    ws = ws_flat[4096:4096 + ws_bytes]                            # (4096-byte offset keeps the 16-byte alignment)
    enc, xo, jac = views["enc"].view(L, cap, 2), views["x01"].view(cap, 3), views["jac"].view(L, cap, 3, 2)
    grad = views["grad"].view(rows, 2)
    grad.zero_()
    cnt = torch.tensor([B, B, 0, 0], dtype=torch.int32, device="cuda")
    d_off = dev(offsets)
    gb.grid_backward_binned_prepare(None, 0.0, d_off, rows, cnt, cap, L, L, S, H, ws, merge_max_res=414, stage=1)
    e.grid_encode_forward_slab(dev(xyz), bound, dev(table), d_off, enc, xo, cnt, cap, cap, L, L, S, H, binned_workspace=ws,
                               dydx=jac)
    gb.grid_backward_binned_prepare(None, 0.0, d_off, rows, cnt, cap, L, L, S, H, ws, stage=2)
    gb.grid_backward_binned_apply(dev(denc), xo, d_off, grad, cnt, cap, cap, L, L, S, H, ws)
    torch.cuda.synchronize()
    assert torch.all(flat[guard] == 123.0), "a kernel wrote outside its output tensor"
    assert torch.all(ws_flat[:4096] == 0x5A) and torch.all(ws_flat[4096 + ws_bytes:] == 0x5A), "workspace overrun"
    ref, ref_j = orc.grid_encode_forward(x01, table, offsets, B, 3, 2, L, L, S, H, True)
    np.testing.assert_allclose(host(enc), ref, rtol=1e-6, atol=1e-6)
    assert np.all(host(enc)[:, 25] == 0) and np.all(host(enc)[:, 26] == 0)
    np.testing.assert_allclose(host(jac), ref_j.reshape(B, L, 3, 2).transpose(1, 0, 2, 3), rtol=1e-5,
                               atol=1e-5 * np.abs(ref_j).max())
    ref_g, _ = orc.grid_encode_backward(denc, x01, table, offsets, B, 3, 2, L, L, S, H)
    np.testing.assert_allclose(host(grad), ref_g, rtol=1e-4, atol=2e-5 * np.abs(ref_g).max())
    last = int(offsets[-1]) - 1
    assert np.abs(ref_g[int(offsets[-2]):]).sum() > 0             # the last level received gradient
    assert host(grad)[last].tolist() == pytest.approx(ref_g[last].tolist(), rel=1e-4, abs=1e-6)


def test_compositor_lists_the_samples_in_front_of_the_early_stop(lib, orc):
    """ngp_x_composite_mse_train_idx: per ray the number of samples up to and including the one that drives the
    transmittance below T_thresh (what the oracle's compositor uses), their indices in ray order, and their total -- and the
    output gradients behind the list are exactly zero."""
    from test_oracle_raymarching import synth_samples
    e = lib.engine_backend
    rng = np.random.default_rng(17)
    N = 3000
    sig, rgb, ts, rays, M = synth_samples(rng, N)
    sig *= 6.0                                                   # dense enough for early stops
    T_thresh = 1e-3
    w_ref, *_ = orc.composite_rays_train_forward(sig, rgb, ts, rays, M, N, T_thresh)
    gt = rng.uniform(0, 1, (N, 4)).astype(np.float32)
    outs = [torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")]
    ds, dc, loss = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda"), torch.zeros(1, device="cuda")
    live_n = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    live_idx = torch.full((M,), -1, dtype=torch.int32, device="cuda")
    live_count = torch.zeros(1, dtype=torch.int32, device="cuda")
    e.composite_mse_train(dev(gt), None, 1.0, dev(sig), dev(rgb), dev(ts), dev(rays), M, N, T_thresh, *outs, ds, dc, loss,
                          live=(live_n, live_idx, live_count))
    # reference: replay the stop rule per ray (alpha, transmittance in float32 like the kernels: the stop index may differ by
    # one where T crosses the threshold within rounding -- none of the 3000 rays does here, checked below through the gradients)
    want_n = np.zeros(N, np.int64)
    for n in range(N):
        o, c = rays[n]
        T = np.float32(1.0)
        used = c
        for k in range(c):
            a = np.float32(1.0) - np.exp(np.float32(-sig[o + k] * ts[o + k, 1]), dtype=np.float32)
            T = np.float32(T * (np.float32(1.0) - a))
            if T < T_thresh:
                used = k + 1
                break
        want_n[n] = used
    got_n = host(live_n).astype(np.int64)
    assert np.abs(got_n - want_n).max() <= 1 and (got_n != want_n).mean() < 0.01
    assert 0.3 * M < got_n.sum() < 0.95 * M                       # stops do happen, and not everywhere
    assert int(live_count[0]) == got_n.sum()
    want_idx = np.concatenate([rays[n, 0] + np.arange(got_n[n]) for n in range(N)])
    np.testing.assert_array_equal(host(live_idx)[:len(want_idx)], want_idx)
    dead = np.ones(M, bool)
    dead[want_idx] = False
    assert np.all(host(ds)[dead] == 0.0) and np.all(host(dc)[dead] == 0.0)
    assert np.count_nonzero(host(ds)[~dead]) > 0.9 * (~dead).sum()


def test_table_backward_over_a_sample_list(lib, orc):
    """ngp_x_grid_backward_binned_apply_mlp_list: fill + reduce over a list of samples (positions by sample, gradient slab in
    list order) against the same call over all samples with zero gradients at the unlisted ones."""
    gb, mb = lib.gridencoder_backend, lib.mlp_backend
    rng = np.random.default_rng(23)
    L, H, B = 16, 16, 20000
    offsets, scale = orc.grid_offsets(desired_resolution=2048)
    S, rows = float(np.log2(scale)), int(offsets[-1])
    o = rng.uniform(0.1, 0.9, (400, 1, 3))
    d = rng.normal(size=(400, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x01 = np.clip(o + d * (np.arange(50)[None, :, None] * 0.0017), 0.001, 0.999).reshape(-1, 3).astype(np.float32)
    live = np.ones(B, bool)
    live.reshape(400, 50)[:, 33:] = False                         # every ray's tail is dead
    idx = np.flatnonzero(live).astype(np.int32)
    g_full = rng.normal(size=(L, B, 2)).astype(np.float32)
    g_full[:, ~live] = 0.0
    g_list = np.zeros_like(g_full)
    g_list[:, :len(idx)] = g_full[:, idx]
    W = [torch.zeros(o_, i_, device="cuda") for o_, i_ in [(64, 32), (64, 64), (16, 64), (64, 31), (64, 64), (3, 64)]]
    mws = torch.zeros(mb.backward_workspace_bytes(B), dtype=torch.uint8, device="cuda")

    def run(grad, sample_index, count):
        out = torch.zeros(rows, 2, device="cuda")
        ws = torch.zeros(gb.backward_workspace_bytes(B, L, rows), dtype=torch.uint8, device="cuda")
        cnt = torch.tensor([count, count, 0, 0], dtype=torch.int32, device="cuda")
        gb.grid_backward_binned_prepare(None, 0.0, dev(offsets), rows, cnt, B, L, L, S, H, ws, stage=1)
        dws = [torch.empty_like(w) for w in W]
        gb.grid_backward_binned_apply(dev(grad), dev(x01), dev(offsets), out, cnt, B, B, L, L, S, H, ws,
                                      mlp_tail=(B, 1.0, dws, mws, None, None), sample_index=sample_index)
        return host(out)
    if gb.backward_needs_counts(B, L, dev(offsets)):
        pytest.skip("global-bins layout: no sample lists")
    full = run(g_full, None, B)
    listed = run(g_list, dev(idx), len(idx))
    ref, _ = orc.grid_encode_backward(g_full, x01, np.zeros((rows, 2), np.float32), offsets, B, 3, 2, L, L, S, H, None, 0, False, 0)
    scale_ = np.abs(ref).max()
    np.testing.assert_allclose(full, ref, rtol=1e-4, atol=2e-5 * scale_)
    np.testing.assert_allclose(listed, full, rtol=1e-5, atol=1e-6 * scale_)


def test_step_over_the_live_sample_list_equals_the_step_over_all_samples(lib, monkeypatch):
    """The fused step's backward over the list of samples in front of the compositor's early stop (default) against the
    same step with the list switched off: identical samples and loss, the same gradient of the table and the MLP weights
    up to summation order -- on a field trained far enough that rays do stop early."""
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=400, fused_mlp=True)
    data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
    warm = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
    warm.train(300)
    torch.cuda.synchronize()
    state = {k: v.clone() for k, v in warm.model.state_dict().items()}
    grid, bits, it = warm.model.density_grid.clone(), warm.model.density_bitfield.clone(), warm.model.iter_density
    res = []
    for live in ("1", "0"):
        monkeypatch.setenv("NGP_LIVE_LIST", live)
        torch.manual_seed(1)
        # (a high threshold: after 300 steps on this small scene no ray reaches the default 1e-4 yet)
        o2 = Options(bound=1.0, num_rays=1024, iters=400, fused_mlp=True, capture_graph=False, fuse_adam=False, T_thresh=0.2)
        model = NeRFNetwork(o2).cuda()
        model.load_state_dict(state)
        model.density_grid.copy_(grid)
        model.density_bitfield.copy_(bits)
        model.iter_density = it
        eng = FusedTrainer(o2, model, data, device="cuda", capacity=1024 * 256, seed=5)
        eng.global_step = 1                                      # (not a refresh step)
        eng.step_ctr.fill_(1)
        eng.train_step()
        torch.cuda.synchronize()
        res.append((float(eng.loss), int(eng.slots[1].arena.counter[0]), int(eng.live_count[0]), eng.gflat.float().clone()))
    (la, na, ca, ga), (lb, nb, cb, gb_) = res
    assert na == nb and abs(la - lb) <= 1e-6 * abs(lb)           # the same batch, the same forward (loss: float atomics)
    assert 0 < ca < 0.97 * na and cb == 0, (ca, na)              # rays do stop early; the second run made no list
    assert float(ga.norm()) > 0
    assert float((ga - gb_).norm()) <= 1e-5 * float(gb_.norm()), float((ga - gb_).norm() / gb_.norm())
