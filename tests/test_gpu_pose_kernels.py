"""Kernels of the fused step's pose-refinement / light-conditioned / HDR side (csrc/pose_kernels.hip and the additions to
csrc/engine_kernels.hip), each against a restatement of the reference lines it replaces:
  * BARF level window from the step counter         oracle.barf_window       (train_utils.py:488, network.py:99-109)
  * slab encoder with the Jacobian + ray gradients   oracle C kernels         (gridencoder.cu:205-247,352-378; raymarching.py:319-329)
  * HDR loss inside the compositor step              oracle.hdr_loss          (train_utils.py:512-536)
  * per-camera pose gradient                         oracle.pose_gradient     (adjoint of train_utils.py:150-160)
  * se(3) update                                     nerf/pose.py (pinned by the reference fixture pose_lie.npz) under
                                                     torch autograd + torch.optim.Adam + ExponentialLR"""
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


@pytest.mark.parametrize("iters,start,end", [(20000, 0.0, 0.33), (3000, 0.1, 0.5), (500, 0.0, 0.0)])
def test_step_window_matches_float16_restatement(lib, orc, iters, start, end):
    e = lib.engine_backend
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    lw = torch.empty(16, device="cuda")
    flags = torch.zeros(2, dtype=torch.int32, device="cuda")
    steps = sorted(set([0, 1, 2, 17, iters // 50, iters // 20, iters // 10, iters // 6, iters // 3 - 1, iters // 3, iters // 3 + 1,
                        iters // 2, iters - 1, iters, iters + 5] + list(range(100, iters, max(iters // 37, 1)))))
    for s in steps:
        ctr.fill_(s)
        e.step_window(ctr, 0, iters, start, end, 16, lw, flags)
        ref, ann = orc.barf_window(s, iters, start, end, 16)
        np.testing.assert_allclose(host(lw), ref, rtol=0, atol=2e-6, err_msg=f"step {s}")   # cosf vs numpy cos
        f = host(flags)
        assert f[1] == s and f[0] == int(ann < np.float16(end)), (s, f, ann)
    # step_offset 1 = what the training step sees (the reference increments global_step first, train_utils.py:887-888);
    # flags[1] stays the count of steps already done (the pose optimiser's Adam / ExponentialLR index)
    ctr.fill_(41)
    e.step_window(ctr, 1, iters, start, end, 16, lw, flags)
    ref, ann = orc.barf_window(42, iters, start, end, 16)
    np.testing.assert_allclose(host(lw), ref, rtol=0, atol=2e-6)
    assert host(flags)[1] == 41 and host(flags)[0] == int(ann < np.float16(end))


@pytest.mark.parametrize("ann,step", [(0.0, 0), (0.1, 100), (0.33, 330), (1.0, 1000)])
def test_baa_window_on_the_slab_matches_the_reference_network(lib, golden_dir, ann, step):
    """BAA-NGP (network.py:77-97): step_window_baa + slab_window on a level-major slab against what the reference's
    common_forward fed its grid MLP (tests/golden/level_windows.npz: features 0.1 .. 3.2, annealing {0, 0.1, 0.33, 1}) --
    and the backward pass is the blend's adjoint."""
    import os
    e = lib.engine_backend
    g = np.load(os.path.join(golden_dir, "level_windows.npz"))
    feat = g["feat"]                                                      # [4, 32] = (2 l + c)
    B, L = feat.shape[0], 16
    stride = B + 3
    slab = torch.full((L, stride, 2), 9.0, device="cuda")
    slab[:, :B] = dev(np.ascontiguousarray(feat.reshape(B, L, 2).transpose(1, 0, 2)))
    ctr = torch.tensor([step], dtype=torch.int32, device="cuda")
    lw = torch.empty(L, device="cuda")
    e.step_window(ctr, 0, 1000, 0.0, 0.33, L, lw, None, baa=True)
    e.slab_window(slab, stride, L, lw, None, B)
    got = host(slab[:, :B]).transpose(1, 0, 2).reshape(B, 2 * L)
    np.testing.assert_allclose(got, g[f"baangp_{ann}"], rtol=1e-6, atol=1e-6)
    assert torch.all(slab[:, B:] == 9.0)                                  # rows beyond M untouched
    # adjoint: <W x, y> == <x, W^T y>
    rng = np.random.default_rng(int(step))
    x, y = rng.normal(size=(L, stride, 2)).astype(np.float32), rng.normal(size=(L, stride, 2)).astype(np.float32)
    wx, wty = dev(x).clone(), dev(y).clone()
    cnt = torch.tensor([B], dtype=torch.int32, device="cuda")
    e.slab_window(wx, stride, L, lw, cnt, stride)                         # (M_dev clamps the launch to B samples)
    e.slab_window(wty, stride, L, lw, cnt, stride, backward=True)
    lhs = float((wx[:, :B].double() * dev(y)[:, :B].double()).sum())
    rhs = float((dev(x)[:, :B].double() * wty[:, :B].double()).sum())
    np.testing.assert_allclose(lhs, rhs, rtol=1e-5, atol=1e-6)
    assert torch.equal(wx[:, B:], dev(x)[:, B:]) and torch.equal(wty[:, B:], dev(y)[:, B:])


def test_slab_jacobian_and_ray_gradients_match_oracle(lib, orc):
    rng = np.random.default_rng(3)
    e = lib.engine_backend
    N, L, H, bound = 700, 16, 16, 2.0
    sig, rgb, ts, rays, M = synth_samples(rng, N, max_cnt=90)
    rays[3] = [M - 2, 9]                                  # a ray whose samples overflow the arena: ignored
    cap = M + 11
    offsets, scale = orc.grid_offsets(desired_resolution=2048 * bound)
    S = float(np.log2(scale))
    table = rng.uniform(-1, 1, (offsets[-1], 2)).astype(np.float32)
    xyz = rng.uniform(-bound * 1.01, bound * 1.01, (cap, 3)).astype(np.float32)
    x01 = ((xyz + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
    ref_out, ref_jac = orc.grid_encode_forward(x01[:M], table, offsets, M, 3, 2, L, L, S, H, True)
    out = torch.full((L, cap, 2), 7.0, device="cuda")
    jac = torch.full((L, cap, 3, 2), 7.0, device="cuda")
    cnt = torch.tensor([M, 0], dtype=torch.int32, device="cuda")
    e.grid_encode_forward_slab(dev(xyz), bound, dev(table), dev(offsets), out, None, cnt, cap, cap, L, L, S, H, dydx=jac)
    np.testing.assert_allclose(host(out)[:, :M], ref_out, rtol=1e-6, atol=1e-6)
    ref_lm = ref_jac.reshape(M, L, 3, 2).transpose(1, 0, 2, 3)             # reference layout [b, l, d, ch] -> level-major
    np.testing.assert_allclose(host(jac)[:, :M], ref_lm, rtol=1e-5, atol=1e-5 * np.abs(ref_lm).max())
    assert torch.all(jac[:, M:] == 7.0)
    outside = ~np.all((x01[:M] >= 0) & (x01[:M] <= 1), axis=1)
    assert outside.sum() > 0 and np.all(host(jac)[:, :M][:, outside] == 0)  # zero Jacobian outside [0,1]^3
    # the counting variant writes the same Jacobian
    ws = torch.empty(lib.gridencoder_backend.backward_workspace_bytes(cap, L, int(offsets[-1])), dtype=torch.uint8, device="cuda")
    lib.gridencoder_backend.grid_backward_binned_prepare(None, 0.0, dev(offsets), int(offsets[-1]), cnt, cap, L, L, S, H, ws,
                                                         merge_max_res=414, stage=1)
    jac2 = torch.empty_like(jac)
    e.grid_encode_forward_slab(dev(xyz), bound, dev(table), dev(offsets), out, None, cnt, cap, cap, L, L, S, H,
                               binned_workspace=ws, dydx=jac2)
    assert torch.equal(jac2[:, :M], jac[:, :M])

    # ray gradients = segment sums over (encoder input backward / (2 bound)) and ts[:,0] * that + d dirs
    denc = rng.normal(size=(L, cap, 2)).astype(np.float32)
    ddirs = rng.normal(size=(cap, 3)).astype(np.float32)
    gx = orc.grid_input_backward(denc[:, :M], ref_jac, M, 3, 2, L) / (2 * bound)
    ref_o, ref_d = orc.march_rays_train_backward(gx.astype(np.float32), ddirs[:M], ts, rays, N, M)
    go, gd = torch.empty(N, 3, device="cuda"), torch.empty(N, 3, device="cuda")
    e.ray_gradients(dev(denc), jac, cap, L, bound, dev(ddirs), dev(ts), dev(rays), N, M, go, gd)
    scale_o, scale_d = np.abs(ref_o).max(), np.abs(ref_d).max()
    np.testing.assert_allclose(host(go), ref_o, rtol=1e-4, atol=1e-5 * scale_o)
    np.testing.assert_allclose(host(gd), ref_d, rtol=1e-4, atol=1e-5 * scale_d)
    assert np.all(host(go)[3] == 0) and np.all(host(gd)[3] == 0)            # the overflowing ray
    e.ray_gradients(dev(denc), jac, cap, L, bound, None, dev(ts), dev(rays), N, M, go, gd)
    ref_o2, ref_d2 = orc.march_rays_train_backward(gx.astype(np.float32), None, ts, rays, N, M)
    np.testing.assert_allclose(host(gd), ref_d2, rtol=1e-4, atol=1e-5 * scale_d)


def test_orientation_term_and_its_direction_gradient_match_autograd(lib, orc):
    """ngp_x_orientation_term on random slabs against the expression of renderer.py:558-571 in torch (float64): the term,
    d term / d dirs with the normal held constant, the tiny-gradient case of F.normalize, and the term's path to the ray
    directions in ngp_x_ray_gradients_terms (term_weight[i] * dterm_ddirs[i] joins d dirs)."""
    rng = np.random.default_rng(5)
    e = lib.engine_backend
    N, L, bound = 300, 16, 2.0
    sig, rgb, ts, rays, M = synth_samples(rng, N, max_cnt=60)
    cap = M + 7
    dh = rng.normal(size=(L, cap, 2)).astype(np.float32)
    jac = rng.normal(size=(L, cap, 3, 2)).astype(np.float32)
    dirs = (rng.normal(size=(cap, 3)) * rng.uniform(0.5, 2.0, (cap, 1))).astype(np.float32)      # un-normalised
    sigma = np.exp(rng.uniform(-20, 5, cap)).astype(np.float32)
    dh[:, 5] = 0.0                                           # a flat density: normalize(0) = 0, the normal is (.5, .5, .5)
    sigma[6] = 1e-30                                         # |d sigma / d xyz| below normalize's eps
    cnt = torch.tensor([M, 0], dtype=torch.int32, device="cuda")
    term, dterm = torch.full((cap,), 7.0, device="cuda"), torch.full((cap, 3), 7.0, device="cuda")
    e.orientation_term(dev(dh), dev(jac), cap, L, bound, dev(sigma), dev(dirs), cnt, cap, term, dterm_ddirs=dterm)
    assert torch.all(term[M:] == 7.0) and torch.all(dterm[M:] == 7.0)
    d = torch.from_numpy(dirs[:M].astype(np.float64)).requires_grad_(True)
    g = torch.einsum("lmc,lmdc->md", torch.from_numpy(dh[:, :M].astype(np.float64)), torch.from_numpy(jac[:, :M].astype(np.float64)))
    g = g * torch.from_numpy(sigma[:M].astype(np.float64)).clamp(np.exp(-80.0), np.exp(80.0))[:, None] / (2 * bound)
    nrm = (-(g / g.norm(dim=-1, keepdim=True).clamp_min(1e-12)) + 1) / 2
    u = d / d.norm(dim=-1, keepdim=True)
    want = torch.clamp((nrm * -u).sum(-1), max=0.0) ** 2
    want.sum().backward()
    np.testing.assert_allclose(host(term)[:M], want.detach().numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(host(dterm)[:M], d.grad.numpy(), rtol=2e-3, atol=2e-5)
    assert float(want[5]) == pytest.approx(min(0.0, float((0.5 * -u[5]).sum())) ** 2, abs=1e-12)
    assert 0.2 < float((want > 0).double().mean()) < 0.9

    # the term's way to the ray directions
    denc = rng.normal(size=(L, cap, 2)).astype(np.float32)
    ddirs = rng.normal(size=(cap, 3)).astype(np.float32)
    tw = rng.uniform(0, 1, cap).astype(np.float32)
    go, gd = torch.empty(N, 3, device="cuda"), torch.empty(N, 3, device="cuda")
    go2, gd2 = torch.empty(N, 3, device="cuda"), torch.empty(N, 3, device="cuda")
    e.ray_gradients(dev(denc), dev(jac), cap, L, bound, dev(ddirs), dev(ts), dev(rays), N, M, go, gd, terms=(dev(tw), dterm))
    both = ddirs.copy()
    both[:M] += tw[:M, None] * host(dterm)[:M]
    e.ray_gradients(dev(denc), dev(jac), cap, L, bound, dev(both), dev(ts), dev(rays), N, M, go2, gd2)
    assert torch.equal(go, go2)
    np.testing.assert_allclose(host(gd), host(gd2), rtol=1e-5, atol=1e-5 * float(gd2.abs().max()))
    e.ray_gradients(dev(denc), dev(jac), cap, L, bound, dev(ddirs), dev(ts), dev(rays), N, M, go2, gd2)      # without the term
    assert float((gd - gd2).abs().max()) > 1e-2 * float(gd2.abs().max())


@pytest.mark.parametrize("T_thresh,random_bg,weighted", [(1e-4, True, False), (1e-8, False, True)])
def test_hdr_loss_step_matches_restatement(lib, orc, T_thresh, random_bg, weighted):
    rng = np.random.default_rng(21)
    e = lib.engine_backend
    N = 1200
    sig, rgb, ts, rays, M = synth_samples(rng, N, max_cnt=120)
    rgb = (rgb * 2.5).astype(np.float32)                     # HDR radiance: predictions above 1 exist
    gt = rng.uniform(0, 1, (N, 4)).astype(np.float32)
    gt[::5, 3] = 1.0
    bg = rng.uniform(0, 1, (N, 3)).astype(np.float32) if random_bg else None
    bg_const = 0.0 if random_bg else 1.0
    exposure = rng.choice([0.25, 1.0, 4.0], N).astype(np.float32)
    weight = rng.uniform(0, 2, (N, 3)).astype(np.float32) if weighted else None     # lossmult (Bayer mask) x loss_weight
    lossmult = (rng.uniform(0, 1, (N, 3)) < 0.5).astype(np.float32) if weighted else None
    if weighted:
        weight = (lossmult * rng.uniform(0.5, 2, (N, 3))).astype(np.float32)
    inv_norm = 1.0 / (lossmult.sum() if weighted else 3 * N)
    # restatement: composite (oracle) -> mix background -> HDR loss -> composite backward (oracle)
    rw, rws, rdep, rimg = orc.composite_rays_train_forward(sig, rgb, ts, rays, M, N, T_thresh)
    b = bg if random_bg else np.full((N, 3), bg_const, np.float32)
    pred = rimg + (1 - rws[:, None]) * b
    gt_rgb = gt[:, :3] * gt[:, 3:] + b * (1 - gt[:, 3:])
    if weighted:
        loss, g_pred = orc.hdr_loss(pred, gt_rgb, exposure, lossmult, weight / np.maximum(lossmult, 1e-30))
    else:
        loss, g_pred = orc.hdr_loss(pred, gt_rgb, exposure)
    assert (pred * exposure[:, None] >= 1).mean() > 0.02 and (pred * exposure[:, None] < 1).mean() > 0.2
    g_ws = -(g_pred * b).sum(1).astype(np.float32)
    rgs, rgc = orc.composite_rays_train_backward(np.zeros(M, np.float32), g_ws, np.zeros(N, np.float32), g_pred, sig, rgb,
                                                 ts, rays, rws, rdep, rimg, M, N, T_thresh)
    ws, dep, img = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
    gs, gc = torch.full((M,), 7.0, device="cuda"), torch.full((M, 3), 7.0, device="cuda")
    loss_dev = torch.zeros(1, device="cuda")
    e.composite_hdr_train(dev(gt), dev(bg) if random_bg else None, bg_const, dev(exposure),
                          dev(weight) if weighted else None, inv_norm, dev(sig), dev(rgb), dev(ts), dev(rays), M, N, T_thresh,
                          ws, dep, img, gs, gc, loss_dev)
    np.testing.assert_allclose(host(img), rimg, rtol=3e-4, atol=1e-5 + 8 * T_thresh)
    np.testing.assert_allclose(float(loss_dev), loss, rtol=2e-3)
    covered = np.zeros(M, bool)
    for n in range(N):
        if rays[n, 1] and rays[n, 0] + rays[n, 1] <= M:
            covered[rays[n, 0]:rays[n, 0] + rays[n, 1]] = True
    # a prediction within rounding of the clip (pred * exposure ~ 1) may fall on either side: compare in the L2 sense
    dgc = host(gc)[covered] - rgc[covered]
    assert np.linalg.norm(dgc) <= 2e-3 * np.linalg.norm(rgc[covered]) + 1e-9
    dgs = host(gs)[covered] - rgs[covered]
    assert np.linalg.norm(dgs) <= 5e-3 * np.linalg.norm(rgs[covered]) + 1e-9


def test_lit_ray_sampler(lib, orc):
    e = lib.engine_backend
    rng = np.random.default_rng(2)
    V, H, W, N = 7, 20, 24, 5000
    images = torch.from_numpy(rng.integers(0, 256, (V, H, W, 4), dtype=np.uint8)).cuda()
    poses = torch.from_numpy(rng.normal(size=(V, 4, 4)).astype(np.float32)).cuda()
    intr = np.array([30.0, 31.0, W / 2, H / 2])
    ld = torch.from_numpy(rng.normal(size=(V, 3)).astype(np.float32)).cuda()
    o, d = torch.empty(N, 3, device="cuda"), torch.empty(N, 3, device="cuda")
    gt, idx = torch.empty(N, 4, device="cuda"), torch.empty(N, 2, dtype=torch.int32, device="cuda")
    rl = torch.empty(N, 3, device="cuda")
    e.sample_rays(images, poses, intr, N, 99, 5, o, d, gt, None, None, idx, view_ldirs=ld, rays_ldir=rl)
    assert torch.equal(rl, ld[idx[:, 0].long()])
    o2, d2, gt2 = torch.empty_like(o), torch.empty_like(d), torch.empty_like(gt)
    e.sample_rays(images, poses, intr, N, 99, 5, o2, d2, gt2, None, None, None)        # same draws without the lights
    assert torch.equal(o, o2) and torch.equal(d, d2) and torch.equal(gt, gt2)


def test_pose_gradient_matches_restatement(lib, orc):
    e = lib.engine_backend
    rng = np.random.default_rng(8)
    V, W, Hh, N = 13, 40, 30, 4096
    idx = np.stack([rng.integers(0, V, N), rng.integers(0, W * Hh, N)], 1).astype(np.int32)
    idx[idx[:, 0] == 4, 0] = 5                                   # camera 4 has no ray in this batch
    go = rng.normal(size=(N, 3)).astype(np.float32)
    gd = rng.normal(size=(N, 3)).astype(np.float32)
    intr = np.array([55.0, 57.0, W / 2 + 0.3, Hh / 2 - 0.2])
    out = torch.full((V, 12), 7.0, device="cuda")
    e.pose_gradient(dev(idx), dev(go), dev(gd), N, V, W, intr, out)
    ref = orc.pose_gradient(idx, go, gd, V, W, intr).reshape(V, 12)
    np.testing.assert_allclose(host(out), ref, rtol=1e-4, atol=1e-4)
    assert np.all(host(out)[4] == 0)
    out2 = torch.empty_like(out)
    e.pose_gradient(dev(idx), dev(go), dev(gd), N, V, W, intr, out2)
    assert torch.equal(out, out2)                                 # fixed summation order


def test_pose_update_matches_torch_autograd_and_adam(lib):
    """xi -> compose(exp(xi), base) and its Adam update against nerf/pose.py under autograd (the module the per-op
    trainer uses; its exponential map is pinned to the reference by tests/golden/pose_lie.npz)."""
    from raw_ngp_amd.nerf import pose as P
    e = lib.engine_backend
    g = torch.Generator().manual_seed(4)
    V, iters, c_lr = 24, 50, 1e-3
    xi0 = torch.randn(V, 6, generator=g) * 0.3
    xi0[:4] = 0.0                                    # zero corrections (where every camera starts): the series branch
    xi0[4:8] *= 1e-3                                 # tiny angles, still the series branch
    xi0[8] = torch.tensor([2.5, -1.0, 0.5, 0.3, 0.2, -0.1])     # a large rotation
    base = P.se3_to_SE3(torch.randn(V, 6, generator=g) * 0.8)   # arbitrary rigid base poses [V,3,4]
    base[:, :, 3] += torch.randn(V, 3, generator=g)
    xi = xi0.clone().cuda()
    m, v = torch.zeros(V, 6, device="cuda"), torch.zeros(V, 6, device="cuda")
    refined = torch.empty(V, 4, 4, device="cuda")
    flags = torch.tensor([1, 0], dtype=torch.int32, device="cuda")
    gamma = 1e-2 ** (1.0 / iters)
    base_d = base.reshape(V, 12).contiguous().cuda()
    e.pose_update(xi, base_d, None, None, None, None, c_lr, gamma, 0.9, 0.999, 1e-8, refined)
    want = P.compose([P.se3_to_SE3(xi0), base])
    np.testing.assert_allclose(host(refined)[:, :3], want.numpy(), rtol=1e-5, atol=2e-6)
    assert torch.all(refined[:, 3] == torch.tensor([0.0, 0, 0, 1], device="cuda"))
    # torch side: the same parameters under Adam + ExponentialLR, gradients of a random linear functional per step
    p = torch.nn.Parameter(xi0.clone())
    opt = torch.optim.Adam([p], lr=c_lr)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma)
    gxi = torch.empty(V, 6, device="cuda")
    for step in range(6):
        G = torch.randn(V, 3, 4, generator=g)
        G[5] = 0.0                                    # a camera without rays still moves with its momentum
        opt.zero_grad()
        (P.compose([P.se3_to_SE3(p), base]) * G).sum().backward()
        flags[1] = step
        e.pose_update(xi, base_d, G.reshape(V, 12).contiguous().cuda(), flags, m, v, c_lr, gamma, 0.9, 0.999, 1e-8, refined, gxi)
        scale = float(p.grad.abs().max())
        np.testing.assert_allclose(host(gxi), p.grad.numpy(), rtol=2e-4, atol=2e-5 * scale, err_msg=f"step {step}")
        opt.step()
        sched.step()
        np.testing.assert_allclose(host(xi), p.detach().numpy(), rtol=1e-4, atol=2e-6, err_msg=f"step {step}")
        np.testing.assert_allclose(host(refined)[:, :3], P.compose([P.se3_to_SE3(p.detach()), base]).numpy(), rtol=1e-4,
                                   atol=5e-6)
    # flags[0] == 0 (annealing past end_annealing): the parameters stay, the refined poses are still written
    before = xi.clone()
    flags[0] = 0
    e.pose_update(xi, base_d, torch.randn(V, 12, device="cuda"), flags, m, v, c_lr, gamma, 0.9, 0.999, 1e-8, refined)
    assert torch.equal(xi, before)
