"""BASELINE-size checks (config 2: 2^18 samples, 4096 rays, L=16 F=2 T=2^19) through properties that need no oracle run:
the CPU restatement takes minutes at these sizes, the identities below take milliseconds and hold for any input.

  * the encoder is linear in the table, and its table backward is the adjoint of that linear map;
  * the binned (bin -> LDS reduce) and the reference-shaped atomic backward are two evaluations of the same sum;
  * SH values satisfy the addition theorem  sum_m Y_lm(d)^2 = (2l+1)/(4 pi);
  * compositing: weights_sum in [0,1], image / depth are the weighted sums of what went in, d image / d rgb = weight;
  * the fused MLP treats every sample independently (a sample's output does not depend on its batch);
  * the ray-ordered sample layout: offsets are the exclusive scan of the counts, ts increase along a ray.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B_FULL = 1 << 18
N_RAYS = 4096


@pytest.fixture(scope="module")
def be():
    from raw_ngp_amd import _lib
    _lib.load()
    return _lib


@pytest.fixture(scope="module")
def grid():
    from raw_ngp_amd.gridencoder.grid import level_table
    scale = float(np.exp2(np.log2(2048 / 16) / 15))
    offsets = torch.from_numpy(level_table(3, 16, scale, 16, 19)).cuda()
    return offsets, float(np.log2(scale)), int(offsets[-1])


def _ray_ordered_points(gen, n_rays, per_ray):
    """Samples like the march produces them: runs of dt-spaced points along rays through [0,1]^3."""
    o = torch.rand(n_rays, 1, 3, device="cuda", generator=gen) * 0.6 + 0.2
    d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, device="cuda", generator=gen), dim=-1)
    t = torch.arange(per_ray, device="cuda").float()[None, :, None] * (3 ** 0.5 / 1024)
    return (o + d * t).clamp(0, 1).reshape(-1, 3).contiguous()


def _forward(be, x, table, offsets, S, B):
    out = torch.empty(16, B, 2, device="cuda")
    be.gridencoder_backend.grid_encode_forward(x, table, offsets, out, B, 3, 2, 16, 16, S, 16, None, 0, False, 0)
    return out


def test_encoder_is_linear_in_the_table_and_backward_is_its_adjoint(be, grid):
    offsets, S, rows = grid
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = _ray_ordered_points(gen, N_RAYS, B_FULL // N_RAYS)
    B = x.shape[0]
    assert B == B_FULL
    t1 = torch.rand(rows, 2, device="cuda", generator=gen) * 2 - 1
    t2 = torch.rand(rows, 2, device="cuda", generator=gen) * 2 - 1
    e1, e2 = _forward(be, x, t1, offsets, S, B), _forward(be, x, t2, offsets, S, B)
    e12 = _forward(be, x, 0.5 * t1 + t2, offsets, S, B)
    # linearity (8 products per output, each rounded once: a few ulps of the largest term)
    assert float((e12 - (0.5 * e1 + e2)).abs().max()) < 2e-6
    assert float(e1.abs().max()) > 0.5

    g = torch.randn(16, B, 2, device="cuda", generator=gen)
    grads = {}
    for binned in (True, False):
        type(be.gridencoder_backend).use_binned_backward = binned
        try:
            gt = torch.zeros(rows, 2, device="cuda")
            be.gridencoder_backend.grid_encode_backward(g, x, t1, offsets, gt, B, 3, 2, 16, 16, S, 16, None, None, 0,
                                                        False, 0)
        finally:
            type(be.gridencoder_backend).use_binned_backward = True
        grads[binned] = gt
    # adjoint identity  <enc(T), G> = <T, enc^T(G)>  in float64
    prod = e1.double() * g.double()
    lhs, norm = float(prod.sum()), float(prod.pow(2).sum().sqrt())        # a sum of 8.4 M random-sign terms ~ norm
    for binned, gt in grads.items():
        rhs = float((t1.double() * gt.double()).sum())
        assert abs(lhs - rhs) <= 1e-4 * norm, (binned, lhs, rhs, norm)     # (a wrong weight or row shows up as O(norm))
    # two evaluations of the same sums: float atomics round per addition, the binned path once per row
    diff = (grads[True] - grads[False]).abs()
    scale = float(grads[False].abs().max())
    assert float(diff.max()) <= 1e-4 * scale
    assert float(diff.mean()) <= 2e-6 * scale
    # total mass: every sample spreads exactly its gradient (trilinear weights sum to 1) over the level's rows
    for lvl in (0, 7, 15):
        lo, hi = int(offsets[lvl]), int(offsets[lvl + 1])
        np.testing.assert_allclose(float(grads[True][lo:hi].double().sum()), float(g[lvl].double().sum()), rtol=0,
                                   atol=1e-3 * float(g[lvl].double().abs().sum()) ** 0.5)


def test_binned_backward_beyond_one_batch_of_tiles(be, grid):
    """More than 2 048 fill tiles (over a million samples: the tile-local reduce takes its directory in batches) with a
    ragged tail: the binned backward and the reference-shaped atomic backward are two evaluations of the same sums."""
    offsets, S, rows = grid
    gen = torch.Generator(device="cuda").manual_seed(3)
    per_ray = 260
    x = _ray_ordered_points(gen, 4040, per_ray)[: 2051 * 512 + 77]
    B = x.shape[0]
    assert B > 2048 * 512
    t1 = torch.zeros(rows, 2, device="cuda")
    g = torch.randn(16, B, 2, device="cuda", generator=gen)
    g[:, torch.arange(B, device="cuda") % per_ray >= 200] = 0.0           # zero-gradient tails: runs that emit nothing
    grads = {}
    for binned in (True, False):
        type(be.gridencoder_backend).use_binned_backward = binned
        try:
            gt = torch.zeros(rows, 2, device="cuda")
            be.gridencoder_backend.grid_encode_backward(g, x, t1, offsets, gt, B, 3, 2, 16, 16, S, 16, None, None, 0,
                                                        False, 0)
        finally:
            type(be.gridencoder_backend).use_binned_backward = True
        grads[binned] = gt
    diff = (grads[True] - grads[False]).abs()
    scale = float(grads[False].abs().max())
    assert scale > 1.0 and float(diff.max()) <= 2e-4 * scale and float(diff.mean()) <= 4e-6 * scale
    for lvl in (0, 9, 15):
        lo, hi = int(offsets[lvl]), int(offsets[lvl + 1])
        np.testing.assert_allclose(float(grads[True][lo:hi].double().sum()), float(g[lvl].double().sum()), rtol=0,
                                   atol=1e-3 * float(g[lvl].double().abs().sum()) ** 0.5)


def test_sh_addition_theorem_at_full_size(be):
    gen = torch.Generator(device="cuda").manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(B_FULL, 3, device="cuda", generator=gen), dim=-1).contiguous()
    out = torch.empty(B_FULL, 16, device="cuda")
    be.shencoder_backend.sh_encode_forward(d, out, B_FULL, 3, 4, None)
    start = 0
    for l in range(4):
        band = out[:, start:start + 2 * l + 1].double().pow(2).sum(-1)
        np.testing.assert_allclose(band.cpu().numpy(), (2 * l + 1) / (4 * np.pi), rtol=2e-5)
        start += 2 * l + 1


def _ragged_rays(gen, n_rays, total):
    cnt = torch.randint(0, 2 * total // n_rays, (n_rays,), device="cuda", generator=gen, dtype=torch.int32)
    cnt[:5] = 0
    off = torch.cumsum(cnt, 0, dtype=torch.int32) - cnt
    return torch.stack([off, cnt], 1).contiguous(), int(cnt.sum())


def test_compositing_identities_at_full_size(be):
    from raw_ngp_amd import raymarching
    gen = torch.Generator(device="cuda").manual_seed(2)
    rays, M = _ragged_rays(gen, N_RAYS, B_FULL)
    assert 0.8 * B_FULL < M < 1.2 * B_FULL
    ridx = torch.repeat_interleave(torch.arange(N_RAYS, device="cuda"), rays[:, 1].long())
    sig = torch.exp(torch.randn(M, device="cuda", generator=gen) * 2).requires_grad_(True)
    rgb = torch.rand(M, 3, device="cuda", generator=gen).requires_grad_(True)
    dt = 2 * 3 ** 0.5 / 1024
    local = torch.arange(M, device="cuda") - rays[:, 0].long()[ridx]
    ts = torch.stack([0.5 + dt * (local.float() + 1), torch.full((M,), dt, device="cuda")], 1).contiguous()
    w, ws, depth, image = raymarching.composite_rays_train(sig, rgb, ts, rays, 1e-4)
    assert float(ws.detach().min()) >= 0 and float(ws.detach().max()) <= 1 + 1e-5
    assert float(w.detach().min()) >= 0
    zero = torch.zeros(N_RAYS, device="cuda", dtype=torch.float64)
    np.testing.assert_allclose(ws.detach().cpu().numpy(),
                               zero.index_add(0, ridx, w.detach().double()).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(depth.detach().cpu().numpy(),
                               zero.index_add(0, ridx, (w.detach() * ts[:, 0]).double()).cpu().numpy(), rtol=1e-5,
                               atol=1e-6)
    img = torch.zeros(N_RAYS, 3, device="cuda", dtype=torch.float64).index_add(
        0, ridx, (w.detach()[:, None] * rgb.detach()).double())
    np.testing.assert_allclose(image.detach().cpu().numpy(), img.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert torch.all(ws[rays[:, 1] == 0] == 0)                 # empty rays
    # rays stop once the transmittance falls under T_thresh: nothing after that sample contributes
    gi = torch.randn(N_RAYS, 3, device="cuda", generator=gen)
    (image * gi).sum().backward()
    np.testing.assert_allclose(rgb.grad.cpu().numpy(), (w.detach()[:, None] * gi[ridx]).cpu().numpy(), rtol=1e-5,
                               atol=1e-7)
    assert torch.isfinite(sig.grad).all() and float(sig.grad.abs().max()) > 0


def test_fused_mlp_treats_samples_independently(be):
    from test_gpu_fused_mlp import make_weights
    mb = be.mlp_backend
    gen = torch.Generator(device="cuda").manual_seed(3)
    M = B_FULL
    enc = torch.randn(16, M, 2, device="cuda", generator=gen) * 0.3
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda", generator=gen), dim=-1).contiguous()
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(make_weights(seed=4), image)
    sigma, rgb = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
    mb.forward(enc, M, dirs, None, M, image, sigma, rgb)
    assert torch.isfinite(sigma).all() and torch.isfinite(rgb).all()
    # the same samples in a different batch: a slice starting off the 32-sample tile grid, of ragged length
    s, n = 100_003, 777
    enc2 = enc[:, s:s + n].contiguous()
    sigma2, rgb2 = torch.empty(n, device="cuda"), torch.empty(n, 3, device="cuda")
    mb.forward(enc2, n, dirs[s:s + n].contiguous(), None, n, image, sigma2, rgb2)
    assert torch.equal(sigma2, sigma[s:s + n]) and torch.equal(rgb2, rgb[s:s + n])
    # and twice the same call: bit-identical (no atomics, no data-dependent order)
    sigma3, rgb3 = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
    mb.forward(enc, M, dirs, None, M, image, sigma3, rgb3)
    assert torch.equal(sigma3, sigma) and torch.equal(rgb3, rgb)


def test_march_layout_invariants_on_the_procedural_scene(be):
    """4096 rays of 800x800 cameras through the scene's own occupancy grid: ray-ordered offsets, monotone ts, samples
    inside occupied cells, and the three march variants agreeing bit for bit."""
    from raw_ngp_amd import raymarching
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    from raw_ngp_amd.raymarching import MarchArena
    opt = Options(bound=1.0)
    data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=2, H=800, W=800)
    occ = data.occupancy_grid(128, 1.0)                                  # [x, y, z] bool
    coords = occ.nonzero().int()
    grid = torch.zeros(128 ** 3, device="cuda")
    grid[raymarching.morton3D(coords).long()] = 1.0
    bits = raymarching.packbits(grid.view(1, -1), 0.5)
    batch = data.sample_rays(N_RAYS, torch.Generator(device="cuda").manual_seed(5))
    ro, rd = batch["rays_o"].contiguous(), batch["rays_d"].contiguous()
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device="cuda")
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.05)
    noises = torch.rand(N_RAYS, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    rb = be.raymarching_backend
    results = []
    for chain_cap in (0, 1026):
        ar = MarchArena(N_RAYS, 1024, 1 << 20, "cuda", chain_cap=chain_cap)
        rb.march_rays_train_arena(ro, rd, None, bits, 1.0, False, 0.0, 1024, N_RAYS, 1, 128, nears, fars, noises,
                                  ar.t_scratch, ar.capacity, ar.xyzs, ar.dirs, ar.ts, None, ar.rays, ar.counter,
                                  ar.ray_idx, None, ar.chain)
        results.append(ar)
    a, b = results
    M = int(a.counter[0])
    assert M == int(a.counter[1]) == int(b.counter[0]) and int(b.counter[2]) == 0 and M > 20_000
    assert torch.equal(a.rays, b.rays) and torch.equal(a.xyzs[:M], b.xyzs[:M]) and torch.equal(a.ts[:M], b.ts[:M])
    cnt, off = a.rays[:, 1].long(), a.rays[:, 0].long()
    assert torch.equal(off, torch.cumsum(cnt, 0) - cnt) and int(cnt.sum()) == M       # ray-ordered, gap-free
    ridx = torch.repeat_interleave(torch.arange(N_RAYS, device="cuda"), cnt)
    assert torch.equal(ridx.int(), a.ray_idx[:M])
    t = a.ts[:M, 0]
    same_ray = ridx[1:] == ridx[:-1]
    assert torch.all(t[1:][same_ray] > t[:-1][same_ray])                               # strictly increasing
    # ts[:, 0] is the parameter AFTER the step the sample was taken at (raymarching.cu:442-459: t += dt precedes ts[0] = t): position = o + (t - dt) d
    t_at = t - a.ts[:M, 1]
    assert torch.all(t_at >= nears[ridx] - 1e-5) and torch.all(t_at <= fars[ridx] + 1e-5)
    np.testing.assert_allclose(a.xyzs[:M].cpu().numpy(),
                               (ro[ridx] + t_at[:, None] * rd[ridx]).clamp(-1, 1).cpu().numpy(), atol=5e-6)
    cell = ((a.xyzs[:M] + 1) * 0.5 * 128).long().clamp(0, 127)
    assert torch.all(occ[cell[:, 0], cell[:, 1], cell[:, 2]])                          # only occupied cells are sampled
