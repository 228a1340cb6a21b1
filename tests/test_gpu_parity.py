"""GPU parity: every C-ABI entry point (called through the `_backend` shims, i.e. exactly what the
op wrappers call) against the CPU oracle on the same seeded inputs.

Bars: integer / byte / index outputs bit-exact; float outputs within the tolerance written next
to each check (fp32 kernels; differences come from fma/evaluation order, `__expf` and -- for the
scatter-add -- the order of float atomics)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_raymarching import brick_bitfield, make_rays, synth_samples  # noqa: E402


@pytest.fixture(scope="module")
def be():
    from raw_ngp_amd import _lib
    _lib.load()
    return _lib


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def host(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- grid encoder

GRID_CASES = [
    # D, C, L, H, log2T, desired, gridtype, align, interp, B
    (3, 2, 16, 16, 19, 2048, 0, False, 0, 20000),     # the north-star configuration
    (3, 2, 16, 16, 19, 4096, 0, False, 0, 4097),      # bound 2 table, ragged B
    (3, 2, 8, 16, 19, 2048, 0, False, 0, 5000),       # plumbing config (L=8)
    (3, 2, 5, 16, 17, 128, 0, False, 1, 3000),        # proposal grid, smoothstep
    (2, 4, 6, 4, 9, 64, 0, True, 0, 1000),
    (3, 1, 6, 4, 9, 64, 1, False, 0, 1000),           # tiled
    (3, 8, 4, 4, 10, 32, 0, True, 1, 777),
    (4, 2, 4, 4, 10, 16, 0, False, 0, 500),
    (5, 2, 3, 3, 10, 8, 0, False, 0, 300),
    (3, 16, 3, 4, 9, 16, 0, False, 0, 300),
    (3, 32, 2, 4, 9, 8, 0, False, 0, 200),
]


def grid_setup(orc, D, C, L, H, log2T, desired, B, seed=0):
    rng = np.random.default_rng(seed)
    offsets, scale = orc.grid_offsets(input_dim=D, num_levels=L, level_dim=C, base_resolution=H,
                                      log2_hashmap_size=log2T, desired_resolution=desired)
    S = float(np.log2(scale))
    table = rng.uniform(-1, 1, (offsets[-1], C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2, 0] = -1e-3      # outside -> zeros
    x[3, D - 1] = 1.0001
    x[4] = 0.5           # exactly on cell boundaries of power-of-two levels
    return offsets, S, table, x


@pytest.mark.parametrize("case", GRID_CASES, ids=lambda c: f"D{c[0]}C{c[1]}L{c[2]}g{c[6]}a{int(c[7])}i{c[8]}")
def test_grid_forward_and_jacobian(be, orc, case):
    D, C, L, H, log2T, desired, gridtype, align, interp, B = case
    offsets, S, table, x = grid_setup(orc, D, C, L, H, log2T, desired, B)
    ref, ref_j = orc.grid_encode_forward(x, table, offsets, B, D, C, L, L, S, H, True, gridtype, align, interp)
    out = torch.empty(L, B, C, device="cuda")
    jac = torch.empty(B, L * D * C, device="cuda")
    be.gridencoder_backend.grid_encode_forward(dev(x), dev(table), dev(offsets), out, B, D, C, L, L, S, H, jac,
                                               gridtype, align, interp)
    # identical operation order in both -> agreement to rounding of the last fma
    np.testing.assert_allclose(host(out), ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(host(jac), ref_j, rtol=1e-5, atol=1e-3 * np.abs(ref_j).max())
    # without dy_dx, and with max_level < L (tail pre-zeroed by the caller)
    out2 = torch.zeros(L, B, C, device="cuda")
    be.gridencoder_backend.grid_encode_forward(dev(x), dev(table), dev(offsets), out2, B, D, C, L, L - 1, S, H, None,
                                               gridtype, align, interp)
    np.testing.assert_allclose(host(out2)[:L - 1], ref[:L - 1], rtol=1e-6, atol=1e-6)
    assert torch.all(out2[L - 1] == 0)


@pytest.mark.parametrize("binned", [True, False], ids=["binned", "atomic"])
@pytest.mark.parametrize("case", GRID_CASES[:7], ids=lambda c: f"D{c[0]}C{c[1]}L{c[2]}g{c[6]}")
def test_grid_backward(be, orc, case, binned, monkeypatch):
    """binned = bin -> LDS-reduce scatter (D3 C2 only, what the op wrapper uses); atomic = reference-shaped."""
    monkeypatch.setattr(type(be.gridencoder_backend), "use_binned_backward", binned)
    D, C, L, H, log2T, desired, gridtype, align, interp, B = case
    offsets, S, table, x = grid_setup(orc, D, C, L, H, log2T, desired, B, seed=1)
    rng = np.random.default_rng(2)
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    _, jac = orc.grid_encode_forward(x, table, offsets, B, D, C, L, L, S, H, True, gridtype, align, interp)
    ref_gt, ref_gi = orc.grid_encode_backward(g, x, table, offsets, B, D, C, L, L, S, H, jac, gridtype, align, interp)
    pre = np.random.default_rng(3).normal(size=(int(offsets[-1]), C)).astype(np.float32)   # "+=" semantics
    gt = dev(pre)
    gi = torch.zeros(B, D, device="cuda")
    be.gridencoder_backend.grid_encode_backward(dev(g), dev(x), dev(table), dev(offsets), gt, B, D, C, L, L, S, H,
                                                dev(jac), gi, gridtype, align, interp)
    # float sums in hardware order vs a double-precision sum: error ~ eps * sum|terms|
    scale = np.abs(ref_gt).max()
    np.testing.assert_allclose(host(gt) - pre, ref_gt, rtol=1e-4, atol=2e-6 * max(scale, 1.0) * 8)
    np.testing.assert_allclose(host(gi), ref_gi, rtol=1e-5, atol=1e-4 * np.abs(ref_gi).max())
    # max_level < L leaves the tail levels' rows untouched
    gt2 = torch.zeros(int(offsets[-1]), C, device="cuda")
    be.gridencoder_backend.grid_encode_backward(dev(g), dev(x), dev(table), dev(offsets), gt2, B, D, C, L, L - 1, S, H,
                                                None, None, gridtype, align, interp)
    assert torch.all(gt2[int(offsets[L - 1]):] == 0)
    np.testing.assert_allclose(host(gt2)[:int(offsets[L - 1])], ref_gt[:int(offsets[L - 1])], rtol=1e-4,
                               atol=2e-6 * max(scale, 1.0) * 8)


@pytest.mark.parametrize("D,C,L,H,log2_T,desired", [(3, 2, 6, 4, 9, 48), (2, 2, 8, 4, 10, 256), (3, 4, 4, 8, 12, 64),
                                                     (2, 1, 5, 16, 8, 128)],
                         ids=["D3C2", "D2C2", "D3C4", "D2C1"])
def test_grid_tv_and_wd(be, orc, D, C, L, H, log2_T, desired):
    offsets, S, table, _ = grid_setup(orc, D, C, L, H, log2_T, desired, 16)
    rng = np.random.default_rng(3)
    B = 4000
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    g0 = rng.normal(size=table.shape).astype(np.float32)
    ref = orc.grad_total_variation(x, table, g0, offsets, 1e-2, B, D, C, L, S, H)
    g = dev(g0)
    be.gridencoder_backend.grad_total_variation(dev(x), dev(table), g, dev(offsets), 1e-2, B, D, C, L, S, H, 0, False)
    np.testing.assert_allclose(host(g), ref, rtol=1e-4, atol=1e-5)
    n = int(offsets[-1])
    ref = orc.grad_weight_decay(table, g0, offsets, 0.1, n, C, L)
    g = dev(g0)
    be.gridencoder_backend.grad_weight_decay(dev(table), g, dev(offsets), 0.1, n, C, L)
    np.testing.assert_allclose(host(g), ref, rtol=1e-6, atol=1e-7)


def test_grid_rejects_bad_arguments(be):
    t = torch.zeros(8, 3, device="cuda")
    off = torch.zeros(3, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="C must be"):
        be.gridencoder_backend.grid_encode_forward(t, torch.zeros(8, 3, device="cuda"), off, t, 8, 3, 3, 2, 2, 1.0, 4,
                                                   None, 0, False, 0)
    with pytest.raises(RuntimeError, match="D must be"):
        be.gridencoder_backend.grid_encode_forward(t, torch.zeros(8, 2, device="cuda"), off, t, 8, 6, 2, 2, 2, 1.0, 4,
                                                   None, 0, False, 0)
    with pytest.raises(RuntimeError, match="contiguous"):
        be.gridencoder_backend.grid_encode_forward(t.t(), torch.zeros(8, 2, device="cuda"), off, t, 8, 3, 2, 2, 2, 1.0,
                                                   4, None, 0, False, 0)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        be.gridencoder_backend.grid_encode_forward(t.cpu(), torch.zeros(8, 2, device="cuda"), off, t, 8, 3, 2, 2, 2,
                                                   1.0, 4, None, 0, False, 0)


# ----------------------------------------------------------------------------- SH / freq

@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_forward_backward(be, orc, degree):
    rng = np.random.default_rng(degree)
    B = 5000
    v = rng.normal(size=(B, 3))
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    v[:3] = [[0, 0, 1], [1, 0, 0], [0, -1, 0]]
    ref, ref_j = orc.sh_encode_forward(v, B, 3, degree, True)
    n = degree * degree
    out = torch.empty(B, n, device="cuda")
    jac = torch.empty(B, 3 * n, device="cuda")
    be.shencoder_backend.sh_encode_forward(dev(v), out, B, 3, degree, jac)
    # fp32 Horner vs the oracle's double evaluation: a few ulp of the largest monomial
    np.testing.assert_allclose(host(out), ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(host(jac), ref_j, rtol=1e-5, atol=2e-5)
    out2 = torch.empty(B, n, device="cuda")
    be.shencoder_backend.sh_encode_forward(dev(v), out2, B, 3, degree, None)
    assert torch.equal(out, out2)
    g = rng.normal(size=(B, n)).astype(np.float32)
    ref_gi = orc.sh_encode_backward(g, v, B, 3, degree, ref_j)
    gi = torch.zeros(B, 3, device="cuda")
    be.shencoder_backend.sh_encode_backward(dev(g), dev(v), B, 3, degree, jac, gi)
    np.testing.assert_allclose(host(gi), ref_gi, rtol=1e-4, atol=1e-4 * np.abs(ref_gi).max())


def test_sh_rejects_bad_degree(be):
    t = torch.zeros(4, 3, device="cuda")
    with pytest.raises(RuntimeError, match="degree"):
        be.shencoder_backend.sh_encode_forward(t, torch.zeros(4, 81, device="cuda"), 4, 3, 9, None)


def test_freq(be, orc):
    rng = np.random.default_rng(0)
    B, D, deg = 3000, 3, 6
    C = D + 2 * D * deg
    x = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    ref = orc.freq_encode_forward(x, B, D, deg, C)
    out = torch.empty(B, C, device="cuda")
    be.freqencoder_backend.freq_encode_forward(dev(x), B, D, deg, C, out)
    np.testing.assert_allclose(host(out), ref, rtol=1e-5, atol=2e-6)
    g = rng.normal(size=(B, C)).astype(np.float32)
    ref_gi = orc.freq_encode_backward(g, ref, B, D, deg, C)
    gi = torch.zeros(B, D, device="cuda")
    be.freqencoder_backend.freq_encode_backward(dev(g), dev(ref), B, D, deg, C, gi)
    np.testing.assert_allclose(host(gi), ref_gi, rtol=1e-5, atol=1e-4)


# ----------------------------------------------------------------------------- integer kernels (bit-exact)

def test_morton_packbits_flatten_bit_exact(be, orc):
    H = 128
    g = np.arange(H, dtype=np.int32)
    coords = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    idx = torch.empty(H ** 3, dtype=torch.int32, device="cuda")
    be.raymarching_backend.morton3D(dev(coords), H ** 3, idx)
    assert np.array_equal(host(idx), orc.morton3D(coords))
    back = torch.empty(H ** 3, 3, dtype=torch.int32, device="cuda")
    be.raymarching_backend.morton3D_invert(idx, H ** 3, back)
    assert np.array_equal(host(back), coords)

    rng = np.random.default_rng(0)
    grid = rng.uniform(-1, 2, (2, H ** 3)).astype(np.float32)
    grid[0, :6] = [0.5, np.nextafter(np.float32(0.5), np.float32(1)), np.nan, np.inf, -np.inf, 0.4999999]
    bits = torch.empty(2 * H ** 3 // 8, dtype=torch.uint8, device="cuda")
    be.raymarching_backend.packbits(dev(grid), 2 * H ** 3 // 8, 0.5, bits)
    assert np.array_equal(host(bits), orc.packbits(grid, 0.5))
    assert np.array_equal(host(bits), np.packbits(grid.reshape(-1) > np.float32(0.5), bitorder="little"))

    cnt = rng.integers(0, 200, 500).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(cnt[:-1])]).astype(np.int32)
    rays = np.stack([off, cnt], 1).astype(np.int32)
    M = int(cnt.sum())
    res = torch.zeros(M, dtype=torch.int32, device="cuda")
    be.raymarching_backend.flatten_rays(dev(rays), 500, M, res)
    assert np.array_equal(host(res), orc.flatten_rays(rays, M))


def test_near_far_and_sph(be, orc):
    rng = np.random.default_rng(1)
    N = 5000
    o, d = make_rays(rng, N)
    d[:500] = rng.normal(size=(500, 3)).astype(np.float32)
    aabb = np.array([-1, -1, -1, 1, 1, 1], dtype=np.float32)
    rn, rf = orc.near_far_from_aabb(o, d, aabb, N, 0.05)
    nears = torch.empty(N, device="cuda")
    fars = torch.empty(N, device="cuda")
    be.raymarching_backend.near_far_from_aabb(dev(o), dev(d), dev(aabb), N, 0.05, nears, fars)
    assert np.array_equal(host(nears), rn) and np.array_equal(host(fars), rf)   # same IEEE ops -> bit-exact
    o2 = rng.uniform(-0.5, 0.5, (N, 3)).astype(np.float32)
    ref = orc.sph_from_ray(o2, d, 3.0, N)
    c = torch.empty(N, 2, device="cuda")
    be.raymarching_backend.sph_from_ray(dev(o2), dev(d), 3.0, N, c)
    np.testing.assert_allclose(host(c), ref, atol=2e-6)    # atan2f implementations differ by ulps


# ----------------------------------------------------------------------------- training march

MARCH_CASES = [
    # N, H, max_steps, C, bound, contract, dt_gamma, ldir
    (4096, 128, 1024, 1, 1.0, False, 0.0, False),       # north-star
    (1000, 128, 1024, 2, 2.0, False, 0.0, True),        # reference default bound, light dirs
    (777, 64, 512, 3, 4.0, False, 1.0 / 128, False),    # cone stepping, 3 cascades
    (500, 64, 256, 2, 8.0, True, 0.0, False),           # contraction
]


def march_inputs(orc, case, seed=0):
    N, H, max_steps, C, bound, contract, dt_gamma, ldir = case
    rng = np.random.default_rng(seed)
    bits, _ = brick_bitfield(orc, H, cascades=C, seed=seed, fill=0.08)
    o, d = make_rays(rng, N, radius=2.5 * bound if not contract else 3.0, jitter=0.6 * bound if not contract else 0.5)
    aabb = np.array([-bound] * 3 + [bound] * 3, dtype=np.float32)
    nears, fars = orc.near_far_from_aabb(o, d, aabb, N, 0.05)
    noises = rng.uniform(0, 1, N).astype(np.float32)
    ld = rng.normal(size=(N, 3)).astype(np.float32) if ldir else None
    return bits, o, d, ld, nears, fars, noises


@pytest.mark.parametrize("case", MARCH_CASES, ids=lambda c: f"N{c[0]}H{c[1]}C{c[3]}b{c[4]}c{int(c[5])}g{c[6] > 0}")
def test_march_rays_train_two_pass(be, orc, case):
    N, H, max_steps, C, bound, contract, dt_gamma, ldir = case
    bits, o, d, ld, nears, fars, noises = march_inputs(orc, case)
    rx, rd, rt, rrays, rl, M = orc.march_rays_train(o, d, ld, bits, bound, contract, dt_gamma, max_steps, C, H, nears,
                                                    fars, noises)
    assert M > 0
    rays = torch.empty(N, 2, dtype=torch.int32, device="cuda")
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    targs = (dev(o), dev(d), dev(ld) if ldir else None, dev(bits), bound, contract, dt_gamma, max_steps, N, C, H,
             dev(nears), dev(fars))
    be.raymarching_backend.march_rays_train(*targs, None, None, None, None, rays, counter, dev(noises))
    assert int(counter.item()) == M
    assert np.array_equal(host(rays), rrays)                       # counts AND ray-ordered offsets, bit-exact
    xyzs = torch.zeros(M, 3, device="cuda")
    dirs = torch.zeros(M, 3, device="cuda")
    ts = torch.zeros(M, 2, device="cuda")
    ldirs = torch.zeros(M, 3, device="cuda") if ldir else None
    be.raymarching_backend.march_rays_train(*targs, xyzs, dirs, ts, ldirs, rays, counter, dev(noises))
    assert np.array_equal(host(xyzs), rx) and np.array_equal(host(dirs), rd) and np.array_equal(host(ts), rt)
    if ldir:
        assert np.array_equal(host(ldirs), rl)


def occupancy_index(be, bits, C, H):
    rb = be.raymarching_backend
    index = torch.zeros(rb.occupancy_index_bytes(C, H) // 4, dtype=torch.int32, device="cuda")
    rb.build_occupancy_index(dev(bits), C, H, index)
    return index


@pytest.mark.parametrize("fill", [0.0, 0.05, 0.6])
def test_occupancy_index_is_the_bitfield(be, fill):
    """Header, block masks, ranks and non-zero 4x4x4 blocks of the LDS index against numpy."""
    rng = np.random.default_rng(5)
    C, H = 2, 64
    words = np.zeros(C * H ** 3 // 64, dtype=np.uint64)
    hot = rng.random(words.size) < fill
    words[hot] = rng.integers(1, 2 ** 63, hot.sum(), dtype=np.uint64)
    bits = words.view(np.uint8)
    idx = host(occupancy_index(be, bits, C, H)).view(np.uint32)
    n_groups = words.size // 32
    assert idx[0] == hot.sum() and idx[1] == words.size
    pairs = idx[4:4 + 2 * n_groups].reshape(n_groups, 2)
    nz = (words != 0).reshape(n_groups, 32)
    np.testing.assert_array_equal(pairs[:, 0], (nz.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(1).astype(np.uint32))
    np.testing.assert_array_equal(pairs[:, 1], np.concatenate([[0], np.cumsum(nz.sum(1))[:-1]]).astype(np.uint32))
    blocks = idx[4 + 2 * n_groups:4 + 2 * n_groups + 2 * int(idx[0])].view(np.uint64)
    np.testing.assert_array_equal(blocks, words[hot])


@pytest.mark.parametrize("mode", ["bitfield", "lds-index", "chain", "chain-index"])
@pytest.mark.parametrize("case", MARCH_CASES, ids=lambda c: f"N{c[0]}H{c[1]}C{c[3]}b{c[4]}c{int(c[5])}g{c[6] > 0}")
def test_march_rays_train_arena(be, orc, case, mode):
    """All first passes (serial on the bitfield, serial on the LDS index, chain-parallel -- which is the one constant-step
    kernel when dt_gamma == 0 -- without and with the index staged in LDS) against the oracle: counts, ray-ordered offsets,
    positions and ts bit for bit."""
    N, H, max_steps, C, bound, contract, dt_gamma, ldir = case
    bits, o, d, ld, nears, fars, noises = march_inputs(orc, case, seed=1)
    index = occupancy_index(be, bits, C, H) if mode in ("lds-index", "chain-index") else None
    chain_cap = max_steps * int(np.ceil(bound)) + 2 if mode.startswith("chain") else 0
    rx, rd, rt, rrays, rl, M = orc.march_rays_train(o, d, ld, bits, bound, contract, dt_gamma, max_steps, C, H, nears,
                                                    fars, noises)
    from raw_ngp_amd.raymarching import MarchArena
    for cap in (M + 1000, max(M // 2, 1)):                          # roomy arena, then one that overflows
        ar = MarchArena(N, max_steps, cap, "cuda", with_ldirs=ldir, chain_cap=chain_cap)
        be.raymarching_backend.march_rays_train_arena(dev(o), dev(d), dev(ld) if ldir else None, dev(bits), bound,
                                                      contract, dt_gamma, max_steps, N, C, H, dev(nears), dev(fars),
                                                      dev(noises), ar.t_scratch, cap, ar.xyzs, ar.dirs, ar.ts,
                                                      ar.ldirs, ar.rays, ar.counter, ar.ray_idx, index, ar.chain)
        written, needed, chain_overflow = host(ar.counter)[:3]
        assert chain_overflow == 0
        assert needed == M
        got = host(ar.rays)
        if cap >= M:
            assert written == M and np.array_equal(got, rrays)
        else:
            keep = (rrays[:, 0] + rrays[:, 1]) <= cap
            assert written == rrays[keep, 1].sum() <= cap
            assert np.array_equal(got[keep], rrays[keep]) and np.all(got[~keep, 1] == 0)
        w = int(written)
        assert np.array_equal(host(ar.xyzs)[:w], rx[:w]) and np.array_equal(host(ar.ts)[:w], rt[:w])
        assert np.array_equal(host(ar.dirs)[:w], rd[:w])
        assert np.array_equal(host(ar.ray_idx)[:w], orc.flatten_rays(rrays, M)[:w])
        if ldir:
            assert np.array_equal(host(ar.ldirs)[:w], rl[:w])


def test_march_arena_in_two_stages(be, orc):
    """The chain-parallel march split where the occupancy grid first matters: stage 1 (candidate parameters: rays, near/far,
    noise) with a bitfield that is still garbage, stage 2 with the real one == the one-call march, bit for bit; and the
    stages refuse to run without chain buffers."""
    case = MARCH_CASES[0]
    N, H, max_steps, C, bound, contract, dt_gamma, ldir = case
    bits, o, d, ld, nears, fars, noises = march_inputs(orc, case, seed=3)
    chain_cap = max_steps * int(np.ceil(bound)) + 2
    from raw_ngp_amd.raymarching import MarchArena
    rb = be.raymarching_backend
    res = []
    for staged in (False, True):
        ar = MarchArena(N, max_steps, 1 << 18, "cuda", with_ldirs=ldir, chain_cap=chain_cap)

        def run(grid, stage):
            rb.march_rays_train_arena(dev(o), dev(d), dev(ld) if ldir else None, grid, bound, contract, dt_gamma, max_steps, N,
                                      C, H, dev(nears), dev(fars), dev(noises), ar.t_scratch, 1 << 18, ar.xyzs, ar.dirs,
                                      ar.ts, ar.ldirs, ar.rays, ar.counter, ar.ray_idx, None, ar.chain, stage=stage)
        if staged:
            run(torch.full_like(dev(bits), 0xA5), 1)
            run(dev(bits), 2)
        else:
            run(dev(bits), 0)
        w = int(host(ar.counter)[0])
        res.append((host(ar.counter)[:3].copy(), host(ar.rays).copy(), host(ar.xyzs)[:w].copy(), host(ar.ts)[:w].copy(),
                    host(ar.dirs)[:w].copy()))
    assert res[0][0][0] > 0
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    ar = MarchArena(N, max_steps, 1 << 18, "cuda", with_ldirs=ldir, chain_cap=0)
    with pytest.raises(RuntimeError):
        rb.march_rays_train_arena(dev(o), dev(d), None, dev(bits), bound, contract, dt_gamma, max_steps, N, C, H, dev(nears),
                                  dev(fars), dev(noises), ar.t_scratch, 1 << 18, ar.xyzs, ar.dirs, ar.ts, ar.ldirs, ar.rays,
                                  ar.counter, ar.ray_idx, None, None, stage=1)


def test_march_arena_index_too_big_for_lds_falls_back_to_bitfield(be, orc):
    """Dense random bitfield: every 4x4x4 block is non-zero (32768 blocks > LDS budget) -> global probes, same result."""
    case = (2000, 128, 1024, 1, 1.0, False, 0.0, False)
    N, H, max_steps, C, bound, contract, dt_gamma, ldir = case
    _, o, d, ld, nears, fars, noises = march_inputs(orc, case, seed=2)
    bits = np.random.default_rng(9).integers(0, 256, H ** 3 // 8, dtype=np.uint8) & np.uint8(0x11)
    rx, rd, rt, rrays, rl, M = orc.march_rays_train(o, d, ld, bits, bound, contract, dt_gamma, max_steps, C, H, nears,
                                                    fars, noises)
    index = occupancy_index(be, bits, C, H)
    assert int(index[0]) > 20000
    from raw_ngp_amd.raymarching import MarchArena
    ar = MarchArena(N, max_steps, M + 10, "cuda")
    be.raymarching_backend.march_rays_train_arena(dev(o), dev(d), None, dev(bits), bound, contract, dt_gamma, max_steps,
                                                  N, C, H, dev(nears), dev(fars), dev(noises), ar.t_scratch, M + 10,
                                                  ar.xyzs, ar.dirs, ar.ts, None, ar.rays, ar.counter, None, index)
    assert int(ar.counter[0]) == M and np.array_equal(host(ar.rays), rrays)
    assert np.array_equal(host(ar.xyzs)[:M], rx) and np.array_equal(host(ar.ts)[:M], rt)


def test_march_chain_reports_a_short_chain_buffer(be, orc):
    case = (512, 64, 256, 1, 1.0, False, 0.0, False)
    N, H, max_steps, C, bound, contract, dt_gamma, ldir = case
    bits, o, d, ld, nears, fars, noises = march_inputs(orc, case, seed=3)
    from raw_ngp_amd.raymarching import MarchArena
    ar = MarchArena(N, max_steps, 1 << 20, "cuda", chain_cap=40)          # rays need up to ~256 candidates
    be.raymarching_backend.march_rays_train_arena(dev(o), dev(d), None, dev(bits), bound, contract, dt_gamma, max_steps,
                                                  N, C, H, dev(nears), dev(fars), dev(noises), ar.t_scratch, 1 << 20,
                                                  ar.xyzs, ar.dirs, ar.ts, None, ar.rays, ar.counter, None, None, ar.chain)
    assert int(ar.counter[2]) == 1
    assert int(host(ar.rays)[:, 1].max()) <= 40


def _step_lengths_with_ties():
    """max_steps values whose step d = 2 sqrt(3) / max_steps (float32, as the kernels compute it) ends in a 1 followed by p
    zero mantissa bits, p = 3..10: then fl(t + d) is a TIE for every t of the binade whose ulp is 2^(p+1) ulp(d) -- binades
    inside the range a ray's parameter really takes (0.05 .. 8) -- which is the one case where the closed form of
    march_const_step_kernel has an irregular first step."""
    picks = {}
    for ms in range(200, 5000):
        dt = np.float32(2.0) * np.float32(1.7320508075688772) / np.float32(ms)
        mant = (int(dt.view(np.uint32)) & 0x7FFFFF) | 0x800000
        p = (mant & -mant).bit_length() - 1
        if 3 <= p <= 10 and p not in picks:
            picks[p] = ms
    return sorted(picks.values())


@pytest.mark.parametrize("staged", [False, True], ids=["bitfield", "lds-index"])
@pytest.mark.parametrize("grid", ["full", "bricks"])
def test_march_constant_step_closed_form_is_the_serial_recurrence(be, orc, grid, staged):
    """dt_gamma == 0: the candidate parameters come from a per-binade closed form instead of t += dt (raymarching.hip:
    march_const_step_kernel).  Against the oracle's serial loop, bit for bit, with ray parameters that start anywhere from
    1e-7 to 7 (up to twenty binades per ray), step lengths that produce round-to-even ties in those binades, a fully
    occupied grid (every candidate is a sample: ts IS the chain) and a sparse one (every jump target comes from the table)."""
    from raw_ngp_amd.raymarching import MarchArena
    H, C, bound, N = 64, 1, 1.0, 768
    rng = np.random.default_rng(11)
    bits = np.full(H ** 3 // 8, 0xFF, dtype=np.uint8) if grid == "full" else brick_bitfield(orc, H, cascades=1, seed=4, fill=0.1)[0]
    steps = _step_lengths_with_ties()
    assert len(steps) >= 6
    index = occupancy_index(be, bits, C, H) if staged else None
    for max_steps in steps + [1024, 333]:
        o, d = make_rays(rng, N, radius=2.5, jitter=0.6)
        nears = np.exp(rng.uniform(np.log(1e-7), np.log(7.0), N)).astype(np.float32)
        nears[:8] = np.float32([0.0, 2.0, 4.0, 0.5, 1.0, 3.9999998, 1.9999999, 0.99999994])
        fars = (nears + rng.uniform(0.05, 3.5, N)).astype(np.float32)
        noises = rng.uniform(0, 1, N).astype(np.float32)
        noises[:8] = 0.0
        rx, rd, rt, rrays, rl, M = orc.march_rays_train(o, d, None, bits, bound, False, 0.0, max_steps, C, H, nears, fars, noises)
        assert M > 0
        ar = MarchArena(N, max_steps, M + 8, "cuda", chain_cap=4 * max_steps)
        be.raymarching_backend.march_rays_train_arena(dev(o), dev(d), None, dev(bits), bound, False, 0.0, max_steps, N, C, H,
                                                      dev(nears), dev(fars), dev(noises), ar.t_scratch, M + 8, ar.xyzs,
                                                      ar.dirs, ar.ts, None, ar.rays, ar.counter, ar.ray_idx, index, ar.chain)
        written, needed, cut = host(ar.counter)[:3]
        assert cut == 0 and needed == M and written == M, (max_steps, written, needed, M, cut)
        assert np.array_equal(host(ar.rays), rrays), max_steps
        assert np.array_equal(host(ar.ts)[:M], rt) and np.array_equal(host(ar.xyzs)[:M], rx), max_steps


def test_march_empty_inputs(be, orc):
    H = 32
    bits = np.zeros(H ** 3 // 8, dtype=np.uint8)
    o = np.array([[0, 0, 3.0]], dtype=np.float32)
    d = np.array([[0, 0.01, -1.0]], dtype=np.float32)
    nears = np.array([2.0], dtype=np.float32)
    fars = np.array([4.0], dtype=np.float32)
    rays = torch.full((1, 2), -7, dtype=torch.int32, device="cuda")
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    be.raymarching_backend.march_rays_train(dev(o), dev(d), None, dev(bits), 1.0, False, 0.0, 64, 1, 1, H, dev(nears),
                                            dev(fars), None, None, None, None, rays, counter,
                                            torch.zeros(1, device="cuda"))
    assert host(rays).tolist() == [[0, 0]] and int(counter.item()) == 0
    # N = 0 is a no-op
    be.raymarching_backend.march_rays_train(dev(o)[:0], dev(d)[:0], None, dev(bits), 1.0, False, 0.0, 64, 0, 1, H,
                                            dev(nears)[:0], dev(fars)[:0], None, None, None, None, rays[:0], counter,
                                            torch.zeros(0, device="cuda"))


# ----------------------------------------------------------------------------- compositing

@pytest.mark.parametrize("T_thresh", [0.0, 1e-8, 1e-4])
def test_composite_train_forward_backward(be, orc, T_thresh):
    rng = np.random.default_rng(5)
    N = 4096
    sig, rgb, ts, rays, M = synth_samples(rng, N)
    rw, rws, rdep, rimg = orc.composite_rays_train_forward(sig, rgb, ts, rays, M, N, T_thresh)
    w = torch.zeros(M, device="cuda")
    ws = torch.empty(N, device="cuda")
    dep = torch.empty(N, device="cuda")
    img = torch.empty(N, 3, device="cuda")
    be.raymarching_backend.composite_rays_train_forward(dev(sig), dev(rgb), dev(ts), dev(rays), M, N, T_thresh, w, ws,
                                                        dep, img)
    # __expf vs libm expf (few ulp on alpha) accumulated over <= 80 samples; where T crosses
    # T_thresh within rounding the stop index may differ by one sample of weight <= T_thresh
    tol = dict(rtol=2e-4, atol=2e-6 + 2 * T_thresh)
    np.testing.assert_allclose(host(w), rw, **tol)
    np.testing.assert_allclose(host(ws), rws, **tol)
    np.testing.assert_allclose(host(dep), rdep, rtol=2e-4, atol=1e-5 + 8 * T_thresh)
    np.testing.assert_allclose(host(img), rimg, **tol)

    gw = rng.normal(size=M).astype(np.float32)
    gws = rng.normal(size=N).astype(np.float32)
    gdep = rng.normal(size=N).astype(np.float32)
    gimg = rng.normal(size=(N, 3)).astype(np.float32)
    rgs, rgc = orc.composite_rays_train_backward(gw, gws, gdep, gimg, sig, rgb, ts, rays, rws, rdep, rimg, M, N, T_thresh)
    gs = torch.zeros(M, device="cuda")
    gc = torch.zeros(M, 3, device="cuda")
    be.raymarching_backend.composite_rays_train_backward(dev(gw), dev(gws), dev(gdep), dev(gimg), dev(sig), dev(rgb),
                                                         dev(ts), dev(rays), dev(rws), dev(rdep), dev(rimg), M, N,
                                                         T_thresh, gs, gc)
    np.testing.assert_allclose(host(gc), rgc, rtol=2e-4, atol=1e-5 + 8 * T_thresh)
    # grad_sigma subtracts nearly equal running sums: bound the error relative to the ray scale
    err = np.abs(host(gs) - rgs)
    assert np.all(err <= 2e-3 * np.abs(rgs) + 1e-3 * np.abs(ts[:, 1]) * 50 + 1e-6)


def test_composite_train_empty_and_overflow(be):
    rays = torch.tensor([[0, 0], [0, 3], [2, 5]], dtype=torch.int32, device="cuda")
    sig = torch.ones(4, device="cuda")
    rgb = torch.ones(4, 3, device="cuda")
    ts = torch.tensor([[0.1, 0.1], [0.2, 0.1], [0.3, 0.1], [0.4, 0.1]], device="cuda")
    w = torch.zeros(4, device="cuda")
    ws = torch.full((3,), 9.0, device="cuda")
    dep = torch.full((3,), 9.0, device="cuda")
    img = torch.full((3, 3), 9.0, device="cuda")
    from raw_ngp_amd import _lib
    _lib.raymarching_backend.composite_rays_train_forward(sig, rgb, ts, rays, 4, 3, 1e-4, w, ws, dep, img)
    assert ws[0] == 0 and ws[2] == 0 and torch.all(img[0] == 0) and torch.all(img[2] == 0) and dep[2] == 0
    assert ws[1] > 0 and torch.all(w[:3] > 0) and w[3] == 0


# ----------------------------------------------------------------------------- inference pair

INFER_CASES = [
    # H, cascades, bound, contract, dt_gamma, n_step, T_thresh
    (128, 1, 1.0, False, 0.0, 4, 1e-2),          # the Function's defaults
    (128, 2, 2.0, False, 1.0 / 128, 8, 1e-8),    # reference default bound (2 cascades), cone stepping, the CLI's T_thresh
    (64, 2, 4.0, True, 0.0, 2, 1e-4),            # contraction (query bound 2 -> 2 cascades, real bound 4)
    (64, 3, 4.0, False, 1.0 / 256, 1, 1e-2),     # three cascades, one sample per round
]


@pytest.mark.parametrize("case", INFER_CASES, ids=lambda c: f"H{c[0]}C{c[1]}b{c[2]}c{int(c[3])}g{c[4] > 0}n{c[5]}")
def test_inference_pair(be, orc, case):
    H, C, bound, contract, dt_gamma, n_step, T_thresh = case
    rng = np.random.default_rng(9)
    N, max_steps = 2000, 1024
    bits, _ = brick_bitfield(orc, H, cascades=C, seed=3, fill=0.08)
    o, d = make_rays(rng, N, radius=2.5 * bound if not contract else 3.0, jitter=0.6 * bound if not contract else 0.5)
    aabb = np.array([-bound] * 3 + [bound] * 3, dtype=np.float32)
    nears, fars = orc.near_far_from_aabb(o, d, aabb, N, 0.05)
    alive = np.arange(N, dtype=np.int32)[::2].copy()
    n_alive = alive.size
    rays_t = nears.copy()
    rays_t[::3] += np.float32(0.37 * bound)          # rays in different states of progress, like later rounds of the loop
    noises = rng.uniform(0, 1, n_alive).astype(np.float32)
    rx, rd, rt = orc.march_rays(n_alive, n_step, alive, rays_t, o, d, bound, contract, dt_gamma, max_steps, C, H, bits, nears,
                                fars, noises)
    assert (rt[:, 1] > 0).sum() > n_alive // 4                                     # live samples ...
    assert contract or (rt[:, 1] == 0).sum() > 0          # ... and zero tails (a contracted ray never leaves the volume)
    M = n_alive * n_step
    xyzs = torch.zeros(M, 3, device="cuda")
    dirs = torch.zeros(M, 3, device="cuda")
    ts = torch.zeros(M, 2, device="cuda")
    be.raymarching_backend.march_rays(n_alive, n_step, dev(alive), dev(rays_t), dev(o), dev(d), bound, contract, dt_gamma,
                                      max_steps, C, H, dev(bits), dev(nears), dev(fars), xyzs, dirs, ts, dev(noises))
    assert np.array_equal(host(xyzs), rx) and np.array_equal(host(dirs), rd) and np.array_equal(host(ts), rt)

    sig = (40.0 * np.exp(-4 * (rx ** 2).sum(1))).astype(np.float32)
    col = (0.5 + 0.5 * np.sin(3 * rx)).astype(np.float32)
    ws = rng.uniform(0, 0.3, N).astype(np.float32)
    dep = rng.uniform(0, 1, N).astype(np.float32)
    img = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    r_alive, r_t, r_ws, r_dep, r_img = alive.copy(), rays_t.copy(), ws.copy(), dep.copy(), img.copy()
    orc.composite_rays(n_alive, n_step, T_thresh, r_alive, r_t, sig, col, rt, r_ws, r_dep, r_img)
    assert (r_alive >= 0).sum() > 0 and (contract or (r_alive < 0).sum() > 0)   # rays go on; some die in this round
    g_alive, g_t, g_ws, g_dep, g_img = dev(alive), dev(rays_t), dev(ws), dev(dep), dev(img)
    be.raymarching_backend.composite_rays(n_alive, n_step, T_thresh, g_alive, g_t, dev(sig), dev(col), ts, g_ws, g_dep, g_img)
    assert np.array_equal(host(g_alive), r_alive)
    np.testing.assert_allclose(host(g_t), r_t, rtol=0, atol=0)
    np.testing.assert_allclose(host(g_ws), r_ws, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(host(g_dep), r_dep, rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(host(g_img), r_img, rtol=2e-4, atol=2e-6)


def test_ray_gradient_segment_sum(be, orc):
    rng = np.random.default_rng(10)
    N = 3000
    cnt = rng.integers(0, 150, N).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(cnt[:-1])]).astype(np.int32)
    M = int(cnt.sum())
    rays = np.stack([off, cnt], 1).astype(np.int32)
    gx = rng.normal(size=(M, 3)).astype(np.float32)
    gd = rng.normal(size=(M, 3)).astype(np.float32)
    ts = rng.uniform(0, 3, (M, 2)).astype(np.float32)
    ro, rd = orc.march_rays_train_backward(gx, gd, ts, rays, N, M)
    go = torch.empty(N, 3, device="cuda")
    gdd = torch.empty(N, 3, device="cuda")
    be.raymarching_backend.march_rays_train_backward(dev(gx), dev(gd), dev(ts), dev(rays), N, M, go, gdd)
    np.testing.assert_allclose(host(go), ro, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(gdd), rd, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("case", MARCH_CASES, ids=lambda c: f"N{c[0]}H{c[1]}C{c[3]}b{c[4]}c{int(c[5])}g{c[6] > 0}")
def test_march_function_single_pass_equals_two_calls(be, orc, case):
    """The autograd op `march_rays_train` marches once (chain-parallel, scratch arena) instead of calling the C ABI twice:
    sizes, contents and the ray gradients of its backward must be identical to the literal two-call protocol."""
    from raw_ngp_amd import raymarching as rm
    N, H, max_steps, C, bound, contract, dt_gamma, ldir = case
    bits, o, d, ld, nears, fars, _ = march_inputs(orc, case, seed=4)
    outs = []
    for single in (True, False):
        rm.raymarching.single_pass = single
        try:
            torch.manual_seed(7)                                  # same perturbation noise in both runs
            ro, rd = dev(o).requires_grad_(True), dev(d).requires_grad_(True)
            xyzs, dirs, ts, rays, ldirs = rm.march_rays_train(ro, rd, dev(ld) if ldir else None, bound, contract, dev(bits),
                                                              C, H, dev(nears), dev(fars), True, dt_gamma, max_steps)
            (xyzs.sum() * 2.0 + (dirs * dirs).sum()).backward()
            outs.append((xyzs, dirs, ts, rays, ldirs, ro.grad, rd.grad))
        finally:
            rm.raymarching.single_pass = True
    a, b = outs
    assert a[0].shape[0] == b[0].shape[0] == int(b[3][:, 1].sum()) > 0
    for x, y in zip(a, b):
        assert (x is None and y is None) or torch.equal(x, y)


def test_grid_op_input_gradient_level_major_jacobian(be, orc):
    """The op keeps its Jacobian level-major on the binned route (coalesced writes) and in the reference layout on the
    atomic route: d loss / d inputs must come out the same bits, and match the oracle's chain rule."""
    from raw_ngp_amd.gridencoder.grid import grid_encode
    D, C, L, H, log2T, desired = 3, 2, 16, 16, 19, 2048
    B = 5000
    offsets, S, table, x = grid_setup(orc, D, C, L, H, log2T, desired, B, seed=5)
    g = np.random.default_rng(6).normal(size=(B, L * C)).astype(np.float32)
    grads = []
    for binned in (True, False):
        type(be.gridencoder_backend).use_binned_backward = binned
        try:
            xi = dev(x).requires_grad_(True)
            emb = dev(table).requires_grad_(True)
            out = grid_encode(xi, emb, dev(offsets), float(2.0 ** S), H, True, 0, False, 0, None if binned else L - 2)
            (out * dev(g)).sum().backward()
            grads.append((xi.grad.clone(), emb.grad.clone()))
        finally:
            type(be.gridencoder_backend).use_binned_backward = True
    # same max_level for the comparison run
    type(be.gridencoder_backend).use_binned_backward = True
    xi = dev(x).requires_grad_(True)
    out = grid_encode(xi, dev(table).requires_grad_(True), dev(offsets), float(2.0 ** S), H, True, 0, False, 0, L - 2)
    (out * dev(g)).sum().backward()
    assert torch.equal(xi.grad, grads[1][0])                       # level-major vs reference layout, max_level = L - 2
    _, jac = orc.grid_encode_forward(x, table, offsets, B, D, C, L, L, S, H, True, 0, False, 0)
    gl = np.ascontiguousarray(g.reshape(B, L, C).transpose(1, 0, 2))
    _, ref_gi = orc.grid_encode_backward(gl, x, table, offsets, B, D, C, L, L, S, H, jac, 0, False, 0)
    np.testing.assert_allclose(host(grads[0][0]), ref_gi, rtol=1e-5, atol=1e-4 * np.abs(ref_gi).max())
