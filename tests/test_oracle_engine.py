"""Oracle pieces of the fused step that have published known answers (CPU)."""
import numpy as np


def test_philox_known_answers(orc):
    """Random123's kat_vectors for philox4x32-10 (the three published counter/key pairs)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = orc.philox4x32_10(np.array(ctr, np.uint32), key)
        assert tuple(int(v) for v in got) == want


def test_sample_rays_matches_reference_get_rays(orc, golden_dir):
    """The ray construction of the sampler against the reference-generated get_rays fixture: feed the fixture's
    pixel coordinates through the same formulas."""
    import os
    g = np.load(os.path.join(golden_dir, "get_rays.npz"))
    H, W = int(g["H"]), int(g["W"])
    rng = np.random.default_rng(0)
    V = g["poses"].shape[0]
    images = rng.integers(0, 256, (V, H, W, 4), dtype=np.uint8)
    N = 4096
    out = orc.sample_rays(images, g["poses"], g["intrinsics"], N, seed=5, draw=3)
    view, pix = out["index"][:, 0], out["index"][:, 1]
    j, i = pix // W, pix % W
    ref_d = np.stack([g[f"rays_d_{v}"] for v in range(V)])[view, pix]      # full-image rays are row-major
    ref_o = np.stack([g[f"rays_o_{v}"] for v in range(V)])[view, pix]
    np.testing.assert_allclose(out["rays_d"], ref_d, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(out["rays_o"], ref_o)
    np.testing.assert_array_equal(out["gt"], images[view, j, i].astype(np.float32) / np.float32(255))
    assert out["noises"].min() >= 0 and out["noises"].max() < 1
    assert out["bg"].min() >= 0 and out["bg"].max() < 1
    # (view, pixel) is uniform: every cell is hit about N / cells times (6-sigma bound), draws differ per draw number
    hist = np.bincount(view * H * W + pix, minlength=V * H * W)
    mean = N / (V * H * W)
    assert abs(hist - mean).max() < 6 * np.sqrt(mean)
    other = orc.sample_rays(images, g["poses"], g["intrinsics"], N, seed=5, draw=4)
    assert (other["index"] != out["index"]).any(1).mean() > 0.9


def test_binned_cell_draws_have_the_law_of_independent_draws(orc):
    """oracle.density_grid_sample(binned=True) -- the draws the device generates in Morton order -- against binned=False, the
    independent draws of nerf/renderer.py:851-866 (Philox in place of torch's generator): the same number of draws per half,
    uniform half sorted by 512-cell block with Poisson(1/4) hits per cell like the independent sample, occupied half only on
    occupied cells, every one of them about equally often; reproducible; a different draw number gives different cells."""
    import numpy as np
    rng = np.random.default_rng(5)
    H = 32
    cells = H ** 3
    grid = rng.uniform(-0.5, 3.0, cells).astype(np.float32)
    grid[rng.random(cells) < 0.4] = -1.0
    n = cells // 4
    args = (grid, H, 1.9375, 0.0625, n, n, False, (7 << 32) | 99)
    bi, bx = orc.density_grid_sample(*args, 3, binned=True)
    ii, ix = orc.density_grid_sample(*args, 3)
    assert bi.shape == ii.shape == (2 * n,) and bx.shape == (2 * n, 3)
    assert np.all(np.diff(bi[:n] >> 3) >= 0)                         # 4096 bins of 8 cells at 32^3
    occ = np.flatnonzero(grid > 0)
    for v in (bi, ii):
        assert np.all(np.isin(v[n:], occ)) and 0 <= v[:n].min() and v[:n].max() < cells
        hits = np.bincount(np.bincount(v[:n], minlength=cells), minlength=6)[:6]
        expect = cells * np.exp(-0.25) * 0.25 ** np.arange(6) / np.array([1, 1, 2, 6, 24, 120])
        assert np.all(np.abs(hits - expect) <= 4 * np.sqrt(expect) + 2)
        per_cell = np.bincount(v[n:], minlength=cells)[occ]
        assert abs(per_cell.mean() - n / len(occ)) < 1e-9 and per_cell.max() <= 12
    rank = np.searchsorted(occ, bi[n:])
    assert np.all(np.diff(rank) >= -(len(occ) // 4096 + 2))          # Morton order up to a bin's width
    # positions stay inside their cells: |x - centre| <= half along every axis
    from oracle.oracle import _compact_bits
    c = np.stack([_compact_bits(bi.astype(np.uint32) >> np.uint32(k)) for k in range(3)], 1).astype(np.float32)
    centre = (2 * c / (H - 1) - 1) * np.float32(1.9375)
    assert np.all(np.abs(bx - centre) <= 0.0625 + 1e-6)
    again, _ = orc.density_grid_sample(*args, 3, binned=True)
    other, _ = orc.density_grid_sample(*args, 4, binned=True)
    assert np.array_equal(again, bi) and not np.array_equal(other, bi)
