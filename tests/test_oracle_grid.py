"""Known-answer tests that pin the CPU oracle's grid encoder (the reference has no tests:
SURVEY.md section 4, so these are authored from first principles, section 8c)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

PRIMES = [1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737]


def py_index(gridtype, D, T, res, c):
    """Scalar transcription of the index rule (gridencoder.cu:61-79) with uint32 wrap."""
    M = 0xFFFFFFFF
    stride, idx, d = 1, 0, 0
    while d < D and stride <= T:
        idx = (idx + c[d] * stride) & M
        stride = (stride * res) & M
        d += 1
    if gridtype == 0 and stride > T:
        idx = 0
        for k in range(D):
            idx ^= (c[k] * PRIMES[k]) & M
    return idx % T


def py_grid_forward(x, table, offsets, L, C, S, H, gridtype=0):
    """Slow float64 reference for tiny cases (align_corners=False, linear)."""
    B, D = x.shape
    out = np.zeros((L, B, C))
    for l in range(L):
        res = int(np.ceil(np.float32(np.exp2(np.float32(l) * np.float32(S))) * np.float32(H)))
        T = int(offsets[l + 1] - offsets[l])
        for b in range(B):
            if np.any(x[b] < 0) or np.any(x[b] > 1):
                continue
            pos = np.clip(x[b].astype(np.float64) * res - 0.5, 0, res - 1)
            cell = np.floor(pos).astype(np.int64)
            f = pos - cell
            for corner in range(1 << D):
                w, c = 1.0, []
                for d in range(D):
                    if corner >> d & 1:
                        w *= f[d]
                        c.append(min(cell[d] + 1, res - 1))
                    else:
                        w *= 1 - f[d]
                        c.append(cell[d])
                row = py_index(gridtype, D, T, res, [int(v) for v in c])
                out[l, b] += w * table[offsets[l] + row]
    return out


def test_offsets_match_survey_table(orc):
    # total rows for bound 1 / bound 2 / plumbing config
    off1, s1 = orc.grid_offsets(desired_resolution=2048)
    off2, s2 = orc.grid_offsets(desired_resolution=4096)
    off3, s3 = orc.grid_offsets(num_levels=8, desired_resolution=2048)
    assert off1[-1] == 6098120 and abs(s1 - 1.381913) < 1e-6
    assert off2[-1] == 6299960 and abs(s2 - 1.447269) < 1e-6
    assert off3[-1] == 2920448 and abs(s3 - 2.0) < 1e-12
    res = orc.grid_resolutions(np.float32(np.log2(s1)), 16, 16)
    assert list(res) == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    res2 = orc.grid_resolutions(np.float32(np.log2(s2)), 16, 16)
    assert list(res2) == [16, 24, 34, 49, 71, 102, 148, 213, 308, 446, 646, 934, 1352, 1956, 2831, 4096]


@pytest.mark.parametrize("D,C", [(3, 2), (2, 4), (3, 1)])
def test_forward_matches_scalar_python(orc, D, C):
    rng = np.random.default_rng(0)
    L, H = 6, 4
    offsets, scale = orc.grid_offsets(input_dim=D, num_levels=L, level_dim=C, base_resolution=H,
                                      log2_hashmap_size=9, desired_resolution=64)
    S = float(np.log2(scale))
    table = rng.uniform(-1, 1, (offsets[-1], C)).astype(np.float32)
    B = 64
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2, 0] = -0.1   # out of range -> zeros
    x[3, 1] = 1.5
    out, _ = orc.grid_encode_forward(x, table, offsets, B, D, C, L, L, S, H)
    ref = py_grid_forward(x, table.astype(np.float64), offsets, L, C, S, H)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=5e-6)
    assert np.all(out[:, 2] == 0) and np.all(out[:, 3] == 0)


def test_dense_levels_match_grid_sample(orc):
    """A dense level is trilinear interpolation of an x-fastest volume with the
    `x*res - 0.5` / border convention = F.grid_sample(align_corners=False, border)."""
    rng = np.random.default_rng(1)
    D, C, L, H = 3, 2, 3, 8
    offsets, scale = orc.grid_offsets(input_dim=D, num_levels=L, level_dim=C, base_resolution=H,
                                      log2_hashmap_size=19, desired_resolution=32)
    S = float(np.log2(scale))
    res = orc.grid_resolutions(np.float32(S), H, L)
    table = rng.uniform(-1, 1, (offsets[-1], C)).astype(np.float32)
    B = 500
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    out, _ = orc.grid_encode_forward(x, table, offsets, B, D, C, L, L, S, H)
    for l in range(L):
        r = int(res[l])
        assert r ** 3 <= offsets[l + 1] - offsets[l]
        vol = torch.from_numpy(table[offsets[l]:offsets[l] + r ** 3]).double().view(r, r, r, C)  # z,y,x,C
        vol = vol.permute(3, 0, 1, 2)[None]
        grid = torch.from_numpy(x).double() * 2 - 1
        samp = F.grid_sample(vol, grid.view(1, B, 1, 1, 3), mode="bilinear", padding_mode="border",
                             align_corners=False)
        samp = samp.view(C, B).t().numpy()
        np.testing.assert_allclose(out[l], samp, rtol=2e-5, atol=2e-6)


def test_max_level_leaves_tail_untouched(orc):
    rng = np.random.default_rng(2)
    D, C, L, H = 3, 2, 4, 4
    offsets, scale = orc.grid_offsets(num_levels=L, base_resolution=H, log2_hashmap_size=8, desired_resolution=32)
    S = float(np.log2(scale))
    table = rng.uniform(-1, 1, (offsets[-1], C)).astype(np.float32)
    x = rng.uniform(0, 1, (16, D)).astype(np.float32)
    full, _ = orc.grid_encode_forward(x, table, offsets, 16, D, C, L, L, S, H)
    part, _ = orc.grid_encode_forward(x, table, offsets, 16, D, C, L, 2, S, H)
    np.testing.assert_array_equal(part[:2], full[:2])
    assert np.all(part[2:] == 0)


@pytest.mark.parametrize("interp", [0, 1])
@pytest.mark.parametrize("align", [False, True])
def test_dy_dx_matches_finite_differences(orc, interp, align):
    rng = np.random.default_rng(3)
    D, C, L, H = 3, 2, 4, 4
    offsets, scale = orc.grid_offsets(num_levels=L, base_resolution=H, log2_hashmap_size=8, desired_resolution=24)
    S = float(np.log2(scale))
    res = orc.grid_resolutions(np.float32(S), H, L)
    table = rng.uniform(-1, 1, (offsets[-1], C)).astype(np.float32)
    B = 40
    # keep points away from cell boundaries of every level so central differences are valid
    # ... and inside [0.5/res, 1 - 0.5/res] of the coarsest level: where the position is clamped
    # the reference still reports the unclamped slope (gridencoder.cu:205-247), finite differences 0
    x = rng.uniform(0.15, 0.85, (B, D)).astype(np.float32)
    eps = 1e-3
    ok = np.ones(B, bool)
    for r in res:
        rr = (r - 1) if align else r
        p = x * rr - (0.0 if align else 0.5)
        ok &= np.all(np.abs(p - np.round(p)) > 4 * eps * rr, axis=1)
    x = x[ok]
    B = x.shape[0]
    assert B > 5
    _, jac = orc.grid_encode_forward(x, table, offsets, B, D, C, L, L, S, H, True, 0, align, interp)
    jac = jac.reshape(B, L, D, C)
    for d in range(D):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += eps
        xm[:, d] -= eps
        fp, _ = orc.grid_encode_forward(xp, table, offsets, B, D, C, L, L, S, H, False, 0, align, interp)
        fm, _ = orc.grid_encode_forward(xm, table, offsets, B, D, C, L, L, S, H, False, 0, align, interp)
        fd = (fp.astype(np.float64) - fm) / (xp[:, d] - xm[:, d]).astype(np.float64)[None, :, None]
        np.testing.assert_allclose(jac[:, :, d, :].transpose(1, 0, 2), fd, rtol=2e-2, atol=2e-2)


def test_backward_is_transpose_of_forward(orc):
    """<grad, forward(table)> == <backward(grad), table>  (forward is linear in the table)."""
    rng = np.random.default_rng(4)
    D, C, L, H = 3, 2, 5, 4
    offsets, scale = orc.grid_offsets(num_levels=L, base_resolution=H, log2_hashmap_size=9, desired_resolution=48)
    S = float(np.log2(scale))
    table = rng.uniform(-1, 1, (offsets[-1], C)).astype(np.float32)
    B = 300
    x = rng.uniform(-0.05, 1.05, (B, D)).astype(np.float32)
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    out, jac = orc.grid_encode_forward(x, table, offsets, B, D, C, L, L, S, H, True)
    gt, gi = orc.grid_encode_backward(g, x, table, offsets, B, D, C, L, L, S, H, jac)
    lhs = np.sum(g.astype(np.float64) * out)
    rhs = np.sum(gt.astype(np.float64) * table)
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
    # input gradient = sum_l,c g * dy_dx
    ref = np.einsum("lbc,bldc->bd", g.astype(np.float64), jac.reshape(B, L, D, C))
    np.testing.assert_allclose(gi, ref, rtol=1e-4, atol=1e-4)


def test_weight_decay_and_tv(orc):
    rng = np.random.default_rng(5)
    D, C, L, H = 3, 2, 3, 4
    offsets, scale = orc.grid_offsets(num_levels=L, base_resolution=H, log2_hashmap_size=7, desired_resolution=16)
    S = float(np.log2(scale))
    n = int(offsets[-1])
    table = rng.uniform(-1, 1, (n, C)).astype(np.float32)
    g0 = rng.normal(size=(n, C)).astype(np.float32)
    g = orc.grad_weight_decay(table, g0, offsets, 0.1, n, C, L)
    T = np.concatenate([np.full(offsets[l + 1] - offsets[l], offsets[l + 1] - offsets[l]) for l in range(L)])
    np.testing.assert_allclose(g, g0 + 2 * 0.1 * table / T[:, None], rtol=1e-6, atol=1e-7)

    # TV on a single interior point of a dense level: gradient lands on the centre row only
    res = orc.grid_resolutions(np.float32(S), H, L)
    x = np.array([[0.4, 0.6, 0.3]], dtype=np.float32)
    gtv = orc.grad_total_variation(x, table, np.zeros_like(table), offsets, 1.0, 1, D, C, L, S, H)
    for l in range(L):
        r, Tl = int(res[l]), int(offsets[l + 1] - offsets[l])
        cell = np.floor(np.clip(x[0] * np.float32(r) - 0.5, 0, r - 1)).astype(int)
        centre = py_index(0, D, Tl, r, [int(v) for v in cell])
        tab = table[offsets[l]:offsets[l + 1]].astype(np.float64)
        s, q = np.zeros(C), np.zeros(C)
        for d in range(D):
            for step in (1, -1):
                c = [int(v) for v in cell]
                c[d] += step
                if c[d] < 0:
                    continue
                dv = tab[centre] - tab[py_index(0, D, Tl, r, c)]
                s += dv
                q += dv * dv
        expect = (1.0 / (2 * D)) * s / np.sqrt(q + 1e-9)
        blk = gtv[offsets[l]:offsets[l + 1]]
        np.testing.assert_allclose(blk[centre], expect, rtol=1e-4, atol=1e-6)
        blk = blk.copy()
        blk[centre] = 0
        assert np.all(blk == 0)
