"""Pins the oracle's SH (and frequency) encoder against scipy and finite differences."""
import os

import numpy as np
import pytest
from scipy import special


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def real_sh_scipy(l, m, v):
    """Real SH in the reference's convention (Condon-Shortley phase kept):
    m>0: sqrt2 Re Y_l^m ; m<0: sqrt2 Im Y_l^|m| ; m=0: Y_l^0."""
    x, y, z = v[:, 0].astype(np.float64), v[:, 1].astype(np.float64), v[:, 2].astype(np.float64)
    polar = np.arccos(np.clip(z, -1, 1))
    azim = np.arctan2(y, x)
    Y = special.sph_harm_y(l, abs(m), polar, azim)
    if m == 0:
        return Y.real
    return np.sqrt(2.0) * (Y.real if m > 0 else Y.imag)


@pytest.mark.parametrize("degree", [1, 2, 4, 6, 8])
def test_values_match_scipy(orc, degree):
    rng = np.random.default_rng(0)
    v = unit(rng, 200)
    out, _ = orc.sh_encode_forward(v, v.shape[0], 3, degree)
    assert out.shape == (200, degree * degree)
    for l in range(degree):
        for m in range(-l, l + 1):
            np.testing.assert_allclose(out[:, l * l + l + m], real_sh_scipy(l, m, v), rtol=2e-5, atol=2e-6,
                                       err_msg=f"l={l} m={m}")


def test_first_order_sign_convention(orc):
    # shencoder.cu:50-54: Y_0 = 0.28209..., Y_1 = 0.48860 * (-y, z, -x)
    v = np.array([[0.36, 0.48, 0.8]], dtype=np.float32)
    out, _ = orc.sh_encode_forward(v, 1, 3, 2)
    np.testing.assert_allclose(out[0], [0.28209479177387814, -0.48860251190291987 * 0.48,
                                        0.48860251190291987 * 0.8, -0.48860251190291987 * 0.36], rtol=1e-6)


def test_polynomial_representatives(orc):
    """Off the sphere the reference's polynomials are functions of z times Re/Im (x+iy)^m
    (e.g. index 6 = 0.9462 z^2 - 0.3154 has NO x,y dependence, shencoder.cu:58)."""
    p = np.array([[0.3, -0.7, 0.5], [1.3, 0.2, -0.4]], dtype=np.float32)
    out, jac = orc.sh_encode_forward(p, 2, 3, 4, True)
    jac = jac.reshape(2, 3, 16)
    z = p[:, 2].astype(np.float64)
    x, y = p[:, 0].astype(np.float64), p[:, 1].astype(np.float64)
    np.testing.assert_allclose(out[:, 6], 0.94617469575755997 * z * z - 0.31539156525251999, rtol=1e-6)
    np.testing.assert_allclose(jac[:, 0, 6], 0.0, atol=1e-7)
    np.testing.assert_allclose(jac[:, 1, 6], 0.0, atol=1e-7)
    np.testing.assert_allclose(jac[:, 2, 6], 2 * 0.94617469575755997 * z, rtol=1e-6)
    np.testing.assert_allclose(out[:, 8], 0.54627421529603959 * (x * x - y * y), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out[:, 9], 0.59004358992664352 * y * (-3 * x * x + y * y), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out[:, 12], 0.3731763325901154 * z * (5 * z * z - 3), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("degree", [4, 8])
def test_jacobian_matches_finite_differences(orc, degree):
    rng = np.random.default_rng(1)
    p = rng.uniform(-1, 1, (50, 3)).astype(np.float32)   # ambient derivative: any point
    _, jac = orc.sh_encode_forward(p, 50, 3, degree, True)
    jac = jac.reshape(50, 3, degree * degree)
    eps = 1e-3
    for d in range(3):
        pp, pm = p.copy(), p.copy()
        pp[:, d] += eps
        pm[:, d] -= eps
        fp, _ = orc.sh_encode_forward(pp, 50, 3, degree)
        fm, _ = orc.sh_encode_forward(pm, 50, 3, degree)
        fd = (fp.astype(np.float64) - fm) / (pp[:, d] - pm[:, d]).astype(np.float64)[:, None]
        np.testing.assert_allclose(jac[:, d], fd, rtol=5e-3, atol=5e-3)


def test_orthonormal_on_sphere(orc):
    # Gauss-Legendre in z times uniform in azimuth integrates degree<=14 polynomials exactly
    nz, na = 16, 32
    zs, wz = np.polynomial.legendre.leggauss(nz)
    az = (np.arange(na) + 0.5) * 2 * np.pi / na
    Z, A = np.meshgrid(zs, az, indexing="ij")
    R = np.sqrt(1 - Z * Z)
    v = np.stack([R * np.cos(A), R * np.sin(A), Z], -1).reshape(-1, 3).astype(np.float32)
    w = (wz[:, None] * np.full((1, na), 2 * np.pi / na)).reshape(-1)
    out, _ = orc.sh_encode_forward(v, v.shape[0], 3, 8)
    G = (out.astype(np.float64) * w[:, None]).T @ out.astype(np.float64)
    np.testing.assert_allclose(G, np.eye(64), atol=2e-5)


def test_backward_accumulates(orc):
    rng = np.random.default_rng(2)
    v = unit(rng, 30)
    out, jac = orc.sh_encode_forward(v, 30, 3, 4, True)
    g = rng.normal(size=(30, 16)).astype(np.float32)
    gi = orc.sh_encode_backward(g, v, 30, 3, 4, jac)
    ref = np.einsum("bc,bdc->bd", g.astype(np.float64), jac.reshape(30, 3, 16))
    np.testing.assert_allclose(gi, ref, rtol=1e-5, atol=1e-5)


def test_freq_encoder(orc):
    rng = np.random.default_rng(3)
    B, D, deg = 20, 3, 4
    C = D + D * 2 * deg
    x = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    out = orc.freq_encode_forward(x, B, D, deg, C)
    ref = [x]
    for f in range(deg):
        ref += [np.sin(x * 2.0 ** f), np.cos(x * 2.0 ** f)]
    np.testing.assert_allclose(out, np.concatenate(ref, 1), rtol=1e-5, atol=1e-6)
    g = rng.normal(size=(B, C)).astype(np.float32)
    gi = orc.freq_encode_backward(g, out, B, D, deg, C)
    expect = g[:, :D].astype(np.float64).copy()
    for f in range(deg):
        s = g[:, D + 2 * f * D: D + 2 * f * D + D]
        c = g[:, D + 2 * f * D + D: D + 2 * f * D + 2 * D]
        expect += 2.0 ** f * (s * np.cos(x * 2.0 ** f) - c * np.sin(x * 2.0 ** f))
    np.testing.assert_allclose(gi, expect, rtol=1e-4, atol=1e-5)


def test_freq_encoder_is_the_references_torch_encoder(orc, golden_dir):
    """The reference holds ONE encoder in Python as well: FreqEncoder_torch (encoding.py:6-50).  Its outputs and autograd
    gradients on seeded inputs (tests/golden/freq_torch.npz, written by oracle/gen_golden.py from the imported reference)
    pin the oracle's restatement of freqencoder.cu -- same column layout [x | sin 2^0 x | cos 2^0 x | sin 2^1 x | ...], the
    CUDA kernel's cos as sin(. + pi / 2)."""
    g = np.load(os.path.join(golden_dir, "freq_torch.npz"))
    for deg in (4, 6, 10):
        x, y, go, gx = g[f"x{deg}"], g[f"y{deg}"], g[f"g{deg}"], g[f"gx{deg}"]
        B, D, C = x.shape[0], 3, 3 + 6 * deg
        out = orc.freq_encode_forward(x, B, D, deg, C)
        # |argument| <= 2^(deg - 1) + pi / 2: float32 sin of a float32 argument, the phase shift rounds at the argument's ulp
        np.testing.assert_allclose(out, y, rtol=0, atol=2e-7 * 2.0 ** deg + 1e-6)
        gi = orc.freq_encode_backward(go, out, B, D, deg, C)
        np.testing.assert_allclose(gi, gx, rtol=2e-5, atol=2e-4 * 2.0 ** (deg - 4))
