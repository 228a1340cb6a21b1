"""Reference-generated fixtures against the HIP path (oracle/gen_golden.py ran the reference's own Python in the build
container and wrote tests/golden/*.npz; the reference itself never travels):

  field_plain.npz / field_rfield.npz   NeRFNetwork.forward (network.py:111-143: MLP class, trunc_exp, clamped-exp colour)
                                       and its autograd gradients  ->  the fused MFMA field kernels, forward and backward
  run_weights.npz                      per-sample weights, weights_sum, depth, image of NeRFRenderer.run()'s compositor
                                       (renderer.py:471-495)  ->  ngp_composite_rays_train_forward and the wave kernel

  wrapper_grid.npz / wrapper_sh.npz    the reference's GridEncoder / _grid_encode and SHEncoder / _sh_encoder Python
                                       (grid.py:24-99,149-174; sphere_harmonics.py:14-89) run on CPU over a `_backend`
                                       backed by this repo's CPU oracle  ->  this repo's wrappers over the HIP library:
                                       pins G0 / S0 (bound map, flatten / permute / reshape, double normalisation, dtype
                                       and which inputs get gradients); the kernels' arithmetic is the oracle's either way

  wrapper_raymarching.npz              the reference's raymarching/raymarching.py wrappers (:32-476: near/far, sph, Morton,
                                       packbits, flatten, the two-call march with ldirs, its ray-gradient backward, the
                                       compositing Function forward / backward, one round of the inference pair) run the
                                       same way over the oracle's C functions  ->  this repo's wrappers over HIP: pins R0
                                       (allocation, casting, call protocol, autograd plumbing); torch_scatter.segment_csr
                                       is restated in the generator as the segmented sum its documentation defines

Tolerances: the MFMA kernels round every operand to f16 (the reference's --fp16 autocast precision): 3e-2 relative on the
activated outputs, 5e-2 in the L2 sense on gradients (a ReLU whose pre-activation is within f16 rounding of zero may
flip).  The compositors are f32: 3e-4 relative (prefix products and __expf against exp(-cumsum))."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("tag", ["plain", "rfield", "plain_exp_softplus", "plain_sigmoid", "plain_softplus_hidden",
                                 "rfield_sigmoid_softplus", "rfield_exp"])
def test_fused_field_matches_the_reference_network(golden_dir, tag):
    """The reference's NeRFNetwork.forward + autograd on CPU (oracle/gen_golden.py: its own MLP class and activations) against
    the fused MFMA kernels.  plain_exp_softplus / plain_sigmoid: the reference's other OUTPUT activations (network.py:115,
    131-135: softplus density with beta = 2 + exp colour; sigmoid colour) through ngp_x_mlp_forward_act / _backward_act;
    rfield_*: the same for the light-conditioned field (ngp_x_mlp_rf_forward_act / _backward_act)."""
    from raw_ngp_amd import _lib
    g = np.load(os.path.join(golden_dir, f"field_{tag}.npz"))
    rf = tag.startswith("rfield")
    # (colour, density, beta, hidden layers): plain_softplus_hidden = internal_activation softplus (network.py:31-34) + softplus density
    act = {"plain_exp_softplus": (1, 1, 2.0, 0), "plain_sigmoid": (2, 0, 1.0, 0), "plain_softplus_hidden": (0, 1, 2.0, 1),
           "rfield_sigmoid_softplus": (2, 1, 2.0, 0), "rfield_exp": (1, 0, 1.0, 0)}.get(tag)
    mb = _lib.mlp_rf_backend if rf else _lib.mlp_backend
    W = [dev(g[f"w{i}"]) for i in range(1, 7)]
    assert tuple(W[3].shape) == ((80, 47) if rf else (64, 31))
    M = g["feat"].shape[0]
    enc = dev(g["feat"].reshape(M, 16, 2).transpose(1, 0, 2))                  # [M, 32] (2 l + c) -> level-major slab
    dirs, ldirs = dev(g["dirs"]), dev(g["ldirs"])
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare(W, image)
    sigma, rgb = torch.empty(M, device="cuda"), torch.empty(M, 3, device="cuda")
    if rf:
        mb.forward(enc, M, dirs, ldirs, None, None, M, image, sigma, rgb, act=act)
    else:
        mb.forward(enc, M, dirs, None, M, image, sigma, rgb, act=act)
    np.testing.assert_allclose(sigma.cpu().numpy(), g["sigma"], rtol=3e-2, atol=1e-4)
    np.testing.assert_allclose(rgb.cpu().numpy(), g["color"], rtol=3e-2, atol=1e-4)
    denc = torch.empty(16, M, 2, device="cuda")
    dws = [torch.empty_like(w) for w in W]
    if rf:
        mb.backward(enc, M, dirs, ldirs, None, dev(g["dsigma"]), dev(g["drgb"]), None, M, image, 1024.0, denc, None, dws, act=act)
    else:
        mb.backward(enc, M, dirs, dev(g["dsigma"]), dev(g["drgb"]), None, M, image, 1024.0, denc, dws, act=act)
    got = denc.permute(1, 0, 2).reshape(M, 32).cpu().numpy()
    assert rel_l2(got, g["dfeat"]) < 5e-2, rel_l2(got, g["dfeat"])
    for i in range(6):
        r = rel_l2(dws[i].cpu().numpy(), g[f"gw{i + 1}"])
        assert r < 5e-2, (i + 1, r)


def test_compositors_match_the_reference_renderer(golden_dir):
    from raw_ngp_amd import _lib
    g = np.load(os.path.join(golden_dir, "run_weights.npz"))
    N, T = g["sigma"].shape
    M = N * T
    sig = g["sigma"].reshape(M)
    rgb = g["color"].reshape(M, 3)
    ts = np.stack([g["t_mid"].reshape(M), g["delta"].reshape(M)], 1).astype(np.float32)
    rays = np.stack([np.arange(N) * T, np.full(N, T)], 1).astype(np.int32)
    want_w = g["weights"].reshape(M)
    tol = dict(rtol=3e-4, atol=2e-6)
    for name in ("reference API", "wave"):
        w = torch.full((M,), 7.0, device="cuda")
        ws, dep, img = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
        if name == "wave":
            _lib.engine_backend.composite_rays_train_forward(dev(sig), dev(rgb), dev(ts), dev(rays), M, N, 0.0, w, ws, dep, img)
        else:
            w.zero_()
            _lib.raymarching_backend.composite_rays_train_forward(dev(sig), dev(rgb), dev(ts), dev(rays), M, N, 0.0, w, ws,
                                                                  dep, img)
        np.testing.assert_allclose(w.cpu().numpy(), want_w, err_msg=name, **tol)
        np.testing.assert_allclose(ws.cpu().numpy(), g["weights_sum"], err_msg=name, **tol)
        np.testing.assert_allclose(dep.cpu().numpy(), g["depth"], err_msg=name, rtol=3e-4, atol=1e-5)
        np.testing.assert_allclose(img.cpu().numpy(), g["image"], err_msg=name, **tol)
    assert g["weights_sum"][5] == 0 and g["weights"][6, 5:].max() < 1e-30       # the fixture's empty ray and its wall


@pytest.mark.parametrize("tag,bound,kw", [
    ("b1", 1.0, dict(num_levels=8, log2_hashmap_size=11, desired_resolution=256)),
    ("b2", 2.0, dict(num_levels=6, log2_hashmap_size=10, desired_resolution=512, interpolation="smoothstep"))])
def test_grid_encoder_wrapper_matches_the_reference_wrapper(golden_dir, tag, bound, kw):
    """G0: GridEncoder.forward + _grid_encode forward / backward as the reference's Python does them (fixture generated by
    running that Python over an oracle-backed `_backend`), here over libngp_hip.so.  Encoders are f32: 1e-5 relative on
    outputs (1e-6 kernel parity + the bound map's rounding), 1e-4 on gradients (atomic / binned summation order)."""
    from raw_ngp_amd.gridencoder import GridEncoder
    g = np.load(os.path.join(golden_dir, "wrapper_grid.npz"))
    enc = GridEncoder(input_dim=3, level_dim=2, base_resolution=16, **kw).cuda()
    np.testing.assert_array_equal(enc.offsets.cpu().numpy(), g[f"{tag}_offsets"])
    with torch.no_grad():
        enc.embeddings.copy_(dev(g[f"{tag}_embeddings"]))
    x, gy = dev(g[f"{tag}_x"]), dev(g[f"{tag}_gy"])
    # positions with gradients
    xr = x.clone().requires_grad_(True)
    out = enc(xr, bound=bound)
    assert out.shape == (5, 7, enc.output_dim) and out.dtype == torch.float32
    gx, ge = torch.autograd.grad((out * gy).sum(), [xr, enc.embeddings])
    want = g[f"{tag}_out"]
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-5 * np.abs(want).max())
    assert np.all(out.detach().cpu().numpy()[0, 1] == 0) and np.all(gx.cpu().numpy()[0, 1] == 0)      # outside the volume
    assert gx.dtype == torch.float32 and gx.shape == x.shape
    assert rel_l2(gx.cpu().numpy(), g[f"{tag}_gx"]) < 1e-4
    assert rel_l2(ge.cpu().numpy(), g[f"{tag}_gemb"]) < 1e-4
    # positions without gradients: same outputs, same table gradient, no position gradient
    out2 = enc(x.clone(), bound=bound)
    (ge2,) = torch.autograd.grad((out2 * gy).sum(), [enc.embeddings])
    assert torch.equal(out2, out.detach())
    assert rel_l2(ge2.cpu().numpy(), g[f"{tag}_gemb"]) < 1e-4
    # max_level
    out3 = enc(x.clone(), bound=bound, max_level=3)
    want3 = g[f"{tag}_out_maxlevel3"]
    np.testing.assert_allclose(out3.detach().cpu().numpy(), want3, rtol=1e-5, atol=1e-5 * np.abs(want3).max())
    assert np.all(out3.detach().cpu().numpy()[..., 6:] == 0) and np.any(want3[..., :6] != 0)


@pytest.mark.parametrize("degree", [4, 6])
def test_sh_encoder_wrapper_matches_the_reference_wrapper(golden_dir, degree):
    """S0: SHEncoder.forward (size division, normalisation, flatten, reshape) + _sh_encoder forward / backward: the
    gradient reaches the UN-normalised directions through the module's normalisation.  f32: 1e-5 / 1e-4."""
    from raw_ngp_amd.shencoder import SHEncoder
    g = np.load(os.path.join(golden_dir, "wrapper_sh.npz"))
    she = SHEncoder(input_dim=3, degree=degree)
    d, gy = dev(g[f"d{degree}_dirs"]), dev(g[f"d{degree}_gy"])
    dr = d.clone().requires_grad_(True)
    out = she(dr, size=float(g["size"]))
    assert out.shape == (3, 11, degree ** 2) and out.dtype == torch.float32
    (gd,) = torch.autograd.grad((out * gy).sum(), [dr])
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"d{degree}_out"], rtol=1e-5, atol=1e-5)
    assert rel_l2(gd.cpu().numpy(), g[f"d{degree}_gdirs"]) < 1e-4
    assert torch.equal(she(d.clone(), size=float(g["size"])), out.detach())


def test_raymarching_wrappers_match_the_reference_wrappers(golden_dir):
    """R0: every wrapper of raymarching/raymarching.py the path uses, called like the reference calls it, on the fixture's
    rays and occupancy grid.  Integer outputs (Morton codes, bitfield, ray offsets / counts, sample -> ray ids) bit for bit;
    march positions, ts and directions to 1e-6 (same float expressions, -ffp-contract=off on both sides); compositing 3e-4
    (wave prefix products, __expf); the march's ray gradients 1e-5 relative."""
    from raw_ngp_amd import raymarching as RM
    g = np.load(os.path.join(golden_dir, "wrapper_raymarching.npz"))
    H, max_steps, n_step = int(g["H"]), int(g["max_steps"]), int(g["n_step"])
    # Morton / packbits / flatten
    idx = RM.morton3D(dev(g["morton_coords"]))
    np.testing.assert_array_equal(idx.cpu().numpy(), g["morton_indices"])
    np.testing.assert_array_equal(RM.morton3D_invert(idx).cpu().numpy(), g["morton_roundtrip"])
    grid = dev(g["grid"].astype(np.float32))
    bits = RM.packbits(grid, 5.0)
    assert bits.dtype == torch.uint8
    np.testing.assert_array_equal(bits.cpu().numpy(), g["bitfield"])
    # near / far (the CUDA flavour: 1 / d, miss -> FLT_MAX) and the background-sphere coordinates
    ro, rd, ld = dev(g["rays_o"]), dev(g["rays_d"]), dev(g["rays_ldir"])
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device="cuda")
    nears, fars = RM.near_far_from_aabb(ro, rd, aabb, 0.05)
    np.testing.assert_allclose(nears.cpu().numpy(), g["nears"], rtol=1e-6)
    np.testing.assert_allclose(fars.cpu().numpy(), g["fars"], rtol=1e-6)
    np.testing.assert_allclose(RM.sph_from_ray(ro * 0.3, rd, 1.5).cpu().numpy(), g["sph"], rtol=1e-5, atol=1e-6)
    nears_f, fars_f = dev(g["nears"]), dev(g["fars"])              # (the fixture's own values from here on)
    # the training march, with and without light directions; both call protocols of this repo's op
    for single in (True, False):
        RM.raymarching.single_pass = single
        try:
            for tag, ldir in (("plain", None), ("lit", ld)):
                o_req, d_req = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
                xyzs, dirs, ts, rays, ldirs = RM.march_rays_train(o_req, d_req, ldir, 1.0, False, bits, 1, H, nears_f, fars_f, False,
                                                                  0.0, max_steps)
                np.testing.assert_array_equal(rays.cpu().numpy(), g[f"march_{tag}_rays"])
                np.testing.assert_allclose(xyzs.detach().cpu().numpy(), g[f"march_{tag}_xyzs"], rtol=1e-6, atol=1e-6)
                np.testing.assert_allclose(ts.detach().cpu().numpy(), g[f"march_{tag}_ts"], rtol=1e-6, atol=1e-7)
                np.testing.assert_allclose(dirs.detach().cpu().numpy(), g[f"march_{tag}_dirs"], rtol=1e-6)
                if ldir is not None:
                    np.testing.assert_allclose(ldirs.detach().cpu().numpy(), g["march_lit_ldirs"], rtol=1e-6)
                else:
                    assert ldirs is None
                    g_o, g_d = torch.autograd.grad([xyzs, dirs], [o_req, d_req], [dev(g["march_gxyzs"]), dev(g["march_gdirs"])])
                    assert rel_l2(g_o.cpu().numpy(), g["march_grays_o"]) < 1e-5
                    assert rel_l2(g_d.cpu().numpy(), g["march_grays_d"]) < 1e-5
                    np.testing.assert_array_equal(RM.flatten_rays(rays, xyzs.shape[0]).cpu().numpy(), g["flatten"])
        finally:
            RM.raymarching.single_pass = True
    # compositing through the autograd.Function, T_thresh = its default
    sig, rgb = dev(g["comp_sigmas"]).requires_grad_(True), dev(g["comp_rgbs"]).requires_grad_(True)
    wts, wsum, dep, img = RM.composite_rays_train(sig, rgb, dev(g["march_plain_ts"]), dev(g["march_plain_rays"]))
    for got, key in ((wts, "comp_weights"), (wsum, "comp_weights_sum"), (dep, "comp_depth"), (img, "comp_image")):
        np.testing.assert_allclose(got.detach().cpu().numpy(), g[key], rtol=3e-4, atol=3e-4)
    gsig, grgb = torch.autograd.grad([wsum, dep, img], [sig, rgb],
                                     [dev(g["comp_g_weights_sum"]), dev(g["comp_g_depth"]), dev(g["comp_g_image"])])
    assert rel_l2(grgb.cpu().numpy(), g["comp_grad_rgbs"]) < 1e-3
    assert rel_l2(gsig.cpu().numpy(), g["comp_grad_sigmas"]) < 3e-3
    # one round of the inference pair
    N = ro.shape[0]
    alive = torch.arange(N, dtype=torch.int32, device="cuda")
    rays_t = nears_f.clone()
    xyz_i, dir_i, ts_i = RM.march_rays(N, n_step, alive, rays_t, ro, rd, 1.0, False, bits, 1, H, nears_f, fars_f, False, 0.0, max_steps)
    np.testing.assert_allclose(xyz_i.cpu().numpy(), g["inf_xyzs"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ts_i.cpu().numpy(), g["inf_ts"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dir_i.cpu().numpy(), g["inf_dirs"], rtol=1e-6)
    ws_i, dep_i, img_i = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, 3, device="cuda")
    RM.composite_rays(N, n_step, alive, rays_t, dev(g["inf_sigmas"]), dev(g["inf_rgbs"]), ts_i, ws_i, dep_i, img_i, 1e-2)
    np.testing.assert_array_equal(alive.cpu().numpy(), g["inf_alive"])
    np.testing.assert_allclose(rays_t.cpu().numpy(), g["inf_rays_t"], rtol=1e-6)
    np.testing.assert_allclose(ws_i.cpu().numpy(), g["inf_weights_sum"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(dep_i.cpu().numpy(), g["inf_depth"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(img_i.cpu().numpy(), g["inf_image"], rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("deg", [4, 6, 10])
def test_frequency_encoder_matches_the_references_torch_encoder(golden_dir, deg):
    """The one encoder the reference ALSO holds in Python: FreqEncoder_torch (encoding.py:6-50), run on CPU by
    oracle/gen_golden.py -> freq_torch.npz.  This repo's FreqEncoder module over ngp_freq_encode_forward / _backward
    (the drop-in for freqencoder.cu) on the same inputs: the column layout is the CUDA kernel's, which is the torch
    encoder's [x | sin 2^0 x | cos 2^0 x | sin 2^1 x | ...] -- the layout map is the identity (checked column by column);
    values to float32 sin of a float32 argument (|argument| up to 2^(deg-1) + pi / 2, cos taken as sin(. + pi / 2) like the
    reference kernel), input gradients through the module's autograd.Function."""
    from raw_ngp_amd.freqencoder import FreqEncoder
    g = np.load(os.path.join(golden_dir, "freq_torch.npz"))
    x, y, go, gx = g[f"x{deg}"], g[f"y{deg}"], g[f"g{deg}"], g[f"gx{deg}"]
    enc = FreqEncoder(input_dim=3, degree=deg).cuda()
    assert enc.output_dim == y.shape[1] == 3 + 6 * deg
    xin = dev(x).requires_grad_(True)
    out = enc(xin)
    got = out.detach().cpu().numpy()
    tol = 2e-7 * 2.0 ** deg + 1e-6
    for c in range(y.shape[1]):                                 # column by column: a layout mistake shows as one bad block
        np.testing.assert_allclose(got[:, c], y[:, c], rtol=0, atol=tol, err_msg=f"column {c}")
    (gi,) = torch.autograd.grad((out * dev(go)).sum(), [xin])
    np.testing.assert_allclose(gi.cpu().numpy(), gx, rtol=2e-5, atol=2e-4 * 2.0 ** (deg - 4))
    # leading dimensions pass through like the reference module's (encoding.py: cat along the last one)
    out3 = enc(dev(x).reshape(4, 24, 3))
    assert out3.shape == (4, 24, enc.output_dim) and torch.equal(out3.reshape(-1, enc.output_dim), out.detach())
