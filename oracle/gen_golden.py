"""Generate golden fixtures from the reference's own Python (build container only).

TEST INFRASTRUCTURE.  Run as `python oracle/gen_golden.py` in the container that has
/root/reference mounted; it imports the reference's pure-Python pieces on CPU (third-party
modules that are not installed are replaced by inert MagicMock entries, SURVEY.md section 8c /
Appendix B), feeds them seeded inputs and writes inputs + outputs to tests/golden/*.npz.
Only data is written -- no reference source text.  The reference's CUDA kernels cannot be
built or run here.  The wrapper-level fixtures (`wrapper_grid.npz`, `wrapper_sh.npz`, `wrapper_raymarching.npz`) run the reference's own
`GridEncoder` / `_grid_encode` and `SHEncoder` / `_sh_encoder` Python (gridencoder/grid.py:24-99,149-174,
shencoder/sphere_harmonics.py:14-89) on CPU with `_backend` replaced by a shim over THIS repo's CPU oracle
(oracle/libngp_oracle.so): what they pin is the wrappers' share -- the [-bound, bound] -> [0, 1] map, flatten / permute /
reshape, the double normalisation, which tensors get gradients and in which dtype -- not the kernels' arithmetic.
"""
import argparse
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

STUBS = ["cv2", "mcubes", "trimesh", "tensorboardX", "torch_efficient_distloss", "torch_scatter", "imageio",
         "torchmetrics", "torchmetrics.functional", "torch_ema", "lpips", "rawpy", "pymeshlab", "easydict",
         "_gridencoder", "_shencoder", "_freqencoder", "_raymarching_mob", "nvdiffrast",
         "nvdiffrast.torch", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for name in STUBS:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    sys.path.insert(0, os.path.join(REF, "barf"))
    sys.path.insert(0, REF)

    import torch
    torch.manual_seed(0)
    rng = np.random.default_rng(0)

    # ---------------------------------------------------------------- renderer helpers
    import nerf.renderer as R
    N = 64
    o = rng.normal(size=(N, 3)).astype(np.float32)
    o = 2.5 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = (rng.uniform(-0.6, 0.6, (N, 3)) - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:8] = rng.normal(size=(8, 3)).astype(np.float32)           # some misses
    d[8] = [0.0, 0.0, -1.0]; o[8] = [0.2, 0.3, 2.0]              # axis-parallel ray
    aabb = np.array([-1, -1, -1, 1, 1, 1], dtype=np.float32)
    near, far = R.near_far_from_aabb(torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(aabb), 0.05)
    np.savez(os.path.join(args.out, "near_far_torch.npz"), rays_o=o, rays_d=d, aabb=aabb, min_near=0.05,
             nears=near.numpy(), fars=far.numpy())

    x = rng.uniform(-3, 3, (200, 3)).astype(np.float32)
    z = R.contract(torch.from_numpy(x))
    xr = R.uncontract(z.clone())
    np.savez(os.path.join(args.out, "contract.npz"), x=x, z=z.numpy(), x_roundtrip=xr.numpy())

    bins = np.sort(rng.uniform(0, 1, (16, 33)).astype(np.float32), axis=1)
    wts = rng.uniform(0, 1, (16, 32)).astype(np.float32)
    wts[3] = 0.0
    samp = R.sample_pdf(torch.from_numpy(bins), torch.from_numpy(wts), 17, perturb=False)
    np.savez(os.path.join(args.out, "sample_pdf.npz"), bins=bins, weights=wts, T=17, out=samp.numpy())

    # ---------------------------------------------------------------- get_rays
    import nerf.train_utils as TU
    poses = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    ang = 0.7
    poses[1, :3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    poses[:, :3, 3] = [[0.1, -0.2, 3.0], [2.0, 0.5, 1.5]]
    intr = np.array([5.5, 5.0, 2.0, 1.5], dtype=np.float32)
    res = {}
    for i in range(2):
        r = TU.get_rays(torch.from_numpy(poses[i:i + 1]), intr, 3, 4, -1)
        res[f"rays_o_{i}"] = r["rays_o"].numpy()
        res[f"rays_d_{i}"] = r["rays_d"].numpy()
    np.savez(os.path.join(args.out, "get_rays.npz"), poses=poses, intrinsics=intr, H=3, W=4, **res)

    # ---------------------------------------------------------------- trunc_exp
    import activation as A
    xe = torch.tensor([-100.0, -80.0, -5.0, 0.0, 1.0, 10.0, 80.0, 85.0], requires_grad=True)
    ye = A.trunc_exp(xe)
    ge = torch.autograd.grad(ye.sum(), xe)[0]
    np.savez(os.path.join(args.out, "trunc_exp.npz"), x=xe.detach().numpy(), y=ye.detach().numpy(), grad=ge.numpy())

    # ---------------------------------------------------------------- MLP + field glue
    import nerf.network as NW
    for act in ("relu", "softplus"):
        opt = types.SimpleNamespace(internal_activation=act, beta=2.0)
        torch.manual_seed(1)
        mlp = NW.MLP(32, 16, 64, 3, opt, bias=False)
        xin = torch.randn(10, 32, requires_grad=True)
        yout = mlp(xin)
        gy = torch.randn_like(yout)
        grads = torch.autograd.grad((yout * gy).sum(), [xin] + list(mlp.parameters()))
        np.savez(os.path.join(args.out, f"mlp_{act}.npz"), x=xin.detach().numpy(), y=yout.detach().numpy(),
                 gy=gy.numpy(), gx=grads[0].numpy(),
                 **{f"w{i}": p.detach().numpy() for i, p in enumerate(mlp.parameters())},
                 **{f"gw{i}": g.numpy() for i, g in enumerate(grads[1:])})

    # ---------------------------------------------------------------- GridEncoder offset tables
    import gridencoder.grid as G
    tabs = {}
    for tag, kw in (("bound1", dict(desired_resolution=2048)), ("bound2", dict(desired_resolution=4096)),
                    ("plumbing_L8", dict(num_levels=8, desired_resolution=2048)),
                    ("prop0", dict(num_levels=5, log2_hashmap_size=17, desired_resolution=128)),
                    ("prop1", dict(num_levels=5, log2_hashmap_size=17, desired_resolution=256))):
        enc = G.GridEncoder(input_dim=3, level_dim=2, base_resolution=16,
                            **{"num_levels": 16, "log2_hashmap_size": 19, **kw})
        tabs[f"{tag}_offsets"] = enc.offsets.numpy()
        tabs[f"{tag}_scale"] = np.float64(enc.per_level_scale)
        tabs[f"{tag}_n_params"] = np.int64(int(enc.n_params))
        tabs[f"{tag}_emb_shape"] = np.array(enc.embeddings.shape)
    np.savez(os.path.join(args.out, "grid_offsets.npz"), **tabs)

    # ---------------------------------------------------------------- the encoder WRAPPERS over an oracle-backed _backend
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import oracle as orc

    class OracleGridBackend:
        """`_gridencoder` (gridencoder/src/bindings.cpp:5-9) over the CPU restatement: same positional arguments, results
        written into the caller's tensors."""

        @staticmethod
        def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype,
                                align_corners, interp):
            out, jac = orc.grid_encode_forward(inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(), B, D, C, L,
                                               max_level, float(S), H, dy_dx is not None, gridtype, align_corners, interp)
            if max_level < L:       # the kernel leaves the levels it skips untouched (the wrapper zero-fills them first)
                out[max_level:] = outputs.detach().numpy()[max_level:]
            outputs.copy_(torch.from_numpy(out))
            if dy_dx is not None:
                dy_dx.copy_(torch.from_numpy(jac))

        @staticmethod
        def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, max_level, S, H, dy_dx,
                                 grad_inputs, gridtype, align_corners, interp):
            assert grad.is_contiguous() and tuple(grad.shape) == (L, B, C)
            ge, gi = orc.grid_encode_backward(grad.numpy(), inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(),
                                              B, D, C, L, max_level, float(S), H,
                                              None if dy_dx is None else dy_dx.numpy(), gridtype, align_corners, interp)
            grad_embeddings.add_(torch.from_numpy(ge))          # (atomicAdd into the zero-initialised tensor)
            if grad_inputs is not None:
                grad_inputs.copy_(torch.from_numpy(gi))

    class OracleShBackend:
        @staticmethod
        def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
            out, jac = orc.sh_encode_forward(inputs.detach().numpy(), B, D, C, dy_dx is not None)
            outputs.copy_(torch.from_numpy(out))
            if dy_dx is not None:
                dy_dx.copy_(torch.from_numpy(jac))

        @staticmethod
        def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
            assert grad.is_contiguous()
            grad_inputs.add_(torch.from_numpy(orc.sh_encode_backward(grad.numpy(), inputs.detach().numpy(), B, D, C,
                                                                     dy_dx.numpy())))

    G._backend = OracleGridBackend
    wg = {}
    for tag, bound, kw in (("b1", 1.0, dict(num_levels=8, log2_hashmap_size=11, desired_resolution=256)),
                           ("b2", 2.0, dict(num_levels=6, log2_hashmap_size=10, desired_resolution=512,
                                            interpolation="smoothstep"))):
        torch.manual_seed(41 if bound == 1.0 else 42)
        enc = G.GridEncoder(input_dim=3, level_dim=2, base_resolution=16, **kw)
        gen = torch.Generator().manual_seed(7)
        with torch.no_grad():
            enc.embeddings.copy_(torch.rand(enc.embeddings.shape, generator=gen) * 2 - 1)
        x = (torch.rand(5, 7, 3, generator=gen) * 2 - 1) * bound
        x[0, 0] = torch.tensor([bound, -bound, 0.0])              # on the faces of the volume
        x[0, 1] = torch.tensor([1.01 * bound, 0.0, 0.0])          # outside: zeros, no gradient
        x[0, 2] = torch.tensor([0.0, 0.0, 0.0])
        gy = torch.randn(5, 7, enc.output_dim, generator=gen)
        wg[f"{tag}_embeddings"] = enc.embeddings.detach().numpy().copy()
        wg[f"{tag}_x"], wg[f"{tag}_gy"] = x.numpy().copy(), gy.numpy()
        wg[f"{tag}_offsets"] = enc.offsets.numpy()
        # (1) positions need gradients (pose refinement): outputs + both gradients
        xr = x.clone().requires_grad_(True)
        out = enc(xr, bound=bound)
        gx, ge = torch.autograd.grad((out * gy).sum(), [xr, enc.embeddings])
        wg[f"{tag}_out"], wg[f"{tag}_gx"], wg[f"{tag}_gemb"] = out.detach().numpy(), gx.numpy(), ge.numpy()
        assert out.dtype == torch.float32 and gx.dtype == torch.float32
        # (2) positions without gradients (the default training path): dy_dx is never formed
        out2 = enc(x.clone(), bound=bound)
        (ge2,) = torch.autograd.grad((out2 * gy).sum(), [enc.embeddings])
        assert torch.equal(out2, out) and torch.equal(ge2, ge)      # same outputs, same table gradient
        # (3) max_level: only the first levels are computed, the others read zero
        out3 = enc(x.clone(), bound=bound, max_level=3)
        wg[f"{tag}_out_maxlevel3"] = out3.detach().numpy()
        # (4) double-precision positions are cast to float32 (custom_fwd(cast_inputs=torch.float32)) only under autocast;
        #     without it the reference's op receives them as given: the wrapper's dtype contract is float32 in
    np.savez_compressed(os.path.join(args.out, "wrapper_grid.npz"), **wg)

    import shencoder.sphere_harmonics as SHM
    SHM._backend = OracleShBackend
    ws = {}
    for degree in (4, 6):
        gen = torch.Generator().manual_seed(100 + degree)
        she = SHM.SHEncoder(input_dim=3, degree=degree)
        d = torch.randn(3, 11, 3, generator=gen) * 2.5             # NOT unit vectors: the module normalises (:81)
        gy = torch.randn(3, 11, degree ** 2, generator=gen)
        dr = d.clone().requires_grad_(True)
        out = she(dr, size=2.0)
        (gd,) = torch.autograd.grad((out * gy).sum(), [dr])
        out_plain = she(d.clone(), size=2.0)                       # no gradient wanted: dy_dx is not formed
        assert torch.equal(out_plain, out.detach())
        ws[f"d{degree}_dirs"], ws[f"d{degree}_gy"] = d.numpy(), gy.numpy()
        ws[f"d{degree}_out"], ws[f"d{degree}_gdirs"] = out.detach().numpy(), gd.numpy()
    np.savez(os.path.join(args.out, "wrapper_sh.npz"), size=2.0, **ws)

    # ---------------------------------------------------------------- the raymarching WRAPPERS over an oracle-backed backend
    # raymarching/raymarching.py:32-476 (R0): allocation, casting, the two-call march protocol with its .item() read, the
    # autograd.Function plumbing of compositing, the ray-gradient backward of the march.  `_raymarching_mob` is replaced by a
    # shim that hands the tensors' storage to the CPU oracle's C functions (same positional arguments as
    # raymarching/src/bindings.cpp:5-19, results written in place); `.cuda()` is the identity on this CPU-only box;
    # torch_scatter.segment_csr (not installable: SURVEY 8c) is restated as the segmented sum its documentation defines,
    # out[i] = sum(src[indptr[i] : indptr[i + 1]]).
    import ctypes
    import raymarching.raymarching as RM
    olib = orc.lib()
    P_, cu, cf, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_int

    def dp(t):
        return None if t is None else P_(t.data_ptr())

    class OracleRayBackend:
        @staticmethod
        def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
            olib.orc_near_far_from_aabb(dp(rays_o), dp(rays_d), dp(aabb), cu(N), cf(min_near), dp(nears), dp(fars))

        @staticmethod
        def sph_from_ray(rays_o, rays_d, radius, N, coords):
            olib.orc_sph_from_ray(dp(rays_o), dp(rays_d), cf(radius), cu(N), dp(coords))

        @staticmethod
        def morton3D(coords, N, indices):
            olib.orc_morton3D(dp(coords.contiguous()), cu(N), dp(indices))

        @staticmethod
        def morton3D_invert(indices, N, coords):
            olib.orc_morton3D_invert(dp(indices.contiguous()), cu(N), dp(coords))

        @staticmethod
        def packbits(grid, N, thresh, bitfield):
            olib.orc_packbits(dp(grid), cu(N), cf(thresh), dp(bitfield))

        @staticmethod
        def flatten_rays(rays, N, M, res):
            olib.orc_flatten_rays(dp(rays), cu(N), cu(M), dp(res))

        @staticmethod
        def march_rays_train(rays_o, rays_d, rays_ldir, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars, xyzs,
                             dirs, ts, ldirs, rays, counter, noises):
            olib.orc_march_rays_train(dp(rays_o), dp(rays_d), dp(rays_ldir), dp(grid), cf(bound), ci(int(contract)),
                                      cf(dt_gamma), cu(max_steps), cu(N), cu(C), cu(H), dp(nears), dp(fars), dp(xyzs), dp(dirs),
                                      dp(ts), dp(ldirs), dp(rays), dp(counter), dp(noises))

        @staticmethod
        def composite_rays_train_forward(sigmas, rgbs, ts, rays, M, N, T_thresh, weights, weights_sum, depth, image):
            olib.orc_composite_rays_train_forward(dp(sigmas), dp(rgbs), dp(ts), dp(rays), cu(M), cu(N), cf(T_thresh), dp(weights),
                                                  dp(weights_sum), dp(depth), dp(image))

        @staticmethod
        def composite_rays_train_backward(gw, gws, gd, gi, sigmas, rgbs, ts, rays, weights_sum, depth, image, M, N, T_thresh,
                                          grad_sigmas, grad_rgbs):
            olib.orc_composite_rays_train_backward(dp(gw), dp(gws), dp(gd), dp(gi), dp(sigmas), dp(rgbs), dp(ts), dp(rays),
                                                   dp(weights_sum), dp(depth), dp(image), cu(M), cu(N), cf(T_thresh),
                                                   dp(grad_sigmas), dp(grad_rgbs))

        @staticmethod
        def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps, C, H, grid, nears,
                       fars, xyzs, dirs, ts, noises):
            olib.orc_march_rays(cu(n_alive), cu(n_step), dp(rays_alive), dp(rays_t), dp(rays_o), dp(rays_d), cf(bound),
                                ci(int(contract)), cf(dt_gamma), cu(max_steps), cu(C), cu(H), dp(grid), dp(nears), dp(fars),
                                dp(xyzs), dp(dirs), dp(ts), dp(noises))

        @staticmethod
        def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image):
            olib.orc_composite_rays(cu(n_alive), cu(n_step), cf(T_thresh), dp(rays_alive), dp(rays_t), dp(sigmas), dp(rgbs), dp(ts),
                                    dp(weights_sum), dp(depth), dp(image))

    def segment_csr(src, indptr):
        out = torch.zeros((indptr.numel() - 1,) + tuple(src.shape[1:]), dtype=src.dtype)
        for i in range(indptr.numel() - 1):
            out[i] = src[int(indptr[i]):int(indptr[i + 1])].sum(0)
        return out

    RM._backend = OracleRayBackend
    RM.get_backend = lambda: OracleRayBackend
    RM.segment_csr = segment_csr
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        wr = {}
        rrng = np.random.default_rng(31)
        Hg, Nr, STEPS = 32, 48, 128
        # a procedural occupancy: a ball of radius 0.55 and a slab, Morton-ordered like the density grid
        cc = (np.stack(np.meshgrid(*[np.arange(Hg)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32))
        ctr = (cc + 0.5) / Hg * 2 - 1
        occ = ((np.linalg.norm(ctr, axis=1) < 0.55) | (np.abs(ctr[:, 2] + 0.7) < 0.08)).astype(np.float32)
        m_idx = RM.morton3D(torch.from_numpy(cc))
        pick = rrng.choice(Hg ** 3, 2048, replace=False)          # (the fixture keeps a sample of the codes)
        wr["morton_coords"], wr["morton_indices"] = cc[pick], m_idx.numpy()[pick]
        wr["morton_roundtrip"] = RM.morton3D_invert(m_idx[torch.from_numpy(pick)]).numpy()
        assert np.array_equal(RM.morton3D_invert(m_idx).numpy(), cc)
        grid = torch.zeros(1, Hg ** 3)
        # (values exactly representable in float16, which is how the fixture stores them; none is near the threshold)
        noise16 = rrng.uniform(0, 1, Hg ** 3).astype(np.float16).astype(np.float32)
        grid[0, m_idx.long()] = torch.from_numpy(occ) * 7.0 + torch.from_numpy(noise16)
        grid = grid.to(torch.float16).float()
        bits = RM.packbits(grid, 5.0)
        wr["grid"], wr["bitfield"] = grid.numpy().astype(np.float16), bits.numpy()
        ro = rrng.normal(size=(Nr, 3)).astype(np.float32)
        ro = 2.4 * ro / np.linalg.norm(ro, axis=1, keepdims=True)
        rd = (rrng.uniform(-0.5, 0.5, (Nr, 3)) - ro).astype(np.float32)
        rd /= np.linalg.norm(rd, axis=1, keepdims=True)
        rd[:6] = rrng.normal(size=(6, 3)).astype(np.float32)        # some rays miss the box
        ld = rrng.normal(size=(Nr, 3)).astype(np.float32)
        aabb_t = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32)
        t_ro, t_rd, t_ld = torch.from_numpy(ro), torch.from_numpy(rd), torch.from_numpy(ld)
        nears_c, fars_c = RM.near_far_from_aabb(t_ro, t_rd, aabb_t, 0.05)
        wr.update(rays_o=ro, rays_d=rd, rays_ldir=ld, nears=nears_c.numpy(), fars=fars_c.numpy())
        wr["sph"] = RM.sph_from_ray(t_ro * 0.3, t_rd, 1.5).numpy()
        for tag, ldir in (("plain", None), ("lit", t_ld)):
            o_req, d_req = t_ro.clone().requires_grad_(True), t_rd.clone().requires_grad_(True)
            xyzs, dirs, ts, rays, ldirs = RM.march_rays_train(o_req, d_req, ldir, 1.0, False, bits, 1, Hg, nears_c, fars_c, False, 0.0,
                                                              STEPS)
            wr[f"march_{tag}_xyzs"], wr[f"march_{tag}_dirs"] = xyzs.detach().numpy(), dirs.detach().numpy()
            wr[f"march_{tag}_ts"], wr[f"march_{tag}_rays"] = ts.detach().numpy(), rays.numpy()
            if ldirs is not None:
                wr[f"march_{tag}_ldirs"] = ldirs.detach().numpy()
            if tag == "plain":       # the march's backward: ray gradients from sample gradients (raymarching.py:319-329)
                gx = torch.from_numpy(rrng.normal(size=tuple(xyzs.shape)).astype(np.float32))
                gd_ = torch.from_numpy(rrng.normal(size=tuple(dirs.shape)).astype(np.float32))
                g_o, g_d = torch.autograd.grad([xyzs, dirs], [o_req, d_req], [gx, gd_])
                wr.update(march_gxyzs=gx.numpy(), march_gdirs=gd_.numpy(), march_grays_o=g_o.numpy(), march_grays_d=g_d.numpy())
                wr["flatten"] = RM.flatten_rays(rays, xyzs.shape[0]).numpy()
                # compositing through the autograd.Function (T_thresh: the Function's own default, 1e-4)
                Mm = xyzs.shape[0]
                sig = torch.from_numpy(rrng.lognormal(0.0, 1.5, Mm).astype(np.float32)).requires_grad_(True)
                rgb = torch.from_numpy(rrng.uniform(0, 1, (Mm, 3)).astype(np.float32)).requires_grad_(True)
                wts, wsum, dep, img = RM.composite_rays_train(sig, rgb, ts.detach(), rays)
                gws_, gdep_, gimg_ = [torch.from_numpy(rrng.normal(size=tuple(t.shape)).astype(np.float32)) for t in (wsum, dep, img)]
                gsig, grgb = torch.autograd.grad([wsum, dep, img], [sig, rgb], [gws_, gdep_, gimg_])
                wr.update(comp_sigmas=sig.detach().numpy(), comp_rgbs=rgb.detach().numpy(), comp_weights=wts.detach().numpy(),
                          comp_weights_sum=wsum.detach().numpy(), comp_depth=dep.detach().numpy(), comp_image=img.detach().numpy(),
                          comp_g_weights_sum=gws_.numpy(), comp_g_depth=gdep_.numpy(), comp_g_image=gimg_.numpy(),
                          comp_grad_sigmas=gsig.numpy(), comp_grad_rgbs=grgb.numpy())
        # the inference pair, one round as renderer.py:575-616 drives it
        n_alive, n_step = Nr, 4
        alive = torch.arange(Nr, dtype=torch.int32)
        rays_t = nears_c.clone()
        xyz_i, dir_i, ts_i = RM.march_rays(n_alive, n_step, alive, rays_t, t_ro, t_rd, 1.0, False, bits, 1, Hg, nears_c, fars_c, False,
                                           0.0, STEPS)
        sig_i = torch.from_numpy(rrng.lognormal(0.0, 1.5, n_alive * n_step).astype(np.float32))
        rgb_i = torch.from_numpy(rrng.uniform(0, 1, (n_alive * n_step, 3)).astype(np.float32))
        ws_i, dep_i, img_i = torch.zeros(Nr), torch.zeros(Nr), torch.zeros(Nr, 3)
        RM.composite_rays(n_alive, n_step, alive, rays_t, sig_i, rgb_i, ts_i, ws_i, dep_i, img_i, 1e-2)
        wr.update(inf_xyzs=xyz_i.numpy(), inf_dirs=dir_i.numpy(), inf_ts=ts_i.numpy(), inf_sigmas=sig_i.numpy(),
                  inf_rgbs=rgb_i.numpy(), inf_alive=alive.numpy(), inf_rays_t=rays_t.numpy(), inf_weights_sum=ws_i.numpy(),
                  inf_depth=dep_i.numpy(), inf_image=img_i.numpy())
        np.savez_compressed(os.path.join(args.out, "wrapper_raymarching.npz"), H=Hg, max_steps=STEPS, n_step=n_step, **wr)
    finally:
        torch.Tensor.cuda = _cuda

    # ---------------------------------------------------------------- checkpoint layout (state_dict keys/shapes)
    import json
    import nerf.network as NW
    layouts = {}
    for tag, kw in (("cuda_ray", dict(cuda_ray=True)), ("sampler", dict(cuda_ray=False)),
                    ("cuda_ray_rfield", dict(cuda_ray=True, rfield=True))):
        o_ = types.SimpleNamespace(**{**dict(
            bound=1.0, cuda_ray=True, min_near=0.05, density_thresh=10, bg_radius=-1, pose_opt="none", rfield=False,
            hashmap_size=19, hashgrid_resolution=2048, contract=False, grid_size=128, device="cpu", fp16=False,
            num_cameras=10, softplus=False, activation="relu", clamped_exp=False, color_act="exp", real_bound=1.0), **kw})
        net = NW.NeRFNetwork(o_)
        layouts[tag] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()]
        del net
    with open(os.path.join(args.out, "state_dict_layout.json"), "w") as f:
        json.dump(layouts, f, indent=1)

    # ---------------------------------------------------------------- BARF / BAA-NGP level windows
    base = dict(bound=1.0, contract=False, grid_size=128, min_near=0.05, density_thresh=10, cuda_ray=True,
                hashmap_size=19, hashgrid_resolution=2048, rfield=False, internal_activation="relu", beta=2.0,
                density_activation="clamped_exp", color_activation="clamped_exp", device="cpu",
                start_annealing=0.0, end_annealing=0.33)
    wres = {}
    for mode in ("barf", "baangp"):
        opt = types.SimpleNamespace(pose_opt="none", **base)
        net = NW.NeRFNetwork(opt)
        opt.pose_opt = mode
        feat = torch.arange(1, 33, dtype=torch.float32).repeat(4, 1) * 0.1

        class FixedFeat(torch.nn.Module):
            def forward(self, x, bound=1):
                return feat.clone()

        net.grid_encoder = FixedFeat()
        seen = []
        net.grid_mlp.register_forward_pre_hook(lambda m, inp: seen.append(inp[0].detach().clone()))
        for ann in (0.0, 0.1, 0.33, 1.0):
            net.update_annealing(np.float16(ann))
            seen.clear()
            net.common_forward(torch.zeros(4, 3))
            wres[f"{mode}_{ann}"] = seen[0].numpy()
    np.savez(os.path.join(args.out, "level_windows.npz"), feat=feat.numpy(), **wres)

    # ---------------------------------------------------------------- the field as the reference evaluates it
    # NeRFNetwork.forward (network.py:111-143) with its own MLP class, trunc_exp and clamped-exp colour, for the plain
    # (31 -> 64 -> 64 -> 3) and the light-conditioned (47 -> 80 -> 80 -> 3) view MLP: inputs, outputs and the gradients
    # of <dsigma, sigma> + <drgb, color>.  The two CUDA-backed encoders are replaced by stand-ins that return given
    # tensors: the hash features are an input of the fixture, the SH features come from the closed-form degree-4
    # polynomials (tests pin those against the kernels separately).  Consumed by the fused MFMA kernels' GPU tests.
    def sh16(d):
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

    class GivenFeatures(torch.nn.Module):
        def __init__(self, feat):
            super().__init__()
            self.feat = feat

        def forward(self, x, bound=1):
            return self.feat

    class ShOfUnit(torch.nn.Module):
        def forward(self, d):
            return sh16(d)

    # (the plain field also with the reference's other OUTPUT activations, network.py:115,131-135: softplus density with
    # beta = 2, exp and sigmoid colour, and softplus HIDDEN layers (network.py:31-34) -- what ngp_x_mlp_forward_act /
    # ngp_x_mlp_backward_act implement)
    for tag, rfield, acts in (("plain", False, {}), ("rfield", True, {}),
                              ("plain_exp_softplus", False, dict(color_activation="exp", density_activation="softplus", beta=2.0)),
                              ("plain_sigmoid", False, dict(color_activation="sigmoid")),
                              ("plain_softplus_hidden", False, dict(internal_activation="softplus", density_activation="softplus",
                                                                   beta=2.0)),
                              # (the light-conditioned field with the other output activations: ngp_x_mlp_rf_forward_act / _backward_act)
                              ("rfield_sigmoid_softplus", True, dict(color_activation="sigmoid", density_activation="softplus",
                                                                     beta=2.0)),
                              ("rfield_exp", True, dict(color_activation="exp"))):
        fopt = types.SimpleNamespace(**{**base, "rfield": rfield, "pose_opt": "none", **acts})
        torch.manual_seed(3 if rfield else 2)
        net = NW.NeRFNetwork(fopt)
        Mf = 96
        g = torch.Generator().manual_seed(17)
        feat = (torch.randn(Mf, 32, generator=g) * 0.5).requires_grad_(True)
        dirs = torch.nn.functional.normalize(torch.randn(Mf, 3, generator=g), dim=-1)
        ldirs = torch.nn.functional.normalize(torch.randn(Mf, 3, generator=g), dim=-1)
        net.grid_encoder = GivenFeatures(feat)
        net.view_encoder = ShOfUnit()
        out = net(torch.zeros(Mf, 3), dirs, ldirs if rfield else None)
        dsig = torch.randn(Mf, generator=g) * 1e-3
        drgb = torch.randn(Mf, 3, generator=g) * 1e-3
        params = [l.weight for l in net.grid_mlp.net] + [l.weight for l in net.view_mlp.net]
        grads = torch.autograd.grad((out["sigma"] * dsig).sum() + (out["color"] * drgb).sum(), [feat] + params)
        np.savez(os.path.join(args.out, f"field_{tag}.npz"), feat=feat.detach().numpy(), dirs=dirs.numpy(),
                 ldirs=ldirs.numpy(), sigma=out["sigma"].detach().numpy(), color=out["color"].detach().numpy(),
                 dsigma=dsig.numpy(), drgb=drgb.numpy(), dfeat=grads[0].numpy(),
                 **{f"w{i + 1}": p.detach().numpy() for i, p in enumerate(params)},
                 **{f"gw{i + 1}": g_.numpy() for i, g_ in enumerate(grads[1:])})

    # ---------------------------------------------------------------- run() with an analytic field
    class Analytic(R.NeRFRenderer):
        def density(self, x, proposal=-1, **kw):
            r2 = (x ** 2).sum(-1)
            return {"sigma": 30.0 * torch.exp(-3.0 * r2) * (1.0 + 0.5 * (proposal + 1))}

        def forward(self, x, d, **kw):
            r2 = (x ** 2).sum(-1)
            return {"sigma": 30.0 * torch.exp(-3.0 * r2),
                    "color": torch.sigmoid(3.0 * x) * (0.75 + 0.25 * d[..., :1])}

    ropt = types.SimpleNamespace(bound=1.0, contract=False, grid_size=128, min_near=0.05, density_thresh=10,
                                 cuda_ray=False, num_steps=[64, 32, 16], background="black", lambda_proposal=0.0,
                                 lambda_distort=0.0, max_ray_batch=4096)
    ren = Analytic(ropt)
    ren.eval()
    NR = 256
    ro = rng.normal(size=(NR, 3)).astype(np.float32)
    ro = 2.2 * ro / np.linalg.norm(ro, axis=1, keepdims=True)
    rd = (rng.uniform(-0.5, 0.5, (NR, 3)) - ro).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    with torch.no_grad():
        outp = ren.run(torch.from_numpy(ro), torch.from_numpy(rd), bg_color=None, perturb=False)
    np.savez(os.path.join(args.out, "run_analytic.npz"), rays_o=ro, rays_d=rd, num_steps=np.array([64, 32, 16]),
             image=outp["image"].numpy(), depth=outp["depth"].numpy(), weights_sum=outp["weights_sum"].numpy())
    # ---------------------------------------------------------------- per-sample weights of run()'s compositor
    # renderer.py:471-495 (alphas, exclusive cumsum transmittance, weights, weights_sum, depth, image) on seeded densities
    # and colours, one sampling level, training mode (results['weights']).  The HIP compositors are compared with it at
    # T_thresh = 0 with ts = (interval midpoint, interval length).
    class Seeded(R.NeRFRenderer):
        def forward(self, x, d, **kw):
            return {"sigma": self._sigma, "color": self._color}

    Tc, Nc = 48, 40
    copt = types.SimpleNamespace(**{**vars(ropt), "num_steps": [Tc]})
    cren = Seeded(copt)
    cren.train()
    crng = np.random.default_rng(23)
    cren._sigma = torch.from_numpy(crng.lognormal(0.0, 2.0, (Nc, Tc)).astype(np.float32))
    cren._sigma[5] = 0.0                                                     # an empty ray
    cren._sigma[6, 3:] = 1e4                                                 # a wall: transmittance underflows behind it
    cren._color = torch.from_numpy(crng.uniform(0, 1, (Nc, Tc, 3)).astype(np.float32))
    co = crng.normal(size=(Nc, 3)).astype(np.float32)
    co = 2.2 * co / np.linalg.norm(co, axis=1, keepdims=True)
    cd = (crng.uniform(-0.5, 0.5, (Nc, 3)) - co).astype(np.float32)
    cd /= np.linalg.norm(cd, axis=1, keepdims=True)
    with torch.no_grad():
        cout = cren.run(torch.from_numpy(co), torch.from_numpy(cd), bg_color=0, perturb=False)
        # the sample positions run() used: its own spacing functions on the uniform bins (renderer.py:441-452)
        nears, fars = R.near_far_from_aabb(torch.from_numpy(co), torch.from_numpy(cd), cren.aabb_train, cren.min_near)
        bins = torch.linspace(0, 1, Tc + 1).unsqueeze(0).expand(Nc, -1)
        real = cren.spacing_fn_inv(cren.spacing_fn(nears) * (1 - bins) + cren.spacing_fn(fars) * bins)
    np.savez(os.path.join(args.out, "run_weights.npz"), sigma=cren._sigma.numpy(), color=cren._color.numpy(),
             t_mid=((real[:, 1:] + real[:, :-1]) / 2).numpy(), delta=(real[:, 1:] - real[:, :-1]).numpy(),
             weights=cout["weights"].numpy(), weights_sum=cout["weights_sum"].numpy(), depth=cout["depth"].numpy(),
             image=cout["image"].numpy())

    # ---------------------------------------------------------------- pose refinement (barf/camera.py)
    import barf.camera as CAM
    prng = np.random.default_rng(11)
    wu = prng.normal(size=(48, 6)).astype(np.float32)
    wu[:8, :3] *= 1e-4                       # small-angle branch
    wu[8:16, :3] *= 1e-9
    wu[16] = 0.0
    wu[40:, :3] *= 2.0                       # up to a few radians
    a34 = CAM.lie.se3_to_SE3(torch.from_numpy(prng.normal(size=(48, 6)).astype(np.float32)))
    with torch.no_grad():
        SE3 = CAM.lie.se3_to_SE3(torch.from_numpy(wu))
        comp = CAM.pose.compose([SE3, a34])
    np.savez(os.path.join(args.out, "pose_lie.npz"), wu=wu, SE3=SE3.numpy(), other=a34.numpy(), composed=comp.numpy())
    # ---------------------------------------------------------------- HDR loss weights (raw/raw_utils.py:30-53)
    # what train_step multiplies the squared residuals with (train_utils.py:520-527): called as it calls them
    import raw.raw_utils as RU
    wrng = np.random.default_rng(17)
    gt = wrng.uniform(0.0, 1.6, (257, 3)).astype(np.float32)
    gt[:5] = [[0.0, 0.5, 1.0], [1.45, 1.4499, 1.4501], [-0.45, -0.4501, -0.4499], [0.05, 0.95, 0.5], [1.0, 1.0, 1.0]]
    tg = torch.from_numpy(gt)
    np.savez(os.path.join(args.out, "loss_weights.npz"), gt_rgb=gt, gaussian=RU.gaussian_weighting(tg).numpy(),
             planck=RU.planck_taper_weighting(tg).numpy(), hanning=RU.hanning_weighting(tg).numpy())
    # ---------------------------------------------------------------- frequency encoder (encoding.py:6-50)
    # FreqEncoder_torch -- the one encoder the reference also holds in Python -- as get_encoder('frequency_torch') builds it
    # (max_freq_log2 = multires - 1, N_freqs = multires, log sampling: bands 2^0 .. 2^(multires-1)); its output layout
    # [x | sin f0 x | cos f0 x | sin f1 x | ...] (blocks of input_dim) is the CUDA kernel's (freqencoder.cu:48-57: column
    # c -> block c / D - 1 = 2 freq + (0 sin, 1 cos), component c % D): the layout map is the identity
    import encoding as E
    frng = np.random.default_rng(23)
    fq = {}
    for multires in (4, 6, 10):
        enc, out_dim = E.get_encoder("frequency_torch", input_dim=3, multires=multires)
        xin = torch.from_numpy(frng.uniform(-1.0, 1.0, (96, 3)).astype(np.float32)).requires_grad_(True)
        yout = enc(xin)
        assert yout.shape[-1] == out_dim == 3 + 3 * 2 * multires
        gout = torch.from_numpy(frng.normal(size=yout.shape).astype(np.float32))
        yout.backward(gout)
        fq.update({f"x{multires}": xin.detach().numpy(), f"y{multires}": yout.detach().numpy(), f"g{multires}": gout.numpy(),
                   f"gx{multires}": xin.grad.numpy()})
    np.savez(os.path.join(args.out, "freq_torch.npz"), **fq)
    print("fixtures written to", args.out)
    for f in sorted(os.listdir(args.out)):
        print("  ", f, os.path.getsize(os.path.join(args.out, f)))


if __name__ == "__main__":
    main()
