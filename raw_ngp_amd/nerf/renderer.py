"""Volume renderer: density-grid guided ray marching (`run_cuda`) and the proposal sampler (`run`).

Host-side mirror of the reference's nerf/renderer.py with the same method names and result keys:
`render` (:374), `run` (:405-513, pure-torch sampler), `run_cuda` (:515-676), `mark_untrained_grid`
(:716-809), `update_extra_state` (:811-897), and the helper functions `near_far_from_aabb`
(:139-158), `contract` / `uncontract` (:77-99), `sample_pdf` (:102-136), `proposal_loss` (:50-74).
Mesh export and nvdiffrast culling are out of scope.  Per-sample work happens in the HIP kernels
reached through raw_ngp_amd.raymarching and the encoders; this file only orchestrates.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import raymarching


def custom_meshgrid(*args):
    return torch.meshgrid(*args, indexing="ij")


# ----------------------------------------------------------------------------- helper functions

def _fp32(fn):
    """The reference runs these helpers with autocast switched off (`@torch.cuda.amp.autocast(enabled=False)`)."""
    return torch.autocast("cuda", enabled=False)(fn)


@_fp32
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.05):
    """Ray / axis-aligned-box slab test (semantics of reference renderer.py:139-158, the torch flavour `run_cuda`
    uses): per axis the two plane parameters, entry = latest of the per-axis earlier ones, exit = earliest of the
    later ones; a miss is reported as near = far = 1e9; directions are offset by 1e-15 before dividing.
    Returns ([N,1], [N,1]) with near clamped to `min_near`."""
    denom = rays_d + 1e-15                          # (a true division: a reciprocal-multiply differs in the last bit)
    lo, hi = (aabb[:3] - rays_o) / denom, (aabb[3:] - rays_o) / denom
    swap = ~(lo < hi)                               # comparison-based (not fmin/fmax): NaN handling as the reference
    entry = torch.where(swap, hi, lo).amax(-1, keepdim=True)
    swap = ~(lo > hi)
    exit_ = torch.where(swap, hi, lo).amin(-1, keepdim=True)
    hit = ~(exit_ < entry)
    entry = torch.where(hit, entry, entry.new_tensor(1e9))
    exit_ = torch.where(hit, exit_, exit_.new_tensor(1e9))
    return entry.clamp(min=min_near), exit_


def _linf_pick(x):
    """|x|_inf per point and a one-hot mask of the coordinate that attains it (first one on ties, as Tensor.max)."""
    mag, arg = x.abs().max(dim=-1, keepdim=True)
    onehot = torch.arange(x.shape[-1], device=x.device).expand_as(x) == arg
    return mag, onehot


@_fp32
def contract(x):
    """L-inf scene contraction (reference renderer.py:77-87): the unit cube maps to itself; outside it the dominant
    coordinate m becomes sign(m)(2 - 1/|m|) and the others are divided by |m|, so everything lands in [-2, 2]^C."""
    flat = x.reshape(-1, x.shape[-1])
    mag, dominant = _linf_pick(flat)
    factor = torch.where(dominant, (2 - 1 / mag) / mag, 1 / mag)
    return torch.where(mag < 1, flat, flat * factor).reshape(x.shape)


@_fp32
def uncontract(z):
    """Inverse of `contract` (reference renderer.py:89-99), with the reference's 1e-8 floors on the denominators."""
    flat = z.reshape(-1, z.shape[-1])
    mag, dominant = _linf_pick(flat)
    factor = torch.where(dominant, 1 / (2 * mag - mag * mag).clamp(min=1e-8), 1 / (2 - mag).clamp(min=1e-8))
    return torch.where(mag < 1, flat, flat * factor).reshape(z.shape)


@_fp32
def sample_pdf(bins, weights, T, perturb=False):
    """Resample T positions from a piecewise-constant density by inverting its CDF (reference renderer.py:102-136):
    bins [N, T0+1] are interval edges, weights [N, T0] their masses (+0.01 each before normalising); the T quantiles
    are the centres of T equal slices of [0,1], each jittered inside its slice when `perturb`."""
    n_rays, n_int = weights.shape
    mass = weights + 0.01
    cdf = (mass / mass.sum(-1, keepdim=True)).cumsum(-1).clamp(max=1)
    cdf = torch.nn.functional.pad(cdf, (1, 0))                                   # cdf at the T0+1 edges, 0 first
    # slice centres through linspace, like the reference: (arange + 0.5) / T rounds differently in the last bit
    q = torch.linspace(0.5 / T, 1 - 0.5 / T, T, device=weights.device, dtype=cdf.dtype).expand(n_rays, T)
    if perturb:
        q = q + (torch.rand_like(q) - 0.5) / T
    hi = torch.searchsorted(cdf, q.contiguous(), right=True).clamp(0, n_int)     # first edge with cdf > q
    lo = (hi - 1).clamp(0, n_int)
    c_lo, c_hi = cdf.gather(-1, lo), cdf.gather(-1, hi)
    e_lo, e_hi = bins.gather(-1, lo), bins.gather(-1, hi)
    w = ((q - c_lo) / (c_hi - c_lo)).nan_to_num().clamp(0, 1)                   # 0/0 on empty intervals -> 0
    return e_lo + w * (e_hi - e_lo)


@_fp32
def proposal_loss(all_bins, all_weights):
    """Inter-level histogram bound of mip-NeRF 360 between the final level and each proposal."""

    def interlevel(t0, w0, t1, w1):
        cw1 = torch.cat([torch.zeros_like(w1[..., :1]), torch.cumsum(w1, dim=-1)], dim=-1)
        lo = (torch.searchsorted(t1[..., :-1].contiguous(), t0[..., :-1].contiguous(), right=True) - 1)
        lo = lo.clamp(0, w1.shape[-1] - 1)
        hi = torch.searchsorted(t1[..., 1:].contiguous(), t0[..., 1:].contiguous(), right=True)
        hi = hi.clamp(0, w1.shape[-1] - 1)
        w = torch.take_along_dim(cw1[..., 1:], hi, dim=-1) - torch.take_along_dim(cw1[..., :-1], lo, dim=-1)
        return (w0 - w).clamp(min=0) ** 2 / (w0 + 1e-8)

    ref_bins, ref_w = all_bins[-1].detach(), all_weights[-1].detach()
    loss = 0
    for bins, weights in zip(all_bins[:-1], all_weights[:-1]):
        loss = loss + interlevel(ref_bins, ref_w, bins, weights).mean()
    return loss


# ----------------------------------------------------------------------------- renderer

class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.real_bound = opt.bound                       # world-space marching bound
        self.bound = 2 if opt.contract else opt.bound     # grid-query bound
        self.cascade = 1 + math.ceil(math.log2(self.bound))
        self.grid_size = opt.grid_size
        self.min_near = opt.min_near
        self.density_thresh = opt.density_thresh

        b = self.real_bound
        aabb = torch.FloatTensor([-b, -b, -b, b, b, b])
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())

        self.cuda_ray = opt.cuda_ray
        if self.cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer("density_bitfield",
                                 torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self._arena = None
        else:
            self.spacing_fn = lambda x: torch.where(x < 1, x / 2, 1 - 1 / (2 * x))
            self.spacing_fn_inv = lambda x: torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))

    def forward(self, x, d, **kwargs):
        raise NotImplementedError()

    def density(self, x, **kwargs):
        raise NotImplementedError()

    def update_aabb(self, aabb):
        if not torch.is_tensor(aabb):
            aabb = torch.from_numpy(aabb).float()
        self.aabb_train = aabb.clamp(-self.real_bound, self.real_bound).to(self.aabb_train.device)
        self.aabb_infer = self.aabb_train.clone()

    # ------------------------------------------------------------------ dispatch
    def render(self, rays_o, rays_d, **kwargs):
        if self.cuda_ray:
            return self.run_cuda(rays_o, rays_d, **kwargs)
        if self.training:
            return self.run(rays_o, rays_d, **kwargs)
        N, device = rays_o.shape[0], rays_o.device          # staged inference of the sampler path
        out = {"depth": torch.empty(N, device=device), "image": torch.empty(N, 3, device=device),
               "weights_sum": torch.empty(N, device=device)}
        step = self.opt.max_ray_batch
        for head in range(0, N, step):
            part = self.run(rays_o[head:head + step], rays_d[head:head + step], **kwargs)
            for k in out:
                out[k][head:head + step] = part[k]
        return out

    # ------------------------------------------------------------------ proposal sampler (pure torch)
    def run(self, rays_o, rays_d, bg_color=None, perturb=False, cam_near_far=None, shading="full",
            update_proposal=True, **kwargs):
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        N, device = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        if cam_near_far is not None:
            nears = torch.maximum(nears, cam_near_far[:, [0]])
            fars = torch.minimum(fars, cam_near_far[:, [1]])
        if bg_color is None:
            bg_color = 1
        s_near, s_far = self.spacing_fn(nears), self.spacing_fn(fars)
        all_bins, all_weights = [], []
        bins = weights = rgbs = rays_t = xyzs = None
        n_levels = len(self.opt.num_steps)
        for it, T in enumerate(self.opt.num_steps):
            if it == 0:
                bins = torch.linspace(0, 1, T + 1, device=device).unsqueeze(0).expand(N, -1)
                if perturb:
                    bins = (bins + (torch.rand_like(bins) - 0.5) / T).clamp(0, 1)
            else:
                bins = sample_pdf(bins, weights, T + 1, perturb).detach()
            real_bins = self.spacing_fn_inv(s_near * (1 - bins) + s_far * bins)
            rays_t = (real_bins[..., 1:] + real_bins[..., :-1]) / 2
            xyzs = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * rays_t.unsqueeze(2)
            query = contract(xyzs) if self.opt.contract else xyzs
            if it != n_levels - 1:
                with torch.set_grad_enabled(update_proposal):
                    sigmas = self.density(query, proposal=it)["sigma"]
            else:
                dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
                dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
                outputs = self(query, dirs, ldir=None, shading=shading)
                sigmas, rgbs = outputs["sigma"], outputs["color"]
            deltas = real_bins[..., 1:] - real_bins[..., :-1]
            ds = deltas * sigmas
            if self.opt.background == "last_sample":
                ds = torch.cat([ds[..., :-1], torch.full_like(ds[..., -1:], torch.inf)], dim=-1)
            alphas = 1 - torch.exp(-ds)
            acc = torch.cumsum(ds[..., :-1], dim=-1)
            trans = torch.exp(-torch.cat([torch.zeros_like(acc[..., :1]), acc], dim=-1))
            weights = (alphas * trans).nan_to_num_(0)
            if self.training:
                all_bins.append(bins)
                all_weights.append(weights)
        weights_sum = weights.sum(dim=-1)
        depth = (weights * rays_t).sum(dim=-1)
        image = (weights.unsqueeze(-1) * rgbs).sum(dim=-2)
        results = {}
        if self.training:
            results["num_points"] = xyzs.shape[0] * xyzs.shape[1]
            results["weights"] = weights
            if self.opt.lambda_proposal > 0 and update_proposal:
                results["proposal_loss"] = proposal_loss(all_bins, all_weights)
        results["weights_sum"] = weights_sum
        results["depth"] = depth
        results["image"] = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return results

    # ------------------------------------------------------------------ density-grid marcher
    def _march_train(self, rays_o, rays_d, rays_ldir, nears, fars, perturb):
        cap = getattr(self.opt, "arena_capacity", 0)
        common = (rays_o, rays_d, rays_ldir, self.real_bound, self.opt.contract, self.density_bitfield, self.cascade,
                  self.grid_size, nears, fars)
        if cap > 0:
            N = rays_o.shape[0]
            if self._arena is None or self._arena.n_rays < N or (rays_ldir is not None) != (self._arena.ldirs is not None):
                self._arena = raymarching.MarchArena(N, self.opt.max_steps, cap, rays_o.device,
                                                     with_ldirs=rays_ldir is not None)
            return raymarching.march_rays_train_arena(*common, self._arena, perturb, self.opt.dt_gamma,
                                                      self.opt.max_steps)
        return raymarching.march_rays_train(*common, perturb, self.opt.dt_gamma, self.opt.max_steps)

    def run_cuda(self, rays_o, rays_d, rays_ldir=None, bg_color=None, perturb=False, cam_near_far=None,
                 update_proposal=True, shading="full", **kwargs):
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        N, device = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        if rays_o.is_cuda:      # the same slab test as one kernel (tested against the torch expression; the per-row
            from .._lib import engine_backend                      # amax / amin reductions cost 0.6 ms on this stack)
            nears, fars = torch.empty(N, device=device), torch.empty(N, device=device)
            engine_backend.near_far_from_aabb_v2(rays_o.detach().float(), rays_d.detach().float(), aabb, N, self.min_near,
                                                 nears, fars)
        else:
            nears, fars = near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
            nears, fars = nears.squeeze(-1), fars.squeeze(-1)
        if cam_near_far is not None:
            nears = torch.maximum(nears, cam_near_far[:, 0])
            fars = torch.minimum(fars, cam_near_far[:, 1])
        if bg_color is None:
            bg_color = 0
        results = {}
        amp = torch.autocast("cuda", enabled=bool(self.opt.fp16))

        if self.training:
            xyzs, dirs, ts, rays, ldirs = self._march_train(rays_o, rays_d, rays_ldir, nears, fars, perturb)
            # (the floor only matters for the arena's unused rows, which may hold zeros: 0/0 there would reach the
            # weight gradients of a torch MLP as NaN * 0)
            dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True).clamp_min(1e-30)
            with amp:
                outputs = self(xyzs, dirs, ldirs, shading=shading)
            sigmas, rgbs = outputs["sigma"], outputs["color"]
            weights, weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, ts, rays,
                                                                                 self.opt.T_thresh)
            results["num_points"] = xyzs.shape[0]
            results["weights"] = weights
            results["weights_sum"] = weights_sum
            if self.opt.lambda_orientation > 0:
                pos = xyzs.clone().requires_grad_(True)
                normals = torch.autograd.grad(self(pos, dirs, ldirs, shading=shading)["sigma"], pos,
                                              grad_outputs=torch.ones_like(sigmas), retain_graph=True)[0]
                normals = (-torch.nn.functional.normalize(normals, dim=-1) + 1) / 2
                n_dot_v = (normals * -dirs).sum(dim=-1)
                results["orientation_loss"] = torch.mean((weights * torch.clamp(n_dot_v, max=0.0) ** 2).sum(dim=-1))
        else:
            weights_sum, depth, image = self._march_infer(rays_o, rays_d, rays_ldir, nears, fars, perturb, shading,
                                                          normals=False)
            if self.opt.compute_normals:
                _, _, nmap = self._march_infer(rays_o, rays_d, rays_ldir, nears, fars, perturb, shading, normals=True)
                results["normals"] = nmap + (1 - weights_sum).unsqueeze(-1) * bg_color

        results["depth"] = depth
        results["image"] = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return results

    def _march_infer(self, rays_o, rays_d, rays_ldir, nears, fars, perturb, shading, normals):
        """Alive-ray loop of renderer.py:573-616 (and its normal-map twin :618-670)."""
        N, device = rays_o.shape[0], rays_o.device
        weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
        depth = torch.zeros(N, dtype=torch.float32, device=device)
        image = torch.zeros(N, 3, dtype=torch.float32, device=device)
        rays_alive = torch.arange(N, dtype=torch.int32, device=device)
        rays_t = nears.clone()
        amp = torch.autocast("cuda", enabled=bool(self.opt.fp16))
        step = 0
        while step < self.opt.max_steps:
            n_alive = rays_alive.shape[0]
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, ts = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d,
                                                    self.real_bound, self.opt.contract, self.density_bitfield,
                                                    self.cascade, self.grid_size, nears, fars,
                                                    perturb if step == 0 else False, self.opt.dt_gamma,
                                                    self.opt.max_steps)
            dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
            with amp:
                ldirs = rays_ldir.repeat(xyzs.shape[0], 1) if self.opt.rfield else None
                outputs = self(xyzs, dirs, ldirs, shading=shading)
                sigmas, colors = outputs["sigma"], outputs["color"]
                if normals:
                    with torch.enable_grad():
                        pos = xyzs.clone().requires_grad_(True)
                        grad = torch.autograd.grad(self(pos, dirs, ldirs, shading=shading)["sigma"], pos,
                                                   grad_outputs=torch.ones_like(sigmas), retain_graph=True)[0]
                    colors = (-torch.nn.functional.normalize(grad, dim=-1) + 1) / 2
            raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, colors, ts, weights_sum, depth,
                                       image, self.opt.T_thresh)
            rays_alive = rays_alive[rays_alive >= 0]
            step += n_step
        return weights_sum, depth, image

    # ------------------------------------------------------------------ density grid upkeep
    def _cell_chunks(self, S):
        ax = torch.arange(self.grid_size, dtype=torch.int32, device=self.aabb_train.device).split(S)
        for xs in ax:
            for ys in ax:
                for zs in ax:
                    xx, yy, zz = custom_meshgrid(xs, ys, zs)
                    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                    yield coords, raymarching.morton3D(coords).long()

    @torch.no_grad()
    def mark_untrained_grid(self, dataset, S=64):
        """Cells no training camera sees (or outside the AABB) get density -1 and are never sampled."""
        poses, intrinsics = dataset.poses, dataset.intrinsics
        cam_near_far = getattr(dataset, "cam_near_far", None)
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        device = self.aabb_train.device
        poses = poses.to(device)
        B = poses.shape[0]
        if isinstance(intrinsics, np.ndarray):
            fx, fy, cx, cy = [torch.tensor(float(v), device=device) for v in intrinsics]
            per_cam = False
        else:
            fx, fy, cx, cy = [c.to(device) for c in torch.chunk(intrinsics, 4, dim=-1)]
            per_cam = True
        mask_cam = torch.zeros_like(self.density_grid)
        mask_aabb = torch.zeros_like(self.density_grid)
        for coords, indices in self._cell_chunks(S):
            world = (2 * coords.float() / (self.grid_size - 1) - 1).unsqueeze(0)
            for cas in range(self.cascade):
                bound = min(2 ** cas, self.bound)
                half = bound / self.grid_size
                cas_world = world * (bound - half)
                inside = ((cas_world >= (self.aabb_train[:3] - half)).sum(-1) == 3) & \
                         ((cas_world <= (self.aabb_train[3:] + half)).sum(-1) == 3)
                mask_aabb[cas, indices] += inside.reshape(-1)
                for head in range(0, B, S):
                    tail = min(head + S, B)
                    cam = (cas_world - poses[head:tail, :3, 3].unsqueeze(1)) @ poses[head:tail, :3, :3]
                    cam[:, :, 2] *= -1
                    kx = (cx[head:tail] / fx[head:tail]) if per_cam else cx / fx
                    ky = (cy[head:tail] / fy[head:tail]) if per_cam else cy / fy
                    near = self.opt.min_near if cam_near_far is None else cam_near_far[head:tail, 0].unsqueeze(1).to(device)
                    seen = (cam[:, :, 2] > near) & (cam[:, :, 0].abs() < kx * cam[:, :, 2] + half * 2) & \
                           (cam[:, :, 1].abs() < ky * cam[:, :, 2] + half * 2)
                    mask_cam[cas, indices] += seen.sum(0).bool().reshape(-1)
        self.density_grid[(mask_cam == 0) | (mask_aabb == 0)] = -1

    def _cell_centres(self, coords, cas):
        bound = min(2 ** cas, self.bound)
        half = bound / self.grid_size
        xyzs = (2 * coords.float() / (self.grid_size - 1) - 1) * (bound - half)
        return xyzs + (torch.rand_like(xyzs) * 2 - 1) * half

    def update_extra_state(self, decay=0.95, S=128):
        """EMA-max refresh of the density grid (full sweep for the first 16 calls, then 2 x H^3/4 cells per
        cascade) followed by re-packing the bitfield with thresh = min(mean density, density_thresh)."""
        if not self.cuda_ray:
            return
        amp = torch.autocast("cuda", enabled=bool(self.opt.fp16))
        with torch.no_grad():
            tmp = -torch.ones_like(self.density_grid)
            if self.iter_density < 16:
                for coords, indices in self._cell_chunks(S):
                    for cas in range(self.cascade):
                        with amp:
                            sig = self.density(self._cell_centres(coords, cas))["sigma"].reshape(-1).detach()
                        tmp[cas, indices] = sig.float()
            else:
                n = self.grid_size ** 3 // 4
                dev = self.aabb_train.device
                for cas in range(self.cascade):
                    coords = torch.randint(0, self.grid_size, (n, 3), device=dev)
                    indices = raymarching.morton3D(coords).long()
                    occ = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                    if occ.shape[0] > 0:
                        pick = occ[torch.randint(0, occ.shape[0], [n], dtype=torch.long, device=dev)]
                        indices = torch.cat([indices, pick], dim=0)
                        coords = torch.cat([coords, raymarching.morton3D_invert(pick)], dim=0)
                    with amp:
                        sig = self.density(self._cell_centres(coords, cas))["sigma"].reshape(-1).detach()
                    tmp[cas, indices] = sig.float()
            valid = (self.density_grid >= 0) & (tmp >= 0)
            self.density_grid[valid] = torch.maximum(self.density_grid[valid] * decay, tmp[valid])
            self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
            self.iter_density += 1
            thresh = min(self.mean_density, self.density_thresh)
            self.density_bitfield = raymarching.packbits(self.density_grid.detach(), thresh, self.density_bitfield)
            self.bitfield_version = getattr(self, "bitfield_version", 0) + 1   # derived structures (LDS index) rebuild
