"""Ray-marching / compositing autograd ops.

Host-side mirror of the reference's raymarching/raymarching.py: the same ten callables
(`near_far_from_aabb`, `sph_from_ray`, `morton3D`, `morton3D_invert`, `packbits`, `flatten_rays`,
`march_rays_train`, `composite_rays_train`, `march_rays`, `composite_rays`) with the same
positional arguments and return values.  Device work = ngp_* of libngp_hip.so through
`raymarching_backend`.

Differences that stay behind the same interface:
  * `march_rays_train.backward` uses our own segmented sum (ngp_x_march_rays_train_backward)
    where the reference imports torch_scatter.segment_csr (raymarching.py:9,319-329);
  * sample offsets are ray-ordered (deterministic), see include/ngp_hip.h;
  * `march_rays_train_arena` (extra) marches into a caller-owned arena without the `.item()`
    host synchronisation of raymarching.py:303.
"""
import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

from .._lib import raymarching_backend

_FWD32 = dict(device_type="cuda", cast_inputs=torch.float32)


def get_backend():
    return raymarching_backend


def _dev(t):
    return t if t.is_cuda else t.cuda()


def _rays3(t):
    return _dev(t).contiguous().view(-1, 3)


# ----------------------------------------------------------------------------- utils

class _near_far_from_aabb(Function):
    @staticmethod
    @custom_fwd(**_FWD32)
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """rays_o/d [N,3], aabb [6] (xmin,ymin,zmin,xmax,ymax,zmax) -> nears [N], fars [N]."""
        rays_o, rays_d = _rays3(rays_o), _rays3(rays_d)
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty_like(nears)
        get_backend().near_far_from_aabb(rays_o, rays_d, _dev(aabb).contiguous(), N, min_near, nears, fars)
        return nears, fars

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_nears, grad_fars):
        return None, None, None, None


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @custom_fwd(**_FWD32)
    def forward(ctx, rays_o, rays_d, radius):
        """Far intersection with the sphere of `radius` -> (theta, phi) in [-1, 1]^2, [N, 2]."""
        rays_o, rays_d = _rays3(rays_o), _rays3(rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        get_backend().sph_from_ray(rays_o, rays_d, radius, N, coords)
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """coords [N,3] int32 in [0,128) -> Morton codes [N] int32."""
        coords = _dev(coords).int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        get_backend().morton3D(coords, N, indices)
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """Morton codes [N] -> coords [N,3] int32."""
        indices = _dev(indices).int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        get_backend().morton3D_invert(indices, N, coords)
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @custom_fwd(**_FWD32)
    def forward(ctx, grid, thresh, bitfield=None):
        """grid [C, H^3] float -> bitfield [C*H^3/8] uint8, bit i of byte n = grid[8n+i] > thresh."""
        grid = _dev(grid).contiguous()
        N = grid.shape[0] * grid.shape[1] // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        get_backend().packbits(grid, N, thresh, bitfield)
        return bitfield


packbits = _packbits.apply


class _flatten_rays(Function):
    @staticmethod
    def forward(ctx, rays, M):
        """rays [N,2] (offset, count) -> per-sample ray id [M]."""
        rays = _dev(rays).contiguous()
        res = torch.zeros(M, dtype=torch.int, device=rays.device)
        get_backend().flatten_rays(rays, rays.shape[0], M, res)
        return res


flatten_rays = _flatten_rays.apply


# ----------------------------------------------------------------------------- training

def _noises(perturb, n, like):
    make = torch.rand if perturb else torch.zeros
    return make(n, dtype=like.dtype, device=like.device)


def _ray_gradients(ctx, dL_dxyzs, dL_ddirs):
    rays, ts = ctx.saved_tensors
    N, M = rays.shape[0], ts.shape[0]
    g_o = torch.empty(N, 3, dtype=ts.dtype, device=ts.device)
    g_d = torch.empty_like(g_o)
    gx = (dL_dxyzs if dL_dxyzs is not None else torch.zeros(M, 3, dtype=ts.dtype, device=ts.device)).contiguous()
    gd = dL_ddirs.contiguous() if dL_ddirs is not None else None
    get_backend().march_rays_train_backward(gx, gd, ts, rays, N, M, g_o, g_d)
    return g_o, g_d


# march_rays_train: one chain-parallel pass into a cached scratch arena instead of the reference's two serial passes
# (count, then write).  Set to False for the literal two-call protocol of the C ABI.
single_pass = True
_SCRATCH = {}


def _scratch_arena(N, max_steps, bound, dev, with_ldirs):
    import math
    key = (dev.index, N, max_steps, int(math.ceil(bound)), with_ldirs)
    ar = _SCRATCH.get(key)
    if ar is None:
        if len(_SCRATCH) >= 2:          # training + evaluation shapes at most; drop the oldest
            _SCRATCH.pop(next(iter(_SCRATCH)))
        ar = MarchArena(N, max_steps, N * max_steps, dev, with_ldirs=with_ldirs,
                        chain_cap=max_steps * int(math.ceil(bound)) + 2)
        _SCRATCH[key] = ar
    return ar


class _march_rays_train(Function):
    @staticmethod
    @custom_fwd(**_FWD32)
    def forward(ctx, rays_o, rays_d, rays_ldir, bound, contract, density_bitfield, C, H, nears, fars, perturb=False,
                dt_gamma=0, max_steps=1024):
        """Occupancy-guided march.  Returns xyzs [M,3], dirs [M,3], ts [M,2] (t_end, dt),
        rays [N,2] int32 (offset, count), ldirs [M,3] | None."""
        rays_o, rays_d = _rays3(rays_o), _rays3(rays_d)
        rays_ldir = _rays3(rays_ldir) if rays_ldir is not None else None
        bitfield = _dev(density_bitfield).contiguous()
        N = rays_o.shape[0]
        dev, dt = rays_o.device, rays_o.dtype

        noises = _noises(perturb, N, rays_o)
        args = (rays_o, rays_d, rays_ldir, bitfield, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars)
        if single_pass and N > 0 and max_steps * int(-(-bound // 1)) + 2 < 65536:      # (chain codes are 16-bit)
            # one chain-parallel march into a worst-case scratch arena (a ray has at most max_steps samples), then
            # exact-size copies: the same bits as the two calls below (tests compare them), ~ 8 x less marching
            ar = _scratch_arena(N, max_steps, bound, dev, rays_ldir is not None)
            get_backend().march_rays_train_arena(*args, noises, ar.t_scratch, ar.capacity, ar.xyzs, ar.dirs, ar.ts,
                                                 ar.ldirs, ar.rays, ar.counter, ar.ray_idx, None, ar.chain)
            written, needed, chain_overflow = ar.counter[:3].tolist()                         # host sync
            if chain_overflow == 0 and written == needed:
                M = written
                xyzs, dirs, ts = ar.xyzs[:M].clone(), ar.dirs[:M].clone(), ar.ts[:M].clone()
                ldirs = ar.ldirs[:M].clone() if rays_ldir is not None else None
                rays = ar.rays[:N].clone()
                ctx.save_for_backward(rays, ts)
                return xyzs, dirs, ts, rays, ldirs
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        rays = torch.empty(N, 2, dtype=torch.int32, device=dev)
        get_backend().march_rays_train(*args, None, None, None, None, rays, counter, noises)   # count + scan
        M = counter.item()                                                                     # host sync

        xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
        dirs = torch.zeros(M, 3, dtype=dt, device=dev)
        ts = torch.zeros(M, 2, dtype=dt, device=dev)
        ldirs = torch.zeros(M, 3, dtype=dt, device=dev) if rays_ldir is not None else None
        if M > 0:
            get_backend().march_rays_train(*args, xyzs, dirs, ts, ldirs, rays, counter, noises)  # write
        ctx.save_for_backward(rays, ts)
        return xyzs, dirs, ts, rays, ldirs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dL_dxyzs, dL_ddirs, dL_dts, dL_drays, dL_dldirs):
        g_o, g_d = _ray_gradients(ctx, dL_dxyzs, dL_ddirs)
        return (g_o, g_d) + (None,) * 11


march_rays_train = _march_rays_train.apply


class MarchArena:
    """Caller-owned sample arena for `march_rays_train_arena` (fixed capacity, reused every step)."""

    def __init__(self, n_rays, max_steps, capacity, device, with_ldirs=False, chain_cap=0):
        """chain_cap > 0 adds the scratch of the chain-parallel first pass (max_steps * ceil(bound) + 2 is safe)."""
        f32 = dict(dtype=torch.float32, device=device)
        self.n_rays, self.max_steps, self.capacity = n_rays, max_steps, capacity
        self.chain = None
        if chain_cap:
            self.chain = (torch.empty(chain_cap, n_rays, **f32),
                          torch.empty(chain_cap, n_rays, dtype=torch.int16, device=device),
                          torch.zeros(n_rays, dtype=torch.int32, device=device))
        self.t_scratch = torch.empty(n_rays * max_steps, **f32)
        self.xyzs = torch.zeros(capacity, 3, **f32)
        self.dirs = torch.zeros(capacity, 3, **f32)
        self.ts = torch.zeros(capacity, 2, **f32)
        self.ldirs = torch.zeros(capacity, 3, **f32) if with_ldirs else None
        self.rays = torch.zeros(n_rays, 2, dtype=torch.int32, device=device)
        self.ray_idx = torch.zeros(capacity, dtype=torch.int32, device=device)
        self.counter = torch.zeros(4, dtype=torch.int32, device=device)   # [written, needed, chain overflow, -]


class _march_rays_train_arena(Function):
    @staticmethod
    @custom_fwd(**_FWD32)
    def forward(ctx, rays_o, rays_d, rays_ldir, bound, contract, density_bitfield, C, H, nears, fars, arena,
                perturb=False, dt_gamma=0, max_steps=1024, noises=None):
        """Same march, no host sync: fills arena.{xyzs,dirs,ts,ldirs,rays,counter,ray_idx} and returns the
        full-capacity tensors (rows >= arena.counter[0] are stale; every consumer reads the counter)."""
        rays_o, rays_d = _rays3(rays_o), _rays3(rays_d)
        rays_ldir = _rays3(rays_ldir) if rays_ldir is not None else None
        N = rays_o.shape[0]
        assert N <= arena.n_rays and max_steps <= arena.max_steps
        if noises is None:
            noises = _noises(perturb, N, rays_o)
        get_backend().march_rays_train_arena(rays_o, rays_d, rays_ldir, _dev(density_bitfield).contiguous(), bound,
                                             contract, dt_gamma, max_steps, N, C, H, nears, fars, noises,
                                             arena.t_scratch, arena.capacity, arena.xyzs, arena.dirs, arena.ts,
                                             arena.ldirs if rays_ldir is not None else None, arena.rays[:N],
                                             arena.counter, arena.ray_idx)
        rays = arena.rays[:N]
        ctx.save_for_backward(rays, arena.ts)
        ctx.mark_non_differentiable(rays)
        # fresh aliases: autograd attaches its node to the returned objects, never to the arena's own
        ldirs = arena.ldirs[:] if rays_ldir is not None else None
        return arena.xyzs[:], arena.dirs[:], arena.ts[:], rays, ldirs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dL_dxyzs, dL_ddirs, dL_dts, dL_drays, dL_dldirs):
        g_o, g_d = _ray_gradients(ctx, dL_dxyzs, dL_ddirs)
        return (g_o, g_d) + (None,) * 13


march_rays_train_arena = _march_rays_train_arena.apply


wave_compositing = True


def _engine():
    from .._lib import engine_backend
    return engine_backend


class _composite_rays_train(Function):
    @staticmethod
    @custom_fwd(**_FWD32)
    def forward(ctx, sigmas, rgbs, ts, rays, T_thresh=1e-4):
        """sigmas [M], rgbs [M,3], ts [M,2], rays [N,2] -> weights [M], weights_sum [N], depth [N], image [N,3]."""
        sigmas = sigmas.float().contiguous()
        rgbs = rgbs.float().contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev, dt = sigmas.device, sigmas.dtype
        weights = torch.zeros(M, dtype=dt, device=dev)      # samples after an early stop keep 0
        weights_sum = torch.empty(N, dtype=dt, device=dev)
        depth = torch.empty(N, dtype=dt, device=dev)
        image = torch.empty(N, 3, dtype=dt, device=dev)
        # wave-per-ray kernels (ngp_x_*): same contract as the reference-shaped thread-per-ray ones of the C ABI
        # (ngp_composite_rays_train_*; `wave_compositing = False` selects those), ~ 10 x faster on ray-ordered samples
        fwd = _engine().composite_rays_train_forward if wave_compositing else get_backend().composite_rays_train_forward
        fwd(sigmas, rgbs, ts.contiguous(), rays, M, N, T_thresh, weights, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, ts, rays, weights_sum, depth, image)
        ctx.dims = (M, N, T_thresh)
        return weights, weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights, grad_weights_sum, grad_depth, grad_image):
        sigmas, rgbs, ts, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        bwd = _engine().composite_rays_train_backward if wave_compositing else get_backend().composite_rays_train_backward
        if M > 0:
            bwd(grad_weights.contiguous(), grad_weights_sum.contiguous(), grad_depth.contiguous(),
                grad_image.contiguous(), sigmas, rgbs, ts.contiguous(), rays, weights_sum, depth, image, M, N, T_thresh,
                grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


# ----------------------------------------------------------------------------- inference

class _march_rays(Function):
    @staticmethod
    @custom_fwd(**_FWD32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, density_bitfield, C, H,
                near, far, perturb=False, dt_gamma=0, max_steps=1024):
        """March each alive ray by <= n_step samples: xyzs/dirs [n_alive*n_step,3], ts [n_alive*n_step,2];
        unused slots stay zero (ts[:,0] == 0 terminates compositing)."""
        rays_o, rays_d = _rays3(rays_o.float()), _rays3(rays_d.float())
        M = n_alive * n_step
        dev, dt = rays_o.device, rays_o.dtype
        xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
        dirs = torch.zeros(M, 3, dtype=dt, device=dev)
        ts = torch.zeros(M, 2, dtype=dt, device=dev)
        noises = _noises(perturb, n_alive, rays_o)
        get_backend().march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma,
                                 max_steps, C, H, density_bitfield, near, far, xyzs, dirs, ts, noises)
        return xyzs, dirs, ts


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    @custom_fwd(**_FWD32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image,
                T_thresh=1e-2):
        """Accumulates into weights_sum / depth / image in place; finished rays get rays_alive = -1."""
        get_backend().composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.float().contiguous(),
                                     rgbs.float().contiguous(), ts, weights_sum, depth, image)
        return tuple()


composite_rays = _composite_rays.apply
