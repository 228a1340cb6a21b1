"""ctypes front-end of the CPU oracle (oracle/ngp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by raw_ngp_amd/.  Every function takes and
returns numpy arrays and mirrors the argument order of the reference's `_backend`
functions so that parity tests read like calls into the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libngp_oracle.so")

c_f = ctypes.c_float
c_u = ctypes.c_uint32
c_i = ctypes.c_int
P = ctypes.c_void_p


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "ngp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libngp_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(P)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return int(lib().orc_max_threads())


# ----------------------------------------------------------------------------- grid

def grid_resolutions(S, H, L):
    res = np.zeros(L, dtype=np.uint32)
    lib().orc_grid_resolutions(c_f(S), c_u(H), c_u(L), _p(res))
    return res


def grid_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2.0, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None):
    """Restates gridencoder/grid.py:106-134 (float64 numpy, as the reference does)."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, resolution ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), float(per_level_scale)


def grid_encode_forward(inputs, embeddings, offsets, B, D, C, L, max_level, S, H, want_dy_dx=False,
                        gridtype=0, align_corners=False, interp=0):
    inputs, embeddings, offsets = _f32(inputs), _f32(embeddings), _i32(offsets)
    outputs = np.zeros((L, B, C), dtype=np.float32)
    dy_dx = np.zeros((B, L * D * C), dtype=np.float32) if want_dy_dx else None
    lib().orc_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), c_u(B), c_u(D),
                                  c_u(C), c_u(L), c_u(max_level), c_f(S), c_u(H), _p(dy_dx),
                                  c_u(gridtype), c_i(int(align_corners)), c_u(interp))
    return outputs, dy_dx


def grid_encode_backward(grad, inputs, embeddings, offsets, B, D, C, L, max_level, S, H, dy_dx=None,
                         gridtype=0, align_corners=False, interp=0):
    grad, inputs, embeddings, offsets = _f32(grad), _f32(inputs), _f32(embeddings), _i32(offsets)
    dy_dx = _f32(dy_dx)
    grad_emb = np.zeros_like(embeddings)
    grad_in = np.zeros((B, D), dtype=np.float32) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_emb),
                                   c_u(B), c_u(D), c_u(C), c_u(L), c_u(max_level), c_f(S), c_u(H),
                                   _p(dy_dx), _p(grad_in), c_u(gridtype), c_i(int(align_corners)),
                                   c_u(interp))
    return grad_emb, grad_in


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype=0,
                         align_corners=False):
    inputs, embeddings, offsets = _f32(inputs), _f32(embeddings), _i32(offsets)
    grad = _f32(grad).copy()
    lib().orc_grad_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets), c_f(weight),
                                   c_u(B), c_u(D), c_u(C), c_u(L), c_f(S), c_u(H), c_u(gridtype),
                                   c_i(int(align_corners)))
    return grad


def grad_weight_decay(embeddings, grad, offsets, weight, B, C, L):
    embeddings, offsets = _f32(embeddings), _i32(offsets)
    grad = _f32(grad).copy()
    lib().orc_grad_weight_decay(_p(embeddings), _p(grad), _p(offsets), c_f(weight), c_u(B), c_u(C), c_u(L))
    return grad


# ----------------------------------------------------------------------------- sh / freq

def sh_encode_forward(inputs, B, D, degree, want_dy_dx=False):
    inputs = _f32(inputs)
    out = np.zeros((B, degree * degree), dtype=np.float32)
    dy_dx = np.zeros((B, D * degree * degree), dtype=np.float32) if want_dy_dx else None
    lib().orc_sh_encode_forward(_p(inputs), _p(out), c_u(B), c_u(D), c_u(degree), _p(dy_dx))
    return out, dy_dx


def sh_encode_backward(grad, inputs, B, D, degree, dy_dx):
    grad, inputs, dy_dx = _f32(grad), _f32(inputs), _f32(dy_dx)
    gi = np.zeros((B, D), dtype=np.float32)
    lib().orc_sh_encode_backward(_p(grad), _p(inputs), c_u(B), c_u(D), c_u(degree), _p(dy_dx), _p(gi))
    return gi


def freq_encode_forward(inputs, B, D, deg, C):
    inputs = _f32(inputs)
    out = np.zeros((B, C), dtype=np.float32)
    lib().orc_freq_encode_forward(_p(inputs), c_u(B), c_u(D), c_u(deg), c_u(C), _p(out))
    return out


def freq_encode_backward(grad, outputs, B, D, deg, C):
    grad, outputs = _f32(grad), _f32(outputs)
    gi = np.zeros((B, D), dtype=np.float32)
    lib().orc_freq_encode_backward(_p(grad), _p(outputs), c_u(B), c_u(D), c_u(deg), c_u(C), _p(gi))
    return gi


# ----------------------------------------------------------------------------- raymarching

def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near):
    rays_o, rays_d, aabb = _f32(rays_o), _f32(rays_d), _f32(aabb)
    nears = np.zeros(N, dtype=np.float32)
    fars = np.zeros(N, dtype=np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), c_u(N), c_f(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius, N):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    coords = np.zeros((N, 2), dtype=np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), c_f(radius), c_u(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _i32(coords)
    N = coords.shape[0]
    out = np.zeros(N, dtype=np.int32)
    lib().orc_morton3D(_p(coords), c_u(N), _p(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices)
    N = indices.shape[0]
    out = np.zeros((N, 3), dtype=np.int32)
    lib().orc_morton3D_invert(_p(indices), c_u(N), _p(out))
    return out


def packbits(grid, thresh):
    grid = _f32(grid)
    N = grid.size // 8
    out = np.zeros(N, dtype=np.uint8)
    lib().orc_packbits(_p(grid), c_u(N), c_f(thresh), _p(out))
    return out


def flatten_rays(rays, M):
    rays = _i32(rays)
    N = rays.shape[0]
    out = np.zeros(M, dtype=np.int32)
    lib().orc_flatten_rays(_p(rays), c_u(N), c_u(M), _p(out))
    return out


def march_rays_train(rays_o, rays_d, rays_ldir, grid, bound, contract, dt_gamma, max_steps, C, H, nears,
                     fars, noises):
    """Both passes of raymarching.py:292-317; returns xyzs, dirs, ts, rays, ldirs, M."""
    rays_o, rays_d, nears, fars, noises = map(_f32, (rays_o, rays_d, nears, fars, noises))
    rays_ldir = _f32(rays_ldir)
    grid = np.ascontiguousarray(grid, dtype=np.uint8)
    N = rays_o.shape[0]
    rays = np.zeros((N, 2), dtype=np.int32)
    counter = np.zeros(1, dtype=np.int32)
    args = (_p(rays_o), _p(rays_d), _p(rays_ldir), _p(grid), c_f(bound), c_i(int(contract)), c_f(dt_gamma),
            c_u(max_steps), c_u(N), c_u(C), c_u(H), _p(nears), _p(fars))
    lib().orc_march_rays_train(*args, None, None, None, None, _p(rays), _p(counter), _p(noises))
    M = int(counter[0])
    xyzs = np.zeros((M, 3), dtype=np.float32)
    dirs = np.zeros((M, 3), dtype=np.float32)
    ts = np.zeros((M, 2), dtype=np.float32)
    ldirs = np.zeros((M, 3), dtype=np.float32) if rays_ldir is not None else None
    lib().orc_march_rays_train(*args, _p(xyzs), _p(dirs), _p(ts), _p(ldirs), _p(rays), _p(counter), _p(noises))
    return xyzs, dirs, ts, rays, ldirs, M


def composite_rays_train_forward(sigmas, rgbs, ts, rays, M, N, T_thresh):
    sigmas, rgbs, ts, rays = _f32(sigmas), _f32(rgbs), _f32(ts), _i32(rays)
    weights = np.zeros(M, dtype=np.float32)
    ws = np.zeros(N, dtype=np.float32)
    depth = np.zeros(N, dtype=np.float32)
    image = np.zeros((N, 3), dtype=np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(ts), _p(rays), c_u(M), c_u(N),
                                           c_f(T_thresh), _p(weights), _p(ws), _p(depth), _p(image))
    return weights, ws, depth, image


def composite_rays_train_backward(grad_weights, grad_ws, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                  weights_sum, depth, image, M, N, T_thresh):
    a = [_f32(x) for x in (grad_weights, grad_ws, grad_depth, grad_image, sigmas, rgbs, ts)]
    rays = _i32(rays)
    b = [_f32(x) for x in (weights_sum, depth, image)]
    gs = np.zeros(M, dtype=np.float32)
    gc = np.zeros((M, 3), dtype=np.float32)
    lib().orc_composite_rays_train_backward(*[_p(x) for x in a], _p(rays), *[_p(x) for x in b], c_u(M),
                                            c_u(N), c_f(T_thresh), _p(gs), _p(gc))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps,
               C, H, grid, nears, fars, noises):
    rays_alive = _i32(rays_alive)
    rays_t, rays_o, rays_d, nears, fars, noises = map(_f32, (rays_t, rays_o, rays_d, nears, fars, noises))
    grid = np.ascontiguousarray(grid, dtype=np.uint8)
    M = n_alive * n_step
    xyzs = np.zeros((M, 3), dtype=np.float32)
    dirs = np.zeros((M, 3), dtype=np.float32)
    ts = np.zeros((M, 2), dtype=np.float32)
    lib().orc_march_rays(c_u(n_alive), c_u(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d),
                         c_f(bound), c_i(int(contract)), c_f(dt_gamma), c_u(max_steps), c_u(C), c_u(H),
                         _p(grid), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(ts), _p(noises))
    return xyzs, dirs, ts


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image):
    """In-place on rays_alive, rays_t, weights_sum, depth, image (must be contiguous numpy arrays of
    int32 / float32), like raymarching.py:447-468."""
    sigmas, rgbs, ts = _f32(sigmas), _f32(rgbs), _f32(ts)
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32),
                  (depth, np.float32), (image, np.float32)):
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"]
    lib().orc_composite_rays(c_u(n_alive), c_u(n_step), c_f(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas),
                             _p(rgbs), _p(ts), _p(weights_sum), _p(depth), _p(image))


def march_rays_train_backward(grad_xyzs, grad_dirs, ts, rays, N, M):
    grad_xyzs, grad_dirs, ts, rays = _f32(grad_xyzs), _f32(grad_dirs), _f32(ts), _i32(rays)
    go = np.zeros((N, 3), dtype=np.float32)
    gd = np.zeros((N, 3), dtype=np.float32)
    lib().orc_march_rays_train_backward(_p(grad_xyzs), _p(grad_dirs), _p(ts), _p(rays), c_u(N), c_u(M),
                                        _p(go), _p(gd))
    return go, gd


# ----------------------------------------------------------------------------- ray batch sampling (engine extension)
def philox4x32_10(counter, key):
    """Philox4x32-10 of Salmon et al. (SC'11, Random123): counter [..., 4] uint32, key (k0, k1) -> [..., 4] uint32."""
    c = np.array(counter, dtype=np.uint64).reshape(-1, 4).copy()
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], 1)
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32).reshape(np.shape(counter))


def sample_rays(images, poses, intrinsics, N, seed, draw):
    """What ngp_x_sample_rays draws and builds: the harness' random (view, pixel) per ray, get_rays
    (nerf/train_utils.py:96-172: pixel centre +0.5, -z forward, y flipped, unnormalised) and the target gather.
    Returns dict(index [N,2], rays_o, rays_d, gt [N,4], noises [N], bg [N,3])."""
    V, H, W, C = images.shape
    fx, fy, cx, cy = [np.float32(v) for v in intrinsics]
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    n = np.arange(N, dtype=np.uint32)
    ctr = np.stack([n, np.full(N, draw, np.uint32), np.zeros(N, np.uint32), np.zeros(N, np.uint32)], 1)
    r = philox4x32_10(ctr, key).astype(np.uint64)
    view = ((r[:, 0] * np.uint64(V)) >> np.uint64(32)).astype(np.int64)
    pix = ((r[:, 1] * np.uint64(H * W)) >> np.uint64(32)).astype(np.int64)
    j, i = pix // W, pix % W
    f32 = np.float32
    dx = (i.astype(f32) + f32(0.5) - cx) / fx
    dy = -((j.astype(f32) + f32(0.5) - cy) / fy)
    dz = np.full(N, -1.0, f32)
    P = poses.astype(f32)[view]
    rays_d = np.stack([(dx * P[:, k, 0] + dy * P[:, k, 1]) + dz * P[:, k, 2] for k in range(3)], 1).astype(f32)
    rays_o = P[:, :3, 3].copy()
    px = images.reshape(V, H * W, C)[view, pix].astype(f32) / f32(255)
    gt = np.concatenate([px[:, :3], px[:, 3:4] if C == 4 else np.ones((N, 1), f32)], 1).astype(f32)
    u01 = lambda x: (x >> np.uint64(8)).astype(f32) * f32(2.0 ** -24)
    ctr[:, 2] = 1
    q = philox4x32_10(ctr, key).astype(np.uint64)
    return {"index": np.stack([view, pix], 1).astype(np.int32), "rays_o": rays_o, "rays_d": rays_d, "gt": gt,
            "noises": u01(r[:, 2]), "bg": np.stack([u01(q[:, k]) for k in range(3)], 1)}


def schedule(step_done, lr0, decay_steps, beta1, beta2):
    """ngp_x_schedule_step: (lr, 1 - beta1^t, 1/sqrt(1 - beta2^t)) for the step after `step_done` finished ones."""
    t = step_done + 1
    return (np.float32(lr0 * 0.1 ** min(step_done / decay_steps, 1.0)), np.float32(1 - beta1 ** t),
            np.float32(1 / np.sqrt(1 - beta2 ** t)))


# ----------------------------------------------------------------------------- density-grid refresh (engine extension)
def _compact_bits(v):
    v = v.astype(np.uint32) & np.uint32(0x49249249)
    v = (v | (v >> np.uint32(2))) & np.uint32(0xc30c30c3)
    v = (v | (v >> np.uint32(4))) & np.uint32(0x0f00f00f)
    v = (v | (v >> np.uint32(8))) & np.uint32(0xff0000ff)
    v = (v | (v >> np.uint32(16))) & np.uint32(0x0000ffff)
    return v


def density_grid_sample(grid_cas, H, span, half, n_uniform, n_occupied, full, seed, draw, binned=False):
    """ngp_x_density_grid_sample: the cell draws of update_extra_state (nerf/renderer.py:851-872) with Philox in
    place of torch's generator.  Returns (indices int32 [n], xyzs f32 [n,3]).
    binned: the draws as the device delivers them when it can (power-of-two H, >= 4096 cells, not a full sweep) -- generated
    bin by bin, in Morton order (density_grid_sample_binned below); otherwise independent draws in draw order."""
    if binned and not full:
        return density_grid_sample_binned(grid_cas, H, span, half, n_uniform, n_occupied, seed, draw)
    n = n_uniform + n_occupied
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    i = np.arange(n, dtype=np.uint32)
    ctr = np.stack([i, np.full(n, draw, np.uint32), np.full(n, 2, np.uint32), np.zeros(n, np.uint32)], 1)
    r = philox4x32_10(ctr, key).astype(np.uint64)
    ctr[:, 2] = 3
    q = philox4x32_10(ctr, key).astype(np.uint64)
    index = np.empty(n, np.int64)
    if full:
        index[:n_uniform] = np.arange(n_uniform)
    else:
        c = [((r[:n_uniform, k] * np.uint64(H)) >> np.uint64(32)).astype(np.uint32) for k in range(3)]
        index[:n_uniform] = morton3D(np.stack(c, 1).astype(np.int32))
    if n_occupied:
        pos = np.flatnonzero(np.asarray(grid_cas).reshape(-1) > 0)          # ascending cell order
        if len(pos):
            pick = ((r[n_uniform:, 0] * np.uint64(len(pos))) >> np.uint64(32)).astype(np.int64)
            index[n_uniform:] = pos[pick]
        else:
            index[n_uniform:] = -1
    live = index >= 0
    cell = np.where(live, index, 0).astype(np.uint32)
    f32 = np.float32
    u01 = lambda x: (x >> np.uint64(8)).astype(f32) * f32(2.0 ** -24)
    xyz = np.zeros((n, 3), f32)
    for k in range(3):
        c = _compact_bits(cell >> np.uint32(k)).astype(f32)
        xyz[:, k] = ((f32(2) * c) / f32(H - 1) - f32(1)) * f32(span) + (u01(q[:, k]) * f32(2) - f32(1)) * f32(half)
    xyz[~live] = 0
    return index.astype(np.int32), xyz


def density_grid_sample_binned(grid_cas, H, span, half, n_uniform, n_occupied, seed, draw):
    """The same distribution, generated in Morton order (density_grid.hip, "draws that are born in Morton order"): n
    independent uniform draws are a multinomial count per bin and, inside every bin, that many independent uniform draws
    from the bin.  4096 bins per half.  (1) counts: histogram of the keys of the stream (i, draw, 2) -- uniform half: top
    12 bits of the Morton index of cell (floor(r0 H / 2^32), floor(r1 H / 2^32), floor(r2 H / 2^32)); occupied half: top 12
    bits of r0.  (2) output slot j lies in the bin whose [base, next base) holds it (base = exclusive running sum of the
    counts, uniform bins first); a fresh number f = first word of the stream (j, draw, 4): uniform half: cell = bin <<
    shift | f >> (32 - shift); occupied half: u = bin << 20 | f >> 12, pick = floor(u n_pos / 2^32) among the occupied
    cells in Morton order.  Jitter from the stream (j, draw, 5)."""
    n = n_uniform + n_occupied
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    cells = H ** 3
    bits = int(np.ceil(np.log2(cells)))
    assert H & (H - 1) == 0 and bits >= 12
    shift = bits - 12
    i = np.arange(n, dtype=np.uint32)
    ctr = np.stack([i, np.full(n, draw, np.uint32), np.full(n, 2, np.uint32), np.zeros(n, np.uint32)], 1)
    r = philox4x32_10(ctr, key).astype(np.uint64)
    c = [((r[:n_uniform, k] * np.uint64(H)) >> np.uint64(32)).astype(np.uint32) for k in range(3)]
    key_u = morton3D(np.stack(c, 1).astype(np.int32)).astype(np.int64) >> shift
    key_o = (r[n_uniform:, 0] >> np.uint64(20)).astype(np.int64)
    counts = np.concatenate([np.bincount(key_u, minlength=4096), np.bincount(key_o, minlength=4096)])
    base = np.concatenate([[0], np.cumsum(counts)])                          # base[4096] == n_uniform
    bins = np.repeat(np.arange(8192), counts)                                # bin of every output slot
    ctr[:, 2] = 4
    f = philox4x32_10(ctr, key).astype(np.uint64)[:, 0]
    ctr[:, 2] = 5
    q = philox4x32_10(ctr, key).astype(np.uint64)
    index = np.empty(n, np.int64)
    bu = bins[:n_uniform].astype(np.uint64)
    index[:n_uniform] = (bu << np.uint64(shift)) | (f[:n_uniform] >> np.uint64(32 - shift) if shift else 0)
    if n_occupied:
        pos = np.flatnonzero(np.asarray(grid_cas).reshape(-1) > 0)
        if len(pos):
            u = ((bins[n_uniform:] - 4096).astype(np.uint64) << np.uint64(20)) | (f[n_uniform:] >> np.uint64(12))
            index[n_uniform:] = pos[((u * np.uint64(len(pos))) >> np.uint64(32)).astype(np.int64)]
        else:
            index[n_uniform:] = -1
    live = index >= 0
    cell = np.where(live, index, 0).astype(np.uint32)
    f32 = np.float32
    u01 = lambda x: (x >> np.uint64(8)).astype(f32) * f32(2.0 ** -24)
    xyz = np.zeros((n, 3), f32)
    for k in range(3):
        cc = _compact_bits(cell >> np.uint32(k)).astype(f32)
        xyz[:, k] = ((f32(2) * cc) / f32(H - 1) - f32(1)) * f32(span) + (u01(q[:, k]) * f32(2) - f32(1)) * f32(half)
    xyz[~live] = 0
    assert base[4096] == n_uniform
    return index.astype(np.int32), xyz


def density_grid_update(grid, tmp, decay):
    """renderer.py:884-887: grid = max(grid*decay, tmp) where both >= 0; returns (new grid, mean of clamp(grid, 0))."""
    g, t = np.asarray(grid, np.float32).copy(), np.asarray(tmp, np.float32)
    valid = (g >= 0) & (t >= 0)
    g[valid] = np.maximum(g[valid] * np.float32(decay), t[valid])
    return g, float(np.clip(g, 0, None).astype(np.float64).mean())


# ------------------------------------------------------------------ pose-refinement / HDR side of the step (numpy)
def hdr_loss(pred_rgb, gt_rgb, exposure, lossmult=None, loss_weight=None):
    """The HDR ("RawNeRF") loss of nerf/train_utils.py:512-536 and its gradient with respect to the prediction:
        clip = min(1, pred * exposure[:, None]);  scaling = 1 / (1e-3 + stop_gradient(clip))
        loss = sum((clip - gt)^2 * scaling^2 * lossmult * loss_weight) / sum(lossmult)
    pred_rgb, gt_rgb [N,3]; exposure [N]; lossmult / loss_weight: None (1.0), scalar or [N,3].  float32 throughout like
    the torch code.  Returns (loss, d loss / d pred [N,3])."""
    f32 = np.float32
    pred = np.asarray(pred_rgb, f32)
    gt = np.asarray(gt_rgb, f32)
    ex = np.asarray(exposure, f32)[:, None]
    mult = np.broadcast_to(np.asarray(1.0 if lossmult is None else lossmult, f32), gt.shape).astype(f32)
    lw = np.broadcast_to(np.asarray(1.0 if loss_weight is None else loss_weight, f32), gt.shape).astype(f32)
    scaled = pred * ex
    clip = np.minimum(f32(1.0), scaled)
    scaling = f32(1.0) / (f32(1e-3) + clip)
    data = (clip - gt) ** 2 * scaling ** 2
    norm = mult.sum(dtype=np.float64)
    loss = float((data * mult * lw).sum(dtype=np.float64) / norm)
    grad = np.where(scaled < 1.0, 2.0 * (clip - gt) * scaling ** 2 * ex, 0.0) * mult * lw / f32(norm)
    return loss, grad.astype(f32)


def barf_window(step, iters, start, end, L=16):
    """Level weights of the BARF window at a training step, with the reference's types: annealing =
    np.clip(step / iters, 0, 1).astype(np.float16) (train_utils.py:488), alpha evaluated with that numpy float16 scalar
    and python floats (network.py:100-105: every operation rounds to half precision), then the float32 cosine ramp of
    :106 and `weights[0:2] = 1` (:108).  Returns (weights per level [L] float32, annealing as np.float16)."""
    annealing = np.clip(step / iters, 0, 1).astype(np.float16)
    if end == 0:
        # the guard `end = 1e-12` underflows to 0 in float16 (0/0 at step 0); in float64 -- NumPy 1.x promotes
        # float16-scalar op python-float to float64 -- it opens every level from step 1 on, which is what it is for
        alpha = np.float32((float(annealing) - start) / (1e-12 - start) * L)
    else:
        alpha = (annealing - np.float16(start)) / np.float16(end - start) * np.float16(L)
        assert alpha.dtype == np.float16
    k = np.arange(L, dtype=np.float32)
    w = (1 - np.cos(np.clip(np.float32(alpha) - k, 0, 1).astype(np.float32) * np.float32(np.pi))) / 2
    w = w.astype(np.float32)
    w[0] = 1.0
    return w, annealing


def grid_input_backward(grad, dy_dx, B, D, C, L):
    """gridencoder.cu:352-378: grad_inputs[b, d] = sum_{l, ch} grad[l, b, ch] * dy_dx[b, l, d, ch] (double accumulation)."""
    g = np.asarray(grad, np.float64).reshape(L, B, C)
    j = np.asarray(dy_dx, np.float64).reshape(B, L, D, C)
    return np.einsum("lbc,bldc->bd", g, j)


def pose_gradient(index, grad_rays_o, grad_rays_d, V, W, intrinsics):
    """Adjoint of get_rays (train_utils.py:150-160) per camera: rays_o = P[:3,3], rays_d = P[:3,:3] @ dir_cam with
    dir_cam = ((i + .5 - cx) / fx, -(j + .5 - cy) / fy, -1).  index [N,2] = (view, pixel).  Returns [V,3,4] float64."""
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    out = np.zeros((V, 3, 4))
    pix = index[:, 1].astype(np.int64)
    jj, ii = pix // W, pix % W
    dc = np.stack([(ii + 0.5 - cx) / fx, -((jj + 0.5 - cy) / fy), -np.ones(len(pix))], 1)
    for n in range(index.shape[0]):
        v = int(index[n, 0])
        out[v, :, :3] += np.outer(grad_rays_d[n].astype(np.float64), dc[n])
        out[v, :, 3] += grad_rays_o[n]
    return out
