"""Procedural "Lego-style" scene: the synthetic stand-in for NeRF-synthetic Lego (no datasets can be
fetched).  A union of opaque axis-aligned bricks with studs inside [-0.8, 0.8]^3, rendered exactly
(ray / box slab tests) into RGBA views with the NeRF-synthetic camera model:
`camera_angle_x` = 0.6911 rad, cameras on the upper hemisphere at radius 4.03 * scale looking at the
origin, cam2world matrices in the Blender convention that get_rays expects (x right, y up, -z forward).
`SyntheticDataset` exposes what the reference's NeRFDataset exposes to the trainer and renderer:
`poses [V,4,4]`, `intrinsics (fx,fy,cx,cy)`, `images [V,H,W,4] uint8`, `H`, `W` (nerf/provider.py:224-254),
and `sample_rays` = its random-image-batch collate (provider.py:281-324) done on the device.
"""
import math

import numpy as np
import torch

from .utils import get_rays

CAMERA_ANGLE_X = 0.6911112070083618   # NeRF-synthetic transforms_*.json


def make_bricks(seed=0, n_bricks=9):
    """Returns boxes [K, 6] (min xyz, max xyz) and albedo [K, 3]: stacked plates + studs."""
    rng = np.random.default_rng(seed)
    boxes, cols = [], []
    palette = np.array([[0.85, 0.12, 0.10], [0.95, 0.78, 0.10], [0.10, 0.35, 0.80], [0.15, 0.60, 0.25],
                        [0.90, 0.90, 0.88], [0.25, 0.25, 0.28]], dtype=np.float32)
    stud = 0.05
    z0 = -0.3
    boxes.append([-0.75, -0.75, z0 - 0.06, 0.75, 0.75, z0])            # base plate
    cols.append(palette[5])
    for k in range(n_bricks):
        w, d = rng.choice([0.2, 0.3, 0.4, 0.6], 2)
        h = rng.choice([0.12, 0.24, 0.36])
        cx, cy = rng.uniform(-0.55, 0.55, 2)
        lift = rng.choice([0.0, 0.12, 0.24, 0.36])
        lo = np.array([cx - w / 2, cy - d / 2, z0 + lift])
        hi = np.array([cx + w / 2, cy + d / 2, z0 + lift + h])
        lo[:2] = np.clip(lo[:2], -0.72, 0.72)
        hi[:2] = np.clip(hi[:2], -0.72, 0.72)
        col = palette[k % 5]
        boxes.append([*lo, *hi])
        cols.append(col)
        nx, ny = max(int((hi[0] - lo[0]) / 0.1), 1), max(int((hi[1] - lo[1]) / 0.1), 1)
        for ix in range(nx):
            for iy in range(ny):
                sx = lo[0] + (ix + 0.5) * (hi[0] - lo[0]) / nx
                sy = lo[1] + (iy + 0.5) * (hi[1] - lo[1]) / ny
                boxes.append([sx - stud / 2, sy - stud / 2, hi[2], sx + stud / 2, sy + stud / 2, hi[2] + 0.03])
                cols.append(col * 0.92)
    return np.asarray(boxes, dtype=np.float32), np.asarray(cols, dtype=np.float32)


def hemisphere_poses(n, radius, seed, min_elev=0.15, max_elev=1.25):
    """cam2world [n,4,4]: camera at `radius` on the upper hemisphere (z up), looking at the origin."""
    rng = np.random.default_rng(seed)
    az = rng.uniform(0, 2 * math.pi, n)
    el = rng.uniform(min_elev, max_elev, n)
    pos = radius * np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)
    fwd = -pos / np.linalg.norm(pos, axis=1, keepdims=True)          # viewing direction
    up = np.array([0, 0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right, axis=1, keepdims=True)
    true_up = np.cross(right, fwd)
    poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    poses[:, :3, 0], poses[:, :3, 1], poses[:, :3, 2], poses[:, :3, 3] = right, true_up, -fwd, pos
    return poses.astype(np.float32)


@torch.no_grad()
def render_boxes(rays_o, rays_d, boxes, albedo, chunk=1 << 18):
    """Exact first-hit rendering: returns rgba [N,4] in [0,1] (straight alpha) and hit distance [N]."""
    out = torch.zeros(rays_o.shape[0], 4, device=rays_o.device)
    depth = torch.full((rays_o.shape[0],), float("inf"), device=rays_o.device)
    light = torch.nn.functional.normalize(torch.tensor([0.4, -0.3, 0.85], device=rays_o.device), dim=0)
    for s in range(0, rays_o.shape[0], chunk):
        o, d = rays_o[s:s + chunk, None, :], rays_d[s:s + chunk, None, :]
        inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
        t0 = (boxes[None, :, :3] - o) * inv
        t1 = (boxes[None, :, 3:] - o) * inv
        tmin, tmax = torch.minimum(t0, t1), torch.maximum(t0, t1)
        tn, axis = tmin.max(-1)
        tf = tmax.min(-1).values
        hit = (tn <= tf) & (tf > 0) & (tn > 0)
        tn = torch.where(hit, tn, torch.full_like(tn, float("inf")))
        t, k = tn.min(-1)
        any_hit = torch.isfinite(t)
        ax = torch.gather(axis, 1, k[:, None]).squeeze(1)
        dn = torch.gather(d.squeeze(1), 1, ax[:, None]).squeeze(1)
        normal = torch.zeros(o.shape[0], 3, device=o.device)
        normal.scatter_(1, ax[:, None], -torch.sign(dn)[:, None])
        vd = torch.nn.functional.normalize(d.squeeze(1), dim=-1)
        diff = (normal * light).sum(-1).clamp(min=0)
        refl = light - 2 * (light * normal).sum(-1, keepdim=True) * normal
        spec = (refl * vd).sum(-1).clamp(min=0) ** 8
        rgb = albedo[k] * (0.45 + 0.55 * diff)[:, None] + 0.2 * spec[:, None]
        out[s:s + chunk, :3] = torch.where(any_hit[:, None], rgb.clamp(0, 1), torch.zeros_like(rgb))
        out[s:s + chunk, 3] = any_hit.float()
        depth[s:s + chunk] = t
    return out, depth


class SyntheticDataset:
    """In-memory dataset with the attributes the reference's trainer / renderer read."""

    def __init__(self, opt, device, ttype="train", n_views=100, H=800, W=800, scale=0.8, seed=0):
        self.opt, self.device, self.training = opt, device, ttype == "train"
        self.H, self.W = H, W
        fl = 0.5 * W / math.tan(0.5 * CAMERA_ANGLE_X)
        self.intrinsics = np.array([fl, fl, W / 2.0, H / 2.0])
        boxes, albedo = make_bricks(seed=0)
        self.boxes, self.albedo = torch.from_numpy(boxes).to(device), torch.from_numpy(albedo).to(device)
        poses = hemisphere_poses(n_views, 4.03 * scale, seed=seed + (0 if self.training else 1000))
        self.poses = torch.from_numpy(poses).to(device)
        self.radius = float(self.poses[:, :3, 3].norm(dim=-1).mean())
        imgs = []
        for v in range(n_views):
            r = get_rays(self.poses[v:v + 1], self.intrinsics, H, W, -1)
            rgba, _ = render_boxes(r["rays_o"].contiguous(), r["rays_d"].contiguous(), self.boxes, self.albedo)
            imgs.append((rgba.view(H, W, 4) * 255 + 0.5).to(torch.uint8))
        self.images = torch.stack(imgs, 0)        # [V, H, W, 4] uint8, straight alpha

    def __len__(self):
        return self.poses.shape[0]

    def sample_rays(self, num_rays, generator=None, pose_fn=None):
        """random_image_batch collate: every ray picks its own (view, pixel).  pose_fn(poses, index): the pose
        optimiser's hook (provider.py:298-300); `self.ldirs` [V,3], when set, yields per-ray light directions
        (colmap_provider.py:619-620)."""
        V = self.poses.shape[0]
        index = torch.randint(0, V, size=(num_rays,), device=self.device, generator=generator)
        poses = self.poses[index]
        if pose_fn is not None:
            poses = pose_fn(poses, index)
        ldirs = getattr(self, "ldirs", None)
        rays = get_rays(poses, self.intrinsics, self.H, self.W, num_rays, generator=generator,
                        ldirs=ldirs[index] if ldirs is not None else None)
        images = self.images[index, rays["j"], rays["i"]].float() / 255
        out = {"rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "images": images, "index": index,
               "H": self.H, "W": self.W}
        if ldirs is not None:
            out["rays_ldir"] = rays["rays_ldir"]
        return out

    def view(self, v):
        ldirs = getattr(self, "ldirs", None)
        rays = get_rays(self.poses[v:v + 1], self.intrinsics, self.H, self.W, -1)
        out = {"rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "images": self.images[v].float() / 255,
               "H": self.H, "W": self.W}
        if ldirs is not None:           # one light per image: the inference march repeats it per sample
            out["rays_ldir"] = ldirs[v:v + 1]
        return out

    def occupancy_grid(self, grid_size=128, bound=1.0):
        """Exact brick occupancy on the density grid's cell centres, [grid_size^3] in x-major (x,y,z) order."""
        g = (torch.arange(grid_size, device=self.device, dtype=torch.float32) + 0.5) / grid_size * 2 * bound - bound
        X, Y, Z = torch.meshgrid(g, g, g, indexing="ij")
        p = torch.stack([X, Y, Z], -1).view(-1, 1, 3)
        half = bound / grid_size
        occ = torch.zeros(p.shape[0], dtype=torch.bool, device=self.device)
        for s in range(0, p.shape[0], 1 << 18):
            q = p[s:s + (1 << 18)]
            inside = ((q + half >= self.boxes[None, :, :3]) & (q - half <= self.boxes[None, :, 3:])).all(-1).any(-1)
            occ[s:s + (1 << 18)] = inside
        return occ.view(grid_size, grid_size, grid_size)
