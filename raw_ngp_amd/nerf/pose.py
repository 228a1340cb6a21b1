"""Camera-pose refinement around the hot path.

Host-side mirror of what the reference wraps around its kernels when `--pose_opt barf|baangp` is on:
  * `se3_to_SE3`, `compose`            -- barf/camera.py:47-63 (compose), :91-102 (exponential map);
  * `CameraOptimizer`                  -- barf/camera_optimizers.py:14-52,94-106: one se(3) correction per camera
    (zero-initialised Embedding), optional pre-generated pose noise, its own Adam (lr `c_lr`) with an exponential
    decay to 1 % over `iters`; `forward(poses, indices)` returns the refined [N,3,4] cam2world matrices;
  * `align_cameras` / `pose_error`     -- what barf/pose_analysis.py reports (rotation error in degrees, translation
    error after a similarity alignment of the camera centres), reduced to the two numbers.
Everything here is ordinary torch: the data-parallel work it triggers -- d loss / d rays through the marcher's
segmented ray-gradient sum, the encoders' input Jacobians and the SH backward -- is in the HIP library.
"""
import numpy as np
import torch
from torch import nn


def _series(theta):
    """sin(t)/t, (1 - cos t)/t^2, (t - sin t)/t^3 with their limits at 0 (a few series terms below 1e-2 rad,
    where the closed forms cancel; the reference sums 11 Taylor terms everywhere, camera.py:130-156)."""
    t2 = theta * theta
    small = theta < 1e-2
    ts = torch.where(small, torch.ones_like(theta), theta)
    a = torch.where(small, 1 - t2 / 6 + t2 * t2 / 120, torch.sin(ts) / ts)
    b = torch.where(small, 0.5 - t2 / 24 + t2 * t2 / 720, (1 - torch.cos(ts)) / (ts * ts))
    c = torch.where(small, 1 / 6 - t2 / 120 + t2 * t2 / 5040, (ts - torch.sin(ts)) / (ts * ts * ts))
    return a, b, c


def skew(w):
    w0, w1, w2 = w.unbind(-1)
    o = torch.zeros_like(w0)
    return torch.stack([torch.stack([o, -w2, w1], -1), torch.stack([w2, o, -w0], -1), torch.stack([-w1, w0, o], -1)], -2)


def se3_to_SE3(wu):
    """[..., 6] (rotation vector w, translation generator u) -> [..., 3, 4] = [exp(w^) | V(w) u]."""
    w, u = wu[..., :3], wu[..., 3:]
    wx = skew(w)
    # |w| through a guarded sqrt: the plain norm has no gradient at 0, where every correction starts
    theta = (w * w).sum(-1).clamp_min(1e-24).sqrt()[..., None, None]
    a, b, c = _series(theta)
    eye = torch.eye(3, device=wu.device, dtype=wu.dtype)
    wx2 = wx @ wx
    R = eye + a * wx + b * wx2
    V = eye + b * wx + c * wx2
    return torch.cat([R, V @ u[..., None]], -1)


def compose(poses):
    """pose_new(x) = pose_n o ... o pose_1(x) for [...,3,4] matrices."""
    out = poses[0]
    for nxt in poses[1:]:
        R = nxt[..., :3] @ out[..., :3]
        t = nxt[..., :3] @ out[..., 3:] + nxt[..., 3:]
        out = torch.cat([R, t], -1)
    return out


class CameraOptimizer(nn.Module):
    def __init__(self, num_cameras, device, opt, seed=0):
        super().__init__()
        self.num_cameras, self.device, self.opt = num_cameras, device, opt
        self.annealing = 0.0
        self.pose_noise = None
        noise = float(getattr(opt, "noise", 0.0) or 0.0)
        if noise > 0:
            g = torch.Generator(device="cpu").manual_seed(seed + 77)
            # the reference concatenates [translation-sized, rotation-sized] noise into se3_to_SE3, whose first three
            # components are the ROTATION vector (camera_optimizers.py:27-36): kept as is
            first = torch.randn(num_cameras, 3, generator=g) * noise * float(getattr(opt, "scale", 1.0))
            second = torch.randn(num_cameras, 3, generator=g) * noise
            self.pose_noise = se3_to_SE3(torch.cat([first, second], -1).to(device))
        self.se3_refine = nn.Embedding(num_cameras, 6, device=device)
        nn.init.zeros_(self.se3_refine.weight)
        c_lr = float(getattr(opt, "c_lr", 1e-3))
        self.optimizer = torch.optim.Adam(self.parameters(), lr=c_lr)
        self.lr_scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, 1e-2 ** (1.0 / opt.iters))

    def update_annealing(self, value):
        self.annealing = value

    def forward(self, poses, indices):
        poses = poses[:, :3, :]
        if self.pose_noise is not None:
            poses = compose([self.pose_noise[indices], poses])
        if getattr(self.opt, "identity", False):
            poses = torch.eye(4, device=self.device)[None, :3, :4].expand(poses.shape[0], 3, 4)
        return compose([se3_to_SE3(self.se3_refine.weight[indices]), poses])

    @torch.no_grad()
    def get_refined_poses(self, poses):
        return self(poses, torch.arange(self.num_cameras, device=self.device))


def align_cameras(pred, gt):
    """Similarity transform (Procrustes on the camera centres) taking `pred` [V,3,4] onto `gt`; returns the aligned
    poses.  Joint pose / field optimisation is only defined up to such a transform."""
    c0, c1 = pred[:, :, 3].double(), gt[:, :3, 3].double()
    m0, m1 = c0.mean(0), c1.mean(0)
    x0, x1 = c0 - m0, c1 - m1
    s0, s1 = x0.pow(2).sum(-1).mean().sqrt(), x1.pow(2).sum(-1).mean().sqrt()
    U, _, Vt = torch.linalg.svd((x0 / s0).T @ (x1 / s1))
    R = (U @ Vt).T
    if torch.det(R) < 0:
        U[:, -1] = -U[:, -1]
        R = (U @ Vt).T
    centres = ((x0 / s0) @ R.T) * s1 + m1
    rot = R.float() @ pred[:, :, :3]
    return torch.cat([rot, centres.float()[..., None]], -1)


def pose_error(pred, gt):
    """(mean rotation error in degrees, mean camera-centre distance) of aligned [V,3,4] poses against gt."""
    aligned = align_cameras(pred, gt)
    rel = aligned[:, :, :3].transpose(-1, -2) @ gt[:, :3, :3]
    cos = ((rel.diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2).clamp(-1, 1)
    return float(torch.rad2deg(torch.acos(cos)).mean()), float((aligned[:, :, 3] - gt[:, :3, 3]).norm(dim=-1).mean())


def synthetic_light_dirs(n_views, seed=0):
    """One unit light direction per view on the upper hemisphere (stands in for the light stage's metadata,
    colmap_provider.py:619: `metadict['ldirs'][index]`)."""
    rng = np.random.default_rng(seed + 4242)
    v = rng.normal(size=(n_views, 3))
    v[:, 2] = np.abs(v[:, 2]) + 0.2
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


__all__ = ["se3_to_SE3", "compose", "skew", "CameraOptimizer", "align_cameras", "pose_error", "synthetic_light_dirs"]
