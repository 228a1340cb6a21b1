"""Ray generation and metrics used by the harness.

`get_rays` follows the reference's nerf/train_utils.py:96-172 (pixel centre +0.5, camera looks
down -z, y flipped, directions NOT normalised); `PSNRMeter` follows :203-248."""
import math
import os
import random

import numpy as np
import torch


def seed_everything(seed):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


@torch.autocast("cuda", enabled=False)
def get_rays(poses, intrinsics, H, W, N=-1, coords=None, ldirs=None, generator=None):
    """poses [B,4,4] cam2world (B = 1, or B = N with one pose per ray), intrinsics (fx, fy, cx, cy).
    N > 0 draws N random pixels (with replacement, like torch.randint in the reference)."""
    device = poses.device
    if isinstance(intrinsics, np.ndarray) or (torch.is_tensor(intrinsics) and intrinsics.dim() == 1):
        fx, fy, cx, cy = [float(v) for v in intrinsics]
    else:
        fx, fy, cx, cy = intrinsics[:, 0], intrinsics[:, 1], intrinsics[:, 2], intrinsics[:, 3]
    results = {}
    if N > 0:
        if coords is not None:
            inds = coords[:, 0] * W + coords[:, 1]
        else:
            inds = torch.randint(0, H * W, size=[N], device=device, generator=generator)
        ii = (inds % W)
        jj = torch.div(inds, W, rounding_mode="floor")
        results["i"], results["j"] = ii.long(), jj.long()
        i, j = ii.float() + 0.5, jj.float() + 0.5
    else:
        j, i = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                              torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
        i, j = i.reshape(-1) + 0.5, j.reshape(-1) + 0.5
    directions = torch.stack(((i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)), dim=-1)
    # rays_d = R @ directions per ray; written as a broadcast multiply-sum: the batched [N,1,3] x [N,3,3]
    # matmul the reference uses (train_utils.py:159) dispatches one tiny GEMM per ray on ROCm (0.3 ms)
    rays_d = (directions.unsqueeze(1) * poses[:, :3, :3]).sum(-1)
    results["rays_o"] = poses[:, :3, 3].expand_as(rays_d)
    results["rays_d"] = rays_d
    results["rays_ldir"] = ldirs.expand_as(rays_d) if ldirs is not None else None
    return results


class PSNRMeter:
    """Per-image PSNR averaged over images, max pixel value 1."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.V, self.N = 0.0, 0

    def update(self, preds, truths):
        if torch.is_tensor(preds):
            preds = preds.detach().float().cpu().numpy()
        if torch.is_tensor(truths):
            truths = truths.detach().float().cpu().numpy()
        psnr = -10 * np.log10(np.mean((preds - truths) ** 2))
        self.V += psnr
        self.N += 1
        return psnr

    def measure(self):
        return self.V / max(self.N, 1)

    def report(self):
        return f"PSNR = {self.measure():.6f}"


# ---- HDR loss of the reference's train_step (train_utils.py:512-536) and its target-dependent weights (raw/raw_utils.py:30-53)
def hdr_loss_weight(kind, gt_rgb):
    """[N,3] weight of the squared residuals (a constant: the reference detaches it), or None for "none".
    gaussian: exp(-(v - peak^2) / (2 sigma^2)) with peak 1, sigma 0.5 (no square on the difference, as the reference has it),
    scaled so that the largest weight in the batch is 1.  hanning: a Hann window over the POSITION of the ray in the
    batch, peak 2, the same for the three channels.  planck: cosine taper around 0.5, zero outside 0.5 +- 0.95, peak 2."""
    if kind == "none":
        return None
    if kind == "gaussian":
        w = torch.exp((1.0 - gt_rgb) / (2 * 0.5 ** 2))
        return (w / w.max()).detach()
    if kind == "hanning":
        n = gt_rgb.shape[0]
        w = 0.5 - 0.5 * torch.cos(2 * math.pi * torch.arange(n, device=gt_rgb.device, dtype=torch.float32) / (n - 1))
        return (2.0 * w / w.max())[:, None].expand(-1, 3).detach()
    if kind == "planck":
        inside = (gt_rgb >= 0.5 - 0.95) & (gt_rgb <= 0.5 + 0.95)
        w = 2.0 * (0.5 + 0.5 * torch.cos((gt_rgb - 0.5) * (math.pi / (2 * 0.95))))
        return torch.where(inside, w, torch.zeros_like(w))
    raise ValueError(f"loss_weight {kind!r}")


def hdr_loss(pred_rgb, gt_rgb, exposure, loss_weight="none", lossmult=None):
    """RawNeRF's clipped, tone-curve-weighted squared error: the prediction is scaled by the ray's exposure and clipped at
    white, the residual weighted by the squared gradient of log(1e-3 + x) at the (detached) clipped value; mean over the
    entries lossmult selects (None: all)."""
    clip = torch.clamp(pred_rgb * exposure[:, None], max=1.0)
    data = (clip - gt_rgb) ** 2 / (1e-3 + clip.detach()) ** 2
    mult = torch.ones_like(gt_rgb) if lossmult is None else torch.broadcast_to(lossmult, gt_rgb.shape)
    w = hdr_loss_weight(loss_weight, gt_rgb)
    return (data * mult * (1.0 if w is None else w)).sum() / mult.sum()

