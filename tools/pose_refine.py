#!/usr/bin/env python3
"""Config-4-style run of the hot path (SURVEY 8d "Config 4", 8f row 3) on the procedural scene: light-direction
conditioning (rfield: second SH call, 47 -> 80 -> 80 -> 3 view MLP), se(3) pose refinement from perturbed cameras with
BARF / BAA-NGP level windows.  Rays carry gradients, so the encoders' input Jacobians (dy_dx), the SH backward and the
marcher's segmented ray-gradient sum are all live.  Prints the pose error before / after and the step rate.

    python tools/pose_refine.py --iters 3000 --noise 0.03
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd.nerf.options import Options  # noqa: E402
from raw_ngp_amd.nerf.network import NeRFNetwork  # noqa: E402
from raw_ngp_amd.nerf.scene import SyntheticDataset  # noqa: E402
from raw_ngp_amd.nerf.trainer import Trainer  # noqa: E402
from raw_ngp_amd.nerf import pose as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--noise", type=float, default=0.03)
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--res", type=int, default=200)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--pose-opt", default="barf", choices=["barf", "baangp", "none"])
    ap.add_argument("--no-rfield", action="store_true")
    ap.add_argument("--c-lr", type=float, default=1e-3)
    ap.add_argument("--log-every", type=int, default=500)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--arena", type=int, default=0, help="sample-arena rows (0 = the reference's two-call march)")
    ap.add_argument("--per-op", action="store_true",
                    help="the per-op autograd path (Trainer: torch MLPs, the reference's call sequence) instead of the fused step")
    ap.add_argument("--hdr", action="store_true", help="HDR loss with per-view exposures (train_utils.py:512-536)")
    ap.add_argument("--timed-steps", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(args.seed)
    opt = Options(bound=1.0, num_rays=args.rays, iters=args.iters, rfield=not args.no_rfield, pose_opt=args.pose_opt,
                  noise=args.noise, c_lr=args.c_lr, arena_capacity=args.arena, image_mode="HDR" if args.hdr else "LDR")
    data = SyntheticDataset(opt, dev, "train", n_views=args.views, H=args.res, W=args.res)
    if opt.rfield:
        data.ldirs = torch.from_numpy(P.synthetic_light_dirs(args.views)).to(dev)
    if args.hdr:
        import numpy as np
        data.exposures = torch.from_numpy(np.random.default_rng(5).choice([0.5, 1.0, 2.0], args.views).astype("float32")).to(dev)
        # what a camera with that exposure records of the same radiance: colour x exposure, clipped at white
        rgb = data.images[..., :3].float() * data.exposures.view(-1, 1, 1, 1)
        data.images[..., :3] = rgb.clamp(max=255).to(torch.uint8)
    model = NeRFNetwork(opt)
    if args.per_op:
        assert not args.hdr, "the HDR loss is implemented in the fused step"
        trainer = Trainer(opt, model, data, dev, seed=args.seed)
    else:
        from raw_ngp_amd.nerf.engine import FusedTrainer
        trainer = FusedTrainer(opt, model, data, device=dev, seed=args.seed)
    co = trainer.pose_optimizer
    report = {"config": vars(args), "step": "per-op" if args.per_op else "fused"}
    if co is not None:
        report["pose_error_start"] = P.pose_error(co.get_refined_poses(data.poses), data.poses)
    t0, last = time.time(), 0
    for it in range(args.iters):
        trainer.train_step()
        if args.log_every and (it + 1) % args.log_every == 0:
            torch.cuda.synchronize()
            dt = time.time() - t0
            err = P.pose_error(co.get_refined_poses(data.poses), data.poses) if co is not None else None
            print(f"[{it + 1}] loss {float(trainer.last_loss):.5f} samples {trainer.last_num_points} "
                  f"{(it + 1 - last) / dt:.0f} steps/s pose error (deg, dist) {err}", flush=True)
            t0, last = time.time(), it + 1
    # steady-state step rate
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.timed_steps):
        trainer.train_step()
    torch.cuda.synchronize()
    ms = (time.time() - t0) / args.timed_steps * 1e3
    report.update(ms_per_step=round(ms, 3), rays_per_s=round(args.rays / ms * 1e3),
                  samples_per_step=int(trainer.last_num_points))
    if co is not None:
        report["pose_error_end"] = P.pose_error(co.get_refined_poses(data.poses), data.poses)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
