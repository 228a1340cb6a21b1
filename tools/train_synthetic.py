#!/usr/bin/env python3
"""Train the field on the procedural Lego-style scene and report PSNR on held-out views.
Usage: python tools/train_synthetic.py --iters 5000 [--res 800] [--views 100] [--fused-mlp]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd.nerf.network import NeRFNetwork  # noqa: E402
from raw_ngp_amd.nerf.options import Options  # noqa: E402
from raw_ngp_amd.nerf.scene import SyntheticDataset  # noqa: E402
from raw_ngp_amd.nerf.engine import FusedTrainer  # noqa: E402
from raw_ngp_amd.nerf.trainer import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--views", type=int, default=100)
    ap.add_argument("--val-views", type=int, default=4)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--log-every", type=int, default=250)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--fused-mlp", action="store_true")
    ap.add_argument("--arena", type=int, default=0)
    ap.add_argument("--engine", action="store_true", help="fused training step (nerf/engine.py)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--torch-refresh", action="store_true", help="density-grid refresh through torch ops")
    ap.add_argument("--torch-sampler", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--write-scene", default="", help="dump the procedural scene as transforms_*.json + PNG to this directory")
    ap.add_argument("--data", default="", help="train from a Blender-format scene directory instead (scale 1, offset 0)")
    ap.add_argument("--bound", type=float, default=1.0, help="scene bound (2 = the reference's default: two cascades)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    opt = Options(bound=args.bound, num_rays=args.rays, iters=args.iters, fp16=args.fp16, fused_mlp=args.fused_mlp,
                  arena_capacity=args.arena, capture_graph=not args.no_graph, device_sampler=not args.torch_sampler,
                  native_grid_refresh=not args.torch_refresh)
    t0 = time.time()
    data = SyntheticDataset(opt, dev, "train", n_views=args.views, H=args.res, W=args.res)
    val = SyntheticDataset(opt, dev, "val", n_views=args.val_views, H=args.res, W=args.res)
    print(f"scene rendered in {time.time() - t0:.1f}s", flush=True)
    if args.write_scene:
        from raw_ngp_amd.nerf.provider import write_blender_scene
        write_blender_scene(args.write_scene, {"train": data, "val": val})
        print(f"scene written to {args.write_scene}", flush=True)
    if args.data:
        from raw_ngp_amd.nerf.provider import BlenderDataset
        data = BlenderDataset(opt, args.data, "train", device=dev, scale=1.0, offset=(0, 0, 0))
        val = BlenderDataset(opt, args.data, "val", device=dev, scale=1.0, offset=(0, 0, 0))
        print(f"loaded {len(data)} train / {len(val)} val frames of {data.H}x{data.W} from {args.data}", flush=True)
    model = NeRFNetwork(opt)
    trainer = FusedTrainer(opt, model, data, device=dev, seed=args.seed, capacity=args.arena or None) if args.engine \
        else Trainer(opt, model, data, device=dev)
    hist = []
    done = 0
    while done < args.iters:
        n = min(args.log_every, args.iters - done)
        torch.cuda.synchronize()
        t1 = time.time()
        trainer.train(n)
        torch.cuda.synchronize()
        dt = time.time() - t1
        done += n
        row = {"iter": done, "loss": float(trainer.last_loss), "samples": int(trainer.last_num_points),
               "ms_per_step": round(dt / n * 1e3, 3), "mean_density": round(float(getattr(trainer, "mean_density", model.mean_density)), 4)}
        hist.append(row)
        print(json.dumps(row), flush=True)
    psnr = trainer.evaluate(val)
    print(json.dumps({"psnr": round(float(psnr), 3), "iters": args.iters, "val_views": args.val_views,
                      "total_s": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
