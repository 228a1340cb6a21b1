#!/usr/bin/env python3
"""Scratch: run the level-split group path eagerly (no capture) to locate a failing launch."""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raw_ngp_amd.nerf.engine import FusedTrainer  # noqa: E402
from raw_ngp_amd.nerf.network import NeRFNetwork  # noqa: E402
from raw_ngp_amd.nerf.options import Options  # noqa: E402
from raw_ngp_amd.nerf.scene import SyntheticDataset  # noqa: E402


class NullGraph:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def main():
    torch.manual_seed(0)
    opt = Options(bound=1.0, num_rays=1024, iters=200, fused_mlp=True)
    data = SyntheticDataset(opt, torch.device("cuda"), "train", n_views=6, H=64, W=64)
    eng = FusedTrainer(opt, NeRFNetwork(opt).cuda(), data, device="cuda", capacity=1024 * 256)
    print("split", eng._level_split, flush=True)
    for _ in range(3):
        eng.train_step()
    torch.cuda.synchronize()
    import raw_ngp_amd._lib as L
    real = L._call

    def loud(name, anchor, *args, **kw):
        try:
            real(name, anchor, *args, **kw)
            torch.cuda.synchronize()
        except Exception:
            print("FAILED in", name, [a for a in args if isinstance(a, int) and a < 10**7], flush=True)
            raise
    L._call = loud
    torch.cuda.graph = NullGraph
    try:
        eng._capture_group(eng.global_step % 2, 4, True, False)
        print("eager group ok", flush=True)
    except Exception:
        traceback.print_exc()


if __name__ == "__main__":
    main()
