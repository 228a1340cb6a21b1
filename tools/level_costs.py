#!/usr/bin/env python3
"""Cost of one tile of each level of the slab encoder, for the points each of its callers hands it.

Trains the bench scene for a few thousand steps, takes (a) the ray-ordered samples of a training step and (b) the two halves
of a density-grid refresh's cell draws (uniform cells / occupied cells, in the order the refresh leaves them), and times the
slab encoder on ONE level at a time, confined to ONE XCD (NGP_PLACE_ONLY_LEVEL, engine_kernels.hip) -- the time an XCD
needs for the whole level.  Prints the per-level times, the cost vectors (NGP_LEVEL_COST_STEP / NGP_LEVEL_COST_REFRESH take them)
and, for each point set, the whole kernel under {fixed pairing, flat costs, measured costs}.

    python tools/level_costs.py [--steps 3000] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd import _lib  # noqa: E402
from raw_ngp_amd._lib import engine_backend as eb  # noqa: E402
from raw_ngp_amd.nerf.engine import FusedTrainer  # noqa: E402
from raw_ngp_amd.nerf.network import NeRFNetwork  # noqa: E402
from raw_ngp_amd.nerf.options import Options  # noqa: E402
from raw_ngp_amd.nerf.scene import SyntheticDataset  # noqa: E402


def timed(fn, reps):
    """us per call, from a graph of `reps` calls (the host cannot launch them as fast as the short ones run)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (3 * reps)


def spread(fn, reps):
    os.environ["NGP_PLACE_SPREAD"] = "1"
    try:
        return timed(fn, reps)
    finally:
        del os.environ["NGP_PLACE_SPREAD"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--bound", type=float, default=1.0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    _lib.load()
    torch.manual_seed(0)
    opt = Options(bound=args.bound, background="random", num_rays=4096, iters=5000)
    data = SyntheticDataset(opt, dev, "train", n_views=100, H=800, W=800)
    tr = FusedTrainer(opt, NeRFNetwork(opt), data, device=dev, seed=0)
    tr.model.mark_untrained_grid(data)
    tr.train(args.steps - args.steps % 16 + 8)          # stop half way between two refreshes
    torch.cuda.synchronize()
    m, L, cap = tr.model, tr.L, tr.cap
    counts = [int(s.arena.counter[1]) for s in tr.slots]
    slot = tr.slots[max(range(len(counts)), key=lambda i: counts[i])]
    n_step = min(max(counts), cap)
    cells = m.grid_size ** 3
    sets = {
        "step samples": (slot.arena.xyzs[:n_step].clone(), tr.level_cost_step),
        "refresh, uniform half": (tr.dg_xyzs[:cells // 4].clone(), tr.level_cost_refresh),
        "refresh, occupied half": (tr.dg_xyzs[cells // 4:cells // 2].clone(), tr.level_cost_refresh),
    }
    enc = torch.empty(L, cap, 2, device=dev)
    offsets = m.grid_encoder.offsets
    flat = [1.0] * L

    def run(x, cost):
        eb.grid_encode_forward_slab(x, m.bound, tr.table, offsets, enc, None, None, x.shape[0], cap, L, L, tr.S, tr.H,
                                    level_cost=cost)

    for name, (x, current) in sets.items():
        n = x.shape[0]
        per = []
        for l in range(L):
            os.environ["NGP_PLACE_ONLY_LEVEL"] = str(l)
            per.append(timed(lambda: run(x, flat), args.reps))
        del os.environ["NGP_PLACE_ONLY_LEVEL"]
        unit = min(per)
        cost = [round(t / unit, 2) for t in per]
        print(f"\n{name}: {n} points")
        print("  one level on one XCD [us]: " + " ".join(f"{t:.1f}" for t in per))
        print(f"  sum / 8 = {sum(per) / 8:.1f} us   fixed pairing max = {max(per[k] + per[L - 1 - k] for k in range(8)):.1f} us")
        print("  cost = [" + ", ".join(f"{c:g}" for c in cost) + "]")
        print(f"  whole kernel: fixed pairing {timed(lambda: run(x, None), args.reps):.1f} us, flat costs "
              f"{timed(lambda: run(x, flat), args.reps):.1f} us, measured costs {timed(lambda: run(x, cost), args.reps):.1f} us"
              + f", every level on every XCD {spread(lambda: run(x, flat), args.reps):.1f} us"
              + (f", engine's costs {timed(lambda: run(x, current), args.reps):.1f} us" if current else ""))


if __name__ == "__main__":
    main()
