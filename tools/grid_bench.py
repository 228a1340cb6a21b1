#!/usr/bin/env python3
"""Standalone driver of the hash-grid forward (slab) and the binned table backward on ray-ordered samples, for
rocprofv3 --pmc passes and quick timing on a fixed synthetic sample set (bench.py itself runs under --pmc only with
--no-graph: the profiler's queue-intercept callback faults on the batched AQL submissions of hipGraph launches --
profiles/r02_pmc_graph_crash_stack.txt; tools/pmc_bench.sh measures the bench's own steady state that way).  Samples: 4096 rays x ~50 steps of dt = 2*sqrt(3)/1024 through a shell
around a sphere, like a trained occupancy grid produces."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd import _lib  # noqa: E402
from raw_ngp_amd._lib import engine_backend as eb, gridencoder_backend as gb  # noqa: E402
from raw_ngp_amd.gridencoder.grid import level_table  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--per-ray", type=int, default=50)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fused-adam", action="store_true", help="apply with Adam inside the reduce kernel (single GPU step)")
    ap.add_argument("--zero-tail", action="store_true",
                    help="the last third of every ray carries a zero gradient (samples behind the compositor's early stop): "
                         "with --per-ray 34 this is the bench's steady state (139 k samples, a third of them without records)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    N, K = args.rays, args.per_ray
    o = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 3.0
    target = (torch.rand(N, 3, device=dev, generator=g) - 0.5) * 0.8
    d = torch.nn.functional.normalize(target - o, dim=-1)
    t0 = (target - o).norm(dim=-1, keepdim=True) - 0.09
    dt = 2 * 3 ** 0.5 / 1024
    t = t0 + dt * torch.arange(K, device=dev).float()[None]
    xyz = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3).clamp(-0.999, 0.999).contiguous()
    B = xyz.shape[0]
    scale = float(np.exp2(np.log2(2048 / 16) / 15))
    offsets_np = level_table(3, 16, scale, 16, 19)
    offsets = torch.from_numpy(offsets_np).to(dev)
    L, H, S = 16, 16, float(np.log2(scale))
    rows = int(offsets_np[-1])
    table = (torch.rand(rows, 2, device=dev, generator=g) - 0.5) * 2e-4
    enc = torch.empty(L, B, 2, device=dev)
    x01 = torch.empty(B, 3, device=dev)
    denc = torch.randn(L, B, 2, device=dev, generator=g)
    if args.zero_tail:
        denc.view(L, N, K, 2)[:, :, (2 * K) // 3:] = 0.0
    grad = torch.zeros(rows, 2, device=dev)
    ws = torch.empty(gb.backward_workspace_bytes(B, L, rows), dtype=torch.uint8, device=dev)
    cnt = torch.tensor([B, B, 0, 0], dtype=torch.int32, device=dev)

    t_m, t_v = torch.zeros_like(table), torch.zeros_like(table)
    hyper = torch.tensor([1e-6, 0.1, 31.6, 0.0], device=dev)       # tiny lr: the table stays where it is
    adam = (table, t_m, t_v, hyper, 0.9, 0.999, 1e-15) if args.fused_adam else None

    def once():
        gb.grid_backward_binned_prepare(None, 0.0, offsets, rows, cnt, B, L, L, S, H, ws, merge_max_res=414, stage=1)
        eb.grid_encode_forward_slab(xyz, 1.0, table, offsets, enc, x01, cnt, B, B, L, L, S, H, binned_workspace=ws)
        gb.grid_backward_binned_prepare(None, 0.0, offsets, rows, cnt, B, L, L, S, H, ws, stage=2,
                                        single_segment=args.fused_adam)
        gb.grid_backward_binned_apply(denc, x01, offsets, grad, cnt, B, B, L, L, S, H, ws, adam=adam)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.iters):
        once()
    torch.cuda.synchronize()
    print(f"{B} samples, {(time.perf_counter() - t1) / args.iters * 1e6:.1f} us per forward + binned backward")

    # the encoder's forward alone, three ways (HIP events around each launch)
    x01c = ((xyz + 1.0) * 0.5).contiguous()
    variants = {
        "slab + count (engine)": lambda: eb.grid_encode_forward_slab(xyz, 1.0, table, offsets, enc, x01, cnt, B, B, L, L, S,
                                                                      H, binned_workspace=ws),
        "slab": lambda: eb.grid_encode_forward_slab(xyz, 1.0, table, offsets, enc, x01, cnt, B, B, L, L, S, H),
        "ngp_grid_encode_forward (reference API)": lambda: gb.grid_encode_forward(x01c, table, offsets, enc, B, 3, 2, L, L,
                                                                                    S, H, None, 0, False, 0),
    }
    for name, fn in variants.items():
        times = []
        for i in range(args.iters + 3):
            if "count" in name:
                gb.grid_backward_binned_prepare(None, 0.0, offsets, rows, cnt, B, L, L, S, H, ws, merge_max_res=414, stage=1)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            if i >= 3:
                times.append(a.elapsed_time(b) * 1e3)
        us = float(np.median(times))
        print(f"forward {name}: {us:.1f} us, {B * 1164 / us / 1e6:.2f} TB/s algorithmic ({B * 1164 / us / 1e6 / 8:.3f} of 8 TB/s)")


if __name__ == "__main__":
    main()
