#!/usr/bin/env python3
"""Per-kernel timing of the C-ABI entry points on synthetic inputs of SURVEY.md section 8d.

Prints one JSON line per kernel: average launch duration (HIP events on torch's current stream,
which is the stream the shims launch on) and achieved algorithmic bandwidth.
Usage: python tools/kernel_bench.py [--B 262144] [--iters 50] [--only grid_fwd,...]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd import _lib  # noqa: E402
from raw_ngp_amd.gridencoder.grid import level_table  # noqa: E402


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3     # seconds


def report(name, sec, bytes_, extra=None):
    row = {"kernel": name, "us": round(sec * 1e6, 2), "GB/s": round(bytes_ / sec / 1e9, 1),
           "frac_of_8TBps": round(bytes_ / sec / 8e12, 4)}
    if extra:
        row.update(extra)
    print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=2 ** 18)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--coherent", action="store_true", help="ray-coherent sample positions instead of uniform")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    want = lambda k: not only or k in only  # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(0)
    B, D, C, L, H = args.B, 3, 2, 16, 16
    scale = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(scale))
    offsets = torch.from_numpy(level_table(D, L, scale, H, 19)).cuda()
    rows = int(offsets[-1])
    table = (torch.rand(rows, C, device="cuda", generator=g) * 2 - 1) * 1e-4
    if args.coherent:
        n_rays = B // 64
        o = torch.rand(n_rays, 1, 3, device="cuda", generator=g)
        d = torch.randn(n_rays, 1, 3, device="cuda", generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        t = torch.arange(64, device="cuda").view(1, 64, 1) * (2 * 3 ** 0.5 / 1024 / 2)
        x = (o * 0.6 + 0.2 + d * t).clamp(0, 1).reshape(B, 3).contiguous()
    else:
        x = torch.rand(B, D, device="cuda", generator=g)
    out = torch.empty(L, B, C, device="cuda")
    gb = _lib.gridencoder_backend

    if want("grid_fwd"):
        sec = timeit(lambda: gb.grid_encode_forward(x, table, offsets, out, B, D, C, L, L, S, H, None, 0, False, 0), args.iters)
        report("grid_encode_forward", sec, B * (12 + L * (64 + 8)))
    if want("grid_fwd_jac"):
        jac = torch.empty(B, L * D * C, device="cuda")
        sec = timeit(lambda: gb.grid_encode_forward(x, table, offsets, out, B, D, C, L, L, S, H, jac, 0, False, 0), args.iters)
        report("grid_encode_forward+dy_dx", sec, B * (12 + L * (64 + 8) + L * D * C * 4))
    if want("grid_bwd"):
        grad = torch.randn(L, B, C, device="cuda", generator=g)
        gt = torch.zeros(rows, C, device="cuda")
        for binned in (True, False):
            type(gb).use_binned_backward = binned
            sec = timeit(lambda: gb.grid_encode_backward(grad, x, table, offsets, gt, B, D, C, L, L, S, H, None, None, 0, False, 0),
                         max(args.iters // 5, 3), warmup=2)
            report("grid_encode_backward(%s)" % ("binned" if binned else "atomics"), sec, B * (12 + L * (8 + 64)))
        type(gb).use_binned_backward = True
    if want("sh"):
        v = torch.randn(B, 3, device="cuda", generator=g)
        v = v / v.norm(dim=-1, keepdim=True)
        o16 = torch.empty(B, 16, device="cuda")
        sec = timeit(lambda: _lib.shencoder_backend.sh_encode_forward(v, o16, B, 3, 4, None), args.iters)
        report("sh_encode_forward(deg4)", sec, B * 76)
    if want("composite"):
        N = 4096
        cnt = torch.full((N,), B // N, dtype=torch.int32, device="cuda")
        off = (torch.cumsum(cnt, 0) - cnt).int()
        rays = torch.stack([off, cnt], 1).contiguous()
        sig = torch.rand(B, device="cuda", generator=g) * 5
        rgb = torch.rand(B, 3, device="cuda", generator=g)
        ts = torch.rand(B, 2, device="cuda", generator=g) * 0.01 + 0.003
        w = torch.zeros(B, device="cuda")
        ws, dep, img = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 3, device="cuda")
        rb = _lib.raymarching_backend
        sec = timeit(lambda: rb.composite_rays_train_forward(sig, rgb, ts, rays, B, N, 1e-8, w, ws, dep, img), args.iters)
        report("composite_rays_train_forward", sec, B * 28 + N * 28)
        gs, gc = torch.zeros(B, device="cuda"), torch.zeros(B, 3, device="cuda")
        gw, gws, gd, gi = torch.zeros(B, device="cuda"), torch.ones(N, device="cuda"), torch.ones(N, device="cuda"), torch.ones(N, 3, device="cuda")
        sec = timeit(lambda: rb.composite_rays_train_backward(gw, gws, gd, gi, sig, rgb, ts, rays, ws, dep, img, B, N, 1e-8, gs, gc), args.iters)
        report("composite_rays_train_backward", sec, B * 44 + N * 48)
    if want("adam_ref"):
        # torch's own fused Adam over the table, for the budget table in DESIGN.md
        p = torch.nn.Parameter(table.clone())
        p.grad = torch.zeros_like(p)
        opt = torch.optim.Adam([p], lr=1e-2, eps=1e-15, fused=True)
        sec = timeit(opt.step, 20)
        report("torch.optim.Adam(fused) over table", sec, rows * C * 4 * 7)
    if want("memset"):
        gt = torch.zeros(rows, C, device="cuda")
        sec = timeit(lambda: gt.zero_(), 50)
        report("zero grad table", sec, rows * C * 4)
    if want("copy"):
        a = torch.empty(256 * 2 ** 20 // 4, device="cuda")
        b_ = torch.empty_like(a)
        sec = timeit(lambda: b_.copy_(a), 20)
        report("torch copy 256MiB (r+w)", sec, 2 * a.numel() * 4)


if __name__ == "__main__":
    main()
