#!/usr/bin/env python3
"""Where a tile of the MLP backward (view kernel) spends its cycles: runs tools/mlp_bench.py's workload on a DIAGNOSTIC
build of the library (-DNGP_STAMP: s_memtime stamps around the sections, see fused_mlp_backward.hip) and prints the share
of each section.  The diagnostic build's run time is not a measurement (its fences forbid overlaps the real kernel has).
    NGP_HIP_LIB=tools/bin/libngp_stamp.so python tools/mlp_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd import _lib  # noqa: E402
from raw_ngp_amd._lib import mlp_backend as mb  # noqa: E402

NAMES = ["prefetch issue + skip test", "input conversion (+ wait for last iteration's loads), SH", "forward recompute",
         "delta6, dW6, delta5", "dW5, delta4", "dW4, d x3, store", "partial-sum flush"]


def main():
    lib = _lib.load()
    assert hasattr(lib, "ngp_dbg_read_stamps"), "needs the -DNGP_STAMP build (NGP_HIP_LIB=...)"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    M = 137000
    shapes = [(64, 32), (64, 64), (16, 64), (64, 31), (64, 64), (3, 64)]
    ws_ = [torch.randn(*s, device=dev, generator=g) * (1.0 / s[1]) ** 0.5 for s in shapes]
    dws = [torch.empty_like(w) for w in ws_]
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device=dev)
    enc = torch.randn(16, M, 2, device=dev, generator=g) * 0.1
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=-1)
    dsigma, drgb = torch.randn(M, device=dev, generator=g) * 1e-3, torch.randn(M, 3, device=dev, generator=g) * 1e-3
    denc = torch.empty_like(enc)
    wsb = torch.empty(mb.backward_workspace_bytes(M), dtype=torch.uint8, device=dev)
    mb.prepare(ws_, image)
    out = (ctypes.c_ulonglong * 16)()
    for _ in range(3):
        mb.backward(enc, M, dirs, dsigma, drgb, None, M, image, 1024.0, denc, dws, wsb)
    torch.cuda.synchronize()
    lib.ngp_dbg_read_stamps(out, 1)
    iters = 10
    for _ in range(iters):
        mb.backward(enc, M, dirs, dsigma, drgb, None, M, image, 1024.0, denc, dws, wsb)
    torch.cuda.synchronize()
    lib.ngp_dbg_read_stamps(out, 1)
    tot = sum(out[:7])
    tiles = (M + 31) // 32
    for i, n in enumerate(NAMES):
        print(f"{out[i] / tot * 100:5.1f} %  {out[i] / iters / (tiles if i < 6 else 1024):9.0f} cycles per {'tile' if i < 6 else 'wave'}   {n}")
    if out[7]:
        print(f"        {out[7] / iters / 1024:9.0f} cycles per wave   arrival skew at the flush's first barrier (not in the total)")
    print(f"total {tot / iters / 1024:.0f} cycles per wave")


if __name__ == "__main__":
    main()
