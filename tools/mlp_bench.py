#!/usr/bin/env python3
"""Standalone driver of the fused MLP kernels (forward, backward) on random slabs, for rocprofv3 passes."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd import _lib  # noqa: E402
from raw_ngp_amd._lib import mlp_backend as mb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=196608)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    M = args.samples
    shapes = [(64, 32), (64, 64), (16, 64), (64, 31), (64, 64), (3, 64)]
    ws_ = [torch.randn(*s, device=dev, generator=g) * (1.0 / s[1]) ** 0.5 for s in shapes]
    dws = [torch.empty_like(w) for w in ws_]
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device=dev)
    enc = torch.randn(16, M, 2, device=dev, generator=g) * 0.1
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=-1)
    sigma, rgb = torch.empty(M, device=dev), torch.empty(M, 3, device=dev)
    dsigma, drgb = torch.randn(M, device=dev, generator=g) * 1e-3, torch.randn(M, 3, device=dev, generator=g) * 1e-3
    denc = torch.empty_like(enc)
    wsb = torch.empty(mb.backward_workspace_bytes(M), dtype=torch.uint8, device=dev)
    cnt = torch.tensor([M, M], dtype=torch.int32, device=dev)

    def once():
        mb.prepare(ws_, image)
        mb.forward(enc, M, dirs, cnt, M, image, sigma, rgb)
        mb.backward(enc, M, dirs, dsigma, drgb, cnt, M, image, 1024.0, denc, dws, wsb)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        once()
    torch.cuda.synchronize()
    print(f"{M} samples: {(time.perf_counter() - t0) / args.iters * 1e6:.1f} us per prepare + forward + backward")


if __name__ == "__main__":
    main()
