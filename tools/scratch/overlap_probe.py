#!/usr/bin/env python3
"""Scratch: how well do the table backward's reduce (+ Adam) and the encoder's forward share the GPU?  (timing only: the
forward reads the table while Adam writes it)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raw_ngp_amd import _lib  # noqa: E402
from raw_ngp_amd._lib import engine_backend as eb, gridencoder_backend as gb  # noqa: E402
from raw_ngp_amd.gridencoder.grid import level_table  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    N, K = 4096, 34
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
    denc.view(L, N, K, 2)[:, :, (2 * K) // 3:] = 0.0
    grad = torch.zeros(rows, 2, device=dev)
    ws = torch.empty(gb.backward_workspace_bytes(B, L, rows), dtype=torch.uint8, device=dev)
    cnt = torch.tensor([B, B, 0, 0], dtype=torch.int32, device=dev)
    t_m, t_v = torch.zeros_like(table), torch.zeros_like(table)
    hyper = torch.tensor([1e-6, 0.1, 31.6, 0.0], device=dev)
    adam = (table, t_m, t_v, hyper, 0.9, 0.999, 1e-15)
    x01b = torch.empty(B, 3, device=dev)

    def prepare():
        gb.grid_backward_binned_prepare(None, 0.0, offsets, rows, cnt, B, L, L, S, H, ws, merge_max_res=414, stage=1)

    def fwd():
        eb.grid_encode_forward_slab(xyz, 1.0, table, offsets, enc, x01b, cnt, B, B, L, L, S, H)

    def apply():
        gb.grid_backward_binned_apply(denc, x01, offsets, grad, cnt, B, B, L, L, S, H, ws, adam=adam)

    eb.grid_encode_forward_slab(xyz, 1.0, table, offsets, enc, x01, cnt, B, B, L, L, S, H)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def body(mode):
        main = torch.cuda.current_stream()
        if mode in ("apply", "both", "serial", "late"):
            apply() if mode != "late" else None
        if mode == "serial":
            fwd()
        if mode == "both":
            sb.wait_stream(main)
        if mode == "fwd":
            fwd()
        if mode == "late":       # the forward starts when the fill is done: beside the reduce only
            gb.grid_backward_binned_apply(denc, x01, offsets, grad, cnt, B, B, L, L, S, H, ws, adam=adam)

    def timed(mode):
        prepare()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            for rep in range(4):
                prepare()
                if mode == "apply":
                    apply()
                elif mode == "fwd":
                    fwd()
                elif mode == "serial":
                    apply()
                    fwd()
                elif mode == "both":
                    sb.wait_stream(main)
                    with torch.cuda.stream(sb):
                        fwd()
                    apply()
                    main.wait_stream(sb)
        ts = []
        for i in range(13):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            b.synchronize()
            if i >= 3:
                ts.append(a.elapsed_time(b) * 1e3 / 4)
        return float(np.median(ts))

    for mode in ("apply", "fwd", "serial", "both"):
        print(f"{mode:8s} {timed(mode):7.1f} us")


if __name__ == "__main__":
    main()
