#!/usr/bin/env python3
"""Per-level, per-phase time of the binned table backward's fill kernel: runs tools/grid_bench.py's ray-ordered samples
(the last third of every ray with a zero gradient, like samples behind the compositor's early stop) through a DIAGNOSTIC
build of the library (-DNGP_STAMP_FILL: s_memtime at the phase boundaries, see grid_backward_binned.hip) and prints, per
level, the workgroups, their mean lifetime and the share of each phase as thread 0 (the scanning wave) sees it.  The
diagnostic build's run time is not a measurement (its waits forbid overlaps the real kernel has).
    make -C raw_ngp_amd/csrc stamp_fill && NGP_HIP_LIB=tools/bin/libngp_stamp_fill.so python tools/fill_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd import _lib  # noqa: E402
from raw_ngp_amd._lib import engine_backend as eb, gridencoder_backend as gb  # noqa: E402
from raw_ngp_amd.gridencoder.grid import level_table  # noqa: E402

# phase names of the global-bins kernel (NGP_BINNED_LOCAL=0) and of the tile-local one (the default)
PHASES_GLOBAL = ["header + zero hist", "x, grad arrive", "weights/runs/hash/hist", "barrier 1 skew", "scan + cursor issue",
                 "cursor RTT + staging", "stream-out"]
PHASES_LOCAL = ["header + loads issued", "-", "loads arrive, weights/runs/hash/hist", "barrier 1 skew",
                "scan + directory + padding", "staging (+ barrier)", "stream-out"]
PHASES = PHASES_GLOBAL if os.environ.get("NGP_BINNED_LOCAL", "1") == "0" else PHASES_LOCAL


def main():
    lib = _lib.load()
    assert hasattr(lib, "ngp_dbg_read_fill_stamps"), "needs the -DNGP_STAMP_FILL build (NGP_HIP_LIB=...)"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    N, K = 4096, 34                                        # ~ 139 k samples, the bench's steady state
    o = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 3.0
    target = (torch.rand(N, 3, device=dev, generator=g) - 0.5) * 0.8
    d = torch.nn.functional.normalize(target - o, dim=-1)
    t0 = (target - o).norm(dim=-1, keepdim=True) - 0.06
    t = t0 + (2 * 3 ** 0.5 / 1024) * torch.arange(K, device=dev).float()[None]
    xyz = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3).clamp(-0.999, 0.999).contiguous()
    B = xyz.shape[0]
    scale = float(np.exp2(np.log2(2048 / 16) / 15))
    offsets_np = level_table(3, 16, scale, 16, 19)
    offsets = torch.from_numpy(offsets_np).to(dev)
    L, H, S, rows = 16, 16, float(np.log2(scale)), int(offsets_np[-1])
    table = (torch.rand(rows, 2, device=dev, generator=g) - 0.5) * 2e-4
    enc, x01 = torch.empty(L, B, 2, device=dev), torch.empty(B, 3, device=dev)
    denc = torch.randn(L, B, 2, device=dev, generator=g)
    denc.view(L, N, K, 2)[:, :, (2 * K) // 3:] = 0.0
    cap = 655360                                           # the engine's arena: launch geometry as in the step
    ws = torch.empty(gb.backward_workspace_bytes(cap, L, rows), dtype=torch.uint8, device=dev)
    pad = lambda a, n: torch.cat([a, torch.zeros(n - a.shape[0], *a.shape[1:], device=dev)]).contiguous()
    xyz_c, x01_c = pad(xyz, cap), torch.empty(cap, 3, device=dev)
    enc_c, denc_c = torch.empty(L, cap, 2, device=dev), torch.zeros(L, cap, 2, device=dev)
    denc_c[:, :B] = denc
    cnt = torch.tensor([B, B, 0, 0], dtype=torch.int32, device=dev)
    t_m, t_v = torch.zeros_like(table), torch.zeros_like(table)
    hyper = torch.tensor([1e-6, 0.1, 31.6, 0.0], device=dev)
    adam = (table, t_m, t_v, hyper, 0.9, 0.999, 1e-15)

    def once():
        gb.grid_backward_binned_prepare(None, 0.0, offsets, rows, cnt, cap, L, L, S, H, ws, merge_max_res=414, stage=1)
        eb.grid_encode_forward_slab(xyz_c, 1.0, table, offsets, enc_c, x01_c, cnt, cap, cap, L, L, S, H, binned_workspace=ws)
        gb.grid_backward_binned_prepare(None, 0.0, offsets, rows, cnt, cap, L, L, S, H, ws, stage=2, single_segment=True)
        gb.grid_backward_binned_apply(denc_c, x01_c, offsets, None, cnt, cap, cap, L, L, S, H, ws, adam=adam)

    out = (ctypes.c_ulonglong * (64 * 8))()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    lib.ngp_dbg_read_fill_stamps(out, 1)
    iters = 10
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    lib.ngp_dbg_read_fill_stamps(out, 1)
    a = np.array(list(out), dtype=np.float64).reshape(64, 8)[:L]
    print(f"{B} samples; s_memtime ticks (shader-clock cycles) per workgroup, mean over {iters} launches")
    print("level   WGs  life  " + "  ".join(f"{p[:22]:>22}" for p in PHASES))
    for lv in range(L):
        n = a[lv, 0]
        if n == 0:
            continue
        ph = a[lv, 1:] / n
        print(f"{lv:5d} {n / iters:5.0f} {ph.sum():5.0f}  " + "  ".join(f"{v:22.1f}" for v in ph))
    tot = a[:, 1:].sum(0)
    print("share of all workgroup lifetimes: " + ", ".join(f"{p}: {v / tot.sum() * 100:.1f} %" for p, v in zip(PHASES, tot)))


if __name__ == "__main__":
    main()
