#!/usr/bin/env python3
"""One-rank RCCL sanity check of the collectives the data-parallel step issues (dtype / op support is validated by the
library whatever the world size): bfloat16 AVG all-reduce in place, f32 AVG, async + wait between graph replays,
MIN / MAX on float64.   python tools/rccl_check.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
dev = torch.device("cuda", 0)
wire = torch.randn(6098120, 2, device=dev).to(torch.bfloat16)
ref = wire.clone()
w = torch.randn(13504, device=dev)
wref = w.clone()
g = torch.cuda.CUDAGraph()
x = torch.zeros(1024, device=dev)
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    x.add_(1.0)
for _ in range(5):
    g.replay()
    small = dist.all_reduce(w, op=dist.ReduceOp.AVG, async_op=True)
    g.replay()
    big = dist.all_reduce(wire, op=dist.ReduceOp.AVG, async_op=True)
    big.wait()
    small.wait()
    g.replay()
torch.cuda.synchronize()
assert torch.equal(wire, ref) and torch.equal(w, wref) and float(x[0]) == 15.0
s = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
lo, hi = s.clone(), s.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN)
dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert torch.equal(lo, hi)
print("rccl ok: bf16 AVG, f32 AVG (async), f64 MIN/MAX, interleaved with graph replays")
dist.destroy_process_group()
