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

# the carrier of the data-parallel step: bare RCCL calls, set up and self-tested the way a multi-rank job does it
# (unique id through the store, worker thread under a deadline, eager + captured + two-stream captured known answers)
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw_ngp_amd import parallel  # noqa: E402
xchg = parallel.guarded_rccl_exchange(dev, timeout_s=120)
assert xchg is not None and xchg.carrier == "rccl" and xchg.capturable
assert xchg.self_test()
flag = torch.tensor([3], dtype=torch.int32, device=dev)
xchg.all_reduce_max(flag)
torch.cuda.synchronize()
assert int(flag) == 3
xchg.close()
print("raw_ngp_amd.parallel: guarded RCCL exchange passed its self-test (eager, graph, two-stream graph) on one rank")
dist.destroy_process_group()
