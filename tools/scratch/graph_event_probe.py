#!/usr/bin/env python3
"""Scratch: can HIP events recorded INSIDE a captured graph be timed after a replay?"""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]


def ev():
    e = ctypes.c_void_p()
    assert hip.hipEventCreate(ctypes.byref(e)) == 0
    return e


x = torch.randn(1 << 24, device="cuda")
y = torch.empty_like(x)
a, b, c = ev(), ev(), ev()
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    st = torch.cuda.current_stream().cuda_stream
    y.copy_(x)
    print("record a:", hip.hipEventRecord(a, st))
    for _ in range(4):
        y.mul_(1.0001)
    print("record b:", hip.hipEventRecord(b, st))
    y.add_(1.0)
    print("record c:", hip.hipEventRecord(c, st))
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    ms = ctypes.c_float()
    rc = hip.hipEventElapsedTime(ctypes.byref(ms), a, b)
    ms2 = ctypes.c_float()
    rc2 = hip.hipEventElapsedTime(ctypes.byref(ms2), b, c)
    print(f"replay {i}: rc {rc} a->b {ms.value * 1e3:.1f} us   rc {rc2} b->c {ms2.value * 1e3:.1f} us")
# reference: eager timing of the same work
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(4):
    y.mul_(1.0001)
e.record()
torch.cuda.synchronize()
print(f"eager 4 x mul_: {s.elapsed_time(e) * 1e3:.1f} us")
