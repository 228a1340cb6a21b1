#!/usr/bin/env python3
"""What does a cross-stream edge cost inside a hipGraph?  A chain of K 'steps' on a main stream (two kernels each: a long one
that streams ~150 MB and a short one) with a short side chain per step, wired four ways:
  none        no side stream at all (the side kernels run on the main stream)
  fork+join   side waits for the main stream at the step's start, main waits for the side stream at the step's end (the engine)
  join        the side stream is forked once, main waits for it at every step's end
  fork        side waits for main at every step's start, main waits for the side stream once, at the end
Prints us per step of a graph replay.   python tools/ubench/graph_edges.py"""
import time

import torch

dev = torch.device("cuda", 0)
K = 30
big = torch.zeros(150 * 1024 * 1024 // 8, device=dev)          # read + write = 150 MB
small = torch.zeros(4096, device=dev)
side_buf = [torch.zeros(1 << 18, device=dev) for _ in range(K + 1)]
main = torch.cuda.Stream(dev)
side = torch.cuda.Stream(dev)


def build(mode):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
            if mode in ("join",):
                side.wait_stream(main)
            for k in range(K):
                if mode in ("fork+join", "fork"):
                    side.wait_stream(main)
                if mode == "none":
                    side_buf[k].add_(1.0)
                    side_buf[k].mul_(0.5)
                else:
                    with torch.cuda.stream(side):
                        side_buf[k].add_(1.0)
                        side_buf[k].mul_(0.5)
                small.add_(1.0)
                big.add_(1.0)
                if mode in ("fork+join", "join"):
                    main.wait_stream(side)
            if mode in ("fork",):
                main.wait_stream(side)
    return g


for mode in ("none", "fork+join", "join", "fork", "none", "fork+join"):
    g = build(mode)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        g.replay()
    torch.cuda.synchronize()
    print(f"{mode:10s} {(time.perf_counter() - t0) / R / K * 1e6:7.2f} us/step")
