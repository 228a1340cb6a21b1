"""Scratch: where does the 2-rank vs 1-rank gradient difference come from?  One process, no collectives: average the
gradients of two 2048-ray steps by hand and compare with one 4096-ray step, at several loss scales."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dp_equivalence_worker as W
from raw_ngp_amd import _lib
_lib.load()
dev = torch.device("cuda")
for ls in (1024.0, 65536.0, 2.0 ** 20):
    grads = []
    for r in range(2):
        opt, data, tr = W.setup(2048, dev, fuse_adam=False, loss_scale=ls)
        batch, noises = W.draw(data, 2048, 100 + r, dev)
        tr.train_step(batch, noises)
        torch.cuda.synchronize()
        grads.append(tr.gflat.clone())
    opt, data, one = W.setup(4096, dev, fuse_adam=False, loss_scale=ls)
    parts = [W.draw(data, 2048, 100 + r, dev) for r in range(2)]
    batch = {k: torch.cat([p[0][k] for p in parts]) for k in parts[0][0]}
    one.train_step(batch, torch.cat([p[1] for p in parts]))
    torch.cuda.synchronize()
    n_t = one.table.numel()
    g2 = (grads[0][:one.gflat.numel()] + grads[1][:one.gflat.numel()]) / 2
    g1 = one.gflat
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"loss_scale {ls}: table rel {rel(g2[:n_t], g1[:n_t]):.3e}  mlp rel {rel(g2[n_t:], g1[n_t:]):.3e}  |g| {float(g1.norm()):.3e}", flush=True)
