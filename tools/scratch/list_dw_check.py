import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_fused_mlp import make_weights
from raw_ngp_amd import _lib
mb = _lib.mlp_backend
M, dead = 30000, 0.33
W = make_weights(seed=3)
g = torch.Generator(device="cuda").manual_seed(M)
stride = M + 9
enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
dirs = torch.randn(M, 3, device="cuda", generator=g)
dsigma = torch.randn(M, device="cuda", generator=g) * 1e-3
drgb = torch.randn(M, 3, device="cuda", generator=g) * 1e-3
live = torch.ones(M, dtype=torch.bool, device="cuda")
at = 0
rng = np.random.default_rng(M)
while at < M:
    n = int(rng.integers(7, 60)); k = int(round(n * (1.0 - dead)))
    live[at + k:at + n] = False; at += n
dsigma[~live] = 0; drgb[~live] = 0
idx = torch.nonzero(live).flatten().to(torch.int32)
image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda"); mb.prepare(W, image)
def run(si, cnt):
    denc = torch.zeros(16, stride, 2, device="cuda"); dws = [torch.empty_like(w) for w in W]
    mb.backward(enc, stride, dirs, dsigma, drgb, cnt, M, image, 1024.0, denc, dws, sample_index=si)
    return dws
a = run(None, None)
pad = torch.zeros(M, dtype=torch.int32, device="cuda"); pad[:idx.numel()] = idx
b = run(pad, torch.tensor([idx.numel()], dtype=torch.int32, device="cuda"))
for i, (x, y) in enumerate(zip(a, b)):
    print(i, "max|dW|", float(x.abs().max()), "max diff", float((x - y).abs().max()), "rel L2", float((x - y).norm() / x.norm()))
