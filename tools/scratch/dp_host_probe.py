"""Scratch: host time of the parts of a timed (probed) data-parallel step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NGP_DP_REHEARSAL"] = "1"
from raw_ngp_amd import _lib, parallel
from raw_ngp_amd.nerf.network import NeRFNetwork
from raw_ngp_amd.nerf.options import Options
from raw_ngp_amd.nerf.scene import SyntheticDataset
from raw_ngp_amd.nerf.engine import FusedTrainer
parallel.init_from_env("cuda")
dev = torch.device("cuda", 0)
_lib.load()
torch.manual_seed(0)
opt = Options(bound=1.0, background="random", num_rays=4096, iters=5000, dp_rehearsal=True)
data = SyntheticDataset(opt, dev, "train", n_views=20, H=400, W=400)
tr = FusedTrainer(opt, NeRFNetwork(opt), data, device=dev)
_lib.set_probe(("ngp_x_grid_backward_binned_apply_mlp", "ngp_x_grid_backward_binned_prepare", "ngp_x_grid_encode_forward_slab"), 5, every=8)
tr.collective_events = []
tr.train(400)
torch.cuda.synchronize()
# instrument: time each part of the next timed steps
import types
orig_capture = tr._capture
def timed_parts(parts, tag):
    out = []
    for i, p in enumerate(parts):
        def f(p=p, i=i):
            t0 = time.perf_counter(); p(); dt = time.perf_counter() - t0
            stats.setdefault((tag, i, getattr(p, "__name__", str(p))[:40]), []).append(dt)
        out.append(f)
    return out
stats = {}
for key in list(tr.graphs):
    if isinstance(key, tuple) and len(key) == 3 and key[1] is True:
        tr.graphs[key] = timed_parts(tr.graphs[key], key)
t0 = time.perf_counter(); tr.train(400); host = time.perf_counter() - t0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"host {host/400*1e3:.4f} ms/step wall {wall/400*1e3:.4f} ms/step")
for k, v in sorted(stats.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
    print(k, len(v), f"{sum(v)/len(v)*1e6:.1f} us")
