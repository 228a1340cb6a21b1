"""Scratch: what fraction of the samples of the light-conditioned / pose / HDR configuration lies behind the early stop?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raw_ngp_amd.nerf import pose as P
from raw_ngp_amd.nerf.engine import FusedTrainer
from raw_ngp_amd.nerf.network import NeRFNetwork
from raw_ngp_amd.nerf.options import Options
from raw_ngp_amd.nerf.scene import SyntheticDataset
dev = torch.device("cuda")
torch.manual_seed(0)
for bound, hdr in ((2.0, True), (1.0, False)):
    opt = Options(bound=bound, num_rays=4096, iters=3000, rfield=True, pose_opt="barf", noise=0.03, c_lr=1e-3,
                  image_mode="HDR" if hdr else "LDR")
    data = SyntheticDataset(opt, dev, "train", n_views=40, H=200, W=200)
    data.ldirs = torch.from_numpy(P.synthetic_light_dirs(40)).to(dev)
    if hdr:
        data.exposures = torch.from_numpy(np.random.default_rng(5).choice([0.5, 1.0, 2.0], 40).astype("float32")).to(dev)
        rgb = data.images[..., :3].float() * data.exposures.view(-1, 1, 1, 1)
        data.images[..., :3] = rgb.clamp(max=255).to(torch.uint8)
    tr = FusedTrainer(opt, NeRFNetwork(opt), data, device=dev, seed=0)
    for it in range(2500):
        tr.train_step()
        if (it + 1) % 500 == 0:
            torch.cuda.synchronize()
            n = int(tr.arena.counter[0])
            live = int((tr.dsigma[:n] != 0).sum() + 0)
            print(f"bound {bound} hdr {hdr} step {it + 1}: samples {n}, with a gradient {live} ({live / max(n, 1):.2f})", flush=True)
