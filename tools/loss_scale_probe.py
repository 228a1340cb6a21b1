"""Gradient error of the fused step against fp32 autograd as a function of the loss scale (static), on an untrained field and
after some training: what the f16 deltas of the fused MLP backward lose to underflow.  Prints one line per scale.
    python tools/loss_scale_probe.py [--rays 4096] [--train 0,300] [--scales 10,16,18,20,22,24]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--train", default="0,300")
    ap.add_argument("--scales", default="10,16,18,20,22,24")
    args = ap.parse_args()
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    dev = torch.device("cuda")
    for warm in [int(t) for t in args.train.split(",")]:
        for e in [int(t) for t in args.scales.split(",")]:
            torch.manual_seed(0)
            opt = Options(bound=1.0, num_rays=args.rays, iters=2000, fused_mlp=True, background="black",
                          loss_scale=2.0 ** e, dynamic_loss_scale=False)
            data = SyntheticDataset(opt, dev, "train", n_views=8, H=128, W=128)
            model = NeRFNetwork(opt).cuda()
            if warm:                    # train with the dynamic default, then measure one batch at the static scale
                o2 = Options(bound=1.0, num_rays=args.rays, iters=2000, fused_mlp=True, background="black")
                tr = FusedTrainer(o2, model, data, device="cuda", capacity=args.rays * 256)
                tr.train(warm)
                torch.cuda.synchronize()
                st = tr.scaler.state()
                del tr
            eng = FusedTrainer(opt, model, data, device="cuda", capacity=args.rays * 256)
            model.train()
            if not warm:
                model.update_extra_state()
            batch = data.sample_rays(opt.num_rays, torch.Generator(device="cuda").manual_seed(1))
            gt = batch["images"]
            eng.forward_backward(batch["rays_o"].contiguous(), batch["rays_d"].contiguous(), gt.contiguous(),
                                 torch.zeros(opt.num_rays, device="cuda"))
            M = int(eng.arena.counter[0])
            opt.fused_mlp = False
            model.zero_grad()
            out = model.render(batch["rays_o"], batch["rays_d"], bg_color=0, perturb=False)
            assert out["num_points"] == M
            loss = ((out["image"] - gt[:, :3] * gt[:, 3:]) ** 2).mean(-1).mean()
            loss.backward()
            ref_t = model.grid_encoder.embeddings.grad
            ref_w = torch.cat([l.weight.grad.reshape(-1) for l in list(model.grid_mlp.net) + list(model.view_mlp.net)])
            rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
            fin = bool(torch.isfinite(eng.table_grad).all() and torch.isfinite(eng.w_grad).all())
            print(f"trained {warm:4d} steps  scale 2^{e:2d}  samples {M:7d}  table {rel(eng.table_grad, ref_t):.2e}  "
                  f"weights {rel(eng.w_grad, ref_w):.2e}  finite {fin}" + (f"  (dynamic run: {st})" if warm else ""), flush=True)
            del eng


if __name__ == "__main__":
    main()
