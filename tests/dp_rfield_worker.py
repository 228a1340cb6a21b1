"""One rank of tests/test_gpu_00_dp.py::test_light_conditioned_pose_step_under_data_parallelism (launched by torchrun):
the configs[3]-shaped fused step (rfield + BARF pose refinement + HDR loss) with the exchange step of data parallelism."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    from raw_ngp_amd import _lib, parallel
    from raw_ngp_amd.nerf import pose as P
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    rank, world, local = parallel.init_from_env("cuda")
    dev = torch.device("cuda", local)
    _lib.load()
    torch.manual_seed(0)
    views = 8
    opt = Options(bound=1.0, num_rays=1024, iters=300, rfield=True, pose_opt="barf", noise=0.03, image_mode="HDR",
                  background="black")
    data = SyntheticDataset(opt, dev, "train", n_views=views, H=64, W=64)
    data.ldirs = torch.from_numpy(P.synthetic_light_dirs(views)).to(dev)
    data.exposures = torch.from_numpy(np.random.default_rng(5).choice([0.5, 1.0, 2.0], views).astype(np.float32)).to(dev)
    tr = FusedTrainer(opt, NeRFNetwork(opt), data, device=dev, seed=0, capacity=1024 * 200)
    assert tr.dp and tr.rfield and tr.pose and tr.hdr and tr.xchg is not None and tr.xchg.R == world
    xi0 = tr.xi.clone()
    first = None
    for it in range(steps):
        loss = tr.train_step()
        if it == 3:
            first = float(loss)
    torch.cuda.synchronize()
    torch.save({"flat": tr.flat.cpu(), "xi": tr.xi.cpu(), "xi0": xi0.cpu(), "poses": tr.poses_refined.cpu(),
                "bitfield": tr.model.density_bitfield.cpu(), "loss": float(tr.loss), "first": first,
                "samples": int(tr.samples_seen)}, os.path.join(out_dir, f"rf{rank}.pt"))
    parallel.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
