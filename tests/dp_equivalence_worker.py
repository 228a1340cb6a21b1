"""One rank of tests/test_gpu_00_dp.py::test_two_ranks_equal_one_rank_on_the_concatenated_batch (launched by torchrun)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def setup(num_rays, dev, **kw):
    from raw_ngp_amd.nerf.engine import FusedTrainer
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    from raw_ngp_amd.nerf.scene import SyntheticDataset
    torch.manual_seed(0)
    # the DEFAULT loss scale (2^16, adapting like the reference's GradScaler): in this scene -- an untrained, dense field, per-
    # sample gradients ~ 1e-7 -- a static 1024 put the f16 deltas of the fused MLP backward into the subnormals, where a batch
    # of 2048 rays (per-ray gradients twice those of a 4096-ray batch) rounds differently: 12 % on the table gradient at
    # 1024, 3e-3 at 2^16, 1e-4 at 2^20.  The test's tolerance is that of the default.
    opt = Options(bound=1.0, num_rays=num_rays, iters=200, background="black", capture_graph=False,
                  **kw)
    data = SyntheticDataset(opt, dev, "train", n_views=6, H=64, W=64)
    model = NeRFNetwork(opt)
    with torch.no_grad():                       # a field with structure, the same in every process
        g = torch.Generator().manual_seed(3)
        model.grid_encoder.embeddings.copy_(torch.rand(model.grid_encoder.embeddings.shape, generator=g) - 0.5)
    return opt, data, FusedTrainer(opt, model, data, device=dev, capacity=4096 * 256)


def draw(data, n, seed, dev):
    b = data.sample_rays(n, torch.Generator(device=dev).manual_seed(seed))
    noises = torch.rand(n, generator=torch.Generator(device=dev).manual_seed(1000 + seed), device=dev)
    return {k: b[k].contiguous() for k in ("rays_o", "rays_d", "images")}, noises


def main():
    out_dir, mode, wire = sys.argv[1], sys.argv[2], sys.argv[3]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    from raw_ngp_amd import _lib, parallel
    rank, world, local = parallel.init_from_env("cuda")
    dev = torch.device("cuda", local)
    _lib.load()
    opt, data, tr = setup(2048, dev, dp_mode=mode, grad_wire=wire)
    assert tr.dp and tr.xchg is not None and tr.xchg.R == world
    flat0 = tr.flat.clone()
    for k in range(steps):
        batch, noises = draw(data, 2048, 100 + rank + 10 * k, dev)
        tr.train_step(batch, noises)
    torch.cuda.synchronize()
    x = tr.xchg
    if tr.split is not None:                # (two level groups: the test that reads `grad` runs the one-group exchange)
        lo, hi, own = 0, 0, tr.gflat[:0]
    else:
        lo, hi = x.shard_bounds(tr.gflat) if mode == "shard" else (0, tr.gflat.numel())
        own = x.shard_of(tr.gflat) if mode == "shard" else tr.gflat
    torch.save({"lo": lo, "hi": hi, "grad": own.float().cpu(), "flat": tr.flat.cpu(), "w": tr.w_flat.cpu(),
                "flat0": flat0.cpu(), "split": tr.split is not None, "n_params": tr.table.numel() + tr.w_flat.numel(),
                "loss": float(tr.loss), "samples": int(tr.arena.counter[0])}, os.path.join(out_dir, f"dp{rank}.pt"))
    parallel.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
