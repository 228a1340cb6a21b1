"""Data-parallel plumbing on CPU: world_size 2 over gloo (the GPU path uses the same code over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from raw_ngp_amd import parallel
    r, w, _ = parallel.init_from_env("cpu")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    torch.manual_seed(100 + rank)                                  # ranks start different ...
    model = torch.nn.Sequential(torch.nn.Linear(8, 16, bias=False), torch.nn.ReLU(), torch.nn.Linear(16, 3, bias=False))
    table = torch.nn.Parameter(torch.randn(1 << 20, 2) * 1e-2)      # "big" tensor: reduced on its own
    model.register_parameter("table", table)
    parallel.broadcast_module(model)                                # ... and are made identical
    w0 = [p.detach().clone() for p in model.parameters()]
    gen = torch.Generator().manual_seed(7 + rank)                   # every rank: its own ray batch
    x = torch.randn(32, 8, generator=gen)
    idx = torch.randint(0, 1 << 20, (32,), generator=gen)
    loss = (model[2](model[1](model[0](x))) ** 2).mean() + (table[idx] ** 2).sum()
    loss.backward()
    local = [p.grad.detach().clone() for p in model.parameters()]
    red = parallel.GradReducer(model)
    assert len(red.big) == 1 and len(red.small) == 2
    red.all_reduce()
    torch.save({"w0": w0, "local": local, "reduced": [p.grad.clone() for p in model.parameters()]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    parallel.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    for a, b in zip(outs[0]["w0"], outs[1]["w0"]):
        assert torch.equal(a, b)                                    # broadcast made the replicas identical
    for k in range(3):
        mean = (outs[0]["local"][k] + outs[1]["local"][k]) / 2
        for r in range(world):
            np.testing.assert_allclose(outs[r]["reduced"][k].numpy(), mean.numpy(), rtol=1e-6, atol=1e-7)
    assert not torch.equal(outs[0]["local"][2], outs[1]["local"][2])   # the shards really differed


def test_single_process_is_a_noop():
    from raw_ngp_amd import parallel
    assert parallel.world_size() == 1 and parallel.rank() == 0
    m = torch.nn.Linear(2, 2)
    parallel.broadcast_module(m)
    parallel.GradReducer(m).all_reduce()
    parallel.barrier()


def _shard_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from raw_ngp_amd import parallel
    parallel.init_from_env("cpu")
    n = 100_003                                                     # not divisible by anything convenient
    N = parallel.padded_numel(n)
    assert N % (4 * world) == 0 and 0 <= N - n < 4 * world
    g0 = torch.Generator().manual_seed(5)
    param = torch.zeros(N)
    param[:n] = torch.randn(n, generator=g0)                        # replicas start identical
    grad = torch.zeros(N)
    grad[:n] = torch.randn(n, generator=torch.Generator().manual_seed(50 + rank))    # every rank: its own gradient
    x = parallel.Exchange("cpu")
    assert x.carrier == "gloo" and not x.capturable and x.self_test()
    # the guarded set-up of the bare-RCCL carrier: not available over gloo -> None on every rank, after the vote, quickly
    assert parallel.guarded_rccl_exchange(torch.device("cpu"), timeout_s=30) is None
    assert parallel.all_ranks_agree(True, "cpu") and not parallel.all_ranks_agree(rank == 0, "cpu")
    lo, hi = x.shard_bounds(param)
    assert (hi - lo) * world == N and lo == rank * (hi - lo)
    m, v = torch.zeros(hi - lo), torch.zeros(hi - lo)               # moments exist for this rank's shard only
    lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-15
    grad0 = grad.clone()
    for t in (1, 2, 3):
        grad.copy_(grad0)
        g = x.reduce_scatter_avg(grad)                              # the mean of this rank's shard
        assert g is x.shard_of(grad)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        param[lo:hi].sub_((lr / (1 - b1 ** t)) * m / (v.sqrt() / (1 - b2 ** t) ** 0.5 + eps))
        x.all_gather(param)
    # whole-buffer averages (the "allreduce" mode, and the 53 KiB of MLP gradients beside a bfloat16 table gradient)
    a, b16 = grad0.clone(), grad0[:1024].to(torch.bfloat16)
    x.all_reduce_avg(a, b16)
    grad = grad0
    torch.save({"avg": a, "avg16": b16.float()}, os.path.join(out_dir, f"avg{rank}.pt"))
    torch.save({"param": param.clone(), "grad": grad.clone()}, os.path.join(out_dir, f"shard{rank}.pt"))
    parallel.barrier()
    dist.destroy_process_group()


def test_sharded_step_equals_adam_on_the_mean_gradient(tmp_path):
    """parallel.Exchange over gloo: reduce_scatter -> Adam on 1/R of the rows -> all_gather leaves every replica with what a single process gets from
    torch.optim.Adam on the mean of the ranks' gradients."""
    world = 2
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"shard{r}.pt"), weights_only=True) for r in range(world)]
    assert torch.equal(outs[0]["param"], outs[1]["param"])          # bit-identical replicas
    n = 100_003
    p = torch.zeros(outs[0]["param"].numel())
    p[:n] = torch.randn(n, generator=torch.Generator().manual_seed(5))
    p = torch.nn.Parameter(p)
    opt = torch.optim.Adam([p], lr=1e-2, eps=1e-15)
    for _ in range(3):
        p.grad = (outs[0]["grad"] + outs[1]["grad"]) / 2
        opt.step()
    np.testing.assert_allclose(outs[0]["param"].numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-7)
    avgs = [torch.load(os.path.join(tmp_path, f"avg{r}.pt"), weights_only=True) for r in range(world)]
    mean = (outs[0]["grad"] + outs[1]["grad"]) / 2
    for r in range(world):
        np.testing.assert_allclose(avgs[r]["avg"].numpy(), mean.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(avgs[r]["avg16"].numpy(), mean[:1024].numpy(), rtol=2e-2, atol=2e-2)
