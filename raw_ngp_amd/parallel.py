"""Ray-batch data parallelism: one process per GPU, torch.distributed over RCCL (backend "nccl").

The hot path shards by ray batch: every rank draws its own rays with a rank-offset
seed, the model (hash table + tiny MLPs + density grid) is replicated, and the only exchange per
step is the gradient all-reduce -- the 12.2 M-parameter hash-table gradient (48.8 MB f32) as ONE
collective on the tensor itself (no flatten copy) plus one small coalesced buffer for the MLP weights.
The reference only has a vestigial DDP wrap (nerf/train_utils.py:384-386) and no launcher; this module
is the working counterpart.  On CPU (tests) the same code runs over gloo.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if is_dist() else 0


def world_size():
    return dist.get_world_size() if is_dist() else 1


def init_from_env(device_type="cuda"):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun env).
    Returns (rank, world_size, local device index); a single process without the env stays un-initialised.
    Rehearsal on a one-GPU box: NGP_DIST_BACKEND=gloo NGP_LOCAL_DEVICE=0 puts every rank on the same device and
    reduces through gloo (RCCL wants one device per rank)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1:
        if os.environ.get("NGP_DP_REHEARSAL") == "1" and device_type == "cuda" and not is_dist():
            # one-rank RCCL group: lets a single GPU run the data-parallel step (real collectives, nothing to average)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        return 0, 1, 0
    r, lr = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if device_type == "cuda":
        lr = int(os.environ.get("NGP_LOCAL_DEVICE", lr))
        torch.cuda.set_device(lr)
        backend = os.environ.get("NGP_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=r, world_size=ws, device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend, rank=r, world_size=ws)
    else:
        dist.init_process_group("gloo", rank=r, world_size=ws)
    return r, ws, lr


def barrier():
    if is_dist():
        dist.barrier()


def broadcast_module(module, src=0):
    """Make parameters and buffers identical on every rank (rank `src` wins)."""
    if not is_dist():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


class GradReducer:
    """Averages gradients across ranks after backward.

    Large gradients (>= `big` elements: the hash-table) are reduced in place, each as its own collective,
    launched first so the wire time overlaps the packing of the small ones; everything else is packed
    into one flat buffer (one collective for all MLP weights)."""

    def __init__(self, module, big=1 << 20):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.big = [p for p in self.params if p.numel() >= big]
        self.small = [p for p in self.params if p.numel() < big]
        self._flat = None

    def all_reduce(self):
        if not is_dist():
            return
        ws = world_size()
        handles = []
        for p in self.big:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            handles.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True))
        small = [p for p in self.small]
        if small:
            n = sum(p.numel() for p in small)
            if self._flat is None or self._flat.numel() != n or self._flat.device != small[0].device:
                self._flat = torch.zeros(n, dtype=small[0].dtype, device=small[0].device)
            off = 0
            for p in small:
                k = p.numel()
                if p.grad is None:
                    self._flat[off:off + k].zero_()
                else:
                    self._flat[off:off + k].copy_(p.grad.reshape(-1))
                off += k
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
            self._flat.div_(ws)
            off = 0
            for p in small:
                k = p.numel()
                if p.grad is None:
                    p.grad = self._flat[off:off + k].view_as(p).clone()
                else:
                    p.grad.copy_(self._flat[off:off + k].view_as(p))
                off += k
        for h in handles:
            h.wait()
        for p in self.big:
            p.grad.div_(ws)
