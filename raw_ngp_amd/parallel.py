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


def padded_numel(n, world=None, align=4):
    """Smallest length >= n that splits into `world` equal shards of a multiple of `align` elements."""
    q = (world or world_size()) * align
    return (n + q - 1) // q * q


class ShardedStep:
    """The exchange of SURVEY 8e's second variant, around a rank-local optimiser:

        reduce_scatter(grad, AVG)  ->  optimiser on this rank's 1/R of the flat parameter  ->  all_gather(param)

    Same wire bytes as an all-reduce of the gradient (which IS a reduce-scatter followed by an all-gather), but the
    optimiser sweep -- parameter, two moments, gradient: 28 B per element of a 12.2 M-element table -- and the moments'
    memory shrink by the number of ranks.  `param` and `grad` are flat tensors of padded_numel() elements (grad may be
    bfloat16: the wire format); the caller runs its optimiser on (param_shard, grad_shard) between the two calls.
    Works over RCCL ("nccl": AVG folded into the collective, in-place gather) and gloo (CPU tests, one-GPU rehearsal)."""

    def __init__(self, param, grad):
        self.R, self.r = world_size(), rank()
        n = param.numel()
        assert param.dim() == 1 and grad.dim() == 1 and grad.numel() == n, "ShardedStep: flat tensors of equal length"
        assert n % (4 * self.R) == 0, "ShardedStep: length must come from padded_numel()"
        self.param, self.grad = param, grad
        self.n_shard = n // self.R
        self.lo = self.r * self.n_shard
        self.param_shard = param[self.lo:self.lo + self.n_shard]                  # a view: updated in place
        self.grad_shard = grad if self.R == 1 else torch.empty(self.n_shard, dtype=grad.dtype, device=grad.device)
        self._nccl = is_dist() and dist.get_backend() == "nccl"
        self._send = None if (self.R == 1 or self._nccl) else torch.empty_like(self.param_shard)

    def reduce_scatter(self):
        if self.R == 1:
            return
        if self._nccl:
            dist.reduce_scatter_tensor(self.grad_shard, self.grad, op=dist.ReduceOp.AVG)
        else:
            dist.reduce_scatter_tensor(self.grad_shard, self.grad)
            self.grad_shard.div_(self.R)

    def all_gather(self):
        if self.R == 1:
            return
        if self._nccl:
            dist.all_gather_into_tensor(self.param, self.param_shard)              # in place: shard r of the output
        else:
            self._send.copy_(self.param_shard)
            dist.all_gather_into_tensor(self.param, self._send)
