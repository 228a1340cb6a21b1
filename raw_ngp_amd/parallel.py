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


class Exchange:
    """The collectives of one data-parallel optimiser step on flat buffers, behind one face for three carriers:

      "rccl"   bare RCCL calls on the current HIP stream (raw_ngp_amd.rccl): in place, capturable -- a hipGraph of several
               training steps contains its reduce-scatters and all-gathers like any other kernel node
      "torch"  torch.distributed on an initialised "nccl" group: eager calls between the captured segments of a step
      "gloo"   torch.distributed over gloo: CPU tests and the two-ranks-on-one-GPU rehearsal (no in-place forms, no AVG)

    reduce_scatter_avg(buf) leaves the mean of shard `rank` in shard_of(buf) -- a view of buf for "rccl" / "torch", a
    buffer of its own for gloo; all_gather(buf) publishes every rank's shard of buf in place; all_reduce_avg(*bufs)
    averages whole buffers.  Buffers are flat, padded_numel() long."""

    def __init__(self, device, carrier=None, uid=None):
        self.R, self.r = world_size(), rank()
        self.device = torch.device(device)
        backend = dist.get_backend() if is_dist() else None
        if carrier is None:
            carrier = "rccl" if (backend == "nccl" and self.device.type == "cuda") else "torch"
        if backend != "nccl" and carrier != "none":
            carrier = "gloo" if is_dist() else "none"
        self.carrier = carrier
        self.capturable = carrier in ("rccl", "none")
        self.comm = None
        self._recv, self._send = {}, {}
        if carrier == "rccl":
            from . import rccl
            self.comm = rccl.Communicator(self.device, uid=uid)

    def close(self, abort=False):
        """Give the RCCL communicator back (abort: without waiting for outstanding work)."""
        if self.comm is not None:
            (self.comm.abort if abort else self.comm.destroy)()
            self.comm = None
            self.carrier, self.capturable = "none", True

    def shard_bounds(self, buf):
        n = buf.numel()
        assert buf.dim() == 1 and n % (4 * self.R) == 0, "Exchange: flat buffers of padded_numel() elements"
        k = n // self.R
        return self.r * k, (self.r + 1) * k

    def shard_of(self, buf):
        """Where reduce_scatter_avg(buf) leaves this rank's result."""
        lo, hi = self.shard_bounds(buf)
        if self.carrier != "gloo":
            return buf[lo:hi]
        key = (buf.data_ptr(), buf.dtype)
        if key not in self._recv:
            self._recv[key] = torch.empty(hi - lo, dtype=buf.dtype, device=buf.device)
        return self._recv[key]

    def reduce_scatter_avg(self, buf):
        out = self.shard_of(buf)
        if self.carrier == "rccl":
            self.comm.reduce_scatter_(buf)
        elif self.carrier == "torch":
            dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.AVG)
        elif self.carrier == "gloo":
            dist.reduce_scatter_tensor(out, buf)
            out.div_(self.R)
        return out

    def all_gather(self, buf):
        lo, hi = self.shard_bounds(buf)
        if self.carrier == "rccl":
            self.comm.all_gather_(buf)
        elif self.carrier == "torch":
            dist.all_gather_into_tensor(buf, buf[lo:hi])
        elif self.carrier == "gloo":
            key = (buf.data_ptr(), buf.dtype)
            if key not in self._send:
                self._send[key] = torch.empty(hi - lo, dtype=buf.dtype, device=buf.device)
            self._send[key].copy_(buf[lo:hi])
            dist.all_gather_into_tensor(buf, self._send[key])

    def all_reduce_avg(self, *bufs):
        if self.carrier == "rccl":
            self.comm.all_reduce_(*bufs)
        elif self.carrier == "torch":
            # blocking collectives are enqueued on the CURRENT stream by ProcessGroupNCCL; one ncclGroup for all of them
            with dist._coalescing_manager(device=self.device):
                for b in bufs:
                    dist.all_reduce(b, op=dist.ReduceOp.AVG)
        elif self.carrier == "gloo":
            for b in bufs:
                if b.dtype == torch.bfloat16 and not b.is_cuda:
                    t = b.float()
                    dist.all_reduce(t)
                    b.copy_(t.div_(self.R))
                else:
                    dist.all_reduce(b)
                    b.div_(self.R)

    def all_reduce_max(self, buf):
        """Element-wise maximum over the ranks, in place (the overflow word of the dynamic loss scale: one int32)."""
        if self.carrier == "rccl":
            from . import rccl
            self.comm.all_reduce_(buf, op=rccl.ncclMax)
        elif self.carrier in ("torch", "gloo"):
            dist.all_reduce(buf, op=dist.ReduceOp.MAX)

    def self_test(self, graph=True, agree=True):
        """Known-answer check of the three collectives on this carrier (rank r contributes r + 1), eagerly and -- for a
        capturable carrier on a GPU -- replayed from a captured graph.  Returns True when every rank agrees that every
        result is right (the verdict itself is reduced over torch.distributed; agree=False: this rank's own verdict)."""
        ok = True
        try:
            n = 4 * self.R * 256
            want = sum(range(1, self.R + 1)) / self.R

            def fill():
                return torch.full((n,), float(self.r + 1), device=self.device)

            def run(a, b):
                self.reduce_scatter_avg(a)
                lo, hi = self.shard_bounds(a)
                if self.carrier == "gloo":
                    a[lo:hi].copy_(self.shard_of(a))
                self.all_gather(a)
                self.all_reduce_avg(b)
            a, b = fill(), fill()
            run(a, b)
            # (RCCL forms the average as a pre-multiplied sum: for rank counts that are not powers of two the result may sit
            # one ulp beside (R + 1) / 2)
            right = lambda t: bool(torch.allclose(t, torch.full_like(t, want), rtol=1e-6, atol=0.0))
            ok = right(a) and right(b)
            if ok and graph and self.capturable and self.device.type == "cuda" and self.carrier == "rccl":
                a.fill_(float(self.r + 1))
                b.fill_(float(self.r + 1))
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    run(a, b)
                for _ in range(2):
                    a.fill_(float(self.r + 1))
                    b.fill_(float(self.r + 1))
                    g.replay()
                    torch.cuda.synchronize(self.device)
                    ok = ok and right(a) and right(b)
                # ... and the shape the step's exchange in two level groups has: collectives on a SECOND stream forked
                # from the capturing one, compute on both, event edges back (engine.py: _on_comm / _split_adam)
                if ok:
                    side, ev1, ev2 = torch.cuda.Stream(device=self.device), torch.cuda.Event(), torch.cuda.Event()
                    c = fill()
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                        main = torch.cuda.current_stream(self.device)
                        a.mul_(2.0)                             # "reduce b"
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            self.reduce_scatter_avg(a)
                            ev1.record(side)
                        c.mul_(3.0)                             # "reduce a", beside the collective
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            self.all_reduce_avg(c)
                            ev2.record(side)
                        main.wait_event(ev1)
                        lo, hi = self.shard_bounds(a)
                        a[lo:hi].add_(1.0)                      # "Adam b"
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            self.all_gather(a)
                        main.wait_event(ev2)
                        c.add_(1.0)
                        main.wait_stream(side)
                    for _ in range(2):
                        a.fill_(float(self.r + 1))
                        c.fill_(float(self.r + 1))
                        g2.replay()
                        torch.cuda.synchronize(self.device)
                        ok = ok and bool(torch.allclose(a, torch.full_like(a, 2.0 * want + 1.0), rtol=1e-6)) \
                            and bool(torch.allclose(c, torch.full_like(c, 3.0 * want + 1.0), rtol=1e-6))
        except Exception as e:      # noqa: BLE001 -- any failure means "do not use this carrier"
            print(f"[rank {self.r}] Exchange.self_test({self.carrier}) failed: {e}", flush=True)
            ok = False
        return all_ranks_agree(ok, self.device) if agree else ok


def all_ranks_agree(ok, device):
    """True when `ok` is true on every rank (reduced over torch.distributed's default group)."""
    if is_dist():
        on = torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")
        v = torch.tensor([1.0 if ok else 0.0], device=on)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        ok = bool(v.item() == 1.0)
    return bool(ok)


def guarded_rccl_exchange(device, timeout_s=None):
    """An Exchange over bare RCCL calls that has passed its self-test on every rank -- or None, on every rank.

    A carrier that has never seen more than one rank must not be able to hang the job it is supposed to speed up, nor to
    take torch.distributed's own communicator down with it:
      * the unique id travels through the process group's key-value STORE, on the main thread -- no collective of ours ever
        touches the default group;
      * communicator set-up and the self-test (eager and replayed from a captured graph) run in a worker thread, the
        self-test on a stream of its own, under a deadline (`NGP_RCCL_TIMEOUT`, 240 s);
      * a rank whose worker has not answered in time ABORTS its communicator if it has one (ncclCommAbort: the kernels of a
        collective that will never complete leave the GPU) and votes "no"; the vote is an all-reduce over the default
        group, which nothing of the worker's has used;
      * unless every rank votes "yes" every rank closes its communicator and the caller falls back to torch.distributed's
        collectives between graph segments -- and says why."""
    import threading
    from . import rccl
    timeout_s = float(os.environ.get("NGP_RCCL_TIMEOUT", "240")) if timeout_s is None else float(timeout_s)
    dev = torch.device(device)
    box, why = {}, None
    try:
        uid = rccl.exchange_unique_id()         # main thread, over the store
    except Exception as e:      # noqa: BLE001 -- no id, no carrier
        uid, why = None, f"unique id: {e}"

    def work():
        try:
            if dev.type == "cuda":
                torch.cuda.set_device(dev)
            x = Exchange(dev, carrier="rccl", uid=uid)
            box["x"] = x
            side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(side):
                box["ok"] = x.carrier == "rccl" and x.self_test(agree=False)
            side.synchronize()
        except Exception as e:      # noqa: BLE001 -- any failure means "do not use this carrier"
            box["ok"] = False
            box["why"] = str(e)

    if uid is not None:
        t = threading.Thread(target=work, daemon=True)
        t.start()
        t.join(timeout_s)
        if t.is_alive():
            why = f"no answer within {timeout_s:.0f} s"
            x = box.get("x")
            if x is not None:               # set up, but stuck in (or before) a collective: get its kernels off the GPU
                x.close(abort=True)
        elif not box.get("ok"):
            why = box.get("why", "self-test: wrong result")
    mine = why is None and bool(box.get("ok"))
    if why is not None:
        print(f"[rank {rank()}] direct RCCL exchange not used: {why}", flush=True)
    if all_ranks_agree(mine, dev):
        return box["x"]
    x = box.get("x")
    if x is not None and mine:              # some other rank failed: give this rank's healthy communicator back
        x.close(abort=True)
    return None
