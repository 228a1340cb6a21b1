"""RCCL, called directly: the collectives of the data-parallel step as plain `nccl*` calls on the caller's HIP stream.

Why not torch.distributed for these: a ProcessGroupNCCL collective brings its own stream hops and event bookkeeping, and
inside a hipGraph capture that bookkeeping is the fragile part.  A bare ncclReduceScatter / ncclAllGather on the capture
stream is one kernel node of the step graph like any other, so a group of training steps -- collectives included --
replays from ONE graph launch (raw_ngp_amd.nerf.engine, `dp_exchange = "rccl"`).  torch.distributed stays what it is good
at: rendezvous (the unique id travels through the process group's key-value store), barriers, the CPU/gloo rehearsal.

The library is the librccl.so PyTorch already links (same symbols, one copy in the process).  Everything here is in place:
  reduce_scatter(buf): every rank passes its full-length buffer, rank r ends with the reduction of shard r AT shard r
  all_gather(buf):     rank r's shard r is published into every rank's buffer
which is what RCCL treats as its in-place forms (recvbuff == sendbuff + rank * count).
"""
import ctypes
import os

import torch

NCCL_UNIQUE_ID_BYTES = 128                     # rccl.h:40
ncclSum, ncclMax, ncclAvg = 0, 2, 4            # rccl.h: ncclRedOp_t
_DTYPES = {torch.float32: 7, torch.bfloat16: 9, torch.float16: 6, torch.int32: 2}   # rccl.h: ncclDataType_t


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("NGP_RCCL_LIB") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = ctypes.CDLL(path)
        vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclGetErrorString.argtypes = [i]
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), i, _UniqueId, i]
        L.ncclCommDestroy.argtypes = [vp]
        L.ncclCommAbort.argtypes = [vp]
        L.ncclReduceScatter.argtypes = [vp, vp, sz, i, i, vp, vp]         # send, recv, recvcount, dtype, op, comm, stream
        L.ncclAllGather.argtypes = [vp, vp, sz, i, vp, vp]                # send, recv, sendcount, dtype, comm, stream
        L.ncclAllReduce.argtypes = [vp, vp, sz, i, i, vp, vp]             # send, recv, count, dtype, op, comm, stream
        L.ncclGroupStart.argtypes = []
        L.ncclGroupEnd.argtypes = []
        for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclReduceScatter", "ncclAllGather",
                     "ncclAllReduce", "ncclGroupStart", "ncclGroupEnd"):
            getattr(L, name).restype = i
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"RCCL: {what} failed: {lib().ncclGetErrorString(rc).decode()}")


_uid_round = [0]


def exchange_unique_id():
    """The communicator's unique id, created on rank 0 and handed to the others through torch.distributed's key-value STORE
    (TCP; the rendezvous the process group itself was built over) -- not through a collective: whatever happens to a
    communicator that is being set up, the default process group has seen no traffic of ours and stays usable (for the
    vote in parallel.guarded_rccl_exchange, for the fallback carrier).  Call on every rank, from the main thread."""
    import torch.distributed as dist
    assert dist.is_initialized(), "rccl: initialise torch.distributed first (its store carries the unique id)"
    store = dist.distributed_c10d._get_default_store()
    key = f"raw_ngp_amd/rccl_uid/{_uid_round[0]}"
    _uid_round[0] += 1
    if dist.get_rank() == 0:
        uid = _UniqueId()
        _check(lib().ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        store.set(key, bytes(uid))
        return bytes(uid)
    return bytes(store.get(key))            # blocks until rank 0 has set it (the store's own timeout applies)


class Communicator:
    """One RCCL communicator over the ranks of torch.distributed's default group (one process per GPU)."""

    def __init__(self, device, uid=None):
        import torch.distributed as dist
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        raw = exchange_unique_id() if uid is None else uid
        assert len(raw) == NCCL_UNIQUE_ID_BYTES
        u = _UniqueId()
        ctypes.memmove(ctypes.byref(u), raw, NCCL_UNIQUE_ID_BYTES)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib().ncclCommInitRank(ctypes.byref(self.comm), self.world, u, self.rank), "ncclCommInitRank")

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _shard(self, buf):
        if not (buf.is_cuda and buf.is_contiguous() and buf.dim() == 1) or buf.dtype not in _DTYPES:
            raise RuntimeError("rccl: flat contiguous device tensor of a supported dtype expected")
        if buf.numel() % self.world:
            raise RuntimeError("rccl: buffer length must be a multiple of the number of ranks")
        n = buf.numel() // self.world
        return n, buf.data_ptr(), buf.data_ptr() + self.rank * n * buf.element_size()

    def _op(self, op):
        # one rank: the average of one contribution is the contribution.  RCCL implements ncclAvg as a pre-multiplied sum,
        # which on a one-rank communicator is a pass of its own over the buffer (oneRankReduce<FuncPreMulSum>: 50 us for the
        # 48.8 MB table gradient) -- work that exists on no multi-rank communicator, where the factor is folded into the
        # reduction's first load.  ncclSum in place is the identity and launches nothing.
        return ncclSum if (op == ncclAvg and self.world == 1) else op

    def reduce_scatter_(self, buf, op=ncclAvg):
        """In place: afterwards buf[rank * n : (rank + 1) * n] holds the reduction of that shard over all ranks."""
        op = self._op(op)
        n, base, mine = self._shard(buf)
        with torch.cuda.device(self.device):
            _check(lib().ncclReduceScatter(base, mine, n, _DTYPES[buf.dtype], op, self.comm, self._stream()), "ncclReduceScatter")

    def all_gather_(self, buf):
        """In place: every rank's shard buf[rank * n : (rank + 1) * n] is published into every rank's buf."""
        n, base, mine = self._shard(buf)
        with torch.cuda.device(self.device):
            _check(lib().ncclAllGather(mine, base, n, _DTYPES[buf.dtype], self.comm, self._stream()), "ncclAllGather")

    def all_reduce_(self, *bufs, op=ncclAvg):
        """In place; several buffers go out as one group (one launch)."""
        L = lib()
        op = self._op(op)
        with torch.cuda.device(self.device):
            if len(bufs) > 1:
                _check(L.ncclGroupStart(), "ncclGroupStart")
            for b in bufs:
                if not (b.is_cuda and b.is_contiguous()) or b.dtype not in _DTYPES:
                    raise RuntimeError("rccl: contiguous device tensor of a supported dtype expected")
                _check(L.ncclAllReduce(b.data_ptr(), b.data_ptr(), b.numel(), _DTYPES[b.dtype], op, self.comm, self._stream()),
                       "ncclAllReduce")
            if len(bufs) > 1:
                _check(L.ncclGroupEnd(), "ncclGroupEnd")

    def destroy(self):
        if self.comm:
            lib().ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()

    def abort(self):
        """Tear the communicator down without waiting for its outstanding work (ncclCommAbort: kernels of a collective that
        will never complete leave the GPU) -- what a rank does with a carrier that missed its deadline."""
        if self.comm:
            lib().ncclCommAbort(self.comm)
            self.comm = ctypes.c_void_p()
