"""Multi-GPU: realisations shard embarrassingly, one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for the tests).

There is no exchange step inside the path: every rank owns a contiguous range of the global
realisation index and draws from the same counter-based stream, so the ensemble is bit-identical for any number
of GPUs.  The only communication is the final gather of the residual arrays to rank 0 named by BASELINE.json's
north_star.  Rank 0's ingress (7 xGMI links) bounds it - 39 GB at config 4, >= 37 ms - so
``generate_gathered`` pipelines it: every rank generates chunk c+1 on its compute stream while chunk c travels, and
rank 0 receives each chunk STRAIGHT into its rows of the final [total, n_toa] tensor (point-to-point receives into
row slices: no per-rank staging buffers, no concatenation - rank 0 holds the ensemble once).
"""
import ctypes
import os

import torch
import torch.distributed as dist


def shard_range(total, rank=None, world=None):
    """[start, stop) of the realisation indices owned by `rank`: contiguous, sizes differ by at most one."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def rng_mode_consistent(engine):
    """every rank must run the same Gaussian-transform mode, or realisation r would depend on which rank drew it
    (the mode is per-engine state: ReplicaEngine.rng_fast).  Raises if the ranks disagree."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    dev = getattr(engine, "comm_device", None) or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu")
    v = int(getattr(engine, "rng_fast", 0))
    t = torch.tensor([v, -v], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0].item()) != -int(t[1].item()):
        raise RuntimeError("ranks disagree on the RNG math mode (engine.rng_fast): the ensemble would not be reproducible")
    return True


def generate_sharded(engine, total, r0=0, td=False):
    """this rank's shard of realisations r0 .. r0+total-1 -> (local tensor [R_loc, n_toa], (start, stop))."""
    rng_mode_consistent(engine)
    start, stop = shard_range(total)
    gen = engine.generate_td if td else engine.generate
    return gen(stop - start, r0=r0 + start), (start, stop)


class _Side:
    """a side stream for the communication calls when the tensors live on a GPU; a no-op on CPU (gloo tests)."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None

    def after_main(self):
        """communication issued from now on starts after everything queued on the current (compute) stream."""
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)

    def __enter__(self):
        if self.cuda:
            self._ctx = torch.cuda.stream(self.stream)
            self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.cuda:
            self._ctx.__exit__(*a)


def generate_gathered(engine, total, r0=0, chunk=256, dst=0, td=False, generate=None, n_cols=None, device=None, dtype=torch.float64):
    """All `total` realisations r0 .. r0+total-1 on rank `dst` as one [total, n_toa] tensor (None on the other ranks),
    generated shard-wise on every rank and gathered chunk by chunk while the next chunk is being generated.

    `generate(n, r0, out)` defaults to engine.generate / engine.generate_td; passing a callable (plus n_cols / device)
    lets the CPU tests drive the pipeline without a GPU."""
    gen = generate or (engine.generate_td if td else engine.generate)
    n_cols = n_cols if n_cols is not None else engine.n_toa
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if not dist.is_initialized() or dist.get_world_size() == 1:
        out = torch.empty((total, n_cols), dtype=dtype, device=device)
        for lo in range(0, total, chunk):
            n = min(chunk, total - lo)
            gen(n, r0=r0 + lo, out=out[lo:lo + n])
        return out
    if engine is not None:
        rng_mode_consistent(engine)
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world) for r in range(world)]
    start, stop = sizes[rank]
    rows = stop - start
    nchunks = max((b - a + chunk - 1) // chunk for a, b in sizes)
    side = _Side(device)
    pending = []
    if rank == dst:
        result = torch.empty((total, n_cols), dtype=dtype, device=device)
        # `result` comes from the caching allocator of the MAIN stream: kernels queued there earlier (a previous call's) may
        # still be using the block, so the receives - issued from the side stream - start only after them
        side.after_main()
        for c in range(nchunks):
            lo = c * chunk
            ops = []
            for src, (a, b) in enumerate(sizes):
                n = min(chunk, (b - a) - lo)
                if src != dst and n > 0:
                    ops.append(dist.P2POp(dist.irecv, result[a + lo:a + lo + n], src))
            if ops:
                with side:                                   # receives run beside this rank's own generation
                    pending += dist.batch_isend_irecv(ops)
            n = min(chunk, rows - lo)
            if n > 0:
                gen(n, r0=r0 + start + lo, out=result[start + lo:start + lo + n])
        for req in pending:
            req.wait()
        return result
    bufs = [torch.empty((min(chunk, max(rows, 1)), n_cols), dtype=dtype, device=device) for _ in range(2)]
    inflight = [None, None]
    for c in range(nchunks):
        lo = c * chunk
        n = min(chunk, rows - lo)
        if n <= 0:
            break
        b = c & 1
        if inflight[b] is not None:                          # the buffer's previous chunk must have left before it is refilled
            for req in inflight[b]:
                req.wait()
        gen(n, r0=r0 + start + lo, out=bufs[b][:n])
        side.after_main()
        with side:
            inflight[b] = dist.batch_isend_irecv([dist.P2POp(dist.isend, bufs[b][:n], dst)])
    for reqs in inflight:
        for req in reqs or []:
            req.wait()
    return None


def gather_to_rank0(local, total=None, dst=0):
    """Gather row-sharded [R_loc, n] tensors (shard_range layout) to rank `dst` in global row order.  Returns the
    [total, n] tensor on `dst`, None elsewhere.  Point-to-point: `dst` receives every shard directly into its rows of
    the result, so it holds the ensemble once (plus nothing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if total is None:
        t = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(t)
        total = int(t.item())
    sizes = [shard_range(total, r, world) for r in range(world)]
    assert local.shape[0] == sizes[rank][1] - sizes[rank][0], "local shard does not match shard_range()"
    local = local.contiguous()
    if rank != dst:
        if local.shape[0]:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst)]):
                req.wait()
        return None
    result = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    ops = [dist.P2POp(dist.irecv, result[a:b], src) for src, (a, b) in enumerate(sizes) if src != dst and b > a]
    reqs = dist.batch_isend_irecv(ops) if ops else []
    result[sizes[dst][0]:sizes[dst][1]] = local
    for req in reqs:
        req.wait()
    return result


# ---------------------------------------------------------------------------------------------------------------------------
# the same gather through the C ABI (pta_gather_rank0): for integrators that bind the library with ctypes and do not want
# torch.distributed on the data path.  The communicator is a plain ncclComm_t created on the librccl.so PyTorch ships.
# ---------------------------------------------------------------------------------------------------------------------------
class RcclComm:
    """An RCCL communicator (ncclComm_t) created through ctypes.  ``unique_id()`` on one rank, hand the 128 bytes to the others by
    any channel (``from_process_group`` uses torch.distributed's object broadcast), then ``RcclComm(rank, world, uid)`` on every
    rank with its GPU current."""

    class _Uid(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    @staticmethod
    def _lib():
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return ctypes.CDLL(path if os.path.exists(path) else "librccl.so")

    @classmethod
    def unique_id(cls):
        uid = cls._Uid()
        rc = cls._lib().ncclGetUniqueId(ctypes.byref(uid))
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId failed ({rc})")
        return ctypes.string_at(ctypes.addressof(uid), 128)

    def __init__(self, rank, world, uid_bytes):
        self.rank, self.world = int(rank), int(world)
        self._rccl = self._lib()
        uid = self._Uid()
        ctypes.memmove(ctypes.addressof(uid), uid_bytes, 128)
        self._comm = ctypes.c_void_p()
        self._rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, self._Uid, ctypes.c_int]
        rc = self._rccl.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed ({rc})")

    @classmethod
    def from_process_group(cls):
        """one communicator per rank of the initialised torch.distributed job (the unique id travels by broadcast_object_list)."""
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0])

    @property
    def ptr(self):
        return self._comm

    def destroy(self):
        if self._comm:
            self._rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            self._rccl.ncclCommDestroy(self._comm)
            self._comm = None


def gather_to_rank0_abi(local, total, comm=None, dst=0):
    """gather_to_rank0 through pta_gather_rank0 (include/pta_replicator_amd.h): ``local`` is this rank's [shard, n] float64 device
    tensor in shard_range() layout; returns the [total, n] tensor on ``dst``, None elsewhere.  ``comm``: an RcclComm (not needed for a
    single rank)."""
    from . import _lib, device as dv
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    a, b = shard_range(total, rank, world)
    assert local.shape[0] == b - a and local.dtype == torch.float64
    local = local.contiguous()
    n = int(local.shape[1])
    out = torch.empty((total, n), dtype=torch.float64, device=local.device) if rank == dst else None
    _lib.call("pta_gather_rank0", comm.ptr if comm is not None else None, rank, world, dst, dv.ptr(local) if b > a else None, int(total), n, n,
              dv.ptr(out) if out is not None else None, n, dv.stream_ptr())
    return out
