"""world_size-2 (and 3) CPU tests of the multi-GPU plumbing over gloo: realisation sharding and the gather of the
residual arrays to rank 0 (the only collective of the path).  The HIP generator is replaced by a
deterministic function of the global realisation index, which is exactly the property the counter-based RNG gives
the real engine (tests/test_gpu_parity.py::test_engine_throughput_mode_equals_replay_of_its_own_draws)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pta_replicator_amd.distributed import gather_to_rank0, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeEngine:
    """stands in for ReplicaEngine on a machine without a GPU (the product refuses to run there): same generate(R, r0, out)
    contract, values a pure function of the global realisation index.  The real engine goes through the very same functions
    under a process group in tests/test_gpu_parity.py::test_engine_through_a_process_group (nccl, -m gpu)."""
    n_toa = 37
    rng_fast = 0
    calls = None

    def generate(self, R, r0=0, out=None):
        r = torch.arange(r0, r0 + R, dtype=torch.float64)[:, None]
        i = torch.arange(self.n_toa, dtype=torch.float64)[None, :]
        v = torch.sin(r * 1.7 + i * 0.3) + r
        if self.calls is not None:
            self.calls.append((r0, R))
        if out is None:
            return v
        out.copy_(v)
        return out


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pta_replicator_amd.distributed import generate_sharded
        eng = _FakeEngine()
        local, (a, b) = generate_sharded(eng, total, r0=100)
        assert (a, b) == shard_range(total, rank, world) and local.shape[0] == b - a
        full = gather_to_rank0(local, total)
        if rank == 0:
            q.put(full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def _worker_pipelined(rank, world, port, total, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pta_replicator_amd.distributed import generate_gathered
        eng = _FakeEngine()
        eng.calls = []
        full = generate_gathered(eng, total, r0=100, chunk=chunk, device="cpu")
        a, b = shard_range(total, rank, world)
        # this rank generated exactly its own shard, in chunks of at most `chunk` rows, in order
        assert eng.calls == [(100 + a + lo, min(chunk, b - a - lo)) for lo in range(0, b - a, chunk)]
        if rank == 0:
            q.put(full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total,chunk", [(2, 10, 2), (2, 7, 3), (3, 8, 1), (3, 2, 4), (2, 33, 5)])
def test_chunked_overlapped_gather(world, total, chunk):
    """generate_gathered: every rank generates its shard chunk by chunk while the previous chunk travels; rank 0 receives
    straight into the final tensor.  Ragged shards (sizes differ by one, ranks with fewer chunks, a rank with NO rows)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, world, port, total, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _FakeEngine().generate(total, r0=100).numpy()
    assert full.shape == ref.shape and np.array_equal(full, ref)


@pytest.mark.parametrize("world,total", [(2, 10), (2, 7), (3, 8)])
def test_sharded_generation_and_gather(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _FakeEngine().generate(total, r0=100).numpy()
    assert full.shape == ref.shape and np.array_equal(full, ref)  # identical to the 1-process ensemble, in order


def test_shard_range_partitions():
    for total in (0, 1, 7, 16384):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
