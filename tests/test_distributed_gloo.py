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
    n_toa = 37

    def generate(self, R, r0=0):
        r = torch.arange(r0, r0 + R, dtype=torch.float64)[:, None]
        i = torch.arange(self.n_toa, dtype=torch.float64)[None, :]
        return torch.sin(r * 1.7 + i * 0.3) + r


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
