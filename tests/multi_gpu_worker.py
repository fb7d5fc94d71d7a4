"""Worker of tests/test_gpu_multi.py (one process per GPU, launched by torch.distributed.run): the REAL engine on every rank, the
pipelined generate + gather, the plain gather and the C-ABI gather over RCCL; rank 0 saves what it received."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import configure_engine, headline_array  # noqa: E402
from pta_replicator_amd.distributed import (RcclComm, gather_to_rank0, gather_to_rank0_abi, generate_gathered, generate_sharded,  # noqa: E402
                                            rng_mode_consistent)
from pta_replicator_amd.engine import ReplicaEngine  # noqa: E402


def main():
    out_dir, total = sys.argv[1], int(sys.argv[2])
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    psrs, noise = headline_array(5, 700)
    eng = configure_engine(ReplicaEngine(psrs, seed=31), noise).prepare()
    assert rng_mode_consistent(eng)
    ones = torch.ones(1, device="cuda")
    dist.all_reduce(ones)
    local, (a, b) = generate_sharded(eng, total, r0=3)
    full_a = gather_to_rank0(local, total)
    full_b = generate_gathered(eng, total, r0=3, chunk=4)            # chunk 4: ranks with fewer chunks, ragged last chunk
    comm = RcclComm.from_process_group()
    full_c = gather_to_rank0_abi(local, total, comm)
    torch.cuda.synchronize()
    eng.prepare_td()
    td_local, _ = generate_sharded(eng, total, r0=3, td=True)
    full_d = gather_to_rank0(td_local, total)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(os.path.join(out_dir, "rank0.npz"), ranks_seen=int(ones.item()), a=full_a.cpu().numpy(), b=full_b.cpu().numpy(),
                 c=full_c.cpu().numpy(), d=full_d.cpu().numpy())
    comm.destroy()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
