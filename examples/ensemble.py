"""Batched ensembles on one MI355X (or sharded over several): 68 synthetic pulsars x 5000 TOAs, HD GWB + RN + white noise.

    python examples/ensemble.py                         # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/ensemble.py   # eight
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import configure_engine, headline_array                      # synthetic array with the NANOGrav 15-yr noise dictionary's shape
from pta_replicator_amd.distributed import generate_sharded
from pta_replicator_amd.engine import ReplicaEngine

world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    torch.distributed.init_process_group("nccl")

psrs, noise = headline_array(68, 5000)
eng = ReplicaEngine(psrs, seed=2026)
configure_engine(eng, noise)   # per-backend EFAC / EQUAD / ECORR, per-pulsar RN, HD GWB: values of ng15_dict.json
eng.prepare()

# this rank's share of 4096 realisations, resident on its GPU: [n_local, 340000] seconds
local, (lo, hi) = generate_sharded(eng, 4096)
rms = float(local.square().mean().sqrt())
print(f"rank {os.environ.get('RANK', '0')}: realisations {lo}..{hi - 1}, residual RMS {rms * 1e6:.3f} us")

# the same realisations as NumPy arrays, streamed over PCIe while the next chunk is generated
if lo == 0:
    n = 0
    for first, block in eng.stream_to_host(960, chunk=240):
        n += len(block)
    print("streamed", n, "realisations to the host")

# write one realisation back into the pulsar objects (one TOA shift per pulsar, like a sequence of add_* calls would leave them)
eng.inject(lo)
print("pulsar 0 residual RMS after inject:", float((psrs[0].residuals.resids_value ** 2).mean() ** 0.5))

if world > 1:
    torch.distributed.destroy_process_group()
