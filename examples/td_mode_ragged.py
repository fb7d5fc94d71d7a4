"""The dense time-domain path on an array whose pulsars have DIFFERENT TOA counts - every real array (ng15: 68 pulsars, 68 counts;
the reference's test_partim: 7758 / 23023 / 35037 TOAs), which the reference handles by looping over pulsars (red_noise.py:286-298).

    python examples/td_mode_ragged.py [n_psr]

All covariances are factored as ONE end-aligned schedule (pta_potrf_ragged): the matrices share the bottom-right corner of a virtual
matrix, so that at every step (one panel counted from the END) all matrices that have been reached present the same panel boundaries,
trailing size and tile grids to the kernels; a matrix enters at the step that contains its first column.  ReplicaEngine picks this
schedule whenever the counts differ (td_potrf_mode = "auto"); td_potrf_mode = "uniform" is rounds 1-3's batch-by-equal-order scheme.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import configure_engine, ragged_array, ragged_counts
from pta_replicator_amd.engine import ReplicaEngine

P = int(sys.argv[1]) if len(sys.argv) > 1 else 42
counts = ragged_counts(P)                          # log-uniform 500 ... 35 000 TOAs, shuffled; 42 pulsars: 340 915 TOAs, 48 GB of factors
psrs, noise = ragged_array(counts)
eng = configure_engine(ReplicaEngine(psrs, seed=2026), noise)
eng.prepare()
for mode in ("ragged", "uniform"):
    eng.td_potrf_mode = mode
    eng.prepare_td()                               # first call of a mode: allocations, plan
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.td_assemble(); eng.td_factorise()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:8s}: assembly + Cholesky of {P} covariances ({min(counts)} ... {max(counts)} TOAs) in {dt * 1e3:.0f} ms = "
          f"{sum(float(n) ** 3 for n in counts) / 3 / dt / 1e12:.1f} TFLOP/s")
eng.td_potrf_mode = "auto"
eng.prepare_td()
x = eng.generate_td(256)
torch.cuda.synchronize()
t0 = time.perf_counter()
x = eng.generate_td(256)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"generate_td(256): {dt * 1e3:.1f} ms = {256 / dt:.0f} whole-array realisations/s, residual RMS {float(x.square().mean().sqrt()) * 1e6:.3f} us")
