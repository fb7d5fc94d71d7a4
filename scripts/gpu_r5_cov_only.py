"""the TD covariance assembly alone, both kernels, on the 68 x 5000 headline array and on the ng15-like ragged array - the command the
round-5 rocprofv3 passes of the assembly run (scripts/gpu_r5_run1.sh)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pta_replicator_amd.engine import ReplicaEngine

res = {}
eng, psrs, noise = bench.build_engine(68, 5000, seed=20260921)
eng.prepare_td()
counts = [int(c) for c in eng.counts]
cov_bytes = 8.0 * sum(n * (n + 1) / 2 for n in counts)
KERNELS = (("walk", "walk", 0), ("tile", "tile", 0))
for kname, kern, var in KERNELS:
    eng.td_cov_walk_variant = var
    eng.td_assemble(kernel=kern)
    t = min(bench._wall(lambda: eng.td_assemble(kernel=kern)) for _ in range(8))
    res[f"uniform_68x5000_{kname}"] = {"ms": t * 1e3, "TBps_algorithmic": cov_bytes / t / 1e12}
if "--ragged" in sys.argv:
    eng.d_Ltd = None
    del eng
    torch.cuda.empty_cache()
    counts = bench.ragged_counts(42)
    psrs, noise = bench.ragged_array(counts)
    eng = bench.configure_engine(ReplicaEngine(psrs, seed=7), noise)
    eng.prepare()
    eng.prepare_td()
    cov_bytes = 8.0 * sum(n * (n + 1) / 2 for n in counts)
    for kname, kern, var in KERNELS:
        eng.td_cov_walk_variant = var
        eng.td_assemble(kernel=kern)
        t = min(bench._wall(lambda: eng.td_assemble(kernel=kern)) for _ in range(3))
        res[f"ragged_ng15like_{kname}"] = {"ms": t * 1e3, "TBps_algorithmic": cov_bytes / t / 1e12}
print(json.dumps(res))
