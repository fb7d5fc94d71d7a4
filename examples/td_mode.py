"""The dense time-domain path on one MI355X: per-pulsar covariance -> batched fp64 Cholesky -> L.z with deviates drawn on chip.

    python examples/td_mode.py

Same array and noise model as examples/ensemble.py (the 68 pulsars / noise values of ng15_dict.json x 5000 TOAs).  The
covariances (13.6 GB in fp64) are assembled and factored once; every realisation afterwards is one triangular product per
pulsar plus the GWB drawn through the factor of its covariance on the 600-sample grid.  Same distribution as
ReplicaEngine.generate(), different deviates.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import configure_engine, headline_array
from pta_replicator_amd.engine import ReplicaEngine

psrs, noise = headline_array(68, 5000)
eng = configure_engine(ReplicaEngine(psrs, seed=2026), noise)
t0 = time.perf_counter()
eng.prepare_td()                                  # assembly + Cholesky of 68 x 5000^2 covariances, GWB grid factor
torch.cuda.synchronize()
print(f"prepare_td: {time.perf_counter() - t0:.2f} s, {eng.d_Ltd.numel() * 8 / 1e9:.1f} GB of factors resident")
t0 = time.perf_counter()
x = eng.generate_td(1024)                         # [1024, 340000] seconds, on the GPU
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"generate_td(1024): {dt * 1e3:.1f} ms = {1024 / dt:.0f} whole-array realisations/s, residual RMS {float(x.square().mean().sqrt()) * 1e6:.3f} us")
# any realisation can be re-derived on the CPU from its deviates (what the parity tests do)
d = eng.dump_draws_td(7)
print("deviates of realisation 7:", len(d["td"]), "pulsar vectors +", d["gwb"].shape, "GWB grid deviates")
# hand a few realisations to an analysis as enterprise-style pulsar objects
ens = eng.to_enterprise(x[:2])
print(ens[0][0], "residual RMS", float((ens[0][0].residuals ** 2).mean() ** 0.5))
