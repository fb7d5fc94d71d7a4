"""BASELINE.json config 5 (200 pulsars x 10000 TOAs, anisotropic l <= 4 GWB) in TD mode: 160 GB of fp64 covariance factors resident
in one MI355X's 288 GB of HBM - assembly, batched Cholesky, and realisations through L.z.  Sanity: finite, reproducible, and one
pulsar's L.z against NumPy/LAPACK on the dumped deviates."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pta_replicator_amd.engine import ReplicaEngine
from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
P, N, lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 10000, 4
rng = np.random.default_rng(200)
raj, decj = rng.uniform(0, 24, P), np.degrees(np.arcsin(rng.uniform(-1, 1, P)))
psrs = []
for a in range(P):
    psr = SimulatedPulsar(toas=ArrayTOAs(np.sort(rng.uniform(53000, 60305, N)), 0.5), name=f"J{a:04d}", loc={"RAJ": raj[a], "DECJ": decj[a]})
    make_ideal(psr); psrs.append(psr)
crng = np.random.default_rng(200)
clm = np.concatenate([[np.sqrt(4 * np.pi)], 0.03 * crng.standard_normal(24)])
eng = ReplicaEngine(psrs, seed=5)
eng.set_white_noise(efac=1.0, log10_equad=-6.5)
eng.set_jitter(log10_ecorr=-6.6, coarsegrain=0.1)
eng.set_red_noise(-14.0, 3.0)
eng.set_gwb(-14.6733, 13. / 3., clm=clm, lmax=lmax)
res = {"P": P, "N": N}
def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return time.perf_counter() - t0, r
res["prepare_s"], _ = wall(eng.prepare)
res["prepare_td_first_s"], _ = wall(eng.prepare_td)      # first touch of 160 GB
res["prepare_td_s"], _ = wall(eng.prepare_td)
res["factor_GB"] = eng.d_Ltd.numel() * 8 / 1e9
res["hbm_allocated_GB"] = torch.cuda.memory_allocated() / 1e9
flop_chol = P * N ** 3 / 3.0
for R in (64, 512):
    t, out = wall(lambda: eng.generate_td(R))
    t, out = wall(lambda: eng.generate_td(R))
    res[f"generate_td_R{R}"] = {"ms": t * 1e3, "realisations_per_s": R / t, "useful_TFLOPs": P * N * N * R / t / 1e12}
assert bool(torch.isfinite(out).all())
assert torch.equal(eng.generate_td(2, r0=3), out[3:5])
# one pulsar against NumPy / LAPACK
a = 123 % P
L = eng.td_factor(a).cpu().numpy()
z = eng.dump_draws_td(1)
n = int(eng.counts[a])
eng0 = None
part = L @ z["td"][a]
# subtract the GWB term by regenerating without it is expensive at this size: compare the factor instead (backward error) and L.z via replay
err_back = None
import ctypes
from pta_replicator_amd import _lib, device as dv
s = dv.stream_ptr()
o = int(eng.off[a])
Cd = dv.zeros((n, n))
phi = (eng.d_amp ** 2).contiguous(); ec2 = (eng.d_ecorr_toa ** 2).contiguous()
_lib.call("pta_td_cov_assemble", ctypes.c_void_p(eng.d_Ft.data_ptr() + 8 * o), eng.n_toa, n, eng.K, ctypes.c_void_p(phi.data_ptr() + 8 * a * eng.K),
          ctypes.c_void_p(eng._td_sigma2.data_ptr() + 8 * o), ctypes.c_void_p(eng.d_epoch_of.data_ptr() + 4 * o), ctypes.c_void_p(ec2.data_ptr() + 8 * o),
          dv.ptr(Cd), n, s)
C = Cd.cpu().numpy(); C = np.tril(C) + np.tril(C, -1).T
Lref = np.linalg.cholesky(C)
res["factor_vs_lapack_rel"] = float(np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)))
res["TFLOPs_of_prepare_td_incl_assembly"] = flop_chol / res["prepare_td_s"] / 1e12
print(json.dumps(res))
