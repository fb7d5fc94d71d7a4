"""TD mode at the headline size (68 x 5000^2): assembly, factorisation (with / without look-ahead), generate_td throughput."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import configure_engine, headline_array
from pta_replicator_amd import _lib, device as dv
from pta_replicator_amd.engine import ReplicaEngine

P = int(sys.argv[1]) if len(sys.argv) > 1 else 68
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
Rs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 1024]
psrs, noise = headline_array(P, N)
eng = configure_engine(ReplicaEngine(psrs, seed=1), noise)
eng.prepare()

def wall(fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

res = {"P": P, "N": N}
t0 = time.perf_counter(); eng.prepare_td(); torch.cuda.synchronize(); res["prepare_td_first_s"] = time.perf_counter() - t0
res["prepare_td_s"] = wall(lambda: eng.prepare_td())
res["prepare_td_nolook_s"] = wall(lambda: eng.prepare_td(lookahead=False))
# factorisation alone on a copy of the assembled covariances
s = dv.stream_ptr()
n, ld = N, eng.td_ld[0]
def assemble():
    phi = (eng.d_amp ** 2).contiguous(); ec2 = (eng.d_ecorr_toa ** 2).contiguous()
    _lib.call("pta_td_cov_assemble_all", dv.ptr(eng.d_Ft), eng.n_toa, eng.plan.rn_k, dv.ptr(phi), dv.ptr(eng._td_sigma2), dv.ptr(eng.d_epoch_of), dv.ptr(ec2),
              dv.ptr(eng.d_Ltd), *[dv.ptr(x) for x in eng._td_layout], P, N, s)
    return phi, ec2
info = dv.zeros((P,), dtype=torch.int32)
for name, flags in (("potrf_default_ms", 0), ("potrf_1chain_ms", _lib.POTRF_NO_LOOKAHEAD), ("potrf_nb768_ms", _lib.POTRF_NB(3)),
                    ("potrf_nb1280_ms", _lib.POTRF_NB(5)), ("potrf_3chains_ms", _lib.POTRF_CHAINS(3)), ("potrf_4chains_ms", _lib.POTRF_CHAINS(4))):
    ts = []
    for rep in range(3):
        assemble(); 
        ts.append(wall(lambda: _lib.call("pta_potrf_batched_ex", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), flags, s)))
    res[name] = [round(t * 1e3, 2) for t in ts]
    res[name.replace("_ms", "_TFLOPs")] = round(P * n ** 3 / 3.0 / min(ts) / 1e12, 2)
res["cov_assemble_ms"] = round(wall(assemble, 2) * 1e3, 2)
assert int(info.abs().sum().item()) == 0
flop = float(sum(int(c) ** 2 for c in eng.counts))     # useful flops per realisation: sum N_a^2
for R in Rs:
    out = dv.empty((R, eng.n_toa))
    eng.generate_td(R, out=out)
    t = wall(lambda: eng.generate_td(R, out=out), 2)
    res[f"generate_td_R{R}"] = {"ms": round(t * 1e3, 2), "realisations_per_s": round(R / t, 1), "useful_TFLOPs": round(flop * R / t / 1e12, 2)}
    eng._gw_saved = None
print(json.dumps(res))
