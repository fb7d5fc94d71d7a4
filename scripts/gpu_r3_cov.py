"""covariance-assembly timing probe (68 x 5000^2 lower triangles, one launch)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import _lib, device as dv
eng, psrs, noise = build_engine(68, 5000, seed=1)
eng.prepare_td()
counts = [int(c) for c in eng.counts]
s = dv.stream_ptr()
phi = (eng.d_amp ** 2).contiguous(); ec2 = (eng.d_ecorr_toa ** 2).contiguous()
def assemble():
    _lib.call("pta_td_cov_assemble_all", dv.ptr(eng.d_Ft), eng.n_toa, eng.K, dv.ptr(phi), dv.ptr(eng._td_sigma2), dv.ptr(eng.d_epoch_of), dv.ptr(ec2),
              dv.ptr(eng.d_Ltd), *[dv.ptr(x) for x in eng._td_layout], eng.P, max(counts), s)
assemble(); torch.cuda.synchronize()
ref = eng.d_Ltd[:5000 * eng.td_ld[0]].clone()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); assemble(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
same = bool(torch.equal(ref, eng.d_Ltd[:5000 * eng.td_ld[0]]))
print(json.dumps({"stagger": os.environ.get("PTA_TUNE_COV_STAGGER"), "ms": [round(t, 3) for t in ts], "GBps": 8.0 * sum(n * (n + 64) / 2 for n in counts) / (min(ts) * 1e-3) / 1e9, "same": same}))
