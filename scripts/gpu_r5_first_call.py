"""where do the 516 ms of the FIRST prepare_td() go (56 ms warm)?  A fresh process, the 68 x 5000 headline array; every
library entry point and every torch allocation of prepare_td() is bracketed by device-wide synchronisations and timed on the host, for the
first and for the second call.  -> profiles/r05_prepare_td_first_call.txt"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from pta_replicator_amd import _lib, device as dv

t_import = time.perf_counter()
eng, psrs, noise = bench.build_engine(68, 5000, seed=20260921)
torch.cuda.synchronize()
out = dv.empty((8, eng.n_toa))
eng.generate(8, out=out)          # throughput mode warmed: the TD path's own first-call costs are what is left
torch.cuda.synchronize()

log = []
orig_call = _lib.call


def timed_call(name, *args):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = orig_call(name, *args)
    torch.cuda.synchronize()
    log.append((name, (time.perf_counter() - t0) * 1e3))
    return r


def run(label):
    log.clear()
    _lib.call = timed_call
    import pta_replicator_amd.engine_td as et
    et._lib.call = timed_call
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.prepare_td()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    _lib.call = orig_call
    et._lib.call = orig_call
    agg = {}
    for name, ms in log:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += ms
    return {"label": label, "total_ms": total, "entry_points": {k: {"calls": v[0], "ms": round(v[1], 3)} for k, v in agg.items()},
            "sum_entry_points_ms": round(sum(v[1] for v in agg.values()), 3)}


res = {"array": "68 x 5000", "factor_GB": 13.6}
# (a) untimed parts measured on their own first: stream creation and a first touch of fresh device memory of the factor buffer's size
torch.cuda.synchronize(); t0 = time.perf_counter()
ss = [torch.cuda.Stream() for _ in range(4)]
torch.cuda.synchronize(); res["four_hip_streams_created_by_torch_ms"] = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize(); t0 = time.perf_counter()
blk = torch.empty((int(13.6e9) // 8,), dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); res["torch_empty_13.6GB_ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); blk.fill_(0.0); torch.cuda.synchronize(); res["first_fill_of_that_block_ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); blk.fill_(0.0); torch.cuda.synchronize(); res["second_fill_ms"] = (time.perf_counter() - t0) * 1e3
del blk
res["first"] = run("first prepare_td() of the process")
res["second"] = run("second prepare_td()")
res["third"] = run("third prepare_td()")
# unsynchronised (as the product runs it)
for lab in ("fourth_unsynchronised", "fifth_unsynchronised"):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.prepare_td(); torch.cuda.synchronize()
    res[lab + "_ms"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out", "r5a"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r5a", "first_call.json"), "w") as fh:
    json.dump(res, fh, indent=1)
lines = ["prepare_td() on the 68 x 5000 array (13.6 GB of factors), fresh process, every entry point bracketed by device-wide synchronisations",
         f"  4 HIP streams created (torch): {res['four_hip_streams_created_by_torch_ms']:.1f} ms; torch.empty(13.6 GB): {res['torch_empty_13.6GB_ms']:.1f} ms; "
         f"first fill of it: {res['first_fill_of_that_block_ms']:.1f} ms; second fill: {res['second_fill_ms']:.1f} ms", ""]
names = list(res["first"]["entry_points"])
lines.append(f"{'entry point':34s} {'calls':>5s} {'first ms':>10s} {'second ms':>10s} {'third ms':>10s}")
for n in names:
    f = res["first"]["entry_points"][n]
    s2 = res["second"]["entry_points"].get(n, {"ms": float('nan')})
    s3 = res["third"]["entry_points"].get(n, {"ms": float('nan')})
    lines.append(f"{n:34s} {f['calls']:5d} {f['ms']:10.2f} {s2['ms']:10.2f} {s3['ms']:10.2f}")
lines.append(f"{'sum of entry points':34s} {'':5s} {res['first']['sum_entry_points_ms']:10.2f} {res['second']['sum_entry_points_ms']:10.2f} {res['third']['sum_entry_points_ms']:10.2f}")
lines.append(f"{'whole prepare_td (host wall)':34s} {'':5s} {res['first']['total_ms']:10.2f} {res['second']['total_ms']:10.2f} {res['third']['total_ms']:10.2f}")
lines.append(f"unsynchronised (product path): {res['fourth_unsynchronised_ms']:.2f} / {res['fifth_unsynchronised_ms']:.2f} ms")
open(os.path.join(ROOT, "gpurun_out", "r5a", "first_call.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
