"""does running two half batches on two streams (chirp-z of one beside the fused synthesis of the other) beat one batch?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import device as dv
e1, psrs, noise = build_engine(68, 5000, seed=1)
e2, _, _ = build_engine(68, 5000, seed=1)
R, K = 1024, 10
out = dv.empty((R, e1.n_toa))
def serial():
    for i in range(K): e1.generate(R, r0=i * R, out=out)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def dual(h=R // 2):
    for i in range(K):
        with torch.cuda.stream(s1): e1.generate(h, r0=i * R, out=out[:h])
        with torch.cuda.stream(s2): e2.generate(R - h, r0=i * R + h, out=out[h:])
def wall(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
res = {"serial_ms_per_1024": wall(serial), "dual_stream_ms_per_1024": wall(dual), "serial_again": wall(serial)}
def quad():
    ss = [s1, s2]
    for i in range(K):
        for j in range(4):
            e = (e1, e2)[j & 1]
            with torch.cuda.stream(ss[j & 1]): e.generate(R // 4, r0=i * R + j * (R // 4), out=out[j * (R // 4):(j + 1) * (R // 4)])
res["dual_stream_quarters_ms_per_1024"] = wall(quad)
print(json.dumps(res))
