"""host-side timing of the legacy-stream draws of replay mode: NumPy RandomState against the native restatement (pta_legacy_randn*)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pta_replicator_amd.white_noise import _legacy_normals


def tm(f, n=20):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return round(min(ts) * 1e6, 1), round(sorted(ts)[len(ts) // 2] * 1e6, 1)


np.random.seed(1)
st = np.random.get_state()
print("get_state us (min, median)", tm(lambda: np.random.get_state(), 200))
print("set_state us", tm(lambda: np.random.set_state(st), 200))
print("numpy 2 x randn(5000) us", tm(lambda: (np.random.randn(5000), np.random.randn(5000)), 100))
print("numpy 136 x randn(3000) us", tm(lambda: [np.random.randn(3000) for _ in range(136)]))
seeds = list(range(68)); counts = [[5000, 5000]] * 68
print("68 streams x 10000 native threads us", tm(lambda: _legacy_normals(seeds, counts)))
print("68 x RandomState(seed).randn x 2 us", tm(lambda: [[np.random.RandomState(s).randn(5000) for _ in range(2)] for s in seeds]))
