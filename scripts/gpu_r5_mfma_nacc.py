"""How many independent accumulators does a wave need to keep the fp64 matrix pipe of gfx950 busy?  pta_microbench kind 8: NACC
accumulators per wave (shared A operand, own B operand each), W waves per SIMD.  -> profiles/r05_mfma_nacc.txt"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pta_replicator_amd import _lib
res = ctypes.c_double(0.0)
rows = []
print("TFLOP/s (peak 78.6): rows = accumulators per wave, columns = waves per SIMD")
print("nacc " + " ".join(f"{w:>7d}" for w in (1, 2, 3, 4, 8)))
for na in (1, 2, 4, 8, 12, 16):
    line = []
    for w in (1, 2, 3, 4, 8):
        _lib.call("pta_microbench", 8, w, 4000, na, ctypes.byref(res))
        line.append(res.value)
    rows.append((na, line))
    print(f"{na:4d} " + " ".join(f"{v:7.1f}" for v in line))
