"""Gaussian generation rate (microbench kind 4) of the table-driven fp64 transform against the polynomial one it replaced and
the fp32 fast mode, at 4 and 8 waves per SIMD."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pta_replicator_amd import _lib
res = {}
for bpc in (4, 8):
    for name, opt in (("table_fp64", 0), ("poly_fp64", 2), ("fast_fp32", 1)):
        out = ctypes.c_double()
        _lib.call("pta_microbench", 4, bpc, 256, opt, ctypes.byref(out))
        res[f"{name}_{bpc}waves_Tnormals_per_s"] = round(out.value, 4)
print(json.dumps(res))
