"""What does HBM take for the STORE PATTERN of the walking assembly kernel?  pta_microbench kind 7: 2048 resident waves, each writing row
segments of `seg` bytes 40 KB apart (a 5000 x 5008 fp64 matrix per block, 34 blocks = 6.8 GB per launch), against the contiguous stream of
kind 2 (k_mb_write).  -> profiles/r05_write_pattern.txt"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pta_replicator_amd import _lib
res = ctypes.c_double(0.0)
out = {}
_lib.call("pta_microbench", 2, 1 << 30, 20, 0, ctypes.byref(res)); out["contiguous_stream_TBps"] = round(res.value, 3)
for lds_kb, tag in ((0, "full_occupancy"), (70, "2_workgroups_per_CU"), (150, "1_workgroup_per_CU")):
    for seg in (128, 256, 512, 1024, 2048, 4096):
        _lib.call("pta_microbench", 7, int(6.8e9), 5, seg | (lds_kb << 16), ctypes.byref(res))
        out[f"{tag}_row_segments_{seg}B_TBps"] = round(res.value, 3)
print(json.dumps(out))
