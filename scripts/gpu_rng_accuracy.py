"""ulp accuracy of the device Box-Muller deviates against an 80-bit longdouble evaluation of the same uniforms."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pta_replicator_amd import _lib, device as dv
from oracle import philox_ref

seed, real, stream, npairs = 77, 5, philox_ref.stream_id(3, 11), 1 << 20
z = dv.empty((2 * npairs,))
_lib.call("pta_rng_fill_normal", seed, real, 1, stream, npairs, 1, dv.ptr(z), None, 2 * npairs, 0, dv.stream_ptr())
got = z.cpu().numpy()
u1, u2 = philox_ref.uniform_pairs(seed, real, stream, npairs)
L = np.longdouble
rad = np.sqrt(L(-2) * np.log(u1.astype(L)))
q = np.rint(4 * u2)
x = L(2) * np.pi.__class__(np.pi) * 0 + (u2.astype(L) - L(0.25) * q.astype(L)) * (L(2) * np.arccos(L(-1)))   # 2 pi in longdouble
sr, cr = np.sin(x), np.cos(x)
k = q.astype(np.int64) & 3
s = np.choose(k, [sr, cr, -sr, -cr]); c = np.choose(k, [cr, -sr, -cr, sr])
ref0, ref1 = (rad * c), (rad * s)
def ulps(g, r):
    r64 = r.astype(np.float64)
    return np.abs((g.astype(L) - r) / np.spacing(np.abs(r64)).astype(L))
e0, e1 = ulps(got[0::2], ref0), ulps(got[1::2], ref1)
# near the zeros of cos/sin the deviate is tiny and absolute error matters: also report absolute error relative to rad
a0 = np.abs(got[0::2].astype(L) - ref0) / rad; a1 = np.abs(got[1::2].astype(L) - ref1) / rad
print(json.dumps({"pairs": npairs, "max_ulp_z0": float(e0.max()), "max_ulp_z1": float(e1.max()), "p999_ulp": float(np.quantile(np.maximum(e0, e1).astype(np.float64), 0.999)),
                  "max_abs_over_rad": float(max(a0.max(), a1.max())), "mean": float(got.mean()), "var": float(got.var())}))
