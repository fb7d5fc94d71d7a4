"""Root cause of round 4's "NaN until a device-wide synchronisation" (DESIGN.md §4.2), reproduced on demand.

The withdrawn column-walking assembly kernel (commit ee73ef5, kept verbatim in scripts/probe_src/nan_repro_r4_walk.hip) loads the A fragments of
its LAST k-step from rows k = 4 * 14 + (lane >> 4) = 56 .. 59 of the [K, N] design matrix without clamping k, relying on "a masked k meets b = 0".
For K = 58 (components = 29 - the one failing case of the round-4 test) rows 58 and 59 lie BEHIND the matrix.  0 x finite = 0, but
0 x NaN = NaN: the result depends on what the caching allocator last left behind d_Ft - in the failing test sequence a recycled
NaN-poisoned factor buffer (the test's own d_Ltd.fill_(nan) clones).  A synchronisation changes the allocation history, not the ordering.

Here the design matrix is made a VIEW into a slab whose tail (the 2 N doubles behind it) is set explicitly:
    tail = NaN  -> the round-4 kernel returns NaN for K = 58, finite for K = 60 / 64;   tail = 0 -> finite everywhere;
    the round-5 kernel (pta_td_cov_assemble_walk: every k clamped before it forms an address) and the tile kernel: finite in all cases.
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pta_replicator_amd import device as dv
from pta_replicator_amd.engine import ReplicaEngine
from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal

probe = ctypes.CDLL(os.path.join(ROOT, "scripts", "probe_src", "libnan_repro_r4_walk.so"))


def build(components):
    rng = np.random.default_rng(components)
    psrs = []
    for a, n in enumerate((777, 90, 1025, 2601)):
        ep = np.sort(rng.uniform(53000, 56000, n // 3 + 1))
        mjd = (ep[:, None] + rng.uniform(0, 0.01, (len(ep), 3))).ravel()[:n]
        p = SimulatedPulsar(toas=ArrayTOAs(mjd, rng.uniform(0.3, 1.5, n)), name=f"J{a:04d}", loc={"RAJ": 2.0 + 3 * a, "DECJ": -20.0 + 25 * a})
        make_ideal(p)
        psrs.append(p)
    eng = ReplicaEngine(psrs, seed=3)
    eng.set_white_noise(efac=1.1, log10_equad=-6.3)
    eng.set_jitter(log10_ecorr=-6.5, coarsegrain=0.1)
    eng.set_red_noise([-13.6, None, -14.0, -13.2], [3.1, None, 4.2, 2.2], components=components)
    eng.prepare().prepare_td()
    return eng


def lower_nans(eng):
    bad = 0
    for a in range(eng.P):
        n, ld, pos = int(eng.counts[a]), eng.td_ld[a], int(eng.td_pos[a])
        v = eng.d_Ltd[pos:pos + n * ld].view(n, ld)[:, :n]
        bad += int((~torch.isfinite(torch.tril(v))).sum().item())
    return bad


out = []
for components in (29, 30, 32):
    eng = build(components)
    K, N = eng.plan.rn_k, eng.n_toa
    Ft = eng.d_Ft.clone()
    for tail in ("nan", "zero"):
        slab = torch.full((K * N + 2 * N + 64,), float("nan") if tail == "nan" else 0.0, dtype=torch.float64, device="cuda")
        slab[:K * N] = Ft.reshape(-1)
        eng.d_Ft = slab[:K * N].view(K, N)
        row = {"components": components, "K": K, "tail_behind_Ft": tail}
        phi = (eng.d_amp ** 2).contiguous()
        ecorr2 = (eng.d_ecorr_toa ** 2).contiguous()
        for name in ("r4_walk", "r5_walk", "tile"):
            eng.d_Ltd.fill_(0.0)
            if name == "r4_walk":
                rc = probe.nan_repro_r4_walk(dv.ptr(eng.d_Ft), ctypes.c_int64(N), K, dv.ptr(phi), dv.ptr(eng._td_sigma2), dv.ptr(eng.d_epoch_of), dv.ptr(ecorr2),
                                             dv.ptr(eng.d_Ltd), *[dv.ptr(x) for x in eng._td_layout], eng.P, int(max(eng.counts)), dv.stream_ptr())
                assert rc == 0
            else:
                eng.td_assemble(kernel="walk" if name == "r5_walk" else "tile")
            torch.cuda.synchronize()
            row[name + "_nonfinite_in_lower_triangles"] = lower_nans(eng)
        out.append(row)
        print(json.dumps(row), flush=True)
ok = all((r["r4_walk_nonfinite_in_lower_triangles"] > 0) == (r["K"] == 58 and r["tail_behind_Ft"] == "nan") for r in out) and \
    all(r["r5_walk_nonfinite_in_lower_triangles"] == 0 and r["tile_nonfinite_in_lower_triangles"] == 0 for r in out)
print("ROOT CAUSE " + ("CONFIRMED" if ok else "NOT CONFIRMED") + ": the round-4 kernel turns NaN exactly when K = 58 and the bytes behind Ft are NaN")
