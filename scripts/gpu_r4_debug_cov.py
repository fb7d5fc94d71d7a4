"""debug: K = 58 column-walking assembly followed by the ragged factorisation reported 'leading minor 1' in the test - which part?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pta_replicator_amd.engine import ReplicaEngine
from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
components = int(sys.argv[1]) if len(sys.argv) > 1 else 29
rng = np.random.default_rng(components)
psrs = []
for a, n in enumerate((777, 90, 1025, 2601)):
    ep = np.sort(rng.uniform(53000, 56000, n // 3 + 1))
    mjd = (ep[:, None] + rng.uniform(0, 0.01, (len(ep), 3))).ravel()[:n]
    p = SimulatedPulsar(toas=ArrayTOAs(mjd, rng.uniform(0.3, 1.5, n)), name=f"J{a:04d}", loc={"RAJ": 2.0 + 3 * a, "DECJ": -20.0 + 25 * a})
    make_ideal(p); psrs.append(p)
eng = ReplicaEngine(psrs, seed=3)
eng.set_white_noise(efac=1.1, log10_equad=-6.3)
eng.set_jitter(log10_ecorr=-6.5, coarsegrain=0.1)
eng.set_red_noise([-13.6, None, -14.0, -13.2], [3.1, None, 4.2, 2.2], components=components)
eng.prepare().prepare_td()
print("first prepare_td ok")
for fill in ("zero", "nan"):
    for variant in (1, 2):
        for mode in ("ragged", "uniform"):
            eng.td_cov_variant = variant
            eng.d_Ltd.fill_(float("nan") if fill == "nan" else 0.0)
            eng.td_assemble()
            torch.cuda.synchronize()
            full = eng.d_Ltd.clone()
            ok = "ok"
            try:
                eng.td_factorise(mode=mode)
            except Exception as e:
                ok = str(e)[-60:]
            print(fill, "variant", variant, mode, "->", ok, flush=True)
            if variant == 2 and mode == "ragged":
                v2 = full
            if variant == 1 and mode == "ragged":
                v1 = full
    same = torch.equal(torch.nan_to_num(v1, nan=12345.0) == 12345.0, torch.nan_to_num(v2, nan=12345.0) == 12345.0)
    d = (torch.nan_to_num(v1, nan=0.0) - torch.nan_to_num(v2, nan=0.0)).abs().max().item()
    print(fill, "same NaN pattern over the WHOLE buffer:", same, "max |v1 - v2|:", d, flush=True)
    if not same:
        bad = torch.nonzero((torch.nan_to_num(v1, nan=12345.0) == 12345.0) != (torch.nan_to_num(v2, nan=12345.0) == 12345.0)).flatten()[:10].cpu().numpy()
        for x in bad:
            a = int(np.searchsorted(eng.td_pos, x, side="right") - 1)
            r, c = divmod(int(x - eng.td_pos[a]), eng.td_ld[a])
            print("   differs at pulsar", a, "row", r, "col", c, "n", int(eng.counts[a]), "v1", v1[x].item(), "v2", v2[x].item())
print("--- the test's flow: clones alive, variant 2, prepare_td() again")
got = {}
for variant in (1, 2):
    eng.td_cov_variant = variant
    eng.d_Ltd.fill_(float("nan"))
    eng.td_assemble()
    got[variant] = eng.d_Ltd.clone()
eng.td_cov_variant = 2
import pta_replicator_amd.engine_td as et
orig = eng.td_factorise
def spy(**kw):
    if os.environ.get("NOSYNC"):
        return orig(**kw)
    torch.cuda.synchronize()
    for a in range(eng.P):
        n, ld, pos = int(eng.counts[a]), eng.td_ld[a], int(eng.td_pos[a])
        v = eng.d_Ltd[pos:pos + eng.td_nst[a] * ld].view(eng.td_nst[a], ld)
        lo = torch.tril(v[:, :eng.td_nst[a]])
        print("  before factorise: pulsar", a, "C[0,0]", v[0, 0].item(), "lower finite:", bool(torch.isfinite(lo).all()), "diag min", torch.diagonal(v[:, :eng.td_nst[a]]).min().item(), flush=True)
    ref = got[2]
    print("  same as the clone assembled by variant 2:", torch.equal(torch.nan_to_num(eng.d_Ltd, nan=7.0), torch.nan_to_num(ref, nan=7.0)) if ref.shape == eng.d_Ltd.shape else "shape differs", flush=True)
    return orig(**kw)
eng.td_factorise = spy
try:
    eng.prepare_td()
    print("prepare_td ok")
except Exception as e:
    print("prepare_td FAILED:", str(e)[-80:])
    torch.cuda.synchronize()
    for a in range(eng.P):
        n, ld, pos = int(eng.counts[a]), eng.td_ld[a], int(eng.td_pos[a])
        v = eng.d_Ltd[pos:pos + eng.td_nst[a] * ld].view(eng.td_nst[a], ld)
        lo = torch.tril(v[:, :eng.td_nst[a]])
        print("  after failure: pulsar", a, "L[0,0]", v[0, 0].item(), "lower finite:", bool(torch.isfinite(lo).all()), "n non-finite", int((~torch.isfinite(lo)).sum().item()))
    # assemble again into a separate buffer and factor with LAPACK to see whether the input was fine
    eng.td_assemble(); torch.cuda.synchronize()
    v = eng.d_Ltd[0:eng.td_nst[0] * eng.td_ld[0]].view(eng.td_nst[0], eng.td_ld[0])[:, :eng.td_nst[0]].cpu().numpy()
    C = np.tril(v) + np.tril(v, -1).T
    print("  LAPACK on the re-assembled J0000:", "ok" if np.all(np.isfinite(np.linalg.cholesky(C))) else "bad", "C[0,0]", C[0, 0])
