"""Generate ``tests/golden/c5_orf_lmax4.npz``: the anisotropic ORF of BASELINE.json config 5 (200 pulsars, l <= 4) computed by
the UNMODIFIED reference module ``/root/reference/pta_replicator/spharmORFbasis.py``.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs ``/root/reference``; ~15 CPU-minutes, spread over the cores):

    python oracle/gen_c5_orf.py

``spharmORFbasis.correlated_basis`` (:385-434) is a pure-Python quadruple loop: 20 100 pairs x 25 (l, m) modes take the
reference ~13 minutes on one core, far beyond what a GPU-box test may spend on the CPU side.  The loop body is therefore run
here once - pair by pair with the reference's own ``calczeta`` / ``arbCompFrame_ORF`` / ``real_rotated_Gammas`` in the order
``correlated_basis`` calls them (a row of pairs per worker process; ``check_against_correlated_basis`` verifies on a 12-pulsar
subset that this reproduces ``correlated_basis`` itself bit for bit) - and the test compares the device ORF, its Cholesky factor
and whole config-5 realisations with the result.

Fixture contents: geometry (``raj`` [h], ``decj`` [deg], seed 200 as SURVEY.md §8d specifies), ``clm`` (isotropic term + 10 %
perturbations, seed 200, redrawn until the ORF is positive definite - the draw index is recorded), ``orf`` = 2 sum_k clm_k
basis_k (red_noise.py:224-226) and ``orf_l[l]`` = the same sum restricted to degree l (pins every degree of the basis at full P).
"""
import importlib.util
import os
import sys
from multiprocessing import Pool

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(_HERE), "tests", "golden", "c5_orf_lmax4.npz")
REF = os.environ.get("PTA_REFERENCE_ROOT", "/root/reference")
P, LMAX, SEED = 200, 4, 200


def _ref():
    spec = importlib.util.spec_from_file_location("ref_spharmORFbasis", os.path.join(REF, "pta_replicator", "spharmORFbasis.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def geometry():
    rng = np.random.default_rng(SEED)
    raj, decj = rng.uniform(0, 24, P), np.degrees(np.arcsin(rng.uniform(-1, 1, P)))
    locs = np.zeros((P, 2))
    locs[:, 0] = raj * np.pi / 12.0                      # red_noise.py:206
    locs[:, 1] = np.pi / 2.0 - decj * np.pi / 180.0      # :207, :223
    return raj, decj, locs


def _row(args):
    """row aa of every basis matrix, pairs (aa, bb >= aa): the loop body of correlated_basis (:399-430)."""
    aa, locs = args
    ref = _ref()
    nb = (LMAX + 1) ** 2
    out = np.zeros((nb, len(locs)))
    for ll in range(LMAX + 1):
        for bb in range(aa, len(locs)):
            plus, neg = [], []
            for mm in range(ll + 1):
                zeta = ref.calczeta(locs[:, 0][aa], locs[:, 0][bb], locs[:, 1][aa], locs[:, 1][bb])
                g = ref.arbCompFrame_ORF(mm, ll, zeta)
                plus.append(g)
                neg.append((-1) ** (mm) * g)
            gamma_ml = neg[1:][::-1] + plus
            for mm in range(2 * ll + 1):
                out[ll * ll + mm, bb] = ref.real_rotated_Gammas(mm - ll, ll, locs[:, 0][aa], locs[:, 0][bb], locs[:, 1][aa],
                                                                locs[:, 1][bb], gamma_ml)
    return aa, out


def basis_rows(locs, procs=None):
    n = len(locs)
    basis = np.zeros(((LMAX + 1) ** 2, n, n))
    with Pool(procs or os.cpu_count()) as pool:
        for aa, row in pool.imap_unordered(_row, [(aa, locs) for aa in range(n)], chunksize=1):
            basis[:, aa, aa:] = row[:, aa:]
            basis[:, aa:, aa] = row[:, aa:]
    return basis


def check_against_correlated_basis(locs):
    sub = locs[:12]
    want = np.array(_ref().correlated_basis(sub, LMAX))
    got = basis_rows(sub, procs=2)
    assert np.array_equal(want, got), "row-wise evaluation differs from correlated_basis"


def main():
    raj, decj, locs = geometry()
    check_against_correlated_basis(locs)
    basis = basis_rows(locs)
    crng = np.random.default_rng(SEED)
    for draw in range(50):
        clm = np.concatenate([[np.sqrt(4 * np.pi)], 0.1 * crng.standard_normal((LMAX + 1) ** 2 - 1)])
        orf = 2 * np.tensordot(clm, basis, axes=1)
        if np.all(np.linalg.eigvalsh(orf) > 1e-6):
            break
    else:
        raise SystemExit("no positive-definite anisotropic ORF found")
    orf_l = np.stack([2 * np.tensordot(clm[ll * ll:(ll + 1) ** 2], basis[ll * ll:(ll + 1) ** 2], axes=1) for ll in range(LMAX + 1)])
    np.savez_compressed(OUT, raj=raj, decj=decj, clm=clm, clm_draw=np.int64(draw), orf=orf, orf_l=orf_l, lmax=np.int64(LMAX))
    print("wrote", OUT, os.path.getsize(OUT), "bytes; min eig", float(np.linalg.eigvalsh(orf).min()), "clm draw", draw)


if __name__ == "__main__":
    main()
