"""Round 6, VERDICT r5 #3: the factorisation of the TD covariances in LEFT-LOOKING panel order (PTA_POTRF_LEFT, with and without the
run-ahead split PTA_POTRF_LEFT_SPLIT) against the right-looking look-ahead schedule (PTA_POTRF_DIAG_AHEAD), same kernels, same workspace.
Per (P, N): min of `reps` timed factorisations per variant (assembly before each, untimed), TFLOP/s, fraction of 78.6, max |L - L_right|
relative to max |L|, and run-to-run bit equality.

    python scripts/gpu_r6_potrf_left.py [P N]... > gpurun_out/r6/potrf_left.jsonl
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import configure_engine, headline_array  # noqa: E402
from pta_replicator_amd import _lib, device as dv  # noqa: E402
from pta_replicator_amd.engine import ReplicaEngine  # noqa: E402

L, S, LA = _lib.POTRF_LEFT, _lib.POTRF_LEFT_SPLIT, _lib.POTRF_DIAG_AHEAD
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(68, 5000)]
reps = int(os.environ.get("REPS", "4"))


def variants(n):
    v = [("right LA c2 (default)", LA), ("right LA c1", LA | _lib.POTRF_CHAINS(1)), ("left c2", L), ("left c1", L | _lib.POTRF_CHAINS(1)),
         ("left c3", L | _lib.POTRF_CHAINS(3)), ("left+split c2", L | S), ("left+split c1", L | S | _lib.POTRF_CHAINS(1)),
         ("left+split c3", L | S | _lib.POTRF_CHAINS(3)),
         ("left+diagLA c2", L | LA), ("left+diagLA c1", L | LA | _lib.POTRF_CHAINS(1)), ("left+diagLA c3", L | LA | _lib.POTRF_CHAINS(3)),
         ("left+diagLA c4", L | LA | _lib.POTRF_CHAINS(4))]
    if os.environ.get("SHORT"):
        return v
    for nb in (2, 3, 6, 8):
        if n > nb * 256 * 2:
            v.append((f"left c2 nb{nb * 256}", L | _lib.POTRF_NB(nb)))
            v.append((f"left+split c2 nb{nb * 256}", L | S | _lib.POTRF_NB(nb)))
            v.append((f"right LA c2 nb{nb * 256}", LA | _lib.POTRF_NB(nb)))
            v.append((f"left+diagLA c2 nb{nb * 256}", L | LA | _lib.POTRF_NB(nb)))
    return v


for P, N in shapes:
    psrs, noise = headline_array(P, N)
    eng = configure_engine(ReplicaEngine(psrs, seed=1), noise)
    eng._gw = None
    eng.prepare()
    eng.prepare_td()
    n, ld = eng.td_nst[0], eng.td_ld[0]
    s = dv.stream_ptr()
    info = dv.zeros((P,), dtype=torch.int32)
    flop = P * float(N) ** 3 / 3.0
    ref = None
    for name, fl in variants(n):
        need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, fl))
        work = dv.empty((need,))
        ts, sums = [], []
        for _ in range(reps):
            eng.td_assemble()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), fl, dv.ptr(work), need, s)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            Lm = eng.d_Ltd.view(P, n, ld)
            sums.append(float(torch.tril(Lm[0, :, :n]).sum().item()) + float(torch.tril(Lm[P - 1, :, :n]).abs().sum().item()))
        bad = int(info.abs().sum().item())
        Lm = eng.d_Ltd.view(P, n, ld)
        cur = [torch.tril(Lm[b, :, :n]).clone() for b in (0, P // 2, P - 1)]
        if ref is None:
            ref = cur
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(cur, ref))
        t = min(ts)
        print(json.dumps({"P": P, "N": N, "variant": name, "flags": hex(fl), "ms": round(t * 1e3, 3), "ms_runs": [round(x * 1e3, 2) for x in ts],
                          "TFLOPs": round(flop / t / 1e12, 2), "frac": round(flop / t / 1e12 / 78.6, 4), "info_nonzero": bad,
                          "max_rel_diff_vs_right": err, "bit_identical_runs": len(set(sums)) == 1}), flush=True)
        del work
    eng.d_Ltd = None
    del eng
    torch.cuda.empty_cache()
