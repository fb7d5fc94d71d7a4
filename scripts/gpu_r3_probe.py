"""Round-3 A/B probe (one gpurun call): register-staged vs LDS-DMA-staged 128 x 128-tile fp64 GEMM over the Cholesky update shapes,
then the whole 68 x 5000^2 batched factorisation with either kernel, chains 1 / 2 / 4."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pta_replicator_amd import _lib, device as dv
s = dv.stream_ptr()
what = sys.argv[1:] or ["gemm", "potrf"]


def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if "gemm" in what:
    for (M, K, b, lower) in [(3968, 1024, 34, 1), (3968, 512, 34, 1), (3968, 256, 34, 1), (3968, 128, 34, 1), (4096, 4096, 16, 0), (2048, 1024, 34, 1)]:
        ld = M + K
        A = torch.randn((b, M, ld), dtype=torch.float64, device="cuda")
        row = {"M": M, "K": K, "batch": b, "lower": lower}
        for algo in (1, 2):
            for beta in (1.0, 0.0):
                t = wall(lambda: _lib.call("pta_dgemm", 1, M, M, K, ctypes.c_double(-1.0), dv.ptr(A), ld, 1, dv.ptr(A), ld, ctypes.c_double(beta),
                                           ctypes.c_void_p(A.data_ptr() + 8 * K), ld, lower, b, M * ld, M * ld, M * ld, algo, s))
                A.normal_()
                row[f"algo{algo}_beta{int(beta)}_TF"] = round(2.0 * M * M * K * b * (0.5 if lower else 1.0) / t / 1e12, 2)
        print(json.dumps(row), flush=True)
        del A

if "potrf" in what:
    from bench import build_engine
    eng, psrs, noise = build_engine(68, 5000, seed=1)
    eng.prepare_td()
    counts = [int(c) for c in eng.counts]
    n, ld, P = counts[0], eng.td_ld[0], eng.P
    phi = (eng.d_amp ** 2).contiguous(); ec2 = (eng.d_ecorr_toa ** 2).contiguous()
    info = dv.zeros((P,), dtype=torch.int32)

    def assemble():
        _lib.call("pta_td_cov_assemble_all", dv.ptr(eng.d_Ft), eng.n_toa, eng.K, dv.ptr(phi), dv.ptr(eng._td_sigma2), dv.ptr(eng.d_epoch_of), dv.ptr(ec2),
                  dv.ptr(eng.d_Ltd), *[dv.ptr(x) for x in eng._td_layout], eng.P, max(counts), s)
    flop = sum(c ** 3 for c in counts) / 3.0
    ref = None
    RS = _lib.POTRF_REG_STAGING
    need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, 0))
    work = dv.empty((need,))
    work.fill_(float("nan"))
    for name, flags, ws in (("reg_2chains", RS, 0), ("glds_2chains", 0, 0), ("ws_1chain", _lib.POTRF_NO_LOOKAHEAD, 1), ("ws_2chains", 0, 1),
                            ("ws_2chains_lockstep", _lib.POTRF_LOCKSTEP, 1), ("ws_3chains", _lib.POTRF_CHAINS(3), 1), ("ws_4chains", _lib.POTRF_CHAINS(4), 1),
                            ("ws_2chains_nb512", _lib.POTRF_NB(2), 2), ("ws_2chains_nb1536", _lib.POTRF_NB(6), 2), ("ws_2chains_nb2048", _lib.POTRF_NB(8), 2),
                            ("glds_2chains_nb2048", _lib.POTRF_NB(8), 0), ("glds_2chains_nb1536", _lib.POTRF_NB(6), 0)):
        if ws == 2:
            need2 = int(_lib.lib.pta_potrf_workspace_doubles(n, P, flags))
            work = dv.empty((need2,)); need = need2
        ts = []
        for _ in range(2):
            assemble(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), flags, dv.ptr(work) if ws else None, need if ws else 0, s)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        L0 = eng.d_Ltd[:n * ld].view(n, ld)[:, :n].tril().clone()
        if ref is None:
            ref = L0
        err = float((L0 - ref).abs().max() / ref.abs().max())
        print(json.dumps({"potrf": name, "ms": [round(t * 1e3, 2) for t in ts], "TFLOPs": round(flop / min(ts) / 1e12, 2), "info_bad": int(info.abs().sum().item()),
                          "max_rel_diff_vs_first": err}), flush=True)
