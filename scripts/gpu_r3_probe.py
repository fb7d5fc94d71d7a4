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
                            ("ws_la_1chain", _lib.POTRF_DIAG_AHEAD | _lib.POTRF_CHAINS(1), 1), ("ws_la_2chains_diag64", _lib.POTRF_DIAG_AHEAD | _lib.POTRF_DIAG64, 1), ("ws_1chain_diag64", _lib.POTRF_NO_LOOKAHEAD | _lib.POTRF_DIAG64, 1),  ("ws_la_2chains", _lib.POTRF_DIAG_AHEAD, 1), ("ws_la_3chains", _lib.POTRF_DIAG_AHEAD | _lib.POTRF_CHAINS(3), 1), ("ws_3chains", _lib.POTRF_CHAINS(3), 1), ("ws_4chains", _lib.POTRF_CHAINS(4), 1),
                            ("ws_la_2chains_nb512", _lib.POTRF_NB(2) | _lib.POTRF_DIAG_AHEAD, 2), ("ws_la_2chains_nb768", _lib.POTRF_NB(3) | _lib.POTRF_DIAG_AHEAD, 2), ("ws_la_2chains_nb1280", _lib.POTRF_NB(5) | _lib.POTRF_DIAG_AHEAD, 2), ("ws_la_2chains_again", _lib.POTRF_DIAG_AHEAD, 2), ("ws_2chains_nb512", _lib.POTRF_NB(2), 2), ("ws_2chains_nb1536", _lib.POTRF_NB(6), 2), ("ws_2chains_nb2048", _lib.POTRF_NB(8), 2),
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

if "solve" in what:
    # the blocked substitution's products: rows x 128 tiles in ONE column of tiles, K = 128 .. 896 - rate against the batch (tile count vs the
    # chip's 512 workgroup slots: 31 x 64 = 1984 = 3.9 rounds, 31 x 68 = 2108 = 4.1, 31 x 66 = 2046 = 4.0)
    M, ld = 3968, 5008
    for b in (64, 66, 68, 82):
        A = torch.randn((b, M + 1024, ld), dtype=torch.float64, device="cuda")
        row = {"rows": M, "N": 128, "batch": b, "tiles": 31 * b}
        for K in (128, 256, 512, 896):
            t = wall(lambda: _lib.call("pta_dgemm", 1, M, 128, K, ctypes.c_double(-1.0), ctypes.c_void_p(A.data_ptr() + 8 * 1024 * ld), ld, 1, dv.ptr(A), ld,
                                       ctypes.c_double(1.0), ctypes.c_void_p(A.data_ptr() + 8 * (1024 * ld + K)), ld, 0, b, (M + 1024) * ld, (M + 1024) * ld,
                                       (M + 1024) * ld, 2, s), reps=5)
            row[f"K{K}_TF"] = round(2.0 * M * 128 * K * b / t / 1e12, 2)
            row[f"K{K}_us"] = round(t * 1e6, 1)
        print(json.dumps(row), flush=True)
        del A

if "potrfloop" in what or "gemmloop" in what:
    # a few seconds of back-to-back work for scripts/gpu_r3_clocks.sh (rocm-smi samples beside it)
    from bench import build_engine
    eng, psrs, noise = build_engine(68, 5000, seed=1)
    eng.prepare_td()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 6.0
    n_it = 0
    if "gemmloop" in what:
        M, K, b = 3968, 1024, 68
        A = torch.randn((b, M, M + K), dtype=torch.float64, device="cuda")
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            for _ in range(10):
                _lib.call("pta_dgemm", 1, M, M, K, ctypes.c_double(-1.0), dv.ptr(A), M + K, 1, dv.ptr(A), M + K, ctypes.c_double(0.0),
                          ctypes.c_void_p(A.data_ptr() + 8 * K), M + K, 1, b, M * (M + K), M * (M + K), M * (M + K), 2, s)
            torch.cuda.synchronize()
            n_it += 1
            if n_it % 5 == 1:
                print("gemm TF", round(10 * M * M * K * b / (time.perf_counter() - t0) / 1e12, 2), flush=True)
    else:
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            eng.prepare_td()
            torch.cuda.synchronize()
            n_it += 1
            if n_it % 10 == 1:
                print("prepare_td ms", round((time.perf_counter() - t0) * 1e3, 2), flush=True)

if "ldsweep" in what:
    # the trailing update of the first panel (rows 3968, lower, beta = 1) against the leading dimension and K (1024 / 1032 = the in-situ first panel)
    M, b = 3968, 68
    for ld in (5000, 5008, 5024, 5056, 5072, 5120, 5136):
        A = torch.randn((b, 5000, ld), dtype=torch.float64, device="cuda")
        row = {"ld": ld}
        for K in (1024, 1032):
            t = wall(lambda: _lib.call("pta_dgemm", 1, M, M, K, ctypes.c_double(-1.0), ctypes.c_void_p(A.data_ptr() + 8 * K * ld), ld, 1,
                                       ctypes.c_void_p(A.data_ptr() + 8 * K * ld), ld, ctypes.c_double(1.0), ctypes.c_void_p(A.data_ptr() + 8 * (K * ld + K)), ld, 1, b,
                                       5000 * ld, 5000 * ld, 5000 * ld, 2, s), reps=3)
            A.normal_()
            row[f"K{K}_TF"] = round(1.0 * M * M * K * b / t / 1e12, 2)
        print(json.dumps(row), flush=True)
        del A
