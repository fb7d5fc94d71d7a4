# round 6: ragged schedule - 1024- against 2048-column panels, 1 / 2 chains, by batch size (arrays of large matrices)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python - <<'PY'
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, ctypes
from pta_replicator_amd import _lib, device as dv
def run(orders, flags, reps=2):
    B = len(orders)
    n = np.array(orders, dtype=np.int32); ld = ((n.astype(np.int64) + 15) // 16 * 16)
    off = np.concatenate([[0], np.cumsum(n.astype(np.int64) * ld)])[:-1].astype(np.int64)
    tot = int(off[-1] + n[-1] * ld[-1])
    A0 = dv.zeros((tot,))
    # SPD by construction: diag-dominant lower triangles (cheap to build on the device): a_ij = 1/(1+|i-j|) scaled + big diagonal
    for b in range(B):
        v = A0[int(off[b]):int(off[b]) + int(n[b]) * int(ld[b])].view(int(n[b]), int(ld[b]))
        i = torch.arange(int(n[b]), device="cuda", dtype=torch.float64)
        blk = 4096
        for r0 in range(0, int(n[b]), blk):
            r1 = min(int(n[b]), r0 + blk)
            v[r0:r1, :int(n[b])] = 1.0 / (1.0 + (i[r0:r1, None] - i[None, :]).abs()) ** 1.5
        v.diagonal().add_(8.0)
    words = int(_lib.lib.pta_potrf_ragged_plan_words(B)); plan = np.zeros(words, dtype=np.int64); need = ctypes.c_int64(0)
    _lib.call("pta_potrf_ragged_plan", dv.hptr(n), dv.hptr(off), dv.hptr(ld), B, flags, dv.hptr(plan), ctypes.byref(need))
    work = dv.empty((need.value,)); info = dv.zeros((B,), dtype=torch.int32); plan_d = dv.i64(plan)
    ts = []
    for _ in range(reps):
        A = A0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.call("pta_potrf_ragged", dv.ptr(A), dv.hptr(plan), dv.ptr(plan_d), dv.ptr(info), dv.ptr(work), need.value, dv.stream_ptr())
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        assert int(info.abs().sum().item()) == 0
        del A
    fl = sum(float(x) ** 3 for x in orders) / 3
    return min(ts) * 1e3, fl / min(ts) / 1e12
sets = {"config2 (3)": [7758, 23022, 35036], "3 x 20000": [20000] * 3, "6 large": [35000, 30000, 26000, 22000, 18000, 14000], "12 large": [int(x) // 2 * 2 for x in np.linspace(12000, 34000, 12)],
        "24 mid": [int(x) // 2 * 2 for x in np.linspace(6000, 24000, 24)]}
for name, orders in sets.items():
    for tag, fl in (("nb1024 c2", _lib.POTRF_NB(4)), ("nb2048 c2", _lib.POTRF_NB(8)), ("nb1024 c1", _lib.POTRF_NB(4) | _lib.POTRF_CHAINS(1)), ("nb2048 c1", _lib.POTRF_NB(8) | _lib.POTRF_CHAINS(1)),
                    ("nb1536 c2", _lib.POTRF_NB(6))):
        ms, tf = run(orders, fl)
        print(f"{name:14s} {tag:10s} {ms:9.2f} ms {tf:6.2f} TFLOP/s {tf/78.6:.3f}", flush=True)
PY
