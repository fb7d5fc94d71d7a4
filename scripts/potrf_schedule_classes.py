"""Kernel classes of ONE uniform-batch factorisation (pta_potrf_batched_ws) from a rocprofv3 kernel trace, for any panel order: the tile
products grouped by what their grid says they are, the diagonal-phase kernels, span and busy time.
usage: potrf_schedule_classes.py <trace dir> [nb=<panel width>]"""
import collections, csv, glob, sys

root = sys.argv[1]
NB = next((int(a[3:]) for a in sys.argv[2:] if a.startswith("nb=")), 1024)
PT = NB // 128
rows = []
for p in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cov = [i for i, r in enumerate(rows) if "k_td_cov" in r["Kernel_Name"]]
names = ("k_diag128", "k_ws_strips", "k_dgemm_glds128", "k_dgemm_mfma", "k_potf2", "k_trsm", "k_syrk64", "k_inv_blocks")
if not cov:
    sys.exit("no assembly launch found in the trace")
seg = [r for r in rows[cov[-1] + 1:] if any(n in r["Kernel_Name"] for n in names)]
if not seg:
    sys.exit("no factorisation kernels after the last assembly")


def cls(r):
    k = r["Kernel_Name"]
    gx, gy = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"])
    if "k_diag128" in k:
        return "diagonal phase: k_diag128 (128-column base case: pivot sweeps + inverse)"
    if "k_ws_strips" in k:
        return "diagonal phase: k_ws_strips"
    if "k_dgemm_glds128" in k:
        tri = PT * (PT + 1) // 2
        if gy == 1 and gx > tri:
            m = int(((8 * gx + 1) ** 0.5 - 1) / 2)
            if m * (m + 1) // 2 == gx:
                return "RIGHT-looking trailing update (triangular grid, K = panel width)"
            if (gx - tri) % PT == 0:
                return "LEFT-looking block-column update U(q) (trapezoid grid: diagonal block + rows below, K = all columns to the left)"
            return "tile product on a 1-D grid (other)"
        if gy == 1 and gx == PT * (PT + 1) // 2:
            return "diagonal block of the next panel (triangular grid of %d tiles: U1 / left-looking last block column)" % gx
        if gx == 1 and gy > PT:
            return "substitution on the rows below the panel (one column tile per launch, K = 128 .. panel width)"
        if gx >= PT - 1 and gy > PT:
            return "block-column update (gx = panel tiles, gy = rows below: LEFT-looking U(q), K = columns to the left; right-looking U2a, K = panel width)"
        return "diagonal phase: 128-tile products of the recursion"
    if "k_dgemm_mfma" in k:
        return "diagonal phase: 64-tile products"
    return "last panel (64-column recursion: " + k.replace("void ", "")[:16] + ")"


agg = collections.OrderedDict()
for r in seg:
    a = agg.setdefault(cls(r), [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
t0 = int(seg[0]["Start_Timestamp"])
span = (max(int(r["End_Timestamp"]) for r in seg) - t0) / 1e6
# time with at least one big tile product (>= 0.3 ms) in flight, and time with none (the exposed diagonal phases / launch chains)
ev = []
for r in seg:
    if "k_dgemm_glds128" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 300000:
        ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy, depth, last = 0, 0, None
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
tot = sum(a[1] for a in agg.values())
queues = collections.Counter(r.get("Queue_Id", "0") for r in seg)
print(f"# {len(seg)} dispatches on {len(queues)} queues, span {span:.2f} ms, sum of durations {tot:.2f} ms; a tile product of >= 0.3 ms in flight for {busy / 1e6:.2f} ms, none for {span - busy / 1e6:.2f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:10.2f} ms {n:6d} dispatches  {k}")
