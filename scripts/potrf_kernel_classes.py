"""Kernel classes of ONE ragged factorisation (pta_potrf_ragged) from a rocprofv3 kernel trace: where the time of the end-aligned
schedule goes.  usage: potrf_kernel_classes.py <trace dir> [first|last]  (which factorisation of the run to take: default last)"""
import collections, csv, glob, sys

root = sys.argv[1]
PT = next((int(a[3:]) for a in sys.argv[2:] if a.startswith("pt=")), 16)   # 128-column tiles per panel: 16 = 2048-column panels (the default for large matrices), 8 = 1024
rows = []
for p in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one factorisation = the dispatches between two hipMemset-of-info markers is not visible here; take the run of potrf kernels that
# follows the LAST assembly launch (td_assemble: k_td_cov_walk / k_td_cov128) of the trace
cov = [i for i, r in enumerate(rows) if "k_td_cov" in r["Kernel_Name"]]   # either assembly kernel (walking / tile)
names = ("k_diag128", "k_ws_strips", "k_dgemm_glds128", "k_dgemm_mfma<", "k_potf2", "k_trsm", "k_syrk64", "k_inv_blocks")
if not cov:
    sys.exit("no assembly launch found in the trace")
pick = cov[0] if "first" in sys.argv[2:] else cov[-1]
seg = []
for r in rows[pick + 1:]:
    if any(n in r["Kernel_Name"] for n in names):
        seg.append(r)
    elif seg and "k_td_cov" in r["Kernel_Name"]:
        break
if not seg:
    sys.exit("no factorisation kernels after the last assembly")


def cls(r):
    k = r["Kernel_Name"]
    gx, gy, gz = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
    if "k_diag128" in k:
        return "diagonal phase: k_diag128 (128-column base case: two pivot sweeps + products + inverse)"
    if "k_ws_strips" in k:
        return "diagonal phase: k_ws_strips (strips [-W L | W])"
    if "k_dgemm_mfma<" in k:
        return "diagonal phase: 64-tile products (rows below a group, < 256 rows)"
    if "k_dgemm_glds128" in k:
        u1 = PT * (PT + 1) // 2
        if gy == 1 and gx > u1:
            return f"trailing updates (lower-triangular tile products, K = {128 * PT})"
        if gy == 1 and gx == u1:
            return f"next panel's diagonal block of the trailing update (U1, {u1} tiles per matrix)"
        if gx == PT and gy >= PT:
            return "trailing updates: sub-diagonal rectangle of the next panel (U2a)"
        if gx == 1 and gy >= PT:
            return f"substitution on the rows below the panel (one product per 128-column block, K = 128 .. {128 * PT})"
        return "diagonal phase: 128-tile products of the recursion"
    return "other potrf kernels (" + k.replace("void ", "")[:24] + ")"


agg = collections.OrderedDict()
for r in seg:
    a = agg.setdefault(cls(r), [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
tot = sum(a[1] for a in agg.values())
queues = collections.Counter(r.get("Queue_Id", "0") for r in seg)
print(f"# {len(seg)} dispatches on {len(queues)} queues, span {span:.2f} ms, sum of durations {tot:.2f} ms (chains and look-ahead overlap: sum > span)")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:10.2f} ms {n:6d} dispatches  {k}")
