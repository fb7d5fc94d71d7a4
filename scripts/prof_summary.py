"""Summarise rocprofv3 CSV output (--output-format csv): per-kernel time table from *kernel_trace.csv, per-kernel counter sums
from *counter_collection.csv, and - with --timeline - how much of the traced span had kernels of >1 queue in flight."""
import collections, csv, glob, os, sys


def short(name, n=70):
    name = name.replace("void ", "")
    return name[:n]


def kernel_trace(path, timeline=False, out=sys.stdout):
    rows = list(csv.DictReader(open(path)))
    if not rows:
        return
    agg = collections.OrderedDict()
    for r in rows:
        k = short(r["Kernel_Name"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        a = agg.setdefault(k, [0, 0.0, r])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    print(f"== {os.path.basename(path)}: kernel, calls, total_ms, avg_ms, pct   (sum {tot:.2f} ms)", file=out)
    for k, (n, t, r) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:72s} {n:6d} {t:10.3f} {t / n:10.4f} {100 * t / tot:6.2f}", file=out)
    print("   launch geometry (first dispatch): grid, workgroup, lds, scratch, vgpr, accum_vgpr, sgpr", file=out)
    for k, (n, t, r) in agg.items():
        g = lambda *names: next((r[x] for x in names if x in r), "?")
        print(f"   {k:60s} grid=({g('Grid_Size_X')},{g('Grid_Size_Y')},{g('Grid_Size_Z')}) wg={g('Workgroup_Size_X')} lds={g('LDS_Block_Size')} "
              f"scratch={g('Scratch_Size')} vgpr={g('VGPR_Count')} agpr={g('Accum_VGPR_Count')} sgpr={g('SGPR_Count')}", file=out)
    if timeline:
        ev = []
        for r in rows:
            q = r.get("Queue_Id", "0")
            ev.append((int(r["Start_Timestamp"]), 1, q))
            ev.append((int(r["End_Timestamp"]), -1, q))
        ev.sort()
        live = collections.Counter()
        last = ev[0][0]
        busy = multi = 0
        for t, d, q in ev:
            nq = sum(1 for v in live.values() if v > 0)
            if nq >= 1:
                busy += t - last
            if nq >= 2:
                multi += t - last
            live[q] += d
            last = t
        span = ev[-1][0] - ev[0][0]
        print(f"   timeline: span {span / 1e6:.2f} ms, some kernel in flight {busy / 1e6:.2f} ms, kernels of >= 2 queues in flight {multi / 1e6:.2f} ms", file=out)
        perq = collections.defaultdict(float)
        for r in rows:
            perq[r.get("Queue_Id", "0")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        print("   per-queue kernel time (ms):", {q: round(v, 2) for q, v in perq.items()}, file=out)


def counters(path, out=sys.stdout):
    rows = csv.DictReader(open(path))
    agg = collections.OrderedDict()
    for r in rows:
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, [set(), collections.Counter()])
        a[0].add(r["Dispatch_Id"])
        a[1][r["Counter_Name"]] += float(r["Counter_Value"])
    print(f"== {os.path.basename(path)}: per-kernel counter sums over all dispatches (dispatch count in brackets)", file=out)
    for k, (disp, c) in agg.items():
        print(f"{k:72s} [{len(disp)}] " + "  ".join(f"{n}={v:.6g}" for n, v in c.items()), file=out)


if __name__ == "__main__":
    root = sys.argv[1]
    tl = "--timeline" in sys.argv
    for p in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        kernel_trace(p, tl)
    for p in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        counters(p)
