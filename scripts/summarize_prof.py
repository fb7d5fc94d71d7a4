"""Condense rocprofv3 (rocpd sqlite) outputs into the small text summaries committed under profiles/.

    python scripts/summarize_prof.py gpurun_out/prof > profiles/rNN_rocprofv3_summary.txt
"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


for db_path in sorted(glob.glob(os.path.join(root, "*", "*.db"))):
    tag = os.path.basename(os.path.dirname(db_path))
    cur = sqlite3.connect(db_path).cursor()
    if tag.startswith("trace"):
        print(f"== rocprofv3 --kernel-trace --stats  ({tag}): kernel, calls, total_ms, avg_ms, pct")
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print(f"{short(name):70s} {calls:6d} {total / 1e3:12.1f} {avg / 1e3:10.2f} {pct:6.2f}")
        print("   launch geometry (first dispatch of each kernel): grid, workgroup, lds, scratch, vgpr, agpr, sgpr")
        seen = set()
        for row in cur.execute("select name,grid_x,grid_y,grid_z,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels"):
            if row[0] in seen:
                continue
            seen.add(row[0])
            print(f"   {short(row[0]):66s} grid=({row[1]},{row[2]},{row[3]}) wg={row[4]} lds={row[5]} scratch={row[6]} vgpr={row[7]} agpr={row[8]} sgpr={row[9]}")
    else:
        print(f"== rocprofv3 --pmc ({tag}): kernel, counter, dispatches, mean value per dispatch")
        q = "select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"
        for name, cname, n, mean in cur.execute(q):
            if "k_" not in name:
                continue
            print(f"{short(name):70s} {cname:26s} {n:5d} {mean:.6g}")
    print()
