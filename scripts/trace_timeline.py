"""print the last N dispatches of a rocprofv3 kernel trace as a timeline (ms from the first of them): start, duration, queue, kernel, grid"""
import csv, glob, sys
root, n = sys.argv[1], int(sys.argv[2])
rows = []
for p in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    k = r["Kernel_Name"].replace("void ", "")
    k = k[:k.index("(")] if "(" in k else k
    gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:8.3f} q{r['Queue_Id']} {k[:44]:44s} {gx}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
