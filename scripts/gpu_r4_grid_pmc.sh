# counters behind the grid's small-N cells: VALU wave-instructions per output element of the fused kernel at N = 122 / 1000 / 5000 (68 pulsars)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/grid_pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/run -o g -- python -c "
import bench
for n in (122, 1000, 5000):
    c = bench.grid_cell(68, n, td=False)
    print(n, c['throughput']['realisations_per_step'], c['throughput']['kernels_ms'])
" > $OUT/run.log 2>&1; echo "rc=$?" >> $OUT/run.log
tail -5 $OUT/run.log
python - <<'PY' > gpurun_out/r04_grid_pmc.txt
import collections, csv, glob
rows = []
for p in glob.glob("gpurun_out/grid_pmc/run/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
dur = {}
for p in glob.glob("gpurun_out/grid_pmc/run/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "k_engine_synth_mfma<false, false>" not in k and "k_gwb_czt<true, false" not in k:
        continue
    key = ("synth" if "synth" in k else "czt", int(r["Grid_Size"]))
    a = agg.setdefault(key, [set(), collections.Counter(), 0.0])
    if r["Dispatch_Id"] not in a[0]:
        a[0].add(r["Dispatch_Id"]); a[2] += dur.get(r["Dispatch_Id"], 0.0)
    a[1][r["Counter_Name"]] += float(r["Counter_Value"])
print("# 68 pulsars x N TOAs, R = 1024: the fused synthesis kernel and the chirp-z kernel by launch size (rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES")
print("# SQ_BUSY_CYCLES; scripts/gpu_r4_grid_pmc.sh).  output elements of the fused kernel = 1024 x 68 x N; its workgroups = 256-TOA tiles x 64 realisation groups.")
print(f"{'kernel':6s} {'grid (threads)':>15s} {'dispatches':>10s} {'avg us':>9s} {'VALU wave-instr':>16s} {'waves':>10s} {'instr/wave':>10s} {'VALU lane-instr per output element':>36s}")
for (kern, grid), (disp, c, t) in sorted(agg.items()):
    n = len(disp)
    insts, waves = c["SQ_INSTS_VALU"] / n, c["SQ_WAVES"] / n
    ntoa = {1114112: 122, 4456448: 1000, 22282240: 5000}.get(grid)
    per = f"{insts * 64 / (1024 * 68 * ntoa):10.1f}  (N = {ntoa})" if (kern == "synth" and ntoa) else ""
    print(f"{kern:6s} {grid:15d} {n:10d} {t / n:9.1f} {insts:16.0f} {waves:10.0f} {insts / max(waves, 1):10.1f} {per:>36s}")
PY
cat gpurun_out/r04_grid_pmc.txt
