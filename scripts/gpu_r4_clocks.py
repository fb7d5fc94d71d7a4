"""Engine clock + board power under each hot kernel: is the chip at 1.8-2.1 GHz under the VALU-bound kernels, and why?

For every load: a one-wave probe (scripts/probe_src/clock_probe.hip) samples (s_memrealtime, s_memtime) every 100 us on a side stream for
~2 s while the load loops on the main stream; the slope between samples is the engine clock (s_memtime = shader cycles, s_memrealtime =
100 MHz).  A host thread samples the hwmon power file (or rocm-smi when there is none) beside it.  Writes gpurun_out/r04_clocks.json and a
text table (-> profiles/r04_clocks.txt).
"""
import ctypes, glob, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import build_engine
from pta_replicator_amd import _lib, device as dv

so = os.path.join(ROOT, "scripts", "probe_src", "libclock_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, so.replace("libclock_probe.so", "clock_probe.hip")])
probe = ctypes.CDLL(so)
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe.clock_probe_load.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

POWER_FILES = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
FREQ_FILES = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))


def read_int(path):
    try:
        with open(path) as fh:
            return int(fh.read().strip())
    except Exception:
        return None


class PowerSampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.samples = []

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            if POWER_FILES:
                v = read_int(POWER_FILES[0])
                f = read_int(FREQ_FILES[0]) if FREQ_FILES else None
                self.samples.append((t, None if v is None else v / 1e6, None if f is None else f / 1e6))
                time.sleep(0.002)
            else:
                try:
                    o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                    j = json.loads(o)
                    c = next(iter(j.values()))
                    pw = next((float(v) for k, v in c.items() if "ower" in k and "W" in k), None)
                    sc = next((v for k, v in c.items() if k.startswith("sclk")), None)
                    self.samples.append((t, pw, sc))
                except Exception as e:
                    self.samples.append((t, None, str(e)[:60]))


side = torch.cuda.Stream()
NS, PERIOD_US = 20000, 100


def measure(name, load, seconds=2.0):
    buf = torch.zeros((NS, 2), dtype=torch.int64, device="cuda")
    load()                                   # warm
    torch.cuda.synchronize()
    ps = PowerSampler(); ps.start()
    time.sleep(0.05)
    rc = probe.clock_probe_launch(buf.data_ptr(), NS, int((seconds + 0.3) * 1e6), PERIOD_US, side.cuda_stream)
    assert rc == 0, rc
    time.sleep(0.1)                          # 0.1 s of idle samples first
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        load(); n += 1
        if n % 8 == 0:
            torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    ps.stop = True; ps.join()
    s = buf.cpu().numpy()
    s = s[s[:, 0] > 0]
    rt, sc = s[:, 0].astype(np.float64), s[:, 1].astype(np.float64)
    ghz = np.diff(sc) / (np.diff(rt) * 10.0)          # cycles per (10 ns tick) -> GHz
    tt = (rt[1:] - rt[0]) / 1e8                        # s since probe start
    busy = (tt > 0.3) & (tt < 0.1 + seconds - 0.2)     # well inside the load window
    idle = tt < 0.08
    pw = [p for (_, p, _) in ps.samples if isinstance(p, float)]
    pt = [t for (t, p, _) in ps.samples if isinstance(p, float)]
    pw_busy = [p for t, p in zip(pt, pw) if t0 + 0.3 < t < t0 + seconds - 0.2]
    rec = {"load": name, "iterations": n, "ms_per_iteration": t_load / n * 1e3,
           "engine_clock_GHz": {"median": float(np.median(ghz[busy])), "p05": float(np.percentile(ghz[busy], 5)), "p95": float(np.percentile(ghz[busy], 95)),
                                "min": float(ghz[busy].min()), "max": float(ghz[busy].max()), "idle_before_load": float(np.median(ghz[idle])) if idle.any() else None,
                                "samples": int(busy.sum())},
           "power_W": {"median": float(np.median(pw_busy)) if pw_busy else None, "max": float(np.max(pw_busy)) if pw_busy else None, "samples": len(pw_busy),
                       "source": POWER_FILES[0] if POWER_FILES else "rocm-smi --showpower"},
           "clock_trace_GHz_every_20ms": [round(float(np.median(ghz[(tt >= a) & (tt < a + 0.02)])), 3) for a in np.arange(0, tt.max() - 0.02, 0.02) if ((tt >= a) & (tt < a + 0.02)).any()]}
    sm = [x for (_, _, x) in ps.samples if x is not None]
    if sm:
        rec["smi_sclk_samples"] = sm[:: max(1, len(sm) // 8)][:8]
    print(json.dumps({k: v for k, v in rec.items() if k != "clock_trace_GHz_every_20ms"}), flush=True)
    return rec


def main():
    eng, psrs, noise = build_engine(68, 5000, seed=20260921)
    R = 1024
    out = dv.empty((R, eng.n_toa))
    s = dv.stream_ptr()
    ws = eng.workspace(R)
    eng.generate(R, out=out)
    P, Nf, npts = eng.P, eng.grid["Nf"], eng.plan.gw_npts
    scratch = dv.empty((1 << 27,))           # 1 GiB
    s = s.value
    loads = [
        ("idle (probe only)", lambda: time.sleep(0.01)),
        ("k_engine_synth_mfma (pta_engine_synth, R=1024)", lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(out), out.stride(0), s)),
        ("k_gwb_czt (pta_gwb_czt, R=1024)", lambda: _lib.call("pta_gwb_czt", eng.seed, 0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt], dv.ptr(ws["G0"]), npts, 0, 0, s)),
        ("headline step (generate 1024)", lambda: eng.generate(R, out=out)),
        ("control: 16 v_fma_f64 chains per lane, 8 blocks/CU", lambda: probe.clock_probe_load(0, scratch.data_ptr(), 0, 20000, 256 * 8, s)),
        ("control: v_mad_u64_u32 chains (Philox multiplier), 8 blocks/CU", lambda: probe.clock_probe_load(1, scratch.data_ptr(), 0, 20000, 256 * 8, s)),
        ("control: fp64 MFMA 4x4 register tile, 2 blocks/CU", lambda: probe.clock_probe_load(2, scratch.data_ptr(), 0, 20000, 256 * 2, s)),
        ("control: streaming stores (1 GiB)", lambda: probe.clock_probe_load(3, scratch.data_ptr(), scratch.numel() // 2, 0, 256 * 8, s)),
        ("pta_rng_fill_normal (Philox + Box-Muller -> HBM, 2 x 64 Mi deviates)", lambda: _lib.call("pta_rng_fill_normal", eng.seed, 0, 1, 0, 1 << 26, 1, dv.ptr(scratch), None, 1 << 27, 0, s)),
    ]
    recs = []
    for name, fn in loads:
        try:
            recs.append(measure(name, fn))
        except Exception as e:
            recs.append({"load": name, "error": str(e)[:300]})
            print(recs[-1], flush=True)
    # TD mode loads: the factorisation and the triangular product
    try:
        eng.prepare_td()
        recs.append(measure("prepare_td (assembly + batched Cholesky, 68 x 5000^2)", eng.prepare_td, seconds=2.0))
        recs.append(measure("generate_td(1024) (L.z)", lambda: eng.generate_td(R, out=out), seconds=2.0))
    except Exception as e:
        recs.append({"load": "td", "error": str(e)[:300]})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_clocks.json"), "w") as fh:
        json.dump({"power_files": POWER_FILES, "freq_files": FREQ_FILES, "records": recs}, fh, indent=1)
    with open(os.path.join(ROOT, "gpurun_out", "r04_clocks.txt"), "w") as fh:
        fh.write("# engine clock by in-kernel probe (s_memtime / s_memrealtime slope, 100 us samples) and board power beside each load; scripts/gpu_r4_clocks.py\n")
        fh.write(f"{'load':58s} {'ms/iter':>9s} {'clk med':>8s} {'p05':>6s} {'p95':>6s} {'min':>6s} {'power W med':>12s} {'max':>6s}\n")
        for r in recs:
            if "error" in r:
                fh.write(f"{r['load']:58s} ERROR {r['error']}\n")
                continue
            c, p = r["engine_clock_GHz"], r["power_W"]
            fh.write(f"{r['load']:58s} {r['ms_per_iteration']:9.3f} {c['median']:8.3f} {c['p05']:6.3f} {c['p95']:6.3f} {c['min']:6.3f} "
                     f"{(p['median'] if p['median'] is not None else float('nan')):12.1f} {(p['max'] if p['max'] is not None else float('nan')):6.1f}\n")
    print(open(os.path.join(ROOT, "gpurun_out", "r04_clocks.txt")).read())


if __name__ == "__main__":
    main()
