"""Record the UNMODIFIED reference's CPU time for the bench workload in the build container (the only host where
/root/reference is mounted): profiles/r06_cpu_baseline_reference.json, carried by bench.py as cpu_baseline.reference_container.

    python scripts/cpu_baseline_reference.py [--subset S]

Same measurement as bench.py's live `cpu_baseline` leg (oracle/cpu_baseline.py in a subprocess, BLAS threads 1 and all cores,
one warm-up + 3 timed repeats) - here kind == "reference" by construction, and by default on ALL pulsars of the array (no
sub-sampling, no scaling: VERDICT r5 #1c)."""
import datetime
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import cpu_baseline, headline_array  # noqa: E402

subset = int(sys.argv[sys.argv.index("--subset") + 1]) if "--subset" in sys.argv else 68
psrs, noise = headline_array(68, 5000)
rec = cpu_baseline(psrs, noise, subset=subset, timeout=3600)
assert rec["kind"] == "reference", "run this where /root/reference is mounted"
rec["date"] = datetime.date.today().isoformat()
rec["host"] = {"container": "build container (no GPU)", "cpus": os.cpu_count(), "machine": platform.machine(), "python": platform.python_version()}
try:
    with open("/proc/cpuinfo") as fh:
        rec["host"]["cpu_model"] = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), None)
except OSError:
    pass
path = os.path.join(ROOT, "profiles", "r06_cpu_baseline_reference.json")
with open(path, "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec, indent=1))
