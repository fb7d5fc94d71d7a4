"""Record the UNMODIFIED reference's CPU time for the bench workload in the build container (the only host where
/root/reference is mounted): profiles/r03_cpu_baseline_reference.json, carried by bench.py as cpu_baseline.reference_container.

    python scripts/cpu_baseline_reference.py

Same measurement as bench.py's live `cpu_baseline` leg (oracle/cpu_baseline.py in a subprocess, BLAS threads 1 and all cores,
one warm-up + 3 timed repeats on the bounded sample) - here kind == "reference" by construction."""
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import cpu_baseline, headline_array  # noqa: E402

psrs, noise = headline_array(68, 5000)
rec = cpu_baseline(psrs, noise)
assert rec["kind"] == "reference", "run this where /root/reference is mounted"
rec["host"] = {"container": "build container (no GPU)", "cpus": os.cpu_count(), "machine": platform.machine(), "python": platform.python_version()}
try:
    with open("/proc/cpuinfo") as fh:
        rec["host"]["cpu_model"] = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), None)
except OSError:
    pass
path = os.path.join(ROOT, "profiles", "r03_cpu_baseline_reference.json")
with open(path, "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec, indent=1))
