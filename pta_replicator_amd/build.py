"""Build libpta_replicator_amd.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m pta_replicator_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU present; the resulting .so is git-ignored but travels to
the GPU box with the source snapshot.  No JIT, no fallback: the Python package refuses to work without it.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libpta_replicator_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: expressions whose float64 association is part of the parity contract (phase arguments,
# white-noise sums) must not be fused behind our back; hot loops call fma()/MFMA explicitly.
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "pta_replicator_amd.h"))
    jobs = []
    for src in sources():
        obj = os.path.join(BUILD, src[:-4] + ".o")
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return src

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for done in ex.map(compile_one, jobs):
            if verbose:
                print(f"[pta_replicator_amd.build] compiled {done}", file=sys.stderr)
    objs = [os.path.join(BUILD, s[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[pta_replicator_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
