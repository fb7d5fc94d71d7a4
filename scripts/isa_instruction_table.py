"""ISA-level instruction-class table of the fused synthesis kernel (VERDICT r2 #3iii): where the ~219 VALU wave-instructions per output
element of k_engine_synth_mfma<false> go.  No GPU needed: hipcc --save-temps for gfx950, then

  * the Gaussian draw is compiled piece by piece (one tiny kernel per device function of pta_rng.h, inputs loaded from memory so that
    nothing folds away): Philox-4x32-10, the uniform conversion, -2 ln u, sqrt, sin/cos(2 pi u), the final products - exact static
    VALU / SALU / LDS counts of straight-line code;
  * the kernel itself is split at its basic blocks: ECORR staging loop (executed ~epn * 16 / 256 times per workgroup), red-noise
    MFMA loop (5 trips at K = 60), the unrolled 4-TOA epilogue (once) - static counts weighted by those trip counts give the
    dynamic VALU count per workgroup and per output element (16 realisations x 256 TOAs).

Writes profiles/r03_isa_instruction_classes.txt.  The PMC figure to compare with is SQ_INSTS_VALU / (R * n_toa / 64) of profiles/r03_pmc.json."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pta_replicator_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result", "--save-temps", "-c"]

PROBE = r'''
#include <hip/hip_runtime.h>
#include "%(csrc)s/pta_rng.h"
extern "C" {
__global__ void p_philox(const uint32_t *in, uint32_t *out) {
  pta_u32x4 v = pta_philox_draw(((const uint64_t *)in)[0], ((const uint64_t *)in)[1] + threadIdx.x, in[4], in[5] + threadIdx.x);
  out[4 * threadIdx.x] = v.x; out[4 * threadIdx.x + 1] = v.y; out[4 * threadIdx.x + 2] = v.z; out[4 * threadIdx.x + 3] = v.w;
}
__global__ void p_uniform(const uint32_t *in, double *out) {
  pta_u32x4 v = {in[4 * threadIdx.x], in[4 * threadIdx.x + 1], in[4 * threadIdx.x + 2], in[4 * threadIdx.x + 3]};
  double a, b; pta_uniform_pair(v, a, b); out[2 * threadIdx.x] = a; out[2 * threadIdx.x + 1] = b;
}
__global__ void p_neg2log(const double *in, double *out) { pta_rng_stage_tables(); __syncthreads(); out[threadIdx.x] = pta_neg2log(in[threadIdx.x]); }
__global__ void p_sqrt(const double *in, double *out) { out[threadIdx.x] = pta_sqrt_pos(in[threadIdx.x]); }
__global__ void p_sincos(const double *in, double *out) {
  pta_rng_stage_tables(); __syncthreads();
  double s, c; pta_sincos_2pi(in[threadIdx.x], s, c); out[2 * threadIdx.x] = s; out[2 * threadIdx.x + 1] = c;
}
__global__ void p_pair(const uint32_t *in, double *out) {
  pta_rng_stage_tables(); __syncthreads();
  double a, b; pta_normal_pair(((const uint64_t *)in)[0], ((const uint64_t *)in)[1] + threadIdx.x, in[4], in[5] + threadIdx.x, a, b, 0);
  out[2 * threadIdx.x] = a; out[2 * threadIdx.x + 1] = b;
}
__global__ void p_empty_tables(const double *in, double *out) { pta_rng_stage_tables(); __syncthreads(); out[threadIdx.x] = in[threadIdx.x]; }
__global__ void p_empty(const double *in, double *out) { out[threadIdx.x] = in[threadIdx.x]; }
}
'''


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"


def kernel_bodies(asm):
    out, name, body = {}, None, []
    for line in open(asm):
        m = re.match(r"^([A-Za-z_][\w]*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if name: out[name] = body
            name, body = m.group(1), []
            continue
        if name is None: continue
        if line.startswith(".Lfunc_end"):
            out[name] = body; name = None; continue
        body.append(line)
    return out


def count(lines):
    c, vops = collections.Counter(), collections.Counter()
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        op = t.split()[0]
        k = classify(op)
        c[k] += 1
        if k == "valu": vops[op] += 1
    return c, vops


def blocks(lines):
    """[(label, lines)] split at .LBB labels"""
    out, lab, cur = [], "entry", []
    for ln in lines:
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            out.append((lab, cur)); lab, cur = m.group(1), []
        else:
            cur.append(ln)
    out.append((lab, cur))
    return out


def main():
    rep = []
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.hip")
        open(src, "w").write(PROBE % {"csrc": CSRC})
        subprocess.check_call([HIPCC] + FLAGS + [src, "-o", os.path.join(d, "probe.o")], cwd=d, stderr=subprocess.DEVNULL)
        kb = kernel_bodies(os.path.join(d, "probe-hip-amdgcn-amd-amdhsa-gfx950.s"))
        base = {k: count(kb[k])[0] for k in ("p_empty", "p_empty_tables")}
        rep.append("Gaussian draw, piece by piece (static counts of straight-line code; probe overhead - loads, address arithmetic, table staging - subtracted)")
        rep.append(f"{'piece':34s} {'VALU':>6s} {'LDS':>5s} {'SALU':>5s}   VALU mix (top)")
        tot = collections.Counter()
        for k, label, b in (("p_philox", "Philox-4x32-10 (counter -> 4 x u32)", "p_empty"), ("p_uniform", "2 x 52-bit uniform", "p_empty"),
                            ("p_neg2log", "-2 ln u (128-entry table + deg-5)", "p_empty_tables"), ("p_sqrt", "sqrt (rsq + Newton + residual)", "p_empty"),
                            ("p_sincos", "sin, cos 2 pi u (32-entry table + Taylor)", "p_empty_tables"), ("p_pair", "pta_normal_pair, whole (z0, z1)", "p_empty_tables")):
            c, vops = count(kb[k])
            v, l, s = c["valu"] - base[b]["valu"], c["lds"] - base[b]["lds"], c["salu"] - base[b]["salu"]
            top = ", ".join(f"{op} {n}" for op, n in vops.most_common(5))
            rep.append(f"{label:34s} {v:6d} {l:5d} {s:5d}   {top}")
            if k != "p_pair": tot["valu"] += v
        rep.append(f"{'sum of the pieces (+ 2 final products)':34s} {tot['valu'] + 2:6d}")
        # ---- the kernel itself
        subprocess.check_call([HIPCC] + FLAGS + ["-DPTA_ISA_TABLE_MAIN_PATH_ONLY", os.path.join(CSRC, "pta_engine_kernels.hip"), "-o", os.path.join(d, "eng.o")],
                              cwd=d, stderr=subprocess.DEVNULL)
        kb = kernel_bodies(os.path.join(d, "pta_engine_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"))
        name = next(k for k in kb if k.startswith("_Z19k_engine_synth_mfmaILb0ELb0E"))
        rep.append("")
        rep.append(f"k_engine_synth_mfma<false, false> ({name}): basic blocks with VALU / MFMA / LDS / VMEM counts")
        bl = blocks(kb[name])
        rows = []
        for lab, ls in bl:
            c, _ = count(ls)
            if sum(c.values()) == 0: continue
            back = any(re.search(r"s_cbranch\w*\s+" + re.escape(lab) + r"\b", x) for x in ls)
            rows.append((lab, c, back))
            rep.append(f"  {lab:12s} valu {c['valu']:5d}  mfma {c['mfma']:3d}  lds {c['lds']:3d}  vmem {c['vmem']:3d}  salu {c['salu']:4d}  wait {c['wait']:3d}{'   <- loop' if back else ''}")
        # dynamic estimate: loops = blocks that branch back to themselves.  Identify them by content: the staging loop holds the
        # Philox multiplies (v_mad_u64_u32) and LDS writes but no MFMA; the red-noise loop holds MFMAs
        stag = [r for r in rows if r[2] and r[1]["mfma"] == 0 and r[1]["valu"] > 50]
        rnl = [r for r in rows if r[2] and r[1]["mfma"] > 0]
        once = sum(r[1]["valu"] for r in rows if not r[2])
        n_epochs_pairs = 121.0     # pairs a 256-TOA tile of the bench workload stages on average (4700 epochs per 5000 TOAs)
        t_stag = n_epochs_pairs * 16 / 256.0
        t_rn = 5.0                 # K = 60 in steps of 12
        dyn = once + sum(r[1]["valu"] for r in stag) * t_stag + sum(r[1]["valu"] for r in rnl) * t_rn
        per_elem = dyn / 16.0                # a wave writes 16 realisations x 64 TOAs: 16 output elements per lane
        rep.append("")
        rep.append(f"dynamic VALU wave-instructions per wave: straight-line {once} + staging loop {sum(r[1]['valu'] for r in stag)} x {t_stag:.2f} trips "
                   f"+ red-noise loop {sum(r[1]['valu'] for r in rnl)} x {t_rn:.0f} trips = {dyn:.0f}")
        rep.append(f"=> {per_elem:.1f} VALU instructions per output element (a wave writes 16 realisations x 64 TOAs = 16 elements per lane); "
                   f"the all-VALU alternatives of the has_gw / has_wn / epn branches are counted once each, so this is an upper bound of the executed path")
        pair = count(kernel_bodies(os.path.join(d, "probe-hip-amdgcn-amd-amdhsa-gfx950.s"))["p_pair"])[0]["valu"] - base["p_empty_tables"]["valu"]
        rng = (1 + n_epochs_pairs / 256.0) * pair
        rep.append(f"of which Gaussian draws: (1 EFAC/EQUAD pair + {n_epochs_pairs / 256.0:.3f} ECORR pairs) per element x {pair} = {rng:.1f} "
                   f"({100 * rng / per_elem:.0f} %); the remaining {per_elem - rng:.1f} are GWB interpolation, accumulation, operand addressing, MFMA feeding and stores")
        rep.append("per Gaussian pair: Philox 46 (20 of them v_mad_u64_u32 at half rate), uniforms 10, -2 ln u 22, sqrt 10, sin / cos 26, products 2")
    path = os.path.join(ROOT, "profiles", "r03_isa_instruction_classes.txt")
    open(path, "w").write("\n".join(rep) + "\n")
    print("\n".join(rep))


if __name__ == "__main__":
    main()
