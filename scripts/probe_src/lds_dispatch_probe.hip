// Throw-away probe: what does a dispatch of 68 workgroups cost as a function of its LDS allocation, code size aside?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_probe scripts/probe_src/lds_dispatch_probe.hip && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KB>
__global__ __launch_bounds__(256) void k_lds(double *out, int spin) {
  __shared__ double buf[KB * 128];
  const int t = threadIdx.x;
  buf[t] = t;
  __syncthreads();
  double a = buf[(t + 1) & 255];
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;   // ~spin dependent fp64 ops
  out[blockIdx.x * 256 + t] = a;
}
template <int KB>
static void run(const char *name, double *out, int spin) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_lds<KB>, dim3(68), dim3(256), 0, 0, out, spin);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("%-28s spin %6d: %.2f us per launch (100 back-to-back launches)\n", name, spin, ms * 10.0f);
  }
}
int main() {
  double *out; hipMalloc(&out, 68 * 256 * 8);
  for (int spin : {0, 4000}) {
    run<1>("LDS 1 KB", out, spin);
    run<32>("LDS 32 KB", out, spin);
    run<64>("LDS 64 KB", out, spin);
    run<100>("LDS 100 KB", out, spin);
    run<131>("LDS 131 KB", out, spin);
    run<158>("LDS 158 KB", out, spin);
  }
  return 0;
}
