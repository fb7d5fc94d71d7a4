// Library plumbing + the host-side (realisation-independent) epoch bucketing.
#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include "pta_common.h"

static thread_local char g_err[512] = "";

void pta_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int pta_abi_version(void) { return PTA_ABI_VERSION; }
extern "C" const char *pta_last_error(void) { return g_err; }

extern "C" int pta_device_info(int *cu_count, int *wavefront, char *arch, int arch_len) {
  int dev = 0;
  PTA_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  PTA_HIP(hipGetDeviceProperties(&p, dev));
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wavefront) *wavefront = p.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return PTA_OK;
}

// Greedy bucketing of white_noise.py:21-31: walk the TOAs in time order; a TOA joins the open
// bucket while (t - bucket_ref) < dt, bucket_ref being the bucket's FIRST TOA; otherwise it opens
// a new bucket.  Output is the column index of the reference's dense U (white_noise.py:37-39).
extern "C" int pta_quantize_epochs(const double *times_host, int N, double dt, const int64_t *order_host,
                                   int32_t *epoch_of_host, int32_t *first_index_host, int *n_epochs) {
  PTA_REQUIRE(times_host && epoch_of_host && first_index_host && n_epochs, PTA_E_ARG, "pta_quantize_epochs: NULL argument");
  PTA_REQUIRE(N > 0, PTA_E_ARG, "pta_quantize_epochs: N must be positive (got %d)", N);
  std::vector<int64_t> order(N);
  if (order_host) {
    std::copy(order_host, order_host + N, order.begin());
    for (int i = 0; i < N; ++i) PTA_REQUIRE(order[i] >= 0 && order[i] < N, PTA_E_ARG, "pta_quantize_epochs: order[%d] out of range", i);
  } else {
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return times_host[a] < times_host[b]; });
  }
  int e = -1;
  double ref = 0.0;
  for (int s = 0; s < N; ++s) {
    int64_t i = order[s];
    if (e < 0 || !(times_host[i] - ref < dt)) {
      ref = times_host[i];
      ++e;
      first_index_host[e] = (int32_t)i;
    }
    epoch_of_host[i] = e;
  }
  *n_epochs = e + 1;
  return PTA_OK;
}

// out[i] = fma(x[i][2], y[2], fma(x[i][1], y[1], x[i][0] * y[0])): the three-term dot product in the association OpenBLAS' ddot
// (= np.dot on two 3-vectors, the call at deterministic.py:364-372 inside the reference's per-source loop) uses on x86-64 -
// verified against np.dot by the Python caller on every catalogue before it trusts this routine.  Lets cw_source_params
// evaluate F+, Fx and cos(mu) of a whole catalogue without a Python-level loop while staying bit-identical to the reference.
extern "C" int pta_dot3_host(const double *x_host, int64_t n, const double *y_host, double *out_host) {
  PTA_REQUIRE(x_host && y_host && out_host, PTA_E_ARG, "pta_dot3_host: NULL argument");
  PTA_REQUIRE(n >= 0, PTA_E_ARG, "pta_dot3_host: n=%lld", (long long)n);
  const double y0 = y_host[0], y1 = y_host[1], y2 = y_host[2];
  for (int64_t i = 0; i < n; ++i) {
    const double *x = x_host + 3 * i;
    const double p0 = x[0] * y0;  // a rounded product (-ffp-contract=off: never fused with what follows)
    out_host[i] = fma(x[2], y2, fma(x[1], y1, p0));
  }
  return PTA_OK;
}

// out[i] = pow(x[i], y) through the C library's scalar pow - what a NumPy float64 SCALAR `x ** y` (the reference's per-source loop
// body, deterministic.py:340-383) and numba's compiled loops call; NumPy's ARRAY power uses its own SIMD kernels, which differ from
// libm by an ulp on ~5 % of arguments (measured), enough to move the evolving-CW phase offsets by 1e-7 relative.
extern "C" int pta_pow_host(const double *x_host, double y, int64_t n, double *out_host) {
  PTA_REQUIRE(x_host && out_host, PTA_E_ARG, "pta_pow_host: NULL argument");
  PTA_REQUIRE(n >= 0, PTA_E_ARG, "pta_pow_host: n=%lld", (long long)n);
  for (int64_t i = 0; i < n; ++i) out_host[i] = pow(x_host[i], y);
  return PTA_OK;
}
