// Library plumbing + the host-side (realisation-independent) epoch bucketing.
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
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

// HOST helper of the ORF set-up: the cosine of the pair separation as spharmORFbasis.py:24 writes it,
//   argument = sin(theta1) * sin(theta2) * cos(phi1 - phi2) + cos(theta1) * cos(theta2)      (left to right, no contraction),
// for every pair a <= b through libm's scalar sin / cos - what the reference's per-pair calczeta() evaluates on NumPy float64 scalars
// (NumPy routes float64 sin / cos to libm; its arccos may not be libm's - AVX512 builds use SVML - so zeta = arccos(argument) and
// cos(zeta) stay with NumPy on the host side: spharmORFbasis.pair_zeta_cos).  same[a*P+b] = 1 where the two positions are identical
// (the reference's exact-equality branch, :23: zeta = 0 whatever the arithmetic would give).  Both outputs are symmetric [P, P].
extern "C" int pta_orf_pair_arguments(const double *locs_host, int P, double *arg_host, uint8_t *same_host) {
  PTA_REQUIRE(locs_host && arg_host && same_host, PTA_E_ARG, "pta_orf_pair_arguments: NULL argument");
  PTA_REQUIRE(P > 0 && P <= 46340, PTA_E_ARG, "pta_orf_pair_arguments: P=%d", P);
  std::vector<double> st(P), ct(P);
  for (int a = 0; a < P; ++a) {
    st[a] = sin(locs_host[2 * a + 1]);
    ct[a] = cos(locs_host[2 * a + 1]);
  }
  for (int a = 0; a < P; ++a) {
    const double p1 = locs_host[2 * a], t1 = locs_host[2 * a + 1];
    for (int b = a; b < P; ++b) {
      const double p2 = locs_host[2 * b], t2 = locs_host[2 * b + 1];
      const double s12 = st[a] * st[b];
      const double c12 = cos(p1 - p2);
      const double left = s12 * c12;
      const double right = ct[a] * ct[b];
      const double arg = left + right;
      const uint8_t same = (p1 == p2 && t1 == t2) ? 1 : 0;
      arg_host[(int64_t)a * P + b] = arg_host[(int64_t)b * P + a] = arg;
      same_host[(int64_t)a * P + b] = same_host[(int64_t)b * P + a] = same;
    }
  }
  return PTA_OK;
}

// ---- NumPy's LEGACY normal stream, natively and on host threads (replay mode of the drop-in API) -------------------------------
// The reference draws every deviate from the global np.random stream, re-seeded per call (white_noise.py:79-80,154-155,
// red_noise.py:112-113): MT19937 seeded by init_genrand(seed), 53-bit doubles (a >> 5, b >> 6), Marsaglia's polar method with the
// second deviate of a pair cached across calls (numpy/random/src/legacy/legacy-distributions.c: legacy_gauss).  A stream is serial,
// but the streams of different pulsars are independent - and RandomState holds the GIL, so Python threads cannot draw them side by
// side (measured 2.3x slower).  This restatement does: one std::thread per share of the streams, libm's log / sqrt as NumPy calls
// them, no contraction (the library is built with -ffp-contract=off).  tests/test_host_logic.py pins it against RandomState value for
// value, including the state the last stream is left in (which the caller installs as the global stream's).
namespace {
struct pta_mt19937 {
  uint32_t key[624];
  int pos;
  int has_gauss;
  double gauss;
  void seed(uint32_t s) {
    for (int i = 0; i < 624; ++i) {
      key[i] = s;
      s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    pos = 624;
    has_gauss = 0;
    gauss = 0.0;
  }
  void refill() {
    const uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    int i = 0;
    uint32_t y;
    for (; i < 624 - 397; ++i) {
      y = (key[i] & UP) | (key[i + 1] & LO);
      key[i] = key[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    for (; i < 623; ++i) {
      y = (key[i] & UP) | (key[i + 1] & LO);
      key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    y = (key[623] & UP) | (key[0] & LO);
    key[623] = key[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    pos = 0;
  }
  uint32_t next() {
    if (pos == 624) refill();
    uint32_t y = key[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  double next_double() {
    const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  double next_gauss() {
    if (has_gauss) {
      const double t = gauss;
      has_gauss = 0;
      gauss = 0.0;
      return t;
    }
    double x1, x2, r2;
    do {
      x1 = 2.0 * next_double() - 1.0;
      x2 = 2.0 * next_double() - 1.0;
      r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    const double f = sqrt(-2.0 * log(r2) / r2);
    gauss = f * x1;
    has_gauss = 1;
    return f * x2;
  }
};
}  // namespace

extern "C" int pta_legacy_randn(const uint32_t *seeds_host, const int64_t *counts_host, const int64_t *offsets_host, int n_streams,
                                double *out_host, uint32_t *last_key_host, int32_t *last_pos_has_host, double *last_gauss_host,
                                int n_threads) {
  PTA_REQUIRE(seeds_host && counts_host && offsets_host && out_host, PTA_E_ARG, "pta_legacy_randn: NULL argument");
  PTA_REQUIRE(n_streams >= 0, PTA_E_ARG, "pta_legacy_randn: n_streams=%d", n_streams);
  for (int i = 0; i < n_streams; ++i)
    PTA_REQUIRE(counts_host[i] >= 0 && offsets_host[i] >= 0, PTA_E_ARG, "pta_legacy_randn: stream %d: count=%lld offset=%lld", i,
                (long long)counts_host[i], (long long)offsets_host[i]);
  if (n_streams == 0) return PTA_OK;
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 16) nt = 16;  // 68 streams of 10 k deviates: 5.3 / 1.5 / 0.7 / 1.8 ms with 1 / 4 / 16 / 64 threads (thread start-up)
  if (nt > n_streams) nt = n_streams;
  auto draw = [&](int i) {
    pta_mt19937 g;
    g.seed(seeds_host[i]);
    double *o = out_host + offsets_host[i];
    for (int64_t k = 0; k < counts_host[i]; ++k) o[k] = g.next_gauss();
    if (i == n_streams - 1 && last_key_host && last_pos_has_host && last_gauss_host) {  // the state the global stream is left in
      memcpy(last_key_host, g.key, sizeof(g.key));
      last_pos_has_host[0] = g.pos;
      last_pos_has_host[1] = g.has_gauss;
      last_gauss_host[0] = g.gauss;
    }
  };
  if (nt == 1) {
    for (int i = 0; i < n_streams; ++i) draw(i);
    return PTA_OK;
  }
  std::atomic<int> next_stream{0};
  std::vector<std::thread> pool;
  pool.reserve(nt);
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([&]() {
      for (int i = next_stream.fetch_add(1); i < n_streams; i = next_stream.fetch_add(1)) draw(i);
    });
  for (auto &th : pool) th.join();
  return PTA_OK;
}
