// The one collective of the path (BASELINE.json north_star): gathering the row-sharded residual arrays to one rank.
// RCCL is bound at RUN time (dlopen of the copy already loaded into the process - PyTorch's librccl.so - else the system one): the
// library itself carries no link-time dependency on it, so single-GPU users never load a communication library.
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <mutex>
#include "pta_common.h"

namespace {
typedef int (*fn_sendrecv)(void *, size_t, int, int, void *, hipStream_t);  // ncclSend / ncclRecv (const void* for send)
typedef int (*fn_group)(void);
typedef const char *(*fn_errstr)(int);
struct rccl_api {
  bool ok = false;
  fn_sendrecv send = nullptr, recv = nullptr;
  fn_group gstart = nullptr, gend = nullptr;
  fn_errstr errstr = nullptr;
};
rccl_api g_rccl;
std::once_flag g_rccl_once;

// the RCCL copy the process ALREADY uses (torch.distributed's, or the one a ctypes integrator loaded by full path), whatever its SONAME:
// the loaded-object list is searched for a file whose BASENAME is librccl.so[.N...] and that very file is re-opened (looking
// it up by the two usual names could miss it and bind a second copy, whose ncclSend would then be handed the first copy's communicator).
// A network plugin (librccl-net.so, librccl_net_ofi.so, ...) also carries "librccl" in its name: the basename test rejects it,
// and every candidate must export ncclSend before it is taken - the walk goes on otherwise.
bool rccl_basename_matches(const char *path) {
  const char *b = strrchr(path, '/');
  b = b ? b + 1 : path;
  if (strncmp(b, "librccl.so", 10) != 0) return false;
  return b[10] == '\0' || b[10] == '.';
}

bool rccl_bind(void *h, rccl_api &r) {
  r.send = (fn_sendrecv)dlsym(h, "ncclSend");
  r.recv = (fn_sendrecv)dlsym(h, "ncclRecv");
  r.gstart = (fn_group)dlsym(h, "ncclGroupStart");
  r.gend = (fn_group)dlsym(h, "ncclGroupEnd");
  r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  return r.send && r.recv && r.gstart && r.gend;
}

int rccl_find_loaded(struct dl_phdr_info *info, size_t, void *data) {
  if (!info->dlpi_name || !rccl_basename_matches(info->dlpi_name)) return 0;
  void *h = dlopen(info->dlpi_name, RTLD_NOW | RTLD_NOLOAD);
  if (h && rccl_bind(h, *(rccl_api *)data)) return 1;  // stop: bound (the handle is kept for the life of the process)
  if (h) dlclose(h);                                    // a rejected candidate: give back the reference RTLD_NOLOAD took
  return 0;                                             // not a usable RCCL: keep walking
}

void rccl_load_once() {
  rccl_api &r = g_rccl;
  if (dl_iterate_phdr(rccl_find_loaded, &r)) {
    r.ok = true;
    return;
  }
  for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {  // nothing loaded yet: the usual names
    void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h && rccl_bind(h, r)) {
      r.ok = true;
      return;
    }
    if (h) dlclose(h);
  }
  r = rccl_api();
}

int rccl_load() {
  std::call_once(g_rccl_once, rccl_load_once);
  return g_rccl.ok ? PTA_OK : PTA_E_ARG;
}
}  // namespace

#define PTA_NCCL(call)                                                                                          \
  do {                                                                                                          \
    int e_ = (call);                                                                                            \
    if (e_ != 0) {                                                                                              \
      pta_set_error("%s failed: %s (%s:%d)", #call, g_rccl.errstr ? g_rccl.errstr(e_) : "rccl error", __FILE__, __LINE__); \
      return PTA_E_HIP;                                                                                         \
    }                                                                                                           \
  } while (0)

// shard of rank r of `total` rows over `world` ranks: contiguous, sizes differ by at most one (distributed.shard_range)
static inline void pta_shard(int64_t total, int r, int world, int64_t *a, int64_t *b) {
  const int64_t base = total / world, rem = total % world;
  *a = r * base + (r < rem ? r : rem);
  *b = *a + base + (r < rem ? 1 : 0);
}

extern "C" int pta_gather_rank0(void *comm, int rank, int world, int dst, const double *local, int64_t total_rows, int64_t n_cols,
                                int64_t ld_local, double *out, int64_t ld_out, void *stream) {
  PTA_REQUIRE(world >= 1 && rank >= 0 && rank < world && dst >= 0 && dst < world, PTA_E_ARG, "pta_gather_rank0: rank=%d world=%d dst=%d", rank,
              world, dst);
  PTA_REQUIRE(total_rows >= 0 && n_cols > 0 && ld_local >= n_cols, PTA_E_ARG, "pta_gather_rank0: total_rows=%lld n_cols=%lld", (long long)total_rows,
              (long long)n_cols);
  PTA_REQUIRE(rank != dst || (out && ld_out >= n_cols), PTA_E_ARG, "pta_gather_rank0: the destination rank needs `out`");
  hipStream_t s = pta_stream(stream);
  int64_t a, b;
  pta_shard(total_rows, rank, world, &a, &b);
  PTA_REQUIRE(b == a || local, PTA_E_ARG, "pta_gather_rank0: NULL local shard");
  if (rank == dst && b > a)  // own rows: one strided device-to-device copy
    PTA_HIP(hipMemcpy2DAsync(out + a * ld_out, ld_out * sizeof(double), local, ld_local * sizeof(double), n_cols * sizeof(double), b - a,
                             hipMemcpyDeviceToDevice, s));
  if (world == 1) return PTA_OK;
  PTA_REQUIRE(comm, PTA_E_ARG, "pta_gather_rank0: NULL communicator");
  PTA_REQUIRE(ld_local == n_cols && (rank != dst || ld_out == n_cols), PTA_E_ARG,
              "pta_gather_rank0: multi-rank gathers need contiguous rows (ld == n_cols)");
  PTA_REQUIRE(rccl_load() == PTA_OK, PTA_E_ARG, "pta_gather_rank0: RCCL (librccl.so) could not be loaded");
  // point-to-point: `dst` posts one receive per peer STRAIGHT into that peer's rows of `out` (no staging, no concatenation - the
  // destination holds the ensemble once); one group = one fused launch
  PTA_NCCL(g_rccl.gstart());
  int e = 0;  // first RCCL error inside the group; the group is CLOSED on every path (an open group would swallow the caller's next calls)
  if (rank == dst) {
    for (int r = 0; r < world && e == 0; ++r) {
      if (r == dst) continue;
      int64_t ra, rb;
      pta_shard(total_rows, r, world, &ra, &rb);
      if (rb > ra) e = g_rccl.recv(out + ra * ld_out, (size_t)((rb - ra) * n_cols), 8 /* ncclFloat64 */, r, comm, s);
    }
  } else if (b > a) {
    e = g_rccl.send(const_cast<double *>(local), (size_t)((b - a) * n_cols), 8, dst, comm, s);
  }
  const int e_end = g_rccl.gend();
  if (e == 0) e = e_end;
  if (e != 0) {
    pta_set_error("pta_gather_rank0: RCCL send / recv failed: %s", g_rccl.errstr ? g_rccl.errstr(e) : "rccl error");
    return PTA_E_HIP;
  }
  return PTA_OK;
}
