// lfx_gather.hip -- the only exchange step of the path: the RCCL gather of the labelled clouds.
#include "lfx_internal.hpp"

#include <rccl/rccl.h>      // types and prototypes only: librccl is opened at run time (lfx_comm_*), not linked
#include <dlfcn.h>

using namespace lfx_host;

// ---------------------------------------------------------------------------- multi-GPU gather over RCCL
namespace
{
struct Rccl
{
  void * lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
};

Rccl * rccl()
{
  static Rccl r;
  if (r.lib || !r.why.empty()) {return &r;}
  // the copy already in the process (PyTorch ships one under the same SONAME) or the ROCm installation's
#ifdef LFX_TEST_HOOKS
  // Test build only (liblfx_testhooks.so, `make testhooks`; the shipped library has no such door): LFX_RCCL_LIB names
  // another library with the same nine entry points -- tests/shim/rccl_shim.cpp lets several processes that share ONE
  // GPU run the N > 1 exchange, which a real RCCL communicator refuses.
  if (const char * named = std::getenv("LFX_RCCL_LIB")) {
    r.lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) {r.why = std::string("cannot open LFX_RCCL_LIB: ") + dlerror(); return &r;}
    std::fprintf(stderr, "liblfx (test hooks): nccl* entry points taken from %s\n", named);
  }
#endif
  for (const char * name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (r.lib) {break;}
    r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!r.lib) {r.why = std::string("cannot open librccl: ") + dlerror(); return &r;}
  bool ok = true;
  auto sym = [&](const char * n) {void * p = dlsym(r.lib, n); if (!p) {ok = false; r.why = std::string("librccl lacks ") + n;} return p;};
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
  r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) {dlclose(r.lib); r.lib = nullptr;}
  return &r;
}
}  // namespace

struct lfx_comm
{
  lfx_ctx * ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  // two slots of totals: two steps' counts may be out at once (lfx_gather_payload2 posts two steps as one group)
  struct Counts
  {
    uint32_t * d_mine = nullptr;      // [2] this rank's totals
    uint32_t * d_totals = nullptr;    // [world][2]
    uint32_t * h_totals = nullptr;    // pinned [world][2]
    hipEvent_t landed = nullptr;      // the totals are in h_totals
    bool pending = false;
  } counts[LFX_GATHER_SLOTS];
  uint64_t stats[LFX_COMM_STATS] = {};   // sends posted, receives posted, bytes sent, bytes received, all-gathers
};

#define LFX_NCCL(ctx, call) \
  do { \
    const ncclResult_t r_ = (call); \
    if (r_ != ncclSuccess) { \
      (ctx)->err = std::string(#call) + ": " + rccl()->GetErrorString(r_); \
      return LFX_ERR_HIP; \
    } \
  } while (0)

extern "C" {

int lfx_comm_unique_id(uint8_t id[LFX_COMM_ID_BYTES])
{
  static_assert(sizeof(ncclUniqueId) == LFX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id) {return LFX_ERR_INVALID_ARGUMENT;}
  Rccl * r = rccl();
  if (!r->lib) {create_error() = r->why; return LFX_ERR_NO_DEVICE;}
  ncclUniqueId u;
  if (r->GetUniqueId(&u) != ncclSuccess) {create_error() = "ncclGetUniqueId failed"; return LFX_ERR_HIP;}
  std::memcpy(id, &u, LFX_COMM_ID_BYTES);
  return LFX_OK;
}

int lfx_comm_create(lfx_ctx * c, const uint8_t id[LFX_COMM_ID_BYTES], int rank, int world, lfx_comm ** out)
{
  if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) {return LFX_ERR_INVALID_ARGUMENT;}
  *out = nullptr;
  Rccl * r = rccl();
  if (!r->lib) {return fail(c, LFX_ERR_NO_DEVICE, r->why);}
  LFX_HIP(c, hipSetDevice(c->device));
  lfx_comm * m = new lfx_comm();
  m->ctx = c; m->rank = rank; m->world = world;
  ncclUniqueId u;
  std::memcpy(&u, id, LFX_COMM_ID_BYTES);
  const ncclResult_t nr = r->CommInitRank(&m->comm, world, u, rank);
  if (nr != ncclSuccess) {
    c->err = std::string("ncclCommInitRank: ") + r->GetErrorString(nr);
    delete m;
    return LFX_ERR_HIP;
  }
  hipError_t e = hipSuccess;
  for (auto & k : m->counts) {
    if (e == hipSuccess) {e = hipMalloc(reinterpret_cast<void **>(&k.d_mine), 16);}
    if (e == hipSuccess) {e = hipMalloc(reinterpret_cast<void **>(&k.d_totals), (size_t)world * 8 + 16);}
    if (e == hipSuccess) {e = hipHostMalloc(reinterpret_cast<void **>(&k.h_totals), (size_t)world * 8 + 16, hipHostMallocDefault);}
    if (e == hipSuccess) {e = hipEventCreateWithFlags(&k.landed, hipEventDisableTiming);}
  }
  if (e != hipSuccess) {
    c->err = std::string("lfx_comm_create: ") + hipGetErrorString(e);
    lfx_comm_destroy(m);
    return LFX_ERR_HIP;
  }
  *out = m;
  return LFX_OK;
}

void lfx_comm_destroy(lfx_comm * m)
{
  if (!m) {return;}
  if (m->comm && rccl()->lib) {(void)rccl()->CommDestroy(m->comm);}
  for (auto & k : m->counts) {
    if (k.d_mine) {(void)hipFree(k.d_mine);}
    if (k.d_totals) {(void)hipFree(k.d_totals);}
    if (k.h_totals) {(void)hipHostFree(k.h_totals);}
    if (k.landed) {(void)hipEventDestroy(k.landed);}
  }
  delete m;
}

int lfx_gather_counts_slot(lfx_ctx * c, lfx_comm * m, uint32_t slot, const uint32_t * d_offsets, uint32_t batch, void * stream)
{
  if (!c || !m || !d_offsets || batch == 0 || slot >= LFX_GATHER_SLOTS) {return LFX_ERR_INVALID_ARGUMENT;}
  Rccl * r = rccl();
  hipStream_t st = static_cast<hipStream_t>(stream);
  lfx_comm::Counts & k = m->counts[slot];
  LFX_HIP(c, hipSetDevice(c->device));
  // totals: d_offsets[batch] (edge) and d_offsets[2 * batch + 1] (surface)
  LFX_HIP(c, hipMemcpyAsync(k.d_mine, d_offsets + batch, 4, hipMemcpyDeviceToDevice, st));
  LFX_HIP(c, hipMemcpyAsync(k.d_mine + 1, d_offsets + 2 * batch + 1, 4, hipMemcpyDeviceToDevice, st));
  LFX_NCCL(c, r->AllGather(k.d_mine, k.d_totals, 2, ncclUint32, m->comm, st));
  m->stats[4]++;
  LFX_HIP(c, hipMemcpyAsync(k.h_totals, k.d_totals, (size_t)m->world * 8, hipMemcpyDeviceToHost, st));
  LFX_HIP(c, hipEventRecord(k.landed, st));
  k.pending = true;
  return LFX_OK;
}

int lfx_gather_counts(lfx_ctx * c, lfx_comm * m, const uint32_t * d_offsets, uint32_t batch, void * stream)
{
  return lfx_gather_counts_slot(c, m, 0u, d_offsets, batch, stream);
}

// One, or two steps' exchanges as ONE group on the communicator.  Every rank takes every decision from the same totals and
// the same capacity_points, so nobody is left waiting in a send or a receive that the other side never posts.
int lfx_gather_payload2(
  lfx_ctx * c, lfx_comm * m, const lfx_gather_step * steps, uint32_t n_steps, uint32_t batch, uint32_t fpp, size_t capacity_points,
  void * stream)
{
  if (!c || !m || !steps || n_steps < 1 || n_steps > LFX_GATHER_SLOTS || batch == 0 || (fpp != 3 && fpp != 4)) {return LFX_ERR_INVALID_ARGUMENT;}
  for (uint32_t i = 0; i < n_steps; i++) {
    const lfx_gather_step & g = steps[i];
    if (!g.d_edge || !g.d_surface || !g.d_offsets || g.dst < 0 || g.dst >= m->world || g.slot >= LFX_GATHER_SLOTS) {return LFX_ERR_INVALID_ARGUMENT;}
    if (m->rank == g.dst && (!g.d_edge_all || !g.d_surface_all || !g.d_offsets_all)) {return LFX_ERR_INVALID_ARGUMENT;}
    if (i == 1 && g.slot == steps[0].slot) {return LFX_ERR_INVALID_ARGUMENT;}
    if (!m->counts[g.slot].pending) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "lfx_gather_payload without lfx_gather_counts");}
  }
  Rccl * r = rccl();
  hipStream_t st = static_cast<hipStream_t>(stream);
  LFX_HIP(c, hipSetDevice(c->device));
  bool fits = true;
  for (uint32_t i = 0; i < n_steps; i++) {
    const lfx_gather_step & g = steps[i];
    lfx_comm::Counts & k = m->counts[g.slot];
    LFX_HIP(c, hipEventSynchronize(k.landed));
    k.pending = false;
    uint64_t sum_e = 0, sum_s = 0;
    for (int q = 0; q < m->world; q++) {
      if (g.counts_out) {g.counts_out[2 * q] = k.h_totals[2 * q]; g.counts_out[2 * q + 1] = k.h_totals[2 * q + 1];}
      sum_e += k.h_totals[2 * q];
      sum_s += k.h_totals[2 * q + 1];
    }
    if (sum_e > capacity_points || sum_s > capacity_points) {fits = false;}
  }
  if (!fits) {return fail(c, LFX_ERR_CAPACITY, "gathered clouds exceed capacity_points of the destination rank");}
  const size_t tab = 2 * ((size_t)batch + 1);
  LFX_NCCL(c, r->GroupStart());
  for (uint32_t i = 0; i < n_steps; i++) {
    const lfx_gather_step & g = steps[i];
    const uint32_t * tot = m->counts[g.slot].h_totals;
    const uint32_t me = tot[2 * m->rank], ms = tot[2 * m->rank + 1];
    if (m->rank != g.dst) {
      LFX_NCCL(c, r->Send(g.d_edge, (size_t)me * fpp, ncclFloat32, g.dst, m->comm, st));
      LFX_NCCL(c, r->Send(g.d_surface, (size_t)ms * fpp, ncclFloat32, g.dst, m->comm, st));
      LFX_NCCL(c, r->Send(g.d_offsets, tab, ncclUint32, g.dst, m->comm, st));
      m->stats[0] += 3;
      m->stats[2] += ((size_t)me + ms) * fpp * 4 + tab * 4;
      continue;
    }
    size_t at_e = 0, at_s = 0;
    for (int q = 0; q < m->world; q++) {
      const uint32_t ne = tot[2 * q], ns = tot[2 * q + 1];
      if (q != g.dst) {
        LFX_NCCL(c, r->Recv(g.d_edge_all + at_e * fpp, (size_t)ne * fpp, ncclFloat32, q, m->comm, st));
        LFX_NCCL(c, r->Recv(g.d_surface_all + at_s * fpp, (size_t)ns * fpp, ncclFloat32, q, m->comm, st));
        LFX_NCCL(c, r->Recv(g.d_offsets_all + (size_t)q * tab, tab, ncclUint32, q, m->comm, st));
        m->stats[1] += 3;
        m->stats[3] += ((size_t)ne + ns) * fpp * 4 + tab * 4;
      }
      at_e += ne;
      at_s += ns;
    }
  }
  LFX_NCCL(c, r->GroupEnd());
  // a destination's own part: plain device copies on the same stream
  for (uint32_t i = 0; i < n_steps; i++) {
    const lfx_gather_step & g = steps[i];
    if (m->rank != g.dst) {continue;}
    const uint32_t * tot = m->counts[g.slot].h_totals;
    const uint32_t me = tot[2 * m->rank], ms = tot[2 * m->rank + 1];
    size_t at_e = 0, at_s = 0;
    for (int q = 0; q < g.dst; q++) {at_e += tot[2 * q]; at_s += tot[2 * q + 1];}
    if (me) {LFX_HIP(c, hipMemcpyAsync(g.d_edge_all + at_e * fpp, g.d_edge, (size_t)me * fpp * 4, hipMemcpyDeviceToDevice, st));}
    if (ms) {LFX_HIP(c, hipMemcpyAsync(g.d_surface_all + at_s * fpp, g.d_surface, (size_t)ms * fpp * 4, hipMemcpyDeviceToDevice, st));}
    LFX_HIP(c, hipMemcpyAsync(g.d_offsets_all + (size_t)g.dst * tab, g.d_offsets, tab * 4, hipMemcpyDeviceToDevice, st));
  }
  return LFX_OK;
}

int lfx_gather_payload(
  lfx_ctx * c, lfx_comm * m, int dst, const float * d_edge, const float * d_surface, const uint32_t * d_offsets,
  uint32_t batch, uint32_t fpp, float * d_edge_all, float * d_surface_all, uint32_t * d_offsets_all, size_t capacity_points,
  uint64_t * counts_out, void * stream)
{
  if (!m) {return LFX_ERR_INVALID_ARGUMENT;}
  const lfx_gather_step g{dst, 0u, d_edge, d_surface, d_offsets, d_edge_all, d_surface_all, d_offsets_all, counts_out};
  return lfx_gather_payload2(c, m, &g, 1u, batch, fpp, capacity_points, stream);
}

int lfx_comm_stats(const lfx_comm * m, uint64_t out[LFX_COMM_STATS])
{
  if (!m || !out) {return LFX_ERR_INVALID_ARGUMENT;}
  std::memcpy(out, m->stats, sizeof(m->stats));
  return LFX_OK;
}

int lfx_gather(
  lfx_ctx * c, lfx_comm * m, int dst, const float * d_edge, const float * d_surface, const uint32_t * d_offsets,
  uint32_t batch, uint32_t fpp, float * d_edge_all, float * d_surface_all, uint32_t * d_offsets_all, size_t capacity_points,
  uint64_t * counts_out, void * stream)
{
  const int rc = lfx_gather_counts(c, m, d_offsets, batch, stream);
  if (rc != LFX_OK) {return rc;}
  return lfx_gather_payload(c, m, dst, d_edge, d_surface, d_offsets, batch, fpp, d_edge_all, d_surface_all, d_offsets_all,
           capacity_points, counts_out, stream);
}

}  // extern "C"
