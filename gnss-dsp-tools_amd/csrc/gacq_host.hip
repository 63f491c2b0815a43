// Host-buffer batched entry points of the C ABI: what a non-torch caller needs to keep one GPU, or all GPUs of a node, busy
// from sample blocks that live in host memory.
//
//   gacq_search_batch        nepoch blocks -> chunks through a ring of pinned staging slots: the H2D copy of chunk c+1 runs on
//                            a copy stream while the kernels of chunk c execute; the last kernel writes its 16-byte peak records
//                            straight into device-visible pinned memory, so there is no D2H copy.  (stream.EpochStreamer is the
//                            torch form of the same pipeline.)
//   gacq_group_*             one process driving ndev GPUs: every device holds all code spectra; the Doppler grid is cut into
//                            contiguous slices, one per device (forward FFTs are not duplicated) -- or the item list when the grid
//                            is too coarse -- every device gets the same samples, and the per-device peaks are merged on the
//                            host in global Doppler order with strict '>' (gacq_finalize), the reference's tie rule
//                            (acquire-gps-l1.py:36-39).  The only cross-device traffic is 16 bytes per (epoch, item) and device.
//                            This replaces the reference's one-process-per-PRN Pool.map (acquire-gps-l1.py:105-108) for a
//                            single-process caller; multi-process callers use torch.distributed / RCCL (sharded.py).
#include "gacq_common.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>

using namespace gacq;

// RCCL, loaded on demand (gacq_group_set_exchange(group, GACQ_EXCHANGE_RCCL)): the library has no link-time dependency on it, a
// caller that never asks for the in-library collective never loads it.  Only the handful of entry points the one exchange step
// needs, declared here with the ABI of rccl.h (ncclResult_t and ncclDataType_t are ints; ncclChar == 0; comms are opaque pointers).
namespace {
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*CommAbort)(void* comm) = nullptr;      // optional: tears a communicator down with collectives still queued (error paths)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllGather)(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok() const { return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather && GetErrorString; }
};
Rccl* rccl_load() {
  static Rccl r;
  if (r.lib) return r.ok() ? &r : nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
  }
  if (!r.lib) return nullptr;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
  r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  return r.ok() ? &r : nullptr;
}
}  // namespace

namespace gacq {

constexpr int kRingDepth = 3;
constexpr size_t kChunkBytes = (size_t)32 << 20;      // samples per staging slot
constexpr int kChunkEpochsMin = 64, kChunkEpochsMax = 1024;      // a call is cut into ~4 chunks (the ring holds 3) of 64..1024 epochs:
                                                                 // enough in flight to hide the copies, launches big enough to batch well

// Staging ring of one device.  pin_in is owned by the ring (single-device calls) or by the group (shared by all members).
struct BatchRing {
  hipStream_t copy = nullptr;
  hipEvent_t h2d[kRingDepth] = {}, done[kRingDepth] = {};
  DevBuf pin_in[kRingDepth], dev_in[kRingDepth], pin_out[kRingDepth];
  DevBuf dev_out[kRingDepth], dev_all[kRingDepth];      // RCCL exchange of a device group: this member's records, every member's records
};

void ring_destroy(gacq_ctx* ctx) {
  BatchRing* r = ctx->ring;
  if (!r) return;
  for (int s = 0; s < kRingDepth; s++) {
    if (r->h2d[s]) (void)hipEventDestroy(r->h2d[s]);
    if (r->done[s]) (void)hipEventDestroy(r->done[s]);
    if (r->pin_in[s].p) (void)hipHostFree(r->pin_in[s].p);
    if (r->dev_in[s].p) (void)hipFree(r->dev_in[s].p);
    if (r->pin_out[s].p) (void)hipHostFree(r->pin_out[s].p);
    if (r->dev_out[s].p) (void)hipFree(r->dev_out[s].p);
    if (r->dev_all[s].p) (void)hipFree(r->dev_all[s].p);
  }
  if (r->copy) (void)hipStreamDestroy(r->copy);
  delete r;
  ctx->ring = nullptr;
}

}  // namespace gacq

namespace {

int ring_get(gacq_ctx* ctx, BatchRing** out) {
  if (!ctx->ring) {
    BatchRing* r = new BatchRing();
    ctx->ring = r;
    GACQ_HIP(ctx, hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking));
    for (int s = 0; s < kRingDepth; s++) {
      GACQ_HIP(ctx, hipEventCreateWithFlags(&r->h2d[s], hipEventDisableTiming));
      GACQ_HIP(ctx, hipEventCreateWithFlags(&r->done[s], hipEventDisableTiming));
    }
  }
  *out = ctx->ring;
  return GACQ_OK;
}

int epochs_per_chunk(size_t nsamp, int nepoch) {
  const size_t by_bytes = std::max<size_t>(1, kChunkBytes / std::max<size_t>(1, nsamp * sizeof(float2)));
  const size_t quarter = std::min<size_t>(kChunkEpochsMax, std::max<size_t>(kChunkEpochsMin, ((size_t)nepoch + 3) / 4));
  return (int)std::min<size_t>((size_t)nepoch, std::min<size_t>(by_bytes, quarter));
}

// Queue one chunk on `sig`'s device: H2D of ne*nsamp samples from pinned `src` on the copy stream, then the search of this
// device's (items, dopplers) share on the compute stream; peaks land in pin_out[slot].  The slot must have been collected.
// to_device: the records stay on the device (dev_out[slot]) for the RCCL exchange of a device group.
int ring_submit(gacq_sig* sig, BatchRing* r, int slot, const void* src, size_t nsamp, int ne, const int* items, int nitems,
                const double* dopplers, int nd, const double* bias, int blocks, bool to_device = false) {
  gacq_ctx* ctx = sig->ctx;
  GACQ_DEVICE(ctx);
  int rc;
  const size_t bytes = sizeof(float2) * nsamp * (size_t)ne;
  if ((rc = ensure(ctx, r->dev_in[slot], bytes)) != GACQ_OK) return rc;
  if (to_device) rc = ensure(ctx, r->dev_out[slot], sizeof(gacq_peak) * (size_t)ne * nitems);
  else rc = ensure_pinned(ctx, r->pin_out[slot], sizeof(gacq_peak) * (size_t)ne * nitems);
  if (rc != GACQ_OK) return rc;
  GACQ_HIP(ctx, hipMemcpyAsync(r->dev_in[slot].p, src, bytes, hipMemcpyHostToDevice, r->copy));
  GACQ_HIP(ctx, hipEventRecord(r->h2d[slot], r->copy));
  // one-way dependency: the compute stream waits for the copy, the copy stream never waits for compute (the host knows a slot
  // is free once its results were collected) -- a device-side wait in that direction cost 0.3 ms per batch
  GACQ_HIP(ctx, hipStreamWaitEvent(ctx->stream, r->h2d[slot], 0));
  rc = gacq_search_batch_dev(sig, r->dev_in[slot].p, nsamp, ne, items, nitems, dopplers, nd, bias, blocks,
                             to_device ? r->dev_out[slot].p : r->pin_out[slot].p);
  if (rc != GACQ_OK) return rc;
  if (!to_device) GACQ_HIP(ctx, hipEventRecord(r->done[slot], ctx->stream));      // (exchange mode: recorded after the collective and the merge)
  return GACQ_OK;
}

// Error paths: a call that fails after its first submit leaves chunks in flight on the copy and compute streams, and the next
// call would memcpy into pinned slots a pending DMA still reads.  Every failing exit drains both streams first.
void ring_drain(gacq_ctx* ctx, BatchRing* r) {
  gacq::DeviceGuard dg(ctx->device);
  if (r && r->copy) (void)hipStreamSynchronize(r->copy);
  (void)hipStreamSynchronize(ctx->stream);
}

int ring_collect(gacq_ctx* ctx, BatchRing* r, int slot) {
  GACQ_DEVICE(ctx);
  GACQ_HIP(ctx, hipEventSynchronize(r->done[slot]));
  return GACQ_OK;
}

void zero_results(gacq_result* out, size_t n) {
  for (size_t i = 0; i < n; i++) { out[i].metric = 0.0; out[i].code_chips = 0.0; out[i].doppler_hz = 0.0; out[i].idx = -1; out[i].d_index = -1; }
}

int check_batch_args(gacq_ctx* ctx, const void* sig, const float* x, size_t nsamp, int nepoch, const int* items, int nitems,
                     const double* dopplers, int nd, int blocks, const void* out, const gacq_sigdesc* desc) {
  if (!sig || !x || !items || !out || nitems <= 0 || nepoch <= 0 || nd < 0 || blocks < 0 || (nd > 0 && !dopplers))
    return set_error(ctx, GACQ_ERR_BAD_ARG, "batched search: bad argument");
  const size_t need = (size_t)(blocks + (desc->pad ? 1 : 0)) * desc->n;
  if (blocks > 0 && nd > 0 && nsamp < need)               // an empty Doppler grid never touches x (acquire-gps-l1.py:25-26,40)
    return set_error(ctx, GACQ_ERR_SHORT_INPUT, "batched search: %zu samples per epoch given, %zu needed for %d block(s) of n=%d%s", nsamp, need,
                     blocks, desc->n, desc->pad ? " (padded: windows span 2n)" : "");
  return GACQ_OK;
}

}  // namespace

extern "C" int gacq_search_batch(gacq_sig* sig, const float* x_iq, size_t nsamp, int nepoch, const int* items, int nitems,
                                 const double* dopplers, int nd, const double* item_bias_hz, int blocks, gacq_result* out) {
  gacq_ctx* ctx = sig ? sig->ctx : nullptr;
  int rc = check_batch_args(ctx, sig, x_iq, nsamp, nepoch, items, nitems, dopplers, nd, blocks, out, sig ? &sig->desc : nullptr);
  if (rc != GACQ_OK) return rc;
  if (nd == 0 || blocks == 0) { zero_results(out, (size_t)nepoch * nitems); return GACQ_OK; }      // acquire-gps-l1.py:25,40
  GACQ_DEVICE(ctx);
  BatchRing* r;
  if ((rc = ring_get(ctx, &r)) != GACQ_OK) return rc;
  const int Ec = epochs_per_chunk(nsamp, nepoch);
  const int nchunk = (nepoch + Ec - 1) / Ec;
  const float2* x = reinterpret_cast<const float2*>(x_iq);
  auto finish = [&](int c) -> int {
    const int slot = c % kRingDepth, e0 = c * Ec, ne = std::min(Ec, nepoch - e0);
    int rcf = ring_collect(ctx, r, slot);
    for (int e = 0; rcf == GACQ_OK && e < ne; e++)
      rcf = gacq_finalize(&sig->desc, (const gacq_peak*)r->pin_out[slot].p + (size_t)e * nitems, 1, nullptr, nitems, dopplers, nd,
                          out + (size_t)(e0 + e) * nitems);
    return rcf;
  };
  auto fail = [&](int code) -> int { ring_drain(ctx, r); return code; };      // nothing of this call stays in flight
  for (int c = 0; c < nchunk; c++) {
    const int slot = c % kRingDepth, e0 = c * Ec, ne = std::min(Ec, nepoch - e0);
    if (c >= kRingDepth && (rc = finish(c - kRingDepth)) != GACQ_OK) return fail(rc);
    const size_t bytes = sizeof(float2) * nsamp * (size_t)ne;
    if ((rc = ensure_pinned(ctx, r->pin_in[slot], bytes)) != GACQ_OK) return fail(rc);
    std::memcpy(r->pin_in[slot].p, x + (size_t)e0 * nsamp, bytes);         // pageable caller memory -> pinned slot (one CPU copy)
    if ((rc = ring_submit(sig, r, slot, r->pin_in[slot].p, nsamp, ne, items, nitems, dopplers, nd, item_bias_hz, blocks)) != GACQ_OK) return fail(rc);
  }
  for (int c = std::max(0, nchunk - kRingDepth); c < nchunk; c++)
    if ((rc = finish(c)) != GACQ_OK) return fail(rc);
  return GACQ_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// device group
// ------------------------------------------------------------------------------------------------------------------------
struct gacq_group {
  std::vector<gacq_ctx*> ctx;
  std::string err;
  DevBuf pin_in[kRingDepth];          // portable pinned staging shared by all members (every device DMAs from it)
  int exchange = GACQ_EXCHANGE_HOST;
  std::vector<void*> comm;            // one RCCL communicator per member (ncclCommInitAll), GACQ_EXCHANGE_RCCL only
};

struct gacq_gsig {
  gacq_group* group = nullptr;
  std::vector<gacq_sig*> sig;         // one per member, same PRN list
  gacq_sigdesc desc{};
};

namespace {

int group_error(gacq_group* g, int code, const char* msg) {
  if (g) g->err = msg ? msg : "";
  return code;
}

int group_fail(gacq_group* g, int k, int rc) {            // member k's message becomes the group's
  const char* m = gacq_last_error(g->ctx[k]);
  g->err = "device member " + std::to_string(k) + ": " + (m ? m : "");
  return rc;
}

}  // namespace

extern "C" {

int gacq_group_create(const int* device_ids, int ndev, gacq_group** out) {
  if (!out || !device_ids || ndev <= 0 || ndev > 64) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_group_create: bad argument");
  *out = nullptr;
  gacq_group* g = new gacq_group();
  for (int k = 0; k < ndev; k++) {
    gacq_ctx* c = nullptr;
    const int rc = gacq_create(device_ids[k], &c);           // the same device may appear twice (two contexts on it)
    if (rc != GACQ_OK) { gacq_group_destroy(g); return rc; }
    g->ctx.push_back(c);
  }
  *out = g;
  return GACQ_OK;
}

void gacq_group_destroy(gacq_group* g) {
  if (!g) return;
  if (!g->comm.empty()) {
    Rccl* nc = rccl_load();
    for (size_t k = 0; k < g->comm.size(); k++) {
      DeviceGuard dg(g->ctx[k]->device);
      (void)hipStreamSynchronize(g->ctx[k]->stream);
      if (nc && g->comm[k]) (void)nc->CommDestroy(g->comm[k]);
    }
  }
  for (gacq_ctx* c : g->ctx) gacq_destroy(c);
  for (DevBuf& b : g->pin_in) if (b.p) (void)hipHostFree(b.p);
  delete g;
}

int gacq_group_size(const gacq_group* g) { return g ? (int)g->ctx.size() : GACQ_ERR_BAD_ARG; }

int gacq_group_set_exchange(gacq_group* g, int mode) {
  if (!g || (mode != GACQ_EXCHANGE_HOST && mode != GACQ_EXCHANGE_RCCL)) return group_error(g, GACQ_ERR_BAD_ARG, "gacq_group_set_exchange: bad argument");
  if (mode == GACQ_EXCHANGE_RCCL && g->comm.empty()) {
    const int G = (int)g->ctx.size();
    std::vector<int> devs(G);
    for (int k = 0; k < G; k++) devs[k] = g->ctx[k]->device;
    for (int a = 0; a < G; a++)
      for (int b = a + 1; b < G; b++)
        if (devs[a] == devs[b]) return group_error(g, GACQ_ERR_UNSUPPORTED, "gacq_group_set_exchange: RCCL needs distinct devices (a device is listed twice)");
    Rccl* nc = rccl_load();
    if (!nc) return group_error(g, GACQ_ERR_UNSUPPORTED, "gacq_group_set_exchange: librccl.so could not be loaded");
    std::vector<void*> comm(G, nullptr);
    const int st = nc->CommInitAll(comm.data(), G, devs.data());
    if (st != 0) {
      // whatever communicators the failed call did create are torn down: none is left behind half-initialised
      for (void* cm : comm) if (cm) (void)(nc->CommAbort ? nc->CommAbort(cm) : nc->CommDestroy(cm));
      g->err = std::string("gacq_group_set_exchange: ncclCommInitAll failed: ") + nc->GetErrorString(st);
      return GACQ_ERR_HIP;
    }
    g->comm = comm;
  }
  g->exchange = mode;
  return GACQ_OK;
}

gacq_ctx* gacq_group_member(gacq_group* g, int k) { return (g && k >= 0 && k < (int)g->ctx.size()) ? g->ctx[k] : nullptr; }

const char* gacq_group_last_error(gacq_group* g) { return g ? g->err.c_str() : gacq_last_error(nullptr); }

int gacq_group_signal_create(gacq_group* g, const gacq_sigdesc* desc, const char* code, const int* prns, int nprn, gacq_gsig** out) {
  if (!g || !out || !desc) return group_error(g, GACQ_ERR_BAD_ARG, "gacq_group_signal_create: bad argument");
  *out = nullptr;
  gacq_gsig* s = new gacq_gsig();
  s->group = g;
  s->desc = *desc;
  for (size_t k = 0; k < g->ctx.size(); k++) {
    gacq_sig* one = nullptr;
    const int rc = gacq_signal_create(g->ctx[k], desc, code, prns, nprn, &one);
    if (rc != GACQ_OK) { group_fail(g, (int)k, rc); gacq_group_signal_destroy(s); return rc; }
    s->sig.push_back(one);
  }
  *out = s;
  return GACQ_OK;
}

void gacq_group_signal_destroy(gacq_gsig* s) {
  if (!s) return;
  for (gacq_sig* one : s->sig) gacq_signal_destroy(one);
  delete s;
}

int gacq_group_search_batch(gacq_gsig* s, const float* x_iq, size_t nsamp, int nepoch, const int* items, int nitems,
                            const double* dopplers, int nd, const double* item_bias_hz, int blocks, gacq_result* out) {
  gacq_group* g = s ? s->group : nullptr;
  if (!g) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_group_search_batch: signal is NULL");
  int rc = check_batch_args(g->ctx[0], s, x_iq, nsamp, nepoch, items, nitems, dopplers, nd, blocks, out, &s->desc);
  if (rc != GACQ_OK) return group_fail(g, 0, rc);
  if (nd == 0 || blocks == 0) { zero_results(out, (size_t)nepoch * nitems); return GACQ_OK; }
  const int G = (int)g->ctx.size();
  // cut the Doppler grid while every device still gets >= 4 bins, otherwise cut the item list
  const bool by_doppler = (G == 1) || nd >= 4 * G || nitems < G;
  std::vector<int> lo(G + 1);
  for (int k = 0; k <= G; k++) lo[k] = (int)(((long)k * (by_doppler ? nd : nitems)) / G);
  std::vector<BatchRing*> ring(G);
  for (int k = 0; k < G; k++) {
    DeviceGuard dg(g->ctx[k]->device);
    if ((rc = ring_get(g->ctx[k], &ring[k])) != GACQ_OK) return group_fail(g, k, rc);
  }
  const int Ec = epochs_per_chunk(nsamp, nepoch);
  const int nchunk = (nepoch + Ec - 1) / Ec;
  const float2* x = reinterpret_cast<const float2*>(x_iq);
  std::vector<gacq_peak> shards;
  auto share = [&](int k, const int*& it, int& nit, const double*& dop, int& ndk, const double*& bias) {
    if (by_doppler) { it = items; nit = nitems; dop = dopplers + lo[k]; ndk = lo[k + 1] - lo[k]; bias = item_bias_hz; }
    else { it = items + lo[k]; nit = lo[k + 1] - lo[k]; dop = dopplers; ndk = nd; bias = item_bias_hz ? item_bias_hz + lo[k] : nullptr; }
  };
  // GACQ_EXCHANGE_RCCL (Doppler slices only: an item split needs no merge): every member's records stay on its device, ONE
  // ncclAllGather per chunk moves them over xGMI, the first member with a slice merges them on the device -- tie-safe, from the samples
  // it holds in its staging slot -- and writes the merged records into its pinned result slot.  The north-star exchange step for a
  // caller without torch; GACQ_EXCHANGE_HOST merges the same records on the host out of pinned memory.
  const bool use_rccl = g->exchange == GACQ_EXCHANGE_RCCL && by_doppler && !g->comm.empty();
  int kmerge = 0;
  while (kmerge < G - 1 && lo[kmerge + 1] == lo[kmerge]) kmerge++;
  auto exchange = [&](int slot, int ne) -> int {
    Rccl* nc = rccl_load();
    if (!nc) return group_error(g, GACQ_ERR_UNSUPPORTED, "gacq_group_search_batch: librccl.so is gone");
    const size_t n = (size_t)ne * nitems;
    for (int k = 0; k < G; k++) {
      gacq_ctx* c = g->ctx[k];
      DeviceGuard dg(c->device);
      int rck = ensure(c, ring[k]->dev_all[slot], sizeof(gacq_peak) * n * G);
      if (rck == GACQ_OK && lo[k + 1] == lo[k]) {
        // a member without a slice sends "nothing found" records: idx = d_index = -1 in every second 8-byte word, metric 0
        rck = ensure(c, ring[k]->dev_out[slot], sizeof(gacq_peak) * n);
        if (rck == GACQ_OK && hipMemset2DAsync(ring[k]->dev_out[slot].p, 16, 0, 8, n, c->stream) != hipSuccess) rck = GACQ_ERR_HIP;
        if (rck == GACQ_OK && hipMemset2DAsync((char*)ring[k]->dev_out[slot].p + 8, 16, 0xff, 8, n, c->stream) != hipSuccess) rck = GACQ_ERR_HIP;
      }
      if (rck != GACQ_OK) return group_fail(g, k, rck);
    }
    int st = nc->GroupStart();
    for (int k = 0; k < G && st == 0; k++) {
      DeviceGuard dg(g->ctx[k]->device);
      st = nc->AllGather(ring[k]->dev_out[slot].p, ring[k]->dev_all[slot].p, sizeof(gacq_peak) * n, /*ncclChar*/ 0, g->comm[k], g->ctx[k]->stream);
    }
    const int st2 = nc->GroupEnd();
    if (st != 0 || st2 != 0) {
      // Some members' streams may now hold a collective whose peers never joined: draining those streams (what fail() does next) would
      // wait for ever.  The communicators are aborted -- queued collectives end -- and the group falls back to the host exchange; the
      // caller sees the error and can call again.
      for (void*& cm : g->comm) {
        if (cm) (void)(nc->CommAbort ? nc->CommAbort(cm) : nc->CommDestroy(cm));
        cm = nullptr;
      }
      g->comm.clear();
      g->exchange = GACQ_EXCHANGE_HOST;
      g->err = std::string("gacq_group_search_batch: ncclAllGather failed (communicators aborted, exchange back to GACQ_EXCHANGE_HOST): ") +
               nc->GetErrorString(st ? st : st2);
      return GACQ_ERR_HIP;
    }
    // the merge (on the member that holds a slice: it has the samples) + the completion events of every member
    gacq_ctx* c0 = g->ctx[kmerge];
    {
      DeviceGuard dg(c0->device);
      int rcm = ensure_pinned(c0, ring[kmerge]->pin_out[slot], sizeof(gacq_peak) * n);
      if (rcm == GACQ_OK)
        rcm = gacq_merge_peaks_tiesafe_dev(s->sig[kmerge], ring[kmerge]->dev_in[slot].p, nsamp, ne, items, nitems, dopplers, nd, item_bias_hz, blocks,
                                           ring[kmerge]->dev_all[slot].p, G, lo.data(), ring[kmerge]->pin_out[slot].p);
      if (rcm != GACQ_OK) return group_fail(g, kmerge, rcm);
    }
    for (int k = 0; k < G; k++) {
      DeviceGuard dg(g->ctx[k]->device);
      if (hipEventRecord(ring[k]->done[slot], g->ctx[k]->stream) != hipSuccess) return group_error(g, GACQ_ERR_HIP, "gacq_group_search_batch: event record failed");
    }
    return GACQ_OK;
  };
  auto finish = [&](int c) -> int {
    const int slot = c % kRingDepth, e0 = c * Ec, ne = std::min(Ec, nepoch - e0);
    for (int k = 0; k < G; k++) {
      if (lo[k + 1] == lo[k] && !use_rccl) continue;
      const int rck = ring_collect(g->ctx[k], ring[k], slot);
      if (rck != GACQ_OK) return group_fail(g, k, rck);
    }
    if (use_rccl) {
      const gacq_peak* merged = (const gacq_peak*)ring[kmerge]->pin_out[slot].p;
      for (int e = 0; e < ne; e++) {
        const int rcf = gacq_finalize(&s->desc, merged + (size_t)e * nitems, 1, nullptr, nitems, dopplers, nd, out + (size_t)(e0 + e) * nitems);
        if (rcf != GACQ_OK) return group_error(g, rcf, gacq_last_error(nullptr));
      }
      return GACQ_OK;
    }
    if (by_doppler && G > 1) {
      // Tie-safe locations across devices: device winners whose metrics come within eps of the best one cannot be ordered in fp32.
      // The host looks for such pairs (two compares per record); only when the chunk has one does a member re-evaluate them in
      // complex128 (gacq_merge_peaks_tiesafe_dev on the samples it still holds in its staging slot) and the merged records
      // replace the host merge below.
      int k0 = 0;
      while (k0 < G && lo[k0 + 1] == lo[k0]) k0++;
      gacq_ctx* c0 = g->ctx[k0];
      bool ambiguous = false;
      if (c0->opt[GACQ_OPT_TIE_SAFE] != 0 && c0->engine != 5 && tie_supported(s->sig[k0]->N)) {
        const double scale = (double)tie_scale_of(c0);
        for (long i = 0; i < (long)ne * nitems && !ambiguous; i++) {
          double best = 0.0, second = 0.0;
          for (int k = 0; k < G; k++) {
            if (lo[k + 1] == lo[k]) continue;
            const gacq_peak& pk = ((const gacq_peak*)ring[k]->pin_out[slot].p)[i];
            if (pk.d_index < 0) continue;
            if (pk.metric > best) { second = best; best = pk.metric; } else if (pk.metric > second) second = pk.metric;
          }
          ambiguous = best > 0.0 && second >= best * scale;
        }
      }
      if (ambiguous) {
        DeviceGuard dg(c0->device);
        const size_t n = (size_t)ne * nitems;
        std::vector<gacq_peak> all((size_t)G * n, gacq_peak{0.0, -1, -1});
        for (int k = 0; k < G; k++)
          if (lo[k + 1] > lo[k]) std::memcpy(&all[(size_t)k * n], ring[k]->pin_out[slot].p, sizeof(gacq_peak) * n);
        int rcm = ensure(c0, c0->chunk_peaks, sizeof(gacq_peak) * all.size());
        if (rcm == GACQ_OK) rcm = ensure(c0, c0->out_peaks, sizeof(gacq_peak) * n);
        if (rcm != GACQ_OK) return group_fail(g, k0, rcm);
        if (hipMemcpyAsync(c0->chunk_peaks.p, all.data(), sizeof(gacq_peak) * all.size(), hipMemcpyHostToDevice, c0->stream) != hipSuccess ||
            hipStreamSynchronize(c0->stream) != hipSuccess)
          return group_error(g, GACQ_ERR_HIP, "gacq_group_search_batch: upload of the device records failed");
        rcm = gacq_merge_peaks_tiesafe_dev(s->sig[k0], ring[k0]->dev_in[slot].p, nsamp, ne, items, nitems, dopplers, nd, item_bias_hz, blocks,
                                           c0->chunk_peaks.p, G, lo.data(), c0->out_peaks.p);
        if (rcm != GACQ_OK) return group_fail(g, k0, rcm);
        std::vector<gacq_peak> merged(n);
        if (hipMemcpyAsync(merged.data(), c0->out_peaks.p, sizeof(gacq_peak) * n, hipMemcpyDeviceToHost, c0->stream) != hipSuccess ||
            hipStreamSynchronize(c0->stream) != hipSuccess)
          return group_error(g, GACQ_ERR_HIP, "gacq_group_search_batch: download of the merged records failed");
        for (int e = 0; e < ne; e++) {
          const int rcf = gacq_finalize(&s->desc, merged.data() + (size_t)e * nitems, 1, nullptr, nitems, dopplers, nd, out + (size_t)(e0 + e) * nitems);
          if (rcf != GACQ_OK) return group_error(g, rcf, gacq_last_error(nullptr));
        }
        return GACQ_OK;
      }
    }
    for (int e = 0; e < ne; e++) {
      gacq_result* dst = out + (size_t)(e0 + e) * nitems;
      if (by_doppler) {
        // shard k covers Doppler indices [lo[k], lo[k+1]); an empty shard contributes "nothing found" records
        shards.assign((size_t)G * nitems, gacq_peak{0.0, -1, -1});
        for (int k = 0; k < G; k++)
          if (lo[k + 1] > lo[k])
            std::memcpy(&shards[(size_t)k * nitems], (const gacq_peak*)ring[k]->pin_out[slot].p + (size_t)e * nitems, sizeof(gacq_peak) * nitems);
        const int rcf = gacq_finalize(&s->desc, shards.data(), G, lo.data(), nitems, dopplers, nd, dst);
        if (rcf != GACQ_OK) return group_error(g, rcf, gacq_last_error(nullptr));
      } else {
        for (int k = 0; k < G; k++) {
          const int nit = lo[k + 1] - lo[k];
          if (!nit) continue;
          const int rcf = gacq_finalize(&s->desc, (const gacq_peak*)ring[k]->pin_out[slot].p + (size_t)e * nit, 1, nullptr, nit, dopplers, nd, dst + lo[k]);
          if (rcf != GACQ_OK) return group_error(g, rcf, gacq_last_error(nullptr));
        }
      }
    }
    return GACQ_OK;
  };
  auto fail = [&](int code) -> int {                    // every member drained: nothing of this call stays in flight anywhere
    for (int k = 0; k < G; k++) ring_drain(g->ctx[k], ring[k]);
    return code;
  };
  for (int c = 0; c < nchunk; c++) {
    const int slot = c % kRingDepth, e0 = c * Ec, ne = std::min(Ec, nepoch - e0);
    if (c >= kRingDepth && (rc = finish(c - kRingDepth)) != GACQ_OK) return fail(rc);
    const size_t bytes = sizeof(float2) * nsamp * (size_t)ne;
    if (bytes > g->pin_in[slot].cap) {
      if (g->pin_in[slot].p) { fail(0); (void)hipHostFree(g->pin_in[slot].p); g->pin_in[slot] = DevBuf(); }
      if (hipHostMalloc(&g->pin_in[slot].p, bytes, hipHostMallocPortable) != hipSuccess)
        return fail(group_error(g, GACQ_ERR_HIP, "gacq_group_search_batch: portable pinned staging allocation failed"));
      g->pin_in[slot].cap = bytes;
    }
    std::memcpy(g->pin_in[slot].p, x + (size_t)e0 * nsamp, bytes);
    for (int k = 0; k < G; k++) {                       // same samples to every device; each searches its share
      const int* it; int nit; const double* dop; int ndk; const double* bias;
      share(k, it, nit, dop, ndk, bias);
      if (nit == 0 || ndk == 0) continue;
      rc = ring_submit(s->sig[k], ring[k], slot, g->pin_in[slot].p, nsamp, ne, it, nit, dop, ndk, bias, blocks, use_rccl);
      if (rc != GACQ_OK) return fail(group_fail(g, k, rc));
    }
    if (use_rccl && (rc = exchange(slot, ne)) != GACQ_OK) return fail(rc);
  }
  for (int c = std::max(0, nchunk - kRingDepth); c < nchunk; c++)
    if ((rc = finish(c)) != GACQ_OK) return fail(rc);
  return GACQ_OK;
}

}  // extern "C"
