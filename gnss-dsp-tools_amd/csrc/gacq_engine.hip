// MI355X (gfx950) acquisition engine: C ABI + the rocFFT pipeline (any FFT length).
//
// Pipeline per batch of epochs (kernels K1..K4 around rocFFT):
//   K1 mix_nco_kernel      y[e,f,d,b,i] = x[e][b*n+i] * tab[floor(f_d*i*1024) & 1023]   acquire-gps-l1.py:28,30-31
//   F1 rocFFT forward      batched, in place                                             acquire-gps-l1.py:32 (fft.fft)
//   K2 conj_mul_kernel     Y[e,p,d,b,k] = C_p[k] * conj(X[e,f(p),d,b,k])                 acquire-gps-l1.py:32
//   F2 rocFFT inverse      batched, in place, unnormalised (1/N folded into K3)          acquire-gps-l1.py:32 (fft.ifft)
//   K3 mag_peak_kernel     q[k] = sum_b |r_b[k]|/N ; (max, first argmax, sum) per row    acquire-gps-l1.py:33-35
//   K4 best_doppler_kernel strict-'>' scan over Doppler bins in order                    acquire-gps-l1.py:36-39
// For power-of-two lengths that fit in LDS the K1+F1 and K2+F2+K3 groups are replaced by the two
// fused LDS-resident FFT kernels in gacq_ldsfft.hip (engine 2).
#include "gacq_common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <mutex>

using namespace gacq;

namespace {
// spin-wait hint and store fence of the host CPU (the BAR / watch path of gacq_search); portable no-ops elsewhere
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#endif
}
inline void cpu_store_fence() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_sfence();
#else
  std::atomic_thread_fence(std::memory_order_seq_cst);
#endif
}
constexpr size_t kPinnedStageMax = 1u << 20;     // host inputs up to 1 MiB are staged through pinned memory
constexpr size_t kBarWriteMax = 1u << 18;        // host inputs up to 256 KiB are written straight into device memory through the BAR
thread_local std::string g_last_error;     // last error of calls made without a ctx, per calling thread
std::once_flag g_rocfft_once;
const char* kStageNames[GACQ_NSTAGES] = {"mix_nco", "rocfft_forward", "conj_mul", "rocfft_inverse",
                                         "mag_peak", "best_doppler", "lds_correlate"};
}  // namespace

namespace gacq {

int set_error(gacq_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  if (ctx) ctx->err = buf;
  return code;
}

// Latency-path upload: the host writes `bytes` straight into fine-grained device memory through the PCIe BAR (large-BAR devices; one
// write-combined memcpy, ~1 us for 32 KB, no staging copy, no DMA engine) and fences, so that the launch doorbell that follows is
// ordered behind the stores.  Returns false when the device (or this allocation) cannot be written that way -- the caller then
// stages through pinned memory.  The caller guarantees that no kernel still reads the buffer (synchronous entry points).
bool bar_write(gacq_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
  if (!ctx->large_bar || bytes > kBarWriteMax || !ctx->opt[GACQ_OPT_BAR_UPLOAD]) return false;
  if (bytes > b.cap) {
    if (b.p) {
      if (hipDeviceSynchronize() != hipSuccess || hipFree(b.p) != hipSuccess) { (void)hipGetLastError(); return false; }
      b.p = nullptr;
      b.cap = 0;
    }
    const size_t want = std::max<size_t>(bytes + bytes / 8, 65536);
    if (hipExtMallocWithFlags(&b.p, want, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      ctx->large_bar = false;            // fall back to the staged copy for good
      return false;
    }
    b.cap = want;
    // isLargeBar says device memory is host-addressable; that THIS allocation is mapped at the same address is checked once per
    // allocation: the host writes a word through the BAR, the runtime copies it back from the device side
    volatile unsigned long long* probe = (volatile unsigned long long*)b.p;
    const unsigned long long token = 0x6761637170726f62ULL;
    unsigned long long back = 0;
    *probe = token;
    cpu_store_fence();
    if (!(hipMemcpy(&back, b.p, 8, hipMemcpyDeviceToHost) == hipSuccess && back == token)) {
      (void)hipGetLastError();
      (void)hipFree(b.p);
      b.p = nullptr;
      b.cap = 0;
      ctx->large_bar = false;
      return false;
    }
  }
  std::memcpy(b.p, src, bytes);
  cpu_store_fence();
  return true;
}

// Completion by watching: the last kernel of a call writes `count` records of `stride` 8-byte words into device-visible pinned
// memory whose word `at` the host pre-set to `sentinel`; true once no sentinel is left, false after `timeout_us` (the caller then
// synchronises the stream, which is also where a faulted launch is reported).
bool watch_records(const volatile unsigned long long* words, size_t count, size_t stride, size_t at, unsigned long long sentinel, int timeout_us) {
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(timeout_us);
  size_t first_pending = 0;
  for (unsigned spin = 0;; spin++) {
    while (first_pending < count && words[first_pending * stride + at] != sentinel) first_pending++;
    if (first_pending == count) { std::atomic_thread_fence(std::memory_order_acquire); return true; }
    cpu_relax();
    if ((spin & 63) == 63 && std::chrono::steady_clock::now() > t_end) return false;
  }
}

int ensure(gacq_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return GACQ_OK;
  if (b.p) {
    // device-wide: the buffer may still be read by work queued on a stream the ctx used before gacq_set_stream
    GACQ_HIP(ctx, hipDeviceSynchronize());
    GACQ_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 8;
  if (hipMalloc(&b.p, want) != hipSuccess) {
    (void)hipGetLastError();
    want = bytes;
    const hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      ctx->alloc_failed = true;            // search_batch_dev retries in smaller passes
      return set_error(ctx, GACQ_ERR_HIP, "hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
    }
  }
  b.cap = want;
  return GACQ_OK;
}

// contexts alive per device (gacq_create / gacq_destroy): they share its memory
static std::atomic<int> g_ctx_on_device[64];

size_t ws_budget(gacq_ctx* ctx) {
  if (!ctx->ws_explicit) {
    const int nctx = std::max(1, g_ctx_on_device[ctx->device & 63].load());
    if (ctx->ws_soft == 0 || nctx > ctx->ws_soft_nctx) {
      // A context holds up to two buffers of this size (forward spectra X, correlation workspace Y), each with 1/8 of headroom.  What is
      // free now plus what this context already holds, a fifth of it left to everything else on the device, split evenly over the
      // contexts: 102 GB for a lone context on an empty 288 GB part (the 32 GiB limit binds), 13 GB each for eight ranks sharing one.
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        ctx->ws_soft = std::max<size_t>((size_t)64 << 20, (size_t)((double)(free_b + ctx->X.cap + ctx->Y.cap) * 0.8 / 2.25 / nctx));
      else
        ctx->ws_soft = ctx->ws_limit;
      ctx->ws_soft_nctx = nctx;
    }
  } else if (ctx->ws_soft == 0) {
    ctx->ws_soft = ctx->ws_limit;
  }
  return std::min(ctx->ws_limit, ctx->ws_soft);
}

int ensure_pinned(gacq_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return GACQ_OK;
  if (b.p) {
    GACQ_HIP(ctx, hipDeviceSynchronize());
    GACQ_HIP(ctx, hipHostFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t want = std::max<size_t>(bytes + bytes / 8, 4096);
  GACQ_HIP(ctx, hipHostMalloc(&b.p, want, hipHostMallocDefault));
  b.cap = want;
  return GACQ_OK;
}

int table_cache(gacq_ctx* ctx, const std::string& key, const void* host, size_t bytes, const void** out) {
  auto it = ctx->tables.find(key);
  if (it == ctx->tables.end()) {
    DevBuf b;
    GACQ_HIP(ctx, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    if (hipMemcpy(b.p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(b.p);
      return set_error(ctx, GACQ_ERR_HIP, "upload of table '%s' failed", key.c_str());
    }
    it = ctx->tables.emplace(key, b).first;
  }
  *out = it->second.p;
  return GACQ_OK;
}

int twiddle_cache(gacq_ctx* ctx, const std::string& key, int N, int count, const float2** out) {
  auto it = ctx->tables.find(key);
  if (it != ctx->tables.end()) { *out = (const float2*)it->second.p; return GACQ_OK; }
  std::vector<float2> h(count);
  for (int k = 0; k < count; k++) {
    const double a = -2.0 * M_PI * (double)k / (double)N;
    h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  const void* p = nullptr;
  int rc = table_cache(ctx, key, h.data(), sizeof(float2) * (size_t)count, &p);
  *out = (const float2*)p;
  return rc;
}

int fft_exec(gacq_ctx* ctx, int N, long batch, bool inverse, void* data, bool fp64) {
  auto key = std::make_pair((long)N * 4 + (inverse ? 1 : 0) + (fp64 ? 2 : 0), batch);
  auto it = ctx->plans.find(key);
  if (it == ctx->plans.end()) {
    FftPlan p;
    size_t len[1] = {(size_t)N};
    GACQ_FFT(ctx, rocfft_plan_create(&p.plan, rocfft_placement_inplace,
                                     inverse ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward,
                                     fp64 ? rocfft_precision_double : rocfft_precision_single, 1, len, (size_t)batch, nullptr));
    GACQ_FFT(ctx, rocfft_execution_info_create(&p.info));
    GACQ_FFT(ctx, rocfft_plan_get_work_buffer_size(p.plan, &p.work_size));
    if (p.work_size) {
      GACQ_HIP(ctx, hipMalloc(&p.work, p.work_size));
      GACQ_FFT(ctx, rocfft_execution_info_set_work_buffer(p.info, p.work, p.work_size));
    }
    it = ctx->plans.emplace(key, p).first;
  }
  FftPlan& p = it->second;
  GACQ_FFT(ctx, rocfft_execution_info_set_stream(p.info, ctx->stream));
  void* bufs[1] = {data};
  GACQ_FFT(ctx, rocfft_execute(p.plan, bufs, nullptr, p.info));
  return GACQ_OK;
}

void stage_begin(gacq_ctx* ctx, int stage) {
  if (!ctx->profiling) return;
  StageEvent ev;
  ev.stage = stage;
  if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return;
  (void)hipEventRecord(ev.a, ctx->stream);
  ctx->pending.push_back(ev);
}

void stage_end(gacq_ctx* ctx) {
  if (!ctx->profiling || ctx->pending.empty()) return;
  (void)hipEventRecord(ctx->pending.back().b, ctx->stream);
}

}  // namespace gacq

// ------------------------------------------------------------------------------------------------
// Kernels of the rocFFT pipeline.  All are pure streaming kernels (HBM-bound): 16 B per lane,
// 256-thread workgroups, one flattened 1-D grid (row-major over rows x chunks).
// ------------------------------------------------------------------------------------------------

// K1: carrier wipe-off with the reference's 10-bit phase-quantised table NCO.
// The table index is computed in fp64 exactly like numpy does: floor((0 + f*i) * 1024) mod 1024
// (gnsstools/nco.py:6-9); one v_mul_f64 per sample is noise next to the HBM traffic.
// DUMP (test hook gacq_debug_nco_indices): the same index expression on the same frequency table, stored as int32 instead of
// being used for the table lookup; x is not read.
template <bool DUMP>
__global__ __launch_bounds__(kBlock) void mix_nco_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                          float2* __restrict__ y, const double* __restrict__ freq,
                                                          const float2* __restrict__ tab, int n, int span, int FD, int B,
                                                          int chunks) {
  __shared__ float2 s_tab[kNcoTableSize];
  for (int i = threadIdx.x; i < kNcoTableSize; i += kBlock) s_tab[i] = tab[i];
  __syncthreads();
  const unsigned blk = blockIdx.x;          // 32-bit index math (64-bit divisions are ~100 scalar ops each)
  const int chunk = (int)(blk % (unsigned)chunks);
  const unsigned row = blk / (unsigned)chunks;            // ((e*FD + fd)*B + b)
  const int b = (int)(row % (unsigned)B);
  const unsigned t = row / (unsigned)B;
  const int fd = (int)(t % (unsigned)FD);
  const long e = t / (unsigned)FD;
  const double f = freq[fd];
  const float2* src = x + e * epoch_stride + (size_t)b * n;
  float2* dst = y + row * (long)span;
  // each thread: 8 samples as 4 x (2 complex = 16 B)
  const int base = chunk * (kBlock * 8);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i0 = base + (j * kBlock + threadIdx.x) * 2;
    if (DUMP) {
      int* out = reinterpret_cast<int*>(y) + row * (long)span;
      if (i0 < span) out[i0] = nco_index(f, (int)i0);
      if (i0 + 1 < span) out[i0 + 1] = nco_index(f, (int)(i0 + 1));
      continue;
    }
    if (i0 + 1 < span) {
      const float4 s = *reinterpret_cast<const float4*>(src + i0);
      const int k0 = nco_index(f, (int)i0);
      const int k1 = nco_index(f, (int)(i0 + 1));
      const float2 w0 = s_tab[k0], w1 = s_tab[k1];
      float4 o;
      o.x = s.x * w0.x - s.y * w0.y;
      o.y = s.x * w0.y + s.y * w0.x;
      o.z = s.z * w1.x - s.w * w1.y;
      o.w = s.z * w1.y + s.w * w1.x;
      *reinterpret_cast<float4*>(dst + i0) = o;
    } else if (i0 < span) {
      const float2 s = src[i0];
      const int k0 = nco_index(f, (int)i0);
      const float2 w0 = s_tab[k0];
      dst[i0] = make_float2(s.x * w0.x - s.y * w0.y, s.x * w0.y + s.y * w0.x);
    }
  }
}

// K2: Y = C_p * conj(X).  group = (e*P + p)*D + d, row_y = (group - g0)*B + b.
__global__ __launch_bounds__(kBlock) void conj_mul_kernel(const float2* __restrict__ X, const float2* __restrict__ C,
                                                           float2* __restrict__ Y, const int* __restrict__ items,
                                                           const int* __restrict__ fset, long g0, int P, int F, int D,
                                                           int B, int N, int chunks) {
  const unsigned blk = blockIdx.x;
  const int chunk = (int)(blk % (unsigned)chunks);
  const unsigned ry = blk / (unsigned)chunks;
  const int b = (int)(ry % (unsigned)B);
  const unsigned g = (unsigned)g0 + ry / (unsigned)B;          // E*P*D < 2^31 (checked by the launcher)
  const int d = (int)(g % (unsigned)D);
  const unsigned ep = g / (unsigned)D;
  const int p = (int)(ep % (unsigned)P);
  const long e = ep / (unsigned)P;
  const float2* xs = X + (((e * F + fset[p]) * D + d) * (long)B + b) * (long)N;
  const float2* cs = C + (long)items[p] * N;
  float2* ys = Y + ry * (long)N;
  const int base = chunk * (kBlock * 8);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i0 = base + (j * kBlock + threadIdx.x) * 2;
    if (i0 + 1 < N) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + i0);
      const float4 cv = *reinterpret_cast<const float4*>(cs + i0);
      float4 o;
      o.x = cv.x * xv.x + cv.y * xv.y;
      o.y = cv.y * xv.x - cv.x * xv.y;
      o.z = cv.z * xv.z + cv.w * xv.w;
      o.w = cv.w * xv.z - cv.z * xv.w;
      *reinterpret_cast<float4*>(ys + i0) = o;
    } else if (i0 < N) {
      const float2 xv = xs[i0], cv = cs[i0];
      ys[i0] = make_float2(cv.x * xv.x + cv.y * xv.y, cv.y * xv.x - cv.x * xv.y);
    }
  }
}

// Block-wide (peak, first index, sum) reduction shared by K3 and the LDS engine.
__device__ __forceinline__ void block_reduce_peak(Top2& top, double& sum) {
  __shared__ float s_peak[kBlock / 64], s_second[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  __shared__ double s_sum[kBlock / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float op = __shfl_down(top.peak, off);
    const int oi = __shfl_down(top.idx, off);
    const float o2 = __shfl_down(top.second, off);
    const double os = __shfl_down(sum, off);
    top.merge(op, oi, o2);
    sum += os;
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_peak[wave] = top.peak; s_idx[wave] = top.idx; s_second[wave] = top.second; s_sum[wave] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      top.merge(s_peak[w], s_idx[w], s_second[w]);
      sum += s_sum[w];
    }
  }
}

// K3: one workgroup per (e,p,d) group: q[k] = sum_b |Y[b][k]| / N, reduced to (max, argmax, sum).
__global__ __launch_bounds__(kBlock) void mag_peak_kernel(const float2* __restrict__ Y, RowRec* __restrict__ rows,
                                                           long g0, int B, int N, float inv_n, float* __restrict__ q_out, float tie_scale) {
  const long gl = blockIdx.x;
  const float2* ys = Y + gl * (long)B * N;
  Top2 top;
  double sum = 0.0;
  for (int k0 = threadIdx.x * 2; k0 < N; k0 += kBlock * 2) {
    float q0 = 0.f, q1 = 0.f;
    if (k0 + 1 < N) {
      for (int b = 0; b < B; b++) {
        const float4 v = *reinterpret_cast<const float4*>(ys + (long)b * N + k0);
        q0 += sqrtf(v.x * v.x + v.y * v.y) * inv_n;
        q1 += sqrtf(v.z * v.z + v.w * v.w) * inv_n;
      }
      if (q_out) { q_out[k0] = q0; q_out[k0 + 1] = q1; }
      top.add(q0, k0);
      top.add(q1, k0 + 1);
      sum += (double)q0 + (double)q1;
    } else {
      for (int b = 0; b < B; b++) {
        const float2 v = ys[(long)b * N + k0];
        q0 += sqrtf(v.x * v.x + v.y * v.y) * inv_n;
      }
      if (q_out) q_out[k0] = q0;
      top.add(q0, k0);
      sum += (double)q0;
    }
  }
  block_reduce_peak(top, sum);
  if (threadIdx.x == 0) {
    RowRec r;
    r.peak = top.peak;
    r.idx = top.tagged(tie_scale);
    r.sum = sum;
    rows[g0 + gl] = r;
  }
}

// K4: per (e,p): scan Doppler bins in order, strict '>' against the running best that starts at 0.
// One wave per (epoch, item): lane l scans bins l, l+64, ... in ascending order with strict '>', then the 64 lane results
// are combined keeping the larger metric and, on equal metrics, the lower bin -- the same winner as the serial scan of
// acquire-gps-l1.py:36-39 (initial (0,0,0): a row that never exceeds 0 reports idx = d_index = -1, mapped to 0 by finalize).
// TIE (tie-safe locations, gacq_tiesafe.hip): the scan also keeps the runner-up metric.  When it comes within eps of the winner, or
// the winning row's own runner-up lag does (kTieBit of its idx), fp32 cannot tell which candidate the reference's fp64 scan picks:
// the pair's record is NOT written here; instead every bin whose metric reaches (1 - eps) x the winner is appended, in ascending
// Doppler order, to the re-evaluation list, and tie_resolve_kernel writes the record from the complex128 values.
__device__ __forceinline__ double row_metric(const RowRec& r, int N, int normalised) {
  return normalised ? (double)r.peak / (r.sum / (double)N) : (double)r.peak;
}
template <bool TIE>
__global__ __launch_bounds__(256) void best_doppler_kernel(const RowRec* __restrict__ rows, gacq_peak* __restrict__ out, long nep,
                                                           int D, int N, int normalised, TieLists tl, gacq_peak* __restrict__ guesses,
                                                           float tie_scale) {
  const long ep = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ep >= nep) return;
  const int lane = threadIdx.x & 63;
  double best = 0.0, second = 0.0;
  int bidx = -1, bd = 0x7fffffff;
  for (int d = lane; d < D; d += 64) {
    const RowRec r = rows[ep * D + d];
    const double m = row_metric(r, N, normalised);
    if (m > best) { second = best; best = m; bidx = r.idx; bd = d; }
    else if (TIE && m > second) second = m;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double om = __shfl_down(best, off);
    const int oi = __shfl_down(bidx, off);
    const int od = __shfl_down(bd, off);
    if (TIE) second = fmax(fmax(second, __shfl_down(second, off)), fmin(best, om));
    if (om > best || (om == best && od < bd)) { best = om; bidx = oi; bd = od; }
  }
  gacq_peak o;
  o.metric = best;
  o.idx = bidx < 0 ? -1 : (bidx & kIdxMask);
  o.d_index = bidx < 0 ? -1 : bd;
  if (!TIE) {
    if (lane == 0) out[ep] = o;
    return;
  }
  best = __shfl(best, 0);
  second = __shfl(second, 0);
  bidx = __shfl(bidx, 0);
  const double thr = best * (double)tie_scale;
  if (bidx < 0 || !((bidx & kTieBit) || second >= thr)) {
    if (lane == 0) out[ep] = o;
    return;
  }
  // ambiguous: list every candidate bin (wave-uniform control flow from here on)
  int cnt = 0;
  for (int d0 = 0; d0 < D; d0 += 64) {
    const int d = d0 + lane;
    const bool c = d < D && row_metric(rows[ep * D + d], N, normalised) >= thr;
    cnt += __popcll(__builtin_amdgcn_ballot_w64(c));
  }
  int slot0 = 0, epslot = 0;
  if (lane == 0) slot0 = (int)atomicAdd(&tl.c->nrows, (unsigned)cnt);
  slot0 = __shfl(slot0, 0);
  if (slot0 + cnt > tl.cap || slot0 < 0) {
    // no room: the pair keeps its fp32 answer; the slots reserved below the capacity are voided
    for (int s = slot0 + lane; s < slot0 + cnt && s < tl.cap && s >= 0; s += 64) { TieRow v; v.ep = -1; v.d = 0; tl.rows[s] = v; }
    if (lane == 0) {
      *(volatile unsigned*)tl.host_full = 1u;      // reaches the host before the record does (gacq_search watches the records)
      __threadfence_system();
      out[ep] = o;
      atomicAdd(&tl.c->overflow, 1ull);
    }
    return;
  }
  if (lane == 0) {
    epslot = (int)atomicAdd(&tl.c->neps, 1u);      // <= the number of listed rows <= cap
    TieEp te;
    te.ep = (int)ep; te.slot0 = slot0; te.cnt = cnt; te.pad = 0;
    tl.eps[epslot] = te;
    guesses[epslot] = o;
  }
  int off = 0;
  for (int d0 = 0; d0 < D; d0 += 64) {
    const int d = d0 + lane;
    const bool c = d < D && row_metric(rows[ep * D + d], N, normalised) >= thr;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(c);
    if (c) { TieRow v; v.ep = (int)ep; v.d = d; tl.rows[slot0 + off + __popcll(mask & ((1ull << lane) - 1ull))] = v; }
    off += __popcll(mask);
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int gacq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

const char* gacq_last_error(gacq_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

const char* gacq_stage_name(int stage) { return (stage >= 0 && stage < GACQ_NSTAGES) ? kStageNames[stage] : nullptr; }

int gacq_create(int device_id, gacq_ctx** out) {
  if (!out) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_create: out is NULL");
  *out = nullptr;
  int ndev = gacq_device_count();
  if (ndev <= 0) return set_error(nullptr, GACQ_ERR_NO_DEVICE, "gacq_create: no HIP device visible (this engine has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_create: device %d out of range [0,%d)", device_id, ndev);
  std::call_once(g_rocfft_once, [] { rocfft_setup(); });
  gacq_ctx* ctx = new gacq_ctx();
  ctx->device = device_id;
  DeviceGuard device_guard_(device_id);
  if (!device_guard_.ok || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return set_error(nullptr, GACQ_ERR_HIP, "gacq_create: cannot create stream on device %d", device_id);
  }
  ctx->stream = ctx->own_stream;
  g_ctx_on_device[device_id & 63]++;
  {
    hipDeviceProp_t prop;
    ctx->large_bar = hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.isLargeBar != 0;
  }
  // NCO phasor table: np.exp(2*pi*1j*k/1024) evaluated in fp64, rounded once to fp32 (gnsstools/nco.py:4)
  std::vector<float2> tab(kNcoTableSize);
  for (int k = 0; k < kNcoTableSize; k++) {
    const double a = 2.0 * M_PI * (double)k * (1.0 / kNcoTableSize);
    tab[k] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  int rc = ensure(ctx, ctx->tab, sizeof(float2) * kNcoTableSize);
  if (rc == GACQ_OK && hipMemcpy(ctx->tab.p, tab.data(), sizeof(float2) * kNcoTableSize, hipMemcpyHostToDevice) != hipSuccess)
    rc = set_error(nullptr, GACQ_ERR_HIP, "gacq_create: table upload failed");
  if (rc != GACQ_OK) { gacq_destroy(ctx); return rc; }
  *out = ctx;
  return GACQ_OK;
}

void gacq_destroy(gacq_ctx* ctx) {
  if (!ctx) return;
  if (ctx->own_stream) g_ctx_on_device[ctx->device & 63]--;
  DeviceGuard device_guard_(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ring_destroy(ctx);
  for (auto& kv : ctx->plans) {
    if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
    if (kv.second.plan) rocfft_plan_destroy(kv.second.plan);
    if (kv.second.work) (void)hipFree(kv.second.work);
  }
  for (auto& ev : ctx->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  DevBuf* bufs[] = {&ctx->tab, &ctx->xstage, &ctx->x32, &ctx->X, &ctx->Y, &ctx->rows, &ctx->freq, &ctx->fset, &ctx->items, &ctx->out_peaks, &ctx->d0, &ctx->partial, &ctx->fe_a, &ctx->fe_b, &ctx->fe_taps, &ctx->chunk_peaks, &ctx->tie, &ctx->tie_scratch, &ctx->tie_q, &ctx->tie_split, &ctx->tie_done2, &ctx->acq_in, &ctx->acq_x};
  for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
  if (ctx->pin_x.p) (void)hipHostFree(ctx->pin_x.p);
  if (ctx->pin_peaks.p) (void)hipHostFree(ctx->pin_peaks.p);
  if (ctx->pin_tie.p) (void)hipHostFree(ctx->pin_tie.p);
  if (ctx->bar_x.p) (void)hipFree(ctx->bar_x.p);
  if (ctx->bar_s.p) (void)hipFree(ctx->bar_s.p);
  for (auto& sl : ctx->grid_slots) for (DevBuf* b : {&sl.dfreq, &sl.dfset, &sl.ditems}) if (b->p) (void)hipFree(b->p);
  for (auto& kv : ctx->tables) if (kv.second.p) (void)hipFree(kv.second.p);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int gacq_set_stream(gacq_ctx* ctx, void* hip_stream) {
  if (!ctx) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_set_stream: ctx is NULL");
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return GACQ_OK;
}

int gacq_use_null_stream(gacq_ctx* ctx) {
  if (!ctx) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_use_null_stream: ctx is NULL");
  ctx->stream = (hipStream_t) nullptr;
  return GACQ_OK;
}

int gacq_set_engine(gacq_ctx* ctx, int engine) {
  if (!ctx || engine < 0 || engine > 5) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_set_engine: engine must be 0..5");
  ctx->engine = engine;
  return GACQ_OK;
}

int gacq_set_workspace_limit(gacq_ctx* ctx, size_t bytes) {
  if (!ctx || bytes < ((size_t)1 << 20)) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_set_workspace_limit: need >= 1 MiB");
  ctx->ws_limit = bytes;
  ctx->ws_explicit = true;
  ctx->ws_soft = 0;
  return GACQ_OK;
}

int gacq_set_option(gacq_ctx* ctx, int option, long value) {
  if (!ctx || option < 0 || option >= GACQ_NOPTS) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_set_option: unknown option %d", option);
  // accepted range per option (GACQ_OPT_* order): switches 0/1(/2), counts bounded by what the kernels' index arithmetic carries
  static const long kMax[GACQ_NOPTS] = {1, 1, 1000, 4096, 4096, 4, 2, 3, 1, 64, 1, 1, 1, 1, 1000000000L, 1L << 24, 1, 1};
  const long lo = (option == GACQ_OPT_LDS_VARIANT) ? -1 : 0;
  if (value < lo || value > kMax[option])
    return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_set_option: value %ld out of range [%ld, %ld] for option %d", value, lo, kMax[option], option);
  ctx->opt[option] = value;
  return GACQ_OK;
}

int gacq_get_option(gacq_ctx* ctx, int option, long* value) {
  if (!ctx || !value || option < 0 || option >= GACQ_NOPTS) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_get_option: unknown option %d", option);
  *value = ctx->opt[option];
  return GACQ_OK;
}

int gacq_set_profiling(gacq_ctx* ctx, int enabled) {
  if (!ctx) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_set_profiling: ctx is NULL");
  ctx->profiling = enabled != 0;
  return GACQ_OK;
}

static int drain_events(gacq_ctx* ctx) {
  if (ctx->pending.empty()) return GACQ_OK;
  GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& ev : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
      ctx->stage_ms[ev.stage] += ms;
      ctx->stage_n[ev.stage] += 1;
    } else {
      (void)hipGetLastError();
    }
    (void)hipEventDestroy(ev.a);
    (void)hipEventDestroy(ev.b);
  }
  ctx->pending.clear();
  return GACQ_OK;
}

int gacq_get_stage_time(gacq_ctx* ctx, int stage, double* total_ms, long* launches) {
  if (!ctx || stage < 0 || stage >= GACQ_NSTAGES) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_get_stage_time: bad stage");
  int rc = drain_events(ctx);
  if (rc != GACQ_OK) return rc;
  if (total_ms) *total_ms = ctx->stage_ms[stage];
  if (launches) *launches = ctx->stage_n[stage];
  return GACQ_OK;
}

int gacq_reset_stage_times(gacq_ctx* ctx) {
  if (!ctx) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_reset_stage_times: ctx is NULL");
  int rc = drain_events(ctx);
  for (int i = 0; i < GACQ_NSTAGES; i++) { ctx->stage_ms[i] = 0; ctx->stage_n[i] = 0; }
  return rc;
}

// ---- signals ------------------------------------------------------------------------------------
static int validate_desc(gacq_ctx* ctx, const gacq_sigdesc* d) {
  if (!d) return set_error(ctx, GACQ_ERR_BAD_ARG, "signal descriptor is NULL");
  if (d->code_length <= 0 || d->n <= 0 || !(d->fs > 0.0)) return set_error(ctx, GACQ_ERR_BAD_ARG, "signal descriptor: code_length, n and fs must be positive");
  if ((long)d->n * 2 > (1L << 28)) return set_error(ctx, GACQ_ERR_BAD_ARG, "signal descriptor: n too large");
  return GACQ_OK;
}

// the replicas as complex rows, zero-extended to N when padded (acquire-beidou-b1i.py:24), on the device
static int upload_replica_rows(gacq_sig* s, float2* dst) {
  gacq_ctx* ctx = s->ctx;
  std::vector<float2> host((size_t)s->nprn * s->N, make_float2(0.f, 0.f));
  for (int p = 0; p < s->nprn; p++)
    for (int i = 0; i < s->desc.n; i++) host[(size_t)p * s->N + i].x = s->replica[(size_t)p * s->desc.n + i];
  GACQ_HIP(ctx, hipMemcpyAsync(dst, host.data(), sizeof(float2) * host.size(), hipMemcpyHostToDevice, ctx->stream));
  GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));      // `host` lives on this frame
  return GACQ_OK;
}

// Code spectra in NATURAL order by rocFFT (c = fft.fft(c), acquire-gps-l1.py:24): what the rocFFT pipeline (engine 1) multiplies with and what
// gacq_signal_spectrum hands out.  Built on first use: creating a rocFFT plan for a new length costs 0.5-2 s per process (runtime-compiled
// kernels), and the default engines never need one -- their code spectra come out of their own forward transforms (build_signal).
static int natural_spectra(gacq_sig* s) {
  if (s->spectra) return GACQ_OK;
  gacq_ctx* ctx = s->ctx;
  const size_t bytes = sizeof(float2) * (size_t)s->nprn * s->N;
  float2* buf = nullptr;
  if (hipMalloc((void**)&buf, bytes) != hipSuccess) return set_error(ctx, GACQ_ERR_HIP, "hipMalloc of %zu bytes for code spectra failed", bytes);
  int rc = upload_replica_rows(s, buf);
  if (rc == GACQ_OK) rc = fft_exec(ctx, s->N, s->nprn, false, buf);
  if (rc == GACQ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "code spectrum FFT failed");
  if (rc != GACQ_OK) { (void)hipFree(buf); return rc; }
  s->spectra = buf;
  return GACQ_OK;
}

// ... and in the [k1][k2] order of the Cooley-Tukey form of engine 3 with rocFFT inner transforms (GACQ_OPT_FUSED_INNER = 0): on first use
static int ct_spectra(gacq_sig* s) {
  if (s->spectra_r31) return GACQ_OK;
  gacq_ctx* ctx = s->ctx;
  const size_t bytes = sizeof(float2) * (size_t)s->nprn * s->N;
  float2 *buf = nullptr, *tmp = nullptr;
  if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&tmp, bytes) != hipSuccess) {
    if (buf) (void)hipFree(buf);
    return set_error(ctx, GACQ_ERR_HIP, "hipMalloc for radix-31 spectra failed");
  }
  int rc = upload_replica_rows(s, tmp);
  if (rc == GACQ_OK) rc = split_forward(ctx, tmp, 0, s->nprn, s->N, s->N, nullptr, 1, 1, nullptr, buf, false);
  if (rc == GACQ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "split-engine code spectrum failed");
  (void)hipFree(tmp);
  if (rc != GACQ_OK) { (void)hipFree(buf); return rc; }
  s->spectra_r31 = buf;
  return GACQ_OK;
}

static int build_signal(gacq_ctx* ctx, const gacq_sigdesc* desc, const std::vector<float>& replicas, int nprn, gacq_sig** out) {
  GACQ_DEVICE(ctx);
  gacq_sig* s = new gacq_sig();
  s->ctx = ctx;
  s->desc = *desc;
  s->nprn = nprn;
  s->N = desc->pad ? 2 * desc->n : desc->n;
  s->replica = replicas;
  const size_t bytes = sizeof(float2) * (size_t)nprn * s->N;
  int rc = GACQ_OK;
  // Every LDS-resident / split engine keeps the code spectra in the order ITS forward transform produces: the replicas go through
  // that transform (no rocFFT plan is created here; natural_spectra() / ct_spectra() make the rocFFT-based forms when something asks).
  const bool own = split_supported(s->N) || lds_supported(s->N);
  float2* tmp = nullptr;
  if (own) {
    if (hipMalloc((void**)&tmp, bytes) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "hipMalloc of %zu bytes for the replicas failed", bytes);
    if (rc == GACQ_OK) rc = upload_replica_rows(s, tmp);
  }
  if (rc == GACQ_OK && split_supported(s->N)) {
    if (s->N / split_radix(s->N) == 4096) {
      // split engine with LDS inner transforms (N = R*4096): code spectra as R lane-pair rows per item
      if (hipMalloc((void**)&s->spectra_split, bytes) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "hipMalloc for split/LDS spectra failed");
      if (rc == GACQ_OK) rc = split_forward(ctx, tmp, 0, nprn, s->N, s->N, nullptr, 1, 1, nullptr, s->spectra_split, false, false);
      if (rc == GACQ_OK) rc = lds_inner_forward(ctx, s->spectra_split, (long)nprn * split_radix(s->N), false);
    }
    if (rc == GACQ_OK && pfa_supported(s->N)) {
      // prime-factor engine: the spectrum order is whatever its forward kernels produce -- run the replicas through them
      if (hipMalloc((void**)&s->spectra_pfa, bytes) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "hipMalloc for prime-factor spectra failed");
      if (rc == GACQ_OK) rc = pfa_forward(ctx, tmp, 0, nprn, s->N, s->N, nullptr, 1, 1, nullptr, s->spectra_pfa, false);
    }
  }
  if (rc == GACQ_OK && lds_supported(s->N)) {
    if (hipMalloc((void**)&s->spectra_lds, bytes) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "hipMalloc for LDS-layout spectra failed");
    else rc = lds_code_spectra(ctx, tmp, s->spectra_lds, nprn, s->N);
    if (rc == GACQ_OK && s->N == 16384) {      // the radix-16 form of the N = 16384 transform has its own spectrum order (GACQ_OPT_LDS_VARIANT = 16)
      if (hipMalloc((void**)&s->spectra_lds16, bytes) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "hipMalloc for LDS-layout spectra failed");
      else rc = lds_code_spectra(ctx, tmp, s->spectra_lds16, nprn, s->N, true);
    }
  }
  if (rc == GACQ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "code spectrum transform failed");
  if (tmp) (void)hipFree(tmp);
  if (rc == GACQ_OK && !own) rc = natural_spectra(s);      // lengths only the rocFFT pipeline serves
  if (rc != GACQ_OK) { if (s->spectra) (void)hipFree(s->spectra); if (s->spectra_lds) (void)hipFree(s->spectra_lds); if (s->spectra_lds16) (void)hipFree(s->spectra_lds16); if (s->spectra_pfa) (void)hipFree(s->spectra_pfa); if (s->spectra_split) (void)hipFree(s->spectra_split); delete s; return rc; }
  *out = s;
  return GACQ_OK;
}

int gacq_signal_create(gacq_ctx* ctx, const gacq_sigdesc* desc, const char* code, const int* prns, int nprn, gacq_sig** out) {
  if (!ctx || !out || !code || !prns || nprn <= 0) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_signal_create: bad argument");
  *out = nullptr;
  int rc = validate_desc(ctx, desc);
  if (rc != GACQ_OK) return rc;
  const int L = gacq_code_length(code);
  if (L < 0) return set_error(ctx, GACQ_ERR_UNKNOWN_CODE, "gacq_signal_create: unknown code '%s'", code);
  if (L != desc->code_length) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_signal_create: descriptor code_length %d != %d of '%s'", desc->code_length, L, code);
  std::vector<float> rep((size_t)nprn * desc->n);
  for (int p = 0; p < nprn; p++) {
    rc = gacq_code_replica(code, prns[p], desc->n, desc->boc, rep.data() + (size_t)p * desc->n);
    if (rc < 0) return set_error(ctx, rc, "gacq_signal_create: no PRN %d in '%s'", prns[p], code);
  }
  return build_signal(ctx, desc, rep, nprn, out);
}

int gacq_signal_create_chips(gacq_ctx* ctx, const gacq_sigdesc* desc, const uint8_t* chips, int nprn, gacq_sig** out) {
  if (!ctx || !out || !chips || nprn <= 0) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_signal_create_chips: bad argument");
  *out = nullptr;
  int rc = validate_desc(ctx, desc);
  if (rc != GACQ_OK) return rc;
  const int L = desc->code_length, n = desc->n;
  const double incr = (double)L / (double)n;
  std::vector<float> rep((size_t)nprn * n);
  for (int p = 0; p < nprn; p++)
    for (int i = 0; i < n; i++) {
      const double pos = incr * (double)i;
      float v = 1.0f - 2.0f * (float)(chips[(size_t)p * L + ((long)std::floor(pos) % L)] & 1);
      if (desc->boc && ((long)std::floor(pos * 2.0) % 2) == 0) v = -v;
      rep[(size_t)p * n + i] = v;
    }
  return build_signal(ctx, desc, rep, nprn, out);
}

void gacq_signal_destroy(gacq_sig* sig) {
  if (!sig) return;
  DeviceGuard device_guard_(sig->ctx->device);
  (void)hipStreamSynchronize(sig->ctx->stream);
  if (sig->spectra) (void)hipFree(sig->spectra);
  if (sig->spectra_lds) (void)hipFree(sig->spectra_lds);
  if (sig->spectra_lds16) (void)hipFree(sig->spectra_lds16);
  if (sig->spectra_r31) (void)hipFree(sig->spectra_r31);
  if (sig->spectra_pfa) (void)hipFree(sig->spectra_pfa);
  if (sig->spectra_split) (void)hipFree(sig->spectra_split);
  if (sig->spectra64) (void)hipFree(sig->spectra64);
  if (sig->spectra64_split) (void)hipFree(sig->spectra64_split);
  delete sig;
}

int gacq_signal_fft_length(const gacq_sig* sig) { return sig ? sig->N : GACQ_ERR_BAD_ARG; }

int gacq_signal_spectrum(gacq_sig* sig, int item, float* out_iq) {
  if (!sig || !out_iq || item < 0 || item >= sig->nprn) return set_error(sig ? sig->ctx : nullptr, GACQ_ERR_BAD_ARG, "gacq_signal_spectrum: bad argument");
  gacq_ctx* ctx = sig->ctx;
  GACQ_DEVICE(ctx);
  GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  { const int rcs = natural_spectra(sig); if (rcs != GACQ_OK) return rcs; }
  GACQ_HIP(ctx, hipMemcpy(out_iq, sig->spectra + (size_t)item * sig->N, sizeof(float2) * sig->N, hipMemcpyDeviceToHost));
  return GACQ_OK;
}

}  // extern "C"

// ---- search ---------------------------------------------------------------------------------------
namespace {

struct Grid {
  int F = 0;                      // distinct forward sets (carrier biases)
  std::vector<int> fset;          // per item
  std::vector<double> freq;       // [F][D] NCO frequency in cycles/sample
};

// f = -(bias + doppler)/fs, evaluated in fp64 in the reference's operation order
// (acquire-gps-l1.py:28, acquire-glonass-l1.py:28)
Grid make_grid(const gacq_sigdesc& d, int nitems, const double* dopplers, int nd, const double* bias) {
  Grid g;
  std::vector<double> biases;
  g.fset.resize(nitems);
  for (int p = 0; p < nitems; p++) {
    const double b = bias ? bias[p] : 0.0;
    int f = -1;
    for (size_t k = 0; k < biases.size(); k++) if (biases[k] == b) { f = (int)k; break; }
    if (f < 0) { f = (int)biases.size(); biases.push_back(b); }
    g.fset[p] = f;
  }
  g.F = (int)biases.size();
  g.freq.resize((size_t)g.F * nd);
  for (int f = 0; f < g.F; f++)
    for (int k = 0; k < nd; k++) {
      const double dop = dopplers[k];
      g.freq[(size_t)f * nd + k] = (biases[f] != 0.0) ? -(biases[f] + dop) / d.fs : -dop / d.fs;
    }
  return g;
}

// Which kernels of the LDS-resident engine serve a search -- decided in ONE place for the launch sequence (launch_search) and for
// the Doppler-slicing decision of gacq_search_batch_dev, which must know whether a forward-spectra buffer exists at all.
struct LdsPath {
  bool use_lds = false;      // whole transform in one workgroup (engine 2, or auto where the length is supported)
  bool fused16k = false;     // N = 16384, one carrier per item: forward + correlate in one kernel
  bool fused4k = false;      // N = 4096, B = 1, one carrier: forward + correlate in one kernel
  bool no_forward_buffer() const { return fused16k || fused4k; }
};
LdsPath lds_path(const gacq_ctx* ctx, int N, int nepoch, int P, int F, int D, int B, bool row_dump) {
  LdsPath p;
  p.use_lds = (ctx->engine == 2) || (ctx->engine == 0 && lds_supported(N) && !row_dump);
  if (!p.use_lds || !lds_supported(N) || row_dump) return p;      // a row dump (engine 2) goes through the two-kernel path
  p.fused16k = lds_fused_supported(ctx, N, P, F);
  p.fused4k = !p.fused16k && lds_fused4k_supported(ctx, N, B, F, (long)nepoch * D);
  return p;
}

// NCO frequencies [F][D], forward-set index and item index per position -> ctx->freq / fset / items on the device (skipped when the
// call repeats the grid uploaded last, as batched loops do)
int upload_grid(gacq_sig* sig, int nepoch, const int* items, int nitems, const double* dopplers, int nd, const double* bias, int* F_out) {
  gacq_ctx* ctx = sig->ctx;
  const gacq_sigdesc& ds = sig->desc;
  const int N = sig->N, D = nd, P = nitems;
  hipStream_t st = ctx->stream;
  Grid g = make_grid(ds, nitems, dopplers, nd, bias);
  const int F = g.F;
  *F_out = F;
  int rc;
  double max_f = 0.0;
  for (double v : g.freq) max_f = std::max(max_f, std::fabs(v));
  if ((long)nepoch * P * D * std::max(1, F) >= (1L << 31))
    return set_error(ctx, GACQ_ERR_UNSUPPORTED, "%d epochs x %d items x %d Doppler bins exceeds the 2^31 rows one call can index", nepoch, P, D);
  if (!nco_range_ok(max_f, N))
    return set_error(ctx, GACQ_ERR_UNSUPPORTED, "NCO frequency %.3g cycles/sample x %d samples exceeds the 32-bit phase-index range", max_f, N);
  const std::vector<int> items_key(items, items + P);
  if (!(g.freq == ctx->up_freq && g.fset == ctx->up_fset && items_key == ctx->up_items)) {
    // not the current grid: one of the earlier ones?  Its device buffers trade places with the current ones -- no upload, no
    // synchronisation; kernels still in flight keep reading the buffers they were given (nothing is freed or rewritten here)
    auto swap_in = [&](gacq_ctx::GridSlot& sl) {
      std::swap(sl.freq, ctx->up_freq); std::swap(sl.fset, ctx->up_fset); std::swap(sl.items, ctx->up_items);
      std::swap(sl.dfreq, ctx->freq); std::swap(sl.dfset, ctx->fset); std::swap(sl.ditems, ctx->items);
    };
    size_t hit = ctx->grid_slots.size();
    for (size_t k = 0; k < ctx->grid_slots.size(); k++)
      if (ctx->grid_slots[k].freq == g.freq && ctx->grid_slots[k].fset == g.fset && ctx->grid_slots[k].items == items_key) { hit = k; break; }
    if (hit < ctx->grid_slots.size()) {
      swap_in(ctx->grid_slots[hit]);
      std::rotate(ctx->grid_slots.begin(), ctx->grid_slots.begin() + hit, ctx->grid_slots.begin() + hit + 1);      // the displaced grid: most recent
    } else if (!ctx->up_freq.empty()) {
      // a new grid: the current one moves into the slots (the oldest slot's buffers are recycled for the upload below, which is
      // ordered behind every kernel that may still read them: same stream)
      if ((int)ctx->grid_slots.size() < gacq_ctx::kGridSlots) ctx->grid_slots.emplace(ctx->grid_slots.begin());
      else std::rotate(ctx->grid_slots.begin(), ctx->grid_slots.end() - 1, ctx->grid_slots.end());
      swap_in(ctx->grid_slots.front());
      ctx->up_freq.clear();            // whatever came back (empty, or the oldest grid) is about to be overwritten
    }
  }
  const void* before[3] = {ctx->freq.p, ctx->fset.p, ctx->items.p};
  if ((rc = ensure(ctx, ctx->freq, sizeof(double) * g.freq.size())) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->fset, sizeof(int) * P)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->items, sizeof(int) * P)) != GACQ_OK) return rc;
  const std::vector<int> items_v(items, items + P);
  if (before[0] != ctx->freq.p || before[1] != ctx->fset.p || before[2] != ctx->items.p) ctx->up_freq.clear();   // reallocated
  if (g.freq != ctx->up_freq || g.fset != ctx->up_fset || items_v != ctx->up_items) {
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->freq.p, g.freq.data(), sizeof(double) * g.freq.size(), hipMemcpyHostToDevice, st));
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->fset.p, g.fset.data(), sizeof(int) * P, hipMemcpyHostToDevice, st));
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->items.p, items, sizeof(int) * P, hipMemcpyHostToDevice, st));
    // the host vectors die with this frame; a repeated grid (batched loops) skips both copy and sync
    GACQ_HIP(ctx, hipStreamSynchronize(st));
    ctx->up_freq = g.freq;
    ctx->up_fset = g.fset;
    ctx->up_items = items_v;
  }
  return GACQ_OK;
}

// xs: the samples as the caller gave them (complex64 or complex128); d_x: their complex64 form for the fp32 engines (== xs.p for
// complex64 input, a rounded copy for complex128 input, unused by engine 5)
int launch_search(gacq_sig* sig, XSrc xs, const float2* d_x, size_t nsamp, int nepoch, const int* items, int nitems,
                  const double* dopplers, int nd, const double* bias, int blocks, gacq_peak* d_out, float* d_qrow) {
  gacq_ctx* ctx = sig->ctx;
  const gacq_sigdesc& ds = sig->desc;
  const int n = ds.n, N = sig->N, B = blocks, D = nd, P = nitems;
  hipStream_t st = ctx->stream;
  int F = 1;
  int rc = upload_grid(sig, nepoch, items, nitems, dopplers, nd, bias, &F);
  if (rc != GACQ_OK) return rc;

  if (ctx->engine == 5) return verify_search(sig, xs, nsamp, nepoch, P, F, D, B, d_out, d_qrow);      // complex128 engine: reads the samples as given

  const LdsPath path = lds_path(ctx, N, nepoch, P, F, D, B, d_qrow != nullptr);
  const bool use_lds = path.use_lds;
  if (ctx->engine == 2 && !lds_supported(N))
    return set_error(ctx, GACQ_ERR_UNSUPPORTED, "engine 2 (LDS FFT) does not support N=%d", N);
  const int R = split_radix(N);
  const bool split_lds_ok = R != 0 && N / R == 4096;
  const bool use_split_lds = (ctx->engine == 4) || (ctx->engine == 0 && split_lds_ok);
  if (ctx->engine == 4 && !split_lds_ok)
    return set_error(ctx, GACQ_ERR_UNSUPPORTED, "engine 4 (split with LDS inner transforms) does not support N=%d", N);
  const bool use_split = use_split_lds || (ctx->engine == 3) || (ctx->engine == 0 && split_supported(N));
  // N = 31 * 1980 / 31 * 990: the twiddle-free prime-factor form (gacq_pfa.hip) unless the caller asks for the Cooley-Tukey forms
  // form with rocFFT inner transforms (GACQ_OPT_FUSED_INNER 0: other arithmetic, the cross-check)
  const bool use_pfa = use_split && !use_split_lds && pfa_supported(N) && ctx->opt[GACQ_OPT_FUSED_INNER] != 0;
  if (ctx->engine == 3 && !split_supported(N))
    return set_error(ctx, GACQ_ERR_UNSUPPORTED, "engine 3 (split with rocFFT inner transforms) does not support N=%d", N);
  // the forms that multiply with rocFFT-made code spectra build them now, on first use (natural order / Cooley-Tukey [k1][k2] order)
  if (!use_lds && !use_split_lds && !use_pfa && (rc = use_split ? ct_spectra(sig) : natural_spectra(sig)) != GACQ_OK) return rc;

  // tie-safe locations: the reducers tag rows whose runner-up lag is within eps of the maximum, best_doppler_kernel lists the
  // (epoch, item) pairs it cannot decide and gacq_tiesafe.hip re-evaluates their candidate rows in complex128
  const bool tie = ctx->opt[GACQ_OPT_TIE_SAFE] != 0 && !d_qrow && tie_supported(N);
  const float tscale = tie ? tie_scale_of(ctx) : 1.0f;      // 1: only exact fp32 duplicates get tagged, and nobody looks
  TieLists tl{};
  gacq_peak* guesses = nullptr;
  if (tie) {
    if ((rc = tie_prepare(sig)) != GACQ_OK) return rc;
    if ((rc = tie_lists(ctx, (long)nepoch * P, N, B, &tl, &guesses)) != GACQ_OK) return rc;
  }

  // epochs per pass so that the forward-spectra buffer respects the workspace limit
  const bool fused16k = path.fused16k;      // one carrier per item: no forward-spectra buffer at all
  const size_t x_epoch_bytes = sizeof(float2) * (size_t)F * D * B * N;
  int Ec = (int)std::max<size_t>(1, std::min<size_t>((size_t)nepoch, ws_budget(ctx) / std::max<size_t>(1, x_epoch_bytes)));
  const bool fused4k = path.fused4k;
  if (fused16k || fused4k) Ec = nepoch;            // nothing but the 16-byte row records is buffered
  if (!fused16k && !fused4k && (rc = ensure(ctx, ctx->X, x_epoch_bytes * Ec)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->rows, sizeof(RowRec) * (size_t)Ec * P * D)) != GACQ_OK) return rc;

  const int chunksN = (N + kBlock * 8 - 1) / (kBlock * 8);
  const float2* lds_spectra = (N == 16384 && ctx->opt[GACQ_OPT_LDS_VARIANT] == 16) ? sig->spectra_lds16 : sig->spectra_lds;
  for (int e0 = 0; e0 < nepoch; e0 += Ec) {
    const int ne = std::min(Ec, nepoch - e0);
    const float2* xe = d_x + (size_t)e0 * nsamp;
    float2* X = (float2*)ctx->X.p;
    RowRec* rows = (RowRec*)ctx->rows.p;
    const long rows_x = (long)ne * F * D * B;
    if (fused16k) {
      stage_begin(ctx, 6);
      rc = lds_fused_search(ctx, xe, nsamp, ne, n, N, lds_spectra, (const int*)ctx->items.p, (const int*)ctx->fset.p,
                            (const double*)ctx->freq.p, (const float2*)ctx->tab.p, P, D, B, rows, tscale);
      stage_end(ctx);
      if (rc != GACQ_OK) return rc;
    } else if (fused4k) {
      stage_begin(ctx, 6);
      rc = lds_fused4k_search(ctx, xe, nsamp, ne, sig->spectra_lds, (const int*)ctx->items.p, (const double*)ctx->freq.p, (const float2*)ctx->tab.p, P, D, rows,
                              tscale);
      stage_end(ctx);
      if (rc != GACQ_OK) return rc;
    } else if (use_lds) {
      stage_begin(ctx, 0);
      rc = lds_forward(ctx, xe, nsamp, ne, n, N, (const double*)ctx->freq.p, F * D, B, (const float2*)ctx->tab.p, X);
      stage_end(ctx);
      if (rc != GACQ_OK) return rc;
      stage_begin(ctx, 6);
      rc = lds_correlate(ctx, X, lds_spectra, (const int*)ctx->items.p, (const int*)ctx->fset.p, ne, P, F, D, B, N, rows, tscale, d_qrow);
      stage_end(ctx);
      if (rc != GACQ_OK) return rc;
    } else {
      if (use_pfa) {
        stage_begin(ctx, 0);
        rc = pfa_forward(ctx, xe, nsamp, rows_x, n, N, (const double*)ctx->freq.p, F * D, B, (const float2*)ctx->tab.p, X, true);
        stage_end(ctx);
        if (rc != GACQ_OK) return rc;
      } else if (use_split) {
        stage_begin(ctx, 0);
        rc = split_forward(ctx, xe, nsamp, rows_x, n, N, (const double*)ctx->freq.p, F * D, B, (const float2*)ctx->tab.p, X, true,
                         !use_split_lds);
        if (rc == GACQ_OK && use_split_lds) rc = lds_inner_forward(ctx, X, rows_x * R, true);
        stage_end(ctx);
        if (rc != GACQ_OK) return rc;
      } else {
        stage_begin(ctx, 0);
        hipLaunchKernelGGL(mix_nco_kernel<false>, dim3((unsigned)(rows_x * chunksN)), dim3(kBlock), 0, st, xe, nsamp, X,
                           (const double*)ctx->freq.p, (const float2*)ctx->tab.p, n, N, F * D, B, chunksN);
        stage_end(ctx);
        GACQ_HIP(ctx, hipGetLastError());
        stage_begin(ctx, 1);
        rc = fft_exec(ctx, N, rows_x, false, X);
        stage_end(ctx);
        if (rc != GACQ_OK) return rc;
      }
      // correlation workspace: chunks of whole (e,p,d) groups, B rows each
      const long groups = (long)ne * P * D;
      const int zpitch = use_pfa ? pfa_row_pitch(N) : 0;       // prime-factor engine: Z' rows are whole reader workgroups of 128-byte lines
      const size_t group_bytes = sizeof(float2) * (size_t)B * (zpitch ? (size_t)zpitch * R : (size_t)N);
      // One pass of Z' is at most 4 GiB, or eight items' worth where that is more (the writers share their forward-spectrum tiles
      // over up to eight items of a pass: a B = 80 search of engine 3 holds 1.4 items in 4 GiB and runs 22 % slower for it).  Beyond
      // that a bigger pass buys nothing (config 5's E1B, 5.24 GB of Z': 2.09-2.12 ms from 3 GiB to 32 GiB per pass, every point a
      // fresh process, profiles/r06_e1b_pass_size_sweep.log) and has cost 25 % on one box (round 5's driver run: 2.42 against 1.93 ms
      // in one 5.24 GB pass), so the workspace limit is the ceiling of a pass, not its size.
      const size_t pass_cap = std::min(ws_budget(ctx), std::max<size_t>((size_t)4 << 30, 8 * group_bytes * (size_t)D));
      long gc = (long)std::max<size_t>(1, std::min<size_t>((size_t)groups, pass_cap / group_bytes));
      gc = (groups + (groups + gc - 1) / gc - 1) / ((groups + gc - 1) / gc);       // equal passes instead of full ones and a sliver
      if ((rc = ensure(ctx, ctx->Y, group_bytes * gc)) != GACQ_OK) return rc;
      float2* Y = (float2*)ctx->Y.p;
      for (long g0 = 0; g0 < groups; g0 += gc) {
        const long ng = std::min(gc, groups - g0);
        if (use_split_lds) {
          stage_begin(ctx, 6);
          rc = lds_inner_correlate(ctx, X, sig->spectra_split, (const int*)ctx->items.p, (const int*)ctx->fset.p, g0, ng, P, F, D, B, R,
                                   N, Y);                                       // K2 + inner inverse FFT + twiddle, fused
          stage_end(ctx);
          if (rc != GACQ_OK) return rc;
          stage_begin(ctx, 4);
          rc = split_inverse_reduce(ctx, Y, rows, g0, ng, B, N, d_qrow, tscale, false);   // outer inverse DFT + |.| + reduce
          stage_end(ctx);
          if (rc != GACQ_OK) return rc;
          continue;
        }
        if (use_pfa) {
          stage_begin(ctx, 6);
          rc = pfa_inner_correlate(ctx, X, sig->spectra_pfa, (const int*)ctx->items.p, (const int*)ctx->fset.p, g0, ng, P, F, D, B, N, Y);   // K2 + inner inverse transforms, in place in LDS
          stage_end(ctx);
          if (rc != GACQ_OK) return rc;
          stage_begin(ctx, 4);
          rc = pfa_inverse_reduce(ctx, Y, rows, g0, ng, B, N, d_qrow, tscale, ds.metric_mode != 0);                 // inverse DFT-31 + |.| + reduce
          stage_end(ctx);
          if (rc != GACQ_OK) return rc;
          continue;
        }
        stage_begin(ctx, 2);
        hipLaunchKernelGGL(conj_mul_kernel, dim3((unsigned)(ng * B * chunksN)), dim3(kBlock), 0, st, X,
                           use_split ? sig->spectra_r31 : sig->spectra, Y,
                           (const int*)ctx->items.p, (const int*)ctx->fset.p, g0, P, F, D, B, N, chunksN);
        stage_end(ctx);
        GACQ_HIP(ctx, hipGetLastError());
        if (use_split) {
          stage_begin(ctx, 3);
          rc = split_inverse_reduce(ctx, Y, rows, g0, ng, B, N, d_qrow, tscale);      // inner inverse FFTs + outer DFT-31 + |.| + reduce
          stage_end(ctx);
          if (rc != GACQ_OK) return rc;
        } else {
          stage_begin(ctx, 3);
          rc = fft_exec(ctx, N, ng * B, true, Y);
          stage_end(ctx);
          if (rc != GACQ_OK) return rc;
          stage_begin(ctx, 4);
          hipLaunchKernelGGL(mag_peak_kernel, dim3((unsigned)ng), dim3(kBlock), 0, st, Y, rows, g0, B, N, 1.0f / (float)N, d_qrow, tscale);
          stage_end(ctx);
          GACQ_HIP(ctx, hipGetLastError());
        }
      }
    }
    const long nep = (long)ne * P;
    stage_begin(ctx, 5);
    if (tie)
      hipLaunchKernelGGL(best_doppler_kernel<true>, dim3((unsigned)((nep + 3) / 4)), dim3(256), 0, st, rows, d_out + (size_t)e0 * P, nep, D, N,
                         ds.metric_mode, tl, guesses, tscale);
    else
      hipLaunchKernelGGL(best_doppler_kernel<false>, dim3((unsigned)((nep + 3) / 4)), dim3(256), 0, st, rows, d_out + (size_t)e0 * P, nep, D, N,
                         ds.metric_mode, tl, guesses, tscale);
    stage_end(ctx);
    GACQ_HIP(ctx, hipGetLastError());
    if (tie && (rc = tie_resolve(sig, tl, guesses, xs.offset((size_t)e0 * nsamp), nsamp, P, D, B, d_out + (size_t)e0 * P)) != GACQ_OK) return rc;
  }
  return GACQ_OK;
}

int check_search_args(gacq_sig* sig, const void* x, size_t nsamp, int nepoch, const int* items, int nitems,
                      const double* dopplers, int nd, int blocks, const void* out) {
  gacq_ctx* ctx = sig ? sig->ctx : nullptr;
  if (!sig || !x || !items || !out || nitems <= 0 || nepoch <= 0 || nd < 0 || blocks < 0 || (nd > 0 && !dopplers))
    return set_error(ctx, GACQ_ERR_BAD_ARG, "search: bad argument");
  for (int p = 0; p < nitems; p++)
    if (items[p] < 0 || items[p] >= sig->nprn) return set_error(ctx, GACQ_ERR_BAD_PRN, "search: item index %d out of range [0,%d)", items[p], sig->nprn);
  for (int k = 0; k < nd; k++)
    if (!std::isfinite(dopplers[k])) return set_error(ctx, GACQ_ERR_BAD_ARG, "search: Doppler value %d is not finite", k);
  const size_t need = (size_t)(blocks + (sig->desc.pad ? 1 : 0)) * sig->desc.n;
  if (blocks > 0 && nd > 0 && nsamp < need)              // an empty Doppler grid never touches x (acquire-gps-l1.py:25-26,40)
    return set_error(ctx, GACQ_ERR_SHORT_INPUT, "search: %zu samples given, %zu needed for %d block(s) of n=%d%s", nsamp, need, blocks,
                     sig->desc.n, sig->desc.pad ? " (padded: windows span 2n)" : "");
  return GACQ_OK;
}

__global__ void zero_peaks_kernel(gacq_peak* out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out[i].metric = 0.0; out[i].idx = -1; out[i].d_index = -1; }
}

// TIE: shard winners whose metrics come within eps of the best one cannot be ordered in fp32 (each is an fp32 value unless its own
// shard re-evaluated it): their rows are listed for re-evaluation in complex128, with GLOBAL Doppler indices, and tie_resolve_kernel
// writes the record -- the same machinery as best_doppler_kernel<true>, one level up.
template <bool TIE>
__global__ void merge_peaks_kernel(const gacq_peak* __restrict__ peaks, gacq_peak* __restrict__ out, long n, int nshard,
                                   const int* __restrict__ d0, TieLists tl, gacq_peak* __restrict__ guesses, float tie_scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  gacq_peak best;
  best.metric = 0.0;
  best.idx = -1;
  best.d_index = -1;
  double second = 0.0;
  for (int s = 0; s < nshard; s++) {
    const gacq_peak k = peaks[(long)s * n + i];
    if (k.d_index < 0) continue;
    if (k.metric > best.metric) { second = best.metric; best.metric = k.metric; best.idx = k.idx; best.d_index = k.d_index + d0[s]; }
    else if (k.metric > second) second = k.metric;
  }
  const double thr = best.metric * (double)tie_scale;
  if (!TIE || best.d_index < 0 || second < thr) { out[i] = best; return; }
  int cnt = 0;
  for (int s = 0; s < nshard; s++) {
    const gacq_peak k = peaks[(long)s * n + i];
    cnt += (k.d_index >= 0 && k.metric >= thr) ? 1 : 0;
  }
  const int slot0 = (int)atomicAdd(&tl.c->nrows, (unsigned)cnt);
  if (slot0 < 0 || slot0 + cnt > tl.cap) {
    for (int sl = slot0; sl < slot0 + cnt && sl < tl.cap && sl >= 0; sl++) { TieRow v; v.ep = -1; v.d = 0; tl.rows[sl] = v; }
    *(volatile unsigned*)tl.host_full = 1u;
    __threadfence_system();
    out[i] = best;
    atomicAdd(&tl.c->overflow, 1ull);
    return;
  }
  const int epslot = (int)atomicAdd(&tl.c->neps, 1u);
  TieEp te;
  te.ep = (int)i; te.slot0 = slot0; te.cnt = cnt; te.pad = 0;
  tl.eps[epslot] = te;
  guesses[epslot] = best;
  int off = 0;
  for (int s = 0; s < nshard; s++) {                   // shards are in global Doppler order: ascending d
    const gacq_peak k = peaks[(long)s * n + i];
    if (k.d_index >= 0 && k.metric >= thr) { TieRow v; v.ep = (int)i; v.d = k.d_index + d0[s]; tl.rows[slot0 + off++] = v; }
  }
}

int upload_d0(gacq_ctx* ctx, const int* shard_d0, int nshard) {
  const void* d0_before = ctx->d0.p;
  int rc = ensure(ctx, ctx->d0, sizeof(int) * 4096);
  if (rc != GACQ_OK) return rc;
  if (d0_before != ctx->d0.p) ctx->up_d0.clear();
  const std::vector<int> d0_v(shard_d0, shard_d0 + nshard);
  if (d0_v != ctx->up_d0) {
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->d0.p, shard_d0, sizeof(int) * nshard, hipMemcpyHostToDevice, ctx->stream));
    GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->up_d0 = d0_v;
  }
  return GACQ_OK;
}

}  // namespace

extern "C" {

static int merge_tiesafe(gacq_sig* sig, XSrc xs, size_t nsamp, int nepoch, const int* items, int nitems, const double* dopplers, int nd,
                         const double* item_bias_hz, int blocks, const void* d_peaks, int nshard, const int* shard_d0, void* d_out);

// complex128 samples -> the complex64 form the fp32 engines search (round to nearest, like x.astype(np.complex64))
__global__ void narrow_x_kernel(const double2* __restrict__ in, float2* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double2 v = in[i]; out[i] = make_float2((float)v.x, (float)v.y); }
}

static int search_batch_dev_once(gacq_sig* sig, XSrc xs, size_t nsamp, int nepoch, const int* items, int nitems,
                                 const double* dopplers, int nd, const double* item_bias_hz, int blocks, void* d_out);

// A search whose workspace could not be allocated (another context, a torch allocator or a co-resident rank took the memory the budget
// was computed from) is run again in passes of half the size, down to 64 MiB: slower, never an error a caller has to handle by hand.
static int search_batch_dev(gacq_sig* sig, XSrc xs, size_t nsamp, int nepoch, const int* items, int nitems,
                            const double* dopplers, int nd, const double* item_bias_hz, int blocks, void* d_out) {
  for (;;) {
    if (sig && sig->ctx) sig->ctx->alloc_failed = false;
    const int rc = search_batch_dev_once(sig, xs, nsamp, nepoch, items, nitems, dopplers, nd, item_bias_hz, blocks, d_out);
    if (rc == GACQ_OK || !sig || !sig->ctx || !sig->ctx->alloc_failed) return rc;
    gacq_ctx* ctx = sig->ctx;
    GACQ_DEVICE(ctx);
    const size_t budget = ws_budget(ctx);
    if (budget <= ((size_t)64 << 20)) return rc;
    (void)hipDeviceSynchronize();
    for (DevBuf* b : {&ctx->X, &ctx->Y}) {
      if (b->p) (void)hipFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
    ctx->ws_soft = budget / 2;
  }
}

static int search_batch_dev_once(gacq_sig* sig, XSrc xs, size_t nsamp, int nepoch, const int* items, int nitems,
                                 const double* dopplers, int nd, const double* item_bias_hz, int blocks, void* d_out) {
  int rc = check_search_args(sig, xs.p, nsamp, nepoch, items, nitems, dopplers, nd, blocks, d_out);
  if (rc != GACQ_OK) return rc;
  gacq_ctx* ctx = sig->ctx;
  GACQ_DEVICE(ctx);
  const void* d_x = xs.p;
  if (xs.wide && ctx->engine != 5 && nd > 0 && blocks > 0) {
    const size_t count = (size_t)nepoch * nsamp;
    if ((rc = ensure(ctx, ctx->x32, sizeof(float2) * count)) != GACQ_OK) return rc;
    hipLaunchKernelGGL(narrow_x_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, (const double2*)xs.p, (float2*)ctx->x32.p, count);
    GACQ_HIP(ctx, hipGetLastError());
    d_x = ctx->x32.p;
  }
  if (nd == 0 || blocks == 0) {
    // empty Doppler grid -> the reference returns its initial (0,0,0) (acquire-gps-l1.py:25,40);
    // zero blocks -> q == 0 everywhere, nothing beats metric 0 either (raw) / NaN never wins (normalised)
    const long n = (long)nepoch * nitems;
    hipLaunchKernelGGL(zero_peaks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (gacq_peak*)d_out, n);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  // One epoch's forward spectra [F][D][B][N] larger than the workspace limit (long integrations: galileo-e1b --time 200 is
  // 9 GB): the Doppler grid is cut into slices that fit, each slice is an ordinary search, and the per-slice peaks are merged
  // in grid order with strict '>' by the shard-merge kernel -- the same winner as one scan over the whole grid.
  int F = 1;
  if (item_bias_hz) {
    std::vector<double> seen;
    for (int p = 0; p < nitems; p++) if (std::find(seen.begin(), seen.end(), item_bias_hz[p]) == seen.end()) seen.push_back(item_bias_hz[p]);
    F = (int)seen.size();
  }
  const size_t bin_bytes = (ctx->engine == 5 ? sizeof(double2) : sizeof(float2)) * (size_t)F * blocks * sig->N;
  const bool no_x = lds_path(ctx, sig->N, nepoch, nitems, F, nd, blocks, false).no_forward_buffer();
  if (!no_x && nd > 1 && bin_bytes * nd > ws_budget(ctx)) {
    const int Dc = (int)std::max<size_t>(1, ws_budget(ctx) / bin_bytes);
    const int nch = (nd + Dc - 1) / Dc;
    const long n = (long)nepoch * nitems;
    if ((rc = ensure(ctx, ctx->chunk_peaks, sizeof(gacq_peak) * (size_t)nch * n)) != GACQ_OK) return rc;
    std::vector<int> d0(nch);
    for (int c = 0; c < nch; c++) {
      d0[c] = c * Dc;
      rc = launch_search(sig, xs, (const float2*)d_x, nsamp, nepoch, items, nitems, dopplers + d0[c], std::min(Dc, nd - d0[c]), item_bias_hz,
                         blocks, (gacq_peak*)ctx->chunk_peaks.p + (size_t)c * n, nullptr);
      if (rc != GACQ_OK) return rc;
    }
    // slice winners within eps of each other are re-evaluated in complex128 like near-tied bins of one scan (tie-safe locations)
    return merge_tiesafe(sig, xs, nsamp, nepoch, items, nitems, dopplers, nd, item_bias_hz, blocks, ctx->chunk_peaks.p, nch, d0.data(), d_out);
  }
  return launch_search(sig, xs, (const float2*)d_x, nsamp, nepoch, items, nitems, dopplers, nd, item_bias_hz, blocks,
                       (gacq_peak*)d_out, nullptr);
}

int gacq_search_batch_dev(gacq_sig* sig, const void* d_x, size_t nsamp, int nepoch, const int* items, int nitems,
                          const double* dopplers, int nd, const double* item_bias_hz, int blocks, void* d_out) {
  return search_batch_dev(sig, XSrc{d_x, 0}, nsamp, nepoch, items, nitems, dopplers, nd, item_bias_hz, blocks, d_out);
}

int gacq_search_batch_dev64(gacq_sig* sig, const void* d_x_c128, size_t nsamp, int nepoch, const int* items, int nitems,
                            const double* dopplers, int nd, const double* item_bias_hz, int blocks, void* d_out) {
  return search_batch_dev(sig, XSrc{d_x_c128, 1}, nsamp, nepoch, items, nitems, dopplers, nd, item_bias_hz, blocks, d_out);
}

int gacq_merge_peaks_dev(gacq_ctx* ctx, const void* d_peaks, int nshard, const int* shard_d0, long n, void* d_out) {
  if (!ctx || !d_peaks || !d_out || !shard_d0 || nshard <= 0 || nshard > 4096 || n <= 0)
    return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_merge_peaks_dev: bad argument");
  GACQ_DEVICE(ctx);
  int rc = upload_d0(ctx, shard_d0, nshard);
  if (rc != GACQ_OK) return rc;
  hipLaunchKernelGGL(merge_peaks_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const gacq_peak*)d_peaks,
                     (gacq_peak*)d_out, n, nshard, (const int*)ctx->d0.p, TieLists{}, (gacq_peak*)nullptr, 1.0f);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int gacq_merge_peaks_tiesafe_dev(gacq_sig* sig, const void* d_x, size_t nsamp, int nepoch, const int* items, int nitems, const double* dopplers,
                                 int nd, const double* item_bias_hz, int blocks, const void* d_peaks, int nshard, const int* shard_d0,
                                 void* d_out) {
  return merge_tiesafe(sig, XSrc{d_x, 0}, nsamp, nepoch, items, nitems, dopplers, nd, item_bias_hz, blocks, d_peaks, nshard, shard_d0, d_out);
}

int gacq_merge_peaks_tiesafe_dev64(gacq_sig* sig, const void* d_x_c128, size_t nsamp, int nepoch, const int* items, int nitems,
                                   const double* dopplers, int nd, const double* item_bias_hz, int blocks, const void* d_peaks, int nshard,
                                   const int* shard_d0, void* d_out) {
  return merge_tiesafe(sig, XSrc{d_x_c128, 1}, nsamp, nepoch, items, nitems, dopplers, nd, item_bias_hz, blocks, d_peaks, nshard, shard_d0, d_out);
}

static int merge_tiesafe(gacq_sig* sig, XSrc xs, size_t nsamp, int nepoch, const int* items, int nitems, const double* dopplers, int nd,
                         const double* item_bias_hz, int blocks, const void* d_peaks, int nshard, const int* shard_d0, void* d_out) {
  const void* d_x = xs.p;
  int rc = check_search_args(sig, d_x, nsamp, nepoch, items, nitems, dopplers, nd, blocks, d_out);
  if (rc != GACQ_OK) return rc;
  gacq_ctx* ctx = sig->ctx;
  if (!d_peaks || !shard_d0 || nshard <= 0 || nshard > 4096) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_merge_peaks_tiesafe_dev: bad argument");
  GACQ_DEVICE(ctx);
  const long n = (long)nepoch * nitems;
  const bool tie = ctx->opt[GACQ_OPT_TIE_SAFE] != 0 && ctx->engine != 5 && nd > 0 && blocks > 0 && tie_supported(sig->N);
  if (!tie) return gacq_merge_peaks_dev(ctx, d_peaks, nshard, shard_d0, n, d_out);
  if ((rc = upload_d0(ctx, shard_d0, nshard)) != GACQ_OK) return rc;
  int F = 1;
  if ((rc = upload_grid(sig, nepoch, items, nitems, dopplers, nd, item_bias_hz, &F)) != GACQ_OK) return rc;      // the FULL grid: global bins
  TieLists tl{};
  gacq_peak* guesses = nullptr;
  if ((rc = tie_prepare(sig)) != GACQ_OK) return rc;
  if ((rc = tie_lists(ctx, n, sig->N, blocks, &tl, &guesses)) != GACQ_OK) return rc;
  hipLaunchKernelGGL(merge_peaks_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const gacq_peak*)d_peaks,
                     (gacq_peak*)d_out, n, nshard, (const int*)ctx->d0.p, tl, guesses, tie_scale_of(ctx));
  GACQ_HIP(ctx, hipGetLastError());
  return tie_resolve(sig, tl, guesses, xs, nsamp, nitems, nd, blocks, (gacq_peak*)d_out);
}

int gacq_finalize(const gacq_sigdesc* desc, const gacq_peak* peaks, int nshard, const int* shard_d0, int nitems,
                  const double* dopplers, int nd, gacq_result* out) {
  if (!desc || !peaks || !out || nshard <= 0 || nitems <= 0 || (nd > 0 && !dopplers) || desc->n <= 0)
    return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_finalize: bad argument");
  const gacq_sigdesc& d = *desc;
  for (int p = 0; p < nitems; p++) {
    double best = 0.0;
    int bidx = -1, bd = -1;
    for (int s = 0; s < nshard; s++) {               // shards in global Doppler order, strict '>'
      const gacq_peak& k = peaks[(size_t)s * nitems + p];
      if (k.d_index >= 0 && k.metric > best) { best = k.metric; bidx = k.idx; bd = k.d_index + (shard_d0 ? shard_d0[s] : 0); }
    }
    gacq_result r;
    r.idx = bidx;
    r.d_index = bd;
    if (bd < 0 || bd >= nd) {
      r.metric = 0.0; r.code_chips = 0.0; r.doppler_hz = 0.0; r.idx = -1; r.d_index = -1;
    } else {
      r.metric = best;
      r.code_chips = (double)d.code_length * ((double)bidx / (double)d.n);      // acquire-gps-l1.py:38
      if (d.fold_code) r.code_chips = std::fmod(r.code_chips, (double)d.code_length);   // acquire-beidou-b1i.py:39
      r.doppler_hz = dopplers[bd];
    }
    out[p] = r;
  }
  return GACQ_OK;
}

// GACQ_WARN_TIE_LIST_FULL for the host-buffer searches: the listing kernels set a pinned host word BEFORE they write the record of a pair
// that found no room, so a caller who has seen every record has seen the word.  It is cleared only with the stream drained (the
// re-evaluation kernels that trail the records are still running when the watched records are complete).
static int tie_list_full_warning(gacq_ctx* ctx) {
  if (!ctx->pin_tie.p || *(volatile unsigned*)ctx->pin_tie.p == 0u) return GACQ_OK;
  GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *(volatile unsigned*)ctx->pin_tie.p = 0u;
  return GACQ_WARN_TIE_LIST_FULL;
}

// One search from host memory, complex64 or (wide) complex128 samples: the body of gacq_search / gacq_search64
static int search_host(gacq_sig* sig, const void* x_iq, bool wide, size_t nsamp, const int* items, int nitems, const double* dopplers,
                       int nd, const double* item_bias_hz, int blocks, gacq_result* out) {
  int rc = check_search_args(sig, x_iq, nsamp, 1, items, nitems, dopplers, nd, blocks, out);
  if (rc != GACQ_OK) return rc;
  gacq_ctx* ctx = sig->ctx;
  if (nd == 0 || blocks == 0) {
    // empty Doppler grid / no block: the reference returns its untouched initial (0,0,0) (acquire-gps-l1.py:25,40) and never
    // looks at x, which may then be shorter than one window (Python callers pass x[:0]) -- nothing is staged or launched
    for (int p = 0; p < nitems; p++) { out[p].metric = 0.0; out[p].code_chips = 0.0; out[p].doppler_hz = 0.0; out[p].idx = -1; out[p].d_index = -1; }
    return GACQ_OK;
  }
  GACQ_DEVICE(ctx);
  const size_t need = (size_t)(blocks + (sig->desc.pad ? 1 : 0)) * sig->desc.n;      // <= nsamp (check_search_args)
  const size_t take = need;
  if ((rc = ensure_pinned(ctx, ctx->pin_peaks, sizeof(gacq_peak) * std::max(1, nitems))) != GACQ_OK) return rc;
  const size_t xbytes = (wide ? sizeof(double2) : sizeof(float2)) * need;
  const void* d_x = nullptr;
  // Small inputs (a 1 ms GPS L1 block is 32 KB): written by the host straight into fine-grained device memory through the PCIe BAR
  // (a 32 KB pinned H2D costs ~7 us of latency in front of the kernels).  This call is synchronous, so the buffer is never
  // rewritten while a kernel still reads it.
  if (bar_write(ctx, ctx->bar_x, x_iq, xbytes)) d_x = ctx->bar_x.p;
  if (!d_x) {
    // Staged path: small inputs go through a pinned staging buffer (one host memcpy, then a true async DMA); a pageable
    // hipMemcpyAsync stages internally and costs ~10 us more per call.  Large inputs are copied directly.
    if ((rc = ensure(ctx, ctx->xstage, xbytes)) != GACQ_OK) return rc;
    const void* src = x_iq;
    if (xbytes <= kPinnedStageMax) {
      if ((rc = ensure_pinned(ctx, ctx->pin_x, xbytes)) != GACQ_OK) return rc;
      std::memcpy(ctx->pin_x.p, x_iq, xbytes);
      src = ctx->pin_x.p;
    }
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->xstage.p, src, xbytes, hipMemcpyHostToDevice, ctx->stream));
    d_x = ctx->xstage.p;
  }
  // the Doppler scan writes its 16-byte records straight into device-visible pinned host memory: no D2H copy
  // Completion (latency path, small inputs only): the host watches the pinned result records themselves instead of asking the
  // runtime (an empty hipStreamSynchronize costs ~3 us here).  Every record is pre-set to a sentinel in both of its 8-byte halves;
  // the last kernel of the search overwrites all of them, so "no sentinel left" means the search is complete.  Bounded: after
  // 200 us (or on any doubt) the call falls back to hipStreamSynchronize, which is also what reports a failed launch.
  gacq_peak* pk = (gacq_peak*)ctx->pin_peaks.p;
  const bool watch = d_x == ctx->bar_x.p && ctx->opt[GACQ_OPT_WATCH_RESULTS] != 0;
  constexpr unsigned long long kSentinelBits = 0x7ff8dead0badbeefULL;      // a NaN payload no search produces
  if (watch)
    for (int p = 0; p < nitems; p++) { std::memcpy(&pk[p].metric, &kSentinelBits, 8); pk[p].idx = -2; pk[p].d_index = -2; }
  rc = search_batch_dev(sig, XSrc{d_x, wide ? 1 : 0}, take, 1, items, nitems, dopplers, nd, item_bias_hz, blocks, ctx->pin_peaks.p);
  if (rc != GACQ_OK) return rc;
  bool complete = false;
  if (watch) {
    const volatile unsigned long long* words = reinterpret_cast<const volatile unsigned long long*>(pk);
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
    int first_pending = 0;
    for (unsigned spin = 0;; spin++) {
      // both 8-byte halves of a record through volatile reads (little endian: d_index is the upper half of the second word)
      while (first_pending < nitems && words[2 * first_pending] != kSentinelBits && (int)(words[2 * first_pending + 1] >> 32) != -2) first_pending++;
      if (first_pending == nitems) { complete = true; break; }
      cpu_relax();
      if ((spin & 63) == 63 && std::chrono::steady_clock::now() > t_end) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!complete) GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  rc = gacq_finalize(&sig->desc, (const gacq_peak*)ctx->pin_peaks.p, 1, nullptr, nitems, dopplers, nd, out);
  return rc != GACQ_OK ? rc : tie_list_full_warning(ctx);
}

int gacq_search(gacq_sig* sig, const float* x_iq, size_t nsamp, const int* items, int nitems, const double* dopplers,
                int nd, const double* item_bias_hz, int blocks, gacq_result* out) {
  return search_host(sig, x_iq, false, nsamp, items, nitems, dopplers, nd, item_bias_hz, blocks, out);
}

// complex128 samples as the reference's search() receives them (np.interp output, acquire-gps-l1.py:94-96): the same call path as
// gacq_search -- BAR upload of small inputs (a 1 ms GPS L1 block is 64 KB here), pinned staging, completion by watching the records
int gacq_search64(gacq_sig* sig, const double* x_iq, size_t nsamp, const int* items, int nitems, const double* dopplers,
                  int nd, const double* item_bias_hz, int blocks, gacq_result* out) {
  return search_host(sig, x_iq, true, nsamp, items, nitems, dopplers, nd, item_bias_hz, blocks, out);
}

// The whole main program of an acquire script in one call, from host memory, without a device-memory framework on the caller's side:
// the int8 block goes up, the front-end (gacq_frontend_dev) and the search (gacq_search_batch_dev) run on it where it lies, the results
// come back.  Synchronous.
int gacq_acquire_int8(gacq_sig* sig, const int8_t* iq_int8, size_t nsamp_in, double fs_in, double carrier_offset_hz, const double* taps,
                      int ntaps, size_t nsamp_out, const int* items, int nitems, const double* dopplers, int nd, const double* item_bias_hz,
                      int blocks, gacq_result* out) {
  if (!sig || !iq_int8 || !taps || nsamp_in == 0 || nsamp_out == 0 || !(fs_in > 0.0) || !std::isfinite(carrier_offset_hz))
    return set_error(sig ? sig->ctx : nullptr, GACQ_ERR_BAD_ARG, "gacq_acquire_int8: bad argument");
  gacq_ctx* ctx = sig->ctx;
  // argument checks of the search against the front-end's output length (short input -> GACQ_ERR_SHORT_INPUT before anything is launched)
  static const float dummy = 0.f;
  int rc = check_search_args(sig, &dummy, nsamp_out, 1, items, nitems, dopplers, nd, blocks, out);
  if (rc != GACQ_OK) return rc;
  GACQ_DEVICE(ctx);
  if ((rc = ensure(ctx, ctx->acq_in, 2 * nsamp_in)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->acq_x, sizeof(float2) * nsamp_out)) != GACQ_OK) return rc;
  if ((rc = ensure_pinned(ctx, ctx->pin_peaks, sizeof(gacq_peak) * std::max(1, nitems))) != GACQ_OK) return rc;
  GACQ_HIP(ctx, hipMemcpyAsync(ctx->acq_in.p, iq_int8, 2 * nsamp_in, hipMemcpyHostToDevice, ctx->stream));
  rc = gacq_frontend_dev(ctx, ctx->acq_in.p, nsamp_in, fs_in, carrier_offset_hz, taps, ntaps, sig->desc.fs, nsamp_out, ctx->acq_x.p);
  if (rc != GACQ_OK) return rc;
  if (nd == 0 || blocks == 0 || nitems == 0) {
    GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < nitems; p++) { out[p].metric = 0.0; out[p].code_chips = 0.0; out[p].doppler_hz = 0.0; out[p].idx = -1; out[p].d_index = -1; }
    return GACQ_OK;
  }
  rc = gacq_search_batch_dev(sig, ctx->acq_x.p, nsamp_out, 1, items, nitems, dopplers, nd, item_bias_hz, blocks, ctx->pin_peaks.p);
  if (rc != GACQ_OK) return rc;
  GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  rc = gacq_finalize(&sig->desc, (const gacq_peak*)ctx->pin_peaks.p, 1, nullptr, nitems, dopplers, nd, out);
  return rc != GACQ_OK ? rc : tie_list_full_warning(ctx);
}

int gacq_debug_fft_plans(gacq_ctx* ctx) {
  if (!ctx) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_debug_fft_plans: bad argument");
  return (int)ctx->plans.size();
}

int gacq_debug_nco_indices(gacq_sig* sig, int kernel, double doppler, double bias_hz, int* idx_out) {
  if (!sig || !idx_out || !std::isfinite(doppler) || !std::isfinite(bias_hz))
    return set_error(sig ? sig->ctx : nullptr, GACQ_ERR_BAD_ARG, "gacq_debug_nco_indices: bad argument");
  gacq_ctx* ctx = sig->ctx;
  GACQ_DEVICE(ctx);
  const int n = sig->desc.n, N = sig->N;
  // the frequency goes through the same host expression as a search (fp64, the reference's operation order)
  const Grid g = make_grid(sig->desc, 1, &doppler, 1, bias_hz != 0.0 ? &bias_hz : nullptr);
  if (!nco_range_ok(std::fabs(g.freq[0]), N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "gacq_debug_nco_indices: NCO phase index out of range");
  int rc;
  if ((rc = ensure(ctx, ctx->freq, sizeof(double))) != GACQ_OK) return rc;
  ctx->up_freq.clear();                                   // the cached search grid is overwritten
  if ((rc = ensure(ctx, ctx->Y, sizeof(int) * (size_t)N)) != GACQ_OK) return rc;
  GACQ_HIP(ctx, hipMemcpyAsync(ctx->freq.p, g.freq.data(), sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  GACQ_HIP(ctx, hipMemsetAsync(ctx->Y.p, 0xff, sizeof(int) * (size_t)N, ctx->stream));
  int* d_idx = (int*)ctx->Y.p;
  switch (kernel) {
    case 1: {
      const int chunksN = (N + kBlock * 8 - 1) / (kBlock * 8);
      hipLaunchKernelGGL(mix_nco_kernel<true>, dim3((unsigned)chunksN), dim3(kBlock), 0, ctx->stream, (const float2*)nullptr, (size_t)0,
                         (float2*)d_idx, (const double*)ctx->freq.p, (const float2*)ctx->tab.p, n, N, 1, 1, chunksN);
      GACQ_HIP(ctx, hipGetLastError());
      break;
    }
    case 2: rc = lds_debug_nco(ctx, N, n, (const double*)ctx->freq.p, false, d_idx); break;
    case 3: rc = split_debug_nco(ctx, N, n, (const double*)ctx->freq.p, d_idx); break;
    case 4: rc = lds_debug_nco(ctx, N, n, (const double*)ctx->freq.p, true, d_idx); break;
    case 5: rc = pfa_debug_nco(ctx, N, n, (const double*)ctx->freq.p, d_idx); break;
    default: rc = set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_debug_nco_indices: kernel must be 1..5");
  }
  if (rc != GACQ_OK) return rc;
  GACQ_HIP(ctx, hipMemcpyAsync(idx_out, d_idx, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
  GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GACQ_OK;
}

int gacq_debug_row(gacq_sig* sig, const float* x_iq, size_t nsamp, int item, double doppler, double bias_hz, int blocks,
                   float* q_out) {
  gacq_peak dummy_out;
  int rc = check_search_args(sig, x_iq, nsamp, 1, &item, 1, &doppler, 1, blocks, &dummy_out);
  if (rc != GACQ_OK) return rc;
  if (!q_out || blocks <= 0) return set_error(sig->ctx, GACQ_ERR_BAD_ARG, "gacq_debug_row: bad argument");
  gacq_ctx* ctx = sig->ctx;
  GACQ_DEVICE(ctx);
  const size_t need = (size_t)(blocks + (sig->desc.pad ? 1 : 0)) * sig->desc.n;
  if ((rc = ensure(ctx, ctx->xstage, sizeof(float2) * need)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->out_peaks, sizeof(gacq_peak))) != GACQ_OK) return rc;
  float* d_q = nullptr;
  GACQ_HIP(ctx, hipMalloc((void**)&d_q, sizeof(float) * sig->N));
  if (hipMemcpyAsync(ctx->xstage.p, x_iq, sizeof(float2) * need, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
    (void)hipFree(d_q);
    return set_error(ctx, GACQ_ERR_HIP, "gacq_debug_row: H2D failed");
  }
  const int saved = ctx->engine;
  // row dump: the engine that was asked for -- rocFFT pipeline (also what auto means here), LDS kernels (two-kernel path), split
  // engines, fp64 pipeline
  ctx->engine = (saved >= 2 && saved <= 5) ? saved : 1;
  rc = launch_search(sig, XSrc{ctx->xstage.p, 0}, (const float2*)ctx->xstage.p, need, 1, &item, 1, &doppler, 1, bias_hz != 0.0 ? &bias_hz : nullptr, blocks,
                     (gacq_peak*)ctx->out_peaks.p, d_q);
  ctx->engine = saved;
  if (rc == GACQ_OK && hipMemcpyAsync(q_out, d_q, sizeof(float) * sig->N, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
    rc = set_error(ctx, GACQ_ERR_HIP, "gacq_debug_row: D2H failed");
  if (rc == GACQ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "gacq_debug_row: sync failed");
  (void)hipFree(d_q);
  return rc;
}

}  // extern "C"
