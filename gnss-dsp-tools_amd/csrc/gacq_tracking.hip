// Batched code correlators for the tracking hand-off (SURVEY.md section 8f "next #4"): the reference's per-signal
//     correlate(x, prn, chips, frac, incr, c[, boc11])         gnsstools/gps/ca.py:120-128 (plain)
//     ... * boc11[int(bp)]                                      gnsstools/gps/l1cd.py:101-112 (BOC(1,1), boc11 = [1,-1])
//     ... * (0.953463 boc11[int(bp)] + 0.301511 boc11[int(bp6)])   gnsstools/galileo/e1b.py:45-58 (CBOC)
//     ... * boc11[int(bp6)] where tmboc_pattern[int(cp % 33)] else boc11[int(bp)]   gnsstools/gps/l1cp.py:210-228 (TMBOC)
//     ... * rz[int(rzp)]                                        gnsstools/gps/l2cm.py:81-92 (rz = [1,0]), l2cl.py (rz = [0,1])
// evaluated for MANY (satellite, tap) pairs over one block of samples in one launch (early/prompt/late of every tracked
// satellite).  The tracking loops themselves (FLL/PLL/DLL feedback, track-gps-l1.py:33-94) are sequential and stay on the host.
//
// The reference advances its phases by repeated fp64 addition (cp = (cp+incr) % L); here sample i uses the closed form
// floor(cp0 + incr*i) mod L.  The two differ only when the accumulated rounding (~1e-11 chips over a block) straddles a chip
// boundary, i.e. with probability ~1e-8 per correlator call; tests/test_tracking.py holds the result to 1e-5 of the reference.
#include "gacq_common.h"
#include "gacq_fft64.h"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace gacq;

namespace {

constexpr int kTrBlock = 256;
constexpr int kTrPer = 16;
constexpr int kTrChunk = kTrBlock * kTrPer;

// tmboc_pattern of gnsstools/gps/l1cp.py:202 as a bit mask (bits 0, 4, 6, 29 set)
constexpr unsigned long long kTmbocMask = (1ull << 0) | (1ull << 4) | (1ull << 6) | (1ull << 29);

struct CorrSpec { const uint8_t* chips; double cp0, bp0, bp60, incr; };

__global__ __launch_bounds__(kTrBlock) void correlate_partial_kernel(const float2* __restrict__ x, long n, const CorrSpec* __restrict__ specs,
                                                                      long L, int kind, int chunks, double2* __restrict__ partial) {
  __shared__ double s_re[kTrBlock / 64], s_im[kTrBlock / 64];
  const long blk = blockIdx.x;
  const int c = (int)(blk % chunks);
  const CorrSpec sp = specs[blk / chunks];
  double ar = 0.0, ai = 0.0;
  const long i0 = (long)c * kTrChunk + threadIdx.x;
  const double inv_l = 1.0 / (double)L;
  // fully unrolled: the 16 chip gathers and the 16 sample loads of a lane are independent, so one round trip covers them all (the call
  // is latency-bound: 36 workgroups on an otherwise idle chip)
#pragma unroll
  for (int j = 0; j < kTrPer; j++) {
    const long i = i0 + (long)j * kTrBlock;
    if (i < n) {
      const double di = (double)i;
      const double pos = sp.cp0 + __dmul_rn(sp.incr, di);
      // floor(pos) mod L without a 64-bit division: the quotient from an fp64 product can be off by one, fixed up by one compare each way
      long idx = (long)floor(pos) - (long)floor(pos * inv_l) * L;
      if (idx < 0) idx += L;
      if (idx >= L) idx -= L;
      float w = sp.chips[idx] ? -1.f : 1.f;                          // 1.0 - 2.0*c[int(cp)]
      if (kind != 0) {
        const long b1 = (long)floor(sp.bp0 + __dmul_rn(2.0 * sp.incr, di)) & 1;        // int(bp), bp = (bp + 2 incr) % 2
        if (kind == 1) {
          w *= b1 ? -1.f : 1.f;                                                        // boc11 = [1, -1]
        } else if (kind == 2 || kind == 3) {
          const long b6 = (long)floor(sp.bp60 + __dmul_rn(12.0 * sp.incr, di)) & 1;    // int(bp6)
          const float s1 = b1 ? -1.f : 1.f, s6 = b6 ? -1.f : 1.f;
          if (kind == 2) w *= 0.953463f * s1 + 0.301511f * s6;                         // CBOC
          else w *= ((kTmbocMask >> (idx % 33)) & 1ull) ? s6 : s1;                     // TMBOC: u = int(cp % 33)
        } else {
          w *= ((kind == 4) == (b1 == 0)) ? 1.f : 0.f;                                 // rz = [1,0] (kind 4) / [0,1] (kind 5)
        }
      }
      const float2 v = x[i];
      ar += (double)(v.x * w);
      ai += (double)(v.y * w);
    }
  }
  ar = gacq::f64::wave_add_f64(ar);                   // DPP: the ds_bpermute chain of a __shfl_down tree is a tenth of this kernel's time
  ai = gacq::f64::wave_add_f64(ai);
  if ((threadIdx.x & 63) == 0) { s_re[threadIdx.x >> 6] = ar; s_im[threadIdx.x >> 6] = ai; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kTrBlock / 64; w++) { ar += s_re[w]; ai += s_im[w]; }
    partial[blk] = make_double2(ar, ai);
  }
}

__global__ void correlate_finish_kernel(const double2* __restrict__ partial, double2* __restrict__ out, int K, int chunks) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double re = 0.0, im = 0.0;
  for (int c = 0; c < chunks; c++) { const double2 p = partial[(long)k * chunks + c]; re += p.x; im += p.y; }
  out[k] = make_double2(re, im);
}

double pymod(double a, double m) { double r = std::fmod(a, m); if (r != 0.0 && ((r < 0.0) != (m < 0.0))) r += m; return r; }

}  // namespace

// x_iq: host block, staged with the specs in one copy -- or d_xdev: the block ALREADY on the device (the front-end's output, the
// samples an acquisition just searched): only the K specs (48 bytes each) travel, which is what the reference's tracking loop
// amounts to once x is resident (track-gps-l1.py:48-50 calls correlate() three times per ms on the same block).
static int correlate_run(gacq_ctx* ctx, const float* x_iq, const float2* d_xdev, size_t n, const char* code, int kind, const int* prns,
                         const double* chips, const double* frac, const double* incr, int K, double* out_iq) {
  if (!ctx || (!x_iq && !d_xdev) || !code || !prns || !chips || !frac || !incr || !out_iq || K <= 0 || n == 0 || kind < 0 || kind > 5)
    return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_correlate_batch: bad argument");
  GACQ_DEVICE(ctx);
  hipStream_t st = ctx->stream;
  const int len = gacq_code_length(code);
  if (len < 0) return set_error(ctx, GACQ_ERR_UNKNOWN_CODE, "gacq_correlate_batch: unknown code '%s'", code);
  const long L = len;
  std::vector<CorrSpec> specs(K);
  // chip tables: the (code, PRN list) of the previous call is kept with its device pointers -- a tracking loop repeats it every block
  const bool same = ctx->tr_code == code && (int)ctx->tr_prns.size() == K && std::equal(prns, prns + K, ctx->tr_prns.begin());
  if (!same) {
    ctx->tr_code.clear();
    ctx->tr_chips.assign(K, nullptr);
    for (int k = 0; k < K; k++) {
      const std::string key = std::string("chips:") + code + ":" + std::to_string(prns[k]);
      const void* dchips = nullptr;
      auto it = ctx->tables.find(key);
      if (it == ctx->tables.end()) {
        std::vector<uint8_t> h(len);
        const int rc = gacq_code_chips(code, prns[k], h.data(), len);
        if (rc < 0) return set_error(ctx, rc, "gacq_correlate_batch: no PRN %d in '%s'", prns[k], code);
        const int rc2 = table_cache(ctx, key, h.data(), (size_t)len, &dchips);
        if (rc2 != GACQ_OK) return rc2;
      } else {
        dchips = it->second.p;
      }
      ctx->tr_chips[k] = dchips;
    }
    ctx->tr_prns.assign(prns, prns + K);
    ctx->tr_code = code;
  }
  for (int k = 0; k < K; k++) {
    const void* dchips = ctx->tr_chips[k];
    const double s = chips[k] + frac[k];
    specs[k].chips = (const uint8_t*)dchips;
    specs[k].cp0 = pymod(s, (double)L);            // cp  = (chips+frac) % code_length
    specs[k].bp0 = pymod(2.0 * s, 2.0);            // bp  = (2*(chips+frac)) % 2
    specs[k].bp60 = pymod(12.0 * s, 2.0);          // bp6 = (12*(chips+frac)) % 2
    specs[k].incr = incr[k];
    if (!(incr[k] >= 0.0) || incr[k] * (double)n > 4.0e15) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_correlate_batch: bad incr");
  }
  const int chunks = (int)((n + kTrChunk - 1) / kTrChunk);
  const size_t npart = (size_t)K * chunks;
  int rc;
  // one pinned staging block [specs | x] -> one H2D copy; results land in device-visible pinned memory (no D2H copy);
  // a single chunk per correlator (blocks up to kTrChunk samples, the tracking case) needs no second kernel
  const size_t spec_bytes = (sizeof(CorrSpec) * (size_t)K + 15) & ~(size_t)15;
  const size_t in_bytes = spec_bytes + (d_xdev ? 0 : sizeof(float2) * (size_t)n);
  if ((rc = ensure(ctx, ctx->xstage, in_bytes)) != GACQ_OK) return rc;
  if ((rc = ensure_pinned(ctx, ctx->pin_x, in_bytes)) != GACQ_OK) return rc;
  if ((rc = ensure_pinned(ctx, ctx->pin_peaks, sizeof(double2) * (size_t)K)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->partial, sizeof(double2) * npart)) != GACQ_OK) return rc;
  // device-resident block: the specs (48 bytes each) are all that travels -- written by the host through the PCIe BAR where the
  // device allows it, and the results are watched for in pinned memory instead of asking the runtime (the latency path of
  // gacq_search): one launch, no copy engine, no stream synchronisation
  const bool bar = d_xdev && bar_write(ctx, ctx->bar_s, specs.data(), sizeof(CorrSpec) * (size_t)K);
  if (!bar) {
    std::memcpy(ctx->pin_x.p, specs.data(), sizeof(CorrSpec) * (size_t)K);
    if (!d_xdev) std::memcpy((char*)ctx->pin_x.p + spec_bytes, x_iq, sizeof(float2) * (size_t)n);
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->xstage.p, ctx->pin_x.p, in_bytes, hipMemcpyHostToDevice, st));
  }
  const CorrSpec* d_specs = bar ? (const CorrSpec*)ctx->bar_s.p : (const CorrSpec*)ctx->xstage.p;
  const float2* d_x = d_xdev ? d_xdev : (const float2*)((const char*)ctx->xstage.p + spec_bytes);
  double2* d_out = (double2*)ctx->pin_peaks.p;
  double2* d_partial = chunks == 1 ? d_out : (double2*)ctx->partial.p;
  constexpr unsigned long long kSentinel = 0x7ff8dead0badbeefULL;      // a NaN payload no correlator sum produces
  const bool watch = bar && ctx->opt[GACQ_OPT_WATCH_RESULTS] != 0;
  if (watch)
    for (int k = 0; k < K; k++) { std::memcpy(&d_out[k].x, &kSentinel, 8); std::memcpy(&d_out[k].y, &kSentinel, 8); }
  hipLaunchKernelGGL(correlate_partial_kernel, dim3((unsigned)npart), dim3(kTrBlock), 0, st, d_x, (long)n, d_specs, L, kind, chunks,
                     d_partial);
  GACQ_HIP(ctx, hipGetLastError());
  if (chunks > 1) {
    hipLaunchKernelGGL(correlate_finish_kernel, dim3((unsigned)((K + 127) / 128)), dim3(128), 0, st, (const double2*)d_partial, d_out, K,
                       chunks);
    GACQ_HIP(ctx, hipGetLastError());
  }
  // both halves of every record: a 16-byte store may land as two 8-byte ones
  const volatile unsigned long long* w = reinterpret_cast<const volatile unsigned long long*>(d_out);
  if (!(watch && watch_records(w, (size_t)K, 2, 0, kSentinel, 200) && watch_records(w, (size_t)K, 2, 1, kSentinel, 200)))
    GACQ_HIP(ctx, hipStreamSynchronize(st));
  std::memcpy(out_iq, ctx->pin_peaks.p, sizeof(double2) * (size_t)K);
  return GACQ_OK;
}

extern "C" int gacq_correlate_batch(gacq_ctx* ctx, const float* x_iq, size_t n, const char* code, int kind, const int* prns,
                                    const double* chips, const double* frac, const double* incr, int K, double* out_iq) {
  return correlate_run(ctx, x_iq, nullptr, n, code, kind, prns, chips, frac, incr, K, out_iq);
}

extern "C" int gacq_correlate_batch_dev(gacq_ctx* ctx, const void* d_x, size_t n, const char* code, int kind, const int* prns,
                                        const double* chips, const double* frac, const double* incr, int K, double* out_iq) {
  if (!d_x) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_correlate_batch_dev: d_x is NULL");
  return correlate_run(ctx, nullptr, (const float2*)d_x, n, code, kind, prns, chips, frac, incr, K, out_iq);
}
