// Front-end of the acquire scripts on the GPU (SURVEY.md section 8f "next #1", component C10):
//   int8 I/Q  ->  carrier-offset wipe-off with the fixed-point table NCO  ->  161-tap FIR applied forward-backward
//   (scipy.signal.filtfilt semantics)  ->  linear-interpolation resample to the signal's internal rate.
// Reference: acquire-gps-l1.py:78-96, gnsstools/io.py:3-12, gnsstools/nco.py:30-41.
//
// All three kernels are streaming kernels (HBM-bound at the input rate; the FIR keeps its taps and a tile + halo in LDS).
//   fe_mix_kernel       x[i] = (I + jQ) * table[((dp + i*df) >> 50) & 1023]        nco.mix_: 50-bit fixed-point phase, int64 wrap
//   fe_fir_kernel<DIR>  one direction of filtfilt over the odd-extended signal, history initialised to the edge value
//                       (that is what filtfilt's lfilter_zi initial condition means for an FIR)
//   fe_resample_kernel  np.interp at t_k = (1/fsr) * k, fp64 positions
#include "gacq_common.h"
#include "gacq_cplx.h"

#include <cmath>

using namespace gacq;

namespace {

constexpr int kFeBlock = 256;
constexpr int kMaxTaps = 512;

// nco.mix_ for sample i: (I + jQ) * table[((dp + i*df) >> 50) & 1023]   (gnsstools/nco.py:30-41); one expression for every kernel
// that mixes, so that the stand-alone mix kernel and the FIR kernel that mixes while it loads its tile produce the same bits
__device__ __forceinline__ float2 mix_sample(const char2* __restrict__ iq, long i, long long dp, long long df, const float2* __restrict__ tab) {
  const char2 s = iq[i];
  const unsigned long long ph = (unsigned long long)dp + (unsigned long long)i * (unsigned long long)df;   // wraps like int64
  const float2 w = tab[(ph >> 50) & (kNcoTableSize - 1)];
  const float re = (float)(signed char)s.x, im = (float)(signed char)s.y;
  // products and FMAs spelled out: left to fp-contract, the two kernels this is inlined into could fuse different halves
  return make_float2(__builtin_fmaf(re, w.x, -(im * w.y)), __builtin_fmaf(re, w.y, im * w.x));
}

__global__ __launch_bounds__(kFeBlock) void fe_mix_kernel(const char2* __restrict__ iq, float2* __restrict__ out, long n, long long dp,
                                                           long long df, const float2* __restrict__ tab) {
  const long i = (long)blockIdx.x * kFeBlock + threadIdx.x;
  if (i >= n) return;
  out[i] = mix_sample(iq, i, dp, df, tab);
}

// the mixed input as an indexable source for odd_ext_at (the FIR kernel that reads int8 samples directly)
struct MixedInput {
  const char2* iq;
  long long dp, df;
  const float2* tab;
  __device__ __forceinline__ float2 operator[](long i) const { return mix_sample(iq, i, dp, df, tab); }
};

// value of the odd extension of x (length n, pad p) at extended index j in [0, n + 2p)   (scipy.signal._arraytools.odd_ext)
template <typename Src>
__device__ __forceinline__ float2 odd_ext_at(const Src& x, long n, int p, long j) {
  if (j < p) {
    const float2 e = x[0], v = x[p - j];
    return make_float2(2.f * e.x - v.x, 2.f * e.y - v.y);
  }
  if (j >= n + p) {
    const float2 e = x[n - 1], v = x[2 * (n - 1) - (j - p)];
    return make_float2(2.f * e.x - v.x, 2.f * e.y - v.y);
  }
  return x[j - p];
}

// PASS 1 (forward):  y1[j] = sum_k h[k] * e[j-k],  e = odd extension, e[m<0] := e[0];   j in [0, L), L = n + 2p
// PASS 2 (backward): y2[j] = sum_k h[k] * y1[j+k], y1[m>=L] := y1[L-1];                 j in [p, p+n) -> out[j-p]
// Each thread produces kOut adjacent outputs from a sliding register window: one LDS read feeds kOut taps' worth of
// FMAs.  The tile is stored with one pad element per 32 (phys = i + i/32) so the stride-kOut lane pattern is
// bank-conflict free; the taps are wave-uniform scalar loads.
constexpr int kOut = 4;
constexpr int kTile = kFeBlock * kOut;
__device__ __forceinline__ int phys(int i) { return i + (i >> 5); }

template <int PASS>
__global__ __launch_bounds__(kFeBlock) void fe_fir_kernel(const float2* __restrict__ in, float2* __restrict__ out, long n, int p,
                                                           const float* __restrict__ taps, int ntaps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* s_x = reinterpret_cast<float2*>(smem);
  const long L = n + 2 * (long)p;
  const long j0 = (long)blockIdx.x * kTile + (PASS == 1 ? 0 : p);      // first output index of this tile (extended coords)
  const int halo = ntaps - 1;
  // tile of inputs: PASS 1 needs e[j0-halo .. j0+kTile-1], PASS 2 needs y1[j0 .. j0+kTile-1+halo]
  for (int m = threadIdx.x; m < kTile + halo; m += kFeBlock) {
    const long idx = (PASS == 1) ? (j0 - halo + m) : (j0 + m);
    float2 v;
    if (PASS == 1) v = odd_ext_at(in, n, p, idx < 0 ? 0 : (idx >= L ? L - 1 : idx));
    else v = in[idx >= L ? L - 1 : idx];
    s_x[phys(m)] = v;
  }
  __syncthreads();
  const int r = threadIdx.x * kOut;
  float ar[kOut], ai[kOut];
#pragma unroll
  for (int c = 0; c < kOut; c++) { ar[c] = 0.f; ai[c] = 0.f; }
  float2 w[kOut];
  if (PASS == 1) {
    // out_c = sum_k h[k] * s[r + c + halo - k]; window w[c] = s[r + c + halo - k]
#pragma unroll
    for (int c = 0; c < kOut; c++) w[c] = s_x[phys(r + c + halo)];
    for (int k = 0; k < ntaps; k++) {
      const float hk = taps[k];
#pragma unroll
      for (int c = 0; c < kOut; c++) { ar[c] = fmaf(hk, w[c].x, ar[c]); ai[c] = fmaf(hk, w[c].y, ai[c]); }
#pragma unroll
      for (int c = kOut - 1; c > 0; c--) w[c] = w[c - 1];                 // next k: every index moves down by one
      const int nxt = r + halo - (k + 1);
      w[0] = s_x[phys(nxt < 0 ? 0 : nxt)];
    }
  } else {
    // out_c = sum_k h[k] * s[r + c + k]; window w[c] = s[r + c + k]
#pragma unroll
    for (int c = 0; c < kOut; c++) w[c] = s_x[phys(r + c)];
    for (int k = 0; k < ntaps; k++) {
      const float hk = taps[k];
#pragma unroll
      for (int c = 0; c < kOut; c++) { ar[c] = fmaf(hk, w[c].x, ar[c]); ai[c] = fmaf(hk, w[c].y, ai[c]); }
#pragma unroll
      for (int c = 0; c < kOut - 1; c++) w[c] = w[c + 1];
      const int nxt = r + kOut + k;                                        // = r + (kOut-1) + (k+1)
      w[kOut - 1] = s_x[phys(nxt > kTile + halo - 1 ? kTile + halo - 1 : nxt)];
    }
  }
  const long jend = (PASS == 1) ? L : (long)p + n;
#pragma unroll
  for (int c = 0; c < kOut; c++) {
    const long j = j0 + r + c;
    if (j < jend) out[PASS == 1 ? j : j - p] = make_float2(ar[c], ai[c]);
  }
}

// The reference's filter length (161 taps in every acquire script, acquire-gps-l1.py:89) gets its own instantiation: five
// outputs per thread, so that the lane stride of the window reads is odd (5 elements = 10 banks: conflict-free without padding),
// the tap loop fully unrolled so that every LDS read is `ds_read_b64 v, vaddr offset:imm` from one per-thread base address, the
// taps in SGPRs, and the accumulators as (re, im) pairs: one v_pk_fma_f32 per output and tap with the tap broadcast from its SGPR.
// Same products, same summation order (k ascending) as the generic kernel: bit-identical outputs, 2.4 x faster
// (the generic loop spends more instructions on window moves and padded-address arithmetic than on FMAs).
constexpr int kOutF = 5;
constexpr int kTileF = kFeBlock * kOutF;
// Src: const float2* (a buffer), or MixedInput for PASS 1 -- then the carrier wipe-off happens while the tile is loaded and the
// mixed signal never exists in HBM (one launch and 95 MB of traffic per 6 M samples less).
template <int PASS, int NTAPS, typename Src>
__global__ __launch_bounds__(kFeBlock) void fe_fir_fixed_kernel(const Src in, float2* __restrict__ out, long n, int p,
                                                                 const float* __restrict__ taps) {
  __shared__ v2 s_x[kTileF + NTAPS - 1];
  constexpr int halo = NTAPS - 1;
  const long L = n + 2 * (long)p;
  const long j0 = (long)blockIdx.x * kTileF + (PASS == 1 ? 0 : p);
  for (int m = threadIdx.x; m < kTileF + halo; m += kFeBlock) {
    const long idx = (PASS == 1) ? (j0 - halo + m) : (j0 + m);
    float2 v;
    if (PASS == 1) v = odd_ext_at(in, n, p, idx < 0 ? 0 : (idx >= L ? L - 1 : idx));
    else v = in[idx >= L ? L - 1 : idx];
    s_x[m] = v2{v.x, v.y};
  }
  __syncthreads();
  const v2* base = s_x + threadIdx.x * kOutF;
  v2 acc[kOutF], w[kOutF];
#pragma unroll
  for (int c = 0; c < kOutF; c++) acc[c] = v2{0.f, 0.f};
  // Taps go in chunks of kChunk (a multiple of kOutF, so the rotating window is back in its starting slots at every chunk
  // boundary): inside a chunk everything is unrolled and every LDS offset is an immediate; a fully unrolled 161-tap body made
  // hipcc hoist all 165 LDS reads to the top (256 VGPRs + spills).
  constexpr int kChunk = 4 * kOutF;
  constexpr int kMain = (NTAPS / kChunk) * kChunk;
#define GACQ_FIR_FMA(C, SLOT, HK) acc[C] = v2{fmaf(HK, w[SLOT].x, acc[C].x), fmaf(HK, w[SLOT].y, acc[C].y)}
  if (PASS == 1) {
    // out_c = sum_k h[k] * s[r + c + halo - k]: the window w[c] = s[r + c + halo - k] moves down by one element per tap
#pragma unroll
    for (int c = 0; c < kOutF; c++) w[c] = base[c + halo];
    const v2* bp = base + halo;                                        // s[r + halo - k0]
    for (int k0 = 0; k0 < kMain; k0 += kChunk, bp -= kChunk) {
#pragma unroll
      for (int kk = 0; kk < kChunk; kk++) {
        const float hk = taps[k0 + kk];
#pragma unroll
        for (int c = 0; c < kOutF; c++) GACQ_FIR_FMA(c, (c + kChunk - kk) % kOutF, hk);
        w[(kOutF - 1 + kChunk - kk) % kOutF] = bp[-(kk + 1)];           // s[r + halo - (k+1)] replaces the element output kOutF-1 just used
      }
    }
#pragma unroll
    for (int kk = 0; kk < NTAPS - kMain; kk++) {
      const float hk = taps[kMain + kk];
#pragma unroll
      for (int c = 0; c < kOutF; c++) GACQ_FIR_FMA(c, (c + kChunk - kk) % kOutF, hk);
      if (kk + 1 < NTAPS - kMain) w[(kOutF - 1 + kChunk - kk) % kOutF] = bp[-(kk + 1)];
    }
  } else {
    // out_c = sum_k h[k] * s[r + c + k]: the window w[c] = s[r + c + k] moves up by one element per tap
#pragma unroll
    for (int c = 0; c < kOutF; c++) w[c] = base[c];
    const v2* bp = base + kOutF;                                       // s[r + kOutF + k0]
    for (int k0 = 0; k0 < kMain; k0 += kChunk, bp += kChunk) {
#pragma unroll
      for (int kk = 0; kk < kChunk; kk++) {
        const float hk = taps[k0 + kk];
#pragma unroll
        for (int c = 0; c < kOutF; c++) GACQ_FIR_FMA(c, (c + kk) % kOutF, hk);
        w[kk % kOutF] = bp[kk];                                        // slot of output 0 at tap k now holds s[r + kOutF + k]
      }
    }
#pragma unroll
    for (int kk = 0; kk < NTAPS - kMain; kk++) {
      const float hk = taps[kMain + kk];
#pragma unroll
      for (int c = 0; c < kOutF; c++) GACQ_FIR_FMA(c, (c + kk) % kOutF, hk);
      if (kk + 1 < NTAPS - kMain) w[kk % kOutF] = bp[kk];
    }
  }
#undef GACQ_FIR_FMA
  const long jend = (PASS == 1) ? L : (long)p + n;
#pragma unroll
  for (int c = 0; c < kOutF; c++) {
    const long j = j0 + threadIdx.x * kOutF + c;
    if (j < jend) out[PASS == 1 ? j : j - p] = make_float2(acc[c].x, acc[c].y);
  }
}

__global__ __launch_bounds__(kFeBlock) void fe_resample_kernel(const float2* __restrict__ y, long n, float2* __restrict__ out, long nout,
                                                                double step) {
  const long k = (long)blockIdx.x * kFeBlock + threadIdx.x;
  if (k >= nout) return;
  const double t = __dmul_rn(step, (double)k);          // (1/fsr)*np.arange(...)   acquire-gps-l1.py:94
  float2 o;
  if (t >= (double)(n - 1)) {
    o = y[n - 1];                                      // np.interp clamps to fp[-1] right of the last sample
  } else {
    const long i = (long)floor(t);
    const float fr = (float)(t - (double)i);
    const float2 a = y[i], b = y[i + 1];
    o = make_float2(fmaf(b.x - a.x, fr, a.x), fmaf(b.y - a.y, fr, a.y));   // slope*(x - xp[i]) + fp[i]
  }
  out[k] = o;
}

}  // namespace

namespace gacq {

// nco.mix(x, -coffset/fs, 0) on int8 I/Q already on the device: dp = floor(p*NT*2^50) = 0, df = floor(f*NT*2^50)   gnsstools/nco.py:33-34
static long long frontend_mix_step(double fs_in, double carrier_offset_hz) {
  const double f = -carrier_offset_hz / fs_in;
  return (long long)std::floor(f * (double)kNcoTableSize * (double)(1LL << 50));
}

int frontend_mix(gacq_ctx* ctx, const void* d_iq_int8, long n, double fs_in, double carrier_offset_hz, float2* d_out) {
  const long long df = frontend_mix_step(fs_in, carrier_offset_hz);
  hipLaunchKernelGGL(fe_mix_kernel, dim3((unsigned)((n + kFeBlock - 1) / kFeBlock)), dim3(kFeBlock), 0, ctx->stream, (const char2*)d_iq_int8, d_out, n,
                     0LL, df, (const float2*)ctx->tab.p);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace gacq

extern "C" {

// scipy.signal.firwin(ntaps, cutoff_norm, window='hann') (low-pass, unity DC gain); cutoff_norm = cutoff / (fs/2)
int gacq_firwin_hann(int ntaps, double cutoff_norm, double* taps) {
  if (ntaps < 3 || ntaps > kMaxTaps || !(cutoff_norm > 0.0 && cutoff_norm < 1.0) || !taps)
    return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_firwin_hann: bad argument");
  const double alpha = 0.5 * (ntaps - 1);
  double sum = 0.0;
  for (int i = 0; i < ntaps; i++) {
    const double m = (double)i - alpha;
    const double arg = cutoff_norm * m;
    const double sinc = (arg == 0.0) ? 1.0 : std::sin(M_PI * arg) / (M_PI * arg);
    const double win = 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)i / (double)(ntaps - 1));      // symmetric Hann
    taps[i] = cutoff_norm * sinc * win;
    sum += taps[i];
  }
  for (int i = 0; i < ntaps; i++) taps[i] /= sum;
  return GACQ_OK;
}

int gacq_frontend_dev(gacq_ctx* ctx, const void* d_iq_int8, size_t nsamp_in, double fs_in, double carrier_offset_hz,
                      const double* taps, int ntaps, double fs_out, size_t nsamp_out, void* d_out) {
  if (!ctx || !d_iq_int8 || !taps || !d_out || ntaps < 1 || ntaps > kMaxTaps || !(fs_in > 0.0) || !(fs_out > 0.0) || nsamp_out == 0)
    return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_frontend_dev: bad argument");
  const int p = 3 * ntaps;                              // filtfilt default padlen = 3*max(len(a), len(b))
  if (nsamp_in <= (size_t)p)
    return set_error(ctx, GACQ_ERR_SHORT_INPUT, "gacq_frontend_dev: %zu input samples, filtfilt needs more than %d", nsamp_in, p);
  GACQ_DEVICE(ctx);
  hipStream_t st = ctx->stream;
  const long n = (long)nsamp_in, L = n + 2L * p;
  int rc;
  if ((rc = ensure(ctx, ctx->fe_a, sizeof(float2) * (size_t)n)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->fe_b, sizeof(float2) * (size_t)L)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->fe_taps, sizeof(float) * kMaxTaps)) != GACQ_OK) return rc;
  std::vector<float> h(ntaps);
  for (int i = 0; i < ntaps; i++) h[i] = (float)taps[i];
  if (h != ctx->up_taps) {
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->fe_taps.p, h.data(), sizeof(float) * ntaps, hipMemcpyHostToDevice, st));
    GACQ_HIP(ctx, hipStreamSynchronize(st));            // h dies with this frame; a repeated filter skips copy and sync
    ctx->up_taps = h;
  }
  float2* a = (float2*)ctx->fe_a.p;
  float2* b = (float2*)ctx->fe_b.p;
  if (ntaps == 161 && !ctx->opt[GACQ_OPT_FE_GENERIC]) {
    // the reference's filter length: mix + forward pass in one kernel, then the backward pass (fe_fir_fixed_kernel)
    MixedInput src;
    src.iq = (const char2*)d_iq_int8;
    src.dp = 0LL;                                                                      // nco.mix(x, f, 0): phase 0
    src.df = gacq::frontend_mix_step(fs_in, carrier_offset_hz);
    src.tab = (const float2*)ctx->tab.p;
    hipLaunchKernelGGL((fe_fir_fixed_kernel<1, 161, MixedInput>), dim3((unsigned)((L + kTileF - 1) / kTileF)), dim3(kFeBlock), 0, st, src, b, n, p,
                       (const float*)ctx->fe_taps.p);
    GACQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL((fe_fir_fixed_kernel<2, 161, const float2*>), dim3((unsigned)((n + kTileF - 1) / kTileF)), dim3(kFeBlock), 0, st,
                       (const float2*)b, a, n, p, (const float*)ctx->fe_taps.p);
    GACQ_HIP(ctx, hipGetLastError());
  } else {
    if ((rc = frontend_mix(ctx, d_iq_int8, n, fs_in, carrier_offset_hz, a)) != GACQ_OK) return rc;
    const int tile_elems = kTile + ntaps;
    const size_t smem = sizeof(float2) * (size_t)(tile_elems + tile_elems / 32 + 2);
    hipLaunchKernelGGL(fe_fir_kernel<1>, dim3((unsigned)((L + kTile - 1) / kTile)), dim3(kFeBlock), smem, st, (const float2*)a, b, n, p,
                       (const float*)ctx->fe_taps.p, ntaps);
    GACQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(fe_fir_kernel<2>, dim3((unsigned)((n + kTile - 1) / kTile)), dim3(kFeBlock), smem, st, (const float2*)b, a, n, p,
                       (const float*)ctx->fe_taps.p, ntaps);
    GACQ_HIP(ctx, hipGetLastError());
  }
  const double fsr = fs_out / fs_in;                    // acquire-gps-l1.py:91
  hipLaunchKernelGGL(fe_resample_kernel, dim3((unsigned)((nsamp_out + kFeBlock - 1) / kFeBlock)), dim3(kFeBlock), 0, st, (const float2*)a, n,
                     (float2*)d_out, (long)nsamp_out, 1.0 / fsr);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // extern "C"
