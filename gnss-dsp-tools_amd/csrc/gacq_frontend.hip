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

#include <cmath>

using namespace gacq;

namespace {

constexpr int kFeBlock = 256;
constexpr int kMaxTaps = 512;

__global__ __launch_bounds__(kFeBlock) void fe_mix_kernel(const char2* __restrict__ iq, float2* __restrict__ out, long n, long long dp,
                                                           long long df, const float2* __restrict__ tab) {
  const long i = (long)blockIdx.x * kFeBlock + threadIdx.x;
  if (i >= n) return;
  const char2 s = iq[i];
  const unsigned long long ph = (unsigned long long)dp + (unsigned long long)i * (unsigned long long)df;   // wraps like int64
  const float2 w = tab[(ph >> 50) & (kNcoTableSize - 1)];
  const float re = (float)(signed char)s.x, im = (float)(signed char)s.y;
  out[i] = make_float2(re * w.x - im * w.y, re * w.y + im * w.x);
}

// value of the odd extension of x (length n, pad p) at extended index j in [0, n + 2p)   (scipy.signal._arraytools.odd_ext)
__device__ __forceinline__ float2 odd_ext_at(const float2* __restrict__ x, long n, int p, long j) {
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
int frontend_mix(gacq_ctx* ctx, const void* d_iq_int8, long n, double fs_in, double carrier_offset_hz, float2* d_out) {
  const double f = -carrier_offset_hz / fs_in;
  const long long df = (long long)std::floor(f * (double)kNcoTableSize * (double)(1LL << 50));
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
  if ((rc = frontend_mix(ctx, d_iq_int8, n, fs_in, carrier_offset_hz, a)) != GACQ_OK) return rc;
  const int tile_elems = kTile + ntaps;
  const size_t smem = sizeof(float2) * (size_t)(tile_elems + tile_elems / 32 + 2);
  hipLaunchKernelGGL(fe_fir_kernel<1>, dim3((unsigned)((L + kTile - 1) / kTile)), dim3(kFeBlock), smem, st, (const float2*)a, b, n, p,
                     (const float*)ctx->fe_taps.p, ntaps);
  GACQ_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(fe_fir_kernel<2>, dim3((unsigned)((n + kTile - 1) / kTile)), dim3(kFeBlock), smem, st, (const float2*)b, a, n, p,
                     (const float*)ctx->fe_taps.p, ntaps);
  GACQ_HIP(ctx, hipGetLastError());
  const double fsr = fs_out / fs_in;                    // acquire-gps-l1.py:91
  hipLaunchKernelGGL(fe_resample_kernel, dim3((unsigned)((nsamp_out + kFeBlock - 1) / kFeBlock)), dim3(kFeBlock), 0, st, (const float2*)a, n,
                     (float2*)d_out, (long)nsamp_out, 1.0 / fsr);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // extern "C"
