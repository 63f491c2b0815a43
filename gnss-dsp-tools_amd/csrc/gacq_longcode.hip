// Time-domain long-code searches (SURVEY.md section 8f "next #3", component C12): acquire-gps-l2cl.py:15-30 (75 L2CM-period
// offsets of the 767 250-chip L2CL code) and acquire-glonass-l1-p.py / -l2-p.py:15-33 (1000 C/A-period offsets of the
// 5.11 M-chip P code), given a prior FFT acquisition.  For every candidate k:
//     q[k] = sum_block | sum_i x[n*block + i] * (1 - 2*chips[floor(phase0[k][block] + incr*i) mod L]) * w[i] |
// with w the table NCO restarted per block.  x*w does not depend on k, so it is formed once (longcode_mix_kernel) and every
// candidate is a +-1 weighted sum over it (longcode_dot_kernel, fp64 accumulation, code indices in fp64 like numpy's).
#include "gacq_common.h"
#include "gacq_fft64.h"

#include <cmath>
#include <cstring>

using namespace gacq;

namespace {

constexpr int kLcBlock = 256;
constexpr int kLcPer = 16;                       // samples per thread per chunk
constexpr int kLcChunk = kLcBlock * kLcPer;     // samples per workgroup

// xw[b*n + i] = x[b*n + i] * tab[floor((f*i)*1024) & 1023]      (nco.nco(f,0,n), same w for every block)
__global__ __launch_bounds__(kLcBlock) void longcode_mix_kernel(const float2* __restrict__ x, float2* __restrict__ xw, long n, int B,
                                                                 double f, const float2* __restrict__ tab) {
  const long g = (long)blockIdx.x * kLcBlock + threadIdx.x;
  if (g >= n * B) return;
  const long i = g % n;
  const int k = nco_index(f, (int)i);
  const float2 s = x[g], w = tab[k];
  xw[g] = make_float2(s.x * w.x - s.y * w.y, s.x * w.y + s.y * w.x);
}

// partial[(k*B + b)*chunks + c] = sum over the chunk's samples of +-xw.  Workgroup = (chunk of 4096 samples, block, group of kc <= kLcKc
// candidates): a thread keeps its 16 samples (as fp64 pairs) and their incr*i products in registers and runs every candidate of the
// group over them -- the samples cross L2 once per group instead of once per candidate, the per-sample product is formed once, and the
// candidates share one barrier.  Per candidate and sample what is left is the index of the reference (floor(ph + incr*i) in fp64, np.mod
// wrap, gnsstools/gps/l2cl.py:57-61), one byte of code, a sign flip and two fp64 additions; the wave sums run on DPP (gacq_fft64.h).
// Round 4: GLONASS P (1000 x 5 x 65536) 0.80 -> see profiles/r04_longcode_dot_kernel.log.
constexpr int kLcKc = 8;
#ifndef GACQ_LC_MINWG
#define GACQ_LC_MINWG 1024      // workgroups below which the candidate group is halved (L2CL, 75 candidates: 79.7 us per call at 1024, 84 at 512 or 2048, 93 at 4096)
#endif
__global__ __launch_bounds__(kLcBlock) void longcode_dot_kernel(const float2* __restrict__ xw, const uint8_t* __restrict__ chips, int L,
                                                                 const double* __restrict__ phase0, double incr, long n, int B,
                                                                 int chunks, int K, int kc, double2* __restrict__ partial) {
  __shared__ double s_re[kLcKc][kLcBlock / 64], s_im[kLcKc][kLcBlock / 64];
  unsigned blk = blockIdx.x;
  const int c = (int)(blk % (unsigned)chunks);
  blk /= (unsigned)chunks;
  const int b = (int)(blk % (unsigned)B);
  const int k0 = (int)(blk / (unsigned)B) * kc;
  const int nk = min(kc, K - k0);
  const float2* src = xw + (long)b * n;
  const long i0 = (long)c * kLcChunk + threadIdx.x;
  double vx[kLcPer], vy[kLcPer], prod[kLcPer];
#pragma unroll
  for (int j = 0; j < kLcPer; j++) {
    const long i = i0 + (long)j * kLcBlock;
    const float2 v = (i < n) ? src[i] : make_float2(0.f, 0.f);       // a sample past the block adds +-0
    vx[j] = (double)v.x;
    vy[j] = (double)v.y;
    prod[j] = __dmul_rn(incr, (double)(i < n ? i : 0));
  }
  for (int kk = 0; kk < nk; kk++) {
    const double ph = phase0[(long)(k0 + kk) * B + b];
    double ar = 0.0, ai = 0.0;
#pragma unroll
    for (int j = 0; j < kLcPer; j++) {
      // idx = floor((chips % L) + frac + incr*i) mod L, fp64 as numpy does it; |ph + incr*i| < 2^31 for every code length in the table.
      // (A branch-free reduction -- conditional -L, conditional +L, rare cases redone -- lets the compiler batch the 16 byte loads, needs
      // 130+ registers for that and is 12 % slower: 268 against 238 us for the GLONASS P shape.)
      int idx = (int)floor(ph + prod[j]);
      if (idx >= L) { idx -= L; if (idx >= L) idx %= L; }
      else if (idx < 0) { idx %= L; if (idx < 0) idx += L; }          // np.mod is floored: a negative start phase wraps upwards
      const unsigned long long flip = (unsigned long long)chips[idx] << 63;       // chip 1 -> -x
      ar += __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, vx[j]) ^ flip);
      ai += __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, vy[j]) ^ flip);
    }
    ar = gacq::f64::wave_add_f64(ar);
    ai = gacq::f64::wave_add_f64(ai);
    if ((threadIdx.x & 63) == 0) { s_re[kk][threadIdx.x >> 6] = ar; s_im[kk][threadIdx.x >> 6] = ai; }
  }
  __syncthreads();
  if ((int)threadIdx.x < nk) {
    double ar = s_re[threadIdx.x][0], ai = s_im[threadIdx.x][0];
    for (int w = 1; w < kLcBlock / 64; w++) { ar += s_re[threadIdx.x][w]; ai += s_im[threadIdx.x][w]; }
    partial[((long)(k0 + (int)threadIdx.x) * B + b) * chunks + c] = make_double2(ar, ai);
  }
}

// q[k] = sum_b | sum_c partial |          (np.absolute(np.sum(p)) accumulated over blocks)
__global__ void longcode_finish_kernel(const double2* __restrict__ partial, double* __restrict__ q, int K, int B, int chunks) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double acc = 0.0;
  for (int b = 0; b < B; b++) {
    double re = 0.0, im = 0.0;
    for (int c = 0; c < chunks; c++) { const double2 p = partial[((long)k * B + b) * chunks + c]; re += p.x; im += p.y; }
    acc += sqrt(re * re + im * im);
  }
  q[k] = acc;
}

int device_chips(gacq_ctx* ctx, const char* code, int prn, const uint8_t** out, long* L) {
  const int len = gacq_code_length(code);
  if (len < 0) return set_error(ctx, GACQ_ERR_UNKNOWN_CODE, "long-code search: unknown code '%s'", code);
  const std::string key = std::string("chips:") + code + ":" + std::to_string(prn);
  *L = len;
  auto it = ctx->tables.find(key);
  if (it != ctx->tables.end()) { *out = (const uint8_t*)it->second.p; return GACQ_OK; }
  std::vector<uint8_t> h(len);
  const int rc = gacq_code_chips(code, prn, h.data(), len);
  if (rc < 0) return set_error(ctx, rc, "long-code search: no PRN %d in '%s'", prn, code);
  const void* p = nullptr;
  const int rc2 = table_cache(ctx, key, h.data(), (size_t)len, &p);
  *out = (const uint8_t*)p;
  return rc2;
}

}  // namespace

// x: host complex64 after the caller's carrier-offset wipe-off (iq_int8 == nullptr), or the raw interleaved int8 I/Q of the
// file, wiped off here on the device with the front-end's fixed-point NCO (nco.mix(x,-coffset/fs,0), acquire-gps-l2cl.py:72)
// d_x: complex64 samples ALREADY on the device (the front-end's output, or gacq_mix_int8_dev's): nothing is uploaded but the K x blocks
// start phases, and the reference's "one x kept in memory from acquisition into the long-code search" (acquire-gps-l2cl.py:60-76)
// holds on the GPU as well.
static int longcode_run(gacq_ctx* ctx, const float* x_iq, const int8_t* iq_int8, double coffset_hz, size_t nsamp, double fs, const char* code,
                        int prn, double carrier_hz, const double* phase0, int K, int blocks, int n, double* q_out, const float2* d_x = nullptr) {
  if (!ctx || (!x_iq && !iq_int8 && !d_x) || !code || !phase0 || !q_out || K <= 0 || blocks < 0 || n <= 0 || !(fs > 0.0) || !std::isfinite(coffset_hz))
    return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_longcode_search: bad argument");
  if (blocks == 0) { memset(q_out, 0, sizeof(double) * K); return GACQ_OK; }
  if (nsamp < (size_t)blocks * n)
    return set_error(ctx, GACQ_ERR_SHORT_INPUT, "gacq_longcode_search: %zu samples given, %zu needed", nsamp, (size_t)blocks * n);
  const double f = -carrier_hz / fs;                                     // nco.nco(-doppler/fs,0,n)  (acquire-gps-l2cl.py:18)
  if (!std::isfinite(f) || !nco_range_ok(std::fabs(f), n))
    return set_error(ctx, GACQ_ERR_UNSUPPORTED, "gacq_longcode_search: NCO phase index out of range");
  GACQ_DEVICE(ctx);
  hipStream_t st = ctx->stream;
  const uint8_t* d_chips;
  long L;
  int rc = device_chips(ctx, code, prn, &d_chips, &L);
  if (rc != GACQ_OK) return rc;
  const double chip_rate = gacq_code_chip_rate(code);
  const double incr = chip_rate / fs;                                   // l2cl.chip_rate/fs  (acquire-gps-l2cl.py:19)
  // the kernel forms floor(phase + incr * i) as a 32-bit integer: start phases as the reference builds them are below two code periods
  const double reach = 2147483000.0 - incr * (double)n;
  for (size_t i = 0; i < (size_t)K * blocks; i++)
    if (!(std::fabs(phase0[i]) < reach))
      return set_error(ctx, GACQ_ERR_BAD_ARG, "long-code search: start phase %g of candidate %zu is not finite or beyond +-2^31 chips", phase0[i], i / blocks);
  const long total = (long)n * blocks;
  const int chunks = (n + kLcChunk - 1) / kLcChunk;
  const size_t npart = (size_t)K * blocks * chunks;
  if (!d_x && (rc = ensure(ctx, ctx->xstage, sizeof(float2) * (size_t)total)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->fe_a, sizeof(float2) * (size_t)total)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->partial, sizeof(double2) * npart + sizeof(double) * (size_t)K * (blocks + 1))) != GACQ_OK) return rc;
  double2* d_partial = (double2*)ctx->partial.p;
  double* d_phase = (double*)(d_partial + npart);
  double* d_q = d_phase + (size_t)K * blocks;
  if (d_x) {
    // device-resident input: nothing to stage
  } else if (iq_int8) {
    // the int8 pairs are staged in the (not yet used) mixed-block buffer, wiped off into xstage as complex64
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->fe_a.p, iq_int8, 2 * (size_t)total, hipMemcpyHostToDevice, st));
    if ((rc = frontend_mix(ctx, ctx->fe_a.p, total, fs, coffset_hz, (float2*)ctx->xstage.p)) != GACQ_OK) return rc;
  } else {
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->xstage.p, x_iq, sizeof(float2) * (size_t)total, hipMemcpyHostToDevice, st));
  }
  GACQ_HIP(ctx, hipMemcpyAsync(d_phase, phase0, sizeof(double) * (size_t)K * blocks, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(longcode_mix_kernel, dim3((unsigned)((total + kLcBlock - 1) / kLcBlock)), dim3(kLcBlock), 0, st,
                     d_x ? d_x : (const float2*)ctx->xstage.p, (float2*)ctx->fe_a.p, (long)n, blocks, f, (const float2*)ctx->tab.p);
  GACQ_HIP(ctx, hipGetLastError());
  // candidates per workgroup: as many as leave a few workgroups per CU (the samples then cross L2 once per group)
  int kc = kLcKc;
  while (kc > 1 && (long)((K + kc - 1) / kc) * blocks * chunks < GACQ_LC_MINWG) kc >>= 1;
  const long ngroups = (K + kc - 1) / kc;
  if (L >= (1L << 30)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "long-code search: code length %ld too large", L);
  hipLaunchKernelGGL(longcode_dot_kernel, dim3((unsigned)(ngroups * blocks * chunks)), dim3(kLcBlock), 0, st, (const float2*)ctx->fe_a.p, d_chips,
                     (int)L, (const double*)d_phase, incr, (long)n, blocks, chunks, K, kc, d_partial);
  GACQ_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(longcode_finish_kernel, dim3((unsigned)((K + 127) / 128)), dim3(128), 0, st, (const double2*)d_partial, d_q, K, blocks,
                     chunks);
  GACQ_HIP(ctx, hipGetLastError());
  GACQ_HIP(ctx, hipMemcpyAsync(q_out, d_q, sizeof(double) * K, hipMemcpyDeviceToHost, st));
  GACQ_HIP(ctx, hipStreamSynchronize(st));
  return GACQ_OK;
}

extern "C" int gacq_longcode_search(gacq_ctx* ctx, const float* x_iq, size_t nsamp, double fs, const char* code, int prn,
                                    double carrier_hz, const double* phase0, int K, int blocks, int n, double* q_out) {
  return longcode_run(ctx, x_iq, nullptr, 0.0, nsamp, fs, code, prn, carrier_hz, phase0, K, blocks, n, q_out);
}

extern "C" int gacq_longcode_search_int8(gacq_ctx* ctx, const int8_t* iq_int8, size_t nsamp, double fs, double carrier_offset_hz,
                                         const char* code, int prn, double carrier_hz, const double* phase0, int K, int blocks, int n,
                                         double* q_out) {
  return longcode_run(ctx, nullptr, iq_int8, carrier_offset_hz, nsamp, fs, code, prn, carrier_hz, phase0, K, blocks, n, q_out);
}

extern "C" int gacq_longcode_search_dev(gacq_ctx* ctx, const void* d_x, size_t nsamp, double fs, const char* code, int prn, double carrier_hz,
                                        const double* phase0, int K, int blocks, int n, double* q_out) {
  if (!d_x) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_longcode_search_dev: d_x is NULL");
  return longcode_run(ctx, nullptr, nullptr, 0.0, nsamp, fs, code, prn, carrier_hz, phase0, K, blocks, n, q_out, (const float2*)d_x);
}

extern "C" int gacq_mix_int8_dev(gacq_ctx* ctx, const void* d_iq_int8, size_t nsamp, double fs, double carrier_offset_hz, void* d_out) {
  if (!ctx || !d_iq_int8 || !d_out || nsamp == 0 || !(fs > 0.0) || !std::isfinite(carrier_offset_hz))
    return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_mix_int8_dev: bad argument");
  GACQ_DEVICE(ctx);
  return frontend_mix(ctx, d_iq_int8, (long)nsamp, fs, carrier_offset_hz, (float2*)d_out);
}
