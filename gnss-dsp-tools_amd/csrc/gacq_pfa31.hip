// Engine 3: FFT lengths N = 31 * M (61380 = 31*1980, 30690 = 31*990) -- the 10.23 Mcps family
// (acquire-gps-l5i.py:19-24 and 18 more scripts) and E6 (acquire-galileo-e6b.py:19-24).
//
// rocFFT has no radix-31 butterfly and falls back to Bluestein for these lengths (three transforms of
// twice the size per FFT).  Here the transform is split Cooley-Tukey style into a hand-written 31-point
// outer DFT and an inner length-M transform that rocFFT does natively (M = 4*5*9*11 or 2*5*9*11):
//
//   n = M n1 + n2, k = k1 + 31 k2:
//   X[k1 + 31 k2] = sum_{n2} W_M^{n2 k2} * ( W_N^{n2 k1} * sum_{n1} x[M n1 + n2] W_31^{n1 k1} )
//
//   forward : pfa_outer_forward_kernel (NCO mix + DFT-31 over the stride-M dimension + twiddle)  K1 + outer
//             rocFFT forward, length M, batch 31*rows, rows contiguous                            inner
//             -> spectrum stored as [k1][k2]; the code spectra use the same order, so the
//                element-wise conj-multiply (K2) is unchanged
//   inverse : rocFFT inverse, length M, batch 31*rows
//             pfa_outer_inverse_kernel (twiddle + inverse DFT-31 + |.|/N + sum over blocks +
//             max/argmax/sum)                                                                     outer + K3
//
// The 31-point DFT uses the conjugate symmetry of W_31: with s_n = x[n] + x[31-n], d_n = x[n] - x[31-n]
//   A_k = x[0] + sum_{n=1..15} cos(2 pi n k/31) s_n,   B_k = sum_{n=1..15} sin(2 pi n k/31) d_n
//   forward X[k] = A_k - i B_k, X[31-k] = A_k + i B_k  (inverse: signs swapped)
// = 30 + 15*30 + 45 packed-f32 instructions per 31 points (real constant x complex value = one v_pk_fma_f32
// with the constant broadcast from an SGPR), a quarter of the 31 x 31 complex products.
#include "gacq_common.h"

#include <cmath>

using namespace gacq;

namespace {

constexpr int kR = 31;
typedef float v2 __attribute__((ext_vector_type(2)));

constexpr float kCos31[16] = {1.f, 0.97952994125249448f, 0.9189578116202306f, 0.82076344120727629f, 0.68896691907568663f,
                              0.52896401032696239f, 0.34730525284482028f, 0.1514277775045767f, -0.050649168838712642f,
                              -0.25065253225872042f, -0.44039415155763439f, -0.61210598254766257f, -0.75875812269279086f,
                              -0.87434661614458209f, -0.95413925640004882f, -0.99486932339189504f};
constexpr float kSin31[16] = {0.f, 0.20129852008866006f, 0.39435585511331855f, 0.57126821509479231f, 0.72479278722911988f,
                              0.84864425749475092f, 0.93775213214708042f, 0.98846832432811138f, 0.99871650717105276f,
                              0.96807711886620429f, 0.89780453957074158f, 0.79077573693769887f, 0.65137248272222226f,
                              0.48530196253108104f, 0.29936312297335804f, 0.10116832198743272f};

__device__ __forceinline__ v2 add_i(v2 a, v2 b) {   // a + i b
  v2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2 sub_i(v2 a, v2 b) {   // a - i b
  v2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2 cmul(v2 a, v2 b) {
  v2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
// acc + x * cs.lo / acc + x * cs.hi / acc - x * cs.hi with cs = (cos, sin) in an SGPR pair, broadcast to both halves
__device__ __forceinline__ v2 fma_c(v2 acc, v2 x, v2 cs) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "s"(cs), "v"(acc));
  return r;
}
__device__ __forceinline__ v2 fma_s(v2 acc, v2 x, v2 cs) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "s"(cs), "v"(acc));
  return r;
}
__device__ __forceinline__ v2 fms_s(v2 acc, v2 x, v2 cs) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(x), "s"(cs), "v"(acc));
  return r;
}
__device__ __forceinline__ v2 mul_s(v2 x, v2 cs) {
  v2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "s"(cs));
  return r;
}
__device__ __forceinline__ v2 mul_s_neg(v2 x, v2 cs) {
  v2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "s"(cs));
  return r;
}

// 31-point DFT of x[0..30]; outputs are handed to sink(k, value) as they are produced (k = 0, then pairs k, 31-k)
// so that callers can twiddle/store/reduce them without holding a second 31-element array.
template <bool INV, class Sink> __device__ __forceinline__ void dft31(const v2 (&x)[kR], Sink&& sink) {
  v2 s[16], d[16];
  v2 sum = x[0];
#pragma unroll
  for (int n = 1; n <= 15; n++) {
    s[n] = x[n] + x[kR - n];
    d[n] = x[n] - x[kR - n];
    sum += s[n];
  }
  sink(0, sum);
#pragma unroll
  for (int k = 1; k <= 15; k++) {
    v2 A = x[0], B;
#pragma unroll
    for (int n = 1; n <= 15; n++) {
      const int m = (n * k) % kR;
      const int mm = m <= 15 ? m : kR - m;
      const v2 cs = {kCos31[mm], kSin31[mm]};
      A = fma_c(A, s[n], cs);
      if (n == 1) B = (m <= 15) ? mul_s(d[n], cs) : mul_s_neg(d[n], cs);
      else B = (m <= 15) ? fma_s(B, d[n], cs) : fms_s(B, d[n], cs);
    }
    sink(k, INV ? add_i(A, B) : sub_i(A, B));
    sink(kR - k, INV ? sub_i(A, B) : add_i(A, B));
  }
}

// w^k for k = 0..30 from base-4 digits: w^k = p[k & 3] * q[k >> 2], p[a] = w^a, q[b] = w^(4b); multiplication depth <= 5
struct TwPow {
  v2 p[4], q[8];
  __device__ __forceinline__ void init(v2 w) {
    p[1] = w;
    p[2] = cmul(w, w);
    p[3] = cmul(p[2], w);
    q[1] = cmul(p[2], p[2]);
    q[2] = cmul(q[1], q[1]);
    q[3] = cmul(q[2], q[1]);
    q[4] = cmul(q[2], q[2]);
    q[5] = cmul(q[4], q[1]);
    q[6] = cmul(q[3], q[3]);
    q[7] = cmul(q[4], q[3]);
  }
  // v * w^k, k compile-time after unrolling
  __device__ __forceinline__ v2 apply(v2 v, int k) const {
    const int a = k & 3, b = k >> 2;
    if (a) v = cmul(v, p[a]);
    if (b) v = cmul(v, q[b]);
    return v;
  }
};

// ---- forward outer stage ---------------------------------------------------------------------------
// grid = rows * chunks; thread -> n2.  MIX: multiply by the table NCO (rows = (e,f,d,b)); otherwise plain rows.
template <bool MIX>
__global__ __launch_bounds__(kBlock) void pfa_outer_forward_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                                    float2* __restrict__ A, const double* __restrict__ freq,
                                                                    const float2* __restrict__ nco_tab,
                                                                    const float2* __restrict__ tw, int n, int M, int FD, int B,
                                                                    int chunks) {
  const long blk = blockIdx.x;
  const int chunk = (int)(blk % chunks);
  const long row = blk / chunks;
  const int n2 = chunk * kBlock + threadIdx.x;
  if (n2 >= M) return;
  const float2* src;
  double f = 0.0;
  if (MIX) {
    const int b = (int)(row % B);
    const long r2 = row / B;
    const int fd = (int)(r2 % FD);
    const long e = r2 / FD;
    f = freq[fd];
    src = x + e * epoch_stride + (size_t)b * n;
  } else {
    src = x + row * (long)(kR * M);
  }
  v2 v[kR];
#pragma unroll
  for (int n1 = 0; n1 < kR; n1++) {
    const int i = M * n1 + n2;
    const float2 sf = src[i];
    v2 sv = {sf.x, sf.y};
    if (MIX) {
      // table NCO, index in fp64 exactly as numpy: floor((0 + f*i)*1024) mod 1024   (gnsstools/nco.py:6-9)
      const long k = (long)floor(__dmul_rn(__dmul_rn(f, (double)i), 1024.0)) & (kNcoTableSize - 1);
      const float2 wf = nco_tab[k];
      const v2 wv = {wf.x, wf.y};
      sv = cmul(sv, wv);
    }
    v[n1] = sv;
  }
  TwPow tp;
  {
    const float2 wf = tw[n2];           // W_N^{n2}
    const v2 wv = {wf.x, wf.y};
    tp.init(wv);
  }
  float2* dst = A + row * (long)(kR * M) + n2;
  dft31<false>(v, [&](int k1, v2 val) {
    const v2 o = tp.apply(val, k1);
    dst[(long)k1 * M] = make_float2(o.x, o.y);
  });
}

// ---- inverse outer stage + magnitude + reduce ------------------------------------------------------------
// Z: [group][b][k1][n2] after the inner inverse transforms (unnormalised).  One workgroup handles 256 values of n2
// of one group and emits a partial (peak, idx, sum) record; idx = M n1 + n2.
__global__ __launch_bounds__(kBlock) void pfa_outer_inverse_kernel(const float2* __restrict__ Z, RowRec* __restrict__ partial,
                                                                    const float2* __restrict__ tw, int M, int B, int chunks,
                                                                    float inv_n, float* __restrict__ q_out) {
  __shared__ float s_peak[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  __shared__ double s_sum[kBlock / 64];
  const long blk = blockIdx.x;
  const int chunk = (int)(blk % chunks);
  const long g = blk / chunks;
  const int n2 = chunk * kBlock + threadIdx.x;
  float peak = -1.0f;
  int idx = 0x7fffffff;
  double sum = 0.0;
  if (n2 < M) {
    TwPow tp;
    {
      const float2 wf = tw[n2];
      const v2 wv = {wf.x, -wf.y};      // conj: W_N^{-n2}
      tp.init(wv);
    }
    float q[kR];
#pragma unroll
    for (int k = 0; k < kR; k++) q[k] = 0.f;
    for (int b = 0; b < B; b++) {
      const float2* src = Z + (g * B + b) * (long)(kR * M) + n2;
      v2 v[kR];
#pragma unroll
      for (int k1 = 0; k1 < kR; k1++) {
        const float2 zf = src[(long)k1 * M];
        const v2 zv = {zf.x, zf.y};
        v[k1] = tp.apply(zv, k1);
      }
      dft31<true>(v, [&](int n1, v2 val) {
        q[n1] += __builtin_amdgcn_sqrtf(val.x * val.x + val.y * val.y) * inv_n;      // np.absolute(ifft(..)), 1/N folded in
      });
    }
#pragma unroll
    for (int n1 = 0; n1 < kR; n1++) {          // ascending idx = M n1 + n2: strict '>' keeps the first maximum
      if (q[n1] > peak) { peak = q[n1]; idx = M * n1 + n2; }
      sum += (double)q[n1];
      if (q_out) q_out[M * n1 + n2] = q[n1];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float op = __shfl_down(peak, off);
    const int oi = __shfl_down(idx, off);
    const double os = __shfl_down(sum, off);
    if (op > peak || (op == peak && oi < idx)) { peak = op; idx = oi; }
    sum += os;
  }
  const int t = threadIdx.x;
  if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = idx; s_sum[t >> 6] = sum; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      if (s_peak[w] > peak || (s_peak[w] == peak && s_idx[w] < idx)) { peak = s_peak[w]; idx = s_idx[w]; }
      sum += s_sum[w];
    }
    RowRec r;
    r.peak = peak;
    r.idx = idx;
    r.sum = sum;
    partial[blk] = r;
  }
}

// partial[(g, chunk)] -> rows[g0 + g]
__global__ void pfa_combine_kernel(const RowRec* __restrict__ partial, RowRec* __restrict__ rows, long g0, long ng, int chunks) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  RowRec best = partial[g * chunks];
  for (int c = 1; c < chunks; c++) {
    const RowRec r = partial[g * chunks + c];
    if (r.peak > best.peak || (r.peak == best.peak && r.idx < best.idx)) { best.peak = r.peak; best.idx = r.idx; }
    best.sum += r.sum;
  }
  rows[g0 + g] = best;
}

struct TwTable { int N; int device; float2* p; };
std::vector<TwTable> g_tables;

int base_twiddles(gacq_ctx* ctx, int N, const float2** out) {
  for (const TwTable& t : g_tables) if (t.N == N && t.device == ctx->device) { *out = t.p; return GACQ_OK; }
  const int M = N / kR;
  std::vector<float2> h(M);
  for (int k = 0; k < M; k++) {
    const double a = -2.0 * M_PI * (double)k / (double)N;      // W_N^k, k < M
    h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  TwTable t{N, ctx->device, nullptr};
  GACQ_HIP(ctx, hipMalloc((void**)&t.p, sizeof(float2) * M));
  GACQ_HIP(ctx, hipMemcpy(t.p, h.data(), sizeof(float2) * M, hipMemcpyHostToDevice));
  g_tables.push_back(t);
  *out = t.p;
  return GACQ_OK;
}

bool smooth(int m) {
  for (int p : {2, 3, 5, 7, 11, 13}) while (m % p == 0) m /= p;
  return m == 1;
}

}  // namespace

namespace gacq {

bool pfa_supported(int N) { return N > 0 && N % kR == 0 && smooth(N / kR) && N / kR >= 64; }

int pfa_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, long rows, int n, int N, const double* d_freq, int FD, int B,
                const float2* tab, float2* X, bool mix) {
  if (!pfa_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "radix-31 engine: N=%d not supported", N);
  const float2* tw;
  int rc = base_twiddles(ctx, N, &tw);
  if (rc != GACQ_OK) return rc;
  const int M = N / kR;
  const int chunks = (M + kBlock - 1) / kBlock;
  if (mix)
    hipLaunchKernelGGL(pfa_outer_forward_kernel<true>, dim3((unsigned)(rows * chunks)), dim3(kBlock), 0, ctx->stream, x, nsamp, X,
                       d_freq, tab, tw, n, M, FD, B, chunks);
  else
    hipLaunchKernelGGL(pfa_outer_forward_kernel<false>, dim3((unsigned)(rows * chunks)), dim3(kBlock), 0, ctx->stream, x, nsamp, X,
                       d_freq, tab, tw, n, M, FD, B, chunks);
  GACQ_HIP(ctx, hipGetLastError());
  return fft_exec(ctx, M, rows * kR, false, X);            // inner transforms, rows contiguous
}

int pfa_inverse_reduce(gacq_ctx* ctx, float2* Y, RowRec* rows, long g0, long ng, int B, int N, float* q_out) {
  if (!pfa_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "radix-31 engine: N=%d not supported", N);
  const float2* tw;
  int rc = base_twiddles(ctx, N, &tw);
  if (rc != GACQ_OK) return rc;
  const int M = N / kR;
  const int chunks = (M + kBlock - 1) / kBlock;
  if ((rc = fft_exec(ctx, M, ng * B * kR, true, Y)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->partial, sizeof(RowRec) * (size_t)ng * chunks)) != GACQ_OK) return rc;
  hipLaunchKernelGGL(pfa_outer_inverse_kernel, dim3((unsigned)(ng * chunks)), dim3(kBlock), 0, ctx->stream, Y, (RowRec*)ctx->partial.p,
                     tw, M, B, chunks, 1.0f / (float)N, q_out);
  GACQ_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(pfa_combine_kernel, dim3((unsigned)((ng + 127) / 128)), dim3(128), 0, ctx->stream, (const RowRec*)ctx->partial.p,
                     rows, g0, ng, chunks);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace gacq
