// Engine 5: complex128 verification pipeline (hard part 1: "keep a complex128 build").
//
// The product engines compute in complex64; the reference computes in complex128 (numpy / scipy.fftpack).  This engine is the
// north-star pipeline K1 -> FFT -> K2 -> IFFT -> K3 -> K4 with every value in fp64 on the device (rocFFT double precision, fp64
// NCO table, fp64 magnitudes and metric), so it agrees with the reference to ~1e-12 instead of ~1e-7.  It exists to bisect:
// when an fp32 engine and the oracle disagree on a near-tied peak, engine 5 says which side the rounding fell on.  Slow by
// design (twice the bytes, a quarter of the flops/s); selected only by gacq_set_engine(ctx, 5), never by auto.
#include "gacq_common.h"
#define GACQ_F64_UNPAIR 1      // fused4k_c128_kernel: no AGPR spills (244 VGPRs), see gacq_fft64.h
#include "gacq_fft64.h"

#include <cmath>

using namespace gacq;

namespace {

struct RowRec64 {
  double peak;
  double sum;
  int idx;
  int pad;
};

// K1: y = x * tab64[floor((f*i)*1024) & 1023]   (acquire-gps-l1.py:28,30-31, gnsstools/nco.py:6-10), samples widened to fp64
__global__ __launch_bounds__(kBlock) void mix64_kernel(XSrc x, size_t epoch_stride, double2* __restrict__ y,
                                                        const double* __restrict__ freq, const double2* __restrict__ tab, int n, int span,
                                                        int FD, int B, unsigned cols) {
  const unsigned row = blockIdx.x / cols;           // ((e*FD + fd)*B + b)
  const int b = (int)(row % (unsigned)B);
  const unsigned t = row / (unsigned)B;
  const int fd = (int)(t % (unsigned)FD);
  const long e = t / (unsigned)FD;
  const double f = freq[fd];
  const int i = (int)(blockIdx.x % cols) * kBlock + threadIdx.x;
  if (i >= span) return;
  const double2 s = ld_x(x, e * epoch_stride + (size_t)b * n + i);
  const double2 w = tab[nco_index(f, i)];
  y[row * (long)span + i] = make_double2(s.x * w.x - s.y * w.y, s.x * w.y + s.y * w.x);
}

// K2: Y = C_p * conj(X)   (acquire-gps-l1.py:32)
__global__ __launch_bounds__(kBlock) void conj_mul64_kernel(const double2* __restrict__ X, const double2* __restrict__ C, double2* __restrict__ Y,
                                                             const int* __restrict__ items, const int* __restrict__ fset, long g0, int P, int F,
                                                             int D, int B, int N, unsigned cols) {
  const unsigned ry = blockIdx.x / cols;
  const int b = (int)(ry % (unsigned)B);
  const unsigned g = (unsigned)g0 + ry / (unsigned)B;
  const int d = (int)(g % (unsigned)D);
  const unsigned ep = g / (unsigned)D;
  const int p = (int)(ep % (unsigned)P);
  const long e = ep / (unsigned)P;
  const int i = (int)(blockIdx.x % cols) * kBlock + threadIdx.x;
  if (i >= N) return;
  const double2 xv = X[(((e * F + fset[p]) * D + d) * (long)B + b) * (long)N + i];
  const double2 cv = C[(long)items[p] * N + i];
  Y[ry * (long)N + i] = make_double2(cv.x * xv.x + cv.y * xv.y, cv.y * xv.x - cv.x * xv.y);
}

// K3: q[k] = sum_b |Y_b[k] / N| in fp64 (the reference's ifft carries the 1/N, then np.absolute, then the sum over blocks)
__global__ __launch_bounds__(kBlock) void mag_peak64_kernel(const double2* __restrict__ Y, RowRec64* __restrict__ rows, long g0, int B, int N,
                                                             float* __restrict__ q_out) {
  __shared__ double s_peak[kBlock], s_sum[kBlock];
  __shared__ int s_idx[kBlock];
  const long gl = blockIdx.x;
  const double2* ys = Y + gl * (long)B * N;
  const double inv_n = 1.0 / (double)N;
  double peak = -1.0, sum = 0.0;
  int idx = 0x7fffffff;
  for (int k = threadIdx.x; k < N; k += kBlock) {
    double q = 0.0;
    for (int b = 0; b < B; b++) {
      const double2 v = ys[(long)b * N + k];
      q += hypot(v.x * inv_n, v.y * inv_n);
    }
    if (q_out) q_out[k] = (float)q;
    if (q > peak) { peak = q; idx = k; }
    sum += q;
  }
  s_peak[threadIdx.x] = peak; s_sum[threadIdx.x] = sum; s_idx[threadIdx.x] = idx;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const double op = s_peak[threadIdx.x + off];
      const int oi = s_idx[threadIdx.x + off];
      if (op > s_peak[threadIdx.x] || (op == s_peak[threadIdx.x] && oi < s_idx[threadIdx.x])) { s_peak[threadIdx.x] = op; s_idx[threadIdx.x] = oi; }
      s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { RowRec64 r; r.peak = s_peak[0]; r.sum = s_sum[0]; r.idx = s_idx[0]; r.pad = 0; rows[g0 + gl] = r; }
}

// K4: strict-'>' scan over the Doppler bins in order, running best starts at 0 (acquire-gps-l1.py:25,36-39)
__global__ void best_doppler64_kernel(const RowRec64* __restrict__ rows, gacq_peak* __restrict__ out, long nep, int D, int N, int normalised) {
  const long ep = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ep >= nep) return;
  double best = 0.0;
  int bidx = -1, bd = -1;
  for (int d = 0; d < D; d++) {
    const RowRec64 r = rows[ep * D + d];
    const double m = normalised ? r.peak / (r.sum / (double)N) : r.peak;      // q[idx]/np.mean(q)
    if (m > best) { best = m; bidx = r.idx; bd = d; }
  }
  gacq_peak o;
  o.metric = best; o.idx = bidx; o.d_index = bd;
  out[ep] = o;
}

// ---- N = 4096, one block, one carrier: the whole row in one workgroup, complex128 ---------------------------------------------------
// The structure of lds_fused4k_kernel (gacq_ldsfft.hip) in the reference's arithmetic type: workgroup = (epoch, Doppler bin, chunk of
// pch items); prologue: x window, table-NCO mix, forward transform, conjugate -- the spectrum stays in registers (the transform's
// output convention is the inverse transform's input convention); then per item C_p * conj(X), inverse transform, |.| / N and the
// (max, first argmax, sum) of the row.  Nothing but x, the complex128 code spectra (natural order: lane t reads C[t + 256 j], 16
// bytes per lane, 1 KiB per wave and instruction) and one 24-byte record per row touches HBM; the rocFFT pipeline this replaces
// moves 5 x 64 KB per row.  64 KB of LDS per workgroup (gacq_fft64.h) -> two workgroups per CU, <= 256 VGPRs.
#ifndef GACQ_C128_PRIO
#define GACQ_C128_PRIO true      // rising wave priorities (gacq_fft64.h F64_PRIO): 13.67 -> 12.88 ms per 1024-epoch step, profiles/r05_headline_kernel_wave_priority_sweep.log
#endif
__global__ __launch_bounds__(kBlock, 2) void fused4k_c128_kernel(XSrc x, size_t epoch_stride, const double2* __restrict__ C,
                                                                 const int* __restrict__ items, const double* __restrict__ freq,
                                                                 const double2* __restrict__ tab, const double2* __restrict__ tw,
                                                                 RowRec64* __restrict__ rows, int E, int P, int D, int pch, int nchunk) {
  using namespace gacq::f64;
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  __shared__ double s_peak[kBlock / 64], s_sum[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  const int t = threadIdx.x;
  // placement as in lds_fused4k_kernel: all workgroups of an epoch run on XCD e % 8 (the sample block is fetched into one L2)
  const int xcd = blockIdx.x & 7;
  const unsigned j8 = blockIdx.x >> 3;
  const unsigned per_epoch = (unsigned)D * (unsigned)nchunk;
  const unsigned e8 = (j8 / per_epoch) * 8 + xcd;
  if (e8 >= (unsigned)E) return;
  const long e = e8;
  const int d = (int)((j8 % per_epoch) / (unsigned)nchunk);
  const int p0 = (int)(j8 % (unsigned)nchunk) * pch;
  const int p1 = min(P, p0 + pch);
  const double2 wa2 = tw[t], wb2 = tw[16 * (t & 15)];
  const cd wa = {wa2.x, wa2.y}, wb = {wb2.x, wb2.y};
  // pass-2 twiddle powers of the inverse transform, conj(W_256^c)^k for the 16 lane classes c = t & 15: built once per workgroup,
  // read back from a 3.8 KB LDS table in the item loop (the trick of lds_fused4k_kernel<.., PREA>)
  __shared__ cd s_tw2[15 * 16];
  if (t < 16) {
    cd pw[15];
    make_powers(pw, conj(wb));                            // lanes 0..15: wb = W_256^t
#pragma unroll
    for (int k = 0; k < 15; k++) s_tw2[16 * k + t] = pw[k];
  }
  cd xr[16];
  {
    const double f = freq[d];
    const XSrc src = x.offset(e * epoch_stride);
    cd v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int i = t + 256 * j;
      const double2 sv = ld_x(src, i);
      const double2 w = tab[nco_index(f, i)];                     // gnsstools/nco.py:6-10
      v[j] = cd{sv.x, sv.y} * cd{w.x, w.y};
    }
    fft4096<false>(v, lds64, wa, wb, t);
#pragma unroll
    for (int j = 0; j < 16; j++) xr[j] = conj(v[rev16(j)]);      // np.conj(fft.fft(b))  acquire-gps-l1.py:32
    __syncthreads();
  }
  const double inv_n = 1.0 / (double)kN;
  for (int p = p0; p < p1; p++) {
    const double2* cp = C + (long)items[p] * kN + t;
    if (GACQ_C128_PRIO) asm volatile("s_setprio 0" ::: "memory");
    cd v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const double2 c = cp[256 * j]; v[j] = cd{c.x, c.y}; }
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = v[j] * xr[j];
    fft4096<true, true, GACQ_C128_PRIO>(v, lds64, wa, wb, t, s_tw2 + (t & 15));
    // (max, first argmax, sum) of the wave's 1024 magnitudes: maximum first (per lane, then over the wave on DPP), location second
    // (one compare per register against the wave-uniform maximum, lane masks folded on the scalar unit) -- wave_first_max's scheme
    double m[16], sum = 0.0, lmax = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) {                                // lane t holds lags t + 256 k, ascending in k
      const cd r = v[rev16(k)];
      m[k] = sqrt_pos(r.x * r.x + r.y * r.y) * inv_n;             // np.absolute(ifft(...)); 1/N is a power of two
      sum += m[k];
      lmax = fmax(lmax, m[k]);
    }
    double peak = wave_max_pos_f64(lmax);
    int idx = 0x7fffffff;
#pragma unroll
    for (int k = 15; k >= 0; k--) {                               // descending: the last assignment is the smallest k
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(m[k] == peak);
      if (mk) idx = 256 * k + (t & ~63) + (int)__builtin_ctzll(mk);
    }
    sum = wave_add_f64(sum);
    if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = idx; s_sum[t >> 6] = sum; }
    __syncthreads();      // also orders this item's exchange-2 reads before the next item's exchange-1 writes
    if (t == 0) {
      for (int w = 1; w < kBlock / 64; w++) {
        if (s_peak[w] > peak || (s_peak[w] == peak && s_idx[w] < idx)) { peak = s_peak[w]; idx = s_idx[w]; }
        sum += s_sum[w];
      }
      RowRec64 r;
      r.peak = peak; r.sum = sum; r.idx = idx; r.pad = 0;
      rows[(e * P + p) * (long)D + d] = r;
    }
    // no barrier here: the scratch is rewritten only after the next item's transform, whose first exchange barrier thread 0 reaches
    // after it has finished reading it
  }
}

// ---- N = R x 4096 (R = 4: BeiDou B1I / B2I, GLONASS L1 / L2; R = 16: Galileo E1B / E1C) in complex128: the split of engine 4 ----------
// n = 4096 n1 + n2, k = k1 + R k2 (the split of gacq_split.hip and of the re-evaluation's tie_recheck_split_kernel, whose arithmetic this
// follows step by step), with every 4096-point transform on the LDS-resident complex128 transform of gacq_fft64.h:
//   forward   (one workgroup per (forward row, k1), shared by all items):  A[k1][n2] = W_N^{n2 k1} sum_n1 x[4096 n1 + n2] nco[.] W_R^{n1 k1},
//             X[k1][k2] = FFT_4096(A[k1])                                         -> Xs, 16 N bytes per forward row
//   writer    (per correlation row, block, k1):  T = FFT_4096(conj(C_p[k1][.]) X[k1][.]);  Z'[k1][n2] = W_N^{n2 k1} T[n2]   (|ifft(Y)| =
//             |fft(conj Y)| / N: both transforms of the chain are forward ones)     -> Z', 16 N bytes per row and block
//   reader    (per correlation row):  DFT_R over k1 for every n2, |.| / N, sum over the blocks, (max, first argmax, sum)
// One Z' round trip of 32 N bytes per row and block instead of the five stage boundaries of the rocFFT pipeline (5 x 32 N), no rocFFT
// plan.  The writer keeps the forward-spectrum row in registers over the items of its chunk (one carrier for all of them) and streams
// the code-spectrum rows; Z' rows of a pass are ordered (epoch, Doppler bin, item), the records go out in the (epoch, item, Doppler bin)
// order of the Doppler scan.
__device__ __forceinline__ f64::cd ldc(const double2* p) { const double2 v = *p; return f64::cd{v.x, v.y}; }
__device__ __forceinline__ void stc(double2* p, f64::cd v) { *p = make_double2(v.x, v.y); }

// C[p][k] (natural order) -> Cs[p][k1][k2] = C[p][k1 + R k2], k2 < M
__global__ __launch_bounds__(kBlock) void c128_split_spectra_kernel(const double2* __restrict__ C, double2* __restrict__ Cs, int R, int M) {
  const long p = blockIdx.x / (unsigned)R;
  const int k1 = (int)(blockIdx.x % (unsigned)R);
  const double2* src = C + p * (long)R * M + k1;
  double2* dst = Cs + (p * R + k1) * (long)M;
  for (int k2 = threadIdx.x; k2 < M; k2 += kBlock) dst[k2] = src[(long)R * k2];
}

// y: the carrier-wiped rows of mix64_kernel (one N-sample window per forward row) -- the table-NCO index arithmetic runs once per sample
// in that streaming kernel instead of once per (sample, k1) here, where two workgroups per CU cannot hide it (GLONASS, a carrier per
// item: forward stage 6.7 -> see profiles/r06_complex128_split_engine.log)
template <int R>
__global__ __launch_bounds__(kBlock, 2) void c128_split_forward_kernel(const double2* __restrict__ y, double2* __restrict__ Xs,
                                                                       const double2* __restrict__ WN, unsigned nrows) {
  using namespace gacq::f64;
  constexpr int M = kN, N = R * kN;
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  const int t = threadIdx.x;
  // the R workgroups of a forward row read the same N mixed samples: placed on ONE XCD (workgroup b runs on XCD b % 8), consecutive in
  // its dispatch order, so that the row crosses the fabric once and the other R - 1 find it in that L2 (GLONASS, a carrier per item:
  // 30 000 rows of 256 KB -- read by k1 = blockIdx % R from four XCDs they came out of HBM four times)
  const unsigned xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3;
  const int k1 = (int)(jx % (unsigned)R);
  const unsigned row = (jx / (unsigned)R) * 8u + xcd;   // ((e*FD + fd)*B + b)
  if (row >= nrows) return;
  const double2* src = y + (long)row * N + t;
  const cd wa = ldc(WN + t * R), wb = ldc(WN + 16 * (t & 15) * R);      // W_4096^t, W_256^(t & 15)
  cd v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) v[j] = cd{0.0, 0.0};
  for (int n1 = 0; n1 < R; n1++) {
    cd sv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) sv[j] = ldc(src + M * n1 + 256 * j);
    const cd wr = ldc(WN + ((n1 * k1) % R) * M);                       // W_R^{n1 k1}
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = v[j] + sv[j] * wr;
  }
  {
    // W_N^{n2 k1} = W_N^{t k1} x W_N^{256 j k1}: one gather per thread and 16 uniform (scalar) loads instead of 16 gathers
    const cd twb = ldc(WN + t * k1);
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = v[j] * (twb * ldc(WN + 256 * j * k1));
  }
  fft4096<false>(v, lds64, wa, wb, t);
  double2* dst = Xs + ((long)row * R + k1) * M + t;
#pragma unroll
  for (int j = 0; j < 16; j++) stc(dst + 256 * j, v[rev16(j)]);                  // X[k1 + R k2], k2 = t + 256 j
}

// grid: (units of the pass) x B x R x nchunk; unit = (epoch, Doppler bin)
template <int R>
__global__ __launch_bounds__(kBlock, 2) void c128_split_corr_kernel(const double2* __restrict__ Xs, const double2* __restrict__ Cs,
                                                                    double2* __restrict__ Z, const int* __restrict__ items,
                                                                    const int* __restrict__ fset, const double2* __restrict__ WN, long u0,
                                                                    int P, int F, int D, int B, int pch, int nchunk) {
  using namespace gacq::f64;
  constexpr int M = kN;
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  const int t = threadIdx.x;
  unsigned blk = blockIdx.x;
  const int k1 = (int)(blk % (unsigned)R);
  blk /= (unsigned)R;
  const int p0 = (int)(blk % (unsigned)nchunk) * pch;
  blk /= (unsigned)nchunk;
  const int b = (int)(blk % (unsigned)B);
  const long ul = blk / (unsigned)B;                    // unit within the pass
  const long u = u0 + ul;
  const long e = u / D;
  const int d = (int)(u % D);
  const int p1 = min(P, p0 + pch);
  const cd wa = ldc(WN + t * R), wb = ldc(WN + 16 * (t & 15) * R);
  // pass-2 twiddle powers W_256^{c k} of the 16 lane classes c = t & 15 from a 3.8 KB LDS table (as in fused4k_c128_kernel): rebuilt
  // per transform they cost the registers the resident forward-spectrum row needs
  __shared__ cd s_tw2[15 * 16];
  if (t < 16) {
    cd pw[15];
    make_powers(pw, wb);
#pragma unroll
    for (int k = 0; k < 15; k++) s_tw2[16 * k + t] = pw[k];
  }
  __syncthreads();
  const cd twb = ldc(WN + t * k1);
  cd xs[16];
  int have = -1;
  for (int p = p0; p < p1; p++) {
    if (fset[p] != have) {
      have = fset[p];
      const double2* xr = Xs + (((((long)e * F + have) * D + d) * B + b) * R + k1) * (long)M + t;
#pragma unroll
      for (int j = 0; j < 16; j++) xs[j] = ldc(xr + 256 * j);
    }
    const double2* cr = Cs + ((long)items[p] * R + k1) * M + t;
    cd y[16];
#pragma unroll
    for (int j = 0; j < 16; j++) y[j] = ldc(cr + 256 * j);
#pragma unroll
    for (int j = 0; j < 16; j++) y[j] = conj(y[j]) * xs[j];                      // conj(C_p * conj(X))   acquire-beidou-b1i.py:32
    if (p > p0) __syncthreads();                                                 // the previous transform's exchange-2 reads are complete
    fft4096<false, true>(y, lds64, wa, wb, t, s_tw2 + (t & 15));
    double2* zr = Z + ((((ul * P + p) * B + b) * R + k1) * (long)M) + t;
    // W_N^{n2 k1}, n2 = t + 256 j, as W_N^{t k1} (one value per thread, resident) x W_N^{256 j k1} (uniform over the workgroup: scalar
    // loads): a second complex product per output instead of 16 gathers behind the transform (two workgroups per CU hide little of a
    // dependent round trip); all 16 resident, their 64 registers push the transform into scratch
#pragma unroll
    for (int j = 0; j < 16; j++) stc(zr + 256 * j, y[rev16(j)] * (twb * ldc(WN + 256 * j * k1)));
  }
}

// one workgroup per correlation row of the pass (Z' order: (unit, item)); lags M n1 + n2
template <int R>
__global__ __launch_bounds__(kBlock, 2) void c128_split_reader_kernel(const double2* __restrict__ Z, RowRec64* __restrict__ rows, long u0, int P,
                                                                      int D, int B, float* __restrict__ q_out) {
  using namespace gacq::f64;
  constexpr int M = kN, N = R * kN;
  __shared__ double s_peak[kBlock / 64], s_sum[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  const int t = threadIdx.x;
  const long gl = blockIdx.x;                           // ul * P + p
  const long ul = gl / P;
  const int p = (int)(gl % P);
  const long u = u0 + ul;
  const long e = u / D;
  const int d = (int)(u % D);
  const double inv_n = 1.0 / (double)N;
  const double2* z0 = Z + gl * (long)B * R * M + t;
  double peak = -1.0, sum = 0.0;
  int idx = 0x7fffffff;
  for (int j = 0; j < 16; j++) {
    const int n2 = t + 256 * j;
    double acc[R];
#pragma unroll
    for (int n1 = 0; n1 < R; n1++) acc[n1] = 0.0;
    for (int b = 0; b < B; b++) {
      cd z[R];
#pragma unroll
      for (int kk = 0; kk < R; kk++) z[kk] = ldc(z0 + ((long)b * R + kk) * M + 256 * j);
      if constexpr (R == 4) {
        dft4<false>(z[0], z[1], z[2], z[3]);
#pragma unroll
        for (int n1 = 0; n1 < 4; n1++) acc[n1] += sqrt_pos(z[n1].x * z[n1].x + z[n1].y * z[n1].y) * inv_n;
      } else {
        dft16<false>(z);
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) { const cd r = z[rev16(n1)]; acc[n1] += sqrt_pos(r.x * r.x + r.y * r.y) * inv_n; }
      }
    }
#pragma unroll
    for (int n1 = 0; n1 < R; n1++) {
      const int lag = M * n1 + n2;
      if (q_out) q_out[lag] = (float)acc[n1];
      if (acc[n1] > peak || (acc[n1] == peak && lag < idx)) { peak = acc[n1]; idx = lag; }      // np.argmax: the first maximum
      sum += acc[n1];
    }
  }
  // (max, first argmax, sum) over the workgroup
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double op = __shfl_down(peak, off), os = __shfl_down(sum, off);
    const int oi = __shfl_down(idx, off);
    if (op > peak || (op == peak && oi < idx)) { peak = op; idx = oi; }
    sum += os;
  }
  if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = idx; s_sum[t >> 6] = sum; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      if (s_peak[w] > peak || (s_peak[w] == peak && s_idx[w] < idx)) { peak = s_peak[w]; idx = s_idx[w]; }
      sum += s_sum[w];
    }
    RowRec64 r;
    r.peak = peak; r.sum = sum; r.idx = idx; r.pad = 0;
    rows[(e * P + p) * (long)D + d] = r;
  }
}

// ---- N = 31 x M (61380 = 31 x 1980, 30690 = 31 x 990: every 10.23 Mcps script, E6, Xona X5) in complex128 ---------------------------
// rocFFT has no radix-31 butterfly and runs these lengths through Bluestein (three transforms of twice the size per FFT): the pipeline
// above manages 1.4e10 cells/s at BASELINE config 4.  The Cooley-Tukey split of gacq_split.hip in fp64 keeps the prime out of rocFFT:
//   n = M n1 + n2, k = k1 + 31 k2:  X[k1 + 31 k2] = sum_n2 W_M^{n2 k2} (W_N^{n2 k1} sum_n1 x[M n1 + n2] W_31^{n1 k1})
//   forward:  c128_r31_forward_kernel (DFT-31 over n1 + twiddle, thread = n2)  ->  rocFFT double, length M (= 4 x 5 x 9 x 11: native radices),
//             batch 31 x rows, in place  ->  X in [k1][k2] order (the code spectra are reordered once to match)
//   inverse:  conj_mul64_kernel  ->  rocFFT inverse length M  ->  c128_r31_reader_kernel (twiddle, inverse DFT-31, |.| / N, sum over the
//             blocks, first-argmax reduce; one workgroup per correlation row)
// The DFT-31 uses the conjugate symmetry of W_31 like dft_prime of gacq_cplx.h: s_n = x[n] + x[31-n], d_n = x[n] - x[31-n],
// A_k = x[0] + sum cos(2 pi nk/31) s_n, B_k = sum sin(2 pi nk/31) d_n, X[k], X[31-k] = A_k -/+ i B_k (inverse: signs swapped).
__constant__ double kCos31[16] = {1.0, 0.97952994125249448, 0.9189578116202306, 0.82076344120727629, 0.68896691907568663, 0.52896401032696239,
                                  0.34730525284482028, 0.1514277775045767, -0.050649168838712642, -0.25065253225872042, -0.44039415155763439,
                                  -0.61210598254766257, -0.75875812269279086, -0.87434661614458209, -0.95413925640004882, -0.99486932339189504};
__constant__ double kSin31[16] = {0.0, 0.20129852008866006, 0.39435585511331855, 0.57126821509479231, 0.72479278722911988, 0.84864425749475092,
                                  0.93775213214708042, 0.98846832432811138, 0.99871650717105276, 0.96807711886620429, 0.89780453957074158,
                                  0.79077573693769887, 0.65137248272222226, 0.48530196253108104, 0.29936312297335804, 0.10116832198743272};

// out(k, X[k]) for k = 0..30 from x[0] and the folded pairs sn[n] = x[n] + x[31-n], dn[n] = x[n] - x[31-n], n = 1..15
template <bool INV, class Sink> __device__ __forceinline__ void dft31_core_f64(f64::cd x0, const f64::cd (&sn)[16], const f64::cd (&dn)[16], Sink&& out) {
  using f64::cd;
  cd sum = x0;
#pragma unroll
  for (int n = 1; n <= 15; n++) sum = sum + sn[n];
  out(0, sum);
#pragma unroll
  for (int k = 1; k <= 15; k++) {
    cd A = x0, B = cd{0.0, 0.0};
#pragma unroll
    for (int n = 1; n <= 15; n++) {
      const int m = (n * k) % 31;
      const int mm = m <= 15 ? m : 31 - m;
      const double c = kCos31[mm], sg = m <= 15 ? kSin31[mm] : -kSin31[mm];
      A.x = __builtin_fma(c, sn[n].x, A.x);
      A.y = __builtin_fma(c, sn[n].y, A.y);
      B.x = __builtin_fma(sg, dn[n].x, B.x);
      B.y = __builtin_fma(sg, dn[n].y, B.y);
    }
    // forward: X[k] = A - i B, X[31-k] = A + i B
    out(k, INV ? f64::add_i(A, B) : f64::sub_i(A, B));
    out(31 - k, INV ? f64::sub_i(A, B) : f64::add_i(A, B));
  }
}
template <bool INV, class Sink> __device__ __forceinline__ void dft31_f64(f64::cd (&x)[31], Sink&& out) {
  f64::cd sn[16], dn[16];
#pragma unroll
  for (int n = 1; n <= 15; n++) {
    sn[n] = x[n] + x[31 - n];
    dn[n] = x[n] - x[31 - n];
  }
  dft31_core_f64<INV>(x[0], sn, dn, out);
}

// grid: rows x chunks; thread = n2.  y: the carrier-wiped rows of mix64_kernel (N samples per forward row)
__global__ __launch_bounds__(kBlock) void c128_r31_forward_kernel(const double2* __restrict__ y, double2* __restrict__ A, const double2* __restrict__ WN,
                                                                   int M, int chunks) {
  using f64::cd;
  const unsigned row = blockIdx.x / (unsigned)chunks;
  const int n2 = (int)(blockIdx.x % (unsigned)chunks) * kBlock + threadIdx.x;
  if (n2 >= M) return;
  const double2* src = y + (long)row * 31 * M + n2;
  cd v[31];
#pragma unroll
  for (int n1 = 0; n1 < 31; n1++) v[n1] = ldc(src + (long)n1 * M);
  double2* dst = A + (long)row * 31 * M + n2;
  // W_N^{n2 k1}: the outputs arrive as k1 = 0, then the pairs (k, 31 - k): w^k by a chain of products from w = W_N^{n2}, and
  // w^{31-k} = W_M^{n2} conj(w^k) -- two coalesced table reads per column instead of 30 gathers with a lane stride of 16 k1 bytes
  // (64 cache lines per wave and load)
  const cd w1 = ldc(WN + n2), w31 = ldc(WN + 31 * n2);
  cd wk = cd{1.0, 0.0};
  dft31_f64<false>(v, [&](int k1, cd val) {
    if (k1 == 0) { stc(dst, val); return; }
    if (k1 <= 15) { wk = wk * w1; stc(dst + (long)k1 * M, val * wk); }
    else stc(dst + (long)k1 * M, val * (w31 * f64::conj(wk)));
  });
}

// one workgroup per correlation row g (Z: [g][b][k1][n2] after the inner inverse transforms, unnormalised); lags M n1 + n2.
// The magnitudes are summed over the blocks in the thread's own 31 LDS slots (in registers, or reduced on the fly for B = 1, the unrolled
// transform spills 0.3-0.9 KB per lane).  The inputs are loaded and folded pair by pair (x[n], x[31-n]).
__global__ __launch_bounds__(kBlock, 2) void c128_r31_reader_kernel(const double2* __restrict__ Z, RowRec64* __restrict__ rows, const double2* __restrict__ WN,
                                                                     long g0, int M, int B, float* __restrict__ q_out) {
  using f64::cd;
  __shared__ double s_peak[kBlock / 64], s_sum[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  __shared__ double s_acc[31 * kBlock];
  const int t = threadIdx.x;
  const long gl = blockIdx.x;
  const long N = 31L * M;
  const double inv_n = 1.0 / (double)N;
  double peak = -1.0, sum = 0.0;
  int idx = 0x7fffffff;
  auto take = [&](double m, int lag) {
    if (q_out) q_out[lag] = (float)m;
    if (m > peak || (m == peak && lag < idx)) { peak = m; idx = lag; }      // np.argmax: the first maximum (lags arrive in any order)
    sum += m;
  };
  for (int n2 = t; n2 < M; n2 += kBlock) {
#pragma unroll
    for (int n1 = 0; n1 < 31; n1++) s_acc[n1 * kBlock + t] = 0.0;
    const cd w1c = f64::conj(ldc(WN + n2)), w31c = f64::conj(ldc(WN + 31 * n2));
    for (int b = 0; b < B; b++) {
      const double2* src = Z + ((gl * B + b) * 31L) * M + n2;
      const cd x0 = ldc(src);
      cd sn[16], dn[16];
      // conj(W_N^{n2 k1}): conj(w)^n by a chain of products, conj(w^{31-n}) = conj(W_M^{n2}) w^n (see c128_r31_forward_kernel).  Ten
      // loads at a time: hoisted all together -- the compiler's choice -- the 60 loads of a column take 240 registers
      cd wn = cd{1.0, 0.0};
#pragma unroll
      for (int n = 1; n <= 15; n++) {
        wn = wn * w1c;
        const cd a = ldc(src + (long)n * M) * wn;
        const cd c = ldc(src + (long)(31 - n) * M) * (w31c * f64::conj(wn));
        sn[n] = a + c;
        dn[n] = a - c;
        if (n % 5 == 0) asm volatile("" : "+v"(sn[n].x), "+v"(dn[n].x) :: "memory");
      }
      dft31_core_f64<true>(x0, sn, dn, [&](int n1, cd val) {
        s_acc[n1 * kBlock + t] += f64::sqrt_pos(val.x * val.x + val.y * val.y) * inv_n;
      });
    }
#pragma unroll
    for (int n1 = 0; n1 < 31; n1++) take(s_acc[n1 * kBlock + t], M * n1 + n2);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double op = __shfl_down(peak, off), os = __shfl_down(sum, off);
    const int oi = __shfl_down(idx, off);
    if (op > peak || (op == peak && oi < idx)) { peak = op; idx = oi; }
    sum += os;
  }
  if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = idx; s_sum[t >> 6] = sum; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      if (s_peak[w] > peak || (s_peak[w] == peak && s_idx[w] < idx)) { peak = s_peak[w]; idx = s_idx[w]; }
      sum += s_sum[w];
    }
    RowRec64 r;
    r.peak = peak; r.sum = sum; r.idx = idx; r.pad = 0;
    rows[g0 + gl] = r;
  }
}

// ---- N = 4096 with several blocks or carriers (GPS L1 / Xona at --time > 1: the reference CLI's own default is --time 80) in complex128 ----
// The two-kernel form of lds_forward_kernel + lds_correlate_kernel on the complex128 transform: forward spectra once per (epoch, carrier,
// Doppler bin, block) -- conj(FFT(x nco)), natural order, 64 KB per row --, then one workgroup per (epoch, Doppler bin, item) that holds the
// item's code spectrum in registers, walks the B rows (C_p conj(X), inverse transform, |.| / N summed over the blocks in registers) and
// reduces.  Nothing but X crosses HBM twice; the rocFFT pipeline it replaces moves five stage boundaries of 32 N per row and block.
__global__ __launch_bounds__(kBlock, 2) void c128_4k_forward_kernel(XSrc x, size_t epoch_stride, double2* __restrict__ X, const double* __restrict__ freq,
                                                                    const double2* __restrict__ tab, const double2* __restrict__ tw, int n, int FD, int B) {
  using namespace gacq::f64;
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  const int t = threadIdx.x;
  const unsigned row = blockIdx.x;          // ((e*FD + fd)*B + b)
  const int b = (int)(row % (unsigned)B);
  const unsigned r2 = row / (unsigned)B;
  const int fd = (int)(r2 % (unsigned)FD);
  const long e = r2 / (unsigned)FD;
  const double f = freq[fd];
  const XSrc src = x.offset(e * epoch_stride + (size_t)b * n);
  const cd wa = ldc(tw + t), wb = ldc(tw + 16 * (t & 15));
  cd v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int i = t + 256 * j;
    const double2 sv = ld_x(src, i);
    const double2 w = tab[nco_index(f, i)];                     // gnsstools/nco.py:6-10
    v[j] = cd{sv.x, sv.y} * cd{w.x, w.y};
  }
  fft4096<false>(v, lds64, wa, wb, t);
  double2* dst = X + (long)row * kN + t;
#pragma unroll
  for (int j = 0; j < 16; j++) stc(dst + 256 * j, conj(v[rev16(j)]));      // np.conj(fft.fft(b))  acquire-gps-l1.py:32
}

// grid: 8 x ceil(units / 8) x P; the P workgroups of an (epoch, Doppler bin) unit run on one XCD and share its B rows in that L2
__global__ __launch_bounds__(kBlock, 2) void c128_4k_correlate_kernel(const double2* __restrict__ X, const double2* __restrict__ C,
                                                                      const int* __restrict__ items, const int* __restrict__ fset,
                                                                      const double2* __restrict__ tw, RowRec64* __restrict__ rows, int E, int P, int F,
                                                                      int D, int B) {
  using namespace gacq::f64;
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  __shared__ double s_peak[kBlock / 64], s_sum[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  __shared__ cd s_tw2[15 * 16];
  const int t = threadIdx.x;
  const unsigned xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3;
  const int p = (int)(jx % (unsigned)P);
  const unsigned u = (jx / (unsigned)P) * 8u + xcd;
  if (u >= (unsigned)E * (unsigned)D) return;
  const long e = u / (unsigned)D;
  const int d = (int)(u % (unsigned)D);
  const cd wa = ldc(tw + t), wb = ldc(tw + 16 * (t & 15));
  if (t < 16) {
    cd pw[15];
    make_powers(pw, conj(wb));                            // pass-2 powers of the inverse transform, as in fused4k_c128_kernel
#pragma unroll
    for (int k = 0; k < 15; k++) s_tw2[16 * k + t] = pw[k];
  }
  cd c[16];
  {
    const double2* cp = C + (long)items[p] * kN + t;
#pragma unroll
    for (int j = 0; j < 16; j++) c[j] = ldc(cp + 256 * j);
  }
  const double2* xs = X + ((((long)e * F + fset[p]) * D + d) * (long)B) * kN + t;
  const double inv_n = 1.0 / (double)kN;
  double q[16];
#pragma unroll
  for (int k = 0; k < 16; k++) q[k] = 0.0;
  __syncthreads();
  for (int b = 0; b < B; b++) {
    cd v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = ldc(xs + (long)b * kN + 256 * j);
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = c[j] * v[j];                             // C_p * np.conj(fft.fft(b))
    if (b > 0) __syncthreads();                                                  // the previous transform's exchange-2 reads are complete
    fft4096<true, true, true>(v, lds64, wa, wb, t, s_tw2 + (t & 15));
    asm volatile("s_setprio 0" ::: "memory");
#pragma unroll
    for (int k = 0; k < 16; k++) {                                               // lane t holds lags t + 256 k
      const cd r = v[rev16(k)];
      q[k] += sqrt_pos(r.x * r.x + r.y * r.y) * inv_n;                           // np.absolute(ifft(...)), summed over the blocks
    }
  }
  double sum = 0.0, lmax = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) { sum += q[k]; lmax = fmax(lmax, q[k]); }
  double peak = wave_max_pos_f64(lmax);
  int idx = 0x7fffffff;
#pragma unroll
  for (int k = 15; k >= 0; k--) {                                                // descending: the last assignment is the smallest k
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(q[k] == peak);
    if (mk) idx = 256 * k + (t & ~63) + (int)__builtin_ctzll(mk);
  }
  sum = wave_add_f64(sum);
  if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = idx; s_sum[t >> 6] = sum; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      if (s_peak[w] > peak || (s_peak[w] == peak && s_idx[w] < idx)) { peak = s_peak[w]; idx = s_idx[w]; }
      sum += s_sum[w];
    }
    RowRec64 r;
    r.peak = peak; r.sum = sum; r.idx = idx; r.pad = 0;
    rows[(e * P + p) * (long)D + d] = r;
  }
}

// W_4096^k in fp64 (sincospi on the exactly reduced argument, device-side)
__global__ void twiddle4096_64_kernel(double2* __restrict__ w) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 4096) return;
  double s, c;
  sincospi(-2.0 * (double)k / 4096.0, &s, &c);
  w[k] = make_double2(c, s);
}

int nco_table64(gacq_ctx* ctx, const double2** out) {
  auto it = ctx->tables.find("nco64");
  if (it != ctx->tables.end()) { *out = (const double2*)it->second.p; return GACQ_OK; }
  std::vector<double2> h(kNcoTableSize);
  for (int k = 0; k < kNcoTableSize; k++) {
    const double a = 2.0 * M_PI * (double)k * (1.0 / kNcoTableSize);       // np.exp(2*pi*1j*np.arange(NT)*(1.0/NT))  nco.py:4
    h[k] = make_double2(std::cos(a), std::sin(a));
  }
  const void* p = nullptr;
  const int rc = table_cache(ctx, "nco64", h.data(), sizeof(double2) * kNcoTableSize, &p);
  *out = (const double2*)p;
  return rc;
}

int twiddle64_4096(gacq_ctx* ctx, const double2** out) {
  auto it = ctx->tables.find("W64_4096");
  if (it == ctx->tables.end()) {
    DevBuf b;
    GACQ_HIP(ctx, hipMalloc(&b.p, sizeof(double2) * 4096));
    b.cap = sizeof(double2) * 4096;
    hipLaunchKernelGGL(twiddle4096_64_kernel, dim3(16), dim3(256), 0, ctx->stream, (double2*)b.p);
    GACQ_HIP(ctx, hipGetLastError());
    it = ctx->tables.emplace("W64_4096", b).first;
  }
  *out = (const double2*)it->second.p;
  return GACQ_OK;
}

}  // namespace

namespace gacq {

int verify_spectra(gacq_sig* s) {
  if (s->spectra64) return GACQ_OK;
  gacq_ctx* ctx = s->ctx;
  const size_t count = (size_t)s->nprn * s->N;
  std::vector<double2> host(count, make_double2(0.0, 0.0));
  for (int p = 0; p < s->nprn; p++)
    for (int i = 0; i < s->desc.n; i++) host[(size_t)p * s->N + i].x = (double)s->replica[(size_t)p * s->desc.n + i];
  // built in a local buffer and published only once the transform has run: a failure (the transform's scratch, a rocFFT plan) must not
  // leave untransformed replicas behind for the next call to take for code spectra
  double2* buf = nullptr;
  GACQ_HIP(ctx, hipMalloc((void**)&buf, sizeof(double2) * count));
  int rc = GACQ_OK;
  if (hipMemcpyAsync(buf, host.data(), sizeof(double2) * count, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
    rc = set_error(ctx, GACQ_ERR_HIP, "complex128 code spectra: upload of the replicas failed");
  // c = fft.fft(c)   acquire-gps-l1.py:24 -- with the library's own complex128 transform where it factors the length (every length of the
  // reference's scripts), so that the default path never pays for a rocFFT plan; rocFFT's double-precision transform otherwise
  if (rc == GACQ_OK) rc = tie_code_spectra64(ctx, buf, s->nprn, s->N);
  if (rc == GACQ_ERR_UNSUPPORTED) rc = fft_exec(ctx, s->N, s->nprn, false, buf, true);
  if (rc == GACQ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "complex128 code spectrum transform failed");
  if (rc != GACQ_OK) {
    (void)hipStreamSynchronize(ctx->stream);      // `host` is still being read by the copy
    (void)hipFree(buf);
    return rc;
  }
  s->spectra64 = buf;
  return GACQ_OK;
}

// the split form of engine 5 (N = 4 x 4096 or 16 x 4096): see c128_split_*_kernel
template <int R>
static int split64_search(gacq_sig* sig, XSrc d_x, size_t nsamp, int nepoch, int P, int F, int D, int B, gacq_peak* d_out, float* d_qrow,
                          const double2* tab) {
  gacq_ctx* ctx = sig->ctx;
  hipStream_t st = ctx->stream;
  const int n = sig->desc.n, N = sig->N;
  int rc;
  const double2* WN;
  if ((rc = twiddles64(ctx, N, &WN)) != GACQ_OK) return rc;
  if (!sig->spectra64_split) {
    double2* buf = nullptr;
    GACQ_HIP(ctx, hipMalloc((void**)&buf, sizeof(double2) * (size_t)sig->nprn * N));
    hipLaunchKernelGGL(c128_split_spectra_kernel, dim3((unsigned)(sig->nprn * R)), dim3(kBlock), 0, st, (const double2*)sig->spectra64, buf, R, f64::kN);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      (void)hipFree(buf);
      return set_error(ctx, GACQ_ERR_HIP, "complex128 split engine: code-spectrum reorder failed");
    }
    sig->spectra64_split = buf;
  }
  GACQ_HIP(ctx, hipFuncSetAttribute((const void*)c128_split_forward_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, f64::kLdsBytes));
  GACQ_HIP(ctx, hipFuncSetAttribute((const void*)c128_split_corr_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, f64::kLdsBytes));
  const size_t x_epoch_bytes = sizeof(double2) * (size_t)F * D * B * N;
  const int Ec = (int)std::max<size_t>(1, std::min<size_t>((size_t)nepoch, ws_budget(ctx) / std::max<size_t>(1, x_epoch_bytes)));
  if ((rc = ensure(ctx, ctx->X, x_epoch_bytes * Ec)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->rows, sizeof(RowRec64) * (size_t)Ec * P * D)) != GACQ_OK) return rc;
  // items per writer workgroup: the forward-spectrum row stays in registers over them (one carrier); with a carrier per item (GLONASS)
  // every item has its own row and nothing is shared
  const int pch = (F == 1) ? std::min(P, 8) : 1;
  const int nchunk = (P + pch - 1) / pch;
  for (int e0 = 0; e0 < nepoch; e0 += Ec) {
    const int ne = std::min(Ec, nepoch - e0);
    double2* Xs = (double2*)ctx->X.p;
    RowRec64* rows = (RowRec64*)ctx->rows.p;
    const long rows_x = (long)ne * F * D * B;
    const unsigned cols = (unsigned)(N / kBlock);
    if (rows_x * cols >= (1L << 31)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "complex128 split engine: too many forward rows in one pass (lower the workspace limit)");
    // carrier wipe-off into the correlation workspace (free until the first Z' pass), then the split forward transform out of it
    if ((rc = ensure(ctx, ctx->Y, x_epoch_bytes * ne)) != GACQ_OK) return rc;
    stage_begin(ctx, 0);
    hipLaunchKernelGGL(mix64_kernel, dim3((unsigned)(rows_x * cols)), dim3(kBlock), 0, st, d_x.offset((size_t)e0 * nsamp), nsamp, (double2*)ctx->Y.p,
                       (const double*)ctx->freq.p, tab, n, N, F * D, B, cols);
    GACQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(c128_split_forward_kernel<R>, dim3((unsigned)(((rows_x + 7) / 8) * 8 * R)), dim3(kBlock), f64::kLdsBytes, st, (const double2*)ctx->Y.p, Xs, WN,
                       (unsigned)rows_x);
    stage_end(ctx);
    GACQ_HIP(ctx, hipGetLastError());
    // Z' passes of whole (epoch, Doppler bin) units, P items each: at most 4 GiB a pass, as in the fp32 split engines
    const long units = (long)ne * D;
    const size_t unit_bytes = sizeof(double2) * (size_t)P * B * N;
    const size_t pass_cap = std::min(ws_budget(ctx), std::max<size_t>((size_t)4 << 30, unit_bytes));
    long uc = (long)std::max<size_t>(1, std::min<size_t>((size_t)units, pass_cap / unit_bytes));
    uc = std::min<long>(uc, std::max<long>(1, ((1L << 31) - 1) / ((long)B * R * nchunk)));
    if (d_qrow) uc = 1;
    if ((rc = ensure(ctx, ctx->Y, unit_bytes * uc)) != GACQ_OK) return rc;
    double2* Z = (double2*)ctx->Y.p;
    for (long u0 = 0; u0 < units; u0 += uc) {
      const long nu = std::min(uc, units - u0);
      stage_begin(ctx, 6);
      hipLaunchKernelGGL(c128_split_corr_kernel<R>, dim3((unsigned)(nu * B * nchunk * R)), dim3(kBlock), f64::kLdsBytes, st, (const double2*)Xs,
                         (const double2*)sig->spectra64_split, Z, (const int*)ctx->items.p, (const int*)ctx->fset.p, WN, u0, P, F, D, B, pch, nchunk);
      stage_end(ctx);
      GACQ_HIP(ctx, hipGetLastError());
      stage_begin(ctx, 4);
      hipLaunchKernelGGL(c128_split_reader_kernel<R>, dim3((unsigned)(nu * P)), dim3(kBlock), 0, st, (const double2*)Z, rows, u0, P, D, B, d_qrow);
      stage_end(ctx);
      GACQ_HIP(ctx, hipGetLastError());
    }
    const long nep = (long)ne * P;
    stage_begin(ctx, 5);
    hipLaunchKernelGGL(best_doppler64_kernel, dim3((unsigned)((nep + 63) / 64)), dim3(64), 0, st, (const RowRec64*)rows, d_out + (size_t)e0 * P, nep, D, N,
                       sig->desc.metric_mode);
    stage_end(ctx);
    GACQ_HIP(ctx, hipGetLastError());
  }
  return GACQ_OK;
}

// the N = 31 x M form of engine 5: see c128_r31_*_kernel
static int r31_64_search(gacq_sig* sig, XSrc d_x, size_t nsamp, int nepoch, int P, int F, int D, int B, gacq_peak* d_out, float* d_qrow,
                         const double2* tab) {
  gacq_ctx* ctx = sig->ctx;
  hipStream_t st = ctx->stream;
  const int n = sig->desc.n, N = sig->N, M = N / 31;
  int rc;
  const double2* WN;
  if ((rc = twiddles64(ctx, N, &WN)) != GACQ_OK) return rc;
  if (!sig->spectra64_split) {
    double2* buf = nullptr;
    GACQ_HIP(ctx, hipMalloc((void**)&buf, sizeof(double2) * (size_t)sig->nprn * N));
    hipLaunchKernelGGL(c128_split_spectra_kernel, dim3((unsigned)(sig->nprn * 31)), dim3(kBlock), 0, st, (const double2*)sig->spectra64, buf, 31, M);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      (void)hipFree(buf);
      return set_error(ctx, GACQ_ERR_HIP, "complex128 radix-31 engine: code-spectrum reorder failed");
    }
    sig->spectra64_split = buf;
  }
  const size_t x_epoch_bytes = sizeof(double2) * (size_t)F * D * B * N;
  const int Ec = (int)std::max<size_t>(1, std::min<size_t>((size_t)nepoch, ws_budget(ctx) / std::max<size_t>(1, x_epoch_bytes)));
  if ((rc = ensure(ctx, ctx->X, x_epoch_bytes * Ec)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->rows, sizeof(RowRec64) * (size_t)Ec * P * D)) != GACQ_OK) return rc;
  const unsigned cols = (unsigned)((N + kBlock - 1) / kBlock);
  const int chunks = (M + kBlock - 1) / kBlock;
  for (int e0 = 0; e0 < nepoch; e0 += Ec) {
    const int ne = std::min(Ec, nepoch - e0);
    double2* X = (double2*)ctx->X.p;
    RowRec64* rows = (RowRec64*)ctx->rows.p;
    const long rows_x = (long)ne * F * D * B;
    if (rows_x * cols >= (1L << 31)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "complex128 radix-31 engine: too many forward rows in one pass (lower the workspace limit)");
    if ((rc = ensure(ctx, ctx->Y, x_epoch_bytes * ne)) != GACQ_OK) return rc;
    stage_begin(ctx, 0);
    hipLaunchKernelGGL(mix64_kernel, dim3((unsigned)(rows_x * cols)), dim3(kBlock), 0, st, d_x.offset((size_t)e0 * nsamp), nsamp, (double2*)ctx->Y.p,
                       (const double*)ctx->freq.p, tab, n, N, F * D, B, cols);
    GACQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(c128_r31_forward_kernel, dim3((unsigned)(rows_x * chunks)), dim3(kBlock), 0, st, (const double2*)ctx->Y.p, X, WN, M, chunks);
    stage_end(ctx);
    GACQ_HIP(ctx, hipGetLastError());
    stage_begin(ctx, 1);
    rc = fft_exec(ctx, M, rows_x * 31, false, X, true);
    stage_end(ctx);
    if (rc != GACQ_OK) return rc;
    const long groups = (long)ne * P * D;
    const size_t group_bytes = sizeof(double2) * (size_t)B * N;
    const size_t pass_cap = std::min(ws_budget(ctx), std::max<size_t>((size_t)4 << 30, 8 * group_bytes * (size_t)D));
    long gc = (long)std::max<size_t>(1, std::min<size_t>((size_t)groups, pass_cap / group_bytes));
    gc = std::min<long>(gc, std::max<long>(1, ((1L << 31) - 1) / ((long)B * cols)));
    if (d_qrow) gc = 1;
    if ((rc = ensure(ctx, ctx->Y, group_bytes * gc)) != GACQ_OK) return rc;
    double2* Y = (double2*)ctx->Y.p;
    for (long g0 = 0; g0 < groups; g0 += gc) {
      const long ng = std::min(gc, groups - g0);
      stage_begin(ctx, 2);
      hipLaunchKernelGGL(conj_mul64_kernel, dim3((unsigned)(ng * B * cols)), dim3(kBlock), 0, st, (const double2*)X, (const double2*)sig->spectra64_split, Y,
                         (const int*)ctx->items.p, (const int*)ctx->fset.p, g0, P, F, D, B, N, cols);
      stage_end(ctx);
      GACQ_HIP(ctx, hipGetLastError());
      stage_begin(ctx, 3);
      rc = fft_exec(ctx, M, ng * B * 31, true, Y, true);
      stage_end(ctx);
      if (rc != GACQ_OK) return rc;
      stage_begin(ctx, 4);
      hipLaunchKernelGGL(c128_r31_reader_kernel, dim3((unsigned)ng), dim3(kBlock), 0, st, (const double2*)Y, rows, WN, g0, M, B, d_qrow);
      stage_end(ctx);
      GACQ_HIP(ctx, hipGetLastError());
    }
    const long nep = (long)ne * P;
    stage_begin(ctx, 5);
    hipLaunchKernelGGL(best_doppler64_kernel, dim3((unsigned)((nep + 63) / 64)), dim3(64), 0, st, (const RowRec64*)rows, d_out + (size_t)e0 * P, nep, D, N,
                       sig->desc.metric_mode);
    stage_end(ctx);
    GACQ_HIP(ctx, hipGetLastError());
  }
  return GACQ_OK;
}

int verify_search(gacq_sig* sig, XSrc d_x, size_t nsamp, int nepoch, int P, int F, int D, int B, gacq_peak* d_out, float* d_qrow) {
  gacq_ctx* ctx = sig->ctx;
  hipStream_t st = ctx->stream;
  const int n = sig->desc.n, N = sig->N;
  int rc;
  if ((rc = verify_spectra(sig)) != GACQ_OK) return rc;
  const double2* tab;
  if ((rc = nco_table64(ctx, &tab)) != GACQ_OK) return rc;
  if (N == 4096 && B == 1 && F == 1 && !d_qrow && ctx->opt[GACQ_OPT_FUSED_C128]) {
    // one kernel per search instead of the five-stage pipeline: the row never leaves the workgroup
    const double2* tw;
    if ((rc = twiddle64_4096(ctx, &tw)) != GACQ_OK) return rc;
    if ((rc = ensure(ctx, ctx->rows, sizeof(RowRec64) * (size_t)nepoch * P * D)) != GACQ_OK) return rc;
    RowRec64* rows = (RowRec64*)ctx->rows.p;
    // one forward transform per workgroup: amortised over up to 32 items while >= ~1024 workgroups remain (2 resident per CU)
    int pch = 4;
    while (pch < 32 && (long)nepoch * D * ((P + 2 * pch - 1) / (2 * pch)) >= 1024) pch *= 2;
    pch = std::min(pch, P);
    const int nchunk = (P + pch - 1) / pch;
    const long e8 = (nepoch + 7) / 8;
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)fused4k_c128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, gacq::f64::kLdsBytes));
    stage_begin(ctx, 6);
    hipLaunchKernelGGL(fused4k_c128_kernel, dim3((unsigned)(8 * e8 * D * nchunk)), dim3(kBlock), gacq::f64::kLdsBytes, st, d_x, nsamp,
                       (const double2*)sig->spectra64, (const int*)ctx->items.p, (const double*)ctx->freq.p, tab, tw, rows, nepoch, P, D, pch, nchunk);
    stage_end(ctx);
    GACQ_HIP(ctx, hipGetLastError());
    const long nep = (long)nepoch * P;
    stage_begin(ctx, 5);
    hipLaunchKernelGGL(best_doppler64_kernel, dim3((unsigned)((nep + 63) / 64)), dim3(64), 0, st, (const RowRec64*)rows, d_out, nep, D, N, sig->desc.metric_mode);
    stage_end(ctx);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  if (N == 4096 && !d_qrow && ctx->opt[GACQ_OPT_FUSED_C128]) {
    // several blocks or carriers: forward spectra once, then one workgroup per (epoch, Doppler bin, item) over its B rows
    const double2* tw;
    if ((rc = twiddle64_4096(ctx, &tw)) != GACQ_OK) return rc;
    const size_t x_epoch_bytes = sizeof(double2) * (size_t)F * D * B * N;
    const int Ec = (int)std::max<size_t>(1, std::min<size_t>((size_t)nepoch, ws_budget(ctx) / std::max<size_t>(1, x_epoch_bytes)));
    if ((rc = ensure(ctx, ctx->X, x_epoch_bytes * Ec)) != GACQ_OK) return rc;
    if ((rc = ensure(ctx, ctx->rows, sizeof(RowRec64) * (size_t)Ec * P * D)) != GACQ_OK) return rc;
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)c128_4k_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, gacq::f64::kLdsBytes));
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)c128_4k_correlate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, gacq::f64::kLdsBytes));
    for (int e0 = 0; e0 < nepoch; e0 += Ec) {
      const int ne = std::min(Ec, nepoch - e0);
      RowRec64* rows = (RowRec64*)ctx->rows.p;
      const long rows_x = (long)ne * F * D * B;
      stage_begin(ctx, 0);
      hipLaunchKernelGGL(c128_4k_forward_kernel, dim3((unsigned)rows_x), dim3(kBlock), gacq::f64::kLdsBytes, st, d_x.offset((size_t)e0 * nsamp), nsamp, (double2*)ctx->X.p,
                         (const double*)ctx->freq.p, tab, tw, n, F * D, B);
      stage_end(ctx);
      GACQ_HIP(ctx, hipGetLastError());
      const long units8 = ((long)ne * D + 7) / 8;
      stage_begin(ctx, 6);
      hipLaunchKernelGGL(c128_4k_correlate_kernel, dim3((unsigned)(8 * units8 * P)), dim3(kBlock), gacq::f64::kLdsBytes, st, (const double2*)ctx->X.p,
                         (const double2*)sig->spectra64, (const int*)ctx->items.p, (const int*)ctx->fset.p, tw, rows, ne, P, F, D, B);
      stage_end(ctx);
      GACQ_HIP(ctx, hipGetLastError());
      const long nep = (long)ne * P;
      stage_begin(ctx, 5);
      hipLaunchKernelGGL(best_doppler64_kernel, dim3((unsigned)((nep + 63) / 64)), dim3(64), 0, st, (const RowRec64*)rows, d_out + (size_t)e0 * P, nep, D, N, sig->desc.metric_mode);
      stage_end(ctx);
      GACQ_HIP(ctx, hipGetLastError());
    }
    return GACQ_OK;
  }
  // N = 4 x 4096 / 16 x 4096: the hand-written split form (one Z' round trip, no rocFFT plan); GACQ_OPT_FUSED_C128 = 0 keeps the pipeline
  if (ctx->opt[GACQ_OPT_FUSED_C128] && N == 4 * f64::kN) return split64_search<4>(sig, d_x, nsamp, nepoch, P, F, D, B, d_out, d_qrow, tab);
  if (ctx->opt[GACQ_OPT_FUSED_C128] && N == 16 * f64::kN) return split64_search<16>(sig, d_x, nsamp, nepoch, P, F, D, B, d_out, d_qrow, tab);
  // N = 31 x M with M in rocFFT's native radices (61380, 30690): the prime stays out of rocFFT (no Bluestein)
  if (ctx->opt[GACQ_OPT_FUSED_C128] && (N == 61380 || N == 30690)) return r31_64_search(sig, d_x, nsamp, nepoch, P, F, D, B, d_out, d_qrow, tab);
  const size_t x_epoch_bytes = sizeof(double2) * (size_t)F * D * B * N;
  const int Ec = (int)std::max<size_t>(1, std::min<size_t>((size_t)nepoch, ws_budget(ctx) / std::max<size_t>(1, x_epoch_bytes)));
  if ((rc = ensure(ctx, ctx->X, x_epoch_bytes * Ec)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->rows, sizeof(RowRec64) * (size_t)Ec * P * D)) != GACQ_OK) return rc;
  const unsigned cols = (unsigned)((N + kBlock - 1) / kBlock);
  for (int e0 = 0; e0 < nepoch; e0 += Ec) {
    const int ne = std::min(Ec, nepoch - e0);
    double2* X = (double2*)ctx->X.p;
    RowRec64* rows = (RowRec64*)ctx->rows.p;
    const long rows_x = (long)ne * F * D * B;
    if (rows_x * cols >= (1L << 31)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "verification engine: too many forward rows in one pass (lower the workspace limit)");
    hipLaunchKernelGGL(mix64_kernel, dim3((unsigned)(rows_x * cols)), dim3(kBlock), 0, st, d_x.offset((size_t)e0 * nsamp), nsamp, X, (const double*)ctx->freq.p, tab, n,
                       N, F * D, B, cols);
    GACQ_HIP(ctx, hipGetLastError());
    if ((rc = fft_exec(ctx, N, rows_x, false, X, true)) != GACQ_OK) return rc;
    const long groups = (long)ne * P * D;
    const size_t group_bytes = sizeof(double2) * (size_t)B * N;
    long gc = (long)std::max<size_t>(1, std::min<size_t>((size_t)groups, ws_budget(ctx) / group_bytes));
    gc = std::min<long>(gc, std::max<long>(1, ((1L << 31) - 1) / ((long)B * cols)));
    if ((rc = ensure(ctx, ctx->Y, group_bytes * gc)) != GACQ_OK) return rc;
    double2* Y = (double2*)ctx->Y.p;
    for (long g0 = 0; g0 < groups; g0 += gc) {
      const long ng = std::min(gc, groups - g0);
      hipLaunchKernelGGL(conj_mul64_kernel, dim3((unsigned)(ng * B * cols)), dim3(kBlock), 0, st, (const double2*)X, (const double2*)sig->spectra64, Y,
                         (const int*)ctx->items.p, (const int*)ctx->fset.p, g0, P, F, D, B, N, cols);
      GACQ_HIP(ctx, hipGetLastError());
      if ((rc = fft_exec(ctx, N, ng * B, true, Y, true)) != GACQ_OK) return rc;
      hipLaunchKernelGGL(mag_peak64_kernel, dim3((unsigned)ng), dim3(kBlock), 0, st, (const double2*)Y, rows, g0, B, N, d_qrow);
      GACQ_HIP(ctx, hipGetLastError());
    }
    const long nep = (long)ne * P;
    hipLaunchKernelGGL(best_doppler64_kernel, dim3((unsigned)((nep + 63) / 64)), dim3(64), 0, st, (const RowRec64*)rows, d_out + (size_t)e0 * P, nep, D, N,
                       sig->desc.metric_mode);
    GACQ_HIP(ctx, hipGetLastError());
  }
  return GACQ_OK;
}

}  // namespace gacq
