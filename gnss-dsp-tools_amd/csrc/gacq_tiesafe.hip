// Tie-safe peak locations: complex128 re-evaluation of near-tied rows, entirely on the device.
//
// The reference computes search() in complex128 and takes np.argmax / the strict-'>' Doppler scan on fp64 values
// (acquire-gps-l1.py:30-39).  The fp32 engines agree with it to ~1e-6 in the metric, which is inside north_star's 1e-5 bar -- but
// when the runner-up lag or bin is closer than that to the winner, fp32 rounding decides which one is reported (about one search
// in 50 000 on noise).  The row reductions therefore flag rows whose runner-up lag is within eps of the maximum, the Doppler scan
// (best_doppler_kernel) does the same across bins and appends every (epoch, item) it cannot decide to a device-side list together
// with its candidate rows; the kernels here then
//   tie_recheck_kernel   recompute exactly those rows in complex128 -- table-NCO mix, FFT, C_p * conj(.), inverse FFT, |.|/N,
//                        sum over blocks, (max, first argmax, sum) -- for any FFT length with prime factors in {2, 3, 5, 7, 11, 13, 31} (a
//                        mixed-radix Stockham transform through a global-memory scratch row; one workgroup per (row, block), the
//                        last one to finish a row adds its blocks up in order; N = 4096 on the LDS-resident transform), and
//   tie_resolve_kernel   redo the strict-'>' scan of the ambiguous pairs on the complex128 values and write their records.
// Nothing returns to the host: the lists are filled, consumed and reset inside the stream, a launch without ambiguous pairs
// costs two empty kernels.  The arithmetic is engine 5's (gacq_verify.hip) with another FFT implementation; both are held to
// the reference's goldens at 1e-10.
#include "gacq_common.h"
#include "gacq_fft64.h"

#include <algorithm>
#include <cmath>

using namespace gacq;

namespace {

constexpr int kTieThreads = 512;      // 8 waves: 2 per SIMD, <= 256 VGPRs each (the radix-31 pass holds 31 complex128 values per lane)
constexpr int kMaxPasses = 20;
struct Radices { int count; unsigned char r[kMaxPasses]; };

using gacq::f64::cd;

// W_N^k = exp(-2 pi i k / N), k < N, evaluated with sincospi on the exactly reduced argument
__global__ void twiddle64_kernel(double2* __restrict__ w, int N) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  double s, c;
  sincospi(-2.0 * (double)k / (double)N, &s, &c);
  w[k] = make_double2(c, s);
}

__device__ __forceinline__ cd ldc(const double2* p) { const double2 v = *p; return cd{v.x, v.y}; }
__device__ __forceinline__ void stc(double2* p, cd v) { *p = make_double2(v.x, v.y); }

// 2-, 4- and 8-point DFTs in place, natural order (the powers of two are the bulk of every length)
template <int R> __device__ __forceinline__ void pow2_dft(cd (&v)[R]) {
  using namespace gacq::f64;
  if constexpr (R == 2) {
    const cd a = v[0] + v[1], b = v[0] - v[1];
    v[0] = a; v[1] = b;
  } else if constexpr (R == 4) {
    dft4<false>(v[0], v[1], v[2], v[3]);
  } else {
    cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4<false>(e0, e1, e2, e3);
    dft4<false>(o0, o1, o2, o3);
    constexpr double h = 0.70710678118654752440;
    o1 = o1 * cd{h, -h};                      // W8^1
    o2 = cd{o2.y, -o2.x};                     // W8^2 = -i
    o3 = o3 * cd{-h, -h};                     // W8^3
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
  }
}

// One radix-R Stockham pass (decimation in frequency, autosort) over the whole row by the whole workgroup:
//   y[q + s (R p + k)] = W_n^{p k} * sum_j a[q + s (p + m j)] W_R^{j k},   n = R m, p < m, q < s
// R = 2, 4, 8: butterflies; the odd primes: the R x R sum with table twiddles, one output at a time (the k loop stays rolled).
template <int R>
__device__ void stockham_pass(const double2* __restrict__ a, double2* __restrict__ y, const double2* __restrict__ WN, int N, int n, int s) {
  using namespace gacq::f64;
  const int m = n / R, step_r = N / R, step_n = N / n;
  constexpr int kIter = R <= 8 ? 4 : 1;              // butterflies in flight per lane: their loads overlap (32 VGPRs each at R = 8)
#pragma unroll kIter
  for (int i = threadIdx.x; i < N / R; i += kTieThreads) {
    const int p = i / s, q = i - p * s;
    cd v[R];
#pragma unroll
    for (int j = 0; j < R; j++) v[j] = ldc(a + q + s * (p + m * j));
    if constexpr (R == 2 || R == 4 || R == 8) {
      pow2_dft<R>(v);
      stc(y + q + s * (R * p), v[0]);
#pragma unroll
      for (int k = 1; k < R; k++) stc(y + q + s * (R * p + k), p ? v[k] * ldc(WN + (int)(((long)p * k) % n) * step_n) : v[k]);
    } else {
#pragma unroll 1
      for (int k = 0; k < R; k++) {
        cd acc = v[0];
#pragma unroll
        for (int j = 1; j < R; j++) acc = acc + v[j] * ldc(WN + ((j * k) % R) * step_r);
        if (k && p) acc = acc * ldc(WN + (int)(((long)p * k) % n) * step_n);
        stc(y + q + s * (R * p + k), acc);
      }
    }
  }
}

// forward FFT of the row in `a` (N values) using `b` as the second buffer; returns the buffer that holds the result
__device__ double2* fft_row(double2* a, double2* b, const double2* __restrict__ WN, int N, const Radices& rad) {
  int n = N, s = 1;
  for (int ip = 0; ip < rad.count; ip++) {
    const int r = rad.r[ip];
    __syncthreads();
    switch (r) {
      case 2: stockham_pass<2>(a, b, WN, N, n, s); break;
      case 3: stockham_pass<3>(a, b, WN, N, n, s); break;
      case 4: stockham_pass<4>(a, b, WN, N, n, s); break;
      case 5: stockham_pass<5>(a, b, WN, N, n, s); break;
      case 7: stockham_pass<7>(a, b, WN, N, n, s); break;
      case 8: stockham_pass<8>(a, b, WN, N, n, s); break;
      case 11: stockham_pass<11>(a, b, WN, N, n, s); break;
      case 13: stockham_pass<13>(a, b, WN, N, n, s); break;
      default: stockham_pass<31>(a, b, WN, N, n, s); break;
    }
    double2* t = a; a = b; b = t;
    n /= r;
    s *= r;
  }
  __syncthreads();
  return a;
}

// Workgroup-wide (max, first argmax, sum) of the magnitudes val(i), i < N, visited in ascending order per thread
template <class F> __device__ __forceinline__ void reduce_row(int N, F val, TieRec* dst) {
  __shared__ double s_peak[kTieThreads], s_sum[kTieThreads];
  __shared__ int s_idx[kTieThreads];
  const int t = threadIdx.x;
  double peak = -1.0, sum = 0.0;
  int idx = 0x7fffffff;
  // eight values per lane are fetched before the first compare: the loads behind val() are independent, the strict-'>' scan is not --
  // taken one at a time, every value costs a full memory round trip on an otherwise idle chip
  constexpr int kBatch = 8;
  for (int i0 = t; i0 < N; i0 += kBatch * (int)blockDim.x) {
    double v[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; u++) {
      const int i = i0 + u * (int)blockDim.x;
      v[u] = i < N ? val(i) : -1.0;
    }
#pragma unroll
    for (int u = 0; u < kBatch; u++) {                                 // ascending i: strict '>' keeps the first maximum
      const int i = i0 + u * (int)blockDim.x;
      if (i < N) {
        if (v[u] > peak) { peak = v[u]; idx = i; }
        sum += v[u];
      }
    }
  }
  s_peak[t] = peak; s_sum[t] = sum; s_idx[t] = idx;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if (t < off) {
      const double op = s_peak[t + off];
      const int oi = s_idx[t + off];
      if (op > s_peak[t] || (op == s_peak[t] && oi < s_idx[t])) { s_peak[t] = op; s_idx[t] = oi; }
      s_sum[t] += s_sum[t + off];
    }
    __syncthreads();
  }
  if (t == 0) { TieRec r; r.peak = s_peak[0]; r.sum = s_sum[0]; r.idx = s_idx[0]; r.pad = 0; *dst = r; }
  __syncthreads();
}

// Hand-over between the workgroups that evaluate the B blocks of one row side by side: each stores its per-block magnitudes
// (plain stores), releases them at agent scope and counts its arrival; the one that finds B - 1 earlier arrivals acquires and
// adds the B rows up in block order (the same order a single workgroup accumulates them in: identical bits) and reduces.
// MI355X_MICROARCH.md, inter-workgroup visibility: plain stores -> barrier -> lane-0 release fence -> s_waitcnt vmcnt(0) -> relaxed
// agent atomic; consumer: agent acquire on one lane -> barrier -> plain loads.
__device__ __forceinline__ bool arrive_last(unsigned* counter, int B) {
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned earlier = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (int)earlier == B - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
  }
  __syncthreads();
  return s_last != 0;
}

// Rows to re-evaluate in this launch.  A list that overflowed is dropped WHOLE: which pairs found room depends on the arrival order of
// the listing kernel's atomics, and a result that differs from run to run (or from rank to rank of a sharded merge, where every rank
// resolves the same gathered records on its own) is worse than fp32 locations that are at least reproducible.  The overflow test
// itself -- all candidate rows of the launch against the capacity -- does not depend on the order.
__device__ __forceinline__ unsigned list_count(const TieLists& tl) {
  const unsigned n = tl.c->nrows;
  return n > (unsigned)tl.cap ? 0u : n;
}

// Work item = (listed row, block) when the per-block magnitude rows fit (qb != nullptr: B workgroups share a row, ~1 / B of the
// latency -- a re-evaluation sits in the stream between two searches), else one listed row with its blocks in sequence.
//   q[k] = sum_b | ifft( C_p * conj(fft(x[b n : b n + N] * nco)) )[k] |   in complex128   (acquire-gps-l1.py:28-35;
// |ifft(Y)| = |fft(conj(Y))| / N, so one forward transform routine serves both directions)
__global__ __launch_bounds__(kTieThreads) void tie_recheck_kernel(TieLists tl, unsigned* __restrict__ done, double* __restrict__ qb,
                                                                   XSrc x, size_t epoch_stride,
                                                                   const double2* __restrict__ C64, const int* __restrict__ items,
                                                                   const int* __restrict__ fset, const double* __restrict__ freq,
                                                                   const double2* __restrict__ tab64, const double2* __restrict__ WN,
                                                                   char* __restrict__ scratch, int n, int N, int P, int D, int B, Radices rad) {
  const unsigned count = list_count(tl);
  const unsigned nwork = qb ? count * (unsigned)B : count;
  char* mine = scratch + (size_t)blockIdx.x * ((size_t)N * 40);
  double2* bufa = reinterpret_cast<double2*>(mine);
  double2* bufb = bufa + N;
  double* qown = reinterpret_cast<double*>(bufb + N);
  const double inv_n = 1.0 / (double)N;
  const int t = threadIdx.x;
  for (unsigned w = blockIdx.x; w < nwork; w += gridDim.x) {
    const unsigned slot = qb ? w / (unsigned)B : w;
    const int b0 = qb ? (int)(w % (unsigned)B) : 0, b1 = qb ? b0 + 1 : B;
    const TieRow row = tl.rows[slot];
    if (row.ep < 0) continue;                                            // voided slot of a pair that did not fit (uniform over the workgroup)
    const long e = row.ep / P;
    const int p = row.ep - (int)e * P;
    const double f = freq[(long)fset[p] * D + row.d];
    const double2* Cp = C64 + (long)items[p] * N;
    double* q = qb ? qb + ((size_t)slot * B + b0) * N : qown;
    for (int b = b0; b < b1; b++) {
      const XSrc src = x.offset(e * epoch_stride + (size_t)b * n);
      for (int i = t; i < N; i += kTieThreads) {
        const double2 sv = ld_x(src, i);
        const double2 wv = tab64[nco_index(f, i)];                    // gnsstools/nco.py:6-10
        bufa[i] = make_double2(sv.x * wv.x - sv.y * wv.y, sv.x * wv.y + sv.y * wv.x);
      }
      double2* X = fft_row(bufa, bufb, WN, N, rad);
      double2* other = (X == bufa) ? bufb : bufa;
      for (int i = t; i < N; i += kTieThreads) {
        const double2 xv = X[i], cv = Cp[i];                          // conj(C_p * conj(X)) = conj(C_p) * X
        X[i] = make_double2(cv.x * xv.x + cv.y * xv.y, cv.x * xv.y - cv.y * xv.x);
      }
      double2* Z = fft_row(X, other, WN, N, rad);
      for (int i = t; i < N; i += kTieThreads) {
        const double m = gacq::f64::sqrt_pos(Z[i].x * Z[i].x + Z[i].y * Z[i].y) * inv_n;
        q[i] = (b == b0) ? m : q[i] + m;
      }
      __syncthreads();                                                 // the row buffers are rewritten by the next block's mix
    }
    if (qb) {
      if (!arrive_last(done + slot, B)) continue;
      const double* q0 = qb + (size_t)slot * B * N;
      reduce_row(N, [&](int i) { double v = q0[i]; for (int b = 1; b < B; b++) v += q0[(size_t)b * N + i]; return v; }, tl.recs + slot);
    } else {
      reduce_row(N, [&](int i) { return q[i]; }, tl.recs + slot);
    }
  }
}

// N = 4096 (GPS L1 C/A, Xona X1: the headline shape, where a step of 32768 searches has a handful of ambiguous pairs): the same work
// items on the LDS-resident complex128 transform of gacq_fft64.h -- ~10 us per (row, block) instead of ~50 us through the
// global-memory Stockham passes above, so that the re-evaluation stays invisible next to the 5.6 ms search it follows.
__global__ __launch_bounds__(256, 1) void tie_recheck4k_kernel(TieLists tl, unsigned* __restrict__ done, double* __restrict__ qb,
                                                               XSrc x, size_t epoch_stride,
                                                               const double2* __restrict__ C64, const int* __restrict__ items,
                                                               const int* __restrict__ fset, const double* __restrict__ freq,
                                                               const double2* __restrict__ tab64, const double2* __restrict__ WN, int n, int P,
                                                               int D, int B) {
  using namespace gacq::f64;
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  __shared__ double s_peak[4], s_sum[4];
  __shared__ int s_idx[4];
  const unsigned count = list_count(tl);
  const unsigned nwork = qb ? count * (unsigned)B : count;
  const int t = threadIdx.x;
  const double2 wa2 = WN[t], wb2 = WN[16 * (t & 15)];
  const cd wa = {wa2.x, wa2.y}, wb = {wb2.x, wb2.y};
  const double inv_n = 1.0 / (double)kN;
  for (unsigned w = blockIdx.x; w < nwork; w += gridDim.x) {
    const unsigned slot = qb ? w / (unsigned)B : w;
    const int b0 = qb ? (int)(w % (unsigned)B) : 0, b1 = qb ? b0 + 1 : B;
    const TieRow row = tl.rows[slot];
    if (row.ep < 0) continue;
    const long e = row.ep / P;
    const int p = row.ep - (int)e * P;
    const double f = freq[(long)fset[p] * D + row.d];
    const double2* cp = C64 + (long)items[p] * kN + t;
    double q[16];
    for (int b = b0; b < b1; b++) {
      const XSrc src = x.offset(e * epoch_stride + (size_t)b * n);
      cd v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int i = t + 256 * j;
        const double2 sv = ld_x(src, i);
        const double2 wv = tab64[nco_index(f, i)];
        v[j] = cd{sv.x, sv.y} * cd{wv.x, wv.y};
      }
      fft4096<false>(v, lds64, wa, wb, t);
      cd y[16];
#pragma unroll
      for (int j = 0; j < 16; j++) { const double2 c = cp[256 * j]; y[j] = cd{c.x, c.y} * conj(v[rev16(j)]); }
      __syncthreads();
      fft4096<true>(y, lds64, wa, wb, t);
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const cd r = y[rev16(k)];
        const double m = sqrt_pos(r.x * r.x + r.y * r.y) * inv_n;
        q[k] = (b == b0) ? m : q[k] + m;
      }
      __syncthreads();
    }
    if (qb) {
      double* mineq = qb + ((size_t)slot * B + b0) * kN;
#pragma unroll
      for (int k = 0; k < 16; k++) mineq[t + 256 * k] = q[k];
      if (!arrive_last(done + slot, B)) continue;
      const double* q0 = qb + (size_t)slot * B * kN;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        double v = q0[t + 256 * k];
        for (int b = 1; b < B; b++) v += q0[(size_t)b * kN + t + 256 * k];
        q[k] = v;
      }
    }
    double peak = -1.0, sum = 0.0;
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (q[k] > peak) { peak = q[k]; idx = t + 256 * k; }
      sum += q[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double op = __shfl_down(peak, off);
      const int oi = __shfl_down(idx, off);
      const double os = __shfl_down(sum, off);
      if (op > peak || (op == peak && oi < idx)) { peak = op; idx = oi; }
      sum += os;
    }
    if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = idx; s_sum[t >> 6] = sum; }
    __syncthreads();
    if (t == 0) {
      for (int w2 = 1; w2 < 4; w2++) {
        if (s_peak[w2] > peak || (s_peak[w2] == peak && s_idx[w2] < idx)) { peak = s_peak[w2]; idx = s_idx[w2]; }
        sum += s_sum[w2];
      }
      TieRec r; r.peak = peak; r.sum = sum; r.idx = idx; r.pad = 0;
      tl.recs[slot] = r;
    }
    __syncthreads();
  }
}

// N = R x 4096, R = 4 or 16 (BeiDou B1I / B2I, GLONASS L1 / L2: 16384; Galileo E1B / E1C: 65536): the row split like engine 4's,
// n = 4096 n1 + n2, k = k1 + R k2, with the 4096-point transforms on the LDS-resident complex128 transform.  Work item = (listed
// row, block, k1): one workgroup forms A[k1][n2] = W_N^{n2 k1} sum_n1 x[4096 n1 + n2] W_R^{n1 k1} straight from the samples (mix
// included), transforms it, multiplies by conj(C_p[k1 + R k2]), transforms again (|ifft(Y)| = |fft(conj Y)| / N; the second
// transform decimates in time, so its inner transforms run over the same k1 rows) and stores W_N^{k1 n2} T_k1[n2]; the last of the
// R workgroups of a (row, block) to arrive does the outer DFT-R and the magnitudes, the last block of a row the ordered sum and
// the reduction.  Two global round trips per block instead of ten Stockham passes, R-fold parallel.  Measured: two GLONASS rows of ten
// blocks (80 work items) in 0.18 ms on an otherwise idle chip (profiles/r04_tie_recheck_kernel_times.log); four dependent phases of
// complex128 work at idle clocks plus two agent-scope hand-overs per row are what is left.
template <int R>
__global__ __launch_bounds__(256, 1) void tie_recheck_split_kernel(TieLists tl, unsigned* __restrict__ done, unsigned* __restrict__ done2,
                                                                   double* __restrict__ qb, double2* __restrict__ zs,
                                                                   XSrc x, size_t epoch_stride,
                                                                   const double2* __restrict__ C64, const int* __restrict__ items,
                                                                   const int* __restrict__ fset, const double* __restrict__ freq,
                                                                   const double2* __restrict__ tab64, const double2* __restrict__ WN, int n, int P,
                                                                   int D, int B) {
  using namespace gacq::f64;
  constexpr int M = kN, N = R * kN;
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  const unsigned count = list_count(tl);
  const unsigned nwork = count * (unsigned)B * (unsigned)R;
  const int t = threadIdx.x;
  const cd wa = ldc(WN + t * R), wb = ldc(WN + 16 * (t & 15) * R);      // W_4096^t, W_256^(t & 15)
  const double inv_n = 1.0 / (double)N;
  for (unsigned w = blockIdx.x; w < nwork; w += gridDim.x) {
    const int k1 = (int)(w % (unsigned)R);
    const unsigned sb = w / (unsigned)R;                                 // (slot, block)
    const unsigned slot = sb / (unsigned)B;
    const int b = (int)(sb % (unsigned)B);
    const TieRow row = tl.rows[slot];
    if (row.ep < 0) continue;
    const long e = row.ep / P;
    const int p = row.ep - (int)e * P;
    const double f = freq[(long)fset[p] * D + row.d];
    const double2* Cp = C64 + (long)items[p] * N;
    const XSrc src = x.offset(e * epoch_stride + (size_t)b * n);
    cd v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = cd{0.0, 0.0};
    for (int n1 = 0; n1 < R; n1++) {
      // the 16 sample loads and the 16 table gathers of one n1 are independent: issued together, one round trip each
      double2 sv[16];
      double2 wv[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int i = M * n1 + t + 256 * j;
        sv[j] = ld_x(src, i);
        wv[j] = tab64[nco_index(f, i)];                                // gnsstools/nco.py:6-10
      }
      const cd wr = ldc(WN + ((n1 * k1) % R) * M);                     // W_R^{n1 k1}
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = v[j] + (cd{sv[j].x, sv[j].y} * cd{wv[j].x, wv[j].y}) * wr;
    }
    {
      cd tw[16];
#pragma unroll
      for (int j = 0; j < 16; j++) tw[j] = ldc(WN + (t + 256 * j) * k1);   // W_N^{n2 k1}  (n2 k1 < N)
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = v[j] * tw[j];
    }
    fft4096<false>(v, lds64, wa, wb, t);
    cd y[16];
#pragma unroll
    for (int j = 0; j < 16; j++) y[j] = ldc(Cp + k1 + R * (t + 256 * j));      // register rev16(j) holds k2 = t + 256 j: X[k1 + R k2]
#pragma unroll
    for (int j = 0; j < 16; j++) y[j] = conj(y[j]) * v[rev16(j)];              // conj(C_p * conj(X))
    __syncthreads();
    fft4096<false>(y, lds64, wa, wb, t);
    double2* zrow = zs + ((size_t)sb * R + k1) * M;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int n2 = t + 256 * j;
      stc(zrow + n2, y[rev16(j)] * ldc(WN + n2 * k1));                 // W_N^{k1 n2} T_k1[n2]
    }
    if (!arrive_last(done2 + sb, R)) continue;
    // outer DFT-R over k1 for every n2, magnitudes of this block's row
    const double2* z0 = zs + (size_t)sb * R * M;
    double* q = qb + (size_t)sb * N;
    constexpr int JB = R == 4 ? 4 : 1;                                 // columns per batch: 16 loads in flight either way
    for (int j0 = 0; j0 < 16; j0 += JB) {
      cd z[JB][R];
#pragma unroll
      for (int jj = 0; jj < JB; jj++)
#pragma unroll
        for (int kk = 0; kk < R; kk++) z[jj][kk] = ldc(z0 + (size_t)kk * M + t + 256 * (j0 + jj));
#pragma unroll
      for (int jj = 0; jj < JB; jj++) {
        const int n2 = t + 256 * (j0 + jj);
        if constexpr (R == 4) {
          dft4<false>(z[jj][0], z[jj][1], z[jj][2], z[jj][3]);
#pragma unroll
          for (int n1 = 0; n1 < 4; n1++) q[M * n1 + n2] = sqrt_pos(z[jj][n1].x * z[jj][n1].x + z[jj][n1].y * z[jj][n1].y) * inv_n;
        } else {
          dft16<false>(z[jj]);
#pragma unroll
          for (int n1 = 0; n1 < 16; n1++) {
            const cd r = z[jj][rev16(n1)];
            q[M * n1 + n2] = sqrt_pos(r.x * r.x + r.y * r.y) * inv_n;
          }
        }
      }
    }
    if (!arrive_last(done + slot, B)) continue;
    const double* q0 = qb + (size_t)slot * B * N;
    reduce_row(N, [&](int i) { double a = q0[i]; for (int bb = 1; bb < B; bb++) a += q0[(size_t)bb * N + i]; return a; }, tl.recs + slot);
  }
}

// Strict-'>' scan of every ambiguous (epoch, item) over its re-evaluated rows, in Doppler order, running best starting at 0
// (acquire-gps-l1.py:25,36-39).  Rows that were not re-evaluated lie more than eps below the fp32 winner and cannot win.  One
// workgroup; resets the list counters for the next launch.  d_shift: Doppler index of the grid's first bin in the caller's
// numbering (0 for a plain search).
__global__ __launch_bounds__(256) void tie_resolve_kernel(TieLists tl, gacq_peak* __restrict__ out, const gacq_peak* __restrict__ fp32_guess, int N,
                                                           int normalised) {
  const unsigned neps = min(tl.c->neps, (unsigned)tl.cap);
  const bool dropped = tl.c->nrows > (unsigned)tl.cap;      // see list_count(): every listed pair keeps its fp32 answer as well
  unsigned moved = 0;
  for (unsigned i = threadIdx.x; i < neps; i += blockDim.x) {
    const TieEp ep = tl.eps[i];
    if (dropped) { out[ep.ep] = fp32_guess[i]; continue; }
    double best = 0.0;
    int bidx = -1, bd = -1;
    for (int s = ep.slot0; s < ep.slot0 + ep.cnt; s++) {
      const TieRec r = tl.recs[s];
      const double m = normalised ? r.peak / (r.sum / (double)N) : r.peak;
      if (m > best) { best = m; bidx = r.idx; bd = tl.rows[s].d; }
    }
    gacq_peak o;
    o.metric = best;
    o.idx = bidx;
    o.d_index = bd;
    const gacq_peak g = fp32_guess[i];
    if (g.idx != bidx || g.d_index != bd) moved++;
    out[ep.ep] = o;
  }
  if (moved) atomicAdd(&tl.c->moved, (unsigned long long)moved);
  __syncthreads();
  if (threadIdx.x == 0) {
    tl.c->flagged += neps;
    if (dropped) tl.c->overflow += neps;      // (the pairs that found no room were counted by the listing kernel)
    else tl.c->rows += tl.c->nrows;
    tl.c->nrows = 0;
    tl.c->neps = 0;
  }
}

// forward transform of one row per workgroup, in place (through a scratch row); see tie_code_spectra64
__global__ __launch_bounds__(kTieThreads) void code_spectra64_kernel(double2* __restrict__ rows, double2* __restrict__ scratch,
                                                                      const double2* __restrict__ WN, int N, Radices rad) {
  double2* a = rows + (size_t)blockIdx.x * N;
  double2* b = scratch + (size_t)blockIdx.x * N;
  const double2* r = fft_row(a, b, WN, N, rad);
  if (r != a)
    for (int i = threadIdx.x; i < N; i += kTieThreads) a[i] = r[i];
}

bool factorise(int N, Radices& rad) {
  rad.count = 0;
  for (int r : {8, 4, 2, 3, 5, 7, 11, 13, 31}) {
    while (N % r == 0) {
      if (rad.count == kMaxPasses) return false;
      rad.r[rad.count++] = (unsigned char)r;
      N /= r;
    }
  }
  return N == 1;
}

}  // namespace

namespace gacq {

bool tie_supported(int N) {
  Radices rad;
  return factorise(N, rad);
}

// W_N^k, k < N, in fp64 (device-side sincospi), cached per context
int twiddles64(gacq_ctx* ctx, int N, const double2** out) {
  const std::string key = "W64_" + std::to_string(N);
  auto wt = ctx->tables.find(key);
  if (wt == ctx->tables.end()) {
    DevBuf b;
    GACQ_HIP(ctx, hipMalloc(&b.p, sizeof(double2) * (size_t)N));
    b.cap = sizeof(double2) * (size_t)N;
    hipLaunchKernelGGL(twiddle64_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, (double2*)b.p, N);
    GACQ_HIP(ctx, hipGetLastError());
    wt = ctx->tables.emplace(key, b).first;
  }
  *out = (const double2*)wt->second.p;
  return GACQ_OK;
}

// The complex128 code spectra c = fft.fft(c) (acquire-gps-l1.py:24) with the library's own transform: rows[p] holds the zero-extended
// replica on entry and its spectrum, natural order, on return.  One workgroup per row, the Stockham passes of the re-evaluation kernel.
// GACQ_ERR_UNSUPPORTED for lengths it does not factor (the caller then falls back to rocFFT's double-precision transform).  Why not
// rocFFT for all of them: creating its plan for a new length costs 0.5-1.7 s (runtime-compiled kernels, nothing cached across processes)
// -- per PROCESS, i.e. per run of an acquire-*.py replacement -- for a transform that is executed once.
int tie_code_spectra64(gacq_ctx* ctx, double2* rows, int nrows, int N) {
  Radices rad;
  if (!factorise(N, rad)) return GACQ_ERR_UNSUPPORTED;
  const double2* WN;
  int rc = twiddles64(ctx, N, &WN);
  if (rc != GACQ_OK) return rc;
  double2* scratch = nullptr;
  GACQ_HIP(ctx, hipMalloc((void**)&scratch, sizeof(double2) * (size_t)nrows * N));
  hipLaunchKernelGGL(code_spectra64_kernel, dim3((unsigned)nrows), dim3(kTieThreads), 0, ctx->stream, rows, scratch, WN, N, rad);
  const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ctx->stream);
  (void)hipFree(scratch);
  if (e1 != hipSuccess || e2 != hipSuccess) return set_error(ctx, GACQ_ERR_HIP, "complex128 code spectra: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  return GACQ_OK;
}

// What the re-evaluation may allocate next to the search's own workspaces: an eighth of the workspace limit the caller has set
// (gacq_set_workspace_limit), between 32 and 256 MiB.  The row buffers of tie_resolve are sized for the LIST CAPACITY (the host does
// not know how many rows a launch will flag), so the automatic capacity is bounded by this budget: pairs beyond it keep their fp32
// answer and are counted (gacq_get_tie_stats()[2]) -- for the long lengths a handful of rows per call is what real data produces.
// A caller who sets the capacity himself (GACQ_OPT_TIE_CAP) gets what he asked for; the side-by-side row buffers are then used up to
// 768 MiB, beyond that the sequential forms (a few rows of scratch).
size_t tie_budget(const gacq_ctx* ctx) {
  if (ctx->opt[GACQ_OPT_TIE_CAP] > 0) return (size_t)768 << 20;
  return std::min<size_t>((size_t)256 << 20, std::max<size_t>((size_t)32 << 20, std::min(ctx->ws_limit, ctx->ws_soft ? ctx->ws_soft : ctx->ws_limit) / 8));
}

int tie_capacity(const gacq_ctx* ctx, long nep, int N, int B) {
  long cap = ctx->opt[GACQ_OPT_TIE_CAP] > 0 ? std::min<long>(ctx->opt[GACQ_OPT_TIE_CAP], 1L << 24) : std::min<long>(64 + nep / 16, 1L << 20);
  // per listed row: B per-block magnitude rows in fp64 (+ the twiddled inner transforms of the split kernels)
  const size_t row = (size_t)std::max(1, B) * (size_t)N * (sizeof(double) + ((N == 4 * gacq::f64::kN || N == 16 * gacq::f64::kN) ? sizeof(double2) : 0)) + 64;
  if (ctx->opt[GACQ_OPT_TIE_CAP] <= 0) cap = std::min<long>(cap, (long)std::max<size_t>(16, tie_budget(ctx) / row));
  return (int)cap;
}

float tie_scale_of(const gacq_ctx* ctx) {
  const long ppb = std::max<long>(0, std::min<long>(ctx->opt[GACQ_OPT_TIE_EPS_PPB], 1000000000L));
  return (float)(1.0 - (double)ppb * 1e-9);
}

// layout of ctx->tie: [TieCounters | fp32 guesses (gacq_peak x cap) | TieEp x cap | TieRow x cap | TieRec x cap]
int tie_lists(gacq_ctx* ctx, long nep, int N, int B, TieLists* out, gacq_peak** guesses) {
  const int cap = tie_capacity(ctx, nep, N, B);
  const size_t bytes = 64 + (size_t)cap * (sizeof(unsigned) * 2 + sizeof(gacq_peak) + sizeof(TieEp) + sizeof(TieRow) + sizeof(TieRec));
  if (cap > ctx->tie_cap || !ctx->tie.p) {
    TieCounters keep{};
    if (ctx->tie.p) {
      GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
      GACQ_HIP(ctx, hipMemcpy(&keep, ctx->tie.p, sizeof keep, hipMemcpyDeviceToHost));
    }
    int rc = ensure(ctx, ctx->tie, bytes);
    if (rc != GACQ_OK) return rc;
    keep.nrows = keep.neps = 0;
    // on the ctx stream (a non-blocking stream is not ordered against the null stream a plain hipMemset runs on), then waited for:
    // `keep` lives on this frame
    GACQ_HIP(ctx, hipMemsetAsync(ctx->tie.p, 0, ctx->tie.cap, ctx->stream));      // the per-row arrival counters start (and are left) at zero
    GACQ_HIP(ctx, hipMemcpyAsync(ctx->tie.p, &keep, sizeof keep, hipMemcpyHostToDevice, ctx->stream));
    GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tie_cap = cap;
  }
  if (!ctx->pin_tie.p) {
    int rc = ensure_pinned(ctx, ctx->pin_tie, 64);
    if (rc != GACQ_OK) return rc;
    *(volatile unsigned*)ctx->pin_tie.p = 0u;
  }
  out->host_full = (unsigned*)ctx->pin_tie.p;
  // the lists are always laid out for the allocated capacity; a smaller request only lowers the fill limit
  char* base = (char*)ctx->tie.p;
  const int lay = ctx->tie_cap;
  out->c = (TieCounters*)base;
  out->done = (unsigned*)(base + 64);                                 // [lay] arrival counters, 8-byte stride keeps what follows aligned
  base += (size_t)lay * sizeof(unsigned) * 2;
  *guesses = (gacq_peak*)(base + 64);
  out->eps = (TieEp*)(base + 64 + (size_t)lay * sizeof(gacq_peak));
  out->rows = (TieRow*)((char*)out->eps + (size_t)lay * sizeof(TieEp));
  out->recs = (TieRec*)((char*)out->rows + (size_t)lay * sizeof(TieRow));
  out->cap = cap;
  return GACQ_OK;
}

int verify_spectra(gacq_sig* s);      // gacq_verify.hip: complex128 code spectra, built on first use

int tie_prepare(gacq_sig* sig) {
  if (!tie_supported(sig->N)) return GACQ_OK;
  return verify_spectra(sig);
}

// Re-evaluate the listed rows and rewrite the records of the ambiguous pairs; asynchronous on the ctx stream.
// ctx->freq / fset / items hold the grid of the search that filled the lists.
int tie_resolve(gacq_sig* sig, const TieLists& tl, const gacq_peak* guesses, XSrc d_x, size_t nsamp, int P, int D, int B, gacq_peak* d_out) {
  gacq_ctx* ctx = sig->ctx;
  const int N = sig->N;
  Radices rad;
  if (!factorise(N, rad)) return set_error(ctx, GACQ_ERR_INTERNAL, "tie_resolve: N=%d has a prime factor > 31", N);
  const void* p = nullptr;
  int rc;
  auto it = ctx->tables.find("nco64");
  if (it == ctx->tables.end()) {
    std::vector<double2> h(kNcoTableSize);
    for (int k = 0; k < kNcoTableSize; k++) {
      const double a = 2.0 * M_PI * (double)k * (1.0 / kNcoTableSize);       // np.exp(2*pi*1j*np.arange(NT)*(1.0/NT))  nco.py:4
      h[k] = make_double2(std::cos(a), std::sin(a));
    }
    if ((rc = table_cache(ctx, "nco64", h.data(), sizeof(double2) * kNcoTableSize, &p)) != GACQ_OK) return rc;
  } else {
    p = it->second.p;
  }
  const double2* tab64 = (const double2*)p;
  const double2* WN;
  if ((rc = twiddles64(ctx, N, &WN)) != GACQ_OK) return rc;
  const int Rs = (N == 4 * gacq::f64::kN) ? 4 : ((N == 16 * gacq::f64::kN) ? 16 : 0);
  const size_t split_bytes = (size_t)tl.cap * B * ((size_t)N * (sizeof(double) + sizeof(double2)) + 64);
  // B > 1: the blocks of a listed row are evaluated side by side by B workgroups when their per-block magnitude rows
  // (capacity x B x N fp64 values) fit into the budget; otherwise one workgroup takes the row's blocks in sequence (same bits)
  double* qb = nullptr;
  const size_t qb_bytes = (size_t)tl.cap * B * N * sizeof(double);
  const size_t budget = tie_budget(ctx);
  if (B > 1 && qb_bytes <= budget && !(Rs && split_bytes <= budget)) {
    if ((rc = ensure(ctx, ctx->tie_q, qb_bytes)) != GACQ_OK) return rc;
    qb = (double*)ctx->tie_q.p;
    // the per-row arrival counters are left at zero by the kernel; cleared per launch all the same (see above)
    GACQ_HIP(ctx, hipMemsetAsync(tl.done, 0, sizeof(unsigned) * (size_t)tl.cap, ctx->stream));
  }
  if (Rs && split_bytes <= budget) {
    // per-(row, block) arrival counters in a buffer of their own (inside the data buffer their place would move with the capacity
    // and land on stale row data).  The kernels leave them at zero; they are cleared per launch all the same (a few KB): a faulted
    // or aborted launch must not leave the next one waiting on stale counts
    const size_t cnt_bytes = (size_t)tl.cap * B * sizeof(unsigned);
    if ((rc = ensure(ctx, ctx->tie_done2, cnt_bytes)) != GACQ_OK) return rc;
    GACQ_HIP(ctx, hipMemsetAsync(ctx->tie_done2.p, 0, cnt_bytes, ctx->stream));
    GACQ_HIP(ctx, hipMemsetAsync(tl.done, 0, sizeof(unsigned) * (size_t)tl.cap, ctx->stream));
    // [per-block magnitude rows | twiddled inner transforms]
    const size_t q_bytes = (size_t)tl.cap * B * N * sizeof(double);
    if ((rc = ensure(ctx, ctx->tie_split, split_bytes)) != GACQ_OK) return rc;
    unsigned* done2 = (unsigned*)ctx->tie_done2.p;
    double* q2 = (double*)ctx->tie_split.p;
    double2* zs = (double2*)((char*)q2 + q_bytes);
    auto kern = Rs == 4 ? tie_recheck_split_kernel<4> : tie_recheck_split_kernel<16>;
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, gacq::f64::kLdsBytes));
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), gacq::f64::kLdsBytes, ctx->stream, tl, tl.done, done2, q2, zs, d_x, nsamp,
                       (const double2*)sig->spectra64, (const int*)ctx->items.p, (const int*)ctx->fset.p, (const double*)ctx->freq.p, tab64, WN,
                       sig->desc.n, P, D, B);
    GACQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(tie_resolve_kernel, dim3(1), dim3(256), 0, ctx->stream, tl, d_out, guesses, N, sig->desc.metric_mode);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  if (N == gacq::f64::kN) {
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)tie_recheck4k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, gacq::f64::kLdsBytes));
    hipLaunchKernelGGL(tie_recheck4k_kernel, dim3(128), dim3(256), gacq::f64::kLdsBytes, ctx->stream, tl, tl.done, qb, d_x, nsamp,
                       (const double2*)sig->spectra64, (const int*)ctx->items.p, (const int*)ctx->fset.p, (const double*)ctx->freq.p, tab64, WN,
                       sig->desc.n, P, D, B);
    GACQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(tie_resolve_kernel, dim3(1), dim3(256), 0, ctx->stream, tl, d_out, guesses, N, sig->desc.metric_mode);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  // workgroups in flight: enough to take a burst of rows side by side, bounded by 64 MiB (or the budget) of row scratch (40 bytes per point)
  const size_t row_bytes = (size_t)N * 40;
  const int G = (int)std::max<size_t>(4, std::min<size_t>(64, std::min<size_t>((size_t)64 << 20, budget) / row_bytes));
  if ((rc = ensure(ctx, ctx->tie_scratch, row_bytes * G)) != GACQ_OK) return rc;
  hipLaunchKernelGGL(tie_recheck_kernel, dim3((unsigned)G), dim3(kTieThreads), 0, ctx->stream, tl, tl.done, qb, d_x, nsamp,
                     (const double2*)sig->spectra64, (const int*)ctx->items.p, (const int*)ctx->fset.p, (const double*)ctx->freq.p, tab64, WN,
                     (char*)ctx->tie_scratch.p, sig->desc.n, N, P, D, B, rad);
  GACQ_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(tie_resolve_kernel, dim3(1), dim3(256), 0, ctx->stream, tl, d_out, guesses, N, sig->desc.metric_mode);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace gacq

extern "C" int gacq_get_tie_stats(gacq_ctx* ctx, long long out[4]) {
  if (!ctx || !out) return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_get_tie_stats: bad argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!ctx->tie.p) return GACQ_OK;
  GACQ_DEVICE(ctx);
  GACQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  TieCounters c;
  GACQ_HIP(ctx, hipMemcpy(&c, ctx->tie.p, sizeof c, hipMemcpyDeviceToHost));
  out[0] = (long long)c.flagged;
  out[1] = (long long)c.rows;
  out[2] = (long long)c.overflow;
  out[3] = (long long)c.moved;
  return GACQ_OK;
}
