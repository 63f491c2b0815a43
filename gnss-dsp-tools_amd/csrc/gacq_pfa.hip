// Prime-factor (Good-Thomas) form of the radix-31 split engine: N = 31 * M with M = 1980 = 11 * 20 * 9 (N = 61380: the 10.23 Mcps
// family zero-padded, acquire-gps-l5i.py:19-24, acquire-beidou-b2ad.py:19-24 and 17 more scripts) or M = 990 = 11 * 10 * 9
// (N = 30690: acquire-galileo-e6b.py:19-24, acquire-xona-x5p.py).  The four factors are pairwise coprime, so the length-N
// transform IS the 4-D transform 31 x 11 x Nb x 9 -- no twiddle factors at any level:
//
//   time index n <-> (n mod 31, n mod 11, n mod Nb, n mod 9) =: (n1, a, b, c)   (Chinese remainder map)
//   X(k1, ka, kb, kc) = sum x[n] W31^{n1 k1} W11^{a ka} WNb^{b kb} W9^{c kc}
//
// which is the DFT of x in an index order nobody needs to know: the code spectra go through the same forward kernels, the
// element-wise product C conj(X) (acquire-gps-l1.py:32) is taken position by position, and the inverse kernels undo the same maps.
// Only two places touch the natural order: the forward outer kernel gathers x (491 KB, L2-resident) and the inverse outer kernel
// labels its outputs with their lag.  What that buys over the Cooley-Tukey form of gacq_split.hip:
//   * no W_N^{n2 k1} twiddles around the DFT-31 (44 complex products per column in the forward and in the inverse outer kernel),
//   * no twiddle stages between the inner passes: 15.7 KB of LDS tables, half of each pass's LDS reads and 25 complex products per
//     butterfly are gone, and the passes work IN PLACE (a butterfly's outputs replace its own inputs), which removes the
//     read-all / barrier / write-all hand-over of the Stockham autosort: 2 barriers per row instead of 5,
//   * an LDS layout L(a,b,c) = a + 11 b + 11 Nb c in which every access of every pass is bank-conflict free (tools/model_pfa.py
//     replays the address patterns; tests/test_pfa_model.py), and exactly M slots long.
//
// Layouts in HBM (all "rows" are one k1 of one forward / correlation row):
//   spectrum rows X, C : Lg(ka,kb,kc) = ka * (Nb*9) + Nb * kc + kb                 (M contiguous, rows M apart)
//   column rows A, Z'  : Lz(a,b,c)    = c * SEG + 11 b + a,  SEG = 11 Nb rounded up to 16 (whole 128-byte lines; A rows: SEG = 11 Nb)
//   rows of either: the DFT-31 runs on slots u with n1 = (M mod 31) u, and its output k' is stored in / loaded from row
//   (M^-1 k') mod 31 (M^-1 = 23 for M = 1980, 15 for M = 990) -- compile-time register permutations.
//
// The DFT-31 of the inverse outer kernel also exists on the matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32): the conjugate-symmetric
// form A_u = v0 + sum cos(2 pi u k/31) s_k, B_u = sum sin(2 pi u k/31) d_k is two real 16 x 16 matrices applied to 16 columns at a time.
#include "gacq_common.h"
#include "gacq_cplx.h"

#include <algorithm>
#include <cmath>

using namespace gacq;

namespace {

constexpr int kR = 31;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int inv_mod(int a, int m) {
  a %= m;
  for (int x = 1; x < m; x++)
    if ((a * x) % m == 1) return x;
  return 0;
}

template <int M_> struct Pfa {
  static constexpr int M = M_, N = kR * M_;
  static constexpr int Na = 11, Nc = 9, Nb = M_ / 99;
  static_assert(Na * Nb * Nc == M_ && (Nb == 20 || Nb == 10), "M = 11 * Nb * 9");
  static constexpr int AB = Na * Nb;                 // columns (a, b) per c-segment: 220 / 110
  static constexpr int BC = Nb * Nc;                 // spectrum positions (kb, kc) per ka-segment: 180 / 90
  static constexpr int SEG = (AB + 15) & ~15;        // 224 / 112: segments of a Z' row start on 128-byte lines
  static constexpr int Mp = (Nc * SEG + 255) & ~255; // Z' row pitch: 2048 / 1024 = 9 segments + a tail of zeros (whole reader workgroups)
  static constexpr int ea = (M / Na) * inv_mod(M / Na, Na);      // CRT idempotents inside M
  static constexpr int eb = (M / Nb) * inv_mod(M / Nb, Nb);
  static constexpr int ec = (M / Nc) * inv_mod(M / Nc, Nc);
  static_assert((ea + eb + ec) % M == 1, "idempotents");
  static constexpr int Mm = M % kR, Minv = inv_mod(M, kR);
  static constexpr int NT = (M_ == 1980) ? 256 : 128;             // threads of the inner kernels: >= AB (pass c), >= BC (pass a), >= 128 (pass b)
  static_assert(SEG + (Mp - Nc * SEG) <= NT && BC <= NT && NT >= 128, "inner kernels: one butterfly per thread and pass, one thread per padding column");
};

// column position of a (padded or unpadded) column row -> time index t = n mod M of the column and the rotation q0 of its 31 lags:
// slot u of the DFT-31 is the sample / lag  n = t + M * ((q0 + u) mod 31)   (n mod 31 = (M mod 31) u)
template <int M, int SEGX> __device__ __forceinline__ bool pfa_column(int pos, int& t, int& q0) {
  using S = Pfa<M>;
  const int c = pos / SEGX, r = pos - c * SEGX;
  t = 0;
  q0 = 0;
  if (c >= S::Nc || r >= S::AB) return false;
  const int b = r / S::Na, a = r - b * S::Na;
  t = (a * S::ea + b * S::eb + c * S::ec) % M;
  q0 = ((kR - t % kR) * S::Minv) % kR;
  return true;
}

// ---- forward outer stage: rotated gather of x (+ table NCO) + DFT-31 -------------------------------------------------------------
// grid = rows * chunks; thread -> column position (unpadded).  MIX: rows = (e,f,d,b), x window of the block times the table NCO
// (acquire-gps-l1.py:28-31, gnsstools/nco.py:6-10); otherwise plain rows of N (the code replicas).  DUMP: test hook
// gacq_debug_nco_indices -- the index expression is stored instead of being used.
template <int M, bool MIX, bool DUMP = false>
__global__ __launch_bounds__(kBlock) void pfa_outer_forward_kernel(const float2* __restrict__ x, size_t epoch_stride, float2* __restrict__ A,
                                                                   const double* __restrict__ freq, const float2* __restrict__ nco_tab, int n,
                                                                   int FD, int B, int chunks) {
  using S = Pfa<M>;
  const unsigned blk = blockIdx.x;
  const int chunk = (int)(blk % (unsigned)chunks);
  const unsigned row = blk / (unsigned)chunks;
  const int pos = chunk * kBlock + threadIdx.x;
  int t, q0;
  if (!pfa_column<M, S::AB>(pos, t, q0)) return;
  const float2* src;
  double f = 0.0;
  if (MIX) {
    const int b = (int)(row % (unsigned)B);
    const unsigned r2 = row / (unsigned)B;
    const int fd = (int)(r2 % (unsigned)FD);
    const long e = r2 / (unsigned)FD;
    f = freq[fd];
    src = x + e * epoch_stride + (size_t)b * n;
  } else {
    src = x + row * (long)S::N;
  }
  // loads first, asm afterwards: the machine scheduler does not move loads across inline asm
  v2 v[kR], w[MIX ? kR : 1];
#pragma unroll
  for (int u = 0; u < kR; u++) {
    int q = q0 + u;
    q -= (q >= kR) ? kR : 0;
    const int i = t + M * q;
    if (DUMP) { reinterpret_cast<int*>(A)[row * (long)S::N + i] = nco_index(f, i); continue; }
    const float2 sf = src[i];
    v[u] = v2{sf.x, sf.y};
    if (MIX) {
      const float2 wf = nco_tab[nco_index(f, i)];      // floor((0 + f*i)*1024) mod 1024 in fp64, as numpy
      w[u] = v2{wf.x, wf.y};
    }
  }
  if (DUMP) return;
  if (MIX) {
#pragma unroll
    for (int u = 0; u < kR; u++) v[u] = cmul(v[u], w[u]);
  }
  float2* dst = A + row * (long)S::N + pos;
  OuterDft<kR, false>::run(v, [&](int kp, v2 val) { dst[(long)((S::Minv * kp) % kR) * M] = make_float2(val.x, val.y); });
}

// ---- inner passes ---------------------------------------------------------------------------------------------------------------
// Pass b (over the Nb dimension) is the one pass that touches neither global layout, so its lane -> butterfly map is free: lane l of
// 32-lane group g takes the g-th butterfly (a, c) whose base address a + AB c is congruent to l mod 32 (at most four per residue for
// both shapes): every ds_read_b64 of a group hits 32 different bank pairs, and since lanes 0-15 / 16-31 hold residues 0-15 / 16-31 the
// 16-lane store groups are conflict free too.  99 butterflies on 128 lanes.
template <int M> __device__ __forceinline__ int pfa_pass_b_base(int tid) {
  using S = Pfa<M>;
  if (tid >= 128) return -1;
  const int l = tid & 31, g = tid >> 5;
  int cnt = 0, pb = -1;
#pragma unroll
  for (int c = 0; c < S::Nc; c++) {
    const int a = (l - (S::AB % 32) * c) & 31;
    if (a < S::Na) {
      if (cnt == g) pb = a + S::AB * c;
      cnt++;
    }
  }
  return pb;
}

template <int M, bool INV> __device__ __forceinline__ void pfa_pass_b(v2* __restrict__ buf, int pb) {
  using S = Pfa<M>;
  if (pb < 0) return;
  v2 x[S::Nb];
#pragma unroll
  for (int t = 0; t < S::Nb; t++) { x[t] = buf[pb + S::Na * t]; GACQ_UNPAIR(); }
  SmallDft<S::Nb, INV>::run(x);
#pragma unroll
  for (int t = 0; t < S::Nb; t++) buf[pb + S::Na * t] = x[t];
}

// forward inner transforms, rows in place: column layout (unpadded) in, spectrum layout out.  Passes c, b, a.
template <int M>
__global__ __launch_bounds__(Pfa<M>::NT) void pfa_inner_forward_kernel(float2* __restrict__ rows, long nrows, int rpw) {
  using S = Pfa<M>;
  __shared__ __attribute__((aligned(16))) v2 buf[2][M];
  const int tid = threadIdx.x;
  const int pb = pfa_pass_b_base<M>(tid);
  const long r0 = (long)blockIdx.x * rpw;
  for (int i = 0; i < rpw && r0 + i < nrows; i++) {
    float2* row = rows + (r0 + i) * (long)M;
    v2* bw = buf[i & 1];
    if (tid < S::AB) {
      v2 y[S::Nc];
#pragma unroll
      for (int t = 0; t < S::Nc; t++) { const float2 z = row[t * S::AB + tid]; y[t] = v2{z.x, z.y}; }
      SmallDft<S::Nc, false>::run(y);
#pragma unroll
      for (int t = 0; t < S::Nc; t++) bw[tid + S::AB * t] = y[t];
    }
    __syncthreads();
    pfa_pass_b<M, false>(bw, pb);
    __syncthreads();
    if (tid < S::BC) {
      v2 xx[S::Na];
#pragma unroll
      for (int t = 0; t < S::Na; t++) { xx[t] = bw[S::Na * tid + t]; GACQ_UNPAIR(); }
      SmallDft<S::Na, false>::run(xx);
#pragma unroll
      for (int t = 0; t < S::Na; t++) row[t * S::BC + tid] = make_float2(xx[t].x, xx[t].y);
    }
  }
}

// K2 + inner inverse transforms (the writer of the Z' round trip):  Z'[g,b,k1][Lz] = IDFT_{11 x Nb x 9}( C_p[k1][.] conj(X[e,f,d,b][k1][.]) ),
// unnormalised.  Workgroup = (k1, chunk of pch consecutive (epoch, item) pairs, DT consecutive Doppler bins, block): the pass-a operands
// of the DT rows X[e,f,d..d+DT-1,b][k1][.] stay in registers while the items change, and every code-spectrum row fetched serves DT
// correlation rows.  Two row buffers: the pass-c reads of a row need no barrier before the next row's pass-a writes.
// Rising wave priority through a row (row start | before the barrier after pass a | before the barrier after pass b), as in the 4096-point
// batch kernel (gacq_ldsfft.hip, F4K_PRIO): the workgroups of a CU stay out of phase.  Worth 1-2 % at M = 1980 and 3-5 % at M = 990
// (profiles/r05_engine3_writer_wave_priority_sweep.log; falling levels and a raise only before the stores measured no better).
#ifndef GACQ_PFA_P0
#define GACQ_PFA_P0 0
#define GACQ_PFA_P1 1
#define GACQ_PFA_P2 2
#endif
#define PFA_PRIO(n) do { if ((n) >= 0) asm volatile("s_setprio %0" :: "n"(n) : "memory"); } while (0)
template <int M, int DT>
__global__ __launch_bounds__(Pfa<M>::NT, DT >= 3 ? 3 : 4) void pfa_inner_corr_kernel(
    const float2* __restrict__ X, const float2* __restrict__ C, float2* __restrict__ Z, const int* __restrict__ items, const int* __restrict__ fset,
    long g0, long ng, long ep_first, int nblk_ep, int pch, int P, int F, int D, int B) {
  using S = Pfa<M>;
  __shared__ __attribute__((aligned(16))) v2 buf[2][M];
  const int tid = threadIdx.x;
  const int pb = pfa_pass_b_base<M>(tid);
  unsigned blk = blockIdx.x;                       // 32-bit index math: 64-bit divisions cost ~100 scalar ops each
  const int b = (int)(blk % (unsigned)B);
  blk /= (unsigned)B;
  const int DG = (D + DT - 1) / DT;
  const int d0 = (int)(blk % (unsigned)DG) * DT;
  blk /= (unsigned)DG;
  const unsigned epc = blk % (unsigned)nblk_ep;
  const int k1 = (int)(blk / (unsigned)nblk_ep);
  const bool act_a = tid < S::BC, act_c = tid < S::AB;
  const float2* have[DT];
  float2 xv[DT][S::Na];
#pragma unroll
  for (int dd = 0; dd < DT; dd++) have[dd] = nullptr;
  const unsigned ep0 = (unsigned)ep_first + epc * (unsigned)pch;       // E * P < 2^31 (checked by the launcher)
  unsigned par = 0;
  float2 cv[S::Na];
  unsigned cv_ep = 0xffffffffu;                      // the (epoch, item) pair whose code-spectrum operands cv holds
  for (int i0 = 0; i0 < pch; i0++) {
    const unsigned ep = ep0 + (unsigned)i0;
    const unsigned e = ep / (unsigned)P;
    const int p = (int)(ep - e * (unsigned)P);
#pragma unroll
    for (int dd = 0; dd < DT; dd++) {
      const int d = d0 + dd;
      const long g = (long)ep * D + d;
      if (!(d < D && g >= g0 && g < g0 + ng)) continue;                    // uniform over the workgroup
      const float2* gx = X + (((((long)e * F + fset[p]) * D + d) * (long)B + b) * kR + k1) * (long)M;
      float2* gz = Z + (((g - g0) * B + b) * kR + k1) * (long)S::Mp;
      v2* bw = buf[par & 1];
      par++;
      PFA_PRIO(GACQ_PFA_P0);
      if (act_a) {
        if (cv_ep != ep) {
          const float2* gc = C + ((long)items[p] * kR + k1) * (long)M;
#pragma unroll
          for (int t = 0; t < S::Na; t++) cv[t] = gc[tid + t * S::BC];
        }
        if (gx != have[dd]) {
#pragma unroll
          for (int t = 0; t < S::Na; t++) xv[dd][t] = gx[tid + t * S::BC];
        }
        v2 x[S::Na];
#pragma unroll
        for (int t = 0; t < S::Na; t++)
          x[t] = v2{cv[t].x * xv[dd][t].x + cv[t].y * xv[dd][t].y, cv[t].y * xv[dd][t].x - cv[t].x * xv[dd][t].y};      // C * conj(X)   acquire-gps-l1.py:32
        SmallDft<S::Na, true>::run(x);
#pragma unroll
        for (int t = 0; t < S::Na; t++) bw[S::Na * tid + t] = x[t];
        if (dd == DT - 1 && i0 + 1 < pch) {
          // the last row of this item has consumed cv: fetch the next item's operands into the same registers now, so that they
          // travel under this row's passes b and c instead of holding up the next row's pass a
          const unsigned e1 = (ep + 1) / (unsigned)P;
          const float2* gc = C + ((long)items[(int)(ep + 1 - e1 * (unsigned)P)] * kR + k1) * (long)M;
#pragma unroll
          for (int t = 0; t < S::Na; t++) cv[t] = gc[tid + t * S::BC];
        }
      }
      cv_ep = (dd == DT - 1 && i0 + 1 < pch) ? ep + 1 : ep;
      have[dd] = gx;
      PFA_PRIO(GACQ_PFA_P1);
      __syncthreads();
      pfa_pass_b<M, true>(bw, pb);
      PFA_PRIO(GACQ_PFA_P2);
      __syncthreads();
      if (act_c) {
        v2 y[S::Nc];
#pragma unroll
        for (int t = 0; t < S::Nc; t++) { y[t] = bw[tid + S::AB * t]; GACQ_UNPAIR(); }
        SmallDft<S::Nc, true>::run(y);
#pragma unroll
        for (int t = 0; t < S::Nc; t++) gz[t * S::SEG + tid] = make_float2(y[t].x, y[t].y);
      } else if (tid < S::SEG) {
        // the padding columns of every segment hold zeros: the reader needs no mask for them (a zero never displaces a maximum)
#pragma unroll
        for (int t = 0; t < S::Nc; t++) gz[t * S::SEG + tid] = make_float2(0.f, 0.f);
      } else if (tid - S::SEG < S::Mp - S::Nc * S::SEG) {
        gz[S::Nc * S::SEG + tid - S::SEG] = make_float2(0.f, 0.f);            // ... and so does the tail of the row
      }
      // (two adjacent columns per thread -- 16-byte LDS reads and 16-byte row stores on half the threads -- measured the same for
      // M = 1980 and 15 % slower for M = 990: the store width is not what paces this kernel)
    }
  }
}

// Z' is read exactly once: non-temporal loads keep the 1-2 GB stream from displacing the code spectra in L2
// (profiles/r02_split_nontemporal_experiment.log).
template <bool NT>
__device__ __forceinline__ v2 ld_stream(const float2* p) {
  if (!NT) { const float2 z = *p; return v2{z.x, z.y}; }
  return __builtin_bit_cast(v2, __builtin_nontemporal_load(reinterpret_cast<const double*>(p)));
}

// (maximum, argmax, runner-up) of magnitudes visited in ANY lag order: on equal maxima the lower lag wins (np.argmax)
// (branch-free: the short-circuit forms compile to exec-mask branches, 75 of them in the unrolled scan)
__device__ __forceinline__ void add_any(Top2& top, float v, int lag) {
  const bool take = (v > top.peak) | ((v == top.peak) & (lag < top.idx));
  top.second = take ? top.peak : fmaxf(top.second, v);
  top.peak = take ? v : top.peak;
  top.idx = take ? lag : top.idx;
}

__device__ __forceinline__ void reduce_and_store(Top2 top, double sum, RowRec* __restrict__ partial, long slot, float tie_scale) {
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
  const int t = threadIdx.x;
  if ((t & 63) == 0) { s_peak[t >> 6] = top.peak; s_idx[t >> 6] = top.idx; s_second[t >> 6] = top.second; s_sum[t >> 6] = sum; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      top.merge(s_peak[w], s_idx[w], s_second[w]);
      sum += s_sum[w];
    }
    RowRec r;
    r.peak = top.peak;
    r.idx = top.tagged(tie_scale);
    r.sum = sum;
    partial[slot] = r;
  }
}

// Running (maximum, lag, runner-up) of one lane with (value, lag) as one 64-bit key -- value bits high, N-1-lag low: one
// v_cmp_gt_u64 decides "larger value, then lower lag" (np.argmax's rule) whatever order the lags arrive in.  Values are >= 0.
struct KeyScan {
  unsigned long long key = 0;                        // (value bits << 32) | (N-1 - lag)
  float second = -1.0f;
  __device__ __forceinline__ void add(float m, unsigned nl) {
    const unsigned long long k = ((unsigned long long)__float_as_uint(m) << 32) | nl;
    second = fmaxf(second, fminf(m, __uint_as_float((unsigned)(key >> 32))));
    key = (k > key) ? k : key;
  }
};

// ---- inverse outer stage + magnitude + reduce (the reader of the Z' round trip), packed-math form --------------------------------
// Z': [group][b][row k1][Lz padded].  One workgroup = 256 column positions of one group; thread = column; output u of the inverse
// DFT-31 is the lag t + M ((q0 + u) mod 31).
// MODE 0: one block, raw metric (acquire-gps-l5i.py:36 -- every 10.23 Mcps script): only the row maximum matters, so the scan runs on
//         |z|^2 (the square root is monotone: one per lane at the end instead of one per lag) and the row sum is not formed.
// MODE 1: one block, max/mean metric (acquire-xona-x5p.py:35): magnitudes and their sum.
// MODE 2: any number of blocks / row dump (gacq_debug_row): magnitudes accumulated over the blocks in registers, scanned at the end.
// Padding columns of a Z' row hold zeros (the writer stores them): they need no masking, a zero never displaces a maximum.
template <int M, int MODE>
__global__ __launch_bounds__(kBlock, MODE == 2 ? 1 : 3) void pfa_outer_inverse_kernel(const float2* __restrict__ Z, RowRec* __restrict__ partial, int B,
                                                                                      int chunks, float inv_n, float* __restrict__ q_out, float tie_scale) {
  using S = Pfa<M>;
  const unsigned blk = blockIdx.x;
  const int chunk = (int)(blk % (unsigned)chunks);
  const long g = (long)(blk / (unsigned)chunks);
  const int pos = chunk * kBlock + threadIdx.x;              // chunks * 256 == Mp
  int t, q0;
  const bool valid = pfa_column<M, S::SEG>(pos, t, q0);      // padding: t = q0 = 0
  const int nl0 = S::N - 1 - (t + M * q0);                   // key low word of output u: N-1 - lag(u) = nl0 - u M (+ N when negative)
  Top2 top;
  double sum = 0.0;
  v2 v[kR];
  if (MODE == 2) {
    float q[kR];
#pragma unroll
    for (int u = 0; u < kR; u++) q[u] = 0.f;
    for (int b = 0; b < B; b++) {
      const float2* src = Z + (g * B + b) * (long)(kR * S::Mp) + pos;
#pragma unroll
      for (int kp = 0; kp < kR; kp++) v[kp] = ld_stream<false>(src + (long)((S::Minv * kp) % kR) * S::Mp);
      OuterDft<kR, true>::run(v, [&](int u, v2 val) {
        q[u] += __builtin_amdgcn_sqrtf(norm2(val)) * inv_n;      // np.absolute(ifft(..)), 1/N folded in
      });
    }
    if (valid) {
#pragma unroll
      for (int u = 0; u < kR; u++) {
        unsigned n = (unsigned)(nl0 - u * M);
        n = min(n, n + (unsigned)S::N);
        const int lag = S::N - 1 - (int)n;
        add_any(top, q[u], lag);
        sum += (double)q[u];
        if (q_out) q_out[lag] = q[u];
      }
    }
  } else {
    const float2* src = Z + g * (long)(kR * S::Mp) + pos;
#pragma unroll
    for (int kp = 0; kp < kR; kp++) v[kp] = ld_stream<true>(src + (long)((S::Minv * kp) % kR) * S::Mp);
    KeyScan ks;
    float sum_f = 0.f;
    OuterDft<kR, true>::run(v, [&](int u, v2 val) {
      float m = norm2(val);
      if (MODE == 1) {
        m = __builtin_amdgcn_sqrtf(m) * inv_n;
        sum_f += m;
      }
      unsigned n = (unsigned)(nl0 - u * M);
      n = min(n, n + (unsigned)S::N);
      ks.add(m, n);
    });
    const float p = __uint_as_float((unsigned)(ks.key >> 32));
    // MODE 0, back to magnitudes: sqrt and the scaling are monotone, so maximum and runner-up are the same elements
    top.peak = (MODE == 0) ? __builtin_amdgcn_sqrtf(p) * inv_n : p;
    top.second = (MODE == 0) ? __builtin_amdgcn_sqrtf(fmaxf(ks.second, 0.f)) * inv_n : ks.second;
    top.idx = S::N - 1 - (int)(unsigned)(ks.key & 0xffffffffu);
    sum = (double)sum_f;
  }
  reduce_and_store(top, sum, partial, g * chunks + chunk, tie_scale);
}

// ---- the same on the matrix pipe ---------------------------------------------------------------------------------------------------
// A wave takes 16 columns at a time; lane (ci = lane & 15, gq = lane >> 4) holds, of column ci, the slots k = 4 kk + gq and 31 - k
// (kk = 0..3), i.e. exactly the K-slices v_mfma_f32_16x16x4_f32 wants in its B operand (B[k = lane >> 4][n = lane & 15]) once they are
// folded to s_k = v_k + v_{31-k}, d_k = v_k - v_{31-k}; the A operand holds the constants cos / sin(2 pi u k / 31), u = lane & 15.
// 16 matrix instructions per 16 columns (cos and sin halves x re and im x 4 K-slices) replace 450 packed FMAs per column; their
// results D[u = 4 gq + i][ci] land in the lane that needs them: z(u) = A_u + i B_u, z(31 - u) = A_u - i B_u.  The VALU keeps the folds,
// the magnitudes and the peak scan.  fp32 in, fp32 accumulate: the same arithmetic as a chain of v_fma_f32.
// Geometry: a workgroup of RW waves covers RW * RT 16-column tiles of a Z' row, RCH workgroups cover the row exactly
// (2048 = 16 * 4 * 2 * 16, 1024 = 8 * 4 * 2 * 16: no tail to mask), and walks `gpw` consecutive groups of its column range.
// (RW = 4: a workgroup puts the same number of waves on every SIMD.  Six-wave workgroups -- the exact cover of 2016 columns -- left
// half the wave slots empty: a second workgroup's 2+2+1+1 waves rarely fit beside the first one's.)
template <int M> struct PfaReader {
  static constexpr int RT = 2;
  static constexpr int RW = 4;
  static constexpr int RCH = Pfa<M>::Mp / (16 * RT * RW);
  static_assert(RCH * RW * RT * 16 == Pfa<M>::Mp, "the reader's tiles cover a Z' row exactly");
};

// One (group, block) of Z' = 31 rows of Mp as a buffer resource: the loads of the reader are `buffer_load_dwordx2 v, voffset, s[rsrc], 0
// offen offset:imm nt` -- a 32-bit per-lane offset that never changes, the tile as the immediate, the group in the SGPR descriptor.
// (With plain pointers the compiler strength-reduces the per-lane addresses of the group loop into sixteen 64-bit VGPR pairs, which
// costs the registers the load pipeline depth needs.)  Z' is read exactly once: non-temporal.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t zrow_rsrc(const float2* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ v2 ld_z(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(v2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 2));
}

// one row (31 x 48 columns of this wave) of loads: 8 per tile and lane
template <int M>
__device__ __forceinline__ void reader_load(v2 (&v)[PfaReader<M>::RT][4], v2 (&vp)[PfaReader<M>::RT][4], __amdgpu_buffer_rsrc_t r,
                                            const unsigned (&offk)[4], const unsigned (&offp)[4]) {
#pragma unroll
  for (int tt = 0; tt < PfaReader<M>::RT; tt++)
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      v[tt][kk] = ld_z(r, offk[kk] + 128u * tt);
      vp[tt][kk] = ld_z(r, offp[kk] + 128u * tt);      // slot 0 is its own partner (see mfma_table)
    }
}

// the inverse DFT-31 of one 16-column tile on the matrix pipe; out(i, m1, m2): |z(u)|^2, |z(31 - u)|^2 for u = 4 gq + i
template <class Out>
__device__ __forceinline__ void reader_tile(const v2 (&v)[4], const v2 (&vp)[4], const float (&ca)[4], const float (&sa)[4], Out&& out) {
  f32x4 aRe = {0.f, 0.f, 0.f, 0.f}, aIm = aRe, bRe = aRe, bIm = aRe;
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const v2 s = v[kk] + vp[kk], d = v[kk] - vp[kk];      // slot 0: s = 2 v0 against a cosine column of 1/2, d = 0
    aRe = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[kk], s.x, aRe, 0, 0, 0);
    aIm = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[kk], s.y, aIm, 0, 0, 0);
    bRe = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[kk], d.x, bRe, 0, 0, 0);
    bIm = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[kk], d.y, bIm, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    // z(u) = A_u + i B_u, z(31 - u) = A_u - i B_u  (plain adds: the accumulator halves are not register pairs)
    const float z1x = aRe[i] - bIm[i], z1y = aIm[i] + bRe[i], z2x = aRe[i] + bIm[i], z2y = aIm[i] - bRe[i];
    out(i, __builtin_fmaf(z1y, z1y, z1x * z1x), __builtin_fmaf(z2y, z2y, z2x * z2x));
  }
}

template <int NW>
__device__ __forceinline__ void reduce_and_store_nw(Top2 top, double sum, RowRec* __restrict__ partial, long slot, float tie_scale, int par) {
  __shared__ float s_peak[2][NW], s_second[2][NW];
  __shared__ int s_idx[2][NW];
  __shared__ double s_sum[2][NW];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float op = __shfl_down(top.peak, off);
    const int oi = __shfl_down(top.idx, off);
    const float o2 = __shfl_down(top.second, off);
    const double os = __shfl_down(sum, off);
    // Top2::merge without its short-circuit branches (six of them per wave reduction otherwise)
    top.second = fmaxf(fmaxf(top.second, o2), fminf(top.peak, op));
    const bool take = (op > top.peak) | ((op == top.peak) & (oi < top.idx));
    top.peak = take ? op : top.peak;
    top.idx = take ? oi : top.idx;
    sum += os;
  }
  const int t = threadIdx.x;
  if ((t & 63) == 0) { s_peak[par][t >> 6] = top.peak; s_idx[par][t >> 6] = top.idx; s_second[par][t >> 6] = top.second; s_sum[par][t >> 6] = sum; }
  __syncthreads();                                   // the two parities alternate: a wave cannot lap this barrier twice
  if (t == 0) {
    for (int w = 1; w < NW; w++) {
      top.merge(s_peak[par][w], s_idx[par][w], s_second[par][w]);
      sum += s_sum[par][w];
    }
    RowRec r;
    r.peak = top.peak;
    r.idx = top.tagged(tie_scale);
    r.sum = sum;
    partial[slot] = r;
  }
}

// MODE 0 / 1 / 2 and the key scan as in pfa_outer_inverse_kernel.
// MODE 0 / 1 are software-pipelined over the (group, tile) stream of the workgroup: a tile's 8 loads per lane are issued three tiles
// before it is transformed (four register sets, straight-line code -- a conditional around a tile lets the compiler sink that tile's
// loads next to their matrix instructions, one round trip per pair of rows).
template <int M, int MODE>
__global__ __launch_bounds__(PfaReader<M>::RW * 64, MODE == 2 ? 3 : 4) void pfa_outer_inverse_mfma_kernel(const float2* __restrict__ Z, RowRec* __restrict__ partial,
                                                                                      const float* __restrict__ cs_tab, int B, long ng, int nslot,
                                                                                      float inv_n, float* __restrict__ q_out, float tie_scale) {
  using S = Pfa<M>;
  using G = PfaReader<M>;
  constexpr int RT = G::RT;
  const unsigned blk = blockIdx.x;
  // The grid is persistent: workgroup (chunk, slot) keeps its column range and walks the groups in blocks of kGB, block slot, slot +
  // nslot, ...: equal shares for every workgroup whatever the number of them the chip holds at once.
  constexpr long kGB = 8;
  const int chunk = (int)(blk % (unsigned)G::RCH);
  const long slot = (long)(blk / (unsigned)G::RCH);
  const long g0 = slot * kGB;
  auto next_group = [&](long g) { return ((g + 1) % kGB) ? g + 1 : g + 1 + kGB * (nslot - 1); };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci = lane & 15, gq = lane >> 4;
  float ca[4], sa[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) { ca[kk] = cs_tab[lane * 8 + kk]; sa[kk] = cs_tab[lane * 8 + 4 + kk]; }
  const int pos0 = (chunk * G::RW + wave) * (16 * RT) + ci;
  // byte offsets of this lane's slots inside a group-block: slot k = 4 kk + gq lives in row (Minv k) mod 31, slot 31 - k in row 31 - that
  unsigned offk[4], offp[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const int rk = (S::Minv * (4 * kk + gq)) % kR;
    offk[kk] = (unsigned)(rk * S::Mp + pos0) * 8u;
    offp[kk] = (unsigned)(((kR - rk) % kR) * S::Mp + pos0) * 8u;
  }
  // key low words: nl(u) = N-1 - lag(u) = t' + M ((q0' - u) mod 31) with t' = M-1 - t, q0' = 30 - q0; this lane's outputs are
  // u = 4 gq + i (nl = na - M i, + N when negative) and 31 - u (nl = nb + M i, - N when past the end)
  int na[RT], nb[RT];
#pragma unroll
  for (int tt = 0; tt < RT; tt++) {
    int t, q0;
    (void)pfa_column<M, S::SEG>(pos0 + 16 * tt, t, q0);        // padding columns: t = q0 = 0, their data are zeros
    const int q0p = kR - 1 - q0;
    int qa = q0p + kR - 4 * gq;
    qa -= (qa >= kR) ? kR : 0;
    int qb = q0p + 4 * gq;
    qb -= (qb >= kR) ? kR : 0;
    na[tt] = (M - 1 - t) + M * qa;
    nb[tt] = (M - 1 - t) + M * qb;
  }
  const long gstride = (long)kR * S::Mp;             // complex elements per (group, block)
  if (MODE == 2) {
    for (long g = g0; g < ng; g = next_group(g)) {
      float q[RT][8];
#pragma unroll
      for (int tt = 0; tt < RT; tt++)
#pragma unroll
        for (int o = 0; o < 8; o++) q[tt][o] = 0.f;
      for (int b = 0; b < B; b++) {
        v2 v[RT][4], vp[RT][4];
        reader_load<M>(v, vp, zrow_rsrc(Z + (g * B + b) * gstride, (int)gstride * 8), offk, offp);
#pragma unroll
        for (int tt = 0; tt < RT; tt++)
          reader_tile(v[tt], vp[tt], ca, sa, [&](int i, float m1, float m2) {
            q[tt][2 * i] += __builtin_amdgcn_sqrtf(m1) * inv_n;              // np.absolute(ifft(..)), 1/N folded in
            q[tt][2 * i + 1] += __builtin_amdgcn_sqrtf(m2) * inv_n;
          });
      }
      Top2 top;
      double sum = 0.0;
#pragma unroll
      for (int tt = 0; tt < RT; tt++) {
        const bool real = pos0 + 16 * tt < S::Nc * S::SEG && (pos0 + 16 * tt) % S::SEG < S::AB;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          unsigned n1 = (unsigned)(na[tt] - M * i), n2 = (unsigned)(nb[tt] + M * i);
          n1 = min(n1, n1 + (unsigned)S::N);
          n2 = min(n2, n2 - (unsigned)S::N);
          const int la = S::N - 1 - (int)n1, lb = S::N - 1 - (int)n2;
          if (real) {
            add_any(top, q[tt][2 * i], la);
            sum += (double)q[tt][2 * i];
            if (q_out) q_out[la] = q[tt][2 * i];
            if (i > 0 || gq > 0) {                       // u = 0 has no partner
              add_any(top, q[tt][2 * i + 1], lb);
              sum += (double)q[tt][2 * i + 1];
              if (q_out) q_out[lb] = q[tt][2 * i + 1];
            }
          }
        }
      }
      reduce_and_store_nw<G::RW>(top, sum, partial, g * G::RCH + chunk, tie_scale, (int)(g & 1));
    }
    return;
  }
  // MODE 0 / 1: a rolling pipeline over the (group, tile) stream, RT register sets, RT - 1 tiles ahead
  v2 vt[RT][4], pt[RT][4];
  auto load_tile = [&](int tt, __amdgpu_buffer_rsrc_t r) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      vt[tt][kk] = ld_z(r, offk[kk] + 128u * tt);
      pt[tt][kk] = ld_z(r, offp[kk] + 128u * tt);
    }
  };
  // a group past the end gets an empty resource: its loads are out of range -- zeros, no memory traffic
  auto group_rsrc = [&](long g) { return zrow_rsrc(Z + min(g, ng - 1) * gstride, g < ng ? (int)gstride * 8 : 0); };
  {
    const __amdgpu_buffer_rsrc_t r0 = group_rsrc(g0);
#pragma unroll
    for (int tt = 0; tt < RT - 1; tt++) load_tile(tt, r0);
  }
  for (long g = g0; g < ng; g = next_group(g)) {
    const __amdgpu_buffer_rsrc_t row = group_rsrc(g);
    const __amdgpu_buffer_rsrc_t rown = group_rsrc(next_group(g));
    KeyScan ks;
    float sum_f = 0.f;
#pragma unroll
    for (int tt = 0; tt < RT; tt++) {
      const int tl = (tt + RT - 1) % RT;                         // the set freed by the previous tile
      load_tile(tl, tt == 0 ? row : rown);
      __builtin_amdgcn_sched_barrier(0);
      reader_tile(vt[tt], pt[tt], ca, sa, [&](int i, float m1, float m2) {
        if (MODE == 1) {
          m1 = __builtin_amdgcn_sqrtf(m1) * inv_n;               // np.absolute(ifft(..)), 1/N folded in
          m2 = __builtin_amdgcn_sqrtf(m2) * inv_n;
        }
        unsigned n1 = (unsigned)(na[tt] - M * i), n2 = (unsigned)(nb[tt] + M * i);
        n1 = min(n1, n1 + (unsigned)S::N);
        n2 = min(n2, n2 - (unsigned)S::N);
        if (i == 0) m2 = (gq == 0) ? 0.f : m2;                   // u = 0 has no partner: a zero at lag(0), which z(0) >= 0 already holds
        ks.add(m1, n1);
        ks.add(m2, n2);
        if (MODE == 1) sum_f += m1 + m2;
      });
    }
    Top2 top;
    const float p = __uint_as_float((unsigned)(ks.key >> 32));
    // MODE 0, back to magnitudes: sqrt and the scaling are monotone, so maximum and runner-up are the same elements
    top.peak = (MODE == 0) ? __builtin_amdgcn_sqrtf(p) * inv_n : p;
    top.second = (MODE == 0) ? __builtin_amdgcn_sqrtf(fmaxf(ks.second, 0.f)) * inv_n : ks.second;
    top.idx = S::N - 1 - (int)(unsigned)(ks.key & 0xffffffffu);
    reduce_and_store_nw<G::RW>(top, (double)sum_f, partial, g * G::RCH + chunk, tie_scale, (int)(g & 1));
  }
}

// cos / sin(2 pi u k / 31) in the lane order of the MFMA A operand: entry [lane][kk] = cos, [lane][4 + kk] = sin with
// u = lane & 15, k = 4 kk + (lane >> 4); column k = 0 is the x0 term (cos 1/2 against 2 v0, sin 0)
int mfma_table(gacq_ctx* ctx, const float** out) {
  float host[64 * 8];
  for (int lane = 0; lane < 64; lane++)
    for (int kk = 0; kk < 4; kk++) {
      const int u = lane & 15, k = 4 * kk + (lane >> 4);
      const double ang = 2.0 * M_PI * (double)((u * k) % kR) / (double)kR;
      host[lane * 8 + kk] = (k == 0) ? 0.5f : (float)std::cos(ang);      // slot 0 is loaded twice and folded to 2 v0 (exact), d_0 = 0
      host[lane * 8 + 4 + kk] = (k == 0) ? 0.f : (float)std::sin(ang);
    }
  const void* p = nullptr;
  const int rc = table_cache(ctx, "pfa_dft31_mfma", host, sizeof host, &p);
  *out = (const float*)p;
  return rc;
}

template <int M>
int forward_t(gacq_ctx* ctx, const float2* x, size_t nsamp, long rows, int n, const double* d_freq, int FD, int B, const float2* tab, float2* X,
              bool mix) {
  using S = Pfa<M>;
  const int chunks = (M + kBlock - 1) / kBlock;
  if (mix)
    hipLaunchKernelGGL((pfa_outer_forward_kernel<M, true>), dim3((unsigned)(rows * chunks)), dim3(kBlock), 0, ctx->stream, x, nsamp, X, d_freq, tab,
                       n, FD, B, chunks);
  else
    hipLaunchKernelGGL((pfa_outer_forward_kernel<M, false>), dim3((unsigned)(rows * chunks)), dim3(kBlock), 0, ctx->stream, x, nsamp, X, d_freq, tab,
                       n, FD, B, chunks);
  GACQ_HIP(ctx, hipGetLastError());
  const long nrows = rows * kR;
  const int rpw = (int)std::max<long>(1, std::min<long>(8, nrows / 2048));
  hipLaunchKernelGGL((pfa_inner_forward_kernel<M>), dim3((unsigned)((nrows + rpw - 1) / rpw)), dim3(S::NT), 0, ctx->stream, X, nrows, rpw);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

template <int M>
int inner_corr_t(gacq_ctx* ctx, const float2* X, const float2* C, const int* d_items, const int* d_fset, long g0, long ng, int P, int F, int D,
                 int B, float2* Z) {
  using S = Pfa<M>;
  // (epoch, item) rows touched by this pass, cut into chunks of pch per workgroup; >= ~2048 workgroups, <= 8 items each
  const long ep_first = g0 / D, ep_last = (g0 + ng - 1) / D;
  const long nep = ep_last - ep_first + 1;
  int pch = (int)std::max<long>(1, std::min<long>(8, nep * D * B * kR / 2048));
  if (ctx->opt[GACQ_OPT_SPLIT_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_SPLIT_PCH];
  const int nblk_ep = (int)((nep + pch - 1) / pch);
  int dt = (D >= 8) ? (M == 1980 ? 3 : 2) : 1;       // Doppler bins per workgroup
  if (ctx->opt[GACQ_OPT_SPLIT_DT] >= 1) dt = (int)ctx->opt[GACQ_OPT_SPLIT_DT];
  if (dt < 1 || dt > 3) return set_error(ctx, GACQ_ERR_BAD_ARG, "split engine: Doppler bins per workgroup must be 1, 2 or 3");
  const int DG = (D + dt - 1) / dt;
  const dim3 grid((unsigned)((long)kR * nblk_ep * DG * B));
#define GACQ_LAUNCH_PFA(DT_)                                                                                                                   \
  hipLaunchKernelGGL((pfa_inner_corr_kernel<M, DT_>), grid, dim3(S::NT), 0, ctx->stream, X, C, Z, d_items, d_fset, g0, ng, ep_first, nblk_ep, pch, \
                     P, F, D, B)
  if (dt == 3) GACQ_LAUNCH_PFA(3);
  else if (dt == 2) GACQ_LAUNCH_PFA(2);
  else GACQ_LAUNCH_PFA(1);
#undef GACQ_LAUNCH_PFA
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

template <int M>
int inverse_t(gacq_ctx* ctx, const float2* Z, RowRec* partial, int B, long ng, float inv_n, float* q_out, float tie_scale, bool need_sum, int* chunks_out) {
  using G = PfaReader<M>;
  const bool b1 = (B == 1) && !q_out;
  if (ctx->opt[GACQ_OPT_SPLIT_MFMA]) {
    *chunks_out = G::RCH;
    const float* cs = nullptr;
    const int rc = mfma_table(ctx, &cs);
    if (rc != GACQ_OK) return rc;
    // persistent grid: as many workgroups as the device holds at once (a wrong guess only costs balance: the groups are dealt out
    // in blocks of 8), a whole number of them per chunk
    const int mode = (b1 && !need_sum) ? 0 : (b1 ? 1 : 2);
    static int resident_of[3] = {0, 0, 0};
    if (!resident_of[mode]) {
      int per_cu = 0, cus = 0;
      const void* k = mode == 0 ? (const void*)pfa_outer_inverse_mfma_kernel<M, 0> : mode == 1 ? (const void*)pfa_outer_inverse_mfma_kernel<M, 1>
                                                                                             : (const void*)pfa_outer_inverse_mfma_kernel<M, 2>;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, G::RW * 64, 0) != hipSuccess || per_cu < 1) per_cu = 2;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 256;
      (void)hipGetLastError();
      resident_of[mode] = per_cu * cus;
    }
    const int resident = resident_of[mode];
    const int nslot = (int)std::max<long>(1, std::min<long>(resident / G::RCH, (ng + 7) / 8));
    const dim3 grid((unsigned)(nslot * G::RCH)), block(G::RW * 64);
    if (b1 && !need_sum) hipLaunchKernelGGL((pfa_outer_inverse_mfma_kernel<M, 0>), grid, block, 0, ctx->stream, Z, partial, cs, B, ng, nslot, inv_n, q_out, tie_scale);
    else if (b1) hipLaunchKernelGGL((pfa_outer_inverse_mfma_kernel<M, 1>), grid, block, 0, ctx->stream, Z, partial, cs, B, ng, nslot, inv_n, q_out, tie_scale);
    else hipLaunchKernelGGL((pfa_outer_inverse_mfma_kernel<M, 2>), grid, block, 0, ctx->stream, Z, partial, cs, B, ng, nslot, inv_n, q_out, tie_scale);
  } else {
    const int chunks = (Pfa<M>::Mp + kBlock - 1) / kBlock;
    *chunks_out = chunks;
    const dim3 grid((unsigned)(ng * chunks));
    if (b1 && !need_sum) hipLaunchKernelGGL((pfa_outer_inverse_kernel<M, 0>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, B, chunks, inv_n, q_out, tie_scale);
    else if (b1) hipLaunchKernelGGL((pfa_outer_inverse_kernel<M, 1>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, B, chunks, inv_n, q_out, tie_scale);
    else hipLaunchKernelGGL((pfa_outer_inverse_kernel<M, 2>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, B, chunks, inv_n, q_out, tie_scale);
  }
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace

namespace gacq {

bool pfa_supported(int N) { return N == 61380 || N == 30690; }

int pfa_row_pitch(int N) { return N == 61380 ? Pfa<1980>::Mp : (N == 30690 ? Pfa<990>::Mp : 0); }

int pfa_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, long rows, int n, int N, const double* d_freq, int FD, int B, const float2* tab,
                float2* X, bool mix) {
  if (N == 61380) return forward_t<1980>(ctx, x, nsamp, rows, n, d_freq, FD, B, tab, X, mix);
  if (N == 30690) return forward_t<990>(ctx, x, nsamp, rows, n, d_freq, FD, B, tab, X, mix);
  return set_error(ctx, GACQ_ERR_UNSUPPORTED, "prime-factor engine: N=%d not supported", N);
}

int pfa_inner_correlate(gacq_ctx* ctx, const float2* X, const float2* C, const int* d_items, const int* d_fset, long g0, long ng, int P, int F,
                        int D, int B, int N, float2* Z) {
  if (N == 61380) return inner_corr_t<1980>(ctx, X, C, d_items, d_fset, g0, ng, P, F, D, B, Z);
  if (N == 30690) return inner_corr_t<990>(ctx, X, C, d_items, d_fset, g0, ng, P, F, D, B, Z);
  return set_error(ctx, GACQ_ERR_UNSUPPORTED, "prime-factor engine: N=%d not supported", N);
}

int pfa_inverse_reduce(gacq_ctx* ctx, const float2* Z, RowRec* rows, long g0, long ng, int B, int N, float* q_out, float tie_scale, bool need_sum) {
  if (!pfa_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "prime-factor engine: N=%d not supported", N);
  const int max_chunks = pfa_row_pitch(N) / (16 * PfaReader<1980>::RT * PfaReader<1980>::RW);      // the finer of the two kernels' partial records per group
  static_assert(PfaReader<1980>::RT * PfaReader<1980>::RW * 16 <= kBlock && PfaReader<990>::RT == PfaReader<1980>::RT, "partial-record count");
  int rc = ensure(ctx, ctx->partial, sizeof(RowRec) * (size_t)ng * max_chunks);
  if (rc != GACQ_OK) return rc;
  RowRec* partial = (RowRec*)ctx->partial.p;
  const float inv_n = 1.0f / (float)N;
  int chunks = 0;
  rc = (N == 61380) ? inverse_t<1980>(ctx, Z, partial, B, ng, inv_n, q_out, tie_scale, need_sum, &chunks)
                    : inverse_t<990>(ctx, Z, partial, B, ng, inv_n, q_out, tie_scale, need_sum, &chunks);
  if (rc != GACQ_OK) return rc;
  return split_combine(ctx, partial, rows, g0, ng, chunks, tie_scale);
}

int pfa_debug_nco(gacq_ctx* ctx, int N, int n, const double* d_freq, int* d_idx) {
  if (N == 61380)
    hipLaunchKernelGGL((pfa_outer_forward_kernel<1980, true, true>), dim3((1980 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                       (const float2*)nullptr, (size_t)0, (float2*)d_idx, d_freq, (const float2*)nullptr, n, 1, 1, (1980 + kBlock - 1) / kBlock);
  else if (N == 30690)
    hipLaunchKernelGGL((pfa_outer_forward_kernel<990, true, true>), dim3((990 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                       (const float2*)nullptr, (size_t)0, (float2*)d_idx, d_freq, (const float2*)nullptr, n, 1, 1, (990 + kBlock - 1) / kBlock);
  else
    return set_error(ctx, GACQ_ERR_UNSUPPORTED, "NCO index dump: prime-factor engine does not support N=%d", N);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace gacq
