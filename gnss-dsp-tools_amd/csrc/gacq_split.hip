// Split engines: FFT length N = R * M, a hand-written R-point OUTER DFT over the stride-M dimension fused with the
// neighbouring element-wise stages, and length-M INNER transforms over contiguous rows.
//
//   n = M n1 + n2, k = k1 + R k2:
//   X[k1 + R k2] = sum_{n2} W_M^{n2 k2} * ( W_N^{n2 k1} * sum_{n1} x[M n1 + n2] W_R^{n1 k1} )
//
//   forward : split_outer_forward_kernel (NCO mix + DFT-R over n1 + twiddle W_N^{n2 k1})            K1 + outer
//             inner forward transforms of length M, batch R*rows  -> spectrum stored as [k1][k2]
//             (the code spectra use the same order, so the element-wise conj-multiply K2 is unchanged)
//   inverse : inner inverse transforms of length M, batch R*rows
//             split_outer_inverse_kernel (twiddle + inverse DFT-R + |.|/N + sum over blocks + max/argmax/sum)  outer + K3
//
// R = 31, M = 1980 / 990 (N = 61380 / 30690: the 10.23 Mcps family, acquire-gps-l5i.py:19-24 and 18 more scripts, and
//   E6, acquire-galileo-e6b.py:19-24).  rocFFT has no radix-31 butterfly and falls back to Bluestein for these lengths
//   (three transforms of twice the size per FFT); M = 4*5*9*11 is native to it.  The DFT-31 uses the conjugate symmetry
//   of W_31 (gacq_cplx.h: dft_prime), a quarter of the 31 x 31 complex products.
// R = 4 / 16 / 20 / 40, M = 4096 (N = 16384 / 65536 / 81920 / 163840: B1I, GLONASS, E1B/E1C, L1C, B1C, L2CM): the inner transforms are single-kernel and the
//   magnitude/reduce stage is fused into the outer inverse DFT, so the correlation workspace is read once less.
#include "gacq_common.h"
#include "gacq_cplx.h"

#include <cmath>

using namespace gacq;

namespace {

// ---- forward outer stage ---------------------------------------------------------------------------
// grid = rows * chunks; thread -> n2.  MIX: multiply by the table NCO (rows = (e,f,d,b)); otherwise plain rows.
// DUMP (test hook gacq_debug_nco_indices, MIX only): the index expression is stored as int32 into A (reinterpreted), x is not read.
template <int R, bool MIX, bool DUMP = false>
__global__ __launch_bounds__(kBlock) void split_outer_forward_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                                    float2* __restrict__ A, const double* __restrict__ freq,
                                                                    const float2* __restrict__ nco_tab,
                                                                    const float2* __restrict__ tw, int n, int M, int FD, int B,
                                                                    int chunks) {
  const unsigned blk = blockIdx.x;                 // 32-bit index math throughout: a 64-bit division is ~100 scalar ops
  const int chunk = (int)(blk % (unsigned)chunks);
  const unsigned row = blk / (unsigned)chunks;
  const int n2 = chunk * kBlock + threadIdx.x;
  if (n2 >= M) return;
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
    src = x + row * (long)(R * M);
  }
  // loads first, asm afterwards: the machine scheduler does not move loads across inline asm
  v2 v[R], w[MIX ? R : 1];
#pragma unroll
  for (int n1 = 0; n1 < R; n1++) {
    const int i = M * n1 + n2;
    if (DUMP) { reinterpret_cast<int*>(A)[row * (long)(R * M) + i] = nco_index(f, (int)i); continue; }
    const float2 sf = src[i];
    v[n1] = v2{sf.x, sf.y};
    if (MIX) {
      // table NCO, index in fp64 exactly as numpy: floor((0 + f*i)*1024) mod 1024   (gnsstools/nco.py:6-9)
      const int k = nco_index(f, (int)i);
      const float2 wf = nco_tab[k];
      w[n1] = v2{wf.x, wf.y};
    }
  }
  if (DUMP) return;
  const float2 twf = tw[n2];            // W_N^{n2}
  if (MIX) {
#pragma unroll
    for (int n1 = 0; n1 < R; n1++) v[n1] = cmul(v[n1], w[n1]);
  }
  TwPow tp;
  tp.init<R - 1>(v2{twf.x, twf.y});
  float2* dst = A + row * (long)(R * M) + n2;
  OuterDft<R, false>::run(v, [&](int k1, v2 val) {
    const v2 o = tp.apply(val, k1);
    dst[(long)k1 * M] = make_float2(o.x, o.y);
  });
}


// Z' is read exactly once: non-temporal loads (`global_load_dwordx2 ... nt`) keep the 1-2 GB stream from displacing the code spectra
// and twiddles in L2 and measured 3-17 % faster than plain loads on the reading side (profiles/r02_split_nontemporal_experiment.log).
template <bool NT>
__device__ __forceinline__ v2 ld_stream(const float2* p) {
  if (!NT) { const float2 z = *p; return v2{z.x, z.y}; }
  return __builtin_bit_cast(v2, __builtin_nontemporal_load(reinterpret_cast<const double*>(p)));
}

// ---- inverse outer stage + magnitude + reduce ------------------------------------------------------------
// Z: [group][b][k1][n2] after the inner inverse transforms (unnormalised).  One workgroup handles 256 values of n2
// of one group and emits a partial (peak, idx, sum) record; idx = M n1 + n2.
// B1 (one block, no row dump): magnitudes are reduced as the DFT produces them instead of being accumulated in q[R]; that
// and the 3-waves-per-SIMD register budget let a third wave hide the R strided loads of the other two.
template <int R, bool TW, bool B1>
__global__ __launch_bounds__(kBlock, (B1 && R >= 31) ? 3 : 1) void split_outer_inverse_kernel(
    const float2* __restrict__ Z, RowRec* __restrict__ partial, const float2* __restrict__ tw, int M, int Mp, int B, int chunks, float inv_n,
    float* __restrict__ q_out, int paired, float tie_scale) {
  __shared__ float s_peak[kBlock / 64], s_second[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  __shared__ double s_sum[kBlock / 64];
  constexpr bool kNT = B1 || R != 31;              // the multi-block R = 31 kernel (E6B, B3I, E5: 0.94 -> 0.99 ms) is the one case that loses
  const unsigned blk = blockIdx.x;
  const int chunk = (int)(blk % (unsigned)chunks);
  const long g = (long)(blk / (unsigned)chunks);
  // position of this thread's column inside a row of Z'.  paired (engine 4, M = 4096): the LDS inner kernel stores a row in its
  // lane-pair layout (n2 = t + 256 j at (j >> 1) * 512 + 2 t + (j & 1), 16 bytes per lane and store); columns are independent here,
  // so the thread simply serves whichever n2 lives at its position.  (The same layout for the Stockham kernel's last pass -- 7 x 16 + 8
  // instead of 15 x 8 bytes per lane -- changed nothing on the writing side and cost the reader 4 %: not kept.)
  const int pos = chunk * kBlock + threadIdx.x;
  const int n2 = paired ? ((pos & 511) >> 1) + 256 * (((pos >> 9) << 1) | (pos & 1)) : pos;
  Top2 top;                                        // (maximum, first argmax, runner-up): the runner-up makes the location tie-safe
  double sum = 0.0;
  if (n2 < M) {
    // loads first, asm afterwards: the machine scheduler does not move loads across inline asm
    v2 v[R];
    {
      const float2* src = Z + (g * B) * (long)(R * Mp) + pos;     // rows are Mp apart (128-byte aligned pitch)
#pragma unroll
      for (int k1 = 0; k1 < R; k1++) v[k1] = ld_stream<kNT>(src + (long)k1 * Mp);
    }
    TwPow tp;
    if (TW) {
      const float2 wf = tw[n2];
      const v2 wv = {wf.x, -wf.y};      // conj: W_N^{-n2}
      tp.init<R - 1>(wv);
    }
    if (B1) {
      if (TW) {
#pragma unroll
        for (int k1 = 1; k1 < R; k1++) v[k1] = tp.apply(v[k1], k1);
      }
      // outputs arrive as n1 = 0, then pairs (k, R-k) for the prime radices, in natural order otherwise: the first half is
      // reduced on the fly, late arrivals (n1 > next expected) are parked so that the scan stays in ascending lag order
      float late[R];
      float sum_f = 0.f;
      int expect = 0;
      OuterDft<R, true>::run(v, [&](int n1, v2 val) {
        const float m = __builtin_amdgcn_sqrtf(norm2(val)) * inv_n;      // np.absolute(ifft(..)), 1/N folded in
        if (n1 == expect) {
          top.add(m, M * n1 + n2);
          sum_f += m;
          expect++;
        } else {
          late[n1] = m;
        }
      });
#pragma unroll
      for (int n1 = 0; n1 < R; n1++) {
        if (n1 >= expect) {                            // compile-time after unrolling: expect is a constant by now
          top.add(late[n1], M * n1 + n2);
          sum_f += late[n1];
        }
      }
      sum = (double)sum_f;
    } else {
    float q[R];
#pragma unroll
    for (int k = 0; k < R; k++) q[k] = 0.f;
    for (int b = 0; b < B; b++) {
      if (b > 0) {
        const float2* src = Z + (g * B + b) * (long)(R * Mp) + pos;
#pragma unroll
        for (int k1 = 0; k1 < R; k1++) v[k1] = ld_stream<kNT>(src + (long)k1 * Mp);
      }
      if (TW) {
#pragma unroll
        for (int k1 = 1; k1 < R; k1++) v[k1] = tp.apply(v[k1], k1);
      }
      OuterDft<R, true>::run(v, [&](int n1, v2 val) {
        q[n1] += __builtin_amdgcn_sqrtf(norm2(val)) * inv_n;      // np.absolute(ifft(..)), 1/N folded in
      });
    }
#pragma unroll
    for (int n1 = 0; n1 < R; n1++) {          // ascending idx = M n1 + n2: strict '>' keeps the first maximum
      top.add(q[n1], M * n1 + n2);
      sum += (double)q[n1];
      if (q_out) q_out[M * n1 + n2] = q[n1];
    }
    }
  }
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
    partial[g * chunks + chunk] = r;      // = blockIdx.x
  }
}

// Phase timing (diagnostic builds only, -DGACQ_PHASE_TIMING; tools/phase_timing.sh): thread 0 of every workgroup accumulates the
// shader-clock cycles between marks and adds them to gacq_phase_cycles[] once at the end (read back with
// gacq_debug_phase_cycles).  Never defined in the product build.
#ifdef GACQ_PHASE_TIMING
__device__ unsigned long long gacq_phase_cycles[32];
struct PhaseAcc { unsigned long long t[16]; };
#define GACQ_MARK(i)                                                                              \
  do {                                                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter();                                 \
    acc_.t[i] += now_ - mark_;                                                                    \
    mark_ = now_;                                                                                 \
  } while (0)
#define GACQ_MARK_ARG , unsigned long long& mark_, PhaseAcc& acc_, int mark_base_
#define GACQ_MARK_PASS(b) , mark_, acc_, b
#else
#define GACQ_MARK(i) do { } while (0)
#define GACQ_MARK_ARG
#define GACQ_MARK_PASS(b)
#endif

// ---- inner inverse transforms fused with the conj-multiply (engine 3, M = 1980 / 990) ---------------------------------
// Z[ry][n2] = IFFT_M( C_p[k1][.] * conj(X[e,f,d,b][k1][.]) )[n2]  (unnormalised), one workgroup per (group, block, k1) row.
// Stockham autosort in LDS (one buffer, see stockham_pass), mixed radices R0*R1*R2*R3 = M; the first pass reads the two spectra from global
// memory and multiplies them (K2), so the product never exists in HBM; the last pass writes the row.  Thread j of a
// radix-R pass with Ns = product of the previous radices:  k = j mod Ns; inputs in[j + t M/R] * W_{Ns R}^{-k t};
// R-point DFT; outputs out[(j div Ns) Ns R + k + t Ns].
// Twiddles of the pass with (Ns, R) live in their own LDS table twp[(t-1) Ns + k] = conj(W_{Ns R}^{k t}), k < Ns: consecutive
// lanes read consecutive words (indexing one shared W_M table by k t M/(Ns R) put up to 32 lanes on one bank).
template <int R, int NT>
__device__ __forceinline__ void fill_pass_twiddles(v2* __restrict__ twp, const float2* __restrict__ twm_g, int Ns, int M) {
  const int step = M / (Ns * R);
  for (int q = threadIdx.x; q < (R - 1) * Ns; q += NT) {
    const int t = q / Ns + 1, k = q - (t - 1) * Ns;
    const float2 w = twm_g[k * t * step];          // k t step < M
    twp[q] = v2{w.x, -w.y};
  }
}

// One radix-R pass, in place: every thread pulls its butterflies' inputs into registers (one LDS read per instruction, GACQ_UNPAIR in
// gacq_cplx.h: config 4 3.17 -> 3.09 ms per step; not in the one-Doppler-bin instantiations, which then spill two registers), the
// workgroup synchronises, then the
// outputs overwrite the same buffer (autosort order).  One buffer instead of a ping-pong pair keeps the workgroup at
// 2 M complex of LDS (row + twiddles) so five of them fit a CU.  LAST writes the row to global memory instead.
template <int R, bool LAST, int M, int NT, bool UNPAIR = true>
__device__ __forceinline__ void stockham_pass(v2* __restrict__ buf, float2* __restrict__ gz, const v2* __restrict__ twp, int Ns, int tid,
                                              bool live GACQ_MARK_ARG) {
  constexpr int nb = M / R;
  constexpr int iters = (nb + NT - 1) / NT;
  v2 x[iters][R];
#pragma unroll
  for (int it = 0; it < iters; it++) {
    const int j = tid + it * NT;
    if (j < nb && live) {
      const int k = j % Ns;
      v2 wv[R];
#pragma unroll
      for (int t = 0; t < R; t++) { x[it][t] = buf[j + t * nb]; if (UNPAIR) GACQ_UNPAIR(); if (t) { wv[t] = twp[(t - 1) * Ns + k]; if (UNPAIR) GACQ_UNPAIR(); } }      // conj(W_{Ns R}^{k t})
#pragma unroll
      for (int t = 1; t < R; t++) x[it][t] = cmul(x[it][t], wv[t]);
      SmallDft<R, true>::run(x[it]);
    }
  }
  GACQ_MARK(mark_base_);                           // LDS reads, twiddle products, R-point DFT
  if (!LAST) __syncthreads();                      // all inputs are in registers
  GACQ_MARK(mark_base_ + 1);
#pragma unroll
  for (int it = 0; it < iters; it++) {
    const int j = tid + it * NT;
    if (j < nb && live) {
      const int k = j % Ns;
      const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
      for (int t = 0; t < R; t++) {
        if (LAST) gz[j0 + t * Ns] = make_float2(x[it][t].x, x[it][t].y);
        else buf[j0 + t * Ns] = x[it][t];
      }
    }
  }
}

// Workgroup = (k1, chunk of pch consecutive (epoch, item) pairs, Doppler bin, block).  TEAMS teams of NT threads (whole waves)
// share one set of per-pass twiddle tables in LDS and work on TEAMS consecutive items at a time, each team in its own row buffer;
// a team keeps the first-pass operands of X[e,f,d,b][k1][.] in registers while its items change (reloaded only when the row
// pointer changes, i.e. across an epoch or frequency-set boundary).  One team per workgroup (31.6 KB for M = 1980) lets five
// workgroups = 15 waves share a CU, and the kernel waits on LDS round trips and barriers two thirds of the time; four teams
// amortise the 15.7 KB of tables over four rows (79 KB per workgroup, two workgroups = 24 waves per CU, the register limit).
// Consecutive workgroups share k1 and the item chunk, so the pch code-spectrum rows they read stay in every XCD's L2.
// [g0, g0+ng) is the range of (e,p,d) groups whose Z rows exist in this workspace pass; anything outside is skipped (the team
// idles through the barriers).
// R3 == 1: three passes (R0, R1, R2).  NT threads per team, chosen close to the butterflies per pass:
//   M = 1980 = 11 * 12 * 15: 180, 165, 132 butterflies -> 192 threads;   M = 990 = 11 * 9 * 10: 90, 110, 99 -> 128 threads.
// Three passes instead of four (11 * 9 * 5 * 4|2) mean one LDS exchange, one twiddle stage and two barriers less per row; the
// composite radices 10, 12 and 15 are coprime products (PfaDft), so they cost no internal twiddles either.
// DT consecutive Doppler bins per workgroup: the first-pass operands of DT rows X[e,f,d..d+DT-1,b][k1][.] stay in registers and
// every code-spectrum row fetched for an item serves DT correlation rows.  Phase timing (profiles/r02_stockham_phase_timing*.log)
// shows a row spending 44 % of its time waiting for the 16 KB of C it reads and 20 % issuing the 16 KB of Z' it writes -- the
// CU's memory pipeline, not arithmetic or LDS, paces this kernel -- so halving the C traffic per row is what pays.
template <int R0, int R1, int R2, int R3, int NT, int TEAMS, int DT>
__global__ __launch_bounds__(NT * TEAMS, (DT == 1 ? 5 : (DT == 2 && R0 * R1 * R2 * R3 == 990 ? 4 : 3))) void split_inner_corr_kernel(const float2* __restrict__ X, const float2* __restrict__ C,
                                                                   float2* __restrict__ Z, const int* __restrict__ items,
                                                                   const int* __restrict__ fset, const float2* __restrict__ twm_g,
                                                                   long g0, long ng, long ep_first, int nblk_ep, int pch, int P, int F,
                                                                   int D, int B, int R, int Mp) {
  constexpr int M = R0 * R1 * R2 * R3;
  constexpr int nb0 = M / R0;
  static_assert(nb0 <= NT, "first pass: one radix-R0 butterfly per thread");
  static_assert(NT % 64 == 0, "teams are whole waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int team = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NT));      // wave-uniform: keeps the row pointers in SGPRs
  const int j = (int)threadIdx.x - team * NT;
  v2* buf0 = reinterpret_cast<v2*>(smem) + team * M;
  v2* tw1 = reinterpret_cast<v2*>(smem) + TEAMS * M;        // per-pass twiddle tables, M - R0 entries in all
  v2* tw2 = tw1 + (R1 - 1) * R0;
  v2* tw3 = tw2 + (R2 - 1) * R0 * R1;
  fill_pass_twiddles<R1, NT * TEAMS>(tw1, twm_g, R0, M);
  fill_pass_twiddles<R2, NT * TEAMS>(tw2, twm_g, R0 * R1, M);
  if (R3 > 1) fill_pass_twiddles<R3, NT * TEAMS>(tw3, twm_g, R0 * R1 * R2, M);
  unsigned blk = blockIdx.x;                       // 32-bit index math: 64-bit divisions cost ~100 scalar ops each
  const int b = (int)(blk % (unsigned)B);
  blk /= (unsigned)B;
  const int DG = (D + DT - 1) / DT;                // Doppler groups
  const int d0 = (int)(blk % (unsigned)DG) * DT;
  blk /= (unsigned)DG;
  const unsigned epc = blk % (unsigned)nblk_ep;
  const int k1 = (int)(blk / (unsigned)nblk_ep);
  const bool act = j < nb0;
  const float2* have[DT];
  float2 xv[DT][R0];
#pragma unroll
  for (int dd = 0; dd < DT; dd++) have[dd] = nullptr;
  const unsigned ep0 = (unsigned)ep_first + epc * (unsigned)pch;       // E * P < 2^31 (checked by the launcher)
#ifdef GACQ_PHASE_TIMING
  PhaseAcc acc_;
#pragma unroll
  for (int i = 0; i < 16; i++) acc_.t[i] = 0;
  unsigned long long mark_ = __builtin_readcyclecounter();
#endif
  for (int i0 = 0; i0 < pch; i0 += TEAMS) {
    const unsigned ep = ep0 + (unsigned)(i0 + team);
    const unsigned e = ep / (unsigned)P;
    const int p = (int)(ep - e * (unsigned)P);
    const bool item_ok = i0 + team < pch;
    float2 cv[R0];
    bool have_c = false;
#pragma unroll
    for (int dd = 0; dd < DT; dd++) {
      const int d = d0 + dd;
      const long g = (long)ep * D + d;
      const bool live = item_ok && d < D && g >= g0 && g < g0 + ng;       // uniform over the team
      float2* gz = nullptr;
      if (live) {
        const float2* gx = X + (((((long)e * F + fset[p]) * D + d) * (long)B + b) * R + k1) * (long)M;
        gz = Z + (((g - g0) * B + b) * R + k1) * (long)Mp;
        if (act) {
          if (!have_c) {
            const float2* gc = C + ((long)items[p] * R + k1) * (long)M;
#pragma unroll
            for (int t = 0; t < R0; t++) cv[t] = gc[j + t * nb0];
          }
          if (gx != have[dd]) {
#pragma unroll
            for (int t = 0; t < R0; t++) xv[dd][t] = gx[j + t * nb0];
          }
          v2 x[R0];
#pragma unroll
          for (int t = 0; t < R0; t++)
            x[t] = v2{cv[t].x * xv[dd][t].x + cv[t].y * xv[dd][t].y, cv[t].y * xv[dd][t].x - cv[t].x * xv[dd][t].y};      // C * conj(X)   acquire-gps-l1.py:32
          SmallDft<R0, true>::run(x);
#pragma unroll
          for (int t = 0; t < R0; t++) buf0[j * R0 + t] = x[t];                                             // Ns = 1: k = 0, no twiddles
        }
        have_c = true;
        have[dd] = gx;
      }
      GACQ_MARK(0);                                // (wait for C, X) C conj(X), DFT-R0, LDS writes
      __syncthreads();
      GACQ_MARK(1);
      stockham_pass<R1, false, M, NT, (DT > 1)>(buf0, nullptr, tw1, R0, j, live GACQ_MARK_PASS(2));
      GACQ_MARK(4);                                // pass-2 outputs written to LDS
      __syncthreads();
      GACQ_MARK(5);
      if (R3 > 1) {
        stockham_pass<R2, false, M, NT, (DT > 1)>(buf0, nullptr, tw2, R0 * R1, j, live GACQ_MARK_PASS(10));
        __syncthreads();
        stockham_pass<(R3 > 1 ? R3 : 2), true, M, NT, (DT > 1)>(buf0, gz, tw3, R0 * R1 * R2, j, live GACQ_MARK_PASS(12));
      } else {
        stockham_pass<R2, true, M, NT, (DT > 1)>(buf0, gz, tw2, R0 * R1, j, live GACQ_MARK_PASS(6));
      }
      GACQ_MARK(8);                                // last pass: row stored to global memory
      __syncthreads();                             // buf0 is rewritten by the next row's first pass
      GACQ_MARK(9);
    }
  }
#ifdef GACQ_PHASE_TIMING
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 16; i++) if (acc_.t[i]) atomicAdd(&gacq_phase_cycles[i], acc_.t[i]);
  }
#endif
}

// partial[(g, chunk)] -> rows[g0 + g]
__global__ void split_combine_kernel(const RowRec* __restrict__ partial, RowRec* __restrict__ rows, long g0, long ng, int chunks, float tie_scale) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const RowRec* part = partial + g * chunks;
  RowRec best;
  combine_tagged(chunks, [&](int c) { return part[c].peak; }, [&](int c) { return part[c].idx; }, tie_scale, best.peak, best.idx);
  best.sum = part[0].sum;
  for (int c = 1; c < chunks; c++) best.sum += part[c].sum;
  rows[g0 + g] = best;
}

// W_N^k for k < M (the per-n2 base twiddles of the outer stage)
int base_twiddles(gacq_ctx* ctx, int N, int M, const float2** out) {
  return twiddle_cache(ctx, "WN_base_" + std::to_string(N), N, M, out);
}

bool smooth(int m) {
  for (int p : {2, 3, 5, 7, 11, 13}) while (m % p == 0) m /= p;
  return m == 1;
}

template <int R>
int launch_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, long rows, int n, int M, const double* d_freq, int FD, int B,
                   const float2* tab, const float2* tw, float2* X, bool mix) {
  const int chunks = (M + kBlock - 1) / kBlock;
  if (mix)
    hipLaunchKernelGGL((split_outer_forward_kernel<R, true>), dim3((unsigned)(rows * chunks)), dim3(kBlock), 0, ctx->stream, x, nsamp,
                       X, d_freq, tab, tw, n, M, FD, B, chunks);
  else
    hipLaunchKernelGGL((split_outer_forward_kernel<R, false>), dim3((unsigned)(rows * chunks)), dim3(kBlock), 0, ctx->stream, x, nsamp,
                       X, d_freq, tab, tw, n, M, FD, B, chunks);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

template <int R>
int launch_inverse(gacq_ctx* ctx, const float2* Z, RowRec* partial, const float2* tw, int M, int Mp, int B, long ng, float inv_n,
                   float* q_out, bool twiddle, int paired, float tie_scale) {
  const int chunks = (M + kBlock - 1) / kBlock;
  const dim3 grid((unsigned)(ng * chunks));
  const bool b1 = (B == 1) && !q_out;
  if (twiddle && b1)
    hipLaunchKernelGGL((split_outer_inverse_kernel<R, true, true>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, tw, M, Mp, B, chunks, inv_n, q_out, paired, tie_scale);
  else if (twiddle)
    hipLaunchKernelGGL((split_outer_inverse_kernel<R, true, false>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, tw, M, Mp, B, chunks, inv_n, q_out, paired, tie_scale);
  else if (b1)
    hipLaunchKernelGGL((split_outer_inverse_kernel<R, false, true>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, tw, M, Mp, B, chunks, inv_n, q_out, paired, tie_scale);
  else
    hipLaunchKernelGGL((split_outer_inverse_kernel<R, false, false>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, tw, M, Mp, B, chunks, inv_n, q_out, paired, tie_scale);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int inner_twiddles(gacq_ctx* ctx, int M, const float2** out) {          // W_M^k, k < M
  return twiddle_cache(ctx, "WM_" + std::to_string(M), M, M, out);
}

}  // namespace

namespace gacq {

bool split_inner_fused_supported(int N) { return N == 61380 || N == 30690; }

#ifdef GACQ_PHASE_TIMING
extern "C" int gacq_debug_phase_cycles(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(gacq_phase_cycles), sizeof(unsigned long long) * 32) != hipSuccess) return GACQ_ERR_HIP;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gacq_phase_cycles), z, sizeof z) != hipSuccess) return GACQ_ERR_HIP;
  }
  return GACQ_OK;
}
#endif

// K2 + inner inverse transforms in one kernel (no Y round trip); Z gets the unnormalised, untwiddled inner IFFTs
int split_row_pitch(int N) {
  if (!split_inner_fused_supported(N)) return 0;
  return (N / 31 + 15) & ~15;                      // 1980 -> 1984, 990 -> 992 complex: rows of Z' start on 128-byte lines
}

int split_inner_correlate(gacq_ctx* ctx, const float2* X, const float2* C, const int* d_items, const int* d_fset, long g0, long ng,
                          int P, int F, int D, int B, int N, float2* Z, int Mp) {
  const int R = 31, M = N / R;
  if (Mp < M) Mp = M;
  const float2* twm;
  int rc = inner_twiddles(ctx, M, &twm);
  if (rc != GACQ_OK) return rc;
  // (epoch, item) rows touched by this pass, cut into chunks of pch per workgroup; >= ~2048 workgroups, <= 8 items each
  const long ep_first = g0 / D, ep_last = (g0 + ng - 1) / D;
  const long nep = ep_last - ep_first + 1;
  int pch = (int)std::max<long>(1, std::min<long>(8, nep * D * B * R / 2048));
  if (ctx->opt[GACQ_OPT_SPLIT_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_SPLIT_PCH];
  int teams = 1;      // measured (profiles/r02_stockham_teams_sweep.log): 2 or 4 rows per workgroup are 5-10 % slower despite 24 instead of 15 waves per CU
  if (ctx->opt[GACQ_OPT_SPLIT_TEAMS] >= 1) teams = (int)ctx->opt[GACQ_OPT_SPLIT_TEAMS];
  if (teams != 1 && teams != 2 && teams != 4) return set_error(ctx, GACQ_ERR_BAD_ARG, "split engine: teams per workgroup must be 1, 2 or 4");
  pch = (pch + teams - 1) / teams * teams;
  const int nblk_ep = (int)((nep + pch - 1) / pch);
  int dt = (teams == 1 && D >= 8) ? (M == 1980 ? 3 : 2) : 1;       // Doppler bins per workgroup (profiles/r02_stockham_doppler_tile_sweep.log)
  if (ctx->opt[GACQ_OPT_SPLIT_DT] >= 1) dt = (int)ctx->opt[GACQ_OPT_SPLIT_DT];
  if (dt < 1 || dt > 3) return set_error(ctx, GACQ_ERR_BAD_ARG, "split engine: Doppler bins per workgroup must be 1, 2 or 3");
  const int DG = (D + dt - 1) / dt;
  const dim3 grid((unsigned)((long)R * nblk_ep * DG * B));
  const size_t smem = sizeof(float2) * ((size_t)teams * M + (size_t)M);
  {
    // teams = 4 with M = 1980 asks for 79 KB of dynamic LDS: fine on gfx950 (160 KB), not on a 64 KB part -- say so instead of
    // failing inside hipFuncSetAttribute with a generic HIP error
    int max_lds = 0;
    if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) == hipSuccess && max_lds > 0 && smem > (size_t)max_lds)
      return set_error(ctx, GACQ_ERR_UNSUPPORTED, "split engine: %d team(s) of M=%d need %zu bytes of LDS per workgroup, the device allows %d",
                       teams, M, smem, max_lds);
  }
#define GACQ_LAUNCH_INNER(KERN, NT)                                                                                                 \
  do {                                                                                                                              \
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                   \
    hipLaunchKernelGGL(KERN, grid, dim3((NT) * teams), smem, ctx->stream, X, C, Z, d_items, d_fset, twm, g0, ng, ep_first, nblk_ep, pch, P, F, D, \
                       B, R, Mp);                                                                                                   \
  } while (0)
#define GACQ_LAUNCH_DT(R1_, R2_, NT_, TEAMS_)                                                                                        \
  do {                                                                                                                              \
    if (dt == 3) GACQ_LAUNCH_INNER((split_inner_corr_kernel<11, R1_, R2_, 1, NT_, TEAMS_, 3>), NT_);                                \
    else if (dt == 2) GACQ_LAUNCH_INNER((split_inner_corr_kernel<11, R1_, R2_, 1, NT_, TEAMS_, 2>), NT_);                           \
    else GACQ_LAUNCH_INNER((split_inner_corr_kernel<11, R1_, R2_, 1, NT_, TEAMS_, 1>), NT_);                                        \
  } while (0)
  if (teams != 1 && dt != 1) return set_error(ctx, GACQ_ERR_BAD_ARG, "split engine: teams > 1 and Doppler tiles > 1 are not combined");
  if (M == 1980) {
    if (teams == 4) GACQ_LAUNCH_INNER((split_inner_corr_kernel<11, 12, 15, 1, 192, 4, 1>), 192);
    else if (teams == 2) GACQ_LAUNCH_INNER((split_inner_corr_kernel<11, 12, 15, 1, 192, 2, 1>), 192);
    else GACQ_LAUNCH_DT(12, 15, 192, 1);
  } else if (M == 990) {
    if (teams == 4) GACQ_LAUNCH_INNER((split_inner_corr_kernel<11, 9, 10, 1, 128, 4, 1>), 128);
    else if (teams == 2) GACQ_LAUNCH_INNER((split_inner_corr_kernel<11, 9, 10, 1, 128, 2, 1>), 128);
    else GACQ_LAUNCH_DT(9, 10, 128, 1);
  } else {
    return set_error(ctx, GACQ_ERR_UNSUPPORTED, "fused inner transforms: M=%d not supported", M);
  }
#undef GACQ_LAUNCH_DT
#undef GACQ_LAUNCH_INNER
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int split_radix(int N) {
  if (N > 0 && N % 31 == 0 && smooth(N / 31) && N / 31 >= 64) return 31;
  if (N == 163840) return 40;
  if (N == 81920) return 20;
  if (N == 65536) return 16;
  if (N == 16384) return 4;
  return 0;
}

bool split_supported(int N) { return split_radix(N) != 0; }

int split_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, long rows, int n, int N, const double* d_freq, int FD, int B,
                const float2* tab, float2* X, bool mix, bool inner) {
  const int R = split_radix(N);
  if (!R) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "split engine: N=%d not supported", N);
  const int M = N / R;
  const float2* tw;
  int rc = base_twiddles(ctx, N, M, &tw);
  if (rc != GACQ_OK) return rc;
  switch (R) {
    case 31: rc = launch_forward<31>(ctx, x, nsamp, rows, n, M, d_freq, FD, B, tab, tw, X, mix); break;
    case 40: rc = launch_forward<40>(ctx, x, nsamp, rows, n, M, d_freq, FD, B, tab, tw, X, mix); break;
    case 20: rc = launch_forward<20>(ctx, x, nsamp, rows, n, M, d_freq, FD, B, tab, tw, X, mix); break;
    case 16: rc = launch_forward<16>(ctx, x, nsamp, rows, n, M, d_freq, FD, B, tab, tw, X, mix); break;
    default: rc = launch_forward<4>(ctx, x, nsamp, rows, n, M, d_freq, FD, B, tab, tw, X, mix); break;
  }
  if (rc != GACQ_OK) return rc;
  if (!inner) return GACQ_OK;
  return fft_exec(ctx, M, rows * R, false, X);            // inner transforms, rows contiguous
}

namespace {
template <int R>
void launch_dump(gacq_ctx* ctx, int n, int M, const double* d_freq, int* d_idx) {
  const int chunks = (M + kBlock - 1) / kBlock;
  hipLaunchKernelGGL((split_outer_forward_kernel<R, true, true>), dim3((unsigned)chunks), dim3(kBlock), 0, ctx->stream, (const float2*)nullptr,
                     (size_t)0, (float2*)d_idx, d_freq, (const float2*)nullptr, (const float2*)nullptr, n, M, 1, 1, chunks);
}
}  // namespace

int split_debug_nco(gacq_ctx* ctx, int N, int n, const double* d_freq, int* d_idx) {
  const int R = split_radix(N);
  if (!R) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "NCO index dump: split engine does not support N=%d", N);
  const int M = N / R;
  switch (R) {
    case 31: launch_dump<31>(ctx, n, M, d_freq, d_idx); break;
    case 40: launch_dump<40>(ctx, n, M, d_freq, d_idx); break;
    case 20: launch_dump<20>(ctx, n, M, d_freq, d_idx); break;
    case 16: launch_dump<16>(ctx, n, M, d_freq, d_idx); break;
    default: launch_dump<4>(ctx, n, M, d_freq, d_idx); break;
  }
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int split_inverse_reduce(gacq_ctx* ctx, float2* Y, RowRec* rows, long g0, long ng, int B, int N, float* q_out, float tie_scale, bool inner,
                         bool twiddle_only, int Mp) {
  const int R = split_radix(N);
  if (!R) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "split engine: N=%d not supported", N);
  const int M = N / R;
  if (Mp <= 0) Mp = M;
  // Y written by lds_inner_correlate_kernel (inner == false, M == 4096): rows in the lane-pair layout
  const int paired = (!inner && !twiddle_only && M == 4096) ? 1 : 0;
  if (Mp != M && !twiddle_only) return set_error(ctx, GACQ_ERR_BAD_ARG, "split engine: padded rows need the fused inner kernel");
  const float2* tw;
  int rc = base_twiddles(ctx, N, M, &tw);
  if (rc != GACQ_OK) return rc;
  const int chunks = (M + kBlock - 1) / kBlock;
  if (inner && !twiddle_only && (rc = fft_exec(ctx, M, ng * B * R, true, Y)) != GACQ_OK) return rc;
  if (twiddle_only) inner = true;              // Y holds untwiddled inner IFFTs: the outer kernel applies W_N^{-n2 k1}
  if ((rc = ensure(ctx, ctx->partial, sizeof(RowRec) * (size_t)ng * chunks)) != GACQ_OK) return rc;
  RowRec* partial = (RowRec*)ctx->partial.p;
  const float inv_n = 1.0f / (float)N;
  switch (R) {
    case 31: rc = launch_inverse<31>(ctx, Y, partial, tw, M, Mp, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
    case 40: rc = launch_inverse<40>(ctx, Y, partial, tw, M, Mp, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
    case 20: rc = launch_inverse<20>(ctx, Y, partial, tw, M, Mp, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
    case 16: rc = launch_inverse<16>(ctx, Y, partial, tw, M, Mp, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
    default: rc = launch_inverse<4>(ctx, Y, partial, tw, M, Mp, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
  }
  if (rc != GACQ_OK) return rc;
  hipLaunchKernelGGL(split_combine_kernel, dim3((unsigned)((ng + 127) / 128)), dim3(128), 0, ctx->stream, (const RowRec*)partial, rows,
                     g0, ng, chunks, tie_scale);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace gacq
