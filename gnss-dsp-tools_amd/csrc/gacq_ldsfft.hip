// LDS-resident FFT engine for gfx950: the whole length-N transform of one correlation row lives in
// one workgroup's registers + LDS, so the stage boundaries of the rocFFT pipeline (mix -> FFT,
// conj-mul -> IFFT -> |.| -> reduce) never touch HBM.
//
//   lds_forward_kernel    x window --(table NCO mix)--> FFT_N --> conj --> X[row]       (K1+F1 fused)
//   lds_correlate_kernel  for p in PRN chunk: q = sum_b | IFFT_N(C_p * X[e,f,d,b]) |/N ; (max, argmax, sum)
//                                                                                        (K2+F2+K3 fused)
//
// Decomposition (N = 4096 = 16*16*16, one wave64-friendly 256-thread workgroup, 16 points per lane):
//   n = n0 + 16 n1 + 256 n2,  k = k0 + 16 k1 + 256 k2
//   pass 1: lane (n0,n1) holds n2=0..15 -> DFT16 over n2 -> k0 ; twiddle W_N^{(n0+16 n1) k0}
//   pass 2: lane (n0,k0) holds n1=0..15 -> DFT16 over n1 -> k1 ; twiddle W_256^{n0 k1}
//   pass 3: lane (k0,k1) holds n0=0..15 -> DFT16 over n0 -> k2
// Global loads/stores are lane-contiguous in every pass-1 load and pass-3 store (index = lane + 256*j);
// the two LDS transposes are bank-conflict free: exchange 1 is lane-contiguous on both sides, exchange 2
// writes with row pitch 257 (odd) so that the 16 lanes of a ds_write_b64 group hit 16 distinct bank pairs.
#include "gacq_common.h"
#include "gacq_cplx.h"
#include "gacq_ldsutil.h"

#include <cmath>
#include <cstdlib>

using namespace gacq;

namespace {

constexpr int kR = 16;                 // points per lane
constexpr int kLdsN = 4096;            // supported length (this round)
constexpr int kPitch = 257;            // exchange-2 row pitch in complex elements
constexpr int kLdsElems = 16 * kPitch; // 4112 complex = 32.9 KB -> 4 workgroups per CU

// v[rev16(k)] *= w^k for k = 1..15, powers built with multiplication depth <= 4 from the table value.
__device__ __forceinline__ void apply_powers(v2 (&v)[kR], v2 w1) {
  const v2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
  v[rev16(1)] = cmul(v[rev16(1)], w1);    v[rev16(2)] = cmul(v[rev16(2)], w2);    v[rev16(3)] = cmul(v[rev16(3)], w3);
  const v2 w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
  v[rev16(4)] = cmul(v[rev16(4)], w4);    v[rev16(5)] = cmul(v[rev16(5)], w5);    v[rev16(6)] = cmul(v[rev16(6)], w6);
  v[rev16(7)] = cmul(v[rev16(7)], w7);
  const v2 w9 = cmul(w8, w1), w10 = cmul(w5, w5), w11 = cmul(w8, w3), w12 = cmul(w6, w6);
  v[rev16(8)] = cmul(v[rev16(8)], w8);    v[rev16(9)] = cmul(v[rev16(9)], w9);    v[rev16(10)] = cmul(v[rev16(10)], w10);
  v[rev16(11)] = cmul(v[rev16(11)], w11); v[rev16(12)] = cmul(v[rev16(12)], w12);
  const v2 w13 = cmul(w8, w5), w14 = cmul(w7, w7), w15 = cmul(w8, w7);
  v[rev16(13)] = cmul(v[rev16(13)], w13); v[rev16(14)] = cmul(v[rev16(14)], w14); v[rev16(15)] = cmul(v[rev16(15)], w15);
}

// w^1..w^15 into pw[0..14] (same product tree as apply_powers)
__device__ __forceinline__ void make_powers(v2 (&pw)[15], v2 w1) {
  pw[0] = w1;
  pw[1] = cmul(w1, w1);        pw[2] = cmul(pw[1], w1);      pw[3] = cmul(pw[1], pw[1]);
  pw[4] = cmul(pw[3], w1);     pw[5] = cmul(pw[2], pw[2]);   pw[6] = cmul(pw[3], pw[2]);   pw[7] = cmul(pw[3], pw[3]);
  pw[8] = cmul(pw[7], w1);     pw[9] = cmul(pw[4], pw[4]);   pw[10] = cmul(pw[7], pw[2]);  pw[11] = cmul(pw[5], pw[5]);
  pw[12] = cmul(pw[7], pw[4]); pw[13] = cmul(pw[6], pw[6]);  pw[14] = cmul(pw[7], pw[6]);
}
__device__ __forceinline__ void apply_table(v2 (&v)[kR], const v2 (&pw)[15]) {
#pragma unroll
  for (int k = 1; k < kR; k++) v[rev16(k)] = cmul(v[rev16(k)], pw[k - 1]);
}

// Length-4096 transform of the 16 values per lane. In: v[j] = x[t + 256 j]; out: v[rev16(k2)] = X[t + 256 k2].
// wa = W_4096^t, wb = W_256^(t & 15) (forward values; conjugated here when INV).
// PRE: bit 0 = the pass-1 powers (W_4096^t)^k, bit 1 = the pass-2 powers (W_256^(t&15))^k come precomputed in *pa / *pb (already
// conjugated for an inverse transform) instead of being rebuilt from wa / wb by 14 complex products per pass.
// bit 2 = the pass-2 powers are read from an LDS table (tb2[16 (k - 1)], tb2 already offset by the lane's class t & 15).
// bit 3 = rising wave priority through the row (F4K_PRIO above).
// Rising wave priority through the segments of a 4096-point row (PRE bit 3 of fft4096; the caller resets it to GACQ_F4K_P4 at the top
// of its row loop): after the exchange-1 writes | after the exchange-2 writes | after the exchange-2 reads have been issued.  The four
// waves of a SIMD belong to four independent workgroups; letting the one that is furthest into its row issue first keeps the workgroups
// out of phase, so one's LDS round trips and barriers fall under another's arithmetic.  Headline step (1024 epochs x 40 bins x 32 PRNs):
// 5.50-5.53 -> 5.30-5.34 ms, falling levels (3-2-1-0, what the 16384-point kernels use inside ONE workgroup) 5.56, levels raised only
// in the last segment 5.48 (profiles/r05_headline_kernel_wave_priority_sweep.log).  -1 = leave the priority alone.
#ifndef GACQ_F4K_P1
#define GACQ_F4K_P1 1
#define GACQ_F4K_P2 2
#define GACQ_F4K_P3 3
#define GACQ_F4K_P4 0
#endif
#ifndef GACQ_F4K_P0
#define GACQ_F4K_P0 -1      // at the first radix-16 pass (sweeps only)
#endif
#define F4K_PRIO(n) do { if ((n) >= 0) asm volatile("s_setprio %0" :: "n"(n) : "memory"); } while (0)
#ifndef GACQ_CORR_PRE
#define GACQ_CORR_PRE 12     // lds_correlate_kernel: 4 = pass-2 powers from an LDS table, 8 = rising priorities (both: -2.5 % at B = 10 / 80)
#endif
// (The same levels in lds_inner_correlate_kernel -- engine 4's writer, store-bound -- and in lds_correlate_kernel with B = 10 measured within
// the run-to-run noise, 0-2 %: not applied there.)
template <bool INV, int PRE = 0>
__device__ __forceinline__ void fft4096(v2 (&v)[kR], v2* lds, v2 wa, v2 wb, const v2 (*pa)[15] = nullptr, const v2 (*pb)[15] = nullptr,
                                        int t = -1, const v2* tb2 = nullptr) {
  if (t < 0) t = threadIdx.x;                       // lane index within the 256-lane group that owns this transform
  if (INV && !(PRE & 1)) wa.y = -wa.y;
  if (INV && !(PRE & 2)) wb.y = -wb.y;
  F4K_PRIO((PRE & 8) ? GACQ_F4K_P0 : -1);
  dft16<INV>(v);
  if (PRE & 1) apply_table(v, *pa); else apply_powers(v, wa);
  {  // exchange 1: (n0,n1;k0) -> (n0,k0;n1)
    const int wbase = (t & 15) + 256 * (t >> 4);
#pragma unroll
    for (int k = 0; k < kR; k++) lds[wbase + 16 * k] = v[rev16(k)];
    F4K_PRIO((PRE & 8) ? GACQ_F4K_P1 : -1);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kR; j++) v[j] = LDS_LD(lds[t + 256 * j]);
  }
  dft16<INV>(v);
  if (PRE & 4) {
#pragma unroll
    for (int k = 1; k < kR; k++) v[rev16(k)] = cmul(v[rev16(k)], LDS_LD(tb2[16 * (k - 1)]));
  } else if (PRE & 2) apply_table(v, *pb);
  else apply_powers(v, wb);
  __syncthreads();   // all exchange-1 reads done before the buffer is reused
  {  // exchange 2: (n0,k0;k1) -> (k0,k1;n0)
    const int wbase = (t >> 4) + kPitch * (t & 15);
#pragma unroll
    for (int k = 0; k < kR; k++) lds[wbase + 16 * k] = v[rev16(k)];
    F4K_PRIO((PRE & 8) ? GACQ_F4K_P2 : -1);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kR; j++) v[j] = LDS_LD(lds[t + kPitch * j]);
    F4K_PRIO((PRE & 8) ? GACQ_F4K_P3 : -1);
  }
  dft16<INV>(v);
}


// Lane-pair layout used for X and C_p inside this engine: natural index i = t + 256 j (t = lane, j = 0..15)
// is stored at (j>>1)*512 + 2 t + (j&1), so one lane's values for j = 2jp, 2jp+1 are 16 contiguous bytes and
// a wave reads/writes 1 KiB per instruction.  Rows are fetched with buffer loads: the row base lives in an
// SGPR resource, the lane offset (16 t) in one VGPR, the piece offset in an SGPR -- no VALU address math.

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float2* row) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, kLdsN * (int)sizeof(float2), 0x00020000);
}
// values j = 2jp (lo) and 2jp+1 (hi) of this lane
__device__ __forceinline__ void ld_pair(__amdgpu_buffer_rsrc_t r, unsigned lane_off, int jp, v2& a, v2& b) {
  const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, (unsigned)jp * 4096u, 0));
  a = q.xy;
  b = q.zw;
}

// ---- forward: one workgroup per (e, f, d, b) row -------------------------------------------------
// DUMP (test hook gacq_debug_nco_indices): the index expression below, on the same frequency table, is stored as int32 into X
// (reinterpreted) and the kernel returns; x is not read.
// PLAIN (code spectra, once per signal): row r of x is transformed as it is -- no carrier wipe-off, no conjugation -- so the code
// spectra come out of the same transform, in the same layout, as the forward spectra they are multiplied with (c = fft.fft(c),
// acquire-gps-l1.py:24), and building a signal needs no rocFFT plan.
template <bool DUMP, bool PLAIN = false>
__global__ __launch_bounds__(kBlock) void lds_forward_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                              float2* __restrict__ X, const double* __restrict__ freq,
                                                              const float2* __restrict__ nco_tab,
                                                              const float2* __restrict__ tw, int n, int FD, int B) {
  __shared__ v2 lds[kLdsElems];
  const int t = threadIdx.x;
  const unsigned row = blockIdx.x;        // ((e*FD + fd)*B + b); 32-bit index math (64-bit divisions are ~100 scalar ops)
  const int b = (int)(row % (unsigned)B);
  const unsigned r2 = row / (unsigned)B;
  const int fd = (int)(r2 % (unsigned)FD);
  const long e = r2 / (unsigned)FD;
  const double f = PLAIN ? 0.0 : freq[fd];
  const float2* src = x + e * epoch_stride + (size_t)b * n;
  v2 v[kR], w[kR];
#pragma unroll
  for (int j = 0; j < kR; j++) {
    const int i = t + 256 * j;
    if (PLAIN) { v[j] = ld2(src + i); continue; }
    // table NCO, index in fp64 exactly as numpy: floor((0 + f*i)*1024) mod 1024   (gnsstools/nco.py:6-9)
    const int k = nco_index(f, (int)i);
    if (DUMP) { reinterpret_cast<int*>(X)[row * (long)kLdsN + i] = k; continue; }
    v[j] = ld2(src + i);
    w[j] = ld2(nco_tab + k);
  }
  if (DUMP) return;
  const v2 twa = ld2(tw + t), twb = ld2(tw + 16 * (t & 15));
  if (!PLAIN) {
#pragma unroll
    for (int j = 0; j < kR; j++) v[j] = cmul(v[j], w[j]);
  }
  fft4096<false>(v, lds, twa, twb);
  float2* dst = X + row * (long)kLdsN;
  const float cs = PLAIN ? 1.f : -1.f;
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++) {
    const v2 a = v[rev16(2 * jp)], b = v[rev16(2 * jp + 1)];
    // store conj(FFT) in the lane-pair layout: np.conj(fft.fft(b))  acquire-gps-l1.py:32   (PLAIN: the transform itself)
    *reinterpret_cast<float4*>(dst + jp * 512 + 2 * t) = make_float4(a.x, cs * a.y, b.x, cs * b.y);
  }
}

// ======================================================================================================
// N = 16384 in ONE 1024-thread workgroup (B1I/B2I padded, GLONASS L1/L2), 16 points per lane, as
//   16 wave-private 1024-point transforms + one radix-16 pass across the waves:
//     n = t + 1024 j  (t = 64 w + l: wave w, lane l; j = register),   k = ka + 16 kk,  kk = k0 + 16 k1 + 256 k2
//   forward (decimation in frequency, natural order in, digit-permuted order out):
//     pass 0  DFT16 over j -> ka, twiddle W_N^{t ka};  exchange 0: lane t sends output ka to wave ka (the only step
//             that crosses waves: one workgroup barrier)
//     then wave ka transforms its 1024 values u[m], m = l + 64 j', WITHOUT any barrier -- a wave runs in lockstep and the
//     LDS executes one wave's accesses in order, so the two transposes inside a wave need no synchronisation:
//     pass 1  DFT16 over j' -> k0, twiddle W_1024^{l k0};   transpose 1: lane (k0, l_lo) collects l = l_lo + 4 l_hi
//     pass 2  DFT16 over l_hi -> k1, twiddle W_64^{l_lo k1}; transpose 2: lane mu = k0 + 16 k1_lo collects (k1_hi, l_lo)
//     pass 3  four DFT4 over l_lo -> k2
//     out: register r = k1_hi + 4 k2 of lane (w, mu) holds X[w + 16 mu + 1024 r]
//   inverse (decimation in time) is the transposed network with conjugated twiddles: it takes exactly that order in and
//   leaves y[t + 1024 j] in register rev16(j) of lane t.  Spectra (X, C_p) therefore live in memory in the order the
//   forward transform produces them ("physical lane-pair layout": element (t, r) at (r >> 1) * 2048 + 2 t + (r & 1)), the
//   pointwise product needs no order at all, and no reordering pass exists anywhere.
// A correlation row costs two workgroup barriers (around the one cross-wave exchange) instead of the seven of the
// 4 x 4096 decomposition of rounds 1-2, and the 16 waves of the CU drift apart everywhere else.
// LDS: 16 regions of 1056 complex (1024 + the padding of the pitch-66 / pitch-65 transposes) = 132 KB + 256 B of reduction
// scratch -> one workgroup (16 waves, 4 per SIMD, <= 128 VGPRs) per CU.  Every LDS access below is (per-lane base) +
// (compile-time offset) and bank-conflict free (checked per 16-lane store group / 32-lane load group).
constexpr int kBig = 16384;
constexpr int kBigThreads = 1024;
constexpr int kRegion = 1056;                               // complex elements per wave region
constexpr int kBigScratch = 16 * kRegion * (int)sizeof(v2); // byte offset of the cross-wave reduction scratch
constexpr int kBigLdsBytes = kBigScratch + 256;

// Phase timing of lds16k_correlate_kernel (diagnostic builds only, -DGACQ_PHASE_TIMING16; tools/phase_timing16.py): lane 0 of every
// wave accumulates the shader-clock cycles between marks into gacq_phase16[wave][phase] (read back with gacq_debug_phase16).
// Never defined in the product build.
#ifdef GACQ_PHASE_TIMING16
__device__ unsigned long long gacq_phase16[16 * 8];
#define GACQ_MARK16(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); acc16_[i] += now_ - mark16_; mark16_ = now_; } while (0)
#else
#define GACQ_MARK16(i) do { } while (0)
#endif

// Progress-based wave priority.  The four waves that share a SIMD do the same work between two workgroup barriers: VALU
// segments separated by LDS round trips.  Left to the default oldest-first arbitration, the two oldest waves ping-pong
// through all their segments (an LDS round trip is longer than a segment, so the VALU idles in between) and then wait at the
// barrier while the two youngest do the same.  A wave that lowers its own priority at the end of every segment -- right
// after issuing the LDS accesses that end it -- hands the VALU to the waves that are behind: the four waves take turns
// segment by segment and every round trip is covered by the three other waves' arithmetic.
#define GACQ_SETPRIO_(n) asm volatile("s_setprio " #n ::: "memory")
#define GACQ_SETPRIO(n) GACQ_SETPRIO_(n)
// Levels of the four segments between two barrier pairs of the inverse transform (last radix-16 pass + magnitudes | C * x +
// radix-4 | radix-16 | radix-16): measured on B1I, 63 items x 200 bins x 10 blocks (profiles/r03_16k_priority_sweep.log):
// none 3.50 ms, 3-2-1-0 3.06-3.09, 0-1-2-3 3.21, 2-3-1-0 2.98-2.99.
// forward + inverse in one kernel (lds16k_fused_kernel): after barrier A | after the sample loads are issued | after the forward
// exchange | after transpose 1 | after transpose 2 (then GACQ_P3 / GACQ_P4 inside the inverse transform)
#ifndef GACQ_QA
#define GACQ_QA 3
#define GACQ_QX 3
#define GACQ_QF 3
#define GACQ_QT1 2
#define GACQ_QT2 2
#endif
#ifndef GACQ_P1
#define GACQ_P1 2
#define GACQ_P2 3
#define GACQ_P3 1
#define GACQ_P4 0
#endif

// v[k] *= w^k, k = 1..15, registers in natural order (the DIT passes twiddle their inputs); same product tree as apply_powers
__device__ __forceinline__ void apply_powers_nat(v2 (&v)[kR], v2 w1) {
  const v2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
  v[1] = cmul(v[1], w1);    v[2] = cmul(v[2], w2);    v[3] = cmul(v[3], w3);
  const v2 w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
  v[4] = cmul(v[4], w4);    v[5] = cmul(v[5], w5);    v[6] = cmul(v[6], w6);    v[7] = cmul(v[7], w7);
  const v2 w9 = cmul(w8, w1), w10 = cmul(w5, w5), w11 = cmul(w8, w3), w12 = cmul(w6, w6);
  v[8] = cmul(v[8], w8);    v[9] = cmul(v[9], w9);    v[10] = cmul(v[10], w10); v[11] = cmul(v[11], w11);
  v[12] = cmul(v[12], w12);
  const v2 w13 = cmul(w8, w5), w14 = cmul(w7, w7), w15 = cmul(w8, w7);
  v[13] = cmul(v[13], w13); v[14] = cmul(v[14], w14); v[15] = cmul(v[15], w15);
}

// Per-lane twiddle bases of the three twiddled passes: W_N^t, W_1024^l = W_N^{16 l}, W_64^{l >> 4} = W_N^{256 (l >> 4)};
// twn holds W_16384^m for m < 1024.  The powers are rebuilt per pass (14 complex products): tables of the two wave-private
// passes in LDS (8.5 KB, one ds_read_b64 per product) were measured -- B1I 3.04 -> 3.51 ms: the LDS pipe, already carrying
// three exchanges per row, is as loaded as the VALU (profiles/r03_16k_*).
struct Tw16k { v2 w0, w1, w2; };
__device__ __forceinline__ Tw16k tw16k_load(const float2* __restrict__ twn, bool conj) {
  const int t = threadIdx.x, l = t & 63;
  Tw16k k;
  k.w0 = ld2(twn + t);
  k.w1 = ld2(twn + 16 * l);
  k.w2 = ld2(twn + 256 * (l >> 4));
  if (conj) { k.w0.y = -k.w0.y; k.w1.y = -k.w1.y; k.w2.y = -k.w2.y; }
  return k;
}

// Forward transform.  In: v[j] = x[t + 1024 j].  Out: v[r] = X[(t >> 6) + 16 (t & 63) + 1024 r].
// The caller guarantees that no wave still uses its region when the exchange-0 stores start (they go to every region).
__device__ __forceinline__ void fft16k_fwd(v2 (&v)[kR], v2* lds, const Tw16k& tw) {
  const int t = threadIdx.x, l = t & 63;
  v2* reg = lds + (t >> 6) * kRegion;
  dft16<false>(v);
  apply_powers(v, tw.w0);
#pragma unroll
  for (int ka = 0; ka < kR; ka++) LDS_ST1(lds[ka * kRegion + t], v[rev16(ka)]);           // exchange 0: output ka -> wave ka
  lds_barrier();
  GACQ_SETPRIO(GACQ_QF);
#pragma unroll
  for (int j = 0; j < kR; j++) v[j] = LDS_LD(reg[l + 64 * j]);
  dft16<false>(v);
  apply_powers(v, tw.w1);
#pragma unroll
  for (int k0 = 0; k0 < kR; k0++) LDS_ST1(reg[66 * k0 + l], v[rev16(k0)]);                // transpose 1, element (k0, l) at 66 k0 + l
#pragma unroll
  for (int lh = 0; lh < kR; lh++) v[lh] = LDS_LD(reg[66 * (l & 15) + (l >> 4) + 4 * lh]);  // lane (k0 = l & 15, l_lo = l >> 4)
  GACQ_SETPRIO(GACQ_QT1);
  dft16<false>(v);
  apply_powers(v, tw.w2);
#pragma unroll
  for (int k1 = 0; k1 < kR; k1++) LDS_ST1(reg[256 * (l >> 4) + (l & 15) + 16 * k1], v[rev16(k1)]);   // transpose 2, (k0, l_lo, k1) at 256 l_lo + 16 k1 + k0
#pragma unroll
  for (int lo = 0; lo < 4; lo++) {
#pragma unroll
    for (int kh = 0; kh < 4; kh++) v[kh + 4 * lo] = LDS_LD(reg[l + 256 * lo + 64 * kh]);     // lane mu = k0 + 16 k1_lo, k1 = k1_lo + 4 kh
  }
  GACQ_SETPRIO(GACQ_QT2);
#pragma unroll
  for (int kh = 0; kh < 4; kh++) dft4<false, false>(v[kh], v[kh + 4], v[kh + 8], v[kh + 12]);   // over l_lo -> k2 at v[kh + 4 k2]
}

// Inverse transform, wave-private part.  In: v[r] = Y[(t >> 6) + 16 (t & 63) + 1024 r]; on return the wave's region holds
// z[l + 64 j'] (its 1024-point inverse transform), ready for the cross-wave exchange.  tw: conjugated bases.
__device__ __forceinline__ void ifft16k_private(v2 (&v)[kR], v2* reg, const Tw16k& tw) {
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int kh = 0; kh < 4; kh++) dft4<true, false>(v[kh], v[kh + 4], v[kh + 8], v[kh + 12]);    // over k2 -> l_lo at v[kh + 4 l_lo]
#pragma unroll
  for (int lo = 0; lo < 4; lo++) {
#pragma unroll
    for (int kh = 0; kh < 4; kh++) LDS_ST1(reg[64 * (l >> 4) + (l & 15) + 256 * kh + 16 * lo], v[kh + 4 * lo]);   // (k0, l_lo, k1) at 64 k1 + 16 l_lo + k0
  }
#pragma unroll
  for (int k1 = 0; k1 < kR; k1++) v[k1] = LDS_LD(reg[l + 64 * k1]);                         // lane (k0 = l & 15, l_lo = l >> 4)
  GACQ_SETPRIO(GACQ_P3);
  apply_powers_nat(v, tw.w2);
  dft16<true>(v);                                                                   // over k1 -> l_hi
#pragma unroll
  for (int lh = 0; lh < kR; lh++) LDS_ST1(reg[65 * (l & 15) + (l >> 4) + 4 * lh], v[rev16(lh)]);   // element (k0, l = l_lo + 4 l_hi) at 65 k0 + l
#pragma unroll
  for (int k0 = 0; k0 < kR; k0++) v[k0] = LDS_LD(reg[l + 65 * k0]);
  GACQ_SETPRIO(GACQ_P4);
  apply_powers_nat(v, tw.w1);
  dft16<true>(v);                                                                   // over k0 -> j'
#pragma unroll
  for (int j = 0; j < kR; j++) LDS_ST1(reg[l + 64 * j], v[rev16(j)]);
}
// cross-wave part: gather the 16 partial transforms of n = t (mod 1024) ...
__device__ __forceinline__ void ifft16k_gather(v2 (&v)[kR], const v2* lds) {
  const int t = threadIdx.x;
#pragma unroll
  for (int ka = 0; ka < kR; ka++) v[ka] = LDS_LD(lds[ka * kRegion + t]);
}
// ... and combine them: v[rev16(j)] = N y[t + 1024 j]
__device__ __forceinline__ void ifft16k_final(v2 (&v)[kR], const Tw16k& tw) {
  apply_powers_nat(v, tw.w0);
  dft16<true>(v);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t big_rsrc(const float2* row) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, kBig * (int)sizeof(float2), 0x00020000);
}
// elements (t, 2 jp) and (t, 2 jp + 1) of a row in the physical lane-pair layout
__device__ __forceinline__ void ld_pair_big(__amdgpu_buffer_rsrc_t r, unsigned lane_off, int jp, v2& a, v2& b) {
  const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, (unsigned)jp * 16384u, 0));
  a = q.xy;
  b = q.zw;
}

// LDS-DMA of one spectrum row into the wave's own region: 8 x 1 KiB, lane l's 16 bytes of piece jp land at
// region + 1024 jp + 16 l -- no VGPRs are tied up while the row is in flight.
__device__ __forceinline__ void dma_row(const float2* __restrict__ row, v2* reg) {
  const char* src = reinterpret_cast<const char*>(row) + (size_t)threadIdx.x * 16;
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++)
    __builtin_amdgcn_global_load_lds((gptr_t)(src + jp * 16384), (lptr_t)(reinterpret_cast<char*>(reg) + jp * 1024), 16, 0, 0);
}
__device__ __forceinline__ void dma_wait_read(v2 (&x)[kR], const v2* reg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const f4* p = reinterpret_cast<const f4*>(reg) + (threadIdx.x & 63);
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++) { const f4 q = p[jp * 64]; x[2 * jp] = q.xy; x[2 * jp + 1] = q.zw; }
  GACQ_SETPRIO(GACQ_P2);
}

// cross-wave (max, first argmax, sum) of one item through the scratch words behind the regions; thread 0 writes the record
__device__ __forceinline__ void big_reduce_store(char* smem, float peak, unsigned widx, float wsum, float tie_scale, RowRec* dst) {
  float* s_peak = reinterpret_cast<float*>(smem + kBigScratch);
  int* s_idx = reinterpret_cast<int*>(smem + kBigScratch + 64);
  double* s_sum = reinterpret_cast<double*>(smem + kBigScratch + 128);
  const int t = threadIdx.x;
  if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = (int)widx; s_sum[t >> 6] = (double)wsum; }
  lds_barrier();
  if (t == 0) {
    RowRec r;
    combine_tagged(kBigThreads / 64, [&](int w) { return s_peak[w]; }, [&](int w) { return s_idx[w]; }, tie_scale, r.peak, r.idx);
    double bs = s_sum[0];
    for (int w = 1; w < kBigThreads / 64; w++) bs += s_sum[w];
    r.sum = bs;
    *dst = r;
  }
}

// forward: one workgroup per (e, f, d, b) row; output conj(FFT) in the physical lane-pair layout, 1 KiB per wave and store
template <bool DUMP, bool PLAIN = false>      // PLAIN: code spectra, see lds_forward_kernel
__global__ __launch_bounds__(kBigThreads) void lds16k_forward_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                                      float2* __restrict__ X, const double* __restrict__ freq,
                                                                      const float2* __restrict__ nco_tab,
                                                                      const float2* __restrict__ twn, int n, int FD, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2* lds = reinterpret_cast<v2*>(smem);
  const int t = threadIdx.x;
  const unsigned row = blockIdx.x;
  const int b = (int)(row % (unsigned)B);
  const unsigned r2 = row / (unsigned)B;
  const int fd = (int)(r2 % (unsigned)FD);
  const long e = r2 / (unsigned)FD;
  const double f = PLAIN ? 0.0 : freq[fd];
  const float2* src = x + e * epoch_stride + (size_t)b * n;
  v2 v[kR], w[kR];
#pragma unroll
  for (int j = 0; j < kR; j++) {
    const int i = t + 1024 * j;
    if (PLAIN) { v[j] = ld2(src + i); continue; }
    const int k = nco_index(f, (int)i);   // gnsstools/nco.py:6-9
    if (DUMP) { reinterpret_cast<int*>(X)[row * (long)kBig + i] = k; continue; }
    v[j] = ld2(src + i);
    w[j] = ld2(nco_tab + k);
  }
  if (DUMP) return;
  const Tw16k tw = tw16k_load(twn, false);
  if (!PLAIN) {
#pragma unroll
    for (int j = 0; j < kR; j++) v[j] = cmul(v[j], w[j]);
  }
  fft16k_fwd(v, lds, tw);
  float2* dst = X + row * (long)kBig;
  const float cs = PLAIN ? 1.f : -1.f;
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++) {
    const v2 a = v[2 * jp], c = v[2 * jp + 1];
    // np.conj(fft.fft(b))  acquire-beidou-b1i.py:32   (PLAIN: the transform itself)
    *reinterpret_cast<float4*>(dst + jp * 2048 + 2 * t) = make_float4(a.x, cs * a.y, c.x, cs * c.y);
  }
}

// correlate: workgroup = (chunk of items, group of `ugroup` (epoch, Doppler) units of one XCD); per (item, unit):
// sum_b |IFFT(C_p * X_b)|/N -> (max, argmax, sum).
// Item-major since round 4: the item's code spectrum is loaded ONCE and stays in registers for every unit of the group and all B
// blocks of each, so a spectrum row crosses into the CU once per (item, group) instead of once per (item, unit) -- B1I (63 spectra =
// 8 MB against 4 MB of L2 per XCD, 200 units) fetched 1.6 GB of code spectra per launch in the unit-major order of round 3, 7.7 x the
// kernel's compulsory bytes.  The workgroups resident on an XCD (consecutive in launch order: same group, different items) walk the
// group's units side by side, so the forward spectrum of the (unit, block) they are all working on is fetched from HBM once and
// served from that XCD's L2 to the rest.
// The forward spectrum of the NEXT row is fetched by LDS-DMA into the wave's own region as soon as the cross-wave exchange of the
// current row has been read out, i.e. under the last radix-16 pass and the magnitudes -- the register file (16 + 16 complex + 16
// accumulators of 128 VGPRs) has no room for a prefetch, the LDS is idle exactly then.  Blocks -> (group, chunk): see lds_correlate().
template <bool QDUMP>
__global__ __launch_bounds__(kBigThreads) void lds16k_correlate_kernel(const float2* __restrict__ X, const float2* __restrict__ C,
                                                                        const int* __restrict__ items, const int* __restrict__ fset,
                                                                        const float2* __restrict__ twn, RowRec* __restrict__ rows,
                                                                        int E, int P, int F, int D, int B, int pch, int nchunk, int ugroup,
                                                                        float tie_scale, float* __restrict__ q_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2* lds = reinterpret_cast<v2*>(smem);
  const int t = threadIdx.x;
  // placement: workgroup b runs on XCD b % 8; XCD x owns the units u = x (mod 8), in groups of `ugroup` consecutive owned units
  const int xcd = blockIdx.x & 7;
  const unsigned j = blockIdx.x >> 3;
  const unsigned grp = j / (unsigned)nchunk;
  const int p0 = (int)(j % (unsigned)nchunk) * pch;
  const int p1 = min(P, p0 + pch);
  const unsigned U = (unsigned)E * (unsigned)D;
  const unsigned u0 = grp * (unsigned)ugroup * 8u + (unsigned)xcd;           // first unit of the group; the i-th is u0 + 8 i
  if (u0 >= U) return;
  const int nu = (int)min((unsigned)ugroup, (U - u0 + 7u) / 8u);
  v2* reg = lds + (t >> 6) * kRegion;
  const Tw16k tw = tw16k_load(twn, true);
  const unsigned lane_off = (unsigned)t * 16u;
  const float inv_n = 1.0f / (float)kBig;
  // first forward-spectrum row of (item p, i-th unit of the group)
  auto unit_row = [&](int p, int i) -> const float2* {
    const unsigned u = u0 + 8u * (unsigned)i;
    const long e = u / (unsigned)D;
    const int d = (int)(u % (unsigned)D);
    return X + (((e * F + fset[p]) * D + d) * (long)B) * kBig;
  };
  const float2* xrow = unit_row(p0, 0);
  dma_row(xrow, reg);
#ifdef GACQ_PHASE_TIMING16
  unsigned long long acc16_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long mark16_ = __builtin_readcyclecounter();
#endif
  for (int p = p0; p < p1; p++) {
    const __amdgpu_buffer_rsrc_t cres = big_rsrc(C + (long)items[p] * kBig);
    v2 c[kR];
#pragma unroll
    for (int jp = 0; jp < kR / 2; jp++) ld_pair_big(cres, lane_off, jp, c[2 * jp], c[2 * jp + 1]);
    for (int i = 0; i < nu; i++) {
      const unsigned u = u0 + 8u * (unsigned)i;
      const long e = u / (unsigned)D;
      const int d = (int)(u % (unsigned)D);
      // the row after this unit's last block: the next unit of the group, else the next item's first unit, else nothing
      const float2* xnext_unit = (i + 1 < nu) ? unit_row(p, i + 1) : ((p + 1 < p1) ? unit_row(p + 1, 0) : nullptr);
      float q[kR];
#pragma unroll
      for (int k = 0; k < kR; k++) q[k] = 0.f;
      for (int b = 0; b < B; b++) {
        v2 v[kR];
        GACQ_MARK16(0);                                    // previous row's tail (reduction, code-spectrum loads)
        dma_wait_read(v, reg);
#pragma unroll
        for (int jj = 0; jj < kR; jj++) v[jj] = cmul(c[jj], v[jj]);
        GACQ_MARK16(1);
        ifft16k_private(v, reg, tw);
        GACQ_MARK16(2);
        lds_barrier();
        GACQ_MARK16(3);
        ifft16k_gather(v, lds);
        lds_barrier();                                     // every wave has read this region: it may be overwritten
        GACQ_SETPRIO(GACQ_P1);
        GACQ_MARK16(4);
        const float2* nx = (b + 1 < B) ? xrow + (long)(b + 1) * kBig : xnext_unit;
        if (nx) dma_row(nx, reg);
        GACQ_MARK16(5);
        ifft16k_final(v, tw);
#pragma unroll
        for (int k = 0; k < kR; k++) {
          const v2 r = v[rev16(k)];
          q[k] += __builtin_amdgcn_sqrtf(norm2(r)) * inv_n;
        }
        GACQ_MARK16(6);
      }
      xrow = xnext_unit;
      if (QDUMP) {                                         // gacq_debug_row: the accumulated magnitude row itself (one row per launch)
#pragma unroll
        for (int k = 0; k < kR; k++) q_out[t + 1024 * k] = q[k];
      }
      float sum_f = q[0];
#pragma unroll
      for (int k = 1; k < kR; k++) sum_f += q[k];
      // lane l of wave w holds lags 64 w + l + 1024 k: first maximum as in lds_correlate_kernel
      float peak;
      unsigned widx;
      wave_first_max(q, (unsigned)__builtin_amdgcn_readfirstlane(t & ~63), 1u, 1024u, tie_scale, peak, widx);
      big_reduce_store(smem, peak, widx, wave_add_f32(sum_f), tie_scale, rows + (e * P + p) * (long)D + d);
    }
  }
#ifdef GACQ_PHASE_TIMING16
  if ((t & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 8; i++) if (acc16_[i]) atomicAdd(&gacq_phase16[(t >> 6) * 8 + i], acc16_[i]);
  }
#endif
}

// Fused search for item lists in which every item has its own carrier (F == P: the GLONASS FDMA channels, or a single
// item): the forward spectrum of (e, f, d, b) is used by exactly one item, so writing it to HBM and reading it back
// (8 N bytes each way per row) buys nothing.  Workgroup = (epoch, Doppler bin, item); per block b: mix + forward transform --
// whose output order is the inverse transform's input order, so the spectrum stays in registers -- conj * C_p, inverse
// transform, |.| accumulated in registers.  Three workgroup barriers per block.  Same arithmetic in the same order as
// lds16k_forward_kernel + lds16k_correlate_kernel.
template <bool DUMP>
__global__ __launch_bounds__(kBigThreads) void lds16k_fused_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                                    const float2* __restrict__ C, const int* __restrict__ items,
                                                                    const int* __restrict__ fset, const double* __restrict__ freq,
                                                                    const float2* __restrict__ nco_tab,
                                                                    const float2* __restrict__ twn, RowRec* __restrict__ rows, int n,
                                                                    int P, int D, int B, float tie_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2* lds = reinterpret_cast<v2*>(smem);
  const int t = threadIdx.x;
  unsigned blk = blockIdx.x;                          // ((e*D + d)*P + p): the P items of one (e, d) run side by side
  const int p = (int)(blk % (unsigned)P);
  blk /= (unsigned)P;
  const int d = (int)(blk % (unsigned)D);
  const long e = blk / (unsigned)D;
  const double f = freq[(long)fset[p] * D + d];
  const __amdgpu_buffer_rsrc_t cres = big_rsrc(C + (long)items[p] * kBig);
  v2* reg = lds + (t >> 6) * kRegion;
  const unsigned lane_off = (unsigned)t * 16u;
  const float inv_n = 1.0f / (float)kBig;
  float q[kR];
#pragma unroll
  for (int k = 0; k < kR; k++) q[k] = 0.f;
  for (int b = 0; b < B; b++) {
    const float2* src = x + e * epoch_stride + (size_t)b * n;
    v2 v[kR], w[kR];
#pragma unroll
    for (int j = 0; j < kR; j++) {
      const int i = t + 1024 * j;
      const int k = nco_index(f, i);                  // gnsstools/nco.py:6-9
      if (DUMP) { reinterpret_cast<int*>(rows)[(long)blockIdx.x * kBig + i] = k; continue; }
      v[j] = ld2(src + i);
      w[j] = ld2(nco_tab + k);
    }
    if (DUMP) return;
    GACQ_SETPRIO(GACQ_QX);
#pragma unroll
    for (int j = 0; j < kR; j++) v[j] = cmul(v[j], w[j]);
    // the twiddle bases are re-read per block (three 8-byte loads, L1 hits): kept live across the block loop, the two sets cost
    // 12 VGPRs the 128-register budget does not have
    const float2* twp = twn;
    asm volatile("" : "+s"(twp));
    fft16k_fwd(v, lds, tw16k_load(twp, false));
    v2 c[kR];
#pragma unroll
    for (int jp = 0; jp < kR / 2; jp++) ld_pair_big(cres, lane_off, jp, c[2 * jp], c[2 * jp + 1]);
    const Tw16k twi = tw16k_load(twp, true);
#pragma unroll
    for (int j = 0; j < kR; j++) v[j] = cmul(c[j], v2{v[j].x, -v[j].y});      // C_p * np.conj(fft.fft(b))
    ifft16k_private(v, reg, twi);
    lds_barrier();
    ifft16k_gather(v, lds);
    lds_barrier();                                    // every wave has read this region: the next block's exchange 0 may overwrite it
    GACQ_SETPRIO(GACQ_QA);
    ifft16k_final(v, twi);
#pragma unroll
    for (int k = 0; k < kR; k++) {
      const v2 r = v[rev16(k)];
      q[k] += __builtin_amdgcn_sqrtf(norm2(r)) * inv_n;
    }
  }
  float sum_f = q[0];
#pragma unroll
  for (int k = 1; k < kR; k++) sum_f += q[k];
  float peak;
  unsigned widx;
  wave_first_max(q, (unsigned)__builtin_amdgcn_readfirstlane(t & ~63), 1u, 1024u, tie_scale, peak, widx);      // lane l of wave w: lags 64 w + l + 1024 k
  big_reduce_store(smem, peak, widx, wave_add_f32(sum_f), tie_scale, rows + (e * P + p) * (long)D + d);
}

// ---- inner transforms of the split engine (N = R * 4096, gacq_split.hip) -----------------------------
// In-place forward FFT of natural-order rows (the outer stage's A[k1][n2]); output in the lane-pair layout,
// conjugated for sample spectra (CONJ) and plain for code spectra.
template <bool CONJ>
__global__ __launch_bounds__(kBlock) void lds_inner_forward_kernel(float2* __restrict__ rows, const float2* __restrict__ tw) {
  __shared__ v2 lds[kLdsElems];
  const int t = threadIdx.x;
  float2* row = rows + (long)blockIdx.x * kLdsN;
  v2 v[kR];
#pragma unroll
  for (int j = 0; j < kR; j++) v[j] = ld2(row + t + 256 * j);
  const v2 twa = ld2(tw + t), twb = ld2(tw + 16 * (t & 15));
  fft4096<false>(v, lds, twa, twb);
  // every lane has read its 16 inputs before the first exchange barrier, so the row can be overwritten in place
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++) {
    const v2 a = v[rev16(2 * jp)], b = v[rev16(2 * jp + 1)];
    *reinterpret_cast<float4*>(row + jp * 512 + 2 * t) = CONJ ? make_float4(a.x, -a.y, b.x, -b.y) : make_float4(a.x, a.y, b.x, b.y);
  }
}

// Z'[(g,b)][k1][n2] = W_N^{-n2 k1} * IFFT_4096( C_p[k1][.] * X[e,f,d,b][k1][.] )[n2]      (K2 + inner inverse + twiddle)
// Workgroup = (k1, chunk of pch consecutive (epoch, item) pairs, Doppler bin, block).  The X row and the W_N twiddle powers
// depend only on the workgroup, so they stay in registers while the items change (X is reloaded only when its row pointer
// changes: epoch or frequency-set boundary inside the chunk).  Consecutive workgroups share (k1, item chunk), so those pch
// code-spectrum rows are hot in every XCD's L2.  [g0, g0+ng) = (e,p,d) groups whose Z rows exist in this workspace pass.
// twn holds W_N^m for m < 256 R.
__global__ __launch_bounds__(kBlock, 4) void lds_inner_correlate_kernel(const float2* __restrict__ X, const float2* __restrict__ C,
                                                                        const int* __restrict__ items, const int* __restrict__ fset,
                                                                        const float2* __restrict__ tw, const float2* __restrict__ twn,
                                                                        float2* __restrict__ Z, long g0, long ng, long ep_first,
                                                                        int nblk_ep, int pch, int P, int F, int D, int B, int R) {
  __shared__ v2 lds[kLdsElems];
  const int t = threadIdx.x;
  unsigned blk = blockIdx.x;                   // 32-bit index math (64-bit divisions are ~100 scalar ops each)
  const int b = (int)(blk % (unsigned)B);
  blk /= (unsigned)B;
  const int d = (int)(blk % (unsigned)D);
  blk /= (unsigned)D;
  const unsigned epc = blk % (unsigned)nblk_ep;
  const int k1 = (int)(blk / (unsigned)nblk_ep);
  const long ep0 = ep_first + (long)epc * pch;
  long e = ep0 / P;
  int p = (int)(ep0 - e * P) - 1;
  const unsigned lane_off = (unsigned)t * 16u;
  const v2 twa = ld2(tw + t), twb = ld2(tw + 16 * (t & 15));
  // lane holds n2 = t + 256 k: W_N^{-k1 (t + 256 k)} = conj(W_N^{k1 t}) * conj(W_N^{256 k1})^k
  v2 base = ld2(twn + k1 * t), step = ld2(twn + 256 * k1);
  base.y = -base.y;
  step.y = -step.y;
  // the 16 output twiddles of this lane depend on (k1, t) only: built once per workgroup (multiplication depth <= 5), one product
  // per output afterwards instead of up to three (base, step^(k&3), step^(k&~3))
  v2 tk[kR];
  {
    TwPow tp;
    tp.init<15>(step);
    tk[0] = base;
#pragma unroll
    for (int k = 1; k < kR; k++) tk[k] = tp.apply(base, k);
  }
  const float2* have = nullptr;
  v2 xr[kR];
  for (int i = 0; i < pch; i++) {
    if (++p == P) { p = 0; e++; }
    const long g = (ep0 + i) * D + d;
    if (g < g0 || g >= g0 + ng) continue;      // uniform over the workgroup
    const float2* xrow = X + ((((e * F + fset[p]) * D + d) * (long)B + b) * R + k1) * kLdsN;
    const __amdgpu_buffer_rsrc_t cres = row_rsrc(C + ((long)items[p] * R + k1) * kLdsN);
    v2 v[kR];
#pragma unroll
    for (int jp = 0; jp < kR / 2; jp++) ld_pair(cres, lane_off, jp, v[2 * jp], v[2 * jp + 1]);      // loads first, asm afterwards
    if (xrow != have) {
      const __amdgpu_buffer_rsrc_t xres = row_rsrc(xrow);
#pragma unroll
      for (int jp = 0; jp < kR / 2; jp++) ld_pair(xres, lane_off, jp, xr[2 * jp], xr[2 * jp + 1]);
      have = xrow;
    }
#pragma unroll
    for (int jj = 0; jj < kR; jj++) v[jj] = cmul(v[jj], xr[jj]);
    fft4096<true>(v, lds, twa, twb);
    // the row goes out in the lane-pair layout (n2 = t + 256 k at (k >> 1) * 512 + 2 t + (k & 1)): one 16-byte store per two
    // outputs, 1 KiB per wave and instruction; split_outer_inverse_kernel (paired) undoes the permutation in its index arithmetic
    float2* dst = Z + (((g - g0) * B + b) * R + k1) * (long)kLdsN;
    if (k1 == 0) {
#pragma unroll
      for (int kp = 0; kp < kR / 2; kp++) {
        const v2 a = v[rev16(2 * kp)], c = v[rev16(2 * kp + 1)];
        *reinterpret_cast<float4*>(dst + kp * 512 + 2 * t) = make_float4(a.x, a.y, c.x, c.y);
      }
    } else {
#pragma unroll
      for (int kp = 0; kp < kR / 2; kp++) {
        const v2 a = cmul(v[rev16(2 * kp)], tk[2 * kp]), c = cmul(v[rev16(2 * kp + 1)], tk[2 * kp + 1]);
        *reinterpret_cast<float4*>(dst + kp * 512 + 2 * t) = make_float4(a.x, a.y, c.x, c.y);
      }
    }
    __syncthreads();                           // exchange-2 reads done before the next item's exchange-1 writes
  }
}

// ---- correlate: workgroup = (epoch, doppler, chunk of items); epochs pinned to XCDs ---------------
// Template knobs (register budget vs occupancy, see DESIGN.md "LDS engine tuning"):
//   MINW    __launch_bounds__ waves per SIMD (4 -> <=128 VGPRs -> 4 workgroups/CU, the LDS limit)
//   B1      single block (B == 1): no q[] accumulator, magnitudes are reduced as they are produced
//   CACHEX  B1 only: keep the forward spectrum X[e,d] in registers across the item loop (halves L2 reads)
//   OPAQUE  recompute the twiddle powers w^1..w^15 per pass instead of letting the compiler hoist all
//           2 x 15 of them out of the item loop (60 VGPRs that would cost two waves of occupancy)
//   PRETW   keep all 2 x 15 twiddle powers in registers for the whole item loop (60 VGPRs, saves 56 ops per row)
template <int MINW, bool B1, bool CACHEX, bool OPAQUE, bool PRETW = false, bool QDUMP = false>
__global__ __launch_bounds__(kBlock, MINW) void lds_correlate_kernel(const float2* __restrict__ X, const float2* __restrict__ C,
                                                                      const int* __restrict__ items, const int* __restrict__ fset,
                                                                      const float2* __restrict__ tw, RowRec* __restrict__ rows,
                                                                      int E, int P, int F, int D, int B, int pch, int nchunk, float tie_scale,
                                                                      float* __restrict__ q_out) {
  __shared__ v2 lds[kLdsElems];
  __shared__ float s_peak[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  __shared__ double s_sum[kBlock / 64];
  const int t = threadIdx.x;
  // XCD-aware placement: workgroup b runs on XCD b%8 (MI355X_MICROARCH.md, workgroup dispatch).  The unit of
  // reuse is one (epoch, Doppler) pair u: its forward spectrum X[u] (B*32 KB) is read by all nchunk workgroups
  // of that unit, so they are all placed on XCD u%8 and share that XCD's L2; the code spectra (P*32 KB) end
  // up resident in every XCD's 4 MB L2.  Works for any E (a single-epoch search still fills all 8 XCDs).
  const int xcd = blockIdx.x & 7;
  const unsigned j = blockIdx.x >> 3;
  const unsigned u = (j / (unsigned)nchunk) * 8 + xcd;
  if (u >= (unsigned)E * (unsigned)D) return;
  const long e = u / (unsigned)D;
  const int d = (int)(u % (unsigned)D);
  const int p0 = (int)(j % (unsigned)nchunk) * pch;
  const int p1 = min(P, p0 + pch);
  v2 wa = ld2(tw + t), wb = ld2(tw + 16 * (t & 15));
  // as in lds_fused4k_kernel: the pass-2 twiddle powers of the inverse transform from a 1.9 KB LDS table built once per workgroup with
  // apply_powers' product tree (bit-identical records), and rising wave priorities through the row (GACQ_CORR_PRE: 4 | 8)
  constexpr int kCorrPre = PRETW ? 0 : GACQ_CORR_PRE;
  __shared__ v2 s_tw2[(kCorrPre & 4) ? 15 * 16 : 1];
  if ((kCorrPre & 4) && t < 16) {
    v2 pw[15];
    make_powers(pw, v2{wb.x, -wb.y});
#pragma unroll
    for (int k = 0; k < 15; k++) s_tw2[16 * k + t] = pw[k];
  }
  v2 pwa[PRETW ? 15 : 1], pwb[PRETW ? 15 : 1];
  if (PRETW) {
    wa.y = -wa.y;                      // inverse transform: conjugate twiddles
    wb.y = -wb.y;
    make_powers(reinterpret_cast<v2(&)[15]>(pwa), wa);
    make_powers(reinterpret_cast<v2(&)[15]>(pwb), wb);
  }
  const float inv_n = 1.0f / (float)kLdsN;
  v2 xr[(B1 && CACHEX) ? kR : 1];
  const unsigned lane_off = (unsigned)t * 16u;
  if (B1 && CACHEX) {
    const __amdgpu_buffer_rsrc_t xres = row_rsrc(X + (((e * F + fset[p0]) * D + d) * (long)B) * kLdsN);   // F == 1 here
#pragma unroll
    for (int jp = 0; jp < kR / 2; jp++) ld_pair(xres, lane_off, jp, xr[2 * jp], xr[2 * jp + 1]);
  }
  for (int p = p0; p < p1; p++) {
    const __amdgpu_buffer_rsrc_t cres = row_rsrc(C + (long)items[p] * kLdsN);
    const float2* xs = X + (((e * F + fset[p]) * D + d) * (long)B) * kLdsN;
    float q[kR];                           // B1: the magnitudes of the one block; else the sum over blocks
    if (!B1) {
#pragma unroll
      for (int k = 0; k < kR; k++) q[k] = 0.f;
    }
    const int nb = B1 ? 1 : B;
    v2 cc[B1 ? 1 : kR];                    // B > 1: the item's code spectrum stays in registers for all blocks
    if (!B1) {
#pragma unroll
      for (int jp = 0; jp < kR / 2; jp++) ld_pair(cres, lane_off, jp, cc[2 * jp], cc[2 * jp + 1]);
    }
    for (int b = 0; b < nb; b++) {
      if (OPAQUE) asm volatile("" : "+v"(wa.x), "+v"(wa.y), "+v"(wb.x), "+v"(wb.y));
      if (kCorrPre & 8) F4K_PRIO(GACQ_F4K_P4);
      v2 v[kR];
      // all loads are issued before the first asm op: the machine scheduler does not move loads across inline asm, so
      // an interleaved load/cmul loop would wait for every load separately
      if (B1 && CACHEX) {
#pragma unroll
        for (int jp = 0; jp < kR / 2; jp++) ld_pair(cres, lane_off, jp, v[2 * jp], v[2 * jp + 1]);
#pragma unroll
        for (int jj = 0; jj < kR; jj++) v[jj] = cmul(v[jj], xr[jj]);
      } else if (B1) {
        const __amdgpu_buffer_rsrc_t xres = row_rsrc(xs + (long)b * kLdsN);
        v2 xv[kR];
#pragma unroll
        for (int jp = 0; jp < kR / 2; jp++) {
          ld_pair(cres, lane_off, jp, v[2 * jp], v[2 * jp + 1]);
          ld_pair(xres, lane_off, jp, xv[2 * jp], xv[2 * jp + 1]);
        }
#pragma unroll
        for (int jj = 0; jj < kR; jj++) v[jj] = cmul(v[jj], xv[jj]);
      } else {
        const __amdgpu_buffer_rsrc_t xres = row_rsrc(xs + (long)b * kLdsN);
#pragma unroll
        for (int jp = 0; jp < kR / 2; jp++) ld_pair(xres, lane_off, jp, v[2 * jp], v[2 * jp + 1]);
#pragma unroll
        for (int jj = 0; jj < kR; jj++) v[jj] = cmul(cc[jj], v[jj]);
      }
      if (!B1 && b > 0) __syncthreads();   // previous transform's exchange-2 reads are complete
      if (PRETW) fft4096<true, 3>(v, lds, wa, wb, reinterpret_cast<const v2(*)[15]>(pwa), reinterpret_cast<const v2(*)[15]>(pwb));
      else fft4096<true, kCorrPre>(v, lds, wa, wb, nullptr, nullptr, -1, s_tw2 + ((kCorrPre & 4) ? (t & 15) : 0));
      if (B1) {
        // magnitudes straight from the transform output; lane holds lags t + 256 k.  The 1/N of ifft is a power of two: it is
        // applied once to the reduced values below instead of to all 16 magnitudes.
#pragma unroll
        for (int k = 0; k < kR; k++) {
          const v2 r = v[rev16(k)];
          q[k] = __builtin_amdgcn_sqrtf(norm2(r));                  // np.absolute(ifft(...)) * N
        }
      } else {
#pragma unroll
        for (int k = 0; k < kR; k++) {
          const v2 r = v[rev16(k)];
          q[k] += __builtin_amdgcn_sqrtf(norm2(r)) * inv_n;
        }
      }
    }
    if (QDUMP) {                                         // gacq_debug_row: the accumulated magnitude row itself (one row per launch)
#pragma unroll
      for (int k = 0; k < kR; k++) q_out[t + 256 * k] = B1 ? q[k] * inv_n : q[k];
    }
    float sum_f = q[0];
#pragma unroll
    for (int k = 1; k < kR; k++) sum_f += q[k];
    float wmaxf;
    unsigned widx;
    wave_first_max(q, (unsigned)__builtin_amdgcn_readfirstlane(t & ~63), 1u, 256u, tie_scale, wmaxf, widx);      // first maximum, like np.argmax
    const unsigned wmax = __builtin_bit_cast(unsigned, B1 ? wmaxf * inv_n : wmaxf);
    const float wsum = wave_add_f32(B1 ? sum_f * inv_n : sum_f);
    if ((t & 63) == 0) { s_peak[t >> 6] = __builtin_bit_cast(float, wmax); s_idx[t >> 6] = (int)widx; s_sum[t >> 6] = (double)wsum; }
    __syncthreads();   // also orders this item's exchange-2 reads before the next item's exchange-1 writes
    if (t == 0) {
      RowRec r;
      combine_tagged(kBlock / 64, [&](int w) { return s_peak[w]; }, [&](int w) { return s_idx[w]; }, tie_scale, r.peak, r.idx);
      r.sum = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
      rows[(e * P + p) * (long)D + d] = r;
    }
  }
}

#ifndef GACQ_PRE4K
#define GACQ_PRE4K 13     // batch kernel: pass-1 powers in registers (1) + pass-2 powers from the LDS table (4) + rising wave priorities (8)
#endif

// ---- N = 4096, one block, one carrier: forward + correlate in ONE kernel ---------------------------------------------------
// Workgroup = (epoch, Doppler bin, chunk of pch items).  Prologue: load the x window, table-NCO mix (fp64 index as in
// lds_forward_kernel), forward FFT, conjugate -- the spectrum never leaves the registers: the transform's output lane/register
// convention (register rev16(k2) of lane t holds X[t + 256 k2]) is the input convention of the inverse transform, so
// xr[j] = conj(v[rev16(j)]) is a compile-time register renaming.  Then the item loop of lds_correlate_kernel<.., B1, CACHEX>.
// No X buffer (84 MB at the bench shape), no forward launch, no launch boundary; the price is one forward transform per
// workgroup instead of one per (epoch, Doppler bin), i.e. nchunk - 1 redundant ones per unit, which is why this kernel is
// launched with larger item chunks (16-32) than the two-kernel path (8).  Same arithmetic in the same order as
// lds_forward_kernel + lds_correlate_kernel: records are bit-identical (test_fused_4096_kernel_equals_two_kernel_path).
// PREA: keep the 15 pass-1 twiddle powers of the inverse transform in registers for the whole item loop (30 VGPRs, 14 complex
// products per row less); the maximum-first peak search freed exactly that much of the 128-register budget of 4 waves per SIMD.
// (Round 3's single-launch instantiation -- the Doppler scan by the workgroup that completes an item's last bin, records handed
// over with agent-scope stores and an arrival counter -- measured slower than the three short launches it replaced (21.8 us of kernel
// against 6.8 + 12.3 + 6.5 us whose launch latencies overlap, profiles/r03_single_search_latency.log) and had no hook for the tie-safe
// re-evaluation; removed in round 6.)
template <int MINW, bool PREA>
__global__ __launch_bounds__(kBlock, MINW) void lds_fused4k_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                                    const float2* __restrict__ C, const int* __restrict__ items,
                                                                    const double* __restrict__ freq, const float2* __restrict__ nco_tab,
                                                                    const float2* __restrict__ tw, RowRec* __restrict__ rows, int E, int P,
                                                                    int D, int pch, int nchunk, int by_epoch, float tie_scale) {
  __shared__ v2 lds[kLdsElems];
  __shared__ float s_peak[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  __shared__ double s_sum[kBlock / 64];
  const int t = threadIdx.x;
  // Placement (workgroup b runs on XCD b % 8).  Batches: all D x nchunk workgroups of an epoch go to XCD e % 8, so the epoch's
  // sample block is fetched into one L2 instead of eight (round 2: 8.3 x the compulsory fetch).  Few epochs: (epoch, Doppler)
  // units are dealt round-robin as in lds_correlate_kernel, so that a single-epoch search still fills all eight XCDs.
  const int xcd = blockIdx.x & 7;
  const unsigned j = blockIdx.x >> 3;
  unsigned u;
  if (by_epoch) {
    const unsigned per_epoch = (unsigned)D * (unsigned)nchunk;
    const unsigned e8 = (j / per_epoch) * 8 + xcd;
    if (e8 >= (unsigned)E) return;
    u = e8 * (unsigned)D + (j % per_epoch) / (unsigned)nchunk;
  } else {
    u = (j / (unsigned)nchunk) * 8 + xcd;
    if (u >= (unsigned)E * (unsigned)D) return;
  }
  const long e = u / (unsigned)D;
  const int d = (int)(u % (unsigned)D);
  const int p0 = (int)(j % (unsigned)nchunk) * pch;
  const int p1 = min(P, p0 + pch);
  v2 wa = ld2(tw + t), wb = ld2(tw + 16 * (t & 15));
  // PREA (the batch kernel): the pass-2 twiddle powers of the inverse transform, conj(W_256^c)^k for the 16 lane classes
  // c = t & 15, are built once per workgroup with the product tree of apply_powers (same values, so the records stay
  // bit-identical to the two-kernel path) and read back from a 1.9 KB LDS table: one ds_read_b64 per product instead of 14
  // extra complex products per row.  (Pass 1's powers are per-lane and stay in registers, see PREA.)
  __shared__ v2 s_tw2[PREA ? 15 * 16 : 1];
  if (PREA && t < 16) {
    v2 pw[15];
    make_powers(pw, v2{wb.x, -wb.y});                  // lanes 0..15: wb = W_256^t
#pragma unroll
    for (int k = 0; k < 15; k++) s_tw2[16 * k + t] = pw[k];
  }
  v2 xr[kR];
  const unsigned lane_off = (unsigned)t * 16u;
  {
    const double f = freq[d];
    const float2* src = x + e * epoch_stride;
    v2 v[kR], w[kR];
#pragma unroll
    for (int jj = 0; jj < kR; jj++) {
      const int i = t + 256 * jj;
      const int k = nco_index(f, i);                  // gnsstools/nco.py:6-9
      v[jj] = ld2(src + i);
      w[jj] = ld2(nco_tab + k);
    }
#pragma unroll
    for (int jj = 0; jj < kR; jj++) v[jj] = cmul(v[jj], w[jj]);
    fft4096<false>(v, lds, wa, wb);
#pragma unroll
    for (int jj = 0; jj < kR; jj++) { const v2 a = v[rev16(jj)]; xr[jj] = v2{a.x, -a.y}; }      // np.conj(fft.fft(b))  acquire-gps-l1.py:32
    __syncthreads();                                  // the forward transform's exchange-2 reads are complete
  }
  const float inv_n = 1.0f / (float)kLdsN;
  v2 pwa[PREA ? 15 : 1];
  if (PREA) make_powers(reinterpret_cast<v2(&)[15]>(pwa), v2{wa.x, -wa.y});      // conjugate: inverse transform
  for (int p = p0; p < p1; p++) {
    if (PREA && (GACQ_PRE4K & 8)) F4K_PRIO(GACQ_F4K_P4);
    // keep the (remaining) twiddle powers out of the loop-invariant set
    if (PREA) asm volatile("" : "+v"(wb.x), "+v"(wb.y));
    else asm volatile("" : "+v"(wa.x), "+v"(wa.y), "+v"(wb.x), "+v"(wb.y));
    v2 v[kR];
    const __amdgpu_buffer_rsrc_t cres = row_rsrc(C + (long)items[p] * kLdsN);
#pragma unroll
    for (int jp = 0; jp < kR / 2; jp++) ld_pair(cres, lane_off, jp, v[2 * jp], v[2 * jp + 1]);
#pragma unroll
    for (int jj = 0; jj < kR; jj++) v[jj] = cmul(v[jj], xr[jj]);
    if (PREA) fft4096<true, GACQ_PRE4K>(v, lds, wa, wb, reinterpret_cast<const v2(*)[15]>(pwa), nullptr, -1, s_tw2 + (t & 15));
    else fft4096<true>(v, lds, wa, wb);
    // lane t holds lags t + 256 k.  The 1/N of ifft is a power of two: applied once to the reduced values.
    float m[kR];
#pragma unroll
    for (int k = 0; k < kR; k++) {
      const v2 r = v[rev16(k)];
      m[k] = __builtin_amdgcn_sqrtf(norm2(r));                      // np.absolute(ifft(...)) * N
    }
    float sum_f = m[0];
#pragma unroll
    for (int k = 1; k < kR; k++) sum_f += m[k];
    float wmaxf;
    unsigned widx;
    wave_first_max(m, (unsigned)__builtin_amdgcn_readfirstlane(t & ~63), 1u, 256u, tie_scale, wmaxf, widx);
    const unsigned wmax = __builtin_bit_cast(unsigned, wmaxf * inv_n);
    const float wsum = wave_add_f32(sum_f * inv_n);
    if ((t & 63) == 0) { s_peak[t >> 6] = __builtin_bit_cast(float, wmax); s_idx[t >> 6] = (int)widx; s_sum[t >> 6] = (double)wsum; }
    __syncthreads();   // also orders this item's exchange-2 reads before the next item's exchange-1 writes
    if (t == 0) {
      RowRec r;
      combine_tagged(kBlock / 64, [&](int w) { return s_peak[w]; }, [&](int w) { return s_idx[w]; }, tie_scale, r.peak, r.idx);
      r.sum = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
      rows[(e * P + p) * (long)D + d] = r;
    }
  }
}

int twiddle_table(gacq_ctx* ctx, const float2** out) { return twiddle_cache(ctx, "W4096", kLdsN, kLdsN, out); }

}  // namespace

namespace gacq {

bool lds_supported(int N) { return N == kLdsN || N == kBig; }

// N = 16384: the radix-32 form of gacq_lds16k.hip unless GACQ_OPT_LDS_VARIANT = 16 selects the radix-16 form of this file (B1I 0-5 %
// slower, GLONASS within 2 % either way in the same process: the in-run A/B of bench.py, roofline.ab.n16384_transform)
static bool radix16_16k(const gacq_ctx* ctx) { return ctx->opt[GACQ_OPT_LDS_VARIANT] == 16; }

#ifdef GACQ_PHASE_TIMING16
extern "C" int gacq_debug_phase16(unsigned long long* out128, int reset) {
  if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(gacq_phase16), sizeof(unsigned long long) * 128) != hipSuccess) return GACQ_ERR_HIP;
  if (reset) {
    unsigned long long z[128] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gacq_phase16), z, sizeof z) != hipSuccess) return GACQ_ERR_HIP;
  }
  return GACQ_OK;
}
#endif

// code spectra straight from the (complex, zero-extended) replica rows with the engine's own forward transform: no rocFFT plan
int lds_code_spectra(gacq_ctx* ctx, const float2* replica_rows, float2* perm, int nprn, int N, bool radix16) {
  if (!lds_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "LDS FFT engine: N=%d not supported", N);
  if (N == kBig && !radix16) return r32_code_spectra(ctx, replica_rows, perm, nprn);
  if (N == kBig) {
    const float2* twn;
    int rcb = twiddle_cache(ctx, "W16384_lo", kBig, 1024, &twn);
    if (rcb != GACQ_OK) return rcb;
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)lds16k_forward_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
    hipLaunchKernelGGL((lds16k_forward_kernel<false, true>), dim3((unsigned)nprn), dim3(kBigThreads), kBigLdsBytes, ctx->stream, replica_rows,
                       (size_t)kBig, perm, (const double*)nullptr, (const float2*)nullptr, twn, kBig, 1, 1);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  const float2* tw;
  int rc = twiddle_table(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  hipLaunchKernelGGL((lds_forward_kernel<false, true>), dim3((unsigned)nprn), dim3(kBlock), 0, ctx->stream, replica_rows, (size_t)kLdsN, perm,
                     (const double*)nullptr, (const float2*)nullptr, tw, kLdsN, 1, 1);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int lds_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, int N, const double* d_freq, int FD,
                int B, const float2* tab, float2* X) {
  if (!lds_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "LDS FFT engine: N=%d not supported", N);
  if (N == kBig && !radix16_16k(ctx)) return r32_forward(ctx, x, nsamp, nepoch, n, d_freq, FD, B, tab, X);
  if (N == kBig) {
    const float2* twn;
    int rcb = twiddle_cache(ctx, "W16384_lo", kBig, 1024, &twn);
    if (rcb != GACQ_OK) return rcb;
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)lds16k_forward_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
    hipLaunchKernelGGL(lds16k_forward_kernel<false>, dim3((unsigned)((long)nepoch * FD * B)), dim3(kBigThreads), kBigLdsBytes, ctx->stream, x,
                       nsamp, X, d_freq, tab, twn, n, FD, B);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  const float2* tw;
  int rc = twiddle_table(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  const long rows = (long)nepoch * FD * B;
  hipLaunchKernelGGL(lds_forward_kernel<false>, dim3((unsigned)rows), dim3(kBlock), 0, ctx->stream, x, nsamp, X, d_freq, tab, tw, n, FD, B);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

bool lds_fused_supported(const gacq_ctx* ctx, int N, int P, int F) { return N == kBig && F == P && ctx->opt[GACQ_OPT_FUSED_16K]; }

int lds_fused_search(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, int N, const float2* spectra, const int* d_items,
                     const int* d_fset, const double* d_freq, const float2* tab, int nitems, int D, int B, RowRec* rows, float tie_scale) {
  if (N != kBig) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "fused LDS search: N=%d not supported", N);
  if (!radix16_16k(ctx)) return r32_fused_search(ctx, x, nsamp, nepoch, n, spectra, d_items, d_fset, d_freq, tab, nitems, D, B, rows, tie_scale);
  const float2* twn;
  int rc = twiddle_cache(ctx, "W16384_lo", kBig, 1024, &twn);
  if (rc != GACQ_OK) return rc;
  GACQ_HIP(ctx, hipFuncSetAttribute((const void*)lds16k_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
  hipLaunchKernelGGL(lds16k_fused_kernel<false>, dim3((unsigned)((long)nepoch * D * nitems)), dim3(kBigThreads), kBigLdsBytes, ctx->stream, x,
                     nsamp, spectra, d_items, d_fset, d_freq, tab, twn, rows, n, nitems, D, B, tie_scale);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

// Worth it for batches only: with few (epoch, Doppler) units every workgroup's own forward transform sits on the critical path
// (single epoch: 32 us fused against 19 us for the two kernels), with many it replaces a launch, 84 MB of X traffic and a tail.
bool lds_fused4k_supported(const gacq_ctx* ctx, int N, int B, int F, long units) {
  return N == kLdsN && B == 1 && F == 1 && (ctx->opt[GACQ_OPT_FUSED_4K] >= 2 || (ctx->opt[GACQ_OPT_FUSED_4K] == 1 && units >= 1024));
}

int lds_fused4k_search(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, const float2* spectra, const int* d_items,
                       const double* d_freq, const float2* tab, int nitems, int D, RowRec* rows, float tie_scale) {
  const float2* tw;
  int rc = twiddle_table(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  const long units = (long)nepoch * D;
  // one forward transform per workgroup: amortise it over up to 32 items while >= ~2048 workgroups remain
  int pch = 8;
  while (pch < 32 && units * ((nitems + 2 * pch - 1) / (2 * pch)) >= 2048) pch *= 2;
  if (ctx->opt[GACQ_OPT_LDS_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_LDS_PCH];
  pch = std::min(pch, nitems);
  const int nchunk = (nitems + pch - 1) / pch;
  const int by_epoch = nepoch >= 64 ? 1 : 0;
  const long units8 = by_epoch ? (long)((nepoch + 7) / 8) * D : (units + 7) / 8;      // units per XCD
  const dim3 grid((unsigned)(8 * units8 * nchunk));
  hipLaunchKernelGGL((lds_fused4k_kernel<4, true>), grid, dim3(kBlock), 0, ctx->stream, x, nsamp, spectra, d_items, d_freq, tab, tw, rows,
                     nepoch, nitems, D, pch, nchunk, by_epoch, tie_scale);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int lds_debug_nco(gacq_ctx* ctx, int N, int n, const double* d_freq, bool fused, int* d_idx) {
  if (!lds_supported(N) || (fused && N != kBig)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "NCO index dump: no %sLDS forward kernel for N=%d", fused ? "fused " : "", N);
  if (N == kBig && !radix16_16k(ctx)) return r32_debug_nco(ctx, n, d_freq, fused, d_idx);
  if (N == kBig && fused) {
    int rc = ensure(ctx, ctx->fset, sizeof(int));
    if (rc != GACQ_OK) return rc;
    ctx->up_fset.clear();
    GACQ_HIP(ctx, hipMemsetAsync(ctx->fset.p, 0, sizeof(int), ctx->stream));
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)lds16k_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
    hipLaunchKernelGGL(lds16k_fused_kernel<true>, dim3(1), dim3(kBigThreads), kBigLdsBytes, ctx->stream, (const float2*)nullptr, (size_t)0,
                       (const float2*)nullptr, (const int*)ctx->fset.p, (const int*)ctx->fset.p, d_freq, (const float2*)nullptr,
                       (const float2*)nullptr, (RowRec*)d_idx, n, 1, 1, 1, 1.0f);
  } else if (N == kBig) {
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)lds16k_forward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
    hipLaunchKernelGGL(lds16k_forward_kernel<true>, dim3(1), dim3(kBigThreads), kBigLdsBytes, ctx->stream, (const float2*)nullptr, (size_t)0,
                       (float2*)d_idx, d_freq, (const float2*)nullptr, (const float2*)nullptr, n, 1, 1);
  } else {
    hipLaunchKernelGGL(lds_forward_kernel<true>, dim3(1), dim3(kBlock), 0, ctx->stream, (const float2*)nullptr, (size_t)0, (float2*)d_idx, d_freq,
                       (const float2*)nullptr, (const float2*)nullptr, n, 1, 1);
  }
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int lds_correlate(gacq_ctx* ctx, const float2* X, const float2* spectra, const int* d_items, const int* d_fset, int nepoch,
                  int nitems, int F, int D, int B, int N, RowRec* rows, float tie_scale, float* q_out) {
  if (q_out && (nepoch != 1 || nitems != 1 || D != 1)) return set_error(ctx, GACQ_ERR_BAD_ARG, "LDS FFT engine: a row dump takes exactly one row");
  if (!lds_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "LDS FFT engine: N=%d not supported", N);
  if (N == kBig && !radix16_16k(ctx)) return r32_correlate(ctx, X, spectra, d_items, d_fset, nepoch, nitems, F, D, B, rows, tie_scale, q_out);
  if (N == kBig) {
    const float2* twn;
    int rcb = twiddle_cache(ctx, "W16384_lo", kBig, 1024, &twn);
    if (rcb != GACQ_OK) return rcb;
    // One 1024-thread workgroup per CU, 32 per XCD.  Workgroup = (one item, a group of G of the XCD's units): the item's code
    // spectrum is read once per workgroup, so G is as large as still leaves ~2 rounds of workgroups per XCD (>= 60; at most 32
    // units), evened out so that the groups of an XCD have the same size where possible.  B1I (63 items, 200 units, B = 10):
    // 25 units per XCD -> G = 25, 63 workgroups of 250 rows per XCD.  Measured (profiles/r04_16k_unit_group_sweep.log): HBM
    // traffic per launch 2.07 GB (round 3, unit-major) -> 1.07 GB (G = 5) -> 0.89 (9) -> 0.78 (13) -> 0.59 GB (25) = 2.2 x the
    // compulsory bytes, kernel time unchanged within 2 % (3.37-3.45 ms in the four-signal step): HBM was never what paced it.
    const long units = (long)nepoch * D;
    int pch = 1;
    if (ctx->opt[GACQ_OPT_LDS_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_LDS_PCH];
    pch = std::min(pch, nitems);
    const int nchunk = (nitems + pch - 1) / pch;
    const long units8 = (units + 7) / 8;                                 // units per XCD
    long g0 = std::max<long>(1, std::min<long>(32, units8 * nchunk / 60));
    int ugroup = (int)((units8 + ((units8 + g0 - 1) / g0) - 1) / ((units8 + g0 - 1) / g0));
    if (ctx->opt[GACQ_OPT_LDS_UGROUP] >= 1) ugroup = (int)std::min<long>(ctx->opt[GACQ_OPT_LDS_UGROUP], units8);
    const long groups = (units8 + ugroup - 1) / ugroup;
    auto kern16 = q_out ? lds16k_correlate_kernel<true> : lds16k_correlate_kernel<false>;
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)kern16, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
    hipLaunchKernelGGL(kern16, dim3((unsigned)(8 * groups * nchunk)), dim3(kBigThreads), kBigLdsBytes, ctx->stream, X,
                       spectra, d_items, d_fset, twn, rows, nepoch, nitems, F, D, B, pch, nchunk, ugroup, tie_scale, q_out);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  const float2* tw;
  int rc = twiddle_table(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  // items per workgroup: keep >= ~2048 workgroups in flight (256 CUs x 4 resident x 2), at most 8 per group
  const long rows_total = (long)nepoch * nitems * D;
  int pch = (int)std::max<long>(1, std::min<long>(8, rows_total / 2048));
  // B > 1: nothing is shared between a workgroup's items (every item walks the unit's B forward-spectrum rows again), but the
  // workgroups of a unit that run side by side on its XCD walk them TOGETHER and share them in that L2: as few items per workgroup as
  // still give it ~8 rows (B = 10, 800 units: 8 -> 1 item per workgroup 1.32 -> 1.23 ms; B = 80: 1 item 0.92 ms, 8 items 1.56 ms --
  // profiles/r05_engine3_writer_wave_priority_sweep.log)
  if (!(B == 1 && F == 1)) pch = std::min(pch, std::max(1, (8 + B - 1) / B));
  if (ctx->opt[GACQ_OPT_LDS_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_LDS_PCH];
  pch = std::min(pch, nitems);
  const int nchunk = (nitems + pch - 1) / pch;
  const long units8 = ((long)nepoch * D + 7) / 8;
  const long grid = 8 * units8 * nchunk;
  // register-cached X needs all items of a workgroup to share one forward set.  <2 waves/SIMD declared, X cached, twiddle powers
  // hoisted> led the round-1 A/B of ten register/occupancy variants (profiles/r01_ab_variants_*.log, all within 5 %); the others
  // are gone from the build.
  const bool b1 = (B == 1) && (F == 1);
  auto kern = q_out ? (b1 ? lds_correlate_kernel<2, true, true, false, false, true> : lds_correlate_kernel<2, false, false, false, false, true>)
                    : (b1 ? lds_correlate_kernel<2, true, true, false> : lds_correlate_kernel<2, false, false, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, X, spectra, d_items, d_fset, tw, rows, nepoch,
                     nitems, F, D, B, pch, nchunk, tie_scale, q_out);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

// ---- inner stages of the split engine (N = R*4096) -------------------------------------------------------
namespace {
// W_N^m for m < 256 R
int big_twiddles(gacq_ctx* ctx, int N, int R, const float2** out) {
  return twiddle_cache(ctx, "WN_lo_" + std::to_string(N), N, 256 * R, out);
}
}  // namespace

int lds_inner_forward(gacq_ctx* ctx, float2* rows, long nrows, bool conj) {
  const float2* tw;
  int rc = twiddle_table(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  if (conj) hipLaunchKernelGGL(lds_inner_forward_kernel<true>, dim3((unsigned)nrows), dim3(kBlock), 0, ctx->stream, rows, tw);
  else hipLaunchKernelGGL(lds_inner_forward_kernel<false>, dim3((unsigned)nrows), dim3(kBlock), 0, ctx->stream, rows, tw);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int lds_inner_correlate(gacq_ctx* ctx, const float2* X, const float2* spectra, const int* d_items, const int* d_fset, long g0,
                        long ng, int P, int F, int D, int B, int R, int N, float2* Z) {
  const float2 *tw, *twn;
  int rc = twiddle_table(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  if ((rc = big_twiddles(ctx, N, R, &twn)) != GACQ_OK) return rc;
  // (epoch, item) rows touched by this pass, cut into chunks of pch per workgroup; >= ~2048 workgroups, <= 8 items each
  const long ep_first = g0 / D, ep_last = (g0 + ng - 1) / D;
  const long nep = ep_last - ep_first + 1;
  int pch = (int)std::max<long>(1, std::min<long>(8, nep * D * B * R / 2048));
  if (ctx->opt[GACQ_OPT_SPLIT_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_SPLIT_PCH];
  const int nblk_ep = (int)((nep + pch - 1) / pch);
  hipLaunchKernelGGL(lds_inner_correlate_kernel, dim3((unsigned)((long)R * nblk_ep * D * B)), dim3(kBlock), 0, ctx->stream, X, spectra,
                     d_items, d_fset, tw, twn, Z, g0, ng, ep_first, nblk_ep, pch, P, F, D, B, R);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace gacq
