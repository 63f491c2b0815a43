// N = 16384 (BeiDou B1I / B2I zero-padded, GLONASS L1 / L2) in ONE 512-thread workgroup, 32 points per lane:
//   16384 = 32 (across the waves) x 32 x 16 (inside a 16-lane group), three passes, TWO exchanges per transform.
//     n = t + 512 j          (t = thread, j = register)
//     k = ka + 32 (k0 + 32 k1)
//   forward (decimation in frequency, natural order in):
//     pass 0  DFT32 over j -> ka, twiddle W_N^{t ka} (table A: 31 values per thread, in registers for the whole kernel);
//             exchange 0: output ka goes to the 16-lane group that owns sub-transform ka (group g of wave w owns ka = 4 w + g) --
//             the only step that crosses waves: one workgroup barrier
//     pass 1  the group's lane lam holds u[lam + 16 j'], DFT32 over j' -> k0, twiddle W_512^{lam k0} (table B: 31 x 16 values in LDS);
//             transpose 1 inside the group (pitch 17, no barrier: a wave's LDS accesses execute in order)
//     pass 2  lane mu holds k0 = mu + 16 h (h = 0, 1) for lam = 0..15: two DFT16 over lam -> k1
//     out: register r = h + 2 k1 of thread t' holds X[(t' >> 4) + 32 (t' & 15) + 512 r]
//   inverse (decimation in time) is the transposed network with conjugated twiddles: it takes exactly that order in and leaves
//   N y[t + 512 j] in register j of thread t.  Spectra (X, C_p) live in memory in the order the forward transform produces them
//   ("physical lane-pair layout": element (t', r) at (r >> 1) * 1024 + 2 t' + (r & 1)): 16-byte accesses, 1 KiB per wave and
//   instruction, no reordering pass anywhere.
// Why this shape, and what it measured (round 6, profiles/r06_16k_radix32_experiments.log): the radix-16 form (gacq_ldsfft.hip: 1024
// threads x 16 points, 16 x 16 x 16 x 4, three exchanges) fills 16 waves x 128 registers with the code spectrum (32), the row (32) and
// the accumulators (16), so its three sets of inter-pass twiddle powers are REBUILT per row (42 of a row's 237 complex products) and
// every cheaper source (LDS or memory tables) cost more than it saved.  Eight waves x 256 registers hold the same row state
// (64 + 64 + 32) plus half of the big twiddle set (table A, 32); the second set is small enough for LDS (table B, 4 KB, 16 distinct
// addresses per read); one exchange and one twiddle set are gone with the fourth pass: 13 % fewer VALU instructions per row
// (SQ_INSTS_VALU 9.75e8 against 1.117e9 per B1I launch) and 2/3 of the LDS store traffic.  What that buys is small -- config 5's B1I
// search 0-5 % faster than the radix-16 form, its GLONASS search within 2 %, alternating in one process on six boxes -- because with two waves per
// SIMD instead of four the LDS round trips and the two workgroup barriers of a row are covered less (VALU pipes 61 % busy against 77 %):
// one row in flight per CU is what the register file allows either way (row + code spectrum + accumulators = 320 KB of its 512), and
// that, not the instruction count, is what paces a row.  The default since round 6; GACQ_OPT_LDS_VARIANT = 16 selects the radix-16
// form.  tools/model_fft16k_r32.py checks the index algebra against numpy.fft and every access pattern for bank conflicts (none;
// SQ_LDS_BANK_CONFLICT = 0 measured).
#include "gacq_common.h"
#include "gacq_cplx.h"
#include "gacq_ldsutil.h"

#include <algorithm>

using namespace gacq;

namespace {

constexpr int kN = 16384;
constexpr int kT = 512;                         // threads per workgroup (8 waves, 2 per SIMD, <= 256 VGPRs)
constexpr int kP = 32;                          // points per lane
constexpr int kS = 560;                         // region stride in complex elements: >= 544 (pitch-17 transpose), = 16 (mod 32)
constexpr int kPitch = 17;
constexpr int kTabB = 32 * kS * (int)sizeof(v2);            // byte offset of table B (W_512^{lam k0}, [k0 - 1][lam])
constexpr int kScratch = kTabB + 31 * 16 * (int)sizeof(v2); // cross-wave reduction scratch
constexpr int kLdsBytes = kScratch + 256;                   // 147 584 B: one workgroup per CU
// Table A (W_N^{t ka}, per thread) is held in registers for the whole kernel for ka = 1..16 (32 VGPRs); W^{t (16 + k)} is applied as
// two products, W^{16 t} then W^{t k}: 15 more complex products per transform (4 % of its packed instructions) than with all 31 values
// resident -- the 30 registers are what lets table B travel in two batches instead of value by value next to a resident code
// spectrum (all 31: 188 bytes of scratch in the row loop).  Every kernel uses the same form, so that the fused kernel and the
// two-kernel path stay bit-identical.
constexpr int kTA = 16;

// cos / sin (2 pi k / 32), k = 0..15
constexpr float kC32[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f,
                            0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f, 0.f, -0.19509032201612819f,
                            -0.38268343236508973f, -0.55557023301960196f, -0.70710678118654746f, -0.83146961230254535f,
                            -0.92387953251128674f, -0.98078528040323043f};
constexpr float kS32[16] = {0.f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654746f,
                            0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.f, 0.98078528040323043f,
                            0.92387953251128674f, 0.83146961230254546f, 0.70710678118654757f, 0.55557023301960218f,
                            0.38268343236508989f, 0.19509032201612861f};

// a * conj(b): the modifiers of cmul with the signs of b.im flipped -- a table of forward twiddles serves the inverse transform too
__device__ __forceinline__ v2 cmulc(v2 a, v2 b) {
  v2 t, r;
  asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
      : "=&v"(t), "=v"(r)
      : "v"(a), "v"(b));
  return r;
}
template <bool CONJ> __device__ __forceinline__ v2 twmul(v2 a, v2 w) { return CONJ ? cmulc(a, w) : cmul(a, w); }
// Two complex products in one statement, multiplies first: each FMA then sits one instruction behind the multiply it depends on
// instead of right behind it (two waves per SIMD hide less of a back-to-back dependency than four).
template <bool CONJ> __device__ __forceinline__ void twmul2(v2& o0, v2 a0, v2 w0, v2& o1, v2 a1, v2 w1) {
  v2 t0, t1, r0, r1;
  if (CONJ)
    asm("v_pk_mul_f32 %0, %4, %5 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %6, %7 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %2, %4, %5, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %6, %7, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(t0), "=&v"(t1), "=&v"(r0), "=v"(r1)
        : "v"(a0), "v"(w0), "v"(a1), "v"(w1));
  else
    asm("v_pk_mul_f32 %0, %4, %5 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
        "v_pk_mul_f32 %1, %6, %7 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
        "v_pk_fma_f32 %2, %4, %5, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %6, %7, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=&v"(t0), "=&v"(t1), "=&v"(r0), "=v"(r1)
        : "v"(a0), "v"(w0), "v"(a1), "v"(w1));
  o0 = r0;
  o1 = r1;
}
// v[ka] *= W_N^{t ka} (CONJ: its conjugate), ka = 1..31
template <bool CONJ> __device__ __forceinline__ void apply_ta(v2 (&v)[kP], const v2 (&ta)[kTA]) {
#pragma unroll
  for (int k = 1; k < 16; k += 2) twmul2<CONJ>(v[k], v[k], ta[k - 1], v[k + 1], v[k + 1], ta[k]);          // ka = 1..16
  v[17] = twmul<CONJ>(v[17], ta[15]);
#pragma unroll
  for (int k = 2; k < 16; k += 2) twmul2<CONJ>(v[16 + k], v[16 + k], ta[15], v[17 + k], v[17 + k], ta[15]);       // W^{16 t} ...
  v[17] = twmul<CONJ>(v[17], ta[0]);
#pragma unroll
  for (int k = 2; k < 16; k += 2) twmul2<CONJ>(v[16 + k], v[16 + k], ta[k - 1], v[17 + k], v[17 + k], ta[k]);     // ... then W^{t k}
}

// X[K] = E[K] + W32^K O[K], X[K + 16] = E[K] - W32^K O[K]; W32^8 = -/+i and the two (1 -/+ i)/sqrt2 rotations cost no multiply
template <bool INV, int K> __device__ __forceinline__ void dft32_combine(v2 (&v)[kP], const v2 (&e)[16], const v2 (&o)[16]) {
  const v2 E = e[rev16(K)], O = o[rev16(K)];
  const v2 hh = {0.70710678118654752f, -0.70710678118654752f};
  if constexpr (K == 0) {
    v[0] = E + O;
    v[16] = E - O;
  } else if constexpr (K == 8) {
    v[8] = INV ? add_i(E, O) : sub_i(E, O);
    v[24] = INV ? sub_i(E, O) : add_i(E, O);
  } else if constexpr (K == 4) {
    const v2 r = INV ? add_i(O, O) : sub_i(O, O);        // O (1 -/+ i); its 1/sqrt2 rides on the FMA
    v[4] = fma_lo(E, r, hh);
    v[20] = fma_hi(E, r, hh);
  } else if constexpr (K == 12) {
    const v2 r = INV ? sub_i(O, O) : add_i(O, O);        // -O (1 +/- i) sqrt2 / 2
    v[12] = fma_hi(E, r, hh);
    v[28] = fma_lo(E, r, hh);
  } else {
    const v2 w = {kC32[K], INV ? kS32[K] : -kS32[K]};
    const v2 p = cmul_k(O, w);
    v[K] = E + p;
    v[K + 16] = E - p;
  }
}
// In-place 32-point DFT, natural order in and out: two DFT16 (even / odd inputs) + the W32 combine.  152 + 58 packed instructions.
template <bool INV> __device__ __forceinline__ void dft32_finish(v2 (&v)[kP], const v2 (&e)[16], const v2 (&o)[16]);
template <bool INV> __device__ __forceinline__ void dft32(v2 (&v)[kP]) {
  v2 e[16], o[16];
#pragma unroll
  for (int m = 0; m < 16; m++) { e[m] = v[2 * m]; o[m] = v[2 * m + 1]; }
  dft16<INV>(e);
  dft16<INV>(o);
  dft32_finish<INV>(v, e, o);
}
template <bool INV> __device__ __forceinline__ void dft32_finish(v2 (&v)[kP], const v2 (&e)[16], const v2 (&o)[16]) {
  dft32_combine<INV, 0>(v, e, o);   dft32_combine<INV, 1>(v, e, o);   dft32_combine<INV, 2>(v, e, o);   dft32_combine<INV, 3>(v, e, o);
  dft32_combine<INV, 4>(v, e, o);   dft32_combine<INV, 5>(v, e, o);   dft32_combine<INV, 6>(v, e, o);   dft32_combine<INV, 7>(v, e, o);
  dft32_combine<INV, 8>(v, e, o);   dft32_combine<INV, 9>(v, e, o);   dft32_combine<INV, 10>(v, e, o);  dft32_combine<INV, 11>(v, e, o);
  dft32_combine<INV, 12>(v, e, o);  dft32_combine<INV, 13>(v, e, o);  dft32_combine<INV, 14>(v, e, o);  dft32_combine<INV, 15>(v, e, o);
}

// Wave priorities (s_setprio) at the segment ends of a row.  Two waves share a SIMD: the one that has just issued the LDS accesses
// that end its segment steps down so that the other's arithmetic runs under the round trip.  Ten assignments of the three levels,
// none included, measured within +-3 % of each other (profiles/r06_16k_radix32_experiments.log, section 7): what the waves wait for
// is the CU's one LDS pipe, which the four SIMDs' waves reach together, not each other's arithmetic.
#ifndef GACQ_R32_PA
#define GACQ_R32_PA 2       // after the cross-wave gather: last DFT32 + magnitudes
#define GACQ_R32_PB 3       // after the staged row has been read: C * x + two DFT16
#define GACQ_R32_PC 1       // after the transpose reads are issued: table B + DFT32
#endif
#define R32_PRIO(n) do { if ((n) >= 0) asm volatile("s_setprio %0" :: "n"(n) : "memory"); } while (0)

// Phase timing of r32_correlate_kernel (diagnostic builds only, -DGACQ_PHASE_TIMING16; tools/phase_timing16.py): lane 0 of every
// wave accumulates the shader-clock cycles between marks into gacq_phase16[wave][phase] (read back with gacq_debug_phase16).
// Never defined in the product build.
#ifdef GACQ_PHASE_TIMING16
__device__ unsigned long long gacq_phase16[16 * 8];
#define R32_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); acc16_[i] += now_ - mark16_; mark16_ = now_; } while (0)
#else
#define R32_MARK(i) do { } while (0)
#endif

struct Lane {
  int t, lam;
  v2* lds;         // all regions
  v2* own;         // region of the sub-transform this 16-lane group owns (ka = t >> 4)
  const v2* tb;    // table B, already offset by lam
  v2* stage;       // the wave's four regions: landing area of the LDS-DMA row prefetch (16 KiB of 17.5)
};
__device__ __forceinline__ Lane lane_setup(char* smem) {
  Lane L;
  L.t = threadIdx.x;
  L.lam = L.t & 15;
  L.lds = reinterpret_cast<v2*>(smem);
  L.own = L.lds + (L.t >> 4) * kS;
  L.tb = reinterpret_cast<const v2*>(smem + kTabB) + L.lam;
  L.stage = L.lds + (L.t >> 6) * 4 * kS;
  return L;
}

// Table A into registers (W_N^{t ka}, ka = 1..31) and table B into LDS (W_512^{lam k0} = W_N^{32 lam k0}), from the full table of
// W_N^m (fp64-evaluated, rounded once): every twiddle of the transform is a table value, none is a product of others.
__device__ __forceinline__ void tables_load(const float2* __restrict__ twN, v2 (&ta)[kTA], char* smem) {
  const int t = threadIdx.x;
#pragma unroll
  for (int ka = 1; ka <= kTA; ka++) ta[ka - 1] = ld2(twN + ((t * ka) & (kN - 1)));
  if (t < 31 * 16) reinterpret_cast<v2*>(smem + kTabB)[t] = ld2(twN + 32 * (t & 15) * ((t >> 4) + 1));      // entry (k0 - 1) * 16 + lam
}

// Forward transform.  In: v[j] = x[t + 512 j].  Out: v[h + 2 k1] = X[(t >> 4) + 32 (t & 15) + 512 (h + 2 k1)].
// The caller guarantees that no wave still uses its regions when the exchange-0 stores start (they go to every region).
__device__ __forceinline__ void fft_r32_fwd(v2 (&v)[kP], const Lane& L, const v2 (&ta)[kTA]) {
  dft32<false>(v);
  apply_ta<false>(v, ta);
#pragma unroll
  for (int ka = 0; ka < kP; ka++) LDS_ST1(L.lds[ka * kS + L.t], v[ka]);                       // exchange 0
  lds_barrier();
  const v2* rd = L.own + L.lam;
#pragma unroll
  for (int jp = 0; jp < kP; jp += 2) v[jp] = LDS_LD(rd[16 * jp]);                             // even inputs first: their DFT16 runs under the rest
#pragma unroll
  for (int jp = 1; jp < kP; jp += 2) v[jp] = LDS_LD(rd[16 * jp]);
  v2 tbv[31];                                                                                 // table B, all of it, under the DFT32
#pragma unroll
  for (int k0 = 1; k0 < kP; k0++) tbv[k0 - 1] = LDS_LD(L.tb[16 * (k0 - 1)]);
  dft32<false>(v);
  v2* wr = L.own + L.lam;
  LDS_ST1(wr[0], v[0]);
#pragma unroll
  for (int k0 = 1; k0 < kP; k0++) LDS_ST1(wr[kPitch * k0], cmul(v[k0], tbv[k0 - 1]));         // transpose 1: element (k0, lam) at 17 k0 + lam
  const v2* rd2 = L.own + kPitch * L.lam;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    v2 a[16];
#pragma unroll
    for (int lp = 0; lp < 16; lp++) a[lp] = LDS_LD(rd2[kPitch * 16 * h + lp]);                // lane mu = lam holds k0 = mu + 16 h
    dft16<false>(a);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) v[h + 2 * k1] = a[rev16(k1)];
  }
}

// Inverse transform, group-private part.  In: v[h + 2 k1] = Y[(t >> 4) + 32 (t & 15) + 512 (h + 2 k1)]; on return the group's region
// holds z[lam + 16 j'] (its 512-point inverse transform), ready for the cross-wave exchange.
// Table B travels in two batches (the even k0 ahead of the transposed row, the odd k0 under the DFT16 of the even half): read value by
// value next to its product -- what the compiler does with the register file full -- every one of the 31 reads exposes a whole LDS
// round trip (B1I correlate 3.25 ms against 2.78 with the multiplies ablated, profiles/r06_16k_radix32_experiments.log).
__device__ __forceinline__ void ifft_r32_private(v2 (&v)[kP], const Lane& L) {
  v2* wr = L.own + kPitch * L.lam;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    v2 a[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) a[k1] = v[h + 2 * k1];
    dft16<true>(a);                                                                           // over k1 -> lam'
#pragma unroll
    for (int lp = 0; lp < 16; lp++) LDS_ST1(wr[kPitch * 16 * h + lp], a[rev16(lp)]);          // element (k0 = mu + 16 h, lam') at 17 k0 + lam'
  }
  const v2* rd = L.own + L.lam;
  v2 e[16], o[16], tbe[15], tbo[16];
#pragma unroll
  for (int m = 1; m < 16; m++) tbe[m - 1] = LDS_LD(L.tb[16 * (2 * m - 1)]);                   // k0 = 2 m
#pragma unroll
  for (int m = 0; m < 16; m++) e[m] = LDS_LD(rd[kPitch * 2 * m]);
#pragma unroll
  for (int m = 0; m < 16; m++) o[m] = LDS_LD(rd[kPitch * (2 * m + 1)]);
  R32_PRIO(GACQ_R32_PC);
  e[1] = cmulc(e[1], tbe[0]);
#pragma unroll
  for (int m = 2; m < 16; m += 2) twmul2<true>(e[m], e[m], tbe[m - 1], e[m + 1], e[m + 1], tbe[m]);
#pragma unroll
  for (int m = 0; m < 16; m++) tbo[m] = LDS_LD(L.tb[16 * (2 * m)]);                           // k0 = 2 m + 1
  dft16<true>(e);
#pragma unroll
  for (int m = 0; m < 16; m += 2) twmul2<true>(o[m], o[m], tbo[m], o[m + 1], o[m + 1], tbo[m + 1]);
  dft16<true>(o);
  dft32_finish<true>(v, e, o);                                                                // over k0 -> j'
  v2* wr2 = L.own + L.lam;
#pragma unroll
  for (int jp = 0; jp < 16; jp++) { LDS_ST1(wr2[16 * jp], v[jp]); LDS_ST1(wr2[16 * (jp + 16)], v[jp + 16]); }      // in the order the combine produces them
}
// cross-wave part: gather the 32 partial transforms of n = t (mod 512) ...
__device__ __forceinline__ void ifft_r32_gather(v2 (&v)[kP], const Lane& L) {
  const v2* rd = L.lds + L.t;
#pragma unroll
  for (int ka = 0; ka < kP; ka += 2) v[ka] = LDS_LD(rd[ka * kS]);                              // even inputs first: their DFT16 runs under the rest
#pragma unroll
  for (int ka = 1; ka < kP; ka += 2) v[ka] = LDS_LD(rd[ka * kS]);
}
// ... and combine them: v[j] = N y[t + 512 j]
__device__ __forceinline__ void ifft_r32_final(v2 (&v)[kP], const v2 (&ta)[kTA]) {
  apply_ta<true>(v, ta);
  dft32<true>(v);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc16k(const float2* row) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, kN * (int)sizeof(float2), 0x00020000);
}
// elements (t, 2 p) and (t, 2 p + 1) of a row in the physical lane-pair layout
__device__ __forceinline__ void ld_pair16k(__amdgpu_buffer_rsrc_t r, unsigned lane_off, int p, v2& a, v2& b) {
  const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, (unsigned)p * 8192u, 0));
  a = q.xy;
  b = q.zw;
}

// LDS-DMA of one spectrum row into the wave's own regions: 16 x 1 KiB, lane l's 16 bytes of piece p land at stage + 1024 p + 16 l --
// no VGPRs are tied up while the row is in flight.
__device__ __forceinline__ void dma_row(const float2* __restrict__ row, v2* stage) {
  const char* src = reinterpret_cast<const char*>(row) + (size_t)threadIdx.x * 16;
#pragma unroll
  for (int p = 0; p < kP / 2; p++)
    __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 8192), (lptr_t)(reinterpret_cast<char*>(stage) + p * 1024), 16, 0, 0);
}
__device__ __forceinline__ void dma_wait_read(v2 (&x)[kP], const v2* stage) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const f4* p = reinterpret_cast<const f4*>(stage) + (threadIdx.x & 63);
#pragma unroll
  for (int pc = 0; pc < kP / 2; pc++) { const f4 q = p[pc * 64]; x[2 * pc] = q.xy; x[2 * pc + 1] = q.zw; }
  R32_PRIO(GACQ_R32_PB);
}

// cross-wave (max, first argmax, sum) of one item through the scratch words behind the tables; thread 0 writes the record
__device__ __forceinline__ void reduce_store(char* smem, float peak, unsigned widx, float wsum, float tie_scale, RowRec* dst) {
  float* s_peak = reinterpret_cast<float*>(smem + kScratch);
  int* s_idx = reinterpret_cast<int*>(smem + kScratch + 64);
  double* s_sum = reinterpret_cast<double*>(smem + kScratch + 128);
  const int t = threadIdx.x;
  if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = (int)widx; s_sum[t >> 6] = (double)wsum; }
  lds_barrier();
  if (t == 0) {
    RowRec r;
    combine_tagged(kT / 64, [&](int w) { return s_peak[w]; }, [&](int w) { return s_idx[w]; }, tie_scale, r.peak, r.idx);
    double bs = s_sum[0];
    for (int w = 1; w < kT / 64; w++) bs += s_sum[w];
    r.sum = bs;
    *dst = r;
  }
}
__device__ __forceinline__ void reduce_item(char* smem, const float (&q)[kP], float tie_scale, RowRec* dst) {
  float sum_f = q[0];
#pragma unroll
  for (int k = 1; k < kP; k++) sum_f += q[k];
  // lane l of wave w holds lags 64 w + l + 512 k: smallest lag among the maxima first (np.argmax, acquire-beidou-b1i.py:34)
  float peak;
  unsigned widx;
  wave_first_max(q, (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x & ~63), 1u, 512u, tie_scale, peak, widx);
  reduce_store(smem, peak, widx, wave_add_f32(sum_f), tie_scale, dst);
}

// carrier wipe-off of one 512 x 32 window: table NCO with the index in fp64 exactly as numpy (gnsstools/nco.py:6-9)
template <bool DUMP>
__device__ __forceinline__ void load_mix(v2 (&v)[kP], const float2* __restrict__ src, const float2* __restrict__ nco_tab, double f, int* dump) {
  const int t = threadIdx.x;
#pragma unroll
  for (int half = 0; half < 2; half++) {       // two halves: 16 sample + 16 table loads in flight, not 32 + 32
    v2 w[kP / 2];
#pragma unroll
    for (int jj = 0; jj < kP / 2; jj++) {
      const int j = half * (kP / 2) + jj;
      const int i = t + kT * j;
      const int k = nco_index(f, i);
      if (DUMP) { dump[i] = k; continue; }
      v[j] = ld2(src + i);
      w[jj] = ld2(nco_tab + k);
    }
    if (DUMP) continue;
#pragma unroll
    for (int jj = 0; jj < kP / 2; jj++) v[half * (kP / 2) + jj] = cmul(v[half * (kP / 2) + jj], w[jj]);
  }
}

// forward: one workgroup per (e, f, d, b) row; output conj(FFT) in the physical lane-pair layout, 1 KiB per wave and store.
// DUMP (test hook gacq_debug_nco_indices): the index expression, on the same frequency table, is stored as int32 into X
// (reinterpreted) and the kernel returns; x is not read.
// PLAIN (code spectra, once per signal): row r of x is transformed as it is -- no carrier wipe-off, no conjugation -- so the code
// spectra come out of the same transform, in the same layout, as the forward spectra they are multiplied with (c = fft.fft(c),
// acquire-beidou-b1i.py:24), and building a signal needs no rocFFT plan.
template <bool DUMP, bool PLAIN>
__global__ __launch_bounds__(kT) void r32_forward_kernel(const float2* __restrict__ x, size_t epoch_stride, float2* __restrict__ X,
                                                         const double* __restrict__ freq, const float2* __restrict__ nco_tab,
                                                         const float2* __restrict__ twN, int n, int FD, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lane L = lane_setup(smem);
  const unsigned row = blockIdx.x;          // ((e*FD + fd)*B + b)
  const int b = (int)(row % (unsigned)B);
  const unsigned r2 = row / (unsigned)B;
  const int fd = (int)(r2 % (unsigned)FD);
  const long e = r2 / (unsigned)FD;
  const float2* src = x + e * epoch_stride + (size_t)b * n;
  v2 v[kP];
  if (PLAIN) {
#pragma unroll
    for (int j = 0; j < kP; j++) v[j] = ld2(src + L.t + kT * j);
  } else {
    load_mix<DUMP>(v, src, nco_tab, freq[fd], reinterpret_cast<int*>(X) + row * (long)kN);
  }
  if (DUMP) return;
  v2 ta[kTA];
  tables_load(twN, ta, smem);                 // table B is first read behind the barrier of exchange 0
  fft_r32_fwd(v, L, ta);
  float2* dst = X + row * (long)kN;
  const float cs = PLAIN ? 1.f : -1.f;
#pragma unroll
  for (int p = 0; p < kP / 2; p++) {
    const v2 a = v[2 * p], c = v[2 * p + 1];
    // np.conj(fft.fft(b))  acquire-beidou-b1i.py:32   (PLAIN: the transform itself)
    *reinterpret_cast<float4*>(dst + p * 1024 + 2 * L.t) = make_float4(a.x, cs * a.y, c.x, cs * c.y);
  }
}

// correlate: workgroup = (chunk of items, group of `ugroup` (epoch, Doppler) units of one XCD); per (item, unit):
// sum_b |IFFT(C_p * X_b)|/N -> (max, argmax, sum)   (acquire-beidou-b1i.py:29-35).
// Item-major: the item's code spectrum is loaded ONCE and stays in registers for every unit of the group and all B blocks of each;
// the workgroups resident on an XCD (consecutive in launch order: same group, different items) walk the group's units side by side,
// so the forward spectrum of the (unit, block) they are all working on is fetched from HBM once and served from that XCD's L2.
// The forward spectrum of the NEXT row is fetched by LDS-DMA into the wave's own regions as soon as the cross-wave exchange of the
// current row has been read out, i.e. under the last DFT32 and the magnitudes.  Two workgroup barriers per row.
template <bool QDUMP>
__global__ __launch_bounds__(kT) void r32_correlate_kernel(const float2* __restrict__ X, const float2* __restrict__ C,
                                                           const int* __restrict__ items, const int* __restrict__ fset,
                                                           const float2* __restrict__ twN, RowRec* __restrict__ rows, int E, int P, int F,
                                                           int D, int B, int pch, int nchunk, int ugroup, float tie_scale,
                                                           float* __restrict__ q_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lane L = lane_setup(smem);
  // placement: workgroup b runs on XCD b % 8; XCD x owns the units u = x (mod 8), in groups of `ugroup` consecutive owned units
  const int xcd = blockIdx.x & 7;
  const unsigned jb = blockIdx.x >> 3;
  const unsigned grp = jb / (unsigned)nchunk;
  const int p0 = (int)(jb % (unsigned)nchunk) * pch;
  const int p1 = min(P, p0 + pch);
  const unsigned U = (unsigned)E * (unsigned)D;
  const unsigned u0 = grp * (unsigned)ugroup * 8u + (unsigned)xcd;           // first unit of the group; the i-th is u0 + 8 i
  if (u0 >= U) return;
  const int nu = (int)min((unsigned)ugroup, (U - u0 + 7u) / 8u);
  const unsigned lane_off = (unsigned)L.t * 16u;
  const float inv_n = 1.0f / (float)kN;
  // first forward-spectrum row of (item p, i-th unit of the group)
  auto unit_row = [&](int p, int i) -> const float2* {
    const unsigned u = u0 + 8u * (unsigned)i;
    const long e = u / (unsigned)D;
    const int d = (int)(u % (unsigned)D);
    return X + (((e * F + fset[p]) * D + d) * (long)B) * kN;
  };
  const float2* xrow = unit_row(p0, 0);
  v2 ta[kTA];
  tables_load(twN, ta, smem);
  dma_row(xrow, L.stage);
  lds_barrier();                                         // table B is in place
#ifdef GACQ_PHASE_TIMING16
  unsigned long long acc16_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long mark16_ = __builtin_readcyclecounter();
#endif
  for (int p = p0; p < p1; p++) {
    const __amdgpu_buffer_rsrc_t cres = row_rsrc16k(C + (long)items[p] * kN);
    v2 c[kP];
#pragma unroll
    for (int pc = 0; pc < kP / 2; pc++) ld_pair16k(cres, lane_off, pc, c[2 * pc], c[2 * pc + 1]);
    for (int i = 0; i < nu; i++) {
      const unsigned u = u0 + 8u * (unsigned)i;
      const long e = u / (unsigned)D;
      const int d = (int)(u % (unsigned)D);
      // the row after this unit's last block: the next unit of the group, else the next item's first unit, else nothing
      const float2* xnext_unit = (i + 1 < nu) ? unit_row(p, i + 1) : ((p + 1 < p1) ? unit_row(p + 1, 0) : nullptr);
      float q[kP];
#pragma unroll
      for (int k = 0; k < kP; k++) q[k] = 0.f;
      for (int b = 0; b < B; b++) {
        v2 v[kP];
        R32_MARK(0);                                       // previous row's tail (reduction, code-spectrum loads)
        dma_wait_read(v, L.stage);
        R32_MARK(1);
#pragma unroll
        for (int r = 0; r < kP; r += 2) twmul2<false>(v[r], c[r], v[r], v[r + 1], c[r + 1], v[r + 1]);
        ifft_r32_private(v, L);
        R32_MARK(3);
        lds_barrier();
        R32_MARK(4);
        ifft_r32_gather(v, L);
        lds_barrier();                                     // every wave has read these regions: they may be overwritten
        R32_PRIO(GACQ_R32_PA);
        R32_MARK(5);
        const float2* nx = (b + 1 < B) ? xrow + (long)(b + 1) * kN : xnext_unit;
        if (nx) dma_row(nx, L.stage);
        ifft_r32_final(v, ta);
#pragma unroll
        for (int k = 0; k < kP; k++) q[k] += __builtin_amdgcn_sqrtf(norm2(v[k])) * inv_n;
        R32_MARK(6);
      }
      xrow = xnext_unit;
      if (QDUMP) {                                         // gacq_debug_row: the accumulated magnitude row itself (one row per launch)
#pragma unroll
        for (int k = 0; k < kP; k++) q_out[L.t + kT * k] = q[k];
      }
      reduce_item(smem, q, tie_scale, rows + (e * P + p) * (long)D + d);
    }
  }
#ifdef GACQ_PHASE_TIMING16
  if ((L.t & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 8; i++) if (acc16_[i]) atomicAdd(&gacq_phase16[(L.t >> 6) * 8 + i], acc16_[i]);
  }
#endif
}

// Fused search for item lists in which every item has its own carrier (F == P: the GLONASS FDMA channels, or a single item): the
// forward spectrum of (e, f, d, b) is used by exactly one item, so writing it to HBM and reading it back (8 N bytes each way per row)
// buys nothing.  Workgroup = (epoch, Doppler bin, item); per block b: mix + forward transform -- whose output order is the inverse
// transform's input order, so the spectrum stays in registers -- conj * C_p, inverse transform, |.| accumulated in registers
// (acquire-glonass-l1.py:28-34).  Three workgroup barriers per block.  Same arithmetic in the same order as r32_forward_kernel +
// r32_correlate_kernel.
template <bool DUMP>
__global__ __launch_bounds__(kT) void r32_fused_kernel(const float2* __restrict__ x, size_t epoch_stride, const float2* __restrict__ C,
                                                       const int* __restrict__ items, const int* __restrict__ fset,
                                                       const double* __restrict__ freq, const float2* __restrict__ nco_tab,
                                                       const float2* __restrict__ twN, RowRec* __restrict__ rows, int n, int P, int D, int B,
                                                       float tie_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lane L = lane_setup(smem);
  unsigned blk = blockIdx.x;                          // ((e*D + d)*P + p): the P items of one (e, d) run side by side
  const int p = (int)(blk % (unsigned)P);
  blk /= (unsigned)P;
  const int d = (int)(blk % (unsigned)D);
  const long e = blk / (unsigned)D;
  const double f = freq[(long)fset[p] * D + d];
  const __amdgpu_buffer_rsrc_t cres = row_rsrc16k(C + (long)items[p] * kN);
  const unsigned lane_off = (unsigned)L.t * 16u;
  const float inv_n = 1.0f / (float)kN;
  v2 ta[kTA];
  if (!DUMP) tables_load(twN, ta, smem);       // table B is first read behind the barrier of exchange 0
  float q[kP];
#pragma unroll
  for (int k = 0; k < kP; k++) q[k] = 0.f;
  for (int b = 0; b < B; b++) {
    const float2* src = x + e * epoch_stride + (size_t)b * n;
    v2 v[kP];
    load_mix<DUMP>(v, src, nco_tab, f, reinterpret_cast<int*>(rows) + (long)blockIdx.x * kN);
    if (DUMP) return;
    R32_PRIO(GACQ_R32_PB);
    fft_r32_fwd(v, L, ta);
#pragma unroll
    for (int pc = 0; pc < kP / 2; pc++) {
      v2 c0, c1;
      ld_pair16k(cres, lane_off, pc, c0, c1);
      v[2 * pc] = cmulc(c0, v[2 * pc]);               // C_p * np.conj(fft.fft(b))
      v[2 * pc + 1] = cmulc(c1, v[2 * pc + 1]);
    }
    ifft_r32_private(v, L);
    lds_barrier();
    ifft_r32_gather(v, L);
    lds_barrier();                                    // every wave has read these regions: the next block's exchange 0 may overwrite them
    R32_PRIO(GACQ_R32_PA);
    ifft_r32_final(v, ta);
#pragma unroll
    for (int k = 0; k < kP; k++) q[k] += __builtin_amdgcn_sqrtf(norm2(v[k])) * inv_n;
  }
  reduce_item(smem, q, tie_scale, rows + (e * P + p) * (long)D + d);
}

int full_twiddles(gacq_ctx* ctx, const float2** out) { return twiddle_cache(ctx, "W16384", kN, kN, out); }

template <class K> int set_lds(gacq_ctx* ctx, K kern) {
  GACQ_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  return GACQ_OK;
}

}  // namespace

namespace gacq {

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

int r32_code_spectra(gacq_ctx* ctx, const float2* replica_rows, float2* perm, int nprn) {
  const float2* tw;
  int rc = full_twiddles(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  if ((rc = set_lds(ctx, r32_forward_kernel<false, true>)) != GACQ_OK) return rc;
  hipLaunchKernelGGL((r32_forward_kernel<false, true>), dim3((unsigned)nprn), dim3(kT), kLdsBytes, ctx->stream, replica_rows, (size_t)kN, perm,
                     (const double*)nullptr, (const float2*)nullptr, tw, kN, 1, 1);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int r32_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, const double* d_freq, int FD, int B, const float2* tab,
                float2* X) {
  const float2* tw;
  int rc = full_twiddles(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  if ((rc = set_lds(ctx, r32_forward_kernel<false, false>)) != GACQ_OK) return rc;
  hipLaunchKernelGGL((r32_forward_kernel<false, false>), dim3((unsigned)((long)nepoch * FD * B)), dim3(kT), kLdsBytes, ctx->stream, x, nsamp, X,
                     d_freq, tab, tw, n, FD, B);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int r32_fused_search(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, const float2* spectra, const int* d_items,
                     const int* d_fset, const double* d_freq, const float2* tab, int nitems, int D, int B, RowRec* rows, float tie_scale) {
  const float2* tw;
  int rc = full_twiddles(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  if ((rc = set_lds(ctx, r32_fused_kernel<false>)) != GACQ_OK) return rc;
  hipLaunchKernelGGL(r32_fused_kernel<false>, dim3((unsigned)((long)nepoch * D * nitems)), dim3(kT), kLdsBytes, ctx->stream, x, nsamp, spectra,
                     d_items, d_fset, d_freq, tab, tw, rows, n, nitems, D, B, tie_scale);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int r32_debug_nco(gacq_ctx* ctx, int n, const double* d_freq, bool fused, int* d_idx) {
  int rc;
  if (fused) {
    if ((rc = ensure(ctx, ctx->fset, sizeof(int))) != GACQ_OK) return rc;
    ctx->up_fset.clear();
    GACQ_HIP(ctx, hipMemsetAsync(ctx->fset.p, 0, sizeof(int), ctx->stream));
    if ((rc = set_lds(ctx, r32_fused_kernel<true>)) != GACQ_OK) return rc;
    hipLaunchKernelGGL(r32_fused_kernel<true>, dim3(1), dim3(kT), kLdsBytes, ctx->stream, (const float2*)nullptr, (size_t)0,
                       (const float2*)nullptr, (const int*)ctx->fset.p, (const int*)ctx->fset.p, d_freq, (const float2*)nullptr,
                       (const float2*)nullptr, (RowRec*)d_idx, n, 1, 1, 1, 1.0f);
  } else {
    if ((rc = set_lds(ctx, r32_forward_kernel<true, false>)) != GACQ_OK) return rc;
    hipLaunchKernelGGL((r32_forward_kernel<true, false>), dim3(1), dim3(kT), kLdsBytes, ctx->stream, (const float2*)nullptr, (size_t)0,
                       (float2*)d_idx, d_freq, (const float2*)nullptr, (const float2*)nullptr, n, 1, 1);
  }
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int r32_correlate(gacq_ctx* ctx, const float2* X, const float2* spectra, const int* d_items, const int* d_fset, int nepoch, int nitems,
                  int F, int D, int B, RowRec* rows, float tie_scale, float* q_out) {
  const float2* tw;
  int rc = full_twiddles(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  // One 512-thread workgroup per CU, 32 per XCD.  Workgroup = (one item, a group of G of the XCD's units): the item's code spectrum
  // is read once per workgroup, so G is as large as still leaves ~2 rounds of workgroups per XCD (>= 60; at most 32 units), evened
  // out so that the groups of an XCD have the same size where possible.  B1I (63 items, 200 units, B = 10): 25 units per XCD ->
  // G = 25, 63 workgroups of 250 rows per XCD (profiles/r04_16k_unit_group_sweep.log: HBM traffic 2.2 x the compulsory bytes).
  const long units = (long)nepoch * D;
  int pch = 1;
  if (ctx->opt[GACQ_OPT_LDS_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_LDS_PCH];
  pch = std::min(pch, nitems);
  const int nchunk = (nitems + pch - 1) / pch;
  const long units8 = (units + 7) / 8;                                 // units per XCD
  const long g0 = std::max<long>(1, std::min<long>(32, units8 * nchunk / 60));
  int ugroup = (int)((units8 + ((units8 + g0 - 1) / g0) - 1) / ((units8 + g0 - 1) / g0));
  if (ctx->opt[GACQ_OPT_LDS_UGROUP] >= 1) ugroup = (int)std::min<long>(ctx->opt[GACQ_OPT_LDS_UGROUP], units8);
  const long groups = (units8 + ugroup - 1) / ugroup;
  auto kern = q_out ? r32_correlate_kernel<true> : r32_correlate_kernel<false>;
  if ((rc = set_lds(ctx, kern)) != GACQ_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * groups * nchunk)), dim3(kT), kLdsBytes, ctx->stream, X, spectra, d_items, d_fset, tw, rows,
                     nepoch, nitems, F, D, B, pch, nchunk, ugroup, tie_scale, q_out);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace gacq
