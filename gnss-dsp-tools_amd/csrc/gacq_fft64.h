// Length-4096 complex128 transform in one 256-thread workgroup (16 points per lane, three radix-16 passes, two LDS exchanges):
// the decomposition of gacq_ldsfft.hip's fft4096 in the reference's own arithmetic type (numpy complex128,
// acquire-gps-l1.py:30-33).  Shared by the fused complex128 search kernel of engine 5 (gacq_verify.hip) and by the re-evaluation
// of near-tied rows (gacq_tiesafe.hip).  Plain fp64 arithmetic (v_fma_f64 / v_add_f64 / v_mul_f64: half the fp32 vector rate on
// this part); real and imaginary parts go through LDS as two 8-byte planes, so the bank-conflict-free index patterns of the fp32
// engine (8-byte elements, row pitch 257 in the second exchange) carry over unchanged.  LDS: 2 x 16 x 257 x 8 B = 65.8 KB per
// workgroup -> two workgroups per CU.
#pragma once
#include <hip/hip_runtime.h>

namespace gacq {
namespace f64 {

constexpr int kN = 4096;
constexpr int kPitch = 257;
constexpr int kPlane = 16 * kPitch;                      // doubles per plane
constexpr int kLdsBytes = 2 * kPlane * (int)sizeof(double);

struct cd { double x, y; };
__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd operator*(cd a, cd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd conj(cd a) { return {a.x, -a.y}; }
__device__ __forceinline__ cd add_i(cd a, cd b) { return {a.x - b.y, a.y + b.x}; }       // a + i b
__device__ __forceinline__ cd sub_i(cd a, cd b) { return {a.x + b.y, a.y - b.x}; }       // a - i b

// sqrt(a) for a >= 0 from v_rsq_f64 (relative error ~2^-23), one Goldschmidt step (-> ~2^-45) and one residual correction
// (d = a - g^2, g += d h: the error squares once more): seven fp64 operations without the scaling branches of the library routine,
// accurate to an ulp or two -- far below the 1e-10 the complex128 paths are held to
__device__ __forceinline__ double sqrt_pos(double a) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, a);
  g = __builtin_fma(d, h, g);
  return a > 0.0 ? g : 0.0;
}

// Wave64 reductions of fp64 values on DPP (no LDS round trips, no divergent compare-and-swap chains): quad_perm xor-1, xor-2,
// row_half_mirror, row_mirror leave every lane of a 16-lane row with the row's result; the four rows are combined through readlane.
__device__ __forceinline__ double dpp_f64(double v, const int ctrl_sel) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(0u, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0u, hi, 0xB1, 0xf, 0xf, false); break;    // quad_perm [1,0,3,2]
    case 1: lo = __builtin_amdgcn_update_dpp(0u, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0u, hi, 0x4E, 0xf, 0xf, false); break;    // quad_perm [2,3,0,1]
    case 2: lo = __builtin_amdgcn_update_dpp(0u, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0u, hi, 0x141, 0xf, 0xf, false); break;  // row_half_mirror
    default: lo = __builtin_amdgcn_update_dpp(0u, lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0u, hi, 0x140, 0xf, 0xf, false); break; // row_mirror
  }
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, lane), hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_add_f64(double v) {      // wave-uniform result
  v += dpp_f64(v, 0);
  v += dpp_f64(v, 1);
  v += dpp_f64(v, 2);
  v += dpp_f64(v, 3);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max_pos_f64(double v) {  // v >= 0: wave-uniform maximum
  v = fmax(v, dpp_f64(v, 0));
  v = fmax(v, dpp_f64(v, 1));
  v = fmax(v, dpp_f64(v, 2));
  v = fmax(v, dpp_f64(v, 3));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

__device__ __forceinline__ constexpr int rev16(int k) { return 4 * (k & 3) + (k >> 2); }

template <bool INV> __device__ __forceinline__ void dft4(cd& a, cd& b, cd& c, cd& d) {
  const cd s0 = a + c, d0 = a - c, s1 = b + d, t = b - d;
  a = s0 + s1;
  c = s0 - s1;
  b = INV ? add_i(d0, t) : sub_i(d0, t);
  d = INV ? sub_i(d0, t) : add_i(d0, t);
}

// W16^m (forward exp(-2 pi i m / 16); INV: conjugate)
template <bool INV> __device__ __forceinline__ cd w16(int m) {
  constexpr double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178, h = 0.70710678118654752440;
  const double re = (m == 1) ? c1 : (m == 2) ? h : (m == 3) ? s1 : (m == 6) ? -h : (m == 9) ? -c1 : 0.0;
  const double im = (m == 1) ? s1 : (m == 2) ? h : (m == 3) ? c1 : (m == 6) ? h : (m == 9) ? -s1 : 0.0;
  return {re, INV ? im : -im};
}

// In-place 16-point DFT; X[k] is left in register rev16(k)
template <bool INV> __device__ __forceinline__ void dft16(cd (&v)[16]) {
#pragma unroll
  for (int n0 = 0; n0 < 4; n0++) dft4<INV>(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);      // A[n0][k0] at v[n0 + 4 k0]
  v[5] = v[5] * w16<INV>(1);
  v[6] = v[6] * w16<INV>(2);
  v[7] = v[7] * w16<INV>(3);
  v[9] = v[9] * w16<INV>(2);
  v[10] = INV ? cd{-v[10].y, v[10].x} : cd{v[10].y, -v[10].x};                              // W16^4 = -i (inverse: +i)
  v[11] = v[11] * w16<INV>(6);
  v[13] = v[13] * w16<INV>(3);
  v[14] = v[14] * w16<INV>(6);
  v[15] = v[15] * w16<INV>(9);
#pragma unroll
  for (int k0 = 0; k0 < 4; k0++) dft4<INV>(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);   // X[k0 + 4 k1] at v[4 k0 + k1]
}

// v[rev16(k)] *= w^k, k = 1..15 (product tree of depth <= 4)
__device__ __forceinline__ void apply_powers(cd (&v)[16], cd w1) {
  const cd w2 = w1 * w1, w3 = w2 * w1, w4 = w2 * w2;
  v[rev16(1)] = v[rev16(1)] * w1;    v[rev16(2)] = v[rev16(2)] * w2;    v[rev16(3)] = v[rev16(3)] * w3;
  const cd w5 = w4 * w1, w6 = w3 * w3, w7 = w4 * w3, w8 = w4 * w4;
  v[rev16(4)] = v[rev16(4)] * w4;    v[rev16(5)] = v[rev16(5)] * w5;    v[rev16(6)] = v[rev16(6)] * w6;    v[rev16(7)] = v[rev16(7)] * w7;
  const cd w9 = w8 * w1, w10 = w5 * w5, w11 = w8 * w3, w12 = w6 * w6;
  v[rev16(8)] = v[rev16(8)] * w8;    v[rev16(9)] = v[rev16(9)] * w9;    v[rev16(10)] = v[rev16(10)] * w10; v[rev16(11)] = v[rev16(11)] * w11;
  v[rev16(12)] = v[rev16(12)] * w12;
  const cd w13 = w8 * w5, w14 = w7 * w7, w15 = w8 * w7;
  v[rev16(13)] = v[rev16(13)] * w13; v[rev16(14)] = v[rev16(14)] * w14; v[rev16(15)] = v[rev16(15)] * w15;
}

// w^1 .. w^15 with the product tree of apply_powers (identical values)
__device__ __forceinline__ void make_powers(cd (&pw)[15], cd w1) {
  pw[0] = w1;
  pw[1] = w1 * w1;         pw[2] = pw[1] * w1;      pw[3] = pw[1] * pw[1];
  pw[4] = pw[3] * w1;      pw[5] = pw[2] * pw[2];   pw[6] = pw[3] * pw[2];   pw[7] = pw[3] * pw[3];
  pw[8] = pw[7] * w1;      pw[9] = pw[4] * pw[4];   pw[10] = pw[7] * pw[2];  pw[11] = pw[5] * pw[5];
  pw[12] = pw[7] * pw[4];  pw[13] = pw[6] * pw[6];  pw[14] = pw[7] * pw[6];
}

// In: v[j] = x[t + 256 j]; out: v[rev16(k2)] = X[t + 256 k2] (unnormalised for INV).  wa = W_4096^t, wb = W_256^(t & 15) (forward
// values).  lds: kLdsBytes.  The caller puts a barrier between two transforms that use the same LDS buffer.
// TAB2: the pass-2 powers (already conjugated for INV) come from an LDS table laid out [k - 1][lane class], tb2 already
// offset by the lane's class t & 15 -- one 16-byte read per product instead of 14 complex products per row to rebuild them.
// One LDS read per instruction where the including unit asks for it (#define GACQ_F64_UNPAIR before the include): paired into
// ds_read2st64_b64 by the compiler the reads run at half rate (MI355X guide, LDS table; GACQ_UNPAIR in gacq_cplx.h).  Same-box A/B on the
// fused search kernel: 13.80 -> 13.60 ms per 1024-epoch step; unpairing the writes as well costs 3 %.  NOT for the re-evaluation
// kernels of gacq_tiesafe.hip: they live on 256 VGPRs + 256 AGPRs + scratch, and with the separators in place this toolchain's
// VGPR-to-AGPR spilling produced wrong rows in tie_recheck4k_kernel (right again with -mllvm -amdgpu-spill-vgpr-to-agpr=0 or with two
// workgroups per CU, i.e. no AGPR spills) -- their speed does not matter, so they keep the compiler's pairing.
#if defined(GACQ_F64_UNPAIR) && !defined(GACQ_LDS_PAIRED)
#define F64_UR() asm volatile("" ::: "memory")
#else
#define F64_UR() do {} while (0)
#endif
// PRIO: rising wave priority through the row (1 after the exchange-1 writes, 2 after the exchange-2 writes, 3 once the exchange-2 reads
// are issued; the caller resets to 0 at the top of its row loop) -- see F4K_PRIO in gacq_ldsfft.hip.
#define F64_PRIO(n) do { if (PRIO) asm volatile("s_setprio %0" :: "n"(n) : "memory"); } while (0)
template <bool INV, bool TAB2 = false, bool PRIO = false>
__device__ __forceinline__ void fft4096(cd (&v)[16], double* lds, cd wa, cd wb, int t, const cd* tb2 = nullptr) {
  double* lre = lds;
  double* lim = lds + kPlane;
  if (INV) { wa.y = -wa.y; wb.y = -wb.y; }
  dft16<INV>(v);
  apply_powers(v, wa);
  {  // exchange 1: (n0, n1; k0) -> (n0, k0; n1)
    const int wbase = (t & 15) + 256 * (t >> 4);
#pragma unroll
    for (int k = 0; k < 16; k++) { lre[wbase + 16 * k] = v[rev16(k)].x; lim[wbase + 16 * k] = v[rev16(k)].y; }
    F64_PRIO(1);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) { v[j].x = lre[t + 256 * j]; F64_UR(); v[j].y = lim[t + 256 * j]; F64_UR(); }
  }
  dft16<INV>(v);
  if (TAB2) {
#pragma unroll
    for (int k = 1; k < 16; k++) v[rev16(k)] = v[rev16(k)] * tb2[16 * (k - 1)];
  } else {
    apply_powers(v, wb);
  }
  __syncthreads();
  {  // exchange 2: (n0, k0; k1) -> (k0, k1; n0), row pitch 257
    const int wbase = (t >> 4) + kPitch * (t & 15);
#pragma unroll
    for (int k = 0; k < 16; k++) { lre[wbase + 16 * k] = v[rev16(k)].x; lim[wbase + 16 * k] = v[rev16(k)].y; }
    F64_PRIO(2);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) { v[j].x = lre[t + kPitch * j]; F64_UR(); v[j].y = lim[t + kPitch * j]; F64_UR(); }
    F64_PRIO(3);
  }
  dft16<INV>(v);
}

}  // namespace f64
}  // namespace gacq
