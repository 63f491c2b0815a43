// Device-side complex arithmetic and small DFTs on packed-f32 VALU ops (shared by the LDS FFT engine and the
// split-radix engines).  A complex value is one 64-bit VGPR pair (re, im); v_pk_{add,mul,fma}_f32 process both
// halves per lane and their op_sel / neg modifiers give the swaps and sign flips of complex products for free.
// hipcc does not find these forms on its own (3-4 instructions + v_mov per complex multiply), hence the asm wrappers.
#pragma once
#include <hip/hip_runtime.h>

// One LDS access per instruction.  Left alone, the compiler pairs neighbouring 8-byte accesses into ds_read2_b64 / ds_read2st64_b64 and
// ds_write2_b64; on gfx950 a paired read takes 8 LDS cycles against 2 + 2 for two ds_read_b64, a paired write 13 against 6 + 6
// (MI355X guide, LDS instruction table).  An empty asm with a memory clobber between two accesses keeps them apart (it orders memory
// instructions only; the VALU work still schedules across it).  Same-box A/B (tools/ab_variants.sh, profiles/r04_lds_unpaired_ab.log):
// reads unpaired -- 4096-point kernels 5.54 -> 5.43 ms per config-2 step; reads and writes unpaired -- 16384-point kernels, config 5
// 7.75 -> 7.43 ms (the 4096-point kernels are 0.7 % faster with their writes left paired: fewer instructions to issue).
// -DGACQ_LDS_PAIRED restores the compiler's pairing for such comparisons.
#ifdef GACQ_LDS_PAIRED
#define GACQ_UNPAIR() do {} while (0)
#else
#define GACQ_UNPAIR() asm volatile("" ::: "memory")
#endif

namespace gacq {

typedef float v2 __attribute__((ext_vector_type(2)));

// |z|^2 as one multiply and one FMA.  Written as x*x + y*y, hipcc's SLP vectoriser emits v_pk_mul_f32 + v_add_f32: 6 issue cycles
// instead of 4 on a part where a packed-f32 instruction costs two plain ones.
__device__ __forceinline__ float norm2(v2 r) { return __builtin_fmaf(r.y, r.y, r.x * r.x); }

// a + i*b = (a.re - b.im, a.im + b.re)
__device__ __forceinline__ v2 add_i(v2 a, v2 b) {
  v2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a - i*b = (a.re + b.im, a.im - b.re)
__device__ __forceinline__ v2 sub_i(v2 a, v2 b) {
  v2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a*b: t = (-a.im*b.im, a.im*b.re); r = (a.re*b.re + t.lo, a.re*b.im + t.hi)
// (one asm statement for both instructions: hipcc pads an s_nop between adjacent asm statements; the RAW dependency on t
// is interlocked by the hardware like any VALU->VALU dependency)
__device__ __forceinline__ v2 cmul(v2 a, v2 b) {
  v2 t, r;
  asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
      "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
      : "=&v"(t), "=v"(r)
      : "v"(a), "v"(b));
  return r;
}
// a*w with w a compile-time constant held in an SGPR pair (one constant-bus operand per instruction)
__device__ __forceinline__ v2 cmul_k(v2 a, v2 w) {
  v2 t, r;
  asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
      "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
      : "=&v"(t), "=v"(r)
      : "v"(a), "s"(w));
  return r;
}
// real constant x complex value, constant pair cs = (c, s) in SGPRs, one half broadcast to both lanes:
// acc + x*cs.lo | acc + x*cs.hi | acc - x*cs.hi | x*cs.hi | -x*cs.hi
__device__ __forceinline__ v2 fma_lo(v2 acc, v2 x, v2 cs) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "s"(cs), "v"(acc));
  return r;
}
__device__ __forceinline__ v2 fma_hi(v2 acc, v2 x, v2 cs) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "s"(cs), "v"(acc));
  return r;
}
__device__ __forceinline__ v2 fms_hi(v2 acc, v2 x, v2 cs) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(x), "s"(cs), "v"(acc));
  return r;
}
__device__ __forceinline__ v2 mul_hi(v2 x, v2 cs) {
  v2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "s"(cs));
  return r;
}
__device__ __forceinline__ v2 mul_hi_neg(v2 x, v2 cs) {
  v2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "s"(cs));
  return r;
}

// Wave64 reductions on DPP (no LDS, no ds_bpermute): quad_perm xor-1, xor-2, row_half_mirror, row_mirror leave every lane
// of a 16-lane row with the row's result (4 one-cycle-issue ops), the four rows are read with v_readlane and combined on the
// scalar unit.  Results are wave-uniform.  s_nop 1 covers the "VALU write -> DPP read" wait states hipcc cannot see in asm.
#define GACQ_DPP_REDUCE(OP)                                                                                   \
  asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"           \
               "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"           \
               "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"               \
               "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                    \
               "s_nop 1"                                                                                      \
               : "+v"(x))
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
  GACQ_DPP_REDUCE("v_max_u32_dpp");
  const unsigned a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
  const unsigned c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
  GACQ_DPP_REDUCE("v_min_u32_dpp");
  const unsigned a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
  const unsigned c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
  return min(min(a, b), min(c, d));
}
__device__ __forceinline__ float wave_add_f32(float v) {
  unsigned x = __builtin_bit_cast(unsigned, v);
  GACQ_DPP_REDUCE("v_add_f32_dpp");
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32)), d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
  return (a + b) + (c + d);
}

// 4-point DFT, forward W4 = -i (INV: +i).  ROTC: input c still needs its -/+i factor (folded into the adds).
template <bool INV, bool ROTC> __device__ __forceinline__ void dft4(v2& a, v2& b, v2& c, v2& d) {
  v2 s0, d0;
  if (ROTC) {
    s0 = INV ? add_i(a, c) : sub_i(a, c);
    d0 = INV ? sub_i(a, c) : add_i(a, c);
  } else {
    s0 = a + c;
    d0 = a - c;
  }
  const v2 s1 = b + d, t = b - d;
  a = s0 + s1;
  c = s0 - s1;
  b = INV ? add_i(d0, t) : sub_i(d0, t);
  d = INV ? sub_i(d0, t) : add_i(d0, t);
}

// W16^m as (re, im): forward exp(-2 pi i m/16), inverse the conjugate
template <bool INV, int M> __device__ __forceinline__ v2 w16() {
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  constexpr float re = (M == 1) ? c1 : (M == 2) ? h : (M == 3) ? s1 : (M == 6) ? -h : (M == 9) ? -c1 : 0.f;
  constexpr float im = (M == 1) ? s1 : (M == 2) ? h : (M == 3) ? c1 : (M == 6) ? h : (M == 9) ? -s1 : 0.f;
  v2 w = {re, INV ? im : -im};
  return w;
}

// In-place 16-point DFT. Input v[n], n = 0..15; output X[k] is left in register v[4*(k&3) + (k>>2)]
// (base-4 digit reversal) -- callers index outputs through rev16().  64 + 16 packed instructions.
__device__ __forceinline__ constexpr int rev16(int k) { return 4 * (k & 3) + (k >> 2); }

// a -+ i*(h*b) with the real scale h = hh.lo = -hh.hi folded into one packed FMA (hh in an SGPR pair):
//   sub_ih(a, b, hh) = (a.re + h b.im, a.im - h b.re)      add_ih(a, b, hh) = (a.re - h b.im, a.im + h b.re)
__device__ __forceinline__ v2 sub_ih(v2 a, v2 b, v2 hh) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(b), "s"(hh), "v"(a));
  return r;
}
__device__ __forceinline__ v2 add_ih(v2 a, v2 b, v2 hh) {
  v2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(b), "s"(hh), "v"(a));
  return r;
}

// The four W16^2 / W16^6 twiddles are (1 -+ i)/sqrt2 rotations: the rotation is one packed add with operand swaps, and its
// 1/sqrt2 rides on the butterfly that consumes it as a packed FMA (fma_lo / fma_hi with hh = (h, -h)) -- 4 instructions per
// 16-point transform less than multiplying them out (76 instead of 80).
template <bool INV> __device__ __forceinline__ void dft16(v2 (&v)[16]) {
#pragma unroll
  for (int n0 = 0; n0 < 4; n0++) dft4<INV, false>(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);   // -> A[n0][k0] at v[n0+4k0]
  const v2 hh = {0.70710678118654752f, -0.70710678118654752f};
  v[5] = cmul_k(v[5], w16<INV, 1>());   v[13] = cmul_k(v[13], w16<INV, 3>());
  v[7] = cmul_k(v[7], w16<INV, 3>());   v[15] = cmul_k(v[15], w16<INV, 9>());
  // v6 W16^2 = h c1, v9 W16^2 = h b2, v11 W16^6 = -h d2, v14 W16^6 = -h c3   (forward W16^2 = h(1 - i), W16^6 = -h(1 + i); inverse: conjugates)
  const v2 c1 = INV ? add_i(v[6], v[6]) : sub_i(v[6], v[6]);
  const v2 b2 = INV ? add_i(v[9], v[9]) : sub_i(v[9], v[9]);
  const v2 d2 = INV ? sub_i(v[11], v[11]) : add_i(v[11], v[11]);
  const v2 c3 = INV ? sub_i(v[14], v[14]) : add_i(v[14], v[14]);
  dft4<INV, false>(v[0], v[1], v[2], v[3]);
  {  // row k0 = 1: a = v4, b = v5, c = h c1, d = v7
    const v2 s0 = fma_lo(v[4], c1, hh), d0 = fma_hi(v[4], c1, hh);
    const v2 s1 = v[5] + v[7], t = v[5] - v[7];
    v[4] = s0 + s1;
    v[6] = s0 - s1;
    v[5] = INV ? add_i(d0, t) : sub_i(d0, t);
    v[7] = INV ? sub_i(d0, t) : add_i(d0, t);
  }
  {  // row k0 = 2: a = v8, b = h b2, c = v10 with its -/+i (W16^4) folded into the adds, d = -h d2
    const v2 s0 = INV ? add_i(v[8], v[10]) : sub_i(v[8], v[10]);
    const v2 d0 = INV ? sub_i(v[8], v[10]) : add_i(v[8], v[10]);
    const v2 s1 = b2 - d2, t = b2 + d2;                    // true values: h s1, h t
    v[8] = fma_lo(s0, s1, hh);
    v[10] = fma_hi(s0, s1, hh);
    v[9] = INV ? add_ih(d0, t, hh) : sub_ih(d0, t, hh);
    v[11] = INV ? sub_ih(d0, t, hh) : add_ih(d0, t, hh);
  }
  {  // row k0 = 3: a = v12, b = v13, c = -h c3, d = v15
    const v2 s0 = fma_hi(v[12], c3, hh), d0 = fma_lo(v[12], c3, hh);
    const v2 s1 = v[13] + v[15], t = v[13] - v[15];
    v[12] = s0 + s1;
    v[14] = s0 - s1;
    v[13] = INV ? add_i(d0, t) : sub_i(d0, t);
    v[15] = INV ? sub_i(d0, t) : add_i(d0, t);
  }                                                                                           // -> X[k0+4k1] at v[4k0+k1]
}

// cos/sin(2 pi m / P) for the odd primes used as outer radices
template <int P> struct Trig;
template <> struct Trig<31> {
  static constexpr float c[16] = {1.f, 0.97952994125249448f, 0.9189578116202306f, 0.82076344120727629f, 0.68896691907568663f,
                                  0.52896401032696239f, 0.34730525284482028f, 0.1514277775045767f, -0.050649168838712642f,
                                  -0.25065253225872042f, -0.44039415155763439f, -0.61210598254766257f, -0.75875812269279086f,
                                  -0.87434661614458209f, -0.95413925640004882f, -0.99486932339189504f};
  static constexpr float s[16] = {0.f, 0.20129852008866006f, 0.39435585511331855f, 0.57126821509479231f, 0.72479278722911988f,
                                  0.84864425749475092f, 0.93775213214708042f, 0.98846832432811138f, 0.99871650717105276f,
                                  0.96807711886620429f, 0.89780453957074158f, 0.79077573693769887f, 0.65137248272222226f,
                                  0.48530196253108104f, 0.29936312297335804f, 0.10116832198743272f};
};
template <> struct Trig<3> {
  static constexpr float c[2] = {1.f, -0.5f};
  static constexpr float s[2] = {0.f, 0.866025404f};
};
template <> struct Trig<11> {
  static constexpr float c[6] = {1.f, 0.841253533f, 0.415415013f, -0.142314838f, -0.654860734f, -0.959492974f};
  static constexpr float s[6] = {0.f, 0.540640817f, 0.909631995f, 0.989821442f, 0.755749574f, 0.281732557f};
};
template <> struct Trig<5> {
  static constexpr float c[3] = {1.f, 0.30901699437494742f, -0.80901699437494742f};
  static constexpr float s[3] = {0.f, 0.95105651629515357f, 0.58778525229247313f};
};

// P-point DFT (P odd prime) through the conjugate symmetry of W_P: with s_n = x[n] + x[P-n], d_n = x[n] - x[P-n]
//   A_k = x[0] + sum_{n=1..h} cos(2 pi n k/P) s_n,   B_k = sum_{n=1..h} sin(2 pi n k/P) d_n,   h = (P-1)/2
//   forward X[k] = A_k - i B_k, X[P-k] = A_k + i B_k (inverse: signs swapped)
// Real constant x complex value = one v_pk_fma_f32 with the constant broadcast from an SGPR: (P-1) + 2 h^2 + ... packed
// instructions, a quarter of the P x P complex products.  Outputs go to sink(k, value) as they are produced.
template <int P, bool INV, class Sink> __device__ __forceinline__ void dft_prime(const v2 (&x)[P], Sink&& sink) {
  constexpr int H = (P - 1) / 2;
  v2 s[H + 1], d[H + 1];
  v2 sum = x[0];
#pragma unroll
  for (int n = 1; n <= H; n++) {
    s[n] = x[n] + x[P - n];
    d[n] = x[n] - x[P - n];
    sum += s[n];
  }
  sink(0, sum);
#pragma unroll
  for (int k = 1; k <= H; k++) {
    v2 A = x[0], B;
#pragma unroll
    for (int n = 1; n <= H; n++) {
      const int m = (n * k) % P;
      const int mm = m <= H ? m : P - m;
      const v2 cs = {Trig<P>::c[mm], Trig<P>::s[mm]};
      A = fma_lo(A, s[n], cs);
      if (n == 1) B = (m <= H) ? mul_hi(d[n], cs) : mul_hi_neg(d[n], cs);
      else B = (m <= H) ? fma_hi(B, d[n], cs) : fms_hi(B, d[n], cs);
    }
    sink(k, INV ? add_i(A, B) : sub_i(A, B));
    sink(P - k, INV ? sub_i(A, B) : add_i(A, B));
  }
}

// Generic R-point DFT used as the OUTER stage of the split engines: x[0..R) -> sink(k, X[k]).
template <int R, bool INV> struct OuterDft;
template <bool INV> struct OuterDft<31, INV> {
  template <class Sink> static __device__ __forceinline__ void run(v2 (&x)[31], Sink&& sink) { dft_prime<31, INV>(x, sink); }
};
template <bool INV> struct OuterDft<4, INV> {
  template <class Sink> static __device__ __forceinline__ void run(v2 (&x)[4], Sink&& sink) {
    dft4<INV, false>(x[0], x[1], x[2], x[3]);
    sink(0, x[0]); sink(1, x[1]); sink(2, x[2]); sink(3, x[3]);
  }
};
template <bool INV> struct OuterDft<16, INV> {
  template <class Sink> static __device__ __forceinline__ void run(v2 (&x)[16], Sink&& sink) {
    dft16<INV>(x);
#pragma unroll
    for (int k = 0; k < 16; k++) sink(k, x[rev16(k)]);
  }
};


// cos/sin(2 pi m / R) over the full period for the composite outer radices 20 = 4*5 and 40 = 8*5
template <int R> struct TrigN;
template <> struct TrigN<8> {
  static constexpr float c[8] = {1.f, 0.707106781f, 0.f, -0.707106781f, -1.f, -0.707106781f, 0.f, 0.707106781f};
  static constexpr float s[8] = {0.f, 0.707106781f, 1.f, 0.707106781f, 0.f, -0.707106781f, -1.f, -0.707106781f};
};
template <> struct TrigN<9> {
  static constexpr float c[9] = {1.f, 0.766044443f, 0.173648178f, -0.5f, -0.939692621f, -0.939692621f, -0.5f, 0.173648178f, 0.766044443f};
  static constexpr float s[9] = {0.f, 0.64278761f, 0.984807753f, 0.866025404f, 0.342020143f, -0.342020143f, -0.866025404f, -0.984807753f, -0.64278761f};
};
template <> struct TrigN<16> {
  static constexpr float c[16] = {1.f, 0.923879533f, 0.707106781f, 0.382683432f, 0.f, -0.382683432f, -0.707106781f, -0.923879533f, -1.f, -0.923879533f, -0.707106781f, -0.382683432f, 0.f, 0.382683432f, 0.707106781f, 0.923879533f};
  static constexpr float s[16] = {0.f, 0.382683432f, 0.707106781f, 0.923879533f, 1.f, 0.923879533f, 0.707106781f, 0.382683432f, 0.f, -0.382683432f, -0.707106781f, -0.923879533f, -1.f, -0.923879533f, -0.707106781f, -0.382683432f};
};
template <> struct TrigN<20> {
  static constexpr float c[20] = {1.f, 0.951056516f, 0.809016994f, 0.587785252f, 0.309016994f, 0.f, -0.309016994f, -0.587785252f, -0.809016994f, -0.951056516f, -1.f, -0.951056516f, -0.809016994f, -0.587785252f, -0.309016994f, 0.f, 0.309016994f, 0.587785252f, 0.809016994f, 0.951056516f};
  static constexpr float s[20] = {0.f, 0.309016994f, 0.587785252f, 0.809016994f, 0.951056516f, 1.f, 0.951056516f, 0.809016994f, 0.587785252f, 0.309016994f, 0.f, -0.309016994f, -0.587785252f, -0.809016994f, -0.951056516f, -1.f, -0.951056516f, -0.809016994f, -0.587785252f, -0.309016994f};
};
template <> struct TrigN<40> {
  static constexpr float c[40] = {1.f, 0.987688341f, 0.951056516f, 0.891006524f, 0.809016994f, 0.707106781f, 0.587785252f, 0.4539905f, 0.309016994f, 0.156434465f, 0.f, -0.156434465f, -0.309016994f, -0.4539905f, -0.587785252f, -0.707106781f, -0.809016994f, -0.891006524f, -0.951056516f, -0.987688341f, -1.f, -0.987688341f, -0.951056516f, -0.891006524f, -0.809016994f, -0.707106781f, -0.587785252f, -0.4539905f, -0.309016994f, -0.156434465f, 0.f, 0.156434465f, 0.309016994f, 0.4539905f, 0.587785252f, 0.707106781f, 0.809016994f, 0.891006524f, 0.951056516f, 0.987688341f};
  static constexpr float s[40] = {0.f, 0.156434465f, 0.309016994f, 0.4539905f, 0.587785252f, 0.707106781f, 0.809016994f, 0.891006524f, 0.951056516f, 0.987688341f, 1.f, 0.987688341f, 0.951056516f, 0.891006524f, 0.809016994f, 0.707106781f, 0.587785252f, 0.4539905f, 0.309016994f, 0.156434465f, 0.f, -0.156434465f, -0.309016994f, -0.4539905f, -0.587785252f, -0.707106781f, -0.809016994f, -0.891006524f, -0.951056516f, -0.987688341f, -1.f, -0.987688341f, -0.951056516f, -0.891006524f, -0.809016994f, -0.707106781f, -0.587785252f, -0.4539905f, -0.309016994f, -0.156434465f};
};

// W_R^m as an SGPR-pair constant (forward exp(-2 pi i m/R); INV: conjugate)
template <int R, bool INV> __device__ __forceinline__ v2 wconst(int m) {
  const v2 w = {TrigN<R>::c[m % R], INV ? TrigN<R>::s[m % R] : -TrigN<R>::s[m % R]};
  return w;
}

// 8-point DFT, natural order in and out (two DFT4 + W8 combine)
template <bool INV> __device__ __forceinline__ void dft8(v2 (&x)[8]) {
  v2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
  dft4<INV, false>(e0, e1, e2, e3);
  dft4<INV, false>(o0, o1, o2, o3);
  o1 = cmul_k(o1, wconst<8, INV>(1));
  o3 = cmul_k(o3, wconst<8, INV>(3));
  x[0] = e0 + o0;  x[4] = e0 - o0;
  x[1] = e1 + o1;  x[5] = e1 - o1;
  x[2] = INV ? add_i(e2, o2) : sub_i(e2, o2);      // W8^2 = -/+ i
  x[6] = INV ? sub_i(e2, o2) : add_i(e2, o2);
  x[3] = e3 + o3;  x[7] = e3 - o3;
}

// R = RA * 5 (RA = 4 or 8): n = 5 a + b, k = ka + RA kb:
//   X[ka + RA kb] = sum_b W5^{b kb} ( W_R^{b ka} sum_a x[5a + b] W_RA^{a ka} )
template <int RA, bool INV, class Sink> __device__ __forceinline__ void dft_ra5(v2 (&x)[RA * 5], Sink&& sink) {
  constexpr int R = RA * 5;
#pragma unroll
  for (int b = 0; b < 5; b++) {
    if (RA == 4) {
      dft4<INV, false>(x[b], x[5 + b], x[10 + b], x[15 + b]);                    // T[b][ka] left at x[5 ka + b]
    } else {
      v2 t[8];
#pragma unroll
      for (int a = 0; a < 8; a++) t[a] = x[5 * a + b];
      dft8<INV>(t);
#pragma unroll
      for (int a = 0; a < 8; a++) x[5 * a + b] = t[a];
    }
    if (b > 0) {
#pragma unroll
      for (int ka = 1; ka < RA; ka++) x[5 * ka + b] = cmul_k(x[5 * ka + b], wconst<R, INV>(b * ka));
    }
  }
#pragma unroll
  for (int ka = 0; ka < RA; ka++) {
    const v2 col[5] = {x[5 * ka], x[5 * ka + 1], x[5 * ka + 2], x[5 * ka + 3], x[5 * ka + 4]};
    dft_prime<5, INV>(col, [&](int kb, v2 val) { sink(ka + RA * kb, val); });
  }
}

template <bool INV> struct OuterDft<20, INV> {
  template <class Sink> static __device__ __forceinline__ void run(v2 (&x)[20], Sink&& sink) { dft_ra5<4, INV>(x, sink); }
};
template <bool INV> struct OuterDft<40, INV> {
  template <class Sink> static __device__ __forceinline__ void run(v2 (&x)[40], Sink&& sink) { dft_ra5<8, INV>(x, sink); }
};

// In-place small DFTs in natural order, the butterflies of the prime-factor engine's inner passes (radices 2, 4, 5, 9, 11 and their coprime products)
template <int R, bool INV> struct SmallDft {
  static __device__ __forceinline__ void run(v2 (&x)[R]) {            // odd primes through the symmetric form
    v2 y[R];
    dft_prime<R, INV>(x, [&](int k, v2 val) { y[k] = val; });
#pragma unroll
    for (int k = 0; k < R; k++) x[k] = y[k];
  }
};
template <bool INV> struct SmallDft<2, INV> {
  static __device__ __forceinline__ void run(v2 (&x)[2]) { const v2 a = x[0] + x[1], b = x[0] - x[1]; x[0] = a; x[1] = b; }
};
template <bool INV> struct SmallDft<4, INV> {
  static __device__ __forceinline__ void run(v2 (&x)[4]) { dft4<INV, false>(x[0], x[1], x[2], x[3]); }
};
template <bool INV> struct SmallDft<9, INV> {                          // 9 = 3 x 3: n = 3 n1 + n2, k = k1 + 3 k2
  static __device__ __forceinline__ void run(v2 (&x)[9]) {
    v2 t[3][3];                                                          // t[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 3; n2++) {
      const v2 col[3] = {x[n2], x[3 + n2], x[6 + n2]};
      dft_prime<3, INV>(col, [&](int k1, v2 val) { t[n2][k1] = (n2 * k1) ? cmul_k(val, wconst<9, INV>(n2 * k1)) : val; });
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; k1++) {
      const v2 row[3] = {t[0][k1], t[1][k1], t[2][k1]};
      dft_prime<3, INV>(row, [&](int k2, v2 val) { x[k1 + 3 * k2] = val; });
    }
  }
};

// Coprime composites by the prime-factor (Good-Thomas) mapping: R = N1 N2 with gcd 1,
//   input  n = (N2 n1 + N1 n2) mod R,   output k = (A k1 + Bk k2) mod R,  A = N2 (N2^-1 mod N1), Bk = N1 (N1^-1 mod N2)
// so that W_R^{nk} = W_N1^{n1 k1} W_N2^{n2 k2}: two layers of small DFTs and no twiddle factors at all (a register
// permutation at compile time).  10 = 2 x 5 and 20 = 4 x 5 are the middle dimensions of the prime-factor engine (gacq_pfa.hip): 990 = 11 * 10 * 9, 1980 = 11 * 20 * 9.
template <int N1, int N2, int A, int Bk, bool INV> struct PfaDft {
  static constexpr int R = N1 * N2;
  static __device__ __forceinline__ void run(v2 (&x)[R]) {
    v2 u[N1][N2];                                                        // u[k1][n2]
#pragma unroll
    for (int n2 = 0; n2 < N2; n2++) {
      v2 col[N1];
#pragma unroll
      for (int n1 = 0; n1 < N1; n1++) col[n1] = x[(N2 * n1 + N1 * n2) % R];
      SmallDft<N1, INV>::run(col);
#pragma unroll
      for (int k1 = 0; k1 < N1; k1++) u[k1][n2] = col[k1];
    }
#pragma unroll
    for (int k1 = 0; k1 < N1; k1++) {
      SmallDft<N2, INV>::run(u[k1]);
#pragma unroll
      for (int k2 = 0; k2 < N2; k2++) x[(A * k1 + Bk * k2) % R] = u[k1][k2];
    }
  }
};
template <bool INV> struct SmallDft<10, INV> : PfaDft<2, 5, 5, 6, INV> {};      // 5^-1 mod 2 = 1, 2^-1 mod 5 = 3
template <bool INV> struct SmallDft<20, INV> : PfaDft<4, 5, 5, 16, INV> {};     // 5^-1 mod 4 = 1, 4^-1 mod 5 = 4

// w^k for k = 0..43 from base-4 digits: w^k = p[k & 3] * q[k >> 2], p[a] = w^a, q[b] = w^(4b); multiplication depth <= 5
struct TwPow {
  v2 p[4], q[11];
  template <int KMAX> __device__ __forceinline__ void init(v2 w) {
    p[1] = w;
    p[2] = cmul(w, w);
    p[3] = cmul(p[2], w);
    q[1] = cmul(p[2], p[2]);
    if (KMAX >= 8) q[2] = cmul(q[1], q[1]);
    if (KMAX >= 12) q[3] = cmul(q[2], q[1]);
    if (KMAX >= 16) q[4] = cmul(q[2], q[2]);
    if (KMAX >= 20) q[5] = cmul(q[4], q[1]);
    if (KMAX >= 24) q[6] = cmul(q[3], q[3]);
    if (KMAX >= 28) q[7] = cmul(q[4], q[3]);
    if (KMAX >= 32) q[8] = cmul(q[4], q[4]);
    if (KMAX >= 36) q[9] = cmul(q[8], q[1]);
    if (KMAX >= 40) q[10] = cmul(q[5], q[5]);
  }
  // v * w^k, k compile-time after unrolling
  __device__ __forceinline__ v2 apply(v2 v, int k) const {
    const int a = k & 3, b = k >> 2;
    if (a) v = cmul(v, p[a]);
    if (b) v = cmul(v, q[b]);
    return v;
  }
};

}  // namespace gacq
