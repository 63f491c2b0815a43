// Helpers shared by the LDS-resident transform kernels (gacq_ldsfft.hip: N = 4096 and the inner rows of the split engines;
// gacq_lds16k.hip: N = 16384): wave-level first maximum, one-instruction LDS accesses, the vmcnt-preserving barrier.
#pragma once
#include "gacq_common.h"
#include "gacq_cplx.h"

namespace gacq {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ v2 ld2(const float2* p) { return *reinterpret_cast<const v2*>(p); }

// (max, first argmax) of the kR x 64 magnitudes a wave holds; lane l of the wave holds lag base + mult * l + kstride * k in
// m[k], with kstride > 63 * mult so that lags grow with k first.  The maximum is found first -- per lane with max3, over the
// wave on DPP -- and only then located: for every k one v_cmp against the wave-uniform maximum gives a lane mask in SGPRs; the
// smallest k with a non-empty mask and its lowest lane are the smallest lag attaining the maximum (np.argmax returns the first
// maximum, acquire-gps-l1.py:34).  24 VALU instructions instead of the 47 of a running (value, index) pair per lane; the
// bookkeeping runs on the scalar unit.  Magnitudes are >= 0, so their bit patterns order like the values.
template <int kR>
__device__ __forceinline__ void wave_first_max(const float (&m)[kR], unsigned base, unsigned mult, unsigned kstride, float tie_scale,
                                               float& wmaxf, unsigned& widx) {
  float lmax = __builtin_fmaxf(__builtin_fmaxf(m[0], m[1]), m[2]);
#pragma unroll
  for (int k = 3; k + 1 < kR; k += 2) lmax = __builtin_fmaxf(__builtin_fmaxf(lmax, m[k]), m[k + 1]);
  lmax = __builtin_fmaxf(lmax, m[kR - 1]);
  wmaxf = __builtin_bit_cast(float, wave_max_u32(__builtin_bit_cast(unsigned, lmax)));
  // Tie-safe locations: the compare runs against thr = (1 - eps) * maximum instead of the maximum itself.  When exactly one entry
  // passes -- all but about one row in 10^4 -- it is the maximum and its mask is its location; the masks of all k are folded on the
  // scalar unit (seen: lanes with an entry, dup: lanes with two) to tell.  Otherwise the row is tagged ambiguous (kTieBit; the
  // Doppler scan decides whether it matters and has it re-evaluated in complex128) and the exact first maximum is located the
  // plain way in a wave-uniform branch.  tie_scale == 1 degenerates to the equality compare.
  const float thr = wmaxf * tie_scale;
  unsigned long long seen = 0, dup = 0;
  unsigned kk = 0;
#pragma unroll
  for (int k = kR - 1; k >= 0; k--) {
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(m[k] >= thr);
    dup |= seen & mk;
    seen |= mk;
    if (mk) kk = (unsigned)k;
  }
  // exactly one entry passed (one k with a non-empty mask, one lane in it): `seen` is that lane's bit.  No entry at all -- a NaN or
  // Inf sample made the maximum NaN, nothing compares >= -- leaves the 0xffffffff sentinel (ctz of 0 is undefined)
  widx = seen ? kstride * kk + base + mult * (unsigned)__builtin_ctzll(seen) : 0xffffffffu;
  if ((dup | (seen & (seen - 1))) != 0) {
    widx = 0xffffffffu;
#pragma unroll
    for (int k = kR - 1; k >= 0; k--) {
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(m[k] == wmaxf);
      if (mk) widx = kstride * k + base + mult * (unsigned)__builtin_ctzll(mk);
    }
    widx |= (unsigned)kTieBit;
  }
}

// LDS accesses of the exchanges, one per instruction (GACQ_UNPAIR, gacq_cplx.h): LDS_LD for every read; LDS_ST1 for the writes of the
// 16384-point transforms (the 4096-point ones are faster with their writes left to the compiler's ds_write2_b64 pairing)
__device__ __forceinline__ v2 lds_ld1(const v2& x) { const v2 r = x; GACQ_UNPAIR(); return r; }
#define LDS_LD(x) lds_ld1(x)
#define LDS_ST1(dst, val) do { (dst) = (val); GACQ_UNPAIR(); } while (0)

// workgroup barrier that does not drain the vector-memory counter: an LDS-DMA in flight survives it (__syncthreads() would
// wait for vmcnt(0) first).  lgkmcnt(0): this wave's LDS stores have been performed before the others are released.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace gacq
