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

#include <cmath>
#include <cstdlib>

using namespace gacq;

namespace {

constexpr int kR = 16;                 // points per lane
constexpr int kLdsN = 4096;            // supported length (this round)
constexpr int kPitch = 257;            // exchange-2 row pitch in complex elements
constexpr int kLdsElems = 16 * kPitch; // 4112 complex = 32.9 KB -> 4 workgroups per CU

// Ablation builds for profiling only (tools/ablate.sh): bit 0 drops the LDS traffic, bit 1 the barriers, bit 2 the
// magnitude/reduce tail, bit 3 the twiddle multiplies, bit 4 the X loads of lds16k_correlate_kernel, bit 5 the workgroup-wide
// barriers of fft16k; bit 6 keeps engine 4's Z' rows in natural order (A/B against the lane-pair stores, correct results with
// -DGACQ_ABL_SPLIT=128 in gacq_split.hip).  Results are wrong by construction; never set in the product build.
#ifndef GACQ_ABL
#define GACQ_ABL 0
#endif

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

// (max, first argmax) of the 16 x 64 magnitudes a wave holds; lane l of the wave holds lag base + mult * l + kstride * k in
// m[k], with kstride > 63 * mult so that lags grow with k first.  The maximum is found first -- per lane with max3, over the
// wave on DPP -- and only then located: for every k one v_cmp against the wave-uniform maximum gives a lane mask in SGPRs; the
// smallest k with a non-empty mask and its lowest lane are the smallest lag attaining the maximum (np.argmax returns the first
// maximum, acquire-gps-l1.py:34).  24 VALU instructions instead of the 47 of a running (value, index) pair per lane; the
// bookkeeping runs on the scalar unit.  Magnitudes are >= 0, so their bit patterns order like the values.
__device__ __forceinline__ void wave_first_max(const float (&m)[kR], unsigned base, unsigned mult, unsigned kstride, float& wmaxf,
                                               unsigned& widx) {
  float lmax = __builtin_fmaxf(__builtin_fmaxf(m[0], m[1]), m[2]);
#pragma unroll
  for (int k = 3; k + 1 < kR; k += 2) lmax = __builtin_fmaxf(__builtin_fmaxf(lmax, m[k]), m[k + 1]);
  lmax = __builtin_fmaxf(lmax, m[kR - 1]);
  wmaxf = __builtin_bit_cast(float, wave_max_u32(__builtin_bit_cast(unsigned, lmax)));
  widx = 0xffffffffu;
#pragma unroll
  for (int k = kR - 1; k >= 0; k--) {                                            // descending: the last assignment is the smallest k
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(m[k] == wmaxf);
    if (mk) widx = kstride * k + base + mult * (unsigned)__builtin_ctzll(mk);
  }
}

// Length-4096 transform of the 16 values per lane. In: v[j] = x[t + 256 j]; out: v[rev16(k2)] = X[t + 256 k2].
// wa = W_4096^t, wb = W_256^(t & 15) (forward values; conjugated here when INV).
// PRE: bit 0 = the pass-1 powers (W_4096^t)^k, bit 1 = the pass-2 powers (W_256^(t&15))^k come precomputed in *pa / *pb (already
// conjugated for an inverse transform) instead of being rebuilt from wa / wb by 14 complex products per pass.
template <bool INV, int PRE = 0>
__device__ __forceinline__ void fft4096(v2 (&v)[kR], v2* lds, v2 wa, v2 wb, const v2 (*pa)[15] = nullptr, const v2 (*pb)[15] = nullptr,
                                        int t = -1) {
  if (t < 0) t = threadIdx.x;                       // lane index within the 256-lane group that owns this transform
  if (INV && !(PRE & 1)) wa.y = -wa.y;
  if (INV && !(PRE & 2)) wb.y = -wb.y;
  dft16<INV>(v);
  if (!(GACQ_ABL & 8)) { if (PRE & 1) apply_table(v, *pa); else apply_powers(v, wa); }
  {  // exchange 1: (n0,n1;k0) -> (n0,k0;n1)
    const int wbase = (t & 15) + 256 * (t >> 4);
    if (!(GACQ_ABL & 1)) {
_Pragma("unroll") for (int k = 0; k < kR; k++) lds[wbase + 16 * k] = v[rev16(k)];
    }
    if (!(GACQ_ABL & 2)) __syncthreads();
    if (!(GACQ_ABL & 1)) {
_Pragma("unroll") for (int j = 0; j < kR; j++) v[j] = lds[t + 256 * j];
    }
  }
  dft16<INV>(v);
  if (!(GACQ_ABL & 8)) { if (PRE & 2) apply_table(v, *pb); else apply_powers(v, wb); }
  if (!(GACQ_ABL & 2)) __syncthreads();   // all exchange-1 reads done before the buffer is reused
  {  // exchange 2: (n0,k0;k1) -> (k0,k1;n0)
    const int wbase = (t >> 4) + kPitch * (t & 15);
    if (!(GACQ_ABL & 1)) {
_Pragma("unroll") for (int k = 0; k < kR; k++) lds[wbase + 16 * k] = v[rev16(k)];
    }
    if (!(GACQ_ABL & 2)) __syncthreads();
    if (!(GACQ_ABL & 1)) {
_Pragma("unroll") for (int j = 0; j < kR; j++) v[j] = lds[t + kPitch * j];
    }
  }
  dft16<INV>(v);
}

__device__ __forceinline__ v2 ld2(const float2* p) { return *reinterpret_cast<const v2*>(p); }

// Lane-pair layout used for X and C_p inside this engine: natural index i = t + 256 j (t = lane, j = 0..15)
// is stored at (j>>1)*512 + 2 t + (j&1), so one lane's values for j = 2jp, 2jp+1 are 16 contiguous bytes and
// a wave reads/writes 1 KiB per instruction.  Rows are fetched with buffer loads: the row base lives in an
// SGPR resource, the lane offset (16 t) in one VGPR, the piece offset in an SGPR -- no VALU address math.
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

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
template <bool DUMP>
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
  const double f = freq[fd];
  const float2* src = x + e * epoch_stride + (size_t)b * n;
  v2 v[kR], w[kR];
#pragma unroll
  for (int j = 0; j < kR; j++) {
    const int i = t + 256 * j;
    // table NCO, index in fp64 exactly as numpy: floor((0 + f*i)*1024) mod 1024   (gnsstools/nco.py:6-9)
    const int k = nco_index(f, (int)i);
    if (DUMP) { reinterpret_cast<int*>(X)[row * (long)kLdsN + i] = k; continue; }
    v[j] = ld2(src + i);
    w[j] = ld2(nco_tab + k);
  }
  if (DUMP) return;
  const v2 twa = ld2(tw + t), twb = ld2(tw + 16 * (t & 15));
#pragma unroll
  for (int j = 0; j < kR; j++) v[j] = cmul(v[j], w[j]);
  fft4096<false>(v, lds, twa, twb);
  float2* dst = X + row * (long)kLdsN;
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++) {
    const v2 a = v[rev16(2 * jp)], b = v[rev16(2 * jp + 1)];
    // store conj(FFT) in the lane-pair layout: np.conj(fft.fft(b))  acquire-gps-l1.py:32
    *reinterpret_cast<float4*>(dst + jp * 512 + 2 * t) = make_float4(a.x, -a.y, b.x, -b.y);
  }
}

// ======================================================================================================
// N = 16384 = 4 x 4096 in ONE 1024-thread workgroup (B1I/B2I padded, GLONASS L1/L2): 16 points per lane,
//   n = 4096 na + m, k = ka + 4 km:  X[ka + 4 km] = FFT4096_m( W_N^{m ka} * sum_na x[4096 na + m] W_4^{na ka} )[km]
//   pass 0  lane t holds x[t + 1024 j], j = 4 na + jj (m = t + 1024 jj): four DFT-4 over na, twiddle W_N^{m ka}
//   exchange 0 through LDS: (ka, m) -> region ka, lane t' = m mod 256 gets m = t' + 256 j
//   then each 256-lane group runs the 4096-point transform of its region (same code as the N = 4096 engine).
// LDS: 4 regions x 32.9 KB = 131.6 KB -> one workgroup (16 waves, 4 per SIMD) per CU.  Output: lane (ka, t'') holds
// X[(ka + 4 t'') + 1024 k2] in register rev16(k2), i.e. exactly the input convention with lane index ka + 4 t''.
constexpr int kBig = 16384;
constexpr int kBigThreads = 1024;
constexpr int kBigLdsBytes = 4 * kLdsElems * (int)sizeof(v2);

template <bool INV>
__device__ __forceinline__ void fft16k(v2 (&v)[kR], v2* lds, const float2* __restrict__ twn /* W_16384^m, m < 1024 */, v2 base) {
  const int t = threadIdx.x;
  const int g = t >> 8, tl = t & 255;
  // issue the table loads of the inner transform before any asm
  v2 wa = ld2(twn + 4 * tl), wb = ld2(twn + 64 * (tl & 15));
  if (INV) base.y = -base.y;
  const v2 p2 = cmul(base, base), p3 = cmul(p2, base);
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
    dft4<INV, false>(v[jj], v[jj + 4], v[jj + 8], v[jj + 12]);                 // -> ka at v[jj + 4 ka]
    v2 a = v[jj + 4], b = v[jj + 8], c = v[jj + 12];
    if (jj) {                                                                  // W_16^{jj ka} part of W_N^{(t + 1024 jj) ka}
      a = cmul_k(a, wconst<16, INV>(jj));
      b = cmul_k(b, wconst<16, INV>(2 * jj));
      c = cmul_k(c, wconst<16, INV>(3 * jj));
    }
    v[jj + 4] = cmul(a, base);
    v[jj + 8] = cmul(b, p2);
    v[jj + 12] = cmul(c, p3);
  }
#pragma unroll
  for (int ka = 0; ka < 4; ka++) {
#pragma unroll
    for (int jj = 0; jj < 4; jj++) lds[ka * kLdsElems + t + 1024 * jj] = v[jj + 4 * ka];
  }
  if (!(GACQ_ABL & 32)) __syncthreads();
  v2* region = lds + g * kLdsElems;
#pragma unroll
  for (int j = 0; j < kR; j++) v[j] = region[tl + 256 * j];
  if (!(GACQ_ABL & 32)) __syncthreads();             // exchange-0 reads complete before the regions are reused
  fft4096<INV>(v, region, wa, wb, nullptr, nullptr, tl);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t big_rsrc(const float2* row) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, kBig * (int)sizeof(float2), 0x00020000);
}
__device__ __forceinline__ void ld_pair_big(__amdgpu_buffer_rsrc_t r, unsigned lane_off, int jp, v2& a, v2& b) {
  const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, (unsigned)jp * 16384u, 0));
  a = q.xy;
  b = q.zw;
}

// forward: one workgroup per (e, f, d, b) row; output conj(FFT) in the 1024-lane pair layout (j>>1)*2048 + 2 lane + (j&1)
template <bool DUMP>
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
  const double f = freq[fd];
  const float2* src = x + e * epoch_stride + (size_t)b * n;
  v2 v[kR], w[kR];
#pragma unroll
  for (int j = 0; j < kR; j++) {
    const int i = t + 1024 * j;
    const int k = nco_index(f, (int)i);   // gnsstools/nco.py:6-9
    if (DUMP) { reinterpret_cast<int*>(X)[row * (long)kBig + i] = k; continue; }
    v[j] = ld2(src + i);
    w[j] = ld2(nco_tab + k);
  }
  if (DUMP) return;
  const v2 base = ld2(twn + t);
#pragma unroll
  for (int j = 0; j < kR; j++) v[j] = cmul(v[j], w[j]);
  fft16k<false>(v, lds, twn, base);
  // lane (ka, t'') holds X[(ka + 4 t'') + 1024 k2]: storing straight from here would write 16-byte pieces at a 64-byte stride
  // (quarter cache lines).  One more pass through LDS puts lane t back on natural indices t + 1024 j, so every wave stores
  // full 1 KiB runs of the lane-pair layout.
  const int lane = (t >> 8) + 4 * (t & 255);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kR; k++) lds[lane + 1024 * k] = v[rev16(k)];
  __syncthreads();
  float2* dst = X + row * (long)kBig;
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++) {
    const v2 a = lds[t + 1024 * (2 * jp)], c = lds[t + 1024 * (2 * jp + 1)];
    *reinterpret_cast<float4*>(dst + jp * 2048 + 2 * t) = make_float4(a.x, -a.y, c.x, -c.y);
  }
}

__global__ __launch_bounds__(kBigThreads) void lds16k_permute_kernel(const float2* __restrict__ nat, float2* __restrict__ perm) {
  const long row = blockIdx.x;
  const int t = threadIdx.x;
  const float2* src = nat + row * kBig;
  float2* dst = perm + row * kBig;
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++) {
    const float2 a = src[t + 1024 * (2 * jp)], b = src[t + 1024 * (2 * jp + 1)];
    *reinterpret_cast<float4*>(dst + jp * 2048 + 2 * t) = make_float4(a.x, a.y, b.x, b.y);
  }
}

// correlate: workgroup = (epoch, Doppler, chunk of items); per item: sum_b |IFFT(C_p * X_b)|/N -> (max, argmax, sum)
__global__ __launch_bounds__(kBigThreads) void lds16k_correlate_kernel(const float2* __restrict__ X, const float2* __restrict__ C,
                                                                        const int* __restrict__ items, const int* __restrict__ fset,
                                                                        const float2* __restrict__ twn, RowRec* __restrict__ rows,
                                                                        int E, int P, int F, int D, int B, int pch, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2* lds = reinterpret_cast<v2*>(smem);
  __shared__ float s_peak[kBigThreads / 64];
  __shared__ int s_idx[kBigThreads / 64];
  __shared__ double s_sum[kBigThreads / 64];
  const int t = threadIdx.x;
  const int xcd = blockIdx.x & 7;
  const unsigned j = blockIdx.x >> 3;
  const unsigned u = (j / (unsigned)nchunk) * 8 + xcd;             // (epoch, Doppler) unit -> XCD, as in lds_correlate_kernel
  if (u >= (unsigned)E * (unsigned)D) return;
  const long e = u / (unsigned)D;
  const int d = (int)(u % (unsigned)D);
  const int p0 = (int)(j % (unsigned)nchunk) * pch;
  const int p1 = min(P, p0 + pch);
  const v2 base = ld2(twn + t);
  const unsigned lane_off = (unsigned)t * 16u;
  const float inv_n = 1.0f / (float)kBig;
  const int lag0 = (t >> 8) + 4 * (t & 255);        // lags of this lane: lag0 + 1024 k
  for (int p = p0; p < p1; p++) {
    const __amdgpu_buffer_rsrc_t cres = big_rsrc(C + (long)items[p] * kBig);
    const float2* xs = X + (((e * F + fset[p]) * D + d) * (long)B) * kBig;
    float q[kR];
    // the code spectrum of the item stays in registers for all B blocks: half the loads (and half the L2 traffic: the 63 x 128 KB
    // of B1I code spectra do not fit the 4 MB L2 next to the forward spectra) of re-reading it per block
    v2 c[kR];
#pragma unroll
    for (int jp = 0; jp < kR / 2; jp++) ld_pair_big(cres, lane_off, jp, c[2 * jp], c[2 * jp + 1]);
#pragma unroll
    for (int k = 0; k < kR; k++) q[k] = 0.f;
    for (int b = 0; b < B; b++) {
      const __amdgpu_buffer_rsrc_t xres = big_rsrc(xs + (long)b * kBig);
      v2 v[kR];
#pragma unroll
      for (int jp = 0; jp < kR / 2; jp++) {
        if (GACQ_ABL & 16) { v[2 * jp] = c[2 * jp + 1]; v[2 * jp + 1] = c[2 * jp]; continue; }      // ablation: no X loads
        ld_pair_big(xres, lane_off, jp, v[2 * jp], v[2 * jp + 1]);      // loads first, asm afterwards
      }
#pragma unroll
      for (int jj = 0; jj < kR; jj++) v[jj] = cmul(c[jj], v[jj]);
      if (!(GACQ_ABL & 32) && (b > 0 || p > p0)) __syncthreads();          // previous transform's last LDS reads are complete
      fft16k<true>(v, lds, twn, base);
#pragma unroll
      for (int k = 0; k < kR; k++) {
        const v2 r = v[rev16(k)];
        q[k] += __builtin_amdgcn_sqrtf(norm2(r)) * inv_n;
      }
    }
    float sum_f = q[0];
#pragma unroll
    for (int k = 1; k < kR; k++) sum_f += q[k];
    // lane l of a wave holds lags lag0 + 1024 k with lag0 = g + 4 (tl_base + l): first maximum as in lds_correlate_kernel
    float peak;
    unsigned widx;
    wave_first_max(q, (unsigned)__builtin_amdgcn_readfirstlane(lag0), 4u, 1024u, peak, widx);
    int idx = (int)widx;
    double sum = (double)wave_add_f32(sum_f);
    if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = idx; s_sum[t >> 6] = sum; }
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < kBigThreads / 64; w++) {
        if (s_peak[w] > peak || (s_peak[w] == peak && s_idx[w] < idx)) { peak = s_peak[w]; idx = s_idx[w]; }
        sum += s_sum[w];
      }
      RowRec r;
      r.peak = peak;
      r.idx = idx;
      r.sum = sum;
      rows[(e * P + p) * (long)D + d] = r;
    }
  }
}

// Fused search for item lists in which every item has its own carrier (F == P: the GLONASS FDMA channels, or a single
// item): the forward spectrum of (e, f, d, b) is used by exactly one item, so writing it to HBM and reading it back
// (8 N bytes each way per row) buys nothing.  Workgroup = (epoch, Doppler bin, item); per block b: mix + forward FFT,
// one LDS pass back to natural lane order, conj * C_p, inverse FFT, |.| accumulated in registers.  Same arithmetic in the
// same order as lds16k_forward_kernel + lds16k_correlate_kernel.
template <bool DUMP>
__global__ __launch_bounds__(kBigThreads) void lds16k_fused_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                                    const float2* __restrict__ C, const int* __restrict__ items,
                                                                    const int* __restrict__ fset, const double* __restrict__ freq,
                                                                    const float2* __restrict__ nco_tab,
                                                                    const float2* __restrict__ twn, RowRec* __restrict__ rows, int n,
                                                                    int P, int D, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2* lds = reinterpret_cast<v2*>(smem);
  __shared__ float s_peak[kBigThreads / 64];
  __shared__ int s_idx[kBigThreads / 64];
  __shared__ double s_sum[kBigThreads / 64];
  const int t = threadIdx.x;
  unsigned blk = blockIdx.x;                          // ((e*D + d)*P + p): the P items of one (e, d) run side by side
  const int p = (int)(blk % (unsigned)P);
  blk /= (unsigned)P;
  const int d = (int)(blk % (unsigned)D);
  const long e = blk / (unsigned)D;
  const double f = freq[(long)fset[p] * D + d];
  const __amdgpu_buffer_rsrc_t cres = big_rsrc(C + (long)items[p] * kBig);
  const v2 base = ld2(twn + t);
  const unsigned lane_off = (unsigned)t * 16u;
  const float inv_n = 1.0f / (float)kBig;
  const int lane = (t >> 8) + 4 * (t & 255);          // natural index (mod 1024) this lane holds after a transform
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
#pragma unroll
    for (int j = 0; j < kR; j++) v[j] = cmul(v[j], w[j]);
    fft16k<false>(v, lds, twn, base);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kR; k++) lds[lane + 1024 * k] = v[rev16(k)];
    __syncthreads();
    v2 c[kR];
#pragma unroll
    for (int jp = 0; jp < kR / 2; jp++) ld_pair_big(cres, lane_off, jp, c[2 * jp], c[2 * jp + 1]);
#pragma unroll
    for (int j = 0; j < kR; j++) { const v2 a = lds[t + 1024 * j]; v[j] = v2{a.x, -a.y}; }      // np.conj(fft.fft(b))
    __syncthreads();                                  // natural-order reads done before the inverse reuses the buffer
#pragma unroll
    for (int j = 0; j < kR; j++) v[j] = cmul(c[j], v[j]);
    fft16k<true>(v, lds, twn, base);
#pragma unroll
    for (int k = 0; k < kR; k++) {
      const v2 r = v[rev16(k)];
      q[k] += __builtin_amdgcn_sqrtf(norm2(r)) * inv_n;
    }
    __syncthreads();                                  // the inverse transform's last LDS reads are complete
  }
  float sum_f = q[0];
#pragma unroll
  for (int k = 1; k < kR; k++) sum_f += q[k];
  float peak;
  unsigned widx;
  wave_first_max(q, (unsigned)__builtin_amdgcn_readfirstlane(lane), 4u, 1024u, peak, widx);      // lane l of the wave: lags lane + 4 l + 1024 k
  int idx = (int)widx;
  double sum = (double)wave_add_f32(sum_f);
  if ((t & 63) == 0) { s_peak[t >> 6] = peak; s_idx[t >> 6] = idx; s_sum[t >> 6] = sum; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < kBigThreads / 64; w++) {
      if (s_peak[w] > peak || (s_peak[w] == peak && s_idx[w] < idx)) { peak = s_peak[w]; idx = s_idx[w]; }
      sum += s_sum[w];
    }
    RowRec r;
    r.peak = peak;
    r.idx = idx;
    r.sum = sum;
    rows[(e * P + p) * (long)D + d] = r;
  }
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
    if (GACQ_ABL & 64) {                        // A/B builds: natural order, 8-byte stores
#pragma unroll
      for (int k = 0; k < kR; k++) { const v2 o = k1 == 0 ? v[rev16(k)] : cmul(v[rev16(k)], tk[k]); dst[t + 256 * k] = make_float2(o.x, o.y); }
    } else if (k1 == 0) {
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

// natural -> lane-pair layout for the code spectra (once per signal)
__global__ __launch_bounds__(kBlock) void lds_permute_kernel(const float2* __restrict__ nat, float2* __restrict__ perm) {
  const long row = blockIdx.x;
  const int t = threadIdx.x;
  const float2* src = nat + row * kLdsN;
  float2* dst = perm + row * kLdsN;
#pragma unroll
  for (int jp = 0; jp < kR / 2; jp++) {
    const float2 a = src[t + 256 * (2 * jp)], b = src[t + 256 * (2 * jp + 1)];
    *reinterpret_cast<float4*>(dst + jp * 512 + 2 * t) = make_float4(a.x, a.y, b.x, b.y);
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
template <int MINW, bool B1, bool CACHEX, bool OPAQUE, bool PRETW = false>
__global__ __launch_bounds__(kBlock, MINW) void lds_correlate_kernel(const float2* __restrict__ X, const float2* __restrict__ C,
                                                                      const int* __restrict__ items, const int* __restrict__ fset,
                                                                      const float2* __restrict__ tw, RowRec* __restrict__ rows,
                                                                      int E, int P, int F, int D, int B, int pch, int nchunk) {
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
      else fft4096<true>(v, lds, wa, wb);
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
    float sum_f = q[0];
#pragma unroll
    for (int k = 1; k < kR; k++) sum_f += q[k];
    float wmaxf;
    unsigned widx;
    wave_first_max(q, (unsigned)__builtin_amdgcn_readfirstlane(t & ~63), 1u, 256u, wmaxf, widx);      // first maximum, like np.argmax
    const unsigned wmax = __builtin_bit_cast(unsigned, B1 ? wmaxf * inv_n : wmaxf);
    const float wsum = wave_add_f32(B1 ? sum_f * inv_n : sum_f);
    if ((t & 63) == 0) { s_peak[t >> 6] = __builtin_bit_cast(float, wmax); s_idx[t >> 6] = (int)widx; s_sum[t >> 6] = (double)wsum; }
    __syncthreads();   // also orders this item's exchange-2 reads before the next item's exchange-1 writes
    if (t == 0) {
      float bp = s_peak[0];
      int bi = s_idx[0];
      double bs = s_sum[0];
      for (int w = 1; w < kBlock / 64; w++) {
        if (s_peak[w] > bp || (s_peak[w] == bp && s_idx[w] < bi)) { bp = s_peak[w]; bi = s_idx[w]; }
        bs += s_sum[w];
      }
      RowRec r;
      r.peak = bp;
      r.idx = bi;
      r.sum = bs;
      rows[(e * P + p) * (long)D + d] = r;
    }
  }
}

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
template <int MINW, bool PREA>
__global__ __launch_bounds__(kBlock, MINW) void lds_fused4k_kernel(const float2* __restrict__ x, size_t epoch_stride,
                                                                    const float2* __restrict__ C, const int* __restrict__ items,
                                                                    const double* __restrict__ freq, const float2* __restrict__ nco_tab,
                                                                    const float2* __restrict__ tw, RowRec* __restrict__ rows, int E, int P,
                                                                    int D, int pch, int nchunk) {
  __shared__ v2 lds[kLdsElems];
  __shared__ float s_peak[kBlock / 64];
  __shared__ int s_idx[kBlock / 64];
  __shared__ double s_sum[kBlock / 64];
  const int t = threadIdx.x;
  const int xcd = blockIdx.x & 7;                     // (epoch, Doppler) unit -> XCD, as in lds_correlate_kernel
  const unsigned j = blockIdx.x >> 3;
  const unsigned u = (j / (unsigned)nchunk) * 8 + xcd;
  if (u >= (unsigned)E * (unsigned)D) return;
  const long e = u / (unsigned)D;
  const int d = (int)(u % (unsigned)D);
  const int p0 = (int)(j % (unsigned)nchunk) * pch;
  const int p1 = min(P, p0 + pch);
  v2 wa = ld2(tw + t), wb = ld2(tw + 16 * (t & 15));
  v2 xr[kR];
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
  const unsigned lane_off = (unsigned)t * 16u;
  v2 pwa[PREA ? 15 : 1];
  if (PREA) make_powers(reinterpret_cast<v2(&)[15]>(pwa), v2{wa.x, -wa.y});      // conjugate: inverse transform
  for (int p = p0; p < p1; p++) {
    const __amdgpu_buffer_rsrc_t cres = row_rsrc(C + (long)items[p] * kLdsN);
    // keep the (remaining) twiddle powers out of the loop-invariant set
    if (PREA) asm volatile("" : "+v"(wb.x), "+v"(wb.y));
    else asm volatile("" : "+v"(wa.x), "+v"(wa.y), "+v"(wb.x), "+v"(wb.y));
    v2 v[kR];
#pragma unroll
    for (int jp = 0; jp < kR / 2; jp++) ld_pair(cres, lane_off, jp, v[2 * jp], v[2 * jp + 1]);
#pragma unroll
    for (int jj = 0; jj < kR; jj++) v[jj] = cmul(v[jj], xr[jj]);
    if (PREA) fft4096<true, 1>(v, lds, wa, wb, reinterpret_cast<const v2(*)[15]>(pwa), nullptr);
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
    wave_first_max(m, (unsigned)__builtin_amdgcn_readfirstlane(t & ~63), 1u, 256u, wmaxf, widx);
    const unsigned wmax = __builtin_bit_cast(unsigned, wmaxf * inv_n);
    const float wsum = wave_add_f32(sum_f * inv_n);
    if ((t & 63) == 0) { s_peak[t >> 6] = __builtin_bit_cast(float, wmax); s_idx[t >> 6] = (int)widx; s_sum[t >> 6] = (double)wsum; }
    __syncthreads();   // also orders this item's exchange-2 reads before the next item's exchange-1 writes
    if (t == 0) {
      float bp = s_peak[0];
      int bi = s_idx[0];
      double bs = s_sum[0];
      for (int w = 1; w < kBlock / 64; w++) {
        if (s_peak[w] > bp || (s_peak[w] == bp && s_idx[w] < bi)) { bp = s_peak[w]; bi = s_idx[w]; }
        bs += s_sum[w];
      }
      RowRec r;
      r.peak = bp;
      r.idx = bi;
      r.sum = bs;
      rows[(e * P + p) * (long)D + d] = r;
    }
  }
}

typedef void (*CorrKernel)(const float2*, const float2*, const int*, const int*, const float2*, RowRec*, int, int, int, int, int, int, int);

struct CorrVariant { const char* name; CorrKernel b1; CorrKernel bn; };

// variant table; index chosen by GACQ_LDS_VARIANT (tuning aid), default = kDefaultVariant
const CorrVariant kVariants[] = {
  {"w1-hoist", lds_correlate_kernel<1, true, false, false>, lds_correlate_kernel<1, false, false, false>},
  {"w2-hoist-cachex", lds_correlate_kernel<2, true, true, false>, lds_correlate_kernel<2, false, false, false>},
  {"w4-opaque-cachex", lds_correlate_kernel<4, true, true, true>, lds_correlate_kernel<4, false, false, true>},
  {"w4-opaque", lds_correlate_kernel<4, true, false, true>, lds_correlate_kernel<4, false, false, true>},
  {"w3-opaque-cachex", lds_correlate_kernel<3, true, true, true>, lds_correlate_kernel<3, false, false, true>},
  {"w2-opaque-cachex", lds_correlate_kernel<2, true, true, true>, lds_correlate_kernel<2, false, false, true>},
  {"w3-pretw-cachex", lds_correlate_kernel<3, true, true, false, true>, lds_correlate_kernel<3, false, false, false, true>},
  {"w2-pretw-cachex", lds_correlate_kernel<2, true, true, false, true>, lds_correlate_kernel<2, false, false, false, true>},
  {"w3-pretw", lds_correlate_kernel<3, true, false, false, true>, lds_correlate_kernel<3, false, false, false, true>},
  {"w4-pretw", lds_correlate_kernel<4, true, false, false, true>, lds_correlate_kernel<4, false, false, false, true>},
};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
constexpr int kDefaultVariant = 1;   // profiles/r01_ab_variants_*.log: all packed-math variants are within 5 %; this one led twice

int twiddle_table(gacq_ctx* ctx, const float2** out) { return twiddle_cache(ctx, "W4096", kLdsN, kLdsN, out); }

}  // namespace

namespace gacq {

bool lds_supported(int N) { return N == kLdsN || N == kBig; }

int lds_prepare_spectra(gacq_ctx* ctx, const float2* natural, float2* perm, int nprn, int N) {
  if (!lds_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "LDS FFT engine: N=%d not supported", N);
  if (N == kBig) {
    hipLaunchKernelGGL(lds16k_permute_kernel, dim3((unsigned)nprn), dim3(kBigThreads), 0, ctx->stream, natural, perm);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  hipLaunchKernelGGL(lds_permute_kernel, dim3((unsigned)nprn), dim3(kBlock), 0, ctx->stream, natural, perm);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int lds_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, int N, const double* d_freq, int FD,
                int B, const float2* tab, float2* X) {
  if (!lds_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "LDS FFT engine: N=%d not supported", N);
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
                     const int* d_fset, const double* d_freq, const float2* tab, int nitems, int D, int B, RowRec* rows) {
  if (N != kBig) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "fused LDS search: N=%d not supported", N);
  const float2* twn;
  int rc = twiddle_cache(ctx, "W16384_lo", kBig, 1024, &twn);
  if (rc != GACQ_OK) return rc;
  GACQ_HIP(ctx, hipFuncSetAttribute((const void*)lds16k_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
  hipLaunchKernelGGL(lds16k_fused_kernel<false>, dim3((unsigned)((long)nepoch * D * nitems)), dim3(kBigThreads), kBigLdsBytes, ctx->stream, x,
                     nsamp, spectra, d_items, d_fset, d_freq, tab, twn, rows, n, nitems, D, B);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

// Worth it for batches only: with few (epoch, Doppler) units every workgroup's own forward transform sits on the critical path
// (single epoch: 32 us fused against 19 us for the two kernels), with many it replaces a launch, 84 MB of X traffic and a tail.
bool lds_fused4k_supported(const gacq_ctx* ctx, int N, int B, int F, long units) {
  return N == kLdsN && B == 1 && F == 1 && (ctx->opt[GACQ_OPT_FUSED_4K] >= 2 || (ctx->opt[GACQ_OPT_FUSED_4K] == 1 && units >= 1024));
}

int lds_fused4k_search(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, const float2* spectra, const int* d_items,
                       const double* d_freq, const float2* tab, int nitems, int D, RowRec* rows) {
  const float2* tw;
  int rc = twiddle_table(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  // one forward transform per workgroup: amortise it over up to 32 items while >= ~2048 workgroups remain
  const long units = (long)nepoch * D;
  int pch = 8;
  while (pch < 32 && units * ((nitems + 2 * pch - 1) / (2 * pch)) >= 2048) pch *= 2;
  if (ctx->opt[GACQ_OPT_LDS_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_LDS_PCH];
  pch = std::min(pch, nitems);
  const int nchunk = (nitems + pch - 1) / pch;
  const long units8 = (units + 7) / 8;
  if (ctx->opt[GACQ_OPT_LDS_VARIANT] == 100)
    hipLaunchKernelGGL((lds_fused4k_kernel<4, false>), dim3((unsigned)(8 * units8 * nchunk)), dim3(kBlock), 0, ctx->stream, x, nsamp, spectra, d_items,
                       d_freq, tab, tw, rows, nepoch, nitems, D, pch, nchunk);
  else
    hipLaunchKernelGGL((lds_fused4k_kernel<4, true>), dim3((unsigned)(8 * units8 * nchunk)), dim3(kBlock), 0, ctx->stream, x, nsamp, spectra, d_items,
                       d_freq, tab, tw, rows, nepoch, nitems, D, pch, nchunk);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

int lds_debug_nco(gacq_ctx* ctx, int N, int n, const double* d_freq, bool fused, int* d_idx) {
  if (!lds_supported(N) || (fused && N != kBig)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "NCO index dump: no %sLDS forward kernel for N=%d", fused ? "fused " : "", N);
  if (N == kBig && fused) {
    int rc = ensure(ctx, ctx->fset, sizeof(int));
    if (rc != GACQ_OK) return rc;
    ctx->up_fset.clear();
    GACQ_HIP(ctx, hipMemsetAsync(ctx->fset.p, 0, sizeof(int), ctx->stream));
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)lds16k_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
    hipLaunchKernelGGL(lds16k_fused_kernel<true>, dim3(1), dim3(kBigThreads), kBigLdsBytes, ctx->stream, (const float2*)nullptr, (size_t)0,
                       (const float2*)nullptr, (const int*)ctx->fset.p, (const int*)ctx->fset.p, d_freq, (const float2*)nullptr,
                       (const float2*)nullptr, (RowRec*)d_idx, n, 1, 1, 1);
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
                  int nitems, int F, int D, int B, int N, RowRec* rows) {
  if (!lds_supported(N)) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "LDS FFT engine: N=%d not supported", N);
  if (N == kBig) {
    const float2* twn;
    int rcb = twiddle_cache(ctx, "W16384_lo", kBig, 1024, &twn);
    if (rcb != GACQ_OK) return rcb;
    // one 1024-thread workgroup per CU: keep >= ~1024 workgroups, at most 8 items per group
    const long rows_total = (long)nepoch * nitems * D;
    int pch = (int)std::max<long>(1, std::min<long>(8, rows_total / 1024));
    pch = std::min(pch, nitems);
    const int nchunk = (nitems + pch - 1) / pch;
    const long units8 = ((long)nepoch * D + 7) / 8;
    GACQ_HIP(ctx, hipFuncSetAttribute((const void*)lds16k_correlate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLdsBytes));
    hipLaunchKernelGGL(lds16k_correlate_kernel, dim3((unsigned)(8 * units8 * nchunk)), dim3(kBigThreads), kBigLdsBytes, ctx->stream, X,
                       spectra, d_items, d_fset, twn, rows, nepoch, nitems, F, D, B, pch, nchunk);
    GACQ_HIP(ctx, hipGetLastError());
    return GACQ_OK;
  }
  const float2* tw;
  int rc = twiddle_table(ctx, &tw);
  if (rc != GACQ_OK) return rc;
  // items per workgroup: keep >= ~2048 workgroups in flight (256 CUs x 4 resident x 2), at most 8 per group
  const long rows_total = (long)nepoch * nitems * D;
  int pch = (int)std::max<long>(1, std::min<long>(8, rows_total / 2048));
  if (ctx->opt[GACQ_OPT_LDS_PCH] >= 1) pch = (int)ctx->opt[GACQ_OPT_LDS_PCH];
  pch = std::min(pch, nitems);
  const int nchunk = (nitems + pch - 1) / pch;
  const long units8 = ((long)nepoch * D + 7) / 8;
  const long grid = 8 * units8 * nchunk;
  int variant = kDefaultVariant;
  if (ctx->opt[GACQ_OPT_LDS_VARIANT] >= 0 && ctx->opt[GACQ_OPT_LDS_VARIANT] < kNumVariants) variant = (int)ctx->opt[GACQ_OPT_LDS_VARIANT];
  // register-cached X needs all items of a workgroup to share one forward set
  const bool b1 = (B == 1) && (F == 1);
  CorrKernel kern = b1 ? kVariants[variant].b1 : kVariants[variant].bn;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, X, spectra, d_items, d_fset, tw, rows, nepoch,
                     nitems, F, D, B, pch, nchunk);
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
