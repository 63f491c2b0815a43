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
// R = 31, any 13-smooth M >= 64 (N = 61380 / 30690: the 10.23 Mcps family, acquire-gps-l5i.py:19-24 and 18 more scripts, and
//   E6, acquire-galileo-e6b.py:19-24).  rocFFT has no radix-31 butterfly and falls back to Bluestein for these lengths
//   (three transforms of twice the size per FFT); M = 4*5*9*11 is native to it.  The DFT-31 uses the conjugate symmetry
//   of W_31 (gacq_cplx.h: dft_prime), a quarter of the 31 x 31 complex products.  For the two lengths the reference uses, auto
//   runs the twiddle-free prime-factor form of gacq_pfa.hip instead (since round 5); this Cooley-Tukey form with rocFFT inner
//   transforms stays as the independent cross-check (GACQ_OPT_FUSED_INNER 0) and for other multiples of 31.
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
    const float2* __restrict__ Z, RowRec* __restrict__ partial, const float2* __restrict__ tw, int M, int B, int chunks, float inv_n,
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
      const float2* src = Z + (g * B) * (long)(R * M) + pos;
#pragma unroll
      for (int k1 = 0; k1 < R; k1++) v[k1] = ld_stream<kNT>(src + (long)k1 * M);
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
        const float2* src = Z + (g * B + b) * (long)(R * M) + pos;
#pragma unroll
        for (int k1 = 0; k1 < R; k1++) v[k1] = ld_stream<kNT>(src + (long)k1 * M);
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
int launch_inverse(gacq_ctx* ctx, const float2* Z, RowRec* partial, const float2* tw, int M, int B, long ng, float inv_n,
                   float* q_out, bool twiddle, int paired, float tie_scale) {
  const int chunks = (M + kBlock - 1) / kBlock;
  const dim3 grid((unsigned)(ng * chunks));
  const bool b1 = (B == 1) && !q_out;
  if (twiddle && b1)
    hipLaunchKernelGGL((split_outer_inverse_kernel<R, true, true>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, tw, M, B, chunks, inv_n, q_out, paired, tie_scale);
  else if (twiddle)
    hipLaunchKernelGGL((split_outer_inverse_kernel<R, true, false>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, tw, M, B, chunks, inv_n, q_out, paired, tie_scale);
  else if (b1)
    hipLaunchKernelGGL((split_outer_inverse_kernel<R, false, true>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, tw, M, B, chunks, inv_n, q_out, paired, tie_scale);
  else
    hipLaunchKernelGGL((split_outer_inverse_kernel<R, false, false>), grid, dim3(kBlock), 0, ctx->stream, Z, partial, tw, M, B, chunks, inv_n, q_out, paired, tie_scale);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace

namespace gacq {

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

int split_inverse_reduce(gacq_ctx* ctx, float2* Y, RowRec* rows, long g0, long ng, int B, int N, float* q_out, float tie_scale, bool inner) {
  const int R = split_radix(N);
  if (!R) return set_error(ctx, GACQ_ERR_UNSUPPORTED, "split engine: N=%d not supported", N);
  const int M = N / R;
  // Y written by lds_inner_correlate_kernel (inner == false, M == 4096): rows in the lane-pair layout
  const int paired = (!inner && M == 4096) ? 1 : 0;
  const float2* tw;
  int rc = base_twiddles(ctx, N, M, &tw);
  if (rc != GACQ_OK) return rc;
  const int chunks = (M + kBlock - 1) / kBlock;
  if (inner && (rc = fft_exec(ctx, M, ng * B * R, true, Y)) != GACQ_OK) return rc;
  if ((rc = ensure(ctx, ctx->partial, sizeof(RowRec) * (size_t)ng * chunks)) != GACQ_OK) return rc;
  RowRec* partial = (RowRec*)ctx->partial.p;
  const float inv_n = 1.0f / (float)N;
  switch (R) {
    case 31: rc = launch_inverse<31>(ctx, Y, partial, tw, M, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
    case 40: rc = launch_inverse<40>(ctx, Y, partial, tw, M, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
    case 20: rc = launch_inverse<20>(ctx, Y, partial, tw, M, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
    case 16: rc = launch_inverse<16>(ctx, Y, partial, tw, M, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
    default: rc = launch_inverse<4>(ctx, Y, partial, tw, M, B, ng, inv_n, q_out, inner, paired, tie_scale); break;
  }
  if (rc != GACQ_OK) return rc;
  return split_combine(ctx, partial, rows, g0, ng, chunks, tie_scale);
}

int split_combine(gacq_ctx* ctx, const RowRec* partial, RowRec* rows, long g0, long ng, int chunks, float tie_scale) {
  hipLaunchKernelGGL(split_combine_kernel, dim3((unsigned)((ng + 127) / 128)), dim3(128), 0, ctx->stream, partial, rows, g0, ng, chunks, tie_scale);
  GACQ_HIP(ctx, hipGetLastError());
  return GACQ_OK;
}

}  // namespace gacq
