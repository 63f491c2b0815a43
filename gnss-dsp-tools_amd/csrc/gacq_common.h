// Internal declarations shared by the engine translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gacq.h"

namespace gacq {

constexpr int kNcoTableSize = 1024;        // gnsstools/nco.py:3
constexpr int kBlock = 256;                // 4 wave64 per workgroup

// Per-(epoch, item, doppler) reduction record written by the magnitude/peak stage.
struct RowRec {
  float peak;    // max_k q[k]
  int idx;       // first k attaining it (np.argmax semantics)
  double sum;    // sum_k q[k]  (for the max/mean metric)
};

// Tie-safe peak locations (gacq_tiesafe.hip).  The reference takes argmax and the strict-'>' Doppler scan in fp64
// (acquire-gps-l1.py:34-39); the fp32 engines can land on another lag / bin when two candidates are closer than their rounding
// error.  Every row reduction therefore also reports whether a second lag comes within `tie_scale` = 1 - eps of the row maximum
// (bit kTieBit of RowRec::idx -- lags are < 2^28), the Doppler scan does the same across bins, and the (epoch, item) pairs that
// are ambiguous get their candidate rows re-evaluated in complex128 on the device before the answer is written.
constexpr int kTieBit = 0x40000000;
constexpr int kIdxMask = kTieBit - 1;
struct TieRow { int ep, d; };                       // row to re-evaluate: ep = epoch * P + item position, d = index into the Doppler grid
struct TieEp { int ep, slot0, cnt, pad; };          // an ambiguous (epoch, item): its candidate rows are slots [slot0, slot0 + cnt), ascending d
struct TieRec { double peak, sum; int idx, pad; };  // complex128 result of a re-evaluated row
struct TieCounters {
  unsigned nrows, neps, pad0, pad1;                 // fill levels of the two lists of the launch in flight (reset by tie_resolve_kernel)
  unsigned long long flagged, rows, overflow, moved;      // cumulative since gacq_create: ambiguous pairs, rows re-evaluated, pairs dropped for lack
                                                          // of list space (they keep their fp32 answer), pairs whose location changed
};
struct TieLists {                                   // kernel argument (by value)
  TieCounters* c;
  unsigned* host_full;                              // pinned host word: set (before the pair's record is written) by a listing kernel that finds no room
  unsigned* done;                                   // per listed row: blocks finished so far (the workgroups of a row hand over through it)
  TieEp* eps;
  TieRow* rows;
  TieRec* recs;
  int cap;
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct FftPlan {
  rocfft_plan plan = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
  size_t work_size = 0;
};

struct StageEvent { int stage; hipEvent_t a, b; };

struct BatchRing;                      // pinned staging ring of the host-buffer batched entry points (gacq_host.hip)

}  // namespace gacq

struct gacq_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  int engine = 0;
  size_t ws_limit = (size_t)32 << 30;      // of 288 GB: a B = 80 search of engine 3 is 22 % faster in 64 GiB passes than in 4 GiB ones (r05_workspace_limit_sweep.log)
  bool ws_explicit = false;                // the caller set the limit (gacq_set_workspace_limit): it is not second-guessed from the free memory
  size_t ws_soft = 0;                      // what the device's free memory allows this context per buffer (0 = not looked yet); halved after a failed allocation
  int ws_soft_nctx = 0;                    // contexts on the device when ws_soft was computed
  bool alloc_failed = false;               // ensure() could not get its bytes: the search is retried in smaller passes
  std::map<std::pair<long, long>, gacq::FftPlan> plans;   // (N * 2 + inverse, batch)
  gacq::DevBuf tab, xstage, x32, X, Y, rows, freq, fset, items, out_peaks, d0, partial, fe_a, fe_b, fe_taps, chunk_peaks;
  gacq::DevBuf tie, tie_scratch, tie_q, tie_split, tie_done2;   // tie-safe re-evaluation: counters + lists, complex128 row scratch, per-block magnitude rows
  int tie_cap = 0;                     // list capacity the `tie` buffer was laid out for
  long opt[GACQ_NOPTS] = {1, 1, -1, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 1, 8000, 0, 1, 0};   // gacq_set_option values (defaults documented in include/gacq.h)
  gacq::DevBuf pin_x, pin_peaks;       // pinned host staging for the host-buffer entry point (gacq_search)
  gacq::DevBuf acq_in, acq_x;          // gacq_acquire_int8: the raw int8 block and its front-end output on the device
  gacq::DevBuf pin_tie;                // pinned host word the listing kernels set when the re-evaluation list is full (GACQ_WARN_TIE_LIST_FULL)
  gacq::DevBuf bar_x;                  // fine-grained device memory the host writes directly through the PCIe BAR (small gacq_search inputs)
  gacq::DevBuf bar_s;                  // the same for the correlator specs of gacq_correlate_batch_dev
  bool large_bar = false;              // hipDeviceProp_t.isLargeBar: device memory is host-addressable
  gacq::BatchRing* ring = nullptr;     // staging ring of gacq_search_batch / gacq_group_search_batch, created on first use
  bool profiling = false;
  double stage_ms[GACQ_NSTAGES] = {0};
  long stage_n[GACQ_NSTAGES] = {0};
  std::vector<gacq::StageEvent> pending;
  // last grid uploaded to freq/fset/items (skip the H2D + sync when a call repeats it, as batched loops do)
  std::vector<double> up_freq;
  std::vector<int> up_fset, up_items, up_d0;
  // ... and the grids before it, with their device buffers: callers that alternate between a few grids -- several signals per step
  // (BASELINE configs 4 and 5), a rank's Doppler slice and the full grid of the tie-safe merge -- find theirs again instead of paying
  // an upload and a stream synchronisation per call (round 4; most recently displaced first, at most kGridSlots)
  struct GridSlot {
    std::vector<double> freq;
    std::vector<int> fset, items;
    gacq::DevBuf dfreq, dfset, ditems;
  };
  static constexpr int kGridSlots = 7;
  std::vector<GridSlot> grid_slots;
  std::vector<float> up_taps;
  // correlator calls repeat their (code, PRN list) every millisecond: the device chip-table pointers of the last call are kept
  std::string tr_code;
  std::vector<int> tr_prns;
  std::vector<const void*> tr_chips;
  // device-resident constant tables (twiddles, long-code chips), keyed by name; owned by the ctx, freed in gacq_destroy
  std::map<std::string, gacq::DevBuf> tables;
};

struct gacq_sig {
  gacq_ctx* ctx = nullptr;
  gacq_sigdesc desc{};
  int nprn = 0;
  int N = 0;                 // FFT length: n or 2n
  float2* spectra = nullptr; // [nprn][N] code spectra C_p = fft(replica), complex64, natural order
  float2* spectra_r31 = nullptr;   // same in the radix-31 engine's [k1][k2] order (only when split_supported(N))
  float2* spectra_pfa = nullptr;   // same in the prime-factor engine's order (only when pfa_supported(N))
  float2* spectra_split = nullptr; // split engine with LDS inner transforms: R lane-pair rows per item (N = R*4096)
  float2* spectra_lds = nullptr;   // same in the LDS engine's lane-pair layout (only when lds_supported(N))
  float2* spectra_lds16 = nullptr; // N = 16384 only: the order of the radix-16 form of the transform (GACQ_OPT_LDS_VARIANT = 16; spectra_lds holds the radix-32 form's, gacq_lds16k.hip)
  double2* spectra64 = nullptr;    // complex128 code spectra of the verification engine (engine 5), built on first use
  double2* spectra64_split = nullptr;   // the same as R rows per item, [p][k1][k2] = C[k1 + R k2] (N = R x 4096: the split form of engine 5), built on first use
  std::vector<float> replica;      // host copy of the +-1 replicas [nprn][n] (source of spectra64)
};

namespace gacq {

// The samples of a search as the caller handed them over: interleaved complex64, or (wide) interleaved complex128 -- what the
// reference's search() actually receives from np.interp (acquire-gps-l1.py:94-96, used at :30-33).  The fp32 engines run on a
// complex64 rounding of wide input; the complex128 engine and the tie-safe re-evaluation read it as given and widen nothing.
struct XSrc {
  const void* p;
  int wide;
  __host__ __device__ XSrc offset(size_t samples) const {
    return XSrc{wide ? (const void*)((const double2*)p + samples) : (const void*)((const float2*)p + samples), wide};
  }
};
__device__ __forceinline__ double2 ld_x(XSrc x, size_t i) {
  if (x.wide) return reinterpret_cast<const double2*>(x.p)[i];
  const float2 s = reinterpret_cast<const float2*>(x.p)[i];
  return make_double2((double)s.x, (double)s.y);
}

int set_error(gacq_ctx* ctx, int code, const char* fmt, ...);
void ring_destroy(gacq_ctx* ctx);
int ensure(gacq_ctx* ctx, DevBuf& b, size_t bytes);
// bytes one forward-spectra / correlation-workspace buffer of this context may take: the workspace limit, and -- unless the caller set
// that limit itself -- no more than its share of the device's free memory (two such buffers per context, every context on the device)
size_t ws_budget(gacq_ctx* ctx);
// latency path: host write through the PCIe BAR into fine-grained device memory / completion by watching pinned result records
bool bar_write(gacq_ctx* ctx, DevBuf& b, const void* src, size_t bytes);
bool watch_records(const volatile unsigned long long* words, size_t count, size_t stride, size_t at, unsigned long long sentinel, int timeout_us);
int ensure_pinned(gacq_ctx* ctx, DevBuf& b, size_t bytes);      // hipHostMalloc'd, device-accessible
// W_N^k = exp(-2 pi i k / N) for k < count, fp64-evaluated and rounded once to fp32; cached per ctx under `key`
int twiddle_cache(gacq_ctx* ctx, const std::string& key, int N, int count, const float2** out);
// arbitrary constant bytes cached per ctx under `key` (uploaded on first use)
int table_cache(gacq_ctx* ctx, const std::string& key, const void* host, size_t bytes, const void** out);
int fft_exec(gacq_ctx* ctx, int N, long batch, bool inverse, void* data, bool fp64 = false);
// engine 5: the pipeline in complex128 on the device (gacq_verify.hip); ctx->freq / items / fset already uploaded
int verify_search(gacq_sig* sig, XSrc d_x, size_t nsamp, int nepoch, int P, int F, int D, int B, gacq_peak* d_out, float* d_qrow);
void stage_begin(gacq_ctx* ctx, int stage);
void stage_end(gacq_ctx* ctx);

// LDS-resident FFT engine (gacq_ldsfft.hip): supported lengths and the two launches.
bool lds_supported(int N);
int lds_code_spectra(gacq_ctx* ctx, const float2* replica_rows, float2* perm, int nprn, int N, bool radix16 = false);      // code spectra in the LDS engines' layout: the replicas through their own forward transform
// X[row][k] = conj(FFT_N(x_window * nco))   rows = ((e*F + f)*D + d)*B + b
int lds_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, int N, const double* d_freq,
                int FD, int B, const float2* tab, float2* X);
// rows[(e*P + p)*D + d] = reduce_k sum_b | IFFT_N(C_p * X[e,f(p),d,b]) | / N
// N = 16384 with one carrier per item (F == P): forward + correlate in one kernel, no X buffer
bool lds_fused_supported(const gacq_ctx* ctx, int N, int P, int F);
int lds_fused_search(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, int N, const float2* spectra, const int* d_items,
                     const int* d_fset, const double* d_freq, const float2* tab, int nitems, int D, int B, RowRec* rows, float tie_scale);
// N = 4096, B == 1, one carrier: forward + correlate in one kernel, no X buffer
bool lds_fused4k_supported(const gacq_ctx* ctx, int N, int B, int F, long units);
int lds_fused4k_search(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, const float2* spectra, const int* d_items,
                       const double* d_freq, const float2* tab, int nitems, int D, RowRec* rows, float tie_scale);
int lds_correlate(gacq_ctx* ctx, const float2* X, const float2* spectra, const int* d_items, const int* d_fset,
                  int nepoch, int nitems, int F, int D, int B, int N, RowRec* rows, float tie_scale, float* q_out = nullptr);

// N = 16384 as 32 x 32 x 16 in one 512-thread workgroup (gacq_lds16k.hip): the launches behind the lds_* entry points above
int r32_code_spectra(gacq_ctx* ctx, const float2* replica_rows, float2* perm, int nprn);
int r32_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, const double* d_freq, int FD, int B, const float2* tab,
                float2* X);
int r32_fused_search(gacq_ctx* ctx, const float2* x, size_t nsamp, int nepoch, int n, const float2* spectra, const int* d_items,
                     const int* d_fset, const double* d_freq, const float2* tab, int nitems, int D, int B, RowRec* rows, float tie_scale);
int r32_correlate(gacq_ctx* ctx, const float2* X, const float2* spectra, const int* d_items, const int* d_fset, int nepoch, int nitems,
                  int F, int D, int B, RowRec* rows, float tie_scale, float* q_out);
int r32_debug_nco(gacq_ctx* ctx, int n, const double* d_freq, bool fused, int* d_idx);

// test hook (gacq_debug_nco_indices): the forward kernel's own NCO index expression for one row, d_idx[N]; fused: the
// one-kernel N = 16384 search
int lds_debug_nco(gacq_ctx* ctx, int N, int n, const double* d_freq, bool fused, int* d_idx);
int split_debug_nco(gacq_ctx* ctx, int N, int n, const double* d_freq, int* d_idx);

// front-end carrier wipe-off alone (gacq_frontend.hip): int8 I/Q on the device -> complex64, fixed-point table NCO
int frontend_mix(gacq_ctx* ctx, const void* d_iq_int8, long n, double fs_in, double carrier_offset_hz, float2* d_out);

// split engines (gacq_split.hip): N = R*M, hand-written outer DFT-R, inner length-M transforms (rocFFT or the 4096 kernels)
bool split_supported(int N);
// outer DFT-31 (+NCO mix when mix) + twiddle, then the inner forward transforms; X in [k1][k2] order
// inner == false: outer stage only (the caller runs the inner transforms itself)
int split_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, long rows, int n, int N, const double* d_freq, int FD, int B,
                const float2* tab, float2* X, bool mix, bool inner = true);
int split_radix(int N);
// inner stages of the split engine on the LDS FFT kernels (M = 4096)
int lds_inner_forward(gacq_ctx* ctx, float2* rows, long nrows, bool conj);
int lds_inner_correlate(gacq_ctx* ctx, const float2* X, const float2* spectra, const int* d_items, const int* d_fset, long g0,
                        long ng, int P, int F, int D, int B, int R, int N, float2* Z);
// inner inverse transforms on Y (in place), then twiddle + inverse DFT-31 + |.|/N + sum over B + reduce -> rows[g0..g0+ng)
// inner == false: Y already holds the twiddled inner inverse transforms (LDS inner path)
int split_inverse_reduce(gacq_ctx* ctx, float2* Y, RowRec* rows, long g0, long ng, int B, int N, float* q_out, float tie_scale, bool inner = true);

// partial[(g, chunk)] -> rows[g0 + g] (tie-tag aware): the last step of every split engine's outer inverse stage
int split_combine(gacq_ctx* ctx, const RowRec* partial, RowRec* rows, long g0, long ng, int chunks, float tie_scale);

// prime-factor form of the radix-31 engine (gacq_pfa.hip): N = 61380 / 30690 as the twiddle-free 4-D transform 31 x 11 x Nb x 9
bool pfa_supported(int N);
int pfa_row_pitch(int N);                        // pitch (complex elements) of a Z' row: 9 segments of whole 128-byte lines
// rotated gather of x (+ NCO mix) + DFT-31, then the in-place inner transforms; X rows in the engine's own spectrum order
int pfa_forward(gacq_ctx* ctx, const float2* x, size_t nsamp, long rows, int n, int N, const double* d_freq, int FD, int B, const float2* tab,
                float2* X, bool mix);
// K2 + inner inverse transforms -> Z' (rows pfa_row_pitch apart)
int pfa_inner_correlate(gacq_ctx* ctx, const float2* X, const float2* C, const int* d_items, const int* d_fset, long g0, long ng, int P, int F,
                        int D, int B, int N, float2* Z);
// inverse DFT-31 (packed math or matrix pipe, GACQ_OPT_SPLIT_MFMA) + |.|/N + sum over B + reduce -> rows[g0..g0+ng)
// need_sum: the row sum feeds the max/mean metric (acquire-gps-l1.py:35); raw-metric signals skip it (RowRec::sum = 0)
int pfa_inverse_reduce(gacq_ctx* ctx, const float2* Z, RowRec* rows, long g0, long ng, int B, int N, float* q_out, float tie_scale, bool need_sum);
int pfa_debug_nco(gacq_ctx* ctx, int N, int n, const double* d_freq, int* d_idx);

// tie-safe re-evaluation (gacq_tiesafe.hip)
int tie_code_spectra64(gacq_ctx* ctx, double2* rows, int nrows, int N);
int twiddles64(gacq_ctx* ctx, int N, const double2** out);      // W_N^k, k < N, in fp64 on the device, cached per context      // gacq_tiesafe.hip: complex128 forward transforms of nrows rows, no rocFFT
bool tie_supported(int N);                       // prime factors of N in {2, 3, 5, 7, 11, 13, 31}: every FFT length of the reference's scripts
float tie_scale_of(const gacq_ctx* ctx);            // 1 - eps from GACQ_OPT_TIE_EPS_PPB
int tie_prepare(gacq_sig* sig);                  // complex128 code spectra on first use
int tie_lists(gacq_ctx* ctx, long nep, int N, int B, TieLists* out, gacq_peak** guesses);      // capacity bounded by the memory budget of the re-evaluation
int tie_resolve(gacq_sig* sig, const TieLists& tl, const gacq_peak* guesses, XSrc d_x, size_t nsamp, int P, int D, int B, gacq_peak* d_out);

// Running (maximum, first argmax, runner-up) of magnitudes visited in ascending lag order, and the merge of two such partial results.
// An exact duplicate of the maximum counts as its runner-up.
struct Top2 {
  float peak = -1.0f;
  int idx = 0x7fffffff;
  float second = -1.0f;
  __device__ __forceinline__ void add(float v, int i) {
    if (v > peak) { second = peak; peak = v; idx = i; } else second = fmaxf(second, v);
  }
  __device__ __forceinline__ void merge(float op, int oi, float os) {
    second = fmaxf(fmaxf(second, os), fminf(peak, op));
    if (op > peak || (op == peak && oi < idx)) { peak = op; idx = oi; }
  }
  // idx with kTieBit set when the runner-up comes within tie_scale of the maximum
  __device__ __forceinline__ int tagged(float tie_scale) const { return idx | ((second >= peak * tie_scale) ? kTieBit : 0); }
};

// Combine nparts partial (maximum, idx | tie bit) results of one row -- waves, workgroups or column chunks.  Conservative: a part
// whose maximum reaches the threshold of the combined maximum counts once, twice if it is ambiguous in itself; two or more make
// the row ambiguous.  On equal maxima the lower lag wins (np.argmax returns the first maximum).
template <class GetPeak, class GetIdx>
__device__ __forceinline__ void combine_tagged(int nparts, GetPeak pk, GetIdx ix, float tie_scale, float& peak, int& tagged_idx) {
  float bp = pk(0);
  int bi = ix(0) & kIdxMask;
  for (int w = 1; w < nparts; w++) {
    const float p = pk(w);
    const int i = ix(w) & kIdxMask;
    if (p > bp || (p == bp && i < bi)) { bp = p; bi = i; }
  }
  const float thr = bp * tie_scale;
  int near = 0;
  for (int w = 0; w < nparts; w++)
    if (pk(w) >= thr) near += 1 + ((ix(w) & kTieBit) != 0);
  peak = bp;
  tagged_idx = bi | (near > 1 ? kTieBit : 0);
}

// Makes ctx's device current for the duration of an entry point and restores the caller's device afterwards
// (a library must not leave a different current device behind in a multi-GPU process).
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = (hipSetDevice(device) == hipSuccess); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define GACQ_DEVICE(ctx)                                                                                  \
  gacq::DeviceGuard device_guard_((ctx)->device);                                                         \
  if (!device_guard_.ok) return gacq::set_error((ctx), GACQ_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->device)

// Table-NCO index of sample i: floor((0 + f*i) * 1024) mod 1024 with the two products rounded to fp64 exactly like numpy's
// `idx = p + f*np.arange(n); np.floor(idx*NT)` (gnsstools/nco.py:7-9).  |f*i*1024| < 2^31 is checked on the host
// (nco_range_ok), which lets the floor go through one f64->i32 conversion instead of the multi-instruction f64->i64 one.
__device__ __forceinline__ int nco_index(double f, int i) {
  return ((int)floor(__dmul_rn(__dmul_rn(f, (double)i), 1024.0))) & (kNcoTableSize - 1);
}
inline bool nco_range_ok(double max_abs_f, long span) { return max_abs_f * (double)span * 1024.0 < 2.0e9; }

#define GACQ_HIP(ctx, call)                                                                         \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return gacq::set_error((ctx), GACQ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                                   \
  } while (0)

#define GACQ_FFT(ctx, call)                                                                         \
  do {                                                                                              \
    rocfft_status s_ = (call);                                                                      \
    if (s_ != rocfft_status_success)                                                                \
      return gacq::set_error((ctx), GACQ_ERR_ROCFFT, "%s failed: rocfft_status %d (%s:%d)", #call, (int)s_, \
                             __FILE__, __LINE__);                                                   \
  } while (0)

}  // namespace gacq
