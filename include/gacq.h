/* gacq.h -- C ABI of the MI355X-native GNSS acquisition engine (libgacq.so).
 *
 * This is the drop-in boundary for ONE hot path of pmonta/GNSS-DSP-tools: the FFT-based parallel
 * code-phase search that every acquire-*.py defines inline as
 *     search(x, prn, doppler_search, ms) -> (metric, code, doppler)      acquire-gps-l1.py:18-40
 * The reference has no FFI of its own (pure Python); the entry points below are what a ctypes
 * binding for that function needs, and gnss-dsp-tools_amd/acquire.py is that binding
 * (INTEGRATION.md shows the stub a maintainer would drop into acquire-gps-l1.py).
 *
 * Conventions: plain pointers and sizes only; every function returns GACQ_OK (0) or a negative
 * GACQ_ERR_* code and never throws across the ABI; gacq_last_error() gives the message.
 * Host buffers are caller-owned; device buffers passed to *_dev entry points are caller-owned
 * device pointers (e.g. torch tensors' data_ptr()); everything else on the device is library-owned.
 * A gacq_ctx is bound to one HIP device and is single-threaded; a gacq_group drives several devices from one thread.
 */
#ifndef GACQ_H
#define GACQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GACQ_OK 0
#define GACQ_ERR_BAD_ARG (-1)
#define GACQ_ERR_UNKNOWN_CODE (-2)
#define GACQ_ERR_BAD_PRN (-3)
#define GACQ_ERR_HIP (-4)
#define GACQ_ERR_ROCFFT (-5)
#define GACQ_ERR_SHORT_INPUT (-6)
#define GACQ_ERR_NO_DEVICE (-7)
#define GACQ_ERR_INTERNAL (-8)
#define GACQ_ERR_UNSUPPORTED (-9)
/* Positive: the call succeeded and its results are valid, with a caveat.  gacq_search / gacq_search64 (the calls that wait for their
 * results) return this when the call's tie-safe re-evaluation list was too small (GACQ_OPT_TIE_CAP): every ambiguous pair of the call
 * kept its fp32 location.  The asynchronous device-buffer calls cannot report it (an overflow of theirs is reported by the next
 * gacq_search / gacq_search64 on the same context, or poll gacq_get_tie_stats()[2] after synchronising); gacq_search_batch and the
 * gacq_group_* calls do not return it either. */
#define GACQ_WARN_TIE_LIST_FULL 1

/* ---------------------------------------------------------------------------------------------
 * PRN chip generators (host only, no GPU needed).  `code` is the reference module name:
 * "gps.ca" == gnsstools/gps/ca.py, "galileo.e1b", "beidou.b1i", "glonass.ca", ...
 * Replaces: <module>.<sig>_code(prn) and <module>.code(prn,0,0,L/n,n)  (gnsstools/gps/ca.py:101-112)
 *           nco.boc11(0,0,L/n,n)                                         (gnsstools/nco.py:12-19)
 * ------------------------------------------------------------------------------------------- */
int gacq_code_count(void);
const char* gacq_code_name(int index);
int gacq_code_length(const char* code);                 /* chips, or GACQ_ERR_UNKNOWN_CODE */
double gacq_code_chip_rate(const char* code);
int gacq_code_prns(const char* code, int* out, int cap);/* returns number of valid PRNs */
int gacq_code_chips(const char* code, int prn, uint8_t* out, int cap);   /* {0,1} per chip */
int gacq_code_replica(const char* code, int prn, int n, int boc, float* out); /* +-1 samples */

/* ---------------------------------------------------------------------------------------------
 * Acquisition engine.
 * ------------------------------------------------------------------------------------------- */
typedef struct gacq_ctx gacq_ctx;
typedef struct gacq_sig gacq_sig;

/* The six knobs that distinguish the 28 FFT search() variants. */
typedef struct gacq_sigdesc {
  int code_length;   /* chips in one code period                      (ca.code_length)            */
  int n;             /* samples per coherent block at the internal fs (acquire-gps-l1.py:20)      */
  int pad;           /* 1: replica zero-extended to N=2n, windows x[b*n:(b+2)*n]                  */
                     /*    (acquire-beidou-b1i.py:24,30); 0: N=n (acquire-gps-l1.py:24,30)        */
  int boc;           /* 1: replica multiplied by nco.boc11 (acquire-galileo-e1b.py:25-26)         */
  int metric_mode;   /* 1: max/mean (acquire-gps-l1.py:35); 0: raw max (acquire-beidou-b1i.py:36) */
  int fold_code;     /* 1: code phase reported modulo code_length (acquire-beidou-b1i.py:39)      */
  double fs;         /* internal sampling rate in Hz                  (acquire-gps-l1.py:19)      */
} gacq_sigdesc;

/* Final per-item answer == the reference's return tuple (acquire-gps-l1.py:40), plus raw indices. */
typedef struct gacq_result {
  double metric;      /* m_metric                                                    */
  double code_chips;  /* m_code   = code_length*(float(idx)/n) [% code_length]       */
  double doppler_hz;  /* m_doppler                                                   */
  int idx;            /* argmax lag of the winning Doppler bin (-1 if no bin won)    */
  int d_index;        /* index of the winning bin in the Doppler grid (-1 if none)   */
} gacq_result;

/* Device-side record: best over a (local) Doppler slice for one (epoch, item). 16 bytes.
 * This is what crosses GPUs (one gather of these). */
typedef struct gacq_peak {
  double metric;      /* 0.0 when nothing beat the initial m_metric = 0 (acquire-gps-l1.py:25) */
  int idx;
  int d_index;
} gacq_peak;

int gacq_device_count(void);
int gacq_create(int device_id, gacq_ctx** out);
void gacq_destroy(gacq_ctx* ctx);
const char* gacq_last_error(gacq_ctx* ctx);             /* ctx may be NULL: last global error */

/* Use the caller's HIP stream (hipStream_t passed as void*; NULL = the ctx-owned stream). */
int gacq_set_stream(gacq_ctx* ctx, void* hip_stream);
/* Launch on the legacy default ("null") stream -- what torch.cuda.current_stream() is unless the caller
 * entered a stream context; needed so that work is ordered with collectives issued on that stream. */
int gacq_use_null_stream(gacq_ctx* ctx);
/* Engine selection: 0 = auto, 1 = rocFFT pipeline (any N), 2 = LDS-resident FFT kernels (N = 4096, 16384),
 * 3 = split engine, outer radix 31/16/4: prime-factor form for N = 61380 / 30690, rocFFT inner transforms otherwise (65536, 16384, ...),
 * 4 = split engine with the inner transforms on the LDS FFT kernels (N = 65536, 16384),
 * 5 = complex128 engine (any N): every value in fp64 on the device, as the reference computes (numpy complex128); agrees with it
 *     to ~1e-12 and is what a near-tie disagreement of an fp32 engine is bisected against.  N = 4096 with one block and one carrier
 *     runs as ONE fused kernel (fp64 transform resident in LDS), with several blocks or carriers as a forward + a correlate kernel on the
 *     same transform; N = 16384 and 65536 (B1I / B2I, GLONASS, E1B / E1C) as the split
 *     form 4 x / 16 x 4096 on the same transform -- forward spectra shared by the items, one Z' round trip of 32 N bytes per row and
 *     block, no rocFFT plan; N = 61380 / 30690 (the 10.23 Mcps scripts, E6, Xona X5) as 31 x M with hand-written fp64 DFT-31 stages
 *     around rocFFT's native length-M double transforms, which keeps the prime 31 out of rocFFT's Bluestein path (all three:
 *     GACQ_OPT_FUSED_C128); every other shape as the five-stage pipeline on rocFFT's double-precision transforms.  Never chosen by auto. */
int gacq_set_engine(gacq_ctx* ctx, int engine);
/* Upper bound in bytes for EACH of the two library-owned work buffers of a context -- the forward spectra and the correlation workspace;
 * both are allocated as searches need them, with 1/8 of headroom, and kept, so a context can hold about 2.25 x this (default 32 GiB
 * of the part's 288 GB).  A search whose forward spectra for one epoch exceed it is cut into Doppler slices that fit; the slices are
 * merged in grid order with strict '>' (acquire-gps-l1.py:36-39), so the result is the one of a single scan.  Until this is called the
 * library also keeps each buffer within the context's share of the device's FREE memory (80 % of it, split over the contexts alive on
 * the device and their two buffers each), so that several contexts, ranks or a torch allocator sharing a device cannot over-commit it;
 * a limit set here is taken as given.  Either way a search whose workspace cannot be allocated is re-run in passes of half the size
 * (down to 64 MiB) before an error is returned.  One pass of the split engines' correlation workspace is at most 4 GiB or eight
 * items' worth, whichever is more: the limit is the ceiling of a pass, not its size. */
int gacq_set_workspace_limit(gacq_ctx* ctx, size_t bytes);
/* Tuning switches of the launch path.  They are ctx state set through this call; nothing in the library reads the
 * environment.  Defaults in brackets. */
#define GACQ_OPT_FUSED_INNER 0  /* [1] engine 3, N = 61380 / 30690: 1 = prime-factor form (no twiddles at any level, conj-multiply fused into the */
                                /*     in-place inner inverse transforms; gacq_pfa.hip); 0 = Cooley-Tukey form with rocFFT inner transforms    */
                                /*     (other arithmetic: the cross-check)                                                                     */
#define GACQ_OPT_FUSED_16K 1    /* [1] N = 16384 with one carrier per item: forward + correlate in one kernel           */
#define GACQ_OPT_LDS_VARIANT 2  /* [-1] N = 16384: 16 = the radix-16 form of the one-workgroup transform (1024 threads x 16 points, three     */
                                /*     exchanges; gacq_ldsfft.hip) instead of the radix-32 form (512 x 32, two exchanges; gacq_lds16k.hip).    */
                                /*     Same results to fp32 rounding; the radix-32 form issues 13 % fewer VALU instructions at half the waves  */
                                /*     and measures 0-5 % faster on B1I, within 2 % on GLONASS (the in-run A/B of bench.py); other values: radix-32 form                */
#define GACQ_OPT_LDS_PCH 3      /* [0 = auto] items per workgroup of the LDS correlate kernels                          */
#define GACQ_OPT_SPLIT_PCH 4    /* [0 = auto] (epoch, item) rows per workgroup of the split engines' inner kernels      */
#define GACQ_OPT_SPLIT_TEAMS 5  /* retired (round 5): belonged to the Stockham inner kernel the prime-factor engine replaced; accepted, ignored */
#define GACQ_OPT_FUSED_4K 6     /* [1] N = 4096, B = 1, one carrier, >= 1024 (epoch, Doppler) units: forward + correlate in  */
                                /*     one kernel (no forward-spectra buffer); 2 = also for small batches                    */
#define GACQ_OPT_SPLIT_DT 7      /* [0 = auto] Doppler bins per workgroup of the prime-factor engine's inner kernel (1, 2 or 3): every   */
                                /*     code-spectrum row fetched serves that many correlation rows                                         */
#define GACQ_OPT_FE_GENERIC 8    /* [0] front-end: 1 = run the any-length mix + FIR kernels even for the reference's 161-tap filter     */
                                /*     (the specialised kernels produce the same bits; this is the A/B and test switch)              */
#define GACQ_OPT_LDS_UGROUP 9    /* [0 = auto, <= 64] N = 16384 correlate kernel: (epoch, Doppler) units a workgroup walks with one item's  */
                                /*     code spectrum held in registers (round 4: item-major order; the spectrum is fetched once per group)  */
#define GACQ_OPT_SEARCH1 10       /* retired (round 6): the single-launch search of round 3 (Doppler scan inside the correlate kernel) measured     */
                                /*     slower than three short launches and had no tie-safe re-evaluation; removed.  Accepted, ignored             */
#define GACQ_OPT_BAR_UPLOAD 11    /* [1] gacq_search: inputs up to 256 KiB are written by the host straight into fine-grained device       */
                                /*     memory through the PCIe BAR (large-BAR devices) instead of pinned staging + DMA                  */
#define GACQ_OPT_WATCH_RESULTS 12 /* [1] gacq_search (BAR upload path only): wait for completion by watching the pinned result records for  */
                                /*     the last kernel's stores (bounded spin, then hipStreamSynchronize) instead of a runtime sync          */
#define GACQ_OPT_TIE_SAFE 13      /* [1] tie-safe peak locations: every (epoch, item) whose winning lag or Doppler bin has a runner-up    */
                                /*     within GACQ_OPT_TIE_EPS_PPB of it is re-evaluated in complex128 on the device (the reference's     */
                                /*     arithmetic type, acquire-gps-l1.py:30-39) before its record is written, so the reported location is */
                                /*     the complex128 one and not whichever side of a near-tie the fp32 rounding fell on                   */
#define GACQ_OPT_TIE_EPS_PPB 14   /* [8000] relative gap, in parts per billion, below which two magnitudes / metrics count as tied         */
                                /*     (8e-6 ~ 6 x the worst fp32-vs-complex128 metric error observed); 1000000000 re-evaluates every row  */
#define GACQ_OPT_TIE_CAP 15       /* [0 = auto: 64 + (epochs x items) / 16] rows one call can re-evaluate.  A call whose candidate rows      */
                                /*     exceed it re-evaluates NONE: every ambiguous pair of the call keeps its fp32 answer (reproducible,    */
                                /*     the same on every rank of a sharded merge -- which pairs would have found room is not) and is         */
                                /*     counted in gacq_get_tie_stats()[2].  The automatic capacity is bounded (never below                  */
                                /*     16) so that the re-evaluation's row buffers (B x N x 8-24 bytes per listed row, allocated for the    */
                                /*     capacity) stay within an eighth of the workspace limit (gacq_set_workspace_limit), between 32 and    */
                                /*     256 MiB; an explicit value is taken as given                                                         */
#define GACQ_OPT_FUSED_C128 16     /* [1] engine 5 on the hand-written complex128 kernels where they exist -- N = 4096, B = 1, one carrier: the     */
                                /*     whole search row in one workgroup; N = 16384 / 65536: the split form (c128_split_*_kernel); N = 61380 /    */
                                /*     30690: the 31 x M form (c128_r31_*_kernel) -- instead of the five-stage rocFFT double-precision pipeline; */
                                /*     0 = always the pipeline (the cross-check)                                                                */
#define GACQ_OPT_SPLIT_MFMA 17     /* [0] prime-factor engine: the inverse DFT-31 of the outer stage as two real 16 x 16 matrices on the       */
                                /*     matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32) instead of packed math on the VALU.  Off: measured     */
                                /*     12-18 % slower -- the stage is paced by its load stream, not by arithmetic                             */
                                /*     (profiles/r05_engine3_prime_factor_vs_cooley_tukey_ab.log)                                             */
#define GACQ_NOPTS 18
int gacq_set_option(gacq_ctx* ctx, int option, long value);
int gacq_get_option(gacq_ctx* ctx, int option, long* value);

/* Build a signal: replicas from the built-in generators -> code spectra C_p resident on the device.
 * Replaces acquire-gps-l1.py:22-24 (c = ca.code(...); c = fft.fft(c)) for every PRN in `prns`. */
int gacq_signal_create(gacq_ctx* ctx, const gacq_sigdesc* desc, const char* code,
                       const int* prns, int nprn, gacq_sig** out);
/* Same, caller supplies the chips ({0,1} bytes, nprn rows of desc->code_length). */
int gacq_signal_create_chips(gacq_ctx* ctx, const gacq_sigdesc* desc, const uint8_t* chips,
                             int nprn, gacq_sig** out);
void gacq_signal_destroy(gacq_sig* sig);
int gacq_signal_fft_length(const gacq_sig* sig);        /* N = n or 2n */
/* Copy the device code spectrum of item `item` back (interleaved complex64, N values). Test hook. */
int gacq_signal_spectrum(gacq_sig* sig, int item, float* out_iq);

/* One search() per item over the same samples (host buffers, synchronous).
 *   x_iq      interleaved complex64, nsamp complex samples at desc->fs; needs (blocks+pad)*n
 *   items     indices into the signal's PRN list (0-based), nitems of them
 *   dopplers  the Doppler grid values np.arange(min,max,incr) produced (acquire-gps-l1.py:26)
 *   item_bias_hz  NULL, or per-item carrier bias added to the Doppler before the NCO
 *                 (GLONASS FDMA: 562500*chan, acquire-glonass-l1.py:28)
 *   blocks    number of non-coherent blocks B (the per-signal ms->B rule stays in the caller)
 * Replaces acquire-gps-l1.py:25-40 for all items at once (and the mp.Pool.map at :105-108).
 * Completion: the call returns when all `nitems` results are in `out`.  For small inputs on large-BAR devices (options
 * GACQ_OPT_BAR_UPLOAD / GACQ_OPT_WATCH_RESULTS, both on by default) it learns that by watching the pinned result records the last
 * kernel writes (bounded: 200 us, then hipStreamSynchronize, which is also where a faulted launch is reported) and does NOT
 * synchronise the ctx stream: trailing empty launches (the tie-safe re-evaluation kernels of a search without near-ties) may still
 * be queued when it returns.  They touch library-owned memory only; any later call on the ctx is ordered behind them. */
int gacq_search(gacq_sig* sig, const float* x_iq, size_t nsamp, const int* items, int nitems,
                const double* dopplers, int nd, const double* item_bias_hz, int blocks,
                gacq_result* out);

/* Batched, device-resident form: nepoch independent sample blocks already in HBM, results left in
 * HBM; asynchronous on the ctx stream.  d_x: complex64 [nepoch][nsamp]; d_out: gacq_peak
 * [nepoch][nitems] holding the best over `dopplers` (a rank's slice of the grid when sharded). */
int gacq_search_batch_dev(gacq_sig* sig, const void* d_x, size_t nsamp, int nepoch,
                          const int* items, int nitems, const double* dopplers, int nd,
                          const double* item_bias_hz, int blocks, void* d_out);

/* The same two entry points for complex128 samples -- the type the reference's search() is actually handed (np.interp's output,
 * acquire-gps-l1.py:94-96, used at :30-33).  x_iq / d_x_c128: interleaved (re, im) doubles.  Nothing is rounded where it matters:
 * engine 5 reads the samples as given; the fp32 engines search a complex64 rounding of them (made on the device) and every
 * (epoch, item) whose winner has a runner-up within GACQ_OPT_TIE_EPS_PPB -- which covers the rounding of the input, ~1e-7 -- is
 * re-evaluated in complex128 FROM THE UNROUNDED SAMPLES (tie-safe locations), so peak locations are those of the reference on its
 * own input and metrics agree within the 1e-5 of the fp32 engines.  gacq_search64 is synchronous (staged H2D copy, no BAR path). */
int gacq_search64(gacq_sig* sig, const double* x_iq, size_t nsamp, const int* items, int nitems,
                  const double* dopplers, int nd, const double* item_bias_hz, int blocks,
                  gacq_result* out);
int gacq_search_batch_dev64(gacq_sig* sig, const void* d_x_c128, size_t nsamp, int nepoch,
                            const int* items, int nitems, const double* dopplers, int nd,
                            const double* item_bias_hz, int blocks, void* d_out);

/* Host-buffer batch: nepoch independent sample blocks in HOST memory (x_iq: complex64 [nepoch][nsamp]) -> out [nepoch][nitems]
 * on the host.  Synchronous for the caller, pipelined inside: the epochs go through a ring of pinned staging slots in chunks,
 * the H2D copy of chunk c+1 runs on a copy stream under the kernels of chunk c, and the peak records are written by the last
 * kernel into device-visible pinned memory (no D2H copy).  Amortises the launch latency of gacq_search over many epochs for
 * callers without a device-memory framework; results are identical to nepoch calls of gacq_search. */
int gacq_search_batch(gacq_sig* sig, const float* x_iq, size_t nsamp, int nepoch, const int* items, int nitems,
                      const double* dopplers, int nd, const double* item_bias_hz, int blocks, gacq_result* out);

/* ---------------------------------------------------------------------------------------------
 * Several GPUs driven from ONE process (no torch, no MPI): a group owns one context per device.  gacq_group_search_batch cuts
 * the Doppler grid into contiguous slices, one per device, while every device still gets >= 4 bins (forward FFTs are not
 * duplicated; every device holds all code spectra), otherwise the item list; all devices read the same samples from one
 * portable pinned staging buffer; the per-device peak records (16 B per epoch and item) come back through pinned host memory
 * and are merged in global Doppler order with strict '>' -- the scan order of acquire-gps-l1.py:36-39, so the result equals
 * a one-device search bit for bit.  device_ids may name a device more than once.  Replaces the mp.Pool.map over PRNs
 * (acquire-gps-l1.py:105-108) for a single-process caller; one-process-per-GPU callers use torch.distributed (sharded.py).
 * ------------------------------------------------------------------------------------------- */
typedef struct gacq_group gacq_group;
typedef struct gacq_gsig gacq_gsig;
int gacq_group_create(const int* device_ids, int ndev, gacq_group** out);
void gacq_group_destroy(gacq_group* group);
int gacq_group_size(const gacq_group* group);
gacq_ctx* gacq_group_member(gacq_group* group, int k);   /* member context, e.g. for gacq_set_engine / gacq_set_option */
/* How the members' peak records meet.  GACQ_EXCHANGE_HOST [default]: each member's last kernel writes its records into pinned host
 * memory and the host merges them (near-ties re-evaluated on a member).  GACQ_EXCHANGE_RCCL: the records stay on the devices, ONE
 * ncclAllGather per chunk moves them over xGMI (single-process communicators from ncclCommInitAll; librccl.so is loaded on this
 * call, the library does not link against it) and a member merges them on the device with the tie-safe merge -- the same result bit
 * for bit.  Needs distinct devices; GACQ_ERR_UNSUPPORTED otherwise or when RCCL cannot be loaded (the group then stays on the host
 * merge).  Applies to Doppler-sliced searches; an item split needs no merge.  EXPERIMENTAL beyond one device: the mode has run on one
 * MI355X (a one-rank collective) and never on several -- no multi-GPU node was available to rounds 1-6.  A failed collective aborts the
 * communicators (ncclCommAbort: nothing stays queued on a member's stream), returns the group to GACQ_EXCHANGE_HOST and reports the
 * error; the call can be repeated. */
#define GACQ_EXCHANGE_HOST 0
#define GACQ_EXCHANGE_RCCL 1
int gacq_group_set_exchange(gacq_group* group, int mode);
const char* gacq_group_last_error(gacq_group* group);
int gacq_group_signal_create(gacq_group* group, const gacq_sigdesc* desc, const char* code, const int* prns, int nprn,
                             gacq_gsig** out);
void gacq_group_signal_destroy(gacq_gsig* sig);
int gacq_group_search_batch(gacq_gsig* sig, const float* x_iq, size_t nsamp, int nepoch, const int* items, int nitems,
                            const double* dopplers, int nd, const double* item_bias_hz, int blocks, gacq_result* out);

/* Tie-safe bookkeeping since gacq_create (synchronises the ctx stream): out[0] ambiguous (epoch, item) pairs found, out[1] rows
 * re-evaluated in complex128, out[2] pairs that kept their fp32 answer because the re-evaluation list was full (GACQ_OPT_TIE_CAP and its
 * memory bound) -- anything but 0 here means locations that are NOT guaranteed to be the complex128 ones: treat it as a warning
 * (bench.py prints one), out[3] pairs whose location the re-evaluation changed.  FFT lengths with a prime factor the complex128 row kernel
 * does not carry (it has 2, 3, 5, 7, 11, 13 and 31: every length the reference's scripts use) are searched without tie-safe locations
 * and are not counted here. */
int gacq_get_tie_stats(gacq_ctx* ctx, long long out[4]);

/* Device-side shard merge: d_peaks [nshard][n] (as gathered from the ranks, shard s covering Doppler
 * indices from shard_d0[s]) -> d_out [n] with global d_index; shards scanned in order with strict '>'
 * so the lowest Doppler bin wins ties exactly like the reference's scan (acquire-gps-l1.py:36-39).
 * Asynchronous on the ctx stream. */
int gacq_merge_peaks_dev(gacq_ctx* ctx, const void* d_peaks, int nshard, const int* shard_d0, long n,
                         void* d_out);

/* The same merge for callers that run with tie-safe locations (GACQ_OPT_TIE_SAFE, the default): shard winners whose metrics come
 * within GACQ_OPT_TIE_EPS_PPB of the best one are re-evaluated in complex128 on this device before the scan decides -- every rank
 * holds the samples and all code spectra, so any rank can do it -- which makes the merged record the one an unsharded tie-safe
 * search writes, bit for bit.  The search arguments are those of the search the shards came from, with the FULL Doppler grid
 * (d_x: [nepoch][nsamp] on the device).  Asynchronous on the ctx stream. */
int gacq_merge_peaks_tiesafe_dev(gacq_sig* sig, const void* d_x, size_t nsamp, int nepoch, const int* items, int nitems,
                                 const double* dopplers, int nd, const double* item_bias_hz, int blocks, const void* d_peaks,
                                 int nshard, const int* shard_d0, void* d_out);
/* ... with the samples in complex128 (see gacq_search_batch_dev64) */
int gacq_merge_peaks_tiesafe_dev64(gacq_sig* sig, const void* d_x_c128, size_t nsamp, int nepoch, const int* items, int nitems,
                                 const double* dopplers, int nd, const double* item_bias_hz, int blocks, const void* d_peaks,
                                 int nshard, const int* shard_d0, void* d_out);

/* Host-side last step (acquire-gps-l1.py:36-40): merge `nshard` peaks per item in shard order with
 * strict '>' (shard s covers Doppler indices [shard_d0[s], ...)), then convert to the reference's
 * return tuple.  peaks: [nshard][nitems]; dopplers: the FULL grid.  Needs no GPU. */
int gacq_finalize(const gacq_sigdesc* desc, const gacq_peak* peaks, int nshard, const int* shard_d0,
                  int nitems, const double* dopplers, int nd, gacq_result* out);

/* ---------------------------------------------------------------------------------------------
 * Front-end (SURVEY.md section 8f "next #1"): what every acquire script does to the file before search().
 * Replaces acquire-gps-l1.py:87-96: nco.mix (fixed-point table NCO, gnsstools/nco.py:30-41),
 * scipy.signal.filtfilt(h,[1],x) with a firwin(161, cutoff, 'hann') low-pass, np.interp resample.
 * ------------------------------------------------------------------------------------------- */
/* scipy.signal.firwin(ntaps, cutoff_norm, window='hann'); cutoff_norm = cutoff_hz / (fs/2). Host only. */
int gacq_firwin_hann(int ntaps, double cutoff_norm, double* taps);
/* d_iq_int8: interleaved signed 8-bit I/Q on the device (nsamp_in complex samples at fs_in);
 * d_out: complex64 [nsamp_out] at fs_out, directly consumable by gacq_search_batch_dev. Asynchronous. */
int gacq_frontend_dev(gacq_ctx* ctx, const void* d_iq_int8, size_t nsamp_in, double fs_in, double carrier_offset_hz,
                      const double* taps, int ntaps, double fs_out, size_t nsamp_out, void* d_out);

/* The main program of an acquire script in one call (acquire-gps-l1.py:78-108: read block, mix / filter / resample, search every item):
 * iq_int8 = nsamp_in interleaved signed 8-bit I/Q samples at fs_in in HOST memory; the block is uploaded, gacq_frontend_dev produces
 * nsamp_out samples at the signal's internal rate on the device and gacq_search_batch_dev searches them there; out[nitems] as
 * gacq_search.  For callers without a device-memory framework (the command-line shim runs on it: no torch import on its path).
 * Synchronous; GACQ_WARN_TIE_LIST_FULL as gacq_search. */
int gacq_acquire_int8(gacq_sig* sig, const int8_t* iq_int8, size_t nsamp_in, double fs_in, double carrier_offset_hz, const double* taps,
                      int ntaps, size_t nsamp_out, const int* items, int nitems, const double* dopplers, int nd,
                      const double* item_bias_hz, int blocks, gacq_result* out);

/* ---------------------------------------------------------------------------------------------
 * Time-domain long-code searches (SURVEY.md section 8f "next #3"): acquire-gps-l2cl.py:15-30,
 * acquire-glonass-l1-p.py / -l2-p.py:15-33.  For candidate k:
 *   q[k] = sum_block | sum_i x[n*block+i] * code[floor(phase0[k*blocks+block] + (chip_rate/fs)*i) mod L] * nco[i] |
 * x_iq: host complex64 at the file rate fs (after the carrier-offset wipe-off); carrier_hz = Doppler (+ FDMA channel bias);
 * phase0: the caller's (chips % L) + frac per (k, block), in chips -- finite and within +-2^31 chips (anything else is
 * GACQ_ERR_BAD_ARG; the reference's expressions stay below two code periods).  Synchronous; q_out[K] in fp64.
 * ------------------------------------------------------------------------------------------- */
int gacq_longcode_search(gacq_ctx* ctx, const float* x_iq, size_t nsamp, double fs, const char* code, int prn,
                         double carrier_hz, const double* phase0, int K, int blocks, int n, double* q_out);
/* Same from the file's raw samples: iq_int8 = interleaved signed 8-bit I/Q (host, nsamp complex samples, gnsstools/io.py:3-12);
 * the carrier-offset wipe-off nco.mix(x,-coffset/fs,0) (acquire-gps-l2cl.py:72, fixed-point table NCO of gnsstools/nco.py:30-41)
 * runs on the device before the search, so the caller does no per-sample work at all. */
int gacq_longcode_search_int8(gacq_ctx* ctx, const int8_t* iq_int8, size_t nsamp, double fs, double carrier_offset_hz,
                              const char* code, int prn, double carrier_hz, const double* phase0, int K, int blocks, int n,
                              double* q_out);

/* ---------------------------------------------------------------------------------------------
 * Batched code correlators for the tracking hand-off (SURVEY.md section 8f "next #4"): the reference's
 * <module>.correlate(x, prn, chips, frac, incr, c[, boc11]) (gnsstools/gps/ca.py:120-128 and its BOC / CBOC / TMBOC /
 * RZ variants) for K (PRN, start phase, rate) triples over one block of n samples in one launch.
 * kind: 0 plain, 1 BOC(1,1), 2 CBOC (e1b/e1c), 3 TMBOC (l1cp), 4 RZ first half chip (l2cm), 5 RZ second half (l2cl).
 * out_iq: K interleaved complex128 sums.  Synchronous; x_iq is a host buffer.
 * ------------------------------------------------------------------------------------------- */
int gacq_correlate_batch(gacq_ctx* ctx, const float* x_iq, size_t n, const char* code, int kind, const int* prns,
                         const double* chips, const double* frac, const double* incr, int K, double* out_iq);

/* Device-resident forms of the two entry points above: d_x is complex64 ALREADY on the device (gacq_frontend_dev's or
 * gacq_mix_int8_dev's output, or the very samples gacq_search_batch_dev just searched) -- the reference keeps one x in memory from
 * acquisition into the long-code search (acquire-gps-l2cl.py:60-76) and into correlate() every millisecond
 * (track-gps-l1.py:48-50).  Only the start phases / the K correlator specs travel to the device; results come back through pinned
 * memory into q_out / out_iq.  Same kernels, byte-identical results. */
int gacq_longcode_search_dev(gacq_ctx* ctx, const void* d_x, size_t nsamp, double fs, const char* code, int prn, double carrier_hz,
                             const double* phase0, int K, int blocks, int n, double* q_out);
int gacq_correlate_batch_dev(gacq_ctx* ctx, const void* d_x, size_t n, const char* code, int kind, const int* prns,
                             const double* chips, const double* frac, const double* incr, int K, double* out_iq);
/* nco.mix(x, -coffset/fs, 0) alone on the device (gnsstools/nco.py:30-41; acquire-gps-l2cl.py:72): d_iq_int8 interleaved signed 8-bit
 * I/Q -> d_out complex64 [nsamp] at the same rate.  Asynchronous on the ctx stream. */
int gacq_mix_int8_dev(gacq_ctx* ctx, const void* d_iq_int8, size_t nsamp, double fs, double carrier_offset_hz, void* d_out);

/* Per-stage GPU time from HIP events recorded on the launch stream (profiling aid for bench.py).
 * Stages: 0 mix/forward, 1 forward FFT (rocFFT), 2 conj-multiply, 3 inverse FFT (rocFFT),
 *         4 magnitude/peak reduce, 5 best-over-Doppler, 6 fused correlate kernel (LDS FFT). */
#define GACQ_NSTAGES 7
int gacq_set_profiling(gacq_ctx* ctx, int enabled);
int gacq_get_stage_time(gacq_ctx* ctx, int stage, double* total_ms, long* launches);
int gacq_reset_stage_times(gacq_ctx* ctx);
const char* gacq_stage_name(int stage);

/* What this device's HBM delivers to a tuned streaming kernel (eight 16-byte non-temporal accesses in flight per lane): kind 0 = fill
 * (stores), 1 = read, 2 = copy (read + write bytes counted), 3 = fill followed by a read of the same buffer (both counted), 4 / 5 / 6 = kinds 0 / 1 / 3 with default-policy instead of non-temporal accesses; `bytes` per launch, `reps` timed launches after two warm-up launches,
 * HIP events on the ctx stream.  The yardstick bench.py holds the HBM-bound kernels against next to the 8 TB/s of the data sheet
 * (roofline.stream_ceiling), measured in the run that reports it.  Synchronous; allocates and frees its own buffers. */
int gacq_stream_probe(gacq_ctx* ctx, int kind, size_t bytes, int reps, double* gbytes_per_s);

/* A HIP stream whose kernels run on a subset of the device's compute units (hipExtStreamCreateWithCUMask; bit b of the mask = logical
 * CU b, which the driver deals round-robin over the XCDs, so the low m bits are m / 8 CUs of every XCD).  For searches of several
 * signals run side by side (the reference runs its PRNs side by side in a Pool, acquire-gps-l1.py:105-108; a cold start searches
 * several signals): the contexts of the memory-bound signals (N = 65536, 61380: Z' round trip through HBM) get a masked stream with
 * gacq_set_stream, the others keep the whole device, and the memory-bound kernels run under the arithmetic of the others
 * (ShardedSearch.search_jobs_async).  The stream is the caller's: destroy it with gacq_stream_destroy after the contexts that used it. */
int gacq_stream_create_cu_mask(int device_id, const uint32_t* cu_mask, int nwords, void** hip_stream_out);
int gacq_stream_destroy(int device_id, void* hip_stream);
/* Where the workgroups of a launch on the ctx stream land: where[i] = xcc_id << 8 | se_id << 5 | sh_id << 4 | cu_id of workgroup i
 * (nworkgroups one-wave workgroups that stay resident 0.2 ms each) -- how tests and tools check what a CU mask selects. */
int gacq_cu_census(gacq_ctx* ctx, int nworkgroups, unsigned* where);

/* Full accumulated magnitude row q[0..N) for one (item, doppler) -- debugging / golden rows. */
int gacq_debug_row(gacq_sig* sig, const float* x_iq, size_t nsamp, int item, double doppler,
                   double bias_hz, int blocks, float* q_out);

/* Table-NCO index vector idx[i] = floor((0 + f*i)*1024) mod 1024, i < N (gnsstools/nco.py:6-9), f = -(bias + doppler)/fs as a
 * search forms it, computed ON THE DEVICE by the forward kernel of the chosen path with the very expression it uses for its
 * table lookup (the kernels are instantiated in a dump mode that stores the index instead of using it).  Index work is
 * bit-exact against the reference; a wrong index moves the metric by less than the 1e-5 tolerance, so it gets its own check.
 * kernel: 1 mix_nco_kernel (rocFFT pipeline), 2 lds_forward_kernel / lds16k_forward_kernel, 3 split_outer_forward_kernel,
 *         4 lds16k_fused_kernel, 5 pfa_outer_forward_kernel.  GACQ_ERR_UNSUPPORTED when that kernel does not serve the signal's N. */
int gacq_debug_nco_indices(gacq_sig* sig, int kernel, double doppler, double bias_hz, int* idx_out);

/* Number of rocFFT plans the context has created so far (a plan for a new length costs 0.5-2 s per process: runtime-compiled kernels).
 * The default engines create none -- code spectra included -- for every FFT length of the reference's scripts; the rocFFT pipeline
 * (engine 1), engine 5 outside N = 4096, GACQ_OPT_FUSED_INNER = 0 and gacq_signal_spectrum do, on first use.  Negative: error. */
int gacq_debug_fft_plans(gacq_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* GACQ_H */
