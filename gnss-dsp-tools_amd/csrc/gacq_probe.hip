// What this device's HBM delivers to tuned streaming kernels, measured where the benchmark runs: the yardstick the HBM-bound kernels
// of the split engines are held against next to the 8 TB/s of the data sheet (bench.py: roofline.stream_ceiling).  Each lane keeps
// eight independent 16-byte accesses in flight, non-temporal (read-once / write-once streams), one 32 KiB tile per workgroup and
// iteration, eight workgroups per CU -- the shape MI355X_MICROARCH.md's 6.3 TB/s float4 copy was measured with; the round-3
// microbenchmark (one 16-byte access per lane and iteration, tools/hbm_bandwidth.hip) stayed 25 % below it for the copy.
#include "gacq_common.h"

using namespace gacq;

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kU = 8;

template <bool NT = true>
__global__ __launch_bounds__(256) void stream_fill_kernel(f4* __restrict__ p, size_t ntiles, float v) {
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    f4* dst = p + tile * (256 * kU) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (NT) __builtin_nontemporal_store(f4{v, v, v, v}, dst + 256 * u);
      else dst[256 * u] = f4{v, v, v, v};
    }
  }
}

template <bool NT = true>
__global__ __launch_bounds__(256) void stream_read_kernel(const f4* __restrict__ p, size_t ntiles, float* __restrict__ sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const f4* src = p + tile * (256 * kU) + threadIdx.x;
    f4 v[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) v[u] = NT ? __builtin_nontemporal_load(src + 256 * u) : src[256 * u];
#pragma unroll
    for (int u = 0; u < kU; u++) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = acc.x;      // never true for the fill pattern: keeps the loads alive
}

__global__ __launch_bounds__(256) void stream_copy_kernel(const f4* __restrict__ a, f4* __restrict__ b, size_t ntiles) {
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const f4* src = a + tile * (256 * kU) + threadIdx.x;
    f4* dst = b + tile * (256 * kU) + threadIdx.x;
    f4 v[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) v[u] = __builtin_nontemporal_load(src + 256 * u);
#pragma unroll
    for (int u = 0; u < kU; u++) __builtin_nontemporal_store(v[u], dst + 256 * u);
  }
}

}  // namespace

extern "C" int gacq_stream_probe(gacq_ctx* ctx, int kind, size_t bytes, int reps, double* gbytes_per_s) {
  const bool plain = kind >= 4;          // 4..6: kinds 0, 1, 3 with default-policy accesses instead of non-temporal ones
  if (plain) kind = kind > 6 ? -1 : (kind == 6 ? 3 : kind - 4);
  if (!ctx || !gbytes_per_s || kind < 0 || kind > 3 || reps <= 0 || reps > 1000 || bytes < ((size_t)1 << 20) || bytes > ((size_t)16 << 30))
    return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_stream_probe: kind 0..3, 1 MiB <= bytes <= 16 GiB, 1 <= reps <= 1000");
  GACQ_DEVICE(ctx);
  const size_t tile_bytes = (size_t)256 * kU * 16;
  const size_t ntiles = bytes / tile_bytes;
  void *a = nullptr, *b = nullptr;
  float* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = GACQ_OK;
  auto cleanup = [&]() {
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  };
  if (hipMalloc(&a, ntiles * tile_bytes) != hipSuccess || (kind == 2 && hipMalloc(&b, ntiles * tile_bytes) != hipSuccess) ||
      hipMalloc((void**)&sink, 4) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    return set_error(ctx, GACQ_ERR_HIP, "gacq_stream_probe: allocation of %zu bytes failed", bytes);
  }
  hipDeviceProp_t prop;
  const int cus = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 256;
  const dim3 grid((unsigned)std::min<size_t>(ntiles, (size_t)cus * 8)), block(256);
  hipStream_t st = ctx->stream;
  auto launch = [&]() {
    if (kind == 3 || kind == 0) {      // 3: write then read back -- what a consumer gets of data a producer has just written
      if (plain) hipLaunchKernelGGL(stream_fill_kernel<false>, grid, block, 0, st, (f4*)a, ntiles, 1.0f);
      else hipLaunchKernelGGL(stream_fill_kernel<true>, grid, block, 0, st, (f4*)a, ntiles, 1.0f);
    }
    if (kind == 3 || kind == 1) {
      if (plain) hipLaunchKernelGGL(stream_read_kernel<false>, grid, block, 0, st, (const f4*)a, ntiles, sink);
      else hipLaunchKernelGGL(stream_read_kernel<true>, grid, block, 0, st, (const f4*)a, ntiles, sink);
    }
    if (kind <= 1 || kind == 3) return;
    hipLaunchKernelGGL(stream_copy_kernel, grid, block, 0, st, (const f4*)a, (f4*)b, ntiles);
  };
  hipLaunchKernelGGL(stream_fill_kernel<true>, grid, block, 0, st, (f4*)a, ntiles, 1.0f);      // defined contents for the read / copy probes
  for (int w = 0; w < 2; w++) launch();
  if (hipEventRecord(e0, st) != hipSuccess) rc = set_error(ctx, GACQ_ERR_HIP, "gacq_stream_probe: event record failed");
  for (int r = 0; r < reps && rc == GACQ_OK; r++) launch();
  float ms = 0.f;
  if (rc == GACQ_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess))
    rc = set_error(ctx, GACQ_ERR_HIP, "gacq_stream_probe: timing failed");
  if (rc == GACQ_OK) *gbytes_per_s = (kind >= 2 ? 2.0 : 1.0) * (double)(ntiles * tile_bytes) * reps / ((double)ms * 1e-3) / 1e9;
  cleanup();
  return rc;
}

// ---- streams restricted to a subset of the CUs --------------------------------------------------------------------------------
// A cold-start search runs signals of two kinds back to back: LDS-resident transforms (N <= 16384; the vector pipes bind them and one
// workgroup owns a CU's register file) and split transforms with a Z' round trip through HBM (N = 65536, 61380; the memory system binds
// them).  Launched on two streams they share the chip: the memory-bound kernels hide under the arithmetic of the others.  A CU mask on the
// memory-bound stream keeps its many small workgroups on a fixed set of CUs, so that they do not keep the large workgroups of the other
// stream from ever finding an empty CU (ShardedSearch.search_jobs_async, bench.py config 5).
namespace {

__global__ void cu_census_kernel(unsigned* __restrict__ out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // stay resident for a while, so that the workgroups of one launch spread over every CU the stream may use
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  // HW_ID (gfx9): cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID: xcc_id [3:0]
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
}

}  // namespace

extern "C" int gacq_stream_create_cu_mask(int device, const uint32_t* cu_mask, int nwords, void** stream_out) {
  if (!stream_out) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_stream_create_cu_mask: stream_out is NULL");
  *stream_out = nullptr;
  if (!cu_mask || nwords <= 0 || nwords > 32 || device < 0 || device >= gacq_device_count())
    return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_stream_create_cu_mask: device in range, 1..32 mask words");
  bool any = false;
  for (int i = 0; i < nwords; i++) any |= cu_mask[i] != 0;
  if (!any) return set_error(nullptr, GACQ_ERR_BAD_ARG, "gacq_stream_create_cu_mask: empty mask");
  DeviceGuard guard(device);
  hipStream_t st = nullptr;
  if (!guard.ok || hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, cu_mask) != hipSuccess) {
    (void)hipGetLastError();
    return set_error(nullptr, GACQ_ERR_HIP, "gacq_stream_create_cu_mask: hipExtStreamCreateWithCUMask failed on device %d", device);
  }
  *stream_out = (void*)st;
  return GACQ_OK;
}

extern "C" int gacq_stream_destroy(int device, void* stream) {
  if (!stream) return GACQ_OK;
  DeviceGuard guard(device);
  if (!guard.ok || hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
    (void)hipGetLastError();
    return set_error(nullptr, GACQ_ERR_HIP, "gacq_stream_destroy: failed on device %d", device);
  }
  return GACQ_OK;
}

extern "C" int gacq_cu_census(gacq_ctx* ctx, int nworkgroups, unsigned* where) {
  if (!ctx || !where || nworkgroups <= 0 || nworkgroups > (1 << 16))
    return set_error(ctx, GACQ_ERR_BAD_ARG, "gacq_cu_census: 1 <= nworkgroups <= 65536");
  GACQ_DEVICE(ctx);
  unsigned* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(unsigned) * nworkgroups) != hipSuccess) { (void)hipGetLastError(); return set_error(ctx, GACQ_ERR_HIP, "gacq_cu_census: allocation failed"); }
  hipLaunchKernelGGL(cu_census_kernel, dim3(nworkgroups), dim3(64), 0, ctx->stream, d, 20000);      // 100 MHz counter: 0.2 ms
  int rc = GACQ_OK;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(where, d, sizeof(unsigned) * nworkgroups, hipMemcpyDeviceToHost) != hipSuccess || hipGetLastError() != hipSuccess)
    rc = set_error(ctx, GACQ_ERR_HIP, "gacq_cu_census: launch failed");
  (void)hipFree(d);
  return rc;
}
