// What this MI355X's HBM delivers to plain streaming kernels: 16-byte stores (fill, fill with non-temporal stores), 16-byte non-temporal
// loads (read) and a copy, 2 GiB each, 10 launches per figure, two grid sizes.  The split engines' two kernels sit on either side of one
// Z' round trip: the writer is bounded by the FILL rate, the reader by the READ rate -- not by the 8 TB/s of the data sheet.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/hbm_bandwidth.hip -o /tmp/hbm_bw && /tmp/hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void fill(float4* p, size_t n, float v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x; for (; i < n; i += s) p[i] = make_float4(v, v, v, v); }
__global__ void fill_nt(float4* p, size_t n, float v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x; for (; i < n; i += s) __builtin_nontemporal_store(f4{v, v, v, v}, reinterpret_cast<f4*>(p + i)); }
__global__ void rsum(const float4* p, size_t n, float* o) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x; float a = 0; for (; i < n; i += s) { f4 q = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + i)); a += q.x + q.y + q.z + q.w; } if (a == 1.2345f) *o = a; }
__global__ void cpy(const float4* a, float4* b, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x; for (; i < n; i += s) b[i] = a[i]; }
int main() {
  size_t bytes = (size_t)2 << 30, n = bytes / 16; float4 *a, *b; float* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256 * 8, 256 * 32}) {
    for (int k = 0; k < 4; k++) {
      const char* nm[4] = {"fill", "fill_nt", "read", "copy"};
      for (int w = 0; w < 2; w++) { if (k == 0) fill<<<grid, 256>>>(a, n, 1.f); if (k == 1) fill_nt<<<grid, 256>>>(a, n, 1.f); if (k == 2) rsum<<<grid, 256>>>(a, n, o); if (k == 3) cpy<<<grid, 256>>>(a, b, n); }
      hipEventRecord(e0);
      for (int r = 0; r < 10; r++) { if (k == 0) fill<<<grid, 256>>>(a, n, 1.f); if (k == 1) fill_nt<<<grid, 256>>>(a, n, 1.f); if (k == 2) rsum<<<grid, 256>>>(a, n, o); if (k == 3) cpy<<<grid, 256>>>(a, b, n); }
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d %-8s %.2f TB/s (%s)\n", grid, nm[k], (k == 3 ? 2.0 : 1.0) * bytes * 10 / (ms * 1e-3) / 1e12, k == 3 ? "read+write bytes" : "bytes");
    }
  }
  return 0;
}
