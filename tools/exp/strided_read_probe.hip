// Experiment (not part of the library): how fast can the access pattern of split_outer_inverse_kernel be read?  A "group" is R rows of
// Mp complex64 values; a thread owns one column (8-byte loads, R in flight) or two adjacent columns (16-byte loads).  Values are
// summed into one float per thread so that nothing is optimised away.  hipcc --offload-arch=gfx950 -O3 strided_read_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, int W>   // W = floats per load: 2 (8 bytes) or 4 (16 bytes)
__global__ __launch_bounds__(256) void read_cols(const float* __restrict__ z, float* __restrict__ out, int Mp, int chunks) {
  typedef float vt __attribute__((ext_vector_type(W)));
  const unsigned blk = blockIdx.x;
  const int chunk = blk % chunks;
  const long g = blk / chunks;
  const long pos = (long)chunk * 256 + threadIdx.x;             // in units of one load
  const vt* src = reinterpret_cast<const vt*>(z + g * (long)R * Mp * 2) + pos;
  vt v[R];
#pragma unroll
  for (int k = 0; k < R; k++) v[k] = __builtin_nontemporal_load(src + (long)k * (Mp * 2 / W));
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < R; k++)
    for (int i = 0; i < W; i++) s += v[k][i];
  if (s == 12345.678f) out[blk * 256 + threadIdx.x] = s;
}

template <int R, int W>
float run(const float* z, float* out, long groups, int M, int Mp, int reps) {
  const int chunks = (M * 2 / W + 255) / 256;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((read_cols<R, W>), dim3((unsigned)(groups * chunks)), dim3(256), 0, 0, z, out, Mp, chunks);
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((read_cols<R, W>), dim3((unsigned)(groups * chunks)), dim3(256), 0, 0, z, out, Mp, chunks);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return (float)((double)groups * R * M * 8.0 * reps / (ms * 1e-3) / 1e9);
}

int main() {
  const int M = 4096, Mp = 4096;
  const long groups16 = 2304;                                   // 2304 x 16 x 4096 x 8 B = 1.2 GB (E1B, 36 items x 64 bins)
  const long groups31 = 1000;                                   // R = 31, M = 1980 (L5I)
  float *z, *out;
  (void)hipMalloc(&z, (size_t)groups16 * 16 * Mp * 8);
  (void)hipMalloc(&out, 64 << 20);
  (void)hipMemset(z, 0, (size_t)groups16 * 16 * Mp * 8);
  for (int rep = 0; rep < 2; rep++) {
    printf("R=16 M=4096: 8-byte loads %.0f GB/s, 16-byte loads %.0f GB/s\n", run<16, 2>(z, out, groups16, M, Mp, 10), run<16, 4>(z, out, groups16, M, Mp, 10));
    printf("R=31 M=1980 (pitch 1984): 8-byte loads %.0f GB/s, 16-byte loads %.0f GB/s\n", run<31, 2>(z, out, groups31, 1980, 1984, 10), run<31, 4>(z, out, groups31, 1980, 1984, 10));
    printf("R=8  M=4096: 8-byte loads %.0f GB/s, 16-byte loads %.0f GB/s\n", run<8, 2>(z, out, groups16 * 2, M, Mp, 10), run<8, 4>(z, out, groups16 * 2, M, Mp, 10));
  }
  return 0;
}
