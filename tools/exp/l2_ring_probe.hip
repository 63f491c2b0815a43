// Experiment (not part of the library): can a producer workgroup hand a tile to a consumer workgroup ON THE SAME XCD through that XCD's L2
// without the data crossing HBM and without an agent-scope release (buffer_wbl2)?  That is what a fused writer/reader of the split engines'
// Z' round trip would need.  Workgroups read their XCC id (s_getreg HW_REG_XCC_ID), take a ticket in their XCD's team and pair up inside
// the team; per iteration each writes its 64 KB slot (plain 16-byte stores, iteration-tagged), waits for its own stores (s_waitcnt
// vmcnt(0)), bumps its flag (agent-scope atomic), polls the partner's flag, reads the partner's slot with sc1 (L1-bypassing) loads and
// counts words that do not carry the expected tag.  2 MB of slots per XCD: the ring fits the 4 MB L2.
//   hipcc --offload-arch=gfx950 -O3 l2_ring_probe.hip -o l2probe && ./l2probe            (and under rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kSlotWords = 64 * 1024 / 8;      // 8-byte words per slot
constexpr int kMaxTeam = 64;

struct Team { unsigned count; unsigned pad[15]; };

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

__global__ __launch_bounds__(256) void ring_kernel(unsigned long long* __restrict__ slots, unsigned* __restrict__ flags, Team* __restrict__ teams,
                                                   unsigned long long* __restrict__ stale, unsigned* __restrict__ census, int iters, int pair) {
  __shared__ unsigned s_rank, s_xcc, s_team;
  const int t = threadIdx.x;
  if (t == 0) {
    s_xcc = xcc_id();
    s_rank = atomicAdd(&teams[s_xcc].count, 1u);
    atomicAdd(&census[s_xcc], 1u);
  }
  __syncthreads();
  const unsigned xcc = s_xcc, rank = s_rank;
  // wait until the whole grid has taken its ticket (the team size is needed to pick a partner): all blocks are resident (1 per CU)
  if (t == 0) {
    unsigned total;
    do {
      total = 0;
      for (int x = 0; x < 8; x++) total += __hip_atomic_load(&teams[x].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_sleep(8);
    } while (total < gridDim.x);
    s_team = __hip_atomic_load(&teams[xcc].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned team = s_team;
  unsigned partner = pair ? (rank ^ 1u) : rank;
  if (partner >= team) partner = rank;             // odd team: the last one reads its own slot
  unsigned long long* mine = slots + ((size_t)xcc * kMaxTeam + rank) * kSlotWords;
  const unsigned long long* theirs = slots + ((size_t)xcc * kMaxTeam + partner) * kSlotWords;
  unsigned* my_flag = flags + (xcc * kMaxTeam + rank) * 16;
  unsigned* their_flag = flags + (xcc * kMaxTeam + partner) * 16;
  unsigned long long bad = 0;
  for (int it = 1; it <= iters; it++) {
    // the partner must have finished READING my slot of the previous iteration before I overwrite it: second flag word
    if (t == 0 && it > 1) {
      while (__hip_atomic_load(their_flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it - 1)) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const unsigned long long tag = ((unsigned long long)it << 32) | (xcc << 16) | rank;
#pragma unroll 8
    for (int w = t * 2; w < kSlotWords; w += 512) {
      ulonglong2 v = make_ulonglong2(tag + ((unsigned long long)w << 48), tag + ((unsigned long long)(w + 1) << 48));
      *reinterpret_cast<ulonglong2*>(mine + w) = v;                           // plain 16-byte stores
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // my stores have reached the L2
    __syncthreads();
    if (t == 0) {
      __hip_atomic_store(my_flag, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const unsigned long long want = ((unsigned long long)it << 32) | (xcc << 16) | partner;
#pragma unroll 8
    for (int w = t; w < kSlotWords; w += 256) {
      const unsigned long long v = __hip_atomic_load(theirs + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dwordx2 sc1
      bad += (v != want + ((unsigned long long)w << 48)) ? 1 : 0;
    }
    __syncthreads();
    if (t == 0) __hip_atomic_store(my_flag + 1, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // done reading their slot
  }
  if (bad) atomicAdd(stale, bad);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  unsigned long long *slots, *stale;
  unsigned *flags, *census;
  Team* teams;
  (void)hipMalloc(&slots, (size_t)8 * kMaxTeam * kSlotWords * 8);
  (void)hipMalloc(&flags, 8 * kMaxTeam * 16 * 4);
  (void)hipMalloc(&teams, sizeof(Team) * 8);
  (void)hipMalloc(&stale, 8);
  (void)hipMalloc(&census, 32);
  for (int pair = 0; pair < 2; pair++) {
    (void)hipMemset(slots, 0, (size_t)8 * kMaxTeam * kSlotWords * 8);
    (void)hipMemset(flags, 0, 8 * kMaxTeam * 16 * 4);
    (void)hipMemset(teams, 0, sizeof(Team) * 8);
    (void)hipMemset(stale, 0, 8);
    (void)hipMemset(census, 0, 32);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(ring_kernel, dim3(cus), dim3(256), 0, 0, slots, flags, teams, stale, census, iters, pair);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h_stale; unsigned h_census[8];
    (void)hipMemcpy(&h_stale, stale, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h_census, census, 32, hipMemcpyDeviceToHost);
    const double bytes = (double)cus * kSlotWords * 8.0 * iters;
    printf("%s: %d workgroups x %d iterations of 64 KB written + 64 KB read back: %.3f ms, %.0f GB/s written + %.0f GB/s read, stale words %llu of %.3g; "
           "workgroups per XCD %u %u %u %u %u %u %u %u\n", pair ? "partner's slot (same XCD)" : "own slot", cus, iters, ms, bytes / (ms * 1e-3) / 1e9,
           bytes / (ms * 1e-3) / 1e9, h_stale, (double)cus * kSlotWords * iters, h_census[0], h_census[1], h_census[2], h_census[3], h_census[4], h_census[5],
           h_census[6], h_census[7]);
  }
  return 0;
}
