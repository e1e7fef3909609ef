// Micro-benchmark behind the voxelizer's design: what does ONE random-address device-scope operation per point cost on
// MI355X, by flavour?  N threads (one per "point") hit pseudo-random 4- or 8-byte words of a table of T bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ void k(void* table, size_t words, int n, uint32_t salt, unsigned long long* sink) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t w = mix(i * 2654435761u + salt) % words;
  unsigned long long r = 0;
  if (MODE == 0) atomicOr((unsigned*)table + w, 1u << (i & 31));                                 // no return, 32 bit
  if (MODE == 1) atomicMin((unsigned long long*)table + w, ((unsigned long long)w << 32) | i);  // no return, 64 bit
  if (MODE == 2) r = atomicCAS((unsigned long long*)table + w, ~0ull, ((unsigned long long)w << 32) | i);  // returning
  if (MODE == 3) r = atomicAdd((unsigned*)table + w, 1u);                                        // returning, 32 bit
  if (MODE == 4) ((unsigned*)table)[w] = i;                                                      // plain store
  if (MODE == 5) r = ((unsigned long long*)table)[w];                                            // plain 8-byte load
  if (MODE == 6) atomicMin((unsigned*)table + w, (unsigned)i);                                   // no return, 32 bit min
  if (MODE == 7) {                                                                               // wave-aggregated: only if needed
    const unsigned old = __hip_atomic_load((unsigned*)table + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old > (unsigned)i) atomicMin((unsigned*)table + w, (unsigned)i);
  }
  if ((MODE == 2 || MODE == 3 || MODE == 5) && r == 0x123456789abcull) *sink = r;
}

template <int MODE>
float run(void* table, size_t bytes, int n, int wordsize, unsigned long long* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t words = bytes / wordsize;
  float best = 1e9f;
  for (int rep = 0; rep < 8; ++rep) {
    hipMemsetAsync(table, 0xFF, bytes, 0);
    hipEventRecord(e0, 0);
    k<MODE><<<(n + 255) / 256, 256>>>(table, words, n, 77u + rep, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  const int n = 1190000;
  unsigned long long* sink; hipMalloc(&sink, 8);
  const size_t sizes[3] = {(size_t)1 << 20, (size_t)16 << 20, (size_t)42 << 20};
  const char* names[8] = {"atomicOr u32 (no return)", "atomicMin u64 (no return)", "atomicCAS u64 (returning)",
                          "atomicAdd u32 (returning)", "plain store u32", "plain load u64", "atomicMin u32 (no return)",
                          "peek + atomicMin u32 if smaller"};
  for (size_t bytes : sizes) {
    void* table; hipMalloc(&table, bytes);
    printf("table %zu MB, %d ops (one per thread), best of 8, us (incl. ~4 us event bracket):\n", bytes >> 20, n);
    float t[8];
    t[0] = run<0>(table, bytes, n, 4, sink); t[1] = run<1>(table, bytes, n, 8, sink);
    t[2] = run<2>(table, bytes, n, 8, sink); t[3] = run<3>(table, bytes, n, 4, sink);
    t[4] = run<4>(table, bytes, n, 4, sink); t[5] = run<5>(table, bytes, n, 8, sink);
    t[6] = run<6>(table, bytes, n, 4, sink); t[7] = run<7>(table, bytes, n, 4, sink);
    for (int m = 0; m < 8; ++m) printf("  %-34s %8.1f us  %6.1f G ops/s\n", names[m], t[m], n / t[m] * 1e-3);
    hipFree(table);
  }
  return 0;
}
