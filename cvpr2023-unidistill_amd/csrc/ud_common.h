// Shared device/host helpers for libunidistill_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "unidistill_hip.h"

#define UD_WAVE 64

#define UD_LAUNCH_CHECK()                                  \
  do {                                                     \
    if (hipGetLastError() != hipSuccess) return UD_ERR_HIP; \
  } while (0)

#define UD_HIP_TRY(expr)                          \
  do {                                            \
    if ((expr) != hipSuccess) return UD_ERR_HIP;  \
  } while (0)

static inline size_t ud_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Bump allocator over the caller-provided workspace.
struct UdArena {
  char* base;
  size_t cap;
  size_t used;
  UdArena(void* p, size_t n) : base((char*)p), cap(n), used(0) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = ud_align_up(count * sizeof(T));
    T* r = (T*)(base + used);
    used += bytes;
    return r;
  }
  bool ok() const { return base != nullptr && used <= cap; }
};

static inline int ud_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// Zero n floats with a KERNEL instead of hipMemsetAsync.  Inside a captured hipGraph (ops/graphed.py) the memset NODES that cleared the
// BatchNorm partial rows of the Winograd launchers were not ordered like their stream counterparts on ROCm 7.2: the replayed fp32
// step diverged from the eager one from the fifth replay on, and stayed bit-identical with either launcher's memset gone
// (UD_WINO_NO_SPLIT=1 / UD_WINO4_SK=0) -- a kernel node keeps the stream order.
static __global__ void k_ud_zero_f32(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
static inline void ud_zero_f32_async(float* p, size_t n, hipStream_t stream) {
  if (!n) return;
  const size_t blocks = (n + 255) / 256;
  k_ud_zero_f32<<<(unsigned)(blocks < 1024 ? blocks : 1024), 256, 0, stream>>>(p, n);
}
#endif


// Guard for "set once" HIP state that is really PER DEVICE (hipFuncSetAttribute: a process that drives several GPUs,
// or several host threads, must not skip it on a device that has not seen it): a bit per device ordinal; racing
// threads may both run the guarded block (idempotent calls), none can skip it.
//   static UdDeviceOnce once;  if (const unsigned long long bit = once.pending()) { ...; once.mark(bit); }
struct UdDeviceOnce {
  std::atomic<unsigned long long> done{0};
  unsigned long long pending() {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long b = 1ull << (d & 63);
    return (done.load(std::memory_order_acquire) & b) ? 0ull : b;
  }
  void mark(unsigned long long b) { done.fetch_or(b, std::memory_order_release); }
};

#ifdef __HIPCC__
__device__ __forceinline__ int ud_lane() { return threadIdx.x & (UD_WAVE - 1); }

// Sum over the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float ud_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Streaming (read-once / write-once) accesses bypass cache allocation: on MI355X a 1 KiB-row
// gather of a 484 MB tensor runs 131 us -> 92 us with nontemporal loads (tools/exp_pool.py).
typedef float ud_vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ud_ldg_stream(const float* p) {
  const ud_vf4 t = __builtin_nontemporal_load(reinterpret_cast<const ud_vf4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void ud_stg_stream(float* p, const float4& v) {
  ud_vf4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<ud_vf4*>(p));
}

// Two floats -> packed bf16 pair (round to nearest even) in ONE v_cvt_pk_bf16_f32; a in the low half.
typedef float ud_vf2 __attribute__((ext_vector_type(2)));
typedef __bf16 ud_vbf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ud_pack_bf16x2(float a, float b) {
  const ud_vf2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ud_vbf2));
}

__device__ __forceinline__ int ud_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#endif
