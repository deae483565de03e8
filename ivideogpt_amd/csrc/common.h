// Shared device/host definitions for the iVideoGPT MI355X engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

namespace ivg {

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum DType : int { F32 = 0, BF16 = 1 };

typedef uint32_t Chunk16 __attribute__((ext_vector_type(4)));  // one 16-byte LDS/global chunk (register-resident)

template <typename T> struct Traits;
template <> struct Traits<float> {
  static constexpr int VEC = 4;  // elements per 16-byte chunk
  static constexpr DType dtype = F32;
};
template <> struct Traits<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr DType dtype = BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE (v_cvt_pk_bf16_f32)

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// SiLU whose result is stored as T: for bf16 the IEEE division sequence (~10 vector instructions) is replaced by v_rcp_f32
// (1 ulp, far below the bf16 rounding that follows); fp32 keeps the exact quotient (the tokenizer's bit-exact ids depend on it)
template <typename T> __device__ __forceinline__ float silu_t(float v) {
  if constexpr (sizeof(T) == 2) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  else return silu_f(v);
}

// wave64 reductions (gfx950 wavefront = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), every lane receiving the total: DPP lane permutes run on the
// VALU -- no LDS round trip as __shfl_xor (ds_bpermute) costs.
template <int CTRL> __device__ __forceinline__ float dpp_add_f32(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float d) {
  d = dpp_add_f32<0xB1>(d);    // quad_perm [1,0,3,2]: lane ^ 1
  d = dpp_add_f32<0x4E>(d);    // quad_perm [2,3,0,1]: lane ^ 2
  d = dpp_add_f32<0x141>(d);   // row_half_mirror: the other quad of the 8-lane half
  d = dpp_add_f32<0x140>(d);   // row_mirror: the other half of the row
  return d;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// One-time per-DEVICE hipFuncSetAttribute(MaxDynamicSharedMemorySize) of a kernel (the attribute is per device; a process may
// hold engines on several GPUs and drive them from several host threads: bench.py --lanes, replica()).  The device's bit is
// published only AFTER the attribute call has returned, and late arrivals wait on the mutex -- a second thread can never launch
// a > 64 KiB dynamic-LDS kernel between another thread's "first time" test and its attribute call.
struct DynLdsOnce { std::atomic<unsigned long long> done{0}; std::mutex mu; };
static inline hipError_t ensure_dyn_lds(DynLdsOnce& g, const void* fn, int bytes) {
  int d = 0;
  (void)hipGetDevice(&d);
  const unsigned long long bit = 1ull << (d & 63);
  if (g.done.load(std::memory_order_acquire) & bit) return hipSuccess;
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.done.load(std::memory_order_relaxed) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) g.done.fetch_or(bit, std::memory_order_release);
  return e;
}

}  // namespace ivg
