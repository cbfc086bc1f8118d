// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16x32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/dualdiff_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define DD_WAVE 64

template <typename T> struct dd_vec;
template <> struct dd_vec<_Float16> { using v8 = f16x8; using v4 = f16x4; };
template <> struct dd_vec<__bf16>   { using v8 = bf16x8; using v4 = bf16x4; };

// DD_DBG_NOMFMA / DD_DBG_NODMA (tools/build_dbg_libs.sh): diagnostic builds that drop one side of the main loop
// (matrix instructions, or the LDS-DMA loads) to see which one bounds a kernel.  Never defined in the product build.
__device__ __forceinline__ f32x4 dd_mfma16(f16x8 a, f16x8 b, f32x4 c) {
#ifdef DD_DBG_NOMFMA
  asm volatile("" ::"v"(a), "v"(b));
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 dd_mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 16-byte vector <-> 8 x T
template <typename T>
__device__ __forceinline__ typename dd_vec<T>::v8 dd_as_v8(u32x4 u) {
  typename dd_vec<T>::v8 r;
  __builtin_memcpy(&r, &u, 16);
  return r;
}
template <typename T>
__device__ __forceinline__ u32x4 dd_as_u4(typename dd_vec<T>::v8 v) {
  u32x4 r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}

__device__ __forceinline__ u32x4 dd_ld16(const void* p) {
  return *reinterpret_cast<const u32x4*>(p);
}
__device__ __forceinline__ void dd_st16(void* p, u32x4 v) {
  *reinterpret_cast<u32x4*>(p) = v;
}

template <typename T>
__device__ __forceinline__ void dd_unpack8(u32x4 u, float (&f)[8]) {
  typename dd_vec<T>::v8 v = dd_as_v8<T>(u);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T>
__device__ __forceinline__ u32x4 dd_pack8(const float (&f)[8]) {
  typename dd_vec<T>::v8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (T)f[i];
  return dd_as_u4<T>(v);
}

// x * sigmoid(x) with v_exp_f32 + the 1-ulp v_rcp_f32 (a true division costs a ~10-instruction fix-up chain)
__device__ __forceinline__ float dd_silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below fp16/bf16 resolution): one rcp, one
// exp and five fmas instead of libm's ~40-instruction erff — the GEGLU epilogue evaluates it on
// 21.5 M gate values per L0 feed-forward.
__device__ __forceinline__ float dd_erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // 1-ulp v_rcp_f32: no division fix-up chain
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = 1.0f - p * t * e;
  return copysignf(r, x);
}
__device__ __forceinline__ float dd_gelu_erf_f(float x) {
  return 0.5f * x * (1.0f + dd_erf_fast(x * 0.70710678118654752440f));
}
// h * gelu(g), the GEGLU gate, in 12 full-rate + 2 transcendental instructions (the form above takes 17 + 2):
//   x Phi(x) = max(x, 0) - |x| * (erfc(|x| / sqrt 2) / 2), erfc by the same Abramowitz-Stegun 7.1.26 polynomial with the
// 1/2 folded into its coefficients and the 1/sqrt 2 into the rational argument — same approximation, same 1.5e-7 bound on
// erf, no sign transfer and no 1 + erf.  (Round 6: the epilogue of the 16800 x 2560 x 320 GEGLU projection evaluates
// 21.5 M gates — 11 us of vector issue at 84 cycles per wave-instruction group.)
__device__ __forceinline__ float dd_geglu_f(float h, float g) {
#ifdef DD_DBG_GEGLU_CHEAP                 // diagnostic build (never the product): what the gate's arithmetic costs a launch
  return h * g;
#endif
  const float ax = fabsf(g);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * (g * g));
  const float q = (p * t) * e;                       // erfc(|g| / sqrt 2) / 2
  return h * fmaf(-ax, q, fmaxf(g, 0.0f));
}

__device__ __forceinline__ float dd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float dd_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// hipGetLastError() is sticky per thread: a benign earlier error of ANY runtime call (e.g. torch's
// hipEventQuery -> hipErrorNotReady) would otherwise be reported as our launch failure.
static inline void dd_clear_error() { (void)hipGetLastError(); }
static inline int dd_check_launch() {
  return hipGetLastError() == hipSuccess ? DD_OK : DD_ERR_LAUNCH;
}
static inline bool dd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Kernels that need more than 64 KB of dynamic LDS must raise hipFuncAttributeMaxDynamicSharedMemorySize,
// and that attribute is PER DEVICE: one flag bit per device ordinal (a process-wide bool would leave a
// second GPU of the same process without it), thread-safe.
static inline void dd_ensure_dyn_lds(const void* kern, size_t smem, std::atomic<uint64_t>& done) {
  if (smem <= 65536) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  done.fetch_or(bit, std::memory_order_release);
}
