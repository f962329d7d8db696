// Shared device/host helpers for libdexbotic_amd (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dexbotic_amd.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define DXA_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (same as torch's float->bfloat16)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair with ONE v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2_t_ __attribute__((ext_vector_type(2)));
  typedef float f32x2_t_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t_){a, b}, bf16x2_t_));
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// value rounded through the storage type (what a torch op in that dtype would hand to the next op)
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }

// 4-wide (8 B bf16 / 16 B fp32) and scalar accessors used by the HBM-bound kernels
template <typename T, int VEC> struct Vec;
template <> struct Vec<float, 4> {
  static __device__ __forceinline__ void ld(float (&o)[4], const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec<bf16_t, 4> {
  static __device__ __forceinline__ void ld(float (&o)[4], const bf16_t* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[4]) {
    uint2 o;
    o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = o;
  }
};
template <> struct Vec<bf16_t, 8> {          // 16 bytes per lane: the widest access, for the HBM-bound row kernels
  static __device__ __forceinline__ void ld(float (&o)[8], const bf16_t* p) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
};
template <typename T> struct Vec<T, 1> {
  static __device__ __forceinline__ void ld(float (&o)[1], const T* p) { o[0] = ldf<T>(p); }
  static __device__ __forceinline__ void st(T* p, const float (&v)[1]) { stf<T>(p, v[0]); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blocks of up to 1024 threads; `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// tanh through v_exp_f32 / v_rcp_f32: 1 - 2 / (1 + e^{2u}); exact limits at +-inf, absolute error ~2e-7 (it is only ever added to 1 or
// squared below).  tanhf is ~20 VALU instructions; the GeGLU backward of pi0's 16384-wide MLP evaluates it 428 M times per layer and
// was VALU-co-bound with it (237 us per launch against 150 for the same bytes in swiglu_bwd_k).
__device__ __forceinline__ float fast_tanhf(float u) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * u));
}

// logistic function through v_exp_f32 / v_rcp_f32 (1 ulp each; exact limits).  Used ONLY where an epilogue's VALU time is exposed (the
// serving-side gate / up epilogue, the decode step).  Tried in swiglu_fwd_k / swiglu_bwd_k / SiLU / quick-GELU as well (round 6): the
// step did not move (232.8 vs 232.9 ms) and three real-width bf16 tests left their bounds (1.5 x the observed distance to the reference
// under autocast: one bf16 step in a few pre-activations per million is enough) — reverted; GELU-tanh keeps fast_tanhf (pi0 + 3 %, all
// bounds held).
__device__ __forceinline__ float fast_sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case DXA_ACT_GELU_ERF: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    case DXA_ACT_GELU_TANH: {
      const float k = 0.79788456080286535588f;  // sqrt(2/pi)
      return 0.5f * x * (1.f + fast_tanhf(k * (x + 0.044715f * x * x * x)));
    }
    case DXA_ACT_QUICK_GELU: return x / (1.f + expf(-1.702f * x));
    case DXA_ACT_SILU: return x / (1.f + expf(-x));
    case DXA_ACT_RELU: return x > 0.f ? x : 0.f;
    case DXA_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    default: return x;
  }
}
__device__ __forceinline__ float act_grad(int act, float x) {
  switch (act) {
    case DXA_ACT_GELU_ERF: {
      const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case DXA_ACT_GELU_TANH: {
      const float k = 0.79788456080286535588f;
      const float u = k * (x + 0.044715f * x * x * x);
      const float t = fast_tanhf(u);
      const float du = k * (1.f + 3.f * 0.044715f * x * x);
      return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
    }
    case DXA_ACT_QUICK_GELU: {
      const float s = 1.f / (1.f + expf(-1.702f * x));
      return s + x * 1.702f * s * (1.f - s);
    }
    case DXA_ACT_SILU: {
      const float s = 1.f / (1.f + expf(-x));
      return s * (1.f + x * (1.f - s));
    }
    case DXA_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case DXA_ACT_SIGMOID: {
      const float s = 1.f / (1.f + expf(-x));
      return s * (1.f - s);
    }
    default: return 1.f;
  }
}

// ---- host side error plumbing ------------------------------------------------------------
void dxa_set_error(const char* fmt, ...);
#define DXA_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      dxa_set_error(__VA_ARGS__);           \
      return DXA_ERR_BAD_ARG;               \
    }                                       \
  } while (0)
#define DXA_CHECK_LAUNCH()                                                 \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      dxa_set_error("%s:%d HIP error: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return DXA_ERR_HIP;                                                  \
    }                                                                      \
  } while (0)

#define DXA_CHECK_HIP(expr)                                                \
  do {                                                                     \
    hipError_t e_ = (expr);                                                \
    if (e_ != hipSuccess) {                                                \
      dxa_set_error("%s:%d HIP error: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return DXA_ERR_HIP;                                                  \
    }                                                                      \
  } while (0)

static inline int dxa_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int dxa_grid1d(int64_t n, int block, int cap = 256 * 16) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
