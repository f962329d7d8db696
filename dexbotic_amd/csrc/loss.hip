// LM-head loss and greedy decode primitives for gfx950 (see include/dexbotic_amd.h).
//   dxa_cross_entropy_fwd/bwd : HF ForCausalLMLoss (transformers/loss/loss_utils.py) as called from
//                               dexbotic/model/dexbotic_arch.py:488 — logits upcast to fp32, mean over the
//                               non-ignored (already shifted) labels.
//   dxa_argmax_rows           : torch.argmax over the vocabulary (first index among equal maxima), the greedy
//                               choice of GenerationMixin.generate(do_sample=False) (discrete_vla_arch.py:33-41).
// All three are one 256-thread workgroup per row streaming the row once (HBM-bound: 152 k logits = 304 KB bf16):
// online (max, sum-exp) pairs per thread folded across the block — no second pass for the maximum.
#include "common.h"

namespace {

template <typename T, int VEC>
__global__ __launch_bounds__(256) void ce_fwd_k(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                float* __restrict__ row_loss, float* __restrict__ lse_out, int64_t V,
                                                int64_t ignore_index) {
  __shared__ float red_m[4], red_s[4];
  const int64_t r = blockIdx.x;
  const T* x = logits + r * ld;
  float m = -INFINITY, s = 0.f;
  for (int64_t i = (int64_t)threadIdx.x * VEC; i < V; i += 256 * VEC) {
    float v[VEC];
    Vec<T, VEC>::ld(v, x + i);
    float vm = v[0];
#pragma unroll
    for (int e = 1; e < VEC; ++e) vm = fmaxf(vm, v[e]);
    const float mn = fmaxf(m, vm);
    float acc = s * expf(m - mn);          // m = -inf on the first visit: s = 0, exp(-inf) = 0
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += expf(v[e] - mn);
    m = mn; s = acc;
  }
  // fold (m, s) pairs: wave shuffle, then the 4 waves through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (mn == -INFINITY) ? 0.f : s * expf(m - mn) + s2 * expf(m2 - mn);
    m = mn;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red_m[w] = m; red_s[w] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = red_m[0], S = red_s[0];
    for (int i = 1; i < 4; ++i) {
      const float mn = fmaxf(M, red_m[i]);
      S = (mn == -INFINITY) ? 0.f : S * expf(M - mn) + red_s[i] * expf(red_m[i] - mn);
      M = mn;
    }
    const float lse = M + logf(S);
    lse_out[r] = lse;
    const int64_t lab = labels[r];
    row_loss[r] = (lab == ignore_index || lab < 0 || lab >= V) ? 0.f : lse - ldf<T>(x + lab);
  }
}

// dlogits = (softmax(x) - onehot(label)) * g ; g = gscale[0] * scale ; ignored rows -> 0.  May run in place.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void ce_bwd_k(const T* logits, int64_t ld, const int64_t* __restrict__ labels,
                                                const float* __restrict__ lse, const float* __restrict__ gscale, float scale,
                                                T* dlogits, int64_t ldd, int64_t V, int64_t ignore_index) {
  const int64_t r = blockIdx.x;
  const T* x = logits + r * ld;
  T* d = dlogits + r * ldd;
  const int64_t lab = labels[r];
  const bool ign = (lab == ignore_index || lab < 0 || lab >= V);
  const float g = ign ? 0.f : (gscale ? gscale[0] : 1.f) * scale;
  const float l = lse[r];
  for (int64_t i = (int64_t)threadIdx.x * VEC; i < V; i += 256 * VEC) {
    float v[VEC];
    Vec<T, VEC>::ld(v, x + i);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float p = ign ? 0.f : expf(v[e] - l);
      v[e] = (p - ((i + e) == lab ? 1.f : 0.f)) * g;
    }
    Vec<T, VEC>::st(d + i, v);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void argmax_rows_k(const T* __restrict__ xin, int64_t ld, int64_t* __restrict__ out,
                                                     int64_t cols) {
  __shared__ float red_v[4];
  __shared__ int64_t red_i[4];
  const T* x = xin + (int64_t)blockIdx.x * ld;
  float best = -INFINITY;
  int64_t idx = INT64_MAX;
  for (int64_t i = (int64_t)threadIdx.x * VEC; i < cols; i += 256 * VEC) {
    float v[VEC];
    Vec<T, VEC>::ld(v, x + i);
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      if (v[e] > best || (v[e] == best && i + e < idx)) { best = v[e]; idx = i + e; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(best, o, 64);
    const int64_t i2 = __shfl_xor(idx, o, 64);
    if (v2 > best || (v2 == best && i2 < idx)) { best = v2; idx = i2; }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red_v[w] = best; red_i[w] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i)
      if (red_v[i] > best || (red_v[i] == best && red_i[i] < idx)) { best = red_v[i]; idx = red_i[i]; }
    out[blockIdx.x] = idx == INT64_MAX ? 0 : idx;       // all -inf / NaN row: index 0 like torch
  }
}

// A handful of rows over a vocabulary (the greedy decode step: ONE row of 152,064 logits): argmax_rows_k walks the row with 256
// threads, 148 dependent iterations of 64-bit compares — 67 us of the 4.5 ms token (profiles/r05_decode_per_token_kernel_stats.txt).
// 1024 threads, 16-byte loads, two loads in flight, 32-bit indices (cols < 2^31), the same (value, then LOWEST index) order.
__global__ __launch_bounds__(1024) void argmax_rows_wide_k(const bf16_t* __restrict__ xin, int64_t ld, int64_t* __restrict__ out,
                                                           int cols) {
  __shared__ float red_v[16];
  __shared__ int red_i[16];
  const bf16_t* x = xin + (int64_t)blockIdx.x * ld;
  float best = -INFINITY;
  int idx = INT_MAX;
  auto take = [&](const float (&v)[8], int i) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (v[e] > best || (v[e] == best && i + e < idx)) { best = v[e]; idx = i + e; }
  };
  int i = (int)threadIdx.x * 8;
  for (; i + 1024 * 8 < cols; i += 2 * 1024 * 8) {             // cols % 8 == 0: a thread's 8 columns are all inside or all outside
    float v0[8], v1[8];
    Vec<bf16_t, 8>::ld(v0, x + i);
    Vec<bf16_t, 8>::ld(v1, x + i + 1024 * 8);
    take(v0, i);
    take(v1, i + 1024 * 8);
  }
  if (i < cols) {
    float v0[8];
    Vec<bf16_t, 8>::ld(v0, x + i);
    take(v0, i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(best, o, 64);
    const int i2 = __shfl_xor(idx, o, 64);
    if (v2 > best || (v2 == best && i2 < idx)) { best = v2; idx = i2; }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red_v[w] = best; red_i[w] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k)
      if (red_v[k] > best || (red_v[k] == best && red_i[k] < idx)) { best = red_v[k]; idx = red_i[k]; }
    out[blockIdx.x] = idx == INT_MAX ? 0 : idx;         // all -inf / NaN row: index 0 like torch
  }
}

inline bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int dxa_cross_entropy_fwd(const void* logits, int64_t ld, const int64_t* labels, float* row_loss, float* lse,
                                     int64_t rows, int64_t V, int64_t ignore_index, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(logits && labels && row_loss && lse && rows >= 0 && V > 0 && ld >= V &&
                (dtype == DXA_F32 || dtype == DXA_BF16), "dxa_cross_entropy_fwd: bad args");
  if (rows == 0) return DXA_OK;
  const size_t es = dtype == DXA_BF16 ? 2 : 4;
  const bool vec = V % 4 == 0 && ld % 4 == 0 && al(logits, 4 * es);
  dim3 grid((unsigned)rows);
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((ce_fwd_k<bf16_t, 4>), grid, dim3(256), 0, ST, (const bf16_t*)logits, ld, labels, row_loss, lse, V, ignore_index);
    else hipLaunchKernelGGL((ce_fwd_k<bf16_t, 1>), grid, dim3(256), 0, ST, (const bf16_t*)logits, ld, labels, row_loss, lse, V, ignore_index);
  } else {
    if (vec) hipLaunchKernelGGL((ce_fwd_k<float, 4>), grid, dim3(256), 0, ST, (const float*)logits, ld, labels, row_loss, lse, V, ignore_index);
    else hipLaunchKernelGGL((ce_fwd_k<float, 1>), grid, dim3(256), 0, ST, (const float*)logits, ld, labels, row_loss, lse, V, ignore_index);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_cross_entropy_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* lse,
                                     const float* gscale, float scale, void* dlogits, int64_t ldd, int64_t rows, int64_t V,
                                     int64_t ignore_index, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(logits && labels && lse && dlogits && rows >= 0 && V > 0 && ld >= V && ldd >= V &&
                (dtype == DXA_F32 || dtype == DXA_BF16), "dxa_cross_entropy_bwd: bad args");
  if (rows == 0) return DXA_OK;
  const size_t es = dtype == DXA_BF16 ? 2 : 4;
  const bool vec = V % 4 == 0 && ld % 4 == 0 && ldd % 4 == 0 && al(logits, 4 * es) && al(dlogits, 4 * es);
  dim3 grid((unsigned)rows);
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((ce_bwd_k<bf16_t, 4>), grid, dim3(256), 0, ST, (const bf16_t*)logits, ld, labels, lse, gscale, scale, (bf16_t*)dlogits, ldd, V, ignore_index);
    else hipLaunchKernelGGL((ce_bwd_k<bf16_t, 1>), grid, dim3(256), 0, ST, (const bf16_t*)logits, ld, labels, lse, gscale, scale, (bf16_t*)dlogits, ldd, V, ignore_index);
  } else {
    if (vec) hipLaunchKernelGGL((ce_bwd_k<float, 4>), grid, dim3(256), 0, ST, (const float*)logits, ld, labels, lse, gscale, scale, (float*)dlogits, ldd, V, ignore_index);
    else hipLaunchKernelGGL((ce_bwd_k<float, 1>), grid, dim3(256), 0, ST, (const float*)logits, ld, labels, lse, gscale, scale, (float*)dlogits, ldd, V, ignore_index);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_argmax_rows(const void* x, int64_t ld, int64_t* out, int64_t rows, int64_t cols, int dtype,
                               dxa_stream_t stream) {
  DXA_CHECK_ARG(x && out && rows >= 0 && cols > 0 && ld >= cols && (dtype == DXA_F32 || dtype == DXA_BF16),
                "dxa_argmax_rows: bad args");
  if (rows == 0) return DXA_OK;
  const size_t es = dtype == DXA_BF16 ? 2 : 4;
  const bool vec = cols % 4 == 0 && ld % 4 == 0 && al(x, 4 * es);
  dim3 grid((unsigned)rows);
  static const bool wide_off = getenv("DXA_ARGMAX_NO_WIDE") != nullptr;
  if (!wide_off && dtype == DXA_BF16 && rows <= 64 && cols >= 16384 && cols < (1ll << 31) && cols % 8 == 0 && ld % 8 == 0 && al(x, 16)) {
    hipLaunchKernelGGL(argmax_rows_wide_k, grid, dim3(1024), 0, ST, (const bf16_t*)x, ld, out, (int)cols);
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((argmax_rows_k<bf16_t, 4>), grid, dim3(256), 0, ST, (const bf16_t*)x, ld, out, cols);
    else hipLaunchKernelGGL((argmax_rows_k<bf16_t, 1>), grid, dim3(256), 0, ST, (const bf16_t*)x, ld, out, cols);
  } else {
    if (vec) hipLaunchKernelGGL((argmax_rows_k<float, 4>), grid, dim3(256), 0, ST, (const float*)x, ld, out, cols);
    else hipLaunchKernelGGL((argmax_rows_k<float, 1>), grid, dim3(256), 0, ST, (const float*)x, ld, out, cols);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
