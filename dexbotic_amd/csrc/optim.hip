// Fused multi-tensor AdamW over a flat parameter arena + global grad-norm clip helpers (HBM-bound).
// One launch updates every parameter: the arena is described by a device chunk table (start, len,
// group), each workgroup owns one chunk; lr / weight-decay come per group, the clip coefficient from a
// device scalar (no host sync between the norm reduction and the update).  fp32 p/g/m/v with 16-byte
// accesses; an optional bf16 shadow of the new parameters is written in the same pass, so the next
// forward reads bf16 weights without a separate cast sweep.
// Semantics = torch.optim.AdamW (decoupled weight decay) after clip_grad_norm_:
//   g *= clip; p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"

namespace {

struct AdamP {
  float* p; const void* g; float* m; float* v; bf16_t* shadow;
  const int64_t* cs; const int32_t* cl; const int32_t* cg;
  float lr[8], wd[8];
  float b1, b2, eps, bc1, bc2;
  const float* clip;
  uint8_t* state;      // per-chunk sparse-table state (dxa_adamw_desc.chunk_state) or null
  const int64_t* cms;  // per-chunk start of the moments in a packed (sharded) m / v, or null: the arena offset
  int n_chunks;        // chunks of this launch: a workgroup takes chunks blockIdx.x, + gridDim.x, ... (grid < n_chunks = a
                       // bounded-footprint launch that leaves the other CUs to a concurrent stream)
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr, float wd, const AdamP& a,
                                      float step_size, float inv_sqrt_bc2) {
  // same operation order as torch's _single_tensor_adamw (lerp_, mul_+addcmul_, sqrt/ bc2_sqrt + eps, addcdiv_)
  p *= 1.f - lr * wd;
  m = m + (1.f - a.b1) * (g - m);
  v = a.b2 * v + (1.f - a.b2) * (g * g);
  const float denom = sqrtf(v) / inv_sqrt_bc2 + a.eps;
  p -= step_size * (m / denom);
}

// 4 consecutive gradients as fp32: fp32 arena (16-byte load) or the bf16 communication copy of it (8-byte load)
template <typename TG> struct GradLd;
template <> struct GradLd<float> {
  typedef float f4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f4 nt4(const float* g, int64_t i) { return __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i); }
  static __device__ __forceinline__ f4 ld4(const float* g, int64_t i) { return reinterpret_cast<const f4*>(g)[i]; }
};
template <> struct GradLd<bf16_t> {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ f4 cvt(u2 v) {
    return (f4){__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                __uint_as_float(v.y & 0xffff0000u)};
  }
  static __device__ __forceinline__ f4 nt4(const bf16_t* g, int64_t i) { return cvt(__builtin_nontemporal_load(reinterpret_cast<const u2*>(g) + i)); }
  static __device__ __forceinline__ f4 ld4(const bf16_t* g, int64_t i) { return cvt(reinterpret_cast<const u2*>(g)[i]); }
};

template <typename TG>
__device__ __forceinline__ void adamw_chunk(const AdamP& a, const int c);

template <typename TG>
__global__ __launch_bounds__(256) void adamw_k(const AdamP a) {
  for (int c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
    adamw_chunk<TG>(a, c);
    if (gridDim.x < (unsigned)a.n_chunks) __syncthreads();     // (the sparse-table vote re-uses its LDS word)
  }
}

template <typename TG>
__device__ __forceinline__ void adamw_chunk(const AdamP& a, const int c) {
  const int64_t start = a.cs[c];
  const int len = a.cl[c];
  const int grp = a.cg[c];
  const float lr = a.lr[grp], wd = a.wd[grp];
  const float clip = a.clip ? a.clip[0] : 1.f;
  const float step_size = lr / a.bc1;
  const float inv_sqrt_bc2 = sqrtf(a.bc2);  // (name kept: it is the divisor sqrt(1-beta2^t))
  float* p = a.p + start;
  const TG* g = reinterpret_cast<const TG*>(a.g) + start;
  const int64_t mstart = a.cms ? a.cms[c] : start;
  float* m = a.m + mstart;
  float* v = a.v + mstart;
  bf16_t* sh = a.shadow ? a.shadow + start : nullptr;
  const bool vec = (((start | mstart) & 3) == 0);
  const int n4 = vec ? (len >> 2) : 0;
  if (a.state != nullptr && a.state[c] == 1 && wd == 0.f) {
    // chunk of a sparsely touched table that never had a gradient (m = v = 0): an all-zero gradient leaves p, m, v unchanged
    // (m' = 0, v' = 0, p' = p - step * 0 / (0 + eps)); one read of g decides, NaN / Inf count as non-zero
    typedef float f4z __attribute__((ext_vector_type(4)));
    int nz = 0;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const f4z gz = GradLd<TG>::ld4(g, i);
      nz |= !(gz.x == 0.f && gz.y == 0.f && gz.z == 0.f && gz.w == 0.f);
    }
    for (int i = (n4 << 2) + threadIdx.x; i < len; i += 256) nz |= !(ldf<TG>(g + i) == 0.f);
    if (!__syncthreads_or(nz)) return;
    if (threadIdx.x == 0) a.state[c] = 2;
  }
  // two float4 groups per thread in flight (8 independent 16-B loads); g, m, v and the stores stream through HBM
  // exactly once per step, so they carry the nontemporal hint and leave L2/MALL to the master weights' neighbours
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  auto upd = [&](f4& pv, const f4& gv, f4& mv, f4& vv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pv[e], me = mv[e], ve = vv[e];
      adam1(pe, gv[e] * clip, me, ve, lr, wd, a, step_size, inv_sqrt_bc2);
      pv[e] = pe; mv[e] = me; vv[e] = ve;
    }
  };
  auto put = [&](int i, const f4& pv, const f4& mv, const f4& vv) {
    __builtin_nontemporal_store(pv, reinterpret_cast<f4*>(p) + i);
    __builtin_nontemporal_store(mv, reinterpret_cast<f4*>(m) + i);
    __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(v) + i);
    if (sh) {
      u2 o;
      o.x = (uint32_t)f2bf(pv.x) | ((uint32_t)f2bf(pv.y) << 16);
      o.y = (uint32_t)f2bf(pv.z) | ((uint32_t)f2bf(pv.w) << 16);
      __builtin_nontemporal_store(o, reinterpret_cast<u2*>(sh) + i);
    }
  };
  int i = threadIdx.x;
  for (; i + 256 < n4; i += 512) {
    f4 p0 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p) + i);
    f4 p1 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p) + i + 256);
    const f4 g0 = GradLd<TG>::nt4(g, i);
    const f4 g1 = GradLd<TG>::nt4(g, i + 256);
    f4 m0 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i);
    f4 m1 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i + 256);
    f4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i);
    f4 v1 = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i + 256);
    upd(p0, g0, m0, v0);
    upd(p1, g1, m1, v1);
    put(i, p0, m0, v0);
    put(i + 256, p1, m1, v1);
  }
  for (; i < n4; i += 256) {
    f4 p0 = reinterpret_cast<const f4*>(p)[i];
    const f4 g0 = GradLd<TG>::ld4(g, i);
    f4 m0 = reinterpret_cast<const f4*>(m)[i];
    f4 v0 = reinterpret_cast<const f4*>(v)[i];
    upd(p0, g0, m0, v0);
    put(i, p0, m0, v0);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < len; i += 256) {
    float pv = p[i], mv = m[i], vv = v[i];
    adam1(pv, ldf<TG>(g + i) * clip, mv, vv, lr, wd, a, step_size, inv_sqrt_bc2);
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (sh) sh[i] = f2bf(pv);
  }
}

// stage 1: up to 4096 blocks, each writes one double partial; stage 2: one block folds them
template <typename TG>
__global__ __launch_bounds__(256) void sumsq_stage1_k(const TG* __restrict__ x, int64_t n, double* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  const bool vec = (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(TG) - 1)) == 0;
  if (vec) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const TG* x4 = x;   // indexed in groups of 4 elements through GradLd
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {     // four independent 16-B loads in flight per thread
      const f4 a = GradLd<TG>::nt4(x4, i), b = GradLd<TG>::nt4(x4, i + stride);
      const f4 c = GradLd<TG>::nt4(x4, i + 2 * stride), d = GradLd<TG>::nt4(x4, i + 3 * stride);
      s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
      s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
      s2 += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
      s3 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    }
    for (; i < n4; i += stride) {
      const f4 a = GradLd<TG>::ld4(x4, i);
      s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    s += s1 + s2 + s3;
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float t = ldf<TG>(x + i); s += t * t; }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float t = ldf<TG>(x + i); s += t * t; }
  }
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = (double)t;
}
__global__ __launch_bounds__(256) void sumsq_stage2_k(const double* __restrict__ part, int nparts, float* __restrict__ out,
                                                      int accumulate) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + (float)red[0] : (float)red[0];
}
// sum(x^2) over up to 4096 short ranges [start, start + len) of one arena: one workgroup per range -> one double partial
// each, stage 2 folds them.  The slots of a gradient bucket that no GEMM epilogue accounts for (norm weights, biases,
// position embeddings ...) are a handful of kilobyte-sized slices: one launch per bucket instead of one per slice.
template <typename TG>
__global__ __launch_bounds__(256) void sumsq_ranges_k(const TG* __restrict__ base, const int64_t* __restrict__ starts,
                                                      const int64_t* __restrict__ lens, double* __restrict__ part) {
  __shared__ float red[16];
  const TG* x = base + starts[blockIdx.x];
  const int64_t n = lens[blockIdx.x];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) { const float t = ldf<TG>(x + i); s += t * t; }
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = (double)t;
}
// out (+)= sum of n floats, one workgroup, double accumulation in a fixed order (the per-tile partials of dxa_gemm's sumsq)
__global__ __launch_bounds__(1024) void sum_f32_k(const float* __restrict__ x, int64_t n, float* __restrict__ out, int accumulate) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + (float)red[0] : (float)red[0];
}
__global__ void clip_coef_k(const float* __restrict__ sumsq, float max_norm, float grad_scale, float* __restrict__ norm_out,
                            float* __restrict__ coef_out) {
  const float norm = sqrtf(sumsq[0]) * grad_scale;
  if (norm_out) norm_out[0] = norm;
  if (coef_out) {
    const float c = max_norm / (norm + 1e-6f);
    coef_out[0] = (c < 1.f ? c : 1.f) * grad_scale;
  }
}
__global__ __launch_bounds__(256) void scale_k(float* __restrict__ x, int64_t n, float s) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= s;
}

__global__ __launch_bounds__(256) void scale_dev_k(float* __restrict__ x, int64_t n, const float* __restrict__ s) {
  const float f = s[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= f;
}

}  // namespace

extern "C" int dxa_adamw(const dxa_adamw_desc* d, dxa_stream_t stream) {
  DXA_CHECK_ARG(d && d->p && d->g && d->m && d->v && d->chunk_start && d->chunk_len && d->chunk_grp,
                "dxa_adamw: null pointer");
  DXA_CHECK_ARG(d->n_chunks >= 0, "dxa_adamw: negative chunk count");
  if (d->n_chunks == 0) return DXA_OK;
  AdamP a;
  DXA_CHECK_ARG(d->g_dtype == DXA_F32 || d->g_dtype == DXA_BF16, "dxa_adamw: bad gradient dtype %d", d->g_dtype);
  a.p = d->p; a.g = d->g; a.m = d->m; a.v = d->v; a.shadow = (bf16_t*)d->shadow;
  a.cs = d->chunk_start; a.cl = d->chunk_len; a.cg = d->chunk_grp;
  for (int i = 0; i < 8; ++i) { a.lr[i] = d->lr[i]; a.wd[i] = d->wd[i]; }
  a.b1 = d->beta1; a.b2 = d->beta2; a.eps = d->eps; a.bc1 = d->bc1; a.bc2 = d->bc2;
  a.clip = d->clip_coef;
  a.state = d->chunk_state;
  a.cms = d->chunk_mv_start;
  a.n_chunks = (int)d->n_chunks;
  // DXA_ADAMW_GRID=<n>: at most n workgroups, each walking chunks n apart (the overlapped update's footprint on the CUs)
  static const int grid_cap = getenv("DXA_ADAMW_GRID") ? atoi(getenv("DXA_ADAMW_GRID")) : 0;
  const unsigned grid = grid_cap > 0 && (int64_t)grid_cap < d->n_chunks ? (unsigned)grid_cap : (unsigned)d->n_chunks;
  if (d->g_dtype == DXA_BF16) hipLaunchKernelGGL(adamw_k<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(adamw_k<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_sumsq(const void* x, int64_t n, int dtype, double* scratch, float* out, int accumulate, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && scratch && out && n >= 0 && (dtype == DXA_F32 || dtype == DXA_BF16), "dxa_sumsq: bad args");
  int nb = dxa_grid1d((n + 3) / 4, 256, 4096);
  if (dtype == DXA_BF16) hipLaunchKernelGGL(sumsq_stage1_k<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n, scratch);
  else hipLaunchKernelGGL(sumsq_stage1_k<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, n, scratch);
  hipLaunchKernelGGL(sumsq_stage2_k, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, nb, out, accumulate);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_sumsq_ranges(const void* base, int dtype, const int64_t* starts, const int64_t* lens, int n_ranges,
                                double* scratch, float* out, int accumulate, dxa_stream_t stream) {
  DXA_CHECK_ARG(base && starts && lens && scratch && out && (dtype == DXA_F32 || dtype == DXA_BF16), "dxa_sumsq_ranges: bad args");
  DXA_CHECK_ARG(n_ranges >= 0 && n_ranges <= 4096, "dxa_sumsq_ranges: 0..4096 ranges (got %d)", n_ranges);
  if (n_ranges == 0) return DXA_OK;
  if (dtype == DXA_BF16) hipLaunchKernelGGL(sumsq_ranges_k<bf16_t>, dim3(n_ranges), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)base, starts, lens, scratch);
  else hipLaunchKernelGGL(sumsq_ranges_k<float>, dim3(n_ranges), dim3(256), 0, (hipStream_t)stream, (const float*)base, starts, lens, scratch);
  hipLaunchKernelGGL(sumsq_stage2_k, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, n_ranges, out, accumulate);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_sum_f32(const float* x, int64_t n, float* out, int accumulate, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && out && n >= 0, "dxa_sum_f32: bad args");
  hipLaunchKernelGGL(sum_f32_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out, accumulate);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_clip_coef(const float* sumsq, float max_norm, float* norm_out, float* coef_out, dxa_stream_t stream) {
  DXA_CHECK_ARG(sumsq, "dxa_clip_coef: null");
  hipLaunchKernelGGL(clip_coef_k, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, 1.f, norm_out, coef_out);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_clip_coef_scaled(const float* sumsq, float max_norm, float grad_scale, float* norm_out, float* coef_out,
                                    dxa_stream_t stream) {
  DXA_CHECK_ARG(sumsq && grad_scale > 0.f, "dxa_clip_coef_scaled: null sum / non-positive scale");
  hipLaunchKernelGGL(clip_coef_k, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, grad_scale, norm_out, coef_out);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_scale(float* x, int64_t n, float s, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && n >= 0, "dxa_scale: bad args");
  if (n == 0) return DXA_OK;
  hipLaunchKernelGGL(scale_k, dim3(dxa_grid1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, s);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_scale_dev(float* x, int64_t n, const float* s, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && s && n >= 0, "dxa_scale_dev: bad args");
  if (n == 0) return DXA_OK;
  hipLaunchKernelGGL(scale_dev_k, dim3(dxa_grid1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, s);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
