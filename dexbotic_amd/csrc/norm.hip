// RMSNorm / LayerNorm forward+backward and column sums (HBM-bound; wave64 shuffle reductions).
// One wavefront owns one row in the forward (no LDS, no barriers): 16-byte loads, fp32 statistics,
// row re-read from L1/L2 for the normalise pass.  The backward keeps one row per wavefront too and
// accumulates the weight/bias gradient partials of a workgroup's rows in a per-workgroup fp32 slab
// (L2 resident), reduced afterwards by dxa_colsum: deterministic, no atomics.
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------- RMSNorm
template <typename T, typename TW, int VEC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_k(const T* __restrict__ x, const TW* __restrict__ w,
                                                     T* __restrict__ y, float* __restrict__ rstd_out,
                                                     int64_t rows, int64_t cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * cols;
  float ss = 0.f;
  for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
    float v[VEC];
    Vec<T, VEC>::ld(v, xr + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) ss += v[i] * v[i];
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)cols + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  T* yr = y + row * cols;
  for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
    float v[VEC], g[VEC];
    Vec<T, VEC>::ld(v, xr + c);
    if (w) {
      Vec<TW, VEC>::ld(g, w + c);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] = 1.f;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = g[i] * rnd<T>(v[i] * rstd);
    Vec<T, VEC>::st(yr + c, v);
  }
}

// partial_dw: [gridDim.x][cols]; every block zeroes its own slab first
template <typename T, typename TW, int VEC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_k(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const TW* __restrict__ w, const float* __restrict__ rstd,
                                                     T* __restrict__ dx, const T* __restrict__ res,
                                                     float* __restrict__ partial, int64_t rows, int64_t cols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* slab = w ? partial + (int64_t)blockIdx.x * cols : nullptr;
  if (w) {
    for (int64_t c = threadIdx.x; c < cols; c += 256) slab[c] = 0.f;
    __syncthreads();
  }
  // rows are dealt round-robin: block b takes rows b*4+wave + k*gridDim.x*4.  The four waves of a block
  // add into the same slab columns -> serialise them wave by wave (deterministic order).
  for (int64_t base = (int64_t)blockIdx.x * 4; base < rows; base += (int64_t)gridDim.x * 4) {
    const int64_t row = base + wave;
    const bool ok = row < rows;
    float rs = 0.f, cterm = 0.f;
    if (ok) {
      rs = rstd[row];
      const T* xr = x + row * cols;
      const T* gr = dy + row * cols;
      float s = 0.f;
      for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
        float xv[VEC], gv[VEC], wv[VEC];
        Vec<T, VEC>::ld(xv, xr + c);
        Vec<T, VEC>::ld(gv, gr + c);
        if (w) Vec<TW, VEC>::ld(wv, w + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s += gv[i] * (w ? wv[i] : 1.f) * (xv[i] * rs);
      }
      cterm = wave_sum(s) / (float)cols;
      T* dxr = dx + row * cols;
      for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
        float xv[VEC], gv[VEC], wv[VEC], o[VEC];
        Vec<T, VEC>::ld(xv, xr + c);
        Vec<T, VEC>::ld(gv, gr + c);
        if (w) Vec<TW, VEC>::ld(wv, w + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float xh = xv[i] * rs;
          o[i] = rs * (gv[i] * (w ? wv[i] : 1.f) - xh * cterm);
        }
        if (res) {
          float rv[VEC];
          Vec<T, VEC>::ld(rv, res + row * cols + c);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] += rv[i];
        }
        Vec<T, VEC>::st(dxr + c, o);
      }
    }
    if (w) {
      for (int turn = 0; turn < 4; ++turn) {
        if (turn == wave && ok) {
          const T* xr = x + row * cols;
          const T* gr = dy + row * cols;
          for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
            float xv[VEC], gv[VEC];
            Vec<T, VEC>::ld(xv, xr + c);
            Vec<T, VEC>::ld(gv, gr + c);
#pragma unroll
            for (int i = 0; i < VEC; ++i) slab[c + i] += gv[i] * rnd<T>(xv[i] * rs);
          }
        }
        __syncthreads();
      }
    }
  }
}

// ----------------------------------------------------------------------------------------- LayerNorm
template <typename T, typename TW, int VEC>
__global__ __launch_bounds__(256) void layernorm_fwd_k(const T* __restrict__ x, const TW* __restrict__ w,
                                                       const TW* __restrict__ b, T* __restrict__ y,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       int64_t rows, int64_t cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * cols;
  float s = 0.f;
  for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
    float v[VEC];
    Vec<T, VEC>::ld(v, xr + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += v[i];
  }
  const float mean = wave_sum(s) / (float)cols;
  float ss = 0.f;
  for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
    float v[VEC];
    Vec<T, VEC>::ld(v, xr + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { const float d = v[i] - mean; ss += d * d; }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)cols + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  T* yr = y + row * cols;
  for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
    float v[VEC], g[VEC], bb[VEC];
    Vec<T, VEC>::ld(v, xr + c);
    if (w) Vec<TW, VEC>::ld(g, w + c);
    if (b) Vec<TW, VEC>::ld(bb, b + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = (v[i] - mean) * rstd * (w ? g[i] : 1.f) + (b ? bb[i] : 0.f);
    Vec<T, VEC>::st(yr + c, v);
  }
}

// partial: [gridDim.x][2*cols] = (dw | db)
template <typename T, typename TW, int VEC>
__global__ __launch_bounds__(256) void layernorm_bwd_k(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const TW* __restrict__ w, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, T* __restrict__ dx,
                                                       const T* __restrict__ res, float* __restrict__ partial,
                                                       int64_t rows, int64_t cols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* slab = partial ? partial + (int64_t)blockIdx.x * 2 * cols : nullptr;
  if (slab) {
    for (int64_t c = threadIdx.x; c < 2 * cols; c += 256) slab[c] = 0.f;
    __syncthreads();
  }
  for (int64_t base = (int64_t)blockIdx.x * 4; base < rows; base += (int64_t)gridDim.x * 4) {
    const int64_t row = base + wave;
    const bool ok = row < rows;
    float rs = 0.f, mu = 0.f;
    if (ok) {
      rs = rstd[row];
      mu = mean[row];
      const T* xr = x + row * cols;
      const T* gr = dy + row * cols;
      float s1 = 0.f, s2 = 0.f;
      for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
        float xv[VEC], gv[VEC], wv[VEC];
        Vec<T, VEC>::ld(xv, xr + c);
        Vec<T, VEC>::ld(gv, gr + c);
        if (w) Vec<TW, VEC>::ld(wv, w + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float g = gv[i] * (w ? wv[i] : 1.f);
          s1 += g;
          s2 += g * ((xv[i] - mu) * rs);
        }
      }
      const float c1 = wave_sum(s1) / (float)cols, c2 = wave_sum(s2) / (float)cols;
      T* dxr = dx + row * cols;
      for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
        float xv[VEC], gv[VEC], wv[VEC], o[VEC];
        Vec<T, VEC>::ld(xv, xr + c);
        Vec<T, VEC>::ld(gv, gr + c);
        if (w) Vec<TW, VEC>::ld(wv, w + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float g = gv[i] * (w ? wv[i] : 1.f);
          o[i] = rs * (g - c1 - (xv[i] - mu) * rs * c2);
        }
        if (res) {
          float rv[VEC];
          Vec<T, VEC>::ld(rv, res + row * cols + c);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] += rv[i];
        }
        Vec<T, VEC>::st(dxr + c, o);
      }
    }
    if (slab) {
      for (int turn = 0; turn < 4; ++turn) {
        if (turn == wave && ok) {
          const T* xr = x + row * cols;
          const T* gr = dy + row * cols;
          for (int64_t c = (int64_t)lane * VEC; c < cols; c += 64 * VEC) {
            float xv[VEC], gv[VEC];
            Vec<T, VEC>::ld(xv, xr + c);
            Vec<T, VEC>::ld(gv, gr + c);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              slab[c + i] += gv[i] * ((xv[i] - mu) * rs);
              slab[cols + c + i] += gv[i];
            }
          }
        }
        __syncthreads();
      }
    }
  }
}

// -------------------------------------------------------------------------------------------- colsum
// stage 1: grid (col tiles of 256, row splits); block 256 = 4 waves; lane owns 4 consecutive columns
// DIRECT: a single row split — the block's sums are the result: written (or accumulated) straight into `part` = out
template <typename T, bool DIRECT = false>
__global__ __launch_bounds__(256) void colsum_stage1_k(const T* __restrict__ x, int64_t ld, float* __restrict__ part,
                                                       int64_t rows, int64_t cols, int vec, int accumulate = 0) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * 256 + lane * 4;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < cols) {
    const int n_ok = (int)min((int64_t)4, cols - c0);
    for (int64_t r = r0 + wave; r < r1; r += 4) {
      const T* p = x + r * ld + c0;
      if (vec && n_ok == 4) {
        float v[4];
        Vec<T, 4>::ld(v, p);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += v[i];
      } else {
        for (int i = 0; i < n_ok; ++i) a[i] += ldf<T>(p + i);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][lane * 4 + i] = a[i];
  __syncthreads();
  const int t = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * 256 + t;
  if (c < cols) {
    const float sum = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    if (DIRECT) part[c] = accumulate ? part[c] + sum : sum;
    else part[(int64_t)blockIdx.y * cols + c] = sum;
  }
}
__global__ void colsum_stage2_k(const float* __restrict__ part, float* __restrict__ out, int nsplit, int64_t cols,
                                int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  // fixed order (split 0, 1, 2, ...), eight independent loads in flight
  float s = 0.f;
  int i = 0;
  for (; i + 8 <= nsplit; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(i + u) * cols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < nsplit; ++i) s += part[(int64_t)i * cols + c];
  out[c] = accumulate ? out[c] + s : s;
}

// Both stages in ONE launch: every (column tile, row split) workgroup publishes its partial sums with write-through (sc1)
// stores and bumps the column tile's arrival counter; the LAST workgroup to arrive adds the splits' partials in split order —
// the same order, hence the same bits, as colsum_stage2_k — and leaves the counter zeroed for the next launch.  Saves one
// launch (~4.5 us of kernel + a boundary) per bias / norm-weight gradient: ~280 of them in a DB-CogACT step.
template <typename T>
__global__ __launch_bounds__(256) void colsum_fused_k(const T* __restrict__ x, int64_t ld, float* __restrict__ part,
                                                      float* __restrict__ out, int64_t rows, int64_t cols, int vec,
                                                      int accumulate, int* __restrict__ cnt) {
  __shared__ float red[4][256];
  __shared__ int last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * 256 + lane * 4;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < cols) {
    const int n_ok = (int)min((int64_t)4, cols - c0);
    for (int64_t r = r0 + wave; r < r1; r += 4) {
      const T* p = x + r * ld + c0;
      if (vec && n_ok == 4) {
        float v[4];
        Vec<T, 4>::ld(v, p);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += v[i];
      } else {
        for (int i = 0; i < n_ok; ++i) a[i] += ldf<T>(p + i);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][lane * 4 + i] = a[i];
  __syncthreads();
  const int t = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * 256 + t;
  if (c < cols)
    __hip_atomic_store(part + (int64_t)blockIdx.y * cols + c, red[0][t] + red[1][t] + red[2][t] + red[3][t], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the write-through stores are acknowledged before the ticket
  __syncthreads();
  if (t == 0) {
    const int arrived = __hip_atomic_fetch_add(cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    last = arrived == (int)gridDim.y;
    if (last) __hip_atomic_store(cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last || c >= cols) return;
  const int nsplit = (int)gridDim.y;
  float s = 0.f;
  int i = 0;
  for (; i + 8 <= nsplit; i += 8) {                         // fixed order (split 0, 1, 2, ...), eight loads in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(part + (int64_t)(i + u) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < nsplit; ++i) s += __hip_atomic_load(part + (int64_t)i * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  out[c] = accumulate ? out[c] + s : s;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

// Fast backward (cols % 4 == 0, cols <= 4096): a wave owns whole rows, keeps its weight/bias-gradient partial
// sums in REGISTERS (16 x 4 columns per lane) over all its rows; no LDS, no barriers, no serialisation between the
// waves of a workgroup until the very end, where the four waves fold into ONE partial row; x/dy are read twice
// (second pass from L1/L2).  `res` (optional) is added to dx: the gradient of the residual branch around the norm'd
// sub-block, which used to be a separate add kernel.  partial: [gridDim.x][(LN ? 2 : 1) * cols].
template <typename T, typename TW, bool LN>
__global__ __launch_bounds__(256) void norm_bwd_fast_k(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const TW* __restrict__ w, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, T* __restrict__ dx,
                                                       const T* __restrict__ res, float* __restrict__ partial,
                                                       int64_t rows, int64_t cols) {
  constexpr int NIT = 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t prow = (int64_t)blockIdx.x * 4 + wave;
  float aw[NIT][4], ab[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) { aw[it][i] = 0.f; ab[it][i] = 0.f; }
  for (int64_t row = prow; row < rows; row += (int64_t)gridDim.x * 4) {
    const float rs = rstd[row];
    const float mu = LN ? mean[row] : 0.f;
    const T* xr = x + row * cols;
    const T* gr = dy + row * cols;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t c = ((int64_t)it * 64 + lane) * 4;
      if (c < cols) {
        float xv[4], gv[4], wv[4] = {1.f, 1.f, 1.f, 1.f};
        Vec<T, 4>::ld(xv, xr + c);
        Vec<T, 4>::ld(gv, gr + c);
        if (w) Vec<TW, 4>::ld(wv, w + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float g = gv[i] * wv[i];
          s1 += g;
          s2 += g * ((xv[i] - mu) * rs);
        }
      }
    }
    const float c1 = LN ? wave_sum(s1) / (float)cols : 0.f;
    const float c2 = wave_sum(s2) / (float)cols;
    T* dxr = dx + row * cols;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t c = ((int64_t)it * 64 + lane) * 4;
      if (c < cols) {
        float xv[4], gv[4], wv[4] = {1.f, 1.f, 1.f, 1.f}, o[4];
        Vec<T, 4>::ld(xv, xr + c);
        Vec<T, 4>::ld(gv, gr + c);
        if (w) Vec<TW, 4>::ld(wv, w + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float xh = (xv[i] - mu) * rs;
          o[i] = rs * (gv[i] * wv[i] - c1 - xh * c2);
          aw[it][i] += gv[i] * (LN ? xh : rnd<T>(xh));
          ab[it][i] += gv[i];
        }
        if (res) {                     // residual branch of the block: dx = norm backward + the gradient that bypassed it
          float rv[4];
          Vec<T, 4>::ld(rv, res + row * cols + c);
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] += rv[i];
        }
        Vec<T, 4>::st(dxr + c, o);
      }
    }
  }
  if (partial) {
    // the four waves fold their register partials through LDS in a fixed order (3, 2, 1, then wave 0 adds its own and
    // writes): ONE partial row per workgroup — a quarter of the former partial traffic and of the column sum after it
    __shared__ float sh[(LN ? 2 : 1) * 4096];
    for (int turn = 3; turn >= 1; --turn) {
      if (wave == turn) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int64_t c = ((int64_t)it * 64 + lane) * 4;
          if (c < cols) {
            float4* pw = reinterpret_cast<float4*>(sh + c);
            float4* pb = reinterpret_cast<float4*>(sh + cols + c);
            if (turn == 3) {
              *pw = make_float4(aw[it][0], aw[it][1], aw[it][2], aw[it][3]);
              if (LN) *pb = make_float4(ab[it][0], ab[it][1], ab[it][2], ab[it][3]);
            } else {
              float4 t = *pw;
              *pw = make_float4(t.x + aw[it][0], t.y + aw[it][1], t.z + aw[it][2], t.w + aw[it][3]);
              if (LN) { t = *pb; *pb = make_float4(t.x + ab[it][0], t.y + ab[it][1], t.z + ab[it][2], t.w + ab[it][3]); }
            }
          }
        }
      }
      __syncthreads();
    }
    if (wave == 0) {
      float* pr = partial + (int64_t)blockIdx.x * (LN ? 2 : 1) * cols;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int64_t c = ((int64_t)it * 64 + lane) * 4;
        if (c < cols) {
          const float4 t = *reinterpret_cast<const float4*>(sh + c);
          *reinterpret_cast<float4*>(pr + c) = make_float4(t.x + aw[it][0], t.y + aw[it][1], t.z + aw[it][2], t.w + aw[it][3]);
          if (LN) {
            const float4 u = *reinterpret_cast<const float4*>(sh + cols + c);
            *reinterpret_cast<float4*>(pr + cols + c) = make_float4(u.x + ab[it][0], u.y + ab[it][1], u.z + ab[it][2], u.w + ab[it][3]);
          }
        }
      }
    }
  }
}

// bf16 rows, cols % 8 == 0, cols <= 4096: the same backward with 16-byte accesses (8 columns per lane per step) and the row
// (x, dy) held in registers between the statistics pass and the dx pass: x and dy are read ONCE.
template <typename TW, bool LN>
__global__ __launch_bounds__(256) void norm_bwd_fast8_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                        const TW* __restrict__ w, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                        const bf16_t* __restrict__ res, float* __restrict__ partial,
                                                        int64_t rows, int64_t cols) {
  constexpr int NIT = 8;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t prow = (int64_t)blockIdx.x * 4 + wave;
  float aw[NIT][8], ab[LN ? NIT : 1][8];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) { aw[it][i] = 0.f; if (LN) ab[it][i] = 0.f; }
  auto unpack = [](const u32x4& v, float (&o)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(v[e] << 16); o[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
  };
  for (int64_t row = prow; row < rows; row += (int64_t)gridDim.x * 4) {
    const float rs = rstd[row];
    const float mu = LN ? mean[row] : 0.f;
    const bf16_t* xr = x + row * cols;
    const bf16_t* gr = dy + row * cols;
    u32x4 xv[NIT], gv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t c = ((int64_t)it * 64 + lane) * 8;
      xv[it] = c < cols ? *reinterpret_cast<const u32x4*>(xr + c) : (u32x4){0u, 0u, 0u, 0u};
      gv[it] = c < cols ? *reinterpret_cast<const u32x4*>(gr + c) : (u32x4){0u, 0u, 0u, 0u};
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t c = ((int64_t)it * 64 + lane) * 8;
      if (c < cols) {
        float xf[8], gf[8], w0[4] = {1.f, 1.f, 1.f, 1.f}, w1[4] = {1.f, 1.f, 1.f, 1.f};
        unpack(xv[it], xf); unpack(gv[it], gf);
        if (w) { Vec<TW, 4>::ld(w0, w + c); Vec<TW, 4>::ld(w1, w + c + 4); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float g = gf[i] * (i < 4 ? w0[i] : w1[i - 4]);
          s1 += g;
          s2 += g * ((xf[i] - mu) * rs);
        }
      }
    }
    const float c1 = LN ? wave_sum(s1) / (float)cols : 0.f;
    const float c2 = wave_sum(s2) / (float)cols;
    bf16_t* dxr = dx + row * cols;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t c = ((int64_t)it * 64 + lane) * 8;
      if (c < cols) {
        float xf[8], gf[8], o[8], w0[4] = {1.f, 1.f, 1.f, 1.f}, w1[4] = {1.f, 1.f, 1.f, 1.f};
        unpack(xv[it], xf); unpack(gv[it], gf);
        if (w) { Vec<TW, 4>::ld(w0, w + c); Vec<TW, 4>::ld(w1, w + c + 4); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xf[i] - mu) * rs;
          o[i] = rs * (gf[i] * (i < 4 ? w0[i] : w1[i - 4]) - c1 - xh * c2);
          aw[it][i] += gf[i] * (LN ? xh : rnd<bf16_t>(xh));
          if (LN) ab[it][i] += gf[i];
        }
        if (res) {
          float rf[8];
          unpack(*reinterpret_cast<const u32x4*>(res + row * cols + c), rf);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += rf[i];
        }
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
        *reinterpret_cast<u32x4*>(dxr + c) = ov;
      }
    }
  }
  if (partial) {
    __shared__ float sh[(LN ? 2 : 1) * 4096];
    for (int turn = 3; turn >= 1; --turn) {
      if (wave == turn) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int64_t c = ((int64_t)it * 64 + lane) * 8;
          if (c < cols) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              sh[c + i] = turn == 3 ? aw[it][i] : sh[c + i] + aw[it][i];
              if (LN) sh[cols + c + i] = turn == 3 ? ab[it][i] : sh[cols + c + i] + ab[it][i];
            }
          }
        }
      }
      __syncthreads();
    }
    if (wave == 0) {
      float* pr = partial + (int64_t)blockIdx.x * (LN ? 2 : 1) * cols;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int64_t c = ((int64_t)it * 64 + lane) * 8;
        if (c < cols) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            pr[c + i] = sh[c + i] + aw[it][i];
            if (LN) pr[cols + c + i] = sh[cols + c + i] + ab[it][i];
          }
        }
      }
    }
  }
}

// RMSNorm backward, bf16 rows, cols % 16 == 0, cols <= 8192 (round 5).  norm_bwd_fast8_k gives a wave a whole row: x and dy
// of the row (2 x 28 registers at 3584 columns) and the wave's column sums (56) leave two waves per SIMD, and every wave walks
// load -> reduce -> load residual -> store for 2-3 rows in sequence: 57 us for 139 MB (2.4 TB/s) on the decoder's [4592, 3584]
// rows.  Here a row is split between TWO waves (columns [0, cols/2) and [cols/2, cols)) of an 8-wave workgroup (4 rows at a
// time, as before): half the registers per wave, so twice the waves — and twice the bytes — in flight per CU; the residual is
// requested together with x and dy; the two halves of the row statistic meet in LDS and are added in a fixed order
// (deterministic).  Same partial-sum layout as the other backward kernels: one row of column sums per workgroup.
// NIT 16-byte chunks per lane: NIT x 64 lanes x 8 columns per half row (4: cols <= 4096); NS rows per workgroup pass (2 NS waves)
template <typename TW, int NIT, int NS>
__global__ __launch_bounds__(NS * 128, (NIT <= 4 ? NS : 2)) void rmsnorm_bwd_split_k(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const TW* __restrict__ w, const float* __restrict__ rstd,
    bf16_t* __restrict__ dx, const bf16_t* __restrict__ res, float* __restrict__ partial, int64_t rows, int64_t cols) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __shared__ float red[2][NS][2];          // [iteration parity][row slot][column half]
  __shared__ float sh[2][NIT * 512];       // hand-over of a row slot's column sums: [column half][column of the half]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = wave >> 1, half = wave & 1;
  const int64_t hc = cols >> 1, col0 = half * hc;
  auto unpack = [](const u32x4& v, float (&o)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(v[e] << 16); o[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
  };
  float aw[NIT][8];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) aw[it][i] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * NS;
  const int64_t iters = (rows + stride - 1) / stride;      // the same trip count for every wave: the loop holds a barrier
  const float inv_cols = 1.f / (float)cols;
  for (int64_t k = 0; k < iters; ++k) {
    const int64_t row = (int64_t)blockIdx.x * NS + slot + k * stride;
    const bool live = row < rows;
    const float rs = live ? rstd[row] : 0.f;
    u32x4 xv[NIT], gv[NIT], rv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t c = ((int64_t)it * 64 + lane) * 8;
      const bool in = live && c < hc;
      const int64_t off = row * cols + col0 + c;
      xv[it] = in ? *reinterpret_cast<const u32x4*>(x + off) : (u32x4){0u, 0u, 0u, 0u};
      gv[it] = in ? *reinterpret_cast<const u32x4*>(dy + off) : (u32x4){0u, 0u, 0u, 0u};
      rv[it] = (in && res) ? *reinterpret_cast<const u32x4*>(res + off) : (u32x4){0u, 0u, 0u, 0u};
    }
    float s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t c = ((int64_t)it * 64 + lane) * 8;
      if (c < hc) {
        float xf[8], gf[8], w0[4] = {1.f, 1.f, 1.f, 1.f}, w1[4] = {1.f, 1.f, 1.f, 1.f};
        unpack(xv[it], xf); unpack(gv[it], gf);
        if (w) { Vec<TW, 4>::ld(w0, w + col0 + c); Vec<TW, 4>::ld(w1, w + col0 + c + 4); }
#pragma unroll
        for (int i = 0; i < 8; ++i) s2 += gf[i] * (i < 4 ? w0[i] : w1[i - 4]) * (xf[i] * rs);
      }
    }
    s2 = wave_sum(s2);
    if (lane == 0) red[k & 1][slot][half] = s2;
    __syncthreads();                       // (parity-indexed slots: the next iteration's writes cannot overtake these reads)
    const float c2 = (red[k & 1][slot][0] + red[k & 1][slot][1]) * inv_cols;
    if (live) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int64_t c = ((int64_t)it * 64 + lane) * 8;
        if (c < hc) {
          float xf[8], gf[8], rf[8], o[8], w0[4] = {1.f, 1.f, 1.f, 1.f}, w1[4] = {1.f, 1.f, 1.f, 1.f};
          unpack(xv[it], xf); unpack(gv[it], gf); unpack(rv[it], rf);
          if (w) { Vec<TW, 4>::ld(w0, w + col0 + c); Vec<TW, 4>::ld(w1, w + col0 + c + 4); }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float xh = xf[i] * rs;
            o[i] = rs * (gf[i] * (i < 4 ? w0[i] : w1[i - 4]) - xh * c2) + rf[i];
            aw[it][i] += gf[i] * rnd<bf16_t>(xh);
          }
          u32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
          *reinterpret_cast<u32x4*>(dx + row * cols + col0 + c) = ov;
        }
      }
    }
  }
  if (partial) {
    // column sums: row slots NS-1 .. 1 hand theirs to slot 0's waves (same column half) through LDS, one after the other — a
    // fixed order of additions — and slot 0 writes the workgroup's row
    for (int turn = NS - 1; turn >= 1; --turn) {
      __syncthreads();
      if (slot == turn) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int64_t c = ((int64_t)it * 64 + lane) * 8;
          if (c < hc) {
            *reinterpret_cast<float4*>(&sh[half][c]) = make_float4(aw[it][0], aw[it][1], aw[it][2], aw[it][3]);
            *reinterpret_cast<float4*>(&sh[half][c + 4]) = make_float4(aw[it][4], aw[it][5], aw[it][6], aw[it][7]);
          }
        }
      }
      __syncthreads();
      if (slot == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int64_t c = ((int64_t)it * 64 + lane) * 8;
          if (c < hc) {
            const float4 a = *reinterpret_cast<const float4*>(&sh[half][c]), b = *reinterpret_cast<const float4*>(&sh[half][c + 4]);
            aw[it][0] += a.x; aw[it][1] += a.y; aw[it][2] += a.z; aw[it][3] += a.w;
            aw[it][4] += b.x; aw[it][5] += b.y; aw[it][6] += b.z; aw[it][7] += b.w;
          }
        }
      }
    }
    if (slot == 0) {
      float* pr = partial + (int64_t)blockIdx.x * cols + col0;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int64_t c = ((int64_t)it * 64 + lane) * 8;
        if (c < hc) {
          *reinterpret_cast<float4*>(pr + c) = make_float4(aw[it][0], aw[it][1], aw[it][2], aw[it][3]);
          *reinterpret_cast<float4*>(pr + c + 4) = make_float4(aw[it][4], aw[it][5], aw[it][6], aw[it][7]);
        }
      }
    }
  }
}

constexpr int NORM_BWD_MAX_BLOCKS = 512;
constexpr int64_t NORM_BWD_FAST_MAX_COLS = 4096;

template <bool LN>
bool launch_norm_bwd_fast(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx,
                          const void* res, float* partial, int64_t rows, int64_t cols, int dtype, int w_dtype, bool vec_ok, hipStream_t st) {
  int64_t g = (rows + 3) / 4;
  if (g > NORM_BWD_MAX_BLOCKS) g = NORM_BWD_MAX_BLOCKS;
  dim3 grid((unsigned)g);
  static const bool split_off = getenv("DXA_NORM_BWD_NO_SPLIT") != nullptr;      // A/B: profiles/r05_norm_bwd_split.txt
  if (!LN && !split_off && vec_ok && dtype == DXA_BF16 && cols % 16 == 0 && cols <= 8192 && (cols * 4) % 16 == 0 &&
      (!partial || (reinterpret_cast<uintptr_t>(partial) & 15) == 0)) {
    // DXA_NORM_BWD_ROWS=3: three rows per workgroup pass (384 threads, 168 registers: no spill) instead of four (512 threads at
    // 128 registers: a dozen spilled dwords) — tuning switch, profiles/r05_norm_bwd_split.txt
    static const int ns = getenv("DXA_NORM_BWD_ROWS") ? atoi(getenv("DXA_NORM_BWD_ROWS")) : 4;
#define LAUNCH_SPLIT(TW_, NIT_, NS_) hipLaunchKernelGGL((rmsnorm_bwd_split_k<TW_, NIT_, NS_>), grid, dim3(NS_ * 128), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const TW_*)w, rstd, (bf16_t*)dx, (const bf16_t*)res, partial, rows, cols)
    if (w_dtype == DXA_BF16) { if (cols > 4096) LAUNCH_SPLIT(bf16_t, 8, 4); else if (ns == 3) LAUNCH_SPLIT(bf16_t, 4, 3); else LAUNCH_SPLIT(bf16_t, 4, 4); }
    else { if (cols > 4096) LAUNCH_SPLIT(float, 8, 4); else if (ns == 3) LAUNCH_SPLIT(float, 4, 3); else LAUNCH_SPLIT(float, 4, 4); }
#undef LAUNCH_SPLIT
    return true;
  }
  if (!vec_ok || cols % 4 != 0 || cols > NORM_BWD_FAST_MAX_COLS) return false;
  if (dtype == DXA_BF16 && cols % 8 == 0 && !LN) {   // RMSNorm: 16-byte accesses, one read of x and dy (LayerNorm's second
                                                     // set of partial sums costs it the registers: measured 27 vs 24 us)
    if (w_dtype == DXA_BF16)
      hipLaunchKernelGGL((norm_bwd_fast8_k<bf16_t, LN>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, mean, rstd, (bf16_t*)dx, (const bf16_t*)res, partial, rows, cols);
    else
      hipLaunchKernelGGL((norm_bwd_fast8_k<float, LN>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const float*)w, mean, rstd, (bf16_t*)dx, (const bf16_t*)res, partial, rows, cols);
    return true;
  }
  if (dtype == DXA_BF16 && w_dtype == DXA_BF16)
    hipLaunchKernelGGL((norm_bwd_fast_k<bf16_t, bf16_t, LN>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, mean, rstd, (bf16_t*)dx, (const bf16_t*)res, partial, rows, cols);
  else if (dtype == DXA_BF16)
    hipLaunchKernelGGL((norm_bwd_fast_k<bf16_t, float, LN>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const float*)w, mean, rstd, (bf16_t*)dx, (const bf16_t*)res, partial, rows, cols);
  else
    hipLaunchKernelGGL((norm_bwd_fast_k<float, float, LN>), grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)w, mean, rstd, (float*)dx, (const float*)res, partial, rows, cols);
  return true;
}

}  // namespace

// number of partial rows both backward kernels write: one per workgroup (the generic fallback launches as many
// workgroups as the fast kernel, one slab each)
extern "C" int dxa_norm_bwd_blocks(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  if (g < 1) g = 1;
  if (g > NORM_BWD_MAX_BLOCKS) g = NORM_BWD_MAX_BLOCKS;
  return (int)g;
}

static int check_norm_dtypes(int dtype, int w_dtype, const char* who) {
  if (!(dtype == DXA_F32 || dtype == DXA_BF16) || !(w_dtype == DXA_F32 || w_dtype == DXA_BF16) ||
      (dtype == DXA_F32 && w_dtype == DXA_BF16)) {
    dxa_set_error("%s: unsupported dtype combination (%d,%d)", who, dtype, w_dtype);
    return DXA_ERR_UNSUPPORTED;
  }
  return DXA_OK;
}

namespace {
// bf16 rows of up to 8192 columns (cols % 8 == 0): the wave requests its whole row at once (16 bytes per lane per request, all
// of them in flight together), keeps it in registers for the statistics and the normalisation: ONE read of x, no second pass
template <typename TW>
__global__ __launch_bounds__(256) void rmsnorm_fwd_fast_k(const bf16_t* __restrict__ x, const TW* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out, int64_t rows,
                                                          int64_t cols, float eps) {
  constexpr int NIT = 16;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * cols;
  u32x4 v[NIT];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t c = ((int64_t)it * 64 + lane) * 8;
    v[it] = c < cols ? *reinterpret_cast<const u32x4*>(xr + c) : (u32x4){0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = __uint_as_float(v[it][e] << 16), b = __uint_as_float(v[it][e] & 0xffff0000u);
      ss += a * a + b * b;
    }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)cols + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  bf16_t* yr = y + row * cols;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t c = ((int64_t)it * 64 + lane) * 8;
    if (c < cols) {
      float g0[4] = {1.f, 1.f, 1.f, 1.f}, g1[4] = {1.f, 1.f, 1.f, 1.f};
      if (w) {
        Vec<TW, 4>::ld(g0, w + c);
        Vec<TW, 4>::ld(g1, w + c + 4);
      }
      const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = __uint_as_float(v[it][e] << 16), b = __uint_as_float(v[it][e] & 0xffff0000u);
        o[e] = pack_bf16x2(g[2 * e] * rnd<bf16_t>(a * rstd), g[2 * e + 1] * rnd<bf16_t>(b * rstd));
      }
      *reinterpret_cast<u32x4*>(yr + c) = o;
    }
  }
}
}  // namespace

extern "C" int dxa_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int64_t cols,
                               float eps, int dtype, int w_dtype, dxa_stream_t stream) {
  if (int rc = check_norm_dtypes(dtype, w_dtype, "dxa_rmsnorm_fwd")) return rc;
  DXA_CHECK_ARG(x && y && rows >= 0 && cols > 0, "dxa_rmsnorm_fwd: bad args");
  if (rows == 0) return DXA_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec_ok = al16(x) && al16(y) && (!w || al16(w));
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == DXA_BF16 && vec_ok && cols % 8 == 0 && cols <= 8192) {
    if (w_dtype == DXA_BF16) hipLaunchKernelGGL((rmsnorm_fwd_fast_k<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, cols, eps);
    else hipLaunchKernelGGL((rmsnorm_fwd_fast_k<float>), grid, dim3(256), 0, st, (const bf16_t*)x, (const float*)w, (bf16_t*)y, rstd, rows, cols, eps);
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  if (dtype == DXA_BF16 && w_dtype == DXA_BF16) {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((rmsnorm_fwd_k<bf16_t, bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, cols, eps);
    else hipLaunchKernelGGL((rmsnorm_fwd_k<bf16_t, bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, cols, eps);
  } else if (dtype == DXA_BF16) {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((rmsnorm_fwd_k<bf16_t, float, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, (const float*)w, (bf16_t*)y, rstd, rows, cols, eps);
    else hipLaunchKernelGGL((rmsnorm_fwd_k<bf16_t, float, 1>), grid, dim3(256), 0, st, (const bf16_t*)x, (const float*)w, (bf16_t*)y, rstd, rows, cols, eps);
  } else {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((rmsnorm_fwd_k<float, float, 4>), grid, dim3(256), 0, st, (const float*)x, (const float*)w, (float*)y, rstd, rows, cols, eps);
    else hipLaunchKernelGGL((rmsnorm_fwd_k<float, float, 1>), grid, dim3(256), 0, st, (const float*)x, (const float*)w, (float*)y, rstd, rows, cols, eps);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx,
                               const void* residual, float* partial_dw, int64_t rows, int64_t cols, int dtype, int w_dtype,
                               dxa_stream_t stream) {
  if (int rc = check_norm_dtypes(dtype, w_dtype, "dxa_rmsnorm_bwd")) return rc;
  DXA_CHECK_ARG(dy && x && rstd && dx && rows >= 0 && cols > 0, "dxa_rmsnorm_bwd: bad args");
  DXA_CHECK_ARG(!w || partial_dw, "dxa_rmsnorm_bwd: partial_dw required when w is given");
  if (rows == 0) return DXA_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec_ok = al16(x) && al16(dy) && al16(dx) && (!w || al16(w)) && (!partial_dw || al16(partial_dw)) && (!residual || al16(residual));
  if (launch_norm_bwd_fast<false>(dy, x, w, nullptr, rstd, dx, residual, w ? partial_dw : nullptr, rows, cols, dtype, w_dtype, vec_ok, st)) {
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  dim3 grid((unsigned)dxa_norm_bwd_blocks(rows));
  if (dtype == DXA_BF16 && w_dtype == DXA_BF16) {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((rmsnorm_bwd_k<bf16_t, bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (bf16_t*)dx, (const bf16_t*)residual, partial_dw, rows, cols);
    else hipLaunchKernelGGL((rmsnorm_bwd_k<bf16_t, bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (bf16_t*)dx, (const bf16_t*)residual, partial_dw, rows, cols);
  } else if (dtype == DXA_BF16) {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((rmsnorm_bwd_k<bf16_t, float, 4>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const float*)w, rstd, (bf16_t*)dx, (const bf16_t*)residual, partial_dw, rows, cols);
    else hipLaunchKernelGGL((rmsnorm_bwd_k<bf16_t, float, 1>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const float*)w, rstd, (bf16_t*)dx, (const bf16_t*)residual, partial_dw, rows, cols);
  } else {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((rmsnorm_bwd_k<float, float, 4>), grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)w, rstd, (float*)dx, (const float*)residual, partial_dw, rows, cols);
    else hipLaunchKernelGGL((rmsnorm_bwd_k<float, float, 1>), grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)w, rstd, (float*)dx, (const float*)residual, partial_dw, rows, cols);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd,
                                 int64_t rows, int64_t cols, float eps, int dtype, int w_dtype,
                                 dxa_stream_t stream) {
  if (int rc = check_norm_dtypes(dtype, w_dtype, "dxa_layernorm_fwd")) return rc;
  DXA_CHECK_ARG(x && y && rows >= 0 && cols > 0, "dxa_layernorm_fwd: bad args");
  if (rows == 0) return DXA_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec_ok = al16(x) && al16(y) && (!w || al16(w)) && (!b || al16(b));
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == DXA_BF16 && w_dtype == DXA_BF16) {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((layernorm_fwd_k<bf16_t, bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
    else hipLaunchKernelGGL((layernorm_fwd_k<bf16_t, bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
  } else if (dtype == DXA_BF16) {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((layernorm_fwd_k<bf16_t, float, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, (const float*)w, (const float*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
    else hipLaunchKernelGGL((layernorm_fwd_k<bf16_t, float, 1>), grid, dim3(256), 0, st, (const bf16_t*)x, (const float*)w, (const float*)b, (bf16_t*)y, mean, rstd, rows, cols, eps);
  } else {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((layernorm_fwd_k<float, float, 4>), grid, dim3(256), 0, st, (const float*)x, (const float*)w, (const float*)b, (float*)y, mean, rstd, rows, cols, eps);
    else hipLaunchKernelGGL((layernorm_fwd_k<float, float, 1>), grid, dim3(256), 0, st, (const float*)x, (const float*)w, (const float*)b, (float*)y, mean, rstd, rows, cols, eps);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean,
                                 const float* rstd, void* dx, const void* residual, float* partial_dwdb, int64_t rows,
                                 int64_t cols, int dtype, int w_dtype, dxa_stream_t stream) {
  if (int rc = check_norm_dtypes(dtype, w_dtype, "dxa_layernorm_bwd")) return rc;
  DXA_CHECK_ARG(dy && x && mean && rstd && dx && rows >= 0 && cols > 0, "dxa_layernorm_bwd: bad args");
  DXA_CHECK_ARG(!w || partial_dwdb, "dxa_layernorm_bwd: partial_dwdb required when w is given");
  if (rows == 0) return DXA_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec_ok = al16(x) && al16(dy) && al16(dx) && (!w || al16(w)) && (!partial_dwdb || al16(partial_dwdb)) && (!residual || al16(residual));
  if (launch_norm_bwd_fast<true>(dy, x, w, mean, rstd, dx, residual, w ? partial_dwdb : nullptr, rows, cols, dtype, w_dtype, vec_ok, st)) {
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  dim3 grid((unsigned)dxa_norm_bwd_blocks(rows));
  float* part = w ? partial_dwdb : nullptr;
  if (dtype == DXA_BF16 && w_dtype == DXA_BF16) {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((layernorm_bwd_k<bf16_t, bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, mean, rstd, (bf16_t*)dx, (const bf16_t*)residual, part, rows, cols);
    else hipLaunchKernelGGL((layernorm_bwd_k<bf16_t, bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, mean, rstd, (bf16_t*)dx, (const bf16_t*)residual, part, rows, cols);
  } else if (dtype == DXA_BF16) {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((layernorm_bwd_k<bf16_t, float, 4>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const float*)w, mean, rstd, (bf16_t*)dx, (const bf16_t*)residual, part, rows, cols);
    else hipLaunchKernelGGL((layernorm_bwd_k<bf16_t, float, 1>), grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const float*)w, mean, rstd, (bf16_t*)dx, (const bf16_t*)residual, part, rows, cols);
  } else {
    if (vec_ok && cols % 4 == 0) hipLaunchKernelGGL((layernorm_bwd_k<float, float, 4>), grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)w, mean, rstd, (float*)dx, (const float*)residual, part, rows, cols);
    else hipLaunchKernelGGL((layernorm_bwd_k<float, float, 1>), grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)w, mean, rstd, (float*)dx, (const float*)residual, part, rows, cols);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

namespace {
// arrival counters of colsum_fused_k: COLSUM_CNT ints per (device, stream), zeroed once (the kernel leaves them zeroed),
// allocated on first use — so the first column sum on a stream must not run under stream capture
constexpr int COLSUM_CNT = 4096;
int colsum_counters(hipStream_t st, int** out) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, int*> tab;
  int dev = 0;
  DXA_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  auto it = tab.find({dev, st});
  if (it == tab.end()) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      *out = nullptr;                      // under capture before the counters exist: the two-launch form needs none
      return DXA_OK;
    }
    int* p = nullptr;
    DXA_CHECK_HIP(hipMalloc((void**)&p, COLSUM_CNT * sizeof(int)));
    DXA_CHECK_HIP(hipMemset(p, 0, COLSUM_CNT * sizeof(int)));
    DXA_CHECK_HIP(hipDeviceSynchronize());
    it = tab.emplace(std::make_pair(dev, st), p).first;
  }
  *out = it->second;
  return DXA_OK;
}
}  // namespace

extern "C" int dxa_colsum(const void* x, int64_t ld, float* out, int64_t rows, int64_t cols, int dtype,
                          int accumulate, float* scratch, size_t scratch_bytes, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && out && rows >= 0 && cols > 0 && ld >= cols, "dxa_colsum: bad args");
  DXA_CHECK_ARG(dtype == DXA_F32 || dtype == DXA_BF16, "dxa_colsum: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  // row splits of >= 32 rows, at most 64 of them: the <= 512 partial rows of a norm backward become 16 x 14 workgroups
  int nsplit = (int)((rows + 31) / 32);
  if (nsplit < 1) nsplit = 1;
  if (nsplit > 64) nsplit = 64;
  DXA_CHECK_ARG(scratch && scratch_bytes >= (size_t)nsplit * cols * sizeof(float),
                "dxa_colsum: scratch too small (need %zu bytes)", (size_t)nsplit * cols * sizeof(float));
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)nsplit);
  const size_t es = dtype == DXA_BF16 ? 2 : 4;
  const int vec = ((reinterpret_cast<uintptr_t>(x) % (4 * es)) == 0) && (ld % 4 == 0);
  if (rows <= 32) {
    // a handful of rows: one launch, the block's sums are the result
    dim3 g1((unsigned)((cols + 255) / 256), 1);
    if (dtype == DXA_BF16)
      hipLaunchKernelGGL((colsum_stage1_k<bf16_t, true>), g1, dim3(256), 0, st, (const bf16_t*)x, ld, out, rows, cols, vec, accumulate);
    else
      hipLaunchKernelGGL((colsum_stage1_k<float, true>), g1, dim3(256), 0, st, (const float*)x, ld, out, rows, cols, vec, accumulate);
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  static const bool two_stage = getenv("DXA_COLSUM_TWO_STAGE") != nullptr;
  int* cnt = nullptr;
  if (!two_stage && grid.x <= COLSUM_CNT) {
    if (int rc = colsum_counters(st, &cnt)) return rc;
  }
  if (cnt != nullptr) {
    if (dtype == DXA_BF16)
      hipLaunchKernelGGL((colsum_fused_k<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, ld, scratch, out, rows, cols, vec, accumulate, cnt);
    else
      hipLaunchKernelGGL((colsum_fused_k<float>), grid, dim3(256), 0, st, (const float*)x, ld, scratch, out, rows, cols, vec, accumulate, cnt);
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  if (dtype == DXA_BF16)
    hipLaunchKernelGGL((colsum_stage1_k<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, ld, scratch, rows, cols, vec);
  else
    hipLaunchKernelGGL((colsum_stage1_k<float>), grid, dim3(256), 0, st, (const float*)x, ld, scratch, rows, cols, vec);
  hipLaunchKernelGGL(colsum_stage2_k, dim3((unsigned)((cols + 63) / 64)), dim3(64), 0, st, scratch, out, nsplit,
                     cols, accumulate);    // one wave per 64 columns: 56 workgroups for 3584 columns instead of 14
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
