// HBM-bound elementwise / data-movement kernels of the CogACT path: RoPE+QKV split, SwiGLU, activation
// helpers, token splice (embedding gather + image-feature insertion), patch im2col, ViT embedding
// assembly, diffusion glue (q_sample, timestep embedding, DiT token assembly, CFG token drop, MSE,
// DDIM update).  8/16-byte vector accesses wherever rows are 4-element aligned; grid-stride loops
// capped at a few workgroups per CU.
#include "common.h"

namespace {

constexpr int TPB = 256;

// ------------------------------------------------------------------------------------------------ RoPE
// work item = (token, head-slot, group of VEC dims in the first half); VEC = 8 for bf16 heads with D % 16 == 0: 16-byte accesses
template <typename T, bool MERGE, int VEC = 4>
__global__ __launch_bounds__(TPB) void rope_k(const T* __restrict__ tok, T* __restrict__ tok_out,
                                              T* __restrict__ q, T* __restrict__ k, T* __restrict__ v,
                                              const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                              const int32_t* __restrict__ pos, int B, int S, int Hq, int Hkv, int D, int Sc, int s0) {
  const int HS = Hq + 2 * Hkv, half = D / 2, qn = half / VEC;
  const int64_t total = (int64_t)B * S * HS * qn;
  const int64_t ld = (int64_t)HS * D;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int qd = (int)(it % qn);
    const int hs = (int)((it / qn) % HS);
    const int64_t t = it / ((int64_t)qn * HS);
    const int b = (int)(t / S), s = (int)(t % S);
    const int d0 = qd * VEC;
    T* hm;  // head-major row of this (b, head, s)
    bool rot = true;
    // head-major tensors hold Sc >= S positions per (b, head); this call's S tokens sit at positions s0 .. s0 + S - 1 (Sc = S, s0 = 0:
    // tensors of their own; otherwise a slice of a tensor several calls share: pi0's two experts, a key / value cache)
    if (hs < Hq) hm = q + (((int64_t)b * Hq + hs) * Sc + s0 + s) * D;
    else if (hs < Hq + Hkv) hm = k + (((int64_t)b * Hkv + (hs - Hq)) * Sc + s0 + s) * D;
    else { hm = v + (((int64_t)b * Hkv + (hs - Hq - Hkv)) * Sc + s0 + s) * D; rot = false; }
    const int64_t toff = t * ld + (int64_t)hs * D;
    float c[VEC], sn[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { c[i] = 1.f; sn[i] = 0.f; }
    if (rot) {
      const int64_t pr = pos ? pos[t] : s;
#pragma unroll
      for (int u = 0; u < VEC / 4; ++u) {
        float c4[4], s4[4];
        Vec<float, 4>::ld(c4, cos_t + pr * half + d0 + 4 * u);
        Vec<float, 4>::ld(s4, sin_t + pr * half + d0 + 4 * u);
#pragma unroll
        for (int i = 0; i < 4; ++i) { c[4 * u + i] = c4[i]; sn[4 * u + i] = s4[i]; }
      }
    }
    float x1[VEC], x2[VEC], o1[VEC], o2[VEC];
    if (!MERGE) {
      Vec<T, VEC>::ld(x1, tok + toff + d0);
      Vec<T, VEC>::ld(x2, tok + toff + d0 + half);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (rot) {
          const float cc = rnd<T>(c[i]), ss = rnd<T>(sn[i]);
          o1[i] = rnd<T>(x1[i] * cc) + rnd<T>(-x2[i] * ss);
          o2[i] = rnd<T>(x2[i] * cc) + rnd<T>(x1[i] * ss);
        } else { o1[i] = x1[i]; o2[i] = x2[i]; }
      }
      Vec<T, VEC>::st(hm + d0, o1);
      Vec<T, VEC>::st(hm + d0 + half, o2);
    } else {
      Vec<T, VEC>::ld(x1, hm + d0);
      Vec<T, VEC>::ld(x2, hm + d0 + half);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (rot) {
          const float cc = rnd<T>(c[i]), ss = rnd<T>(sn[i]);
          o1[i] = x1[i] * cc + x2[i] * ss;
          o2[i] = x2[i] * cc - x1[i] * ss;
        } else { o1[i] = x1[i]; o2[i] = x2[i]; }
      }
      Vec<T, VEC>::st(tok_out + toff + d0, o1);
      Vec<T, VEC>::st(tok_out + toff + d0 + half, o2);
    }
  }
}

// ---------------------------------------------------------------------------------------------- SwiGLU
template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void swiglu_fwd_k(const T* __restrict__ gu, T* __restrict__ out, int64_t rows, int64_t F) {
  const int64_t per = F / VEC, total = rows * per;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t r = it / per, c = (it % per) * VEC;
    float g[VEC], u[VEC], o[VEC];
    Vec<T, VEC>::ld(g, gu + r * 2 * F + c);
    Vec<T, VEC>::ld(u, gu + r * 2 * F + F + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = rnd<T>(g[i] / (1.f + expf(-g[i]))) * u[i];
    Vec<T, VEC>::st(out + r * F + c, o);
  }
}
template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void swiglu_bwd_k(const T* __restrict__ gu, const T* __restrict__ dout,
                                                    T* __restrict__ dgu, int64_t rows, int64_t F) {
  const int64_t per = F / VEC, total = rows * per;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t r = it / per, c = (it % per) * VEC;
    float g[VEC], u[VEC], d[VEC], dg[VEC], du[VEC];
    Vec<T, VEC>::ld(g, gu + r * 2 * F + c);
    Vec<T, VEC>::ld(u, gu + r * 2 * F + F + c);
    Vec<T, VEC>::ld(d, dout + r * F + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float s = 1.f / (1.f + expf(-g[i]));
      du[i] = d[i] * rnd<T>(g[i] * s);
      dg[i] = d[i] * u[i] * (s * (1.f + g[i] * (1.f - s)));
    }
    Vec<T, VEC>::st(dgu + r * 2 * F + c, dg);
    Vec<T, VEC>::st(dgu + r * 2 * F + F + c, du);
  }
}

// gated MLP with any gate activation: out = act(gate) * up (Gemma's GeGLU = gelu_pytorch_tanh, HF gemma/modeling_gemma.py
// GemmaMLP as used by pi0_arch.py:205-208); same [gate ; up] packing as SwiGLU
template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void glu_fwd_k(const T* __restrict__ gu, T* __restrict__ out, int64_t rows, int64_t F, int act) {
  const int64_t per = F / VEC, total = rows * per;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t r = it / per, c = (it % per) * VEC;
    float g[VEC], u[VEC], o[VEC];
    Vec<T, VEC>::ld(g, gu + r * 2 * F + c);
    Vec<T, VEC>::ld(u, gu + r * 2 * F + F + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = rnd<T>(act_fwd(act, g[i])) * u[i];
    Vec<T, VEC>::st(out + r * F + c, o);
  }
}
template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void glu_bwd_k(const T* __restrict__ gu, const T* __restrict__ dout, T* __restrict__ dgu,
                                                 int64_t rows, int64_t F, int act) {
  const int64_t per = F / VEC, total = rows * per;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t r = it / per, c = (it % per) * VEC;
    float g[VEC], u[VEC], d[VEC], dg[VEC], du[VEC];
    Vec<T, VEC>::ld(g, gu + r * 2 * F + c);
    Vec<T, VEC>::ld(u, gu + r * 2 * F + F + c);
    Vec<T, VEC>::ld(d, dout + r * F + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      du[i] = d[i] * rnd<T>(act_fwd(act, g[i]));
      dg[i] = d[i] * u[i] * act_grad(act, g[i]);
    }
    Vec<T, VEC>::st(dgu + r * 2 * F + c, dg);
    Vec<T, VEC>::st(dgu + r * 2 * F + F + c, du);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void act_fwd_k(const T* __restrict__ x, T* __restrict__ y, int64_t n, int act) {
  const int64_t total = n / VEC;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    float v[VEC];
    Vec<T, VEC>::ld(v, x + it * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = act_fwd(act, v[i]);
    Vec<T, VEC>::st(y + it * VEC, v);
  }
}
template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void act_bwd_k(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                 int64_t n, int act) {
  const int64_t total = n / VEC;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    float v[VEC], g[VEC];
    Vec<T, VEC>::ld(v, x + it * VEC);
    Vec<T, VEC>::ld(g, dy + it * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = g[i] * act_grad(act, v[i]);
    Vec<T, VEC>::st(dx + it * VEC, v);
  }
}
template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void add_k(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t n) {
  const int64_t total = n / VEC;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    float x[VEC], y[VEC];
    Vec<T, VEC>::ld(x, a + it * VEC);
    Vec<T, VEC>::ld(y, b + it * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] += y[i];
    Vec<T, VEC>::st(o + it * VEC, x);
  }
}
template <typename TS, typename TD, int VEC>
__global__ __launch_bounds__(TPB) void cast_k(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  const int64_t total = n / VEC;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    float x[VEC];
    Vec<TS, VEC>::ld(x, s + it * VEC);
    Vec<TD, VEC>::st(d + it * VEC, x);
  }
}
template <typename TS, typename TD>
__global__ __launch_bounds__(TPB) void copy2d_k(const TS* __restrict__ s, int64_t lds, TD* __restrict__ d, int64_t ldd,
                                                int64_t rows, int64_t cols, int64_t cols_padded) {
  const int64_t total = rows * cols_padded;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t r = it / cols_padded, c = it % cols_padded;
    stf<TD>(d + r * ldd + c, c < cols ? ldf<TS>(s + r * lds + c) : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------- splice
constexpr int64_t PLAN_PAD = INT64_MIN;
template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void splice_fwd_k(const int64_t* __restrict__ plan, const T* __restrict__ embed,
                                                    const T* __restrict__ img, T* __restrict__ out, int64_t n_rows, int64_t d) {
  const int64_t per = d / VEC, total = n_rows * per;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t r = it / per, c = (it % per) * VEC;
    const int64_t p = plan[r];
    float v[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = 0.f;
    if (p >= 0) Vec<T, VEC>::ld(v, embed + p * d + c);
    else if (p != PLAN_PAD) Vec<T, VEC>::ld(v, img + (-1 - p) * d + c);
    Vec<T, VEC>::st(out + r * d + c, v);
  }
}
// Backward of the splice.  One workgroup per (output row r, 256*VEC-column chunk).  Image rows are used once: plain store.
// Token rows add into the dense fp32 embedding gradient WITHOUT atomics: the workgroup of the FIRST row carrying a token
// id sums, in ascending row order, every row with that id (ballot over 256-row windows of the plan) and adds the total
// to the gradient row; workgroups of later duplicates exit.  The plan (n_rows int64, L2 resident) is re-scanned per
// workgroup: a few us in total, and the result is bitwise reproducible whatever the dispatch order.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void splice_bwd_k(const int64_t* __restrict__ plan, const T* __restrict__ dout,
                                                    float* __restrict__ d_embed, T* __restrict__ d_img, int64_t n_rows, int64_t d) {
  __shared__ unsigned long long s_mask[4];
  __shared__ int s_dup;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int64_t r = blockIdx.x;
  const int64_t c = ((int64_t)blockIdx.y * 256 + tid) * VEC;
  const bool col_ok = c < d;
  const int64_t p = plan[r];
  if (p == PLAN_PAD) return;
  if (p < 0) {
    if (d_img && col_ok) {
      float v[VEC];
      Vec<T, VEC>::ld(v, dout + r * d + c);
      Vec<T, VEC>::st(d_img + (-1 - p) * d + c, v);
    }
    return;
  }
  if (!d_embed) return;
  if (tid == 0) s_dup = 0;
  __syncthreads();
  for (int64_t i = tid; i < r; i += 256)
    if (plan[i] == p) s_dup = 1;
  __syncthreads();
  if (s_dup) return;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  for (int64_t base = r - (r & 255); base < n_rows; base += 256) {
    const int64_t i = base + tid;
    const bool m = i >= r && i < n_rows && plan[i] == p;
    const unsigned long long bal = __ballot(m);
    if ((tid & 63) == 0) s_mask[wave] = bal;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned long long mm = s_mask[w];
      while (mm) {
        const int b = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        if (col_ok) {
          float v[VEC];
          Vec<T, VEC>::ld(v, dout + (base + w * 64 + b) * d + c);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += v[e];
        }
      }
    }
    __syncthreads();
  }
  if (col_ok) {
    float* g = d_embed + p * d + c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] += acc[e];
  }
}
// zero the embedding-gradient rows a previous plan touched (sparse re-zero instead of a 2 GB memset per step)
__global__ __launch_bounds__(256) void zero_rows_k(const int64_t* __restrict__ plan, float* __restrict__ g, int64_t n_rows, int64_t d) {
  const int64_t r = blockIdx.x;
  const int64_t p = plan[r];
  if (p < 0) return;
  for (int64_t c = threadIdx.x; c < d; c += 256) g[p * d + c] = 0.f;
}
template <typename TS, typename TD>
__global__ __launch_bounds__(TPB) void gather_rows_k(const TS* __restrict__ x, const int64_t* __restrict__ idx,
                                                     TD* __restrict__ out, int64_t n, int64_t d) {
  const int64_t total = n * d;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t r = it / d, c = it % d;
    stf<TD>(out + it, ldf<TS>(x + idx[r] * d + c));
  }
}
// dx [R, d] fully written: row r = sum of dout[i] with idx[i] == r, else 0
template <typename TS, typename TD>
__global__ __launch_bounds__(TPB) void scatter_rows_k(const TS* __restrict__ dout, const int64_t* __restrict__ idx,
                                                      TD* __restrict__ dx, int64_t n, int64_t R, int64_t d) {
  const int64_t total = R * d;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t r = it / d, c = it % d;
    float s = 0.f;
    for (int64_t i = 0; i < n; ++i)
      if (idx[i] == r) s += ldf<TS>(dout + i * d + c);
    stf<TD>(dx + it, s);
  }
}

// ----------------------------------------------------------------------------------------- ViT front-end
template <typename TI, typename T>
__global__ __launch_bounds__(TPB) void im2col_k(const TI* __restrict__ img, T* __restrict__ rows, int N, int H, int W,
                                                int P, int64_t ld) {
  const int gh = H / P, gw = W / P, KK = 3 * P * P;
  const int64_t total = (int64_t)N * gh * gw * ld;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int col = (int)(it % ld);
    const int64_t row = it / ld;
    float v = 0.f;
    if (col < KK) {
      const int c = col / (P * P), py = (col / P) % P, px = col % P;
      const int gx = (int)(row % gw), gy = (int)((row / gw) % gh);
      const int64_t n = row / ((int64_t)gw * gh);
      v = ldf<TI>(img + ((n * 3 + c) * H + (gy * P + py)) * (int64_t)W + gx * P + px);
    }
    stf<T>(rows + it, v);
  }
}
template <typename T, typename TW>
__global__ __launch_bounds__(TPB) void vit_embed_fwd_k(const T* __restrict__ patch, const TW* __restrict__ cls,
                                                       const TW* __restrict__ pos, T* __restrict__ x, int N, int np, int C) {
  const int64_t total = (int64_t)N * (np + 1) * C;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int c = (int)(it % C);
    const int t = (int)((it / C) % (np + 1));
    const int64_t n = it / ((int64_t)C * (np + 1));
    const float base = t == 0 ? ldf<TW>(cls + c) : ldf<T>(patch + (n * np + (t - 1)) * C + c);
    stf<T>(x + it, rnd<T>(base) + ldf<TW>(pos + (int64_t)t * C + c));
  }
}
template <typename T>
__global__ __launch_bounds__(TPB) void vit_embed_bwd_k(const T* __restrict__ dx, T* __restrict__ dpatch, int N, int np, int C) {
  const int64_t total = (int64_t)N * np * C;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int c = (int)(it % C);
    const int p = (int)((it / C) % np);
    const int64_t n = it / ((int64_t)C * np);
    dpatch[it] = dx[(n * (np + 1) + p + 1) * C + c];
  }
}

// --------------------------------------------------------------------------------------- diffusion glue
__global__ __launch_bounds__(TPB) void qsample_k(const float* __restrict__ x0, const float* __restrict__ noise,
                                                 const float* __restrict__ a, const float* __restrict__ s,
                                                 float* __restrict__ xt, int64_t N, int64_t per) {
  const int64_t total = N * per;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t n = it / per;
    xt[it] = a[n] * x0[it] + s[n] * noise[it];
  }
}
__global__ __launch_bounds__(TPB) void timestep_embedding_k(const float* __restrict__ t, const float* __restrict__ freqs,
                                                            float* __restrict__ out, int64_t N, int half) {
  const int64_t total = N * half;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t n = it / half;
    const int j = (int)(it % half);
    const float arg = t[n] * freqs[j];
    out[n * 2 * half + j] = cosf(arg);
    out[n * 2 * half + half + j] = sinf(arg);
  }
}
__global__ __launch_bounds__(TPB) void dit_assemble_fwd_k(const float* __restrict__ xe, const float* __restrict__ te,
                                                          const float* __restrict__ ze, const float* __restrict__ pos,
                                                          float* __restrict__ h, int N, int T, int hd) {
  const int64_t total = (int64_t)N * (T + 1) * hd;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int c = (int)(it % hd);
    const int t = (int)((it / hd) % (T + 1));
    const int64_t n = it / ((int64_t)hd * (T + 1));
    const float base = t == 0 ? te[n * hd + c] + ze[n * hd + c] : xe[(n * T + (t - 1)) * hd + c];
    h[it] = base + pos[(int64_t)t * hd + c];
  }
}
__global__ __launch_bounds__(TPB) void dit_assemble_bwd_k(const float* __restrict__ dh, float* __restrict__ dxe,
                                                          float* __restrict__ dc, int N, int T, int hd) {
  const int64_t total = (int64_t)N * (T + 1) * hd;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int c = (int)(it % hd);
    const int t = (int)((it / hd) % (T + 1));
    const int64_t n = it / ((int64_t)hd * (T + 1));
    if (t == 0) dc[n * hd + c] = dh[it];
    else dxe[(n * T + (t - 1)) * hd + c] = dh[it];
  }
}
__global__ __launch_bounds__(TPB) void token_drop_k(const float* __restrict__ z, const float* __restrict__ unc,
                                                    const uint8_t* __restrict__ drop, float* __restrict__ out, int64_t N, int64_t d) {
  const int64_t total = N * d;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t n = it / d;
    out[it] = drop[n] ? unc[it % d] : z[it];
  }
}
// dz[n] = drop ? 0 : dout[n];  dunc[c] (+)= sum_{n dropped} dout[n,c]
__global__ __launch_bounds__(TPB) void token_drop_bwd_k(const float* __restrict__ dout, const uint8_t* __restrict__ drop,
                                                        float* __restrict__ dz, float* __restrict__ dunc, int64_t N, int64_t d,
                                                        int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (c >= d) return;
  float s = 0.f;
  for (int64_t n = 0; n < N; ++n) {
    const float g = dout[n * d + c];
    if (drop[n]) { s += g; if (dz) dz[n * d + c] = 0.f; }
    else if (dz) dz[n * d + c] = g;
  }
  if (dunc) dunc[c] = accumulate ? dunc[c] + s : s;
}
__global__ __launch_bounds__(1024) void mse_loss_k(const float* __restrict__ pred, const float* __restrict__ target,
                                                   float* __restrict__ loss, float* __restrict__ dpred, int64_t n, float gscale) {
  __shared__ float red[16];
  float s = 0.f;
  const float k = 2.f / (float)n * gscale;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float d = pred[i] - target[i];
    s += d * d;
    if (dpred) dpred[i] = k * d;
  }
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = tot / (float)n;
}
// sample-weighted eps-MSE of HybridCogACT (hybrid_cogact_arch.py:168-173): loss = sum_r w_r mean_c(d_rc^2) / (sum_r w_r + 1e-6)
__global__ __launch_bounds__(1024) void mse_loss_rows_k(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const float* __restrict__ row_w, float* __restrict__ loss,
                                                        float* __restrict__ dpred, int64_t rows, int64_t cols, float gscale) {
  __shared__ float red[16];
  float ws = 0.f;
  for (int64_t r = threadIdx.x; r < rows; r += 1024) ws += row_w[r];
  const float wsum = block_sum(ws, red) + 1e-6f;
  __syncthreads();
  const int64_t n = rows * cols;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float w = row_w[i / cols], d = pred[i] - target[i];
    s += w * d * d;
    if (dpred) dpred[i] = 2.f * w * d / ((float)cols * wsum) * gscale;
  }
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = tot / ((float)cols * wsum);
}
__global__ __launch_bounds__(TPB) void ddim_step_k(float* __restrict__ x, const float* __restrict__ mo, int64_t B, int64_t per,
                                                   int use_cfg, float cfg_scale, float c_recip, float c_recipm1, float ab_prev) {
  const int64_t total = B * per;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    float eps;
    if (use_cfg) {
      const float ec = mo[it], eu = mo[total + it];
      eps = eu + cfg_scale * (ec - eu);
    } else {
      eps = mo[it];
    }
    const float xv = x[it];
    const float x0 = c_recip * xv - c_recipm1 * eps;
    const float eps2 = (c_recip * xv - x0) / c_recipm1;
    const float xn = x0 * sqrtf(ab_prev) + sqrtf(1.f - ab_prev - 0.f) * eps2;
    x[it] = xn;
    if (use_cfg) x[total + it] = xn;
  }
}

inline bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline bool ok_dtype(int d) { return d == DXA_F32 || d == DXA_BF16; }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int dxa_rope_split_at(const void* qkv, void* q, void* k, void* v, const float* cos_t, const float* sin_t,
                                 const int32_t* pos, int B, int S, int Hq, int Hkv, int D, int S_cap, int s0, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(qkv && q && k && v && cos_t && sin_t && ok_dtype(dtype), "dxa_rope_split: bad args");
  DXA_CHECK_ARG(D % 8 == 0, "dxa_rope_split: head_dim must be a multiple of 8 (got %d)", D);
  DXA_CHECK_ARG(s0 >= 0 && S >= 0 && s0 + S <= S_cap, "dxa_rope_split: positions %d .. %d outside the %d the head-major tensors hold", s0, s0 + S, S_cap);
  const int64_t total = (int64_t)B * S * (Hq + 2 * Hkv) * (D / 8);
  if (total == 0) return DXA_OK;
  dim3 grid(dxa_grid1d(total, TPB));
  const bool wide = dtype == DXA_BF16 && D % 16 == 0 && al(qkv, 16) && al(q, 16) && al(k, 16) && al(v, 16);
  if (wide)
    hipLaunchKernelGGL((rope_k<bf16_t, false, 8>), dim3(dxa_grid1d(total / 2, TPB)), dim3(TPB), 0, ST, (const bf16_t*)qkv, (bf16_t*)nullptr, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, cos_t, sin_t, pos, B, S, Hq, Hkv, D, S_cap, s0);
  else if (dtype == DXA_BF16)
    hipLaunchKernelGGL((rope_k<bf16_t, false>), grid, dim3(TPB), 0, ST, (const bf16_t*)qkv, (bf16_t*)nullptr, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, cos_t, sin_t, pos, B, S, Hq, Hkv, D, S_cap, s0);
  else
    hipLaunchKernelGGL((rope_k<float, false>), grid, dim3(TPB), 0, ST, (const float*)qkv, (float*)nullptr, (float*)q, (float*)k, (float*)v, cos_t, sin_t, pos, B, S, Hq, Hkv, D, S_cap, s0);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_rope_split(const void* qkv, void* q, void* k, void* v, const float* cos_t, const float* sin_t,
                              const int32_t* pos, int B, int S, int Hq, int Hkv, int D, int dtype, dxa_stream_t stream) {
  return dxa_rope_split_at(qkv, q, k, v, cos_t, sin_t, pos, B, S, Hq, Hkv, D, S, 0, dtype, stream);
}
extern "C" int dxa_rope_merge_at(const void* dq, const void* dk, const void* dv, void* dqkv, const float* cos_t,
                                 const float* sin_t, const int32_t* pos, int B, int S, int Hq, int Hkv, int D, int S_cap, int s0,
                                 int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(dq && dk && dv && dqkv && cos_t && sin_t && ok_dtype(dtype), "dxa_rope_merge: bad args");
  DXA_CHECK_ARG(D % 8 == 0, "dxa_rope_merge: head_dim must be a multiple of 8 (got %d)", D);
  DXA_CHECK_ARG(s0 >= 0 && S >= 0 && s0 + S <= S_cap, "dxa_rope_merge: positions %d .. %d outside the %d the head-major tensors hold", s0, s0 + S, S_cap);
  const int64_t total = (int64_t)B * S * (Hq + 2 * Hkv) * (D / 8);
  if (total == 0) return DXA_OK;
  dim3 grid(dxa_grid1d(total, TPB));
  const bool wide = dtype == DXA_BF16 && D % 16 == 0 && al(dqkv, 16) && al(dq, 16) && al(dk, 16) && al(dv, 16);
  if (wide)
    hipLaunchKernelGGL((rope_k<bf16_t, true, 8>), dim3(dxa_grid1d(total / 2, TPB)), dim3(TPB), 0, ST, (const bf16_t*)nullptr, (bf16_t*)dqkv, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, cos_t, sin_t, pos, B, S, Hq, Hkv, D, S_cap, s0);
  else if (dtype == DXA_BF16)
    hipLaunchKernelGGL((rope_k<bf16_t, true>), grid, dim3(TPB), 0, ST, (const bf16_t*)nullptr, (bf16_t*)dqkv, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, cos_t, sin_t, pos, B, S, Hq, Hkv, D, S_cap, s0);
  else
    hipLaunchKernelGGL((rope_k<float, true>), grid, dim3(TPB), 0, ST, (const float*)nullptr, (float*)dqkv, (float*)dq, (float*)dk, (float*)dv, cos_t, sin_t, pos, B, S, Hq, Hkv, D, S_cap, s0);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_rope_merge(const void* dq, const void* dk, const void* dv, void* dqkv, const float* cos_t,
                              const float* sin_t, const int32_t* pos, int B, int S, int Hq, int Hkv, int D, int dtype,
                              dxa_stream_t stream) {
  return dxa_rope_merge_at(dq, dk, dv, dqkv, cos_t, sin_t, pos, B, S, Hq, Hkv, D, S, 0, dtype, stream);
}

extern "C" int dxa_swiglu_fwd(const void* gu, void* out, int64_t rows, int64_t F, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(gu && out && rows >= 0 && F > 0 && ok_dtype(dtype), "dxa_swiglu_fwd: bad args");
  if (rows == 0) return DXA_OK;
  const bool vec = F % 4 == 0 && al(gu, 16) && al(out, 16);
  if (dtype == DXA_BF16) {
    if (vec && F % 8 == 0) hipLaunchKernelGGL((swiglu_fwd_k<bf16_t, 8>), dim3(dxa_grid1d(rows * F / 8, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (bf16_t*)out, rows, F);
    else if (vec) hipLaunchKernelGGL((swiglu_fwd_k<bf16_t, 4>), dim3(dxa_grid1d(rows * F / 4, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (bf16_t*)out, rows, F);
    else hipLaunchKernelGGL((swiglu_fwd_k<bf16_t, 1>), dim3(dxa_grid1d(rows * F, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (bf16_t*)out, rows, F);
  } else {
    if (vec) hipLaunchKernelGGL((swiglu_fwd_k<float, 4>), dim3(dxa_grid1d(rows * F / 4, TPB)), dim3(TPB), 0, ST, (const float*)gu, (float*)out, rows, F);
    else hipLaunchKernelGGL((swiglu_fwd_k<float, 1>), dim3(dxa_grid1d(rows * F, TPB)), dim3(TPB), 0, ST, (const float*)gu, (float*)out, rows, F);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_swiglu_bwd(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t F, int dtype,
                              dxa_stream_t stream) {
  DXA_CHECK_ARG(gu && dout && dgu && rows >= 0 && F > 0 && ok_dtype(dtype), "dxa_swiglu_bwd: bad args");
  if (rows == 0) return DXA_OK;
  const bool vec = F % 4 == 0 && al(gu, 16) && al(dout, 16) && al(dgu, 16);
  if (dtype == DXA_BF16) {
    if (vec && F % 8 == 0) hipLaunchKernelGGL((swiglu_bwd_k<bf16_t, 8>), dim3(dxa_grid1d(rows * F / 8, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (const bf16_t*)dout, (bf16_t*)dgu, rows, F);
    else if (vec) hipLaunchKernelGGL((swiglu_bwd_k<bf16_t, 4>), dim3(dxa_grid1d(rows * F / 4, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (const bf16_t*)dout, (bf16_t*)dgu, rows, F);
    else hipLaunchKernelGGL((swiglu_bwd_k<bf16_t, 1>), dim3(dxa_grid1d(rows * F, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (const bf16_t*)dout, (bf16_t*)dgu, rows, F);
  } else {
    if (vec) hipLaunchKernelGGL((swiglu_bwd_k<float, 4>), dim3(dxa_grid1d(rows * F / 4, TPB)), dim3(TPB), 0, ST, (const float*)gu, (const float*)dout, (float*)dgu, rows, F);
    else hipLaunchKernelGGL((swiglu_bwd_k<float, 1>), dim3(dxa_grid1d(rows * F, TPB)), dim3(TPB), 0, ST, (const float*)gu, (const float*)dout, (float*)dgu, rows, F);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_glu_fwd(const void* gu, void* out, int64_t rows, int64_t F, int act, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(gu && out && rows >= 0 && F > 0 && ok_dtype(dtype), "dxa_glu_fwd: bad args");
  if (rows == 0) return DXA_OK;
  const bool vec = F % 4 == 0 && al(gu, 16) && al(out, 16);
  if (dtype == DXA_BF16) {
    if (vec && F % 8 == 0) hipLaunchKernelGGL((glu_fwd_k<bf16_t, 8>), dim3(dxa_grid1d(rows * F / 8, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (bf16_t*)out, rows, F, act);
    else if (vec) hipLaunchKernelGGL((glu_fwd_k<bf16_t, 4>), dim3(dxa_grid1d(rows * F / 4, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (bf16_t*)out, rows, F, act);
    else hipLaunchKernelGGL((glu_fwd_k<bf16_t, 1>), dim3(dxa_grid1d(rows * F, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (bf16_t*)out, rows, F, act);
  } else {
    if (vec) hipLaunchKernelGGL((glu_fwd_k<float, 4>), dim3(dxa_grid1d(rows * F / 4, TPB)), dim3(TPB), 0, ST, (const float*)gu, (float*)out, rows, F, act);
    else hipLaunchKernelGGL((glu_fwd_k<float, 1>), dim3(dxa_grid1d(rows * F, TPB)), dim3(TPB), 0, ST, (const float*)gu, (float*)out, rows, F, act);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_glu_bwd(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t F, int act, int dtype,
                           dxa_stream_t stream) {
  DXA_CHECK_ARG(gu && dout && dgu && rows >= 0 && F > 0 && ok_dtype(dtype), "dxa_glu_bwd: bad args");
  if (rows == 0) return DXA_OK;
  const bool vec = F % 4 == 0 && al(gu, 16) && al(dout, 16) && al(dgu, 16);
  if (dtype == DXA_BF16) {
    if (vec && F % 8 == 0) hipLaunchKernelGGL((glu_bwd_k<bf16_t, 8>), dim3(dxa_grid1d(rows * F / 8, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (const bf16_t*)dout, (bf16_t*)dgu, rows, F, act);
    else if (vec) hipLaunchKernelGGL((glu_bwd_k<bf16_t, 4>), dim3(dxa_grid1d(rows * F / 4, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (const bf16_t*)dout, (bf16_t*)dgu, rows, F, act);
    else hipLaunchKernelGGL((glu_bwd_k<bf16_t, 1>), dim3(dxa_grid1d(rows * F, TPB)), dim3(TPB), 0, ST, (const bf16_t*)gu, (const bf16_t*)dout, (bf16_t*)dgu, rows, F, act);
  } else {
    if (vec) hipLaunchKernelGGL((glu_bwd_k<float, 4>), dim3(dxa_grid1d(rows * F / 4, TPB)), dim3(TPB), 0, ST, (const float*)gu, (const float*)dout, (float*)dgu, rows, F, act);
    else hipLaunchKernelGGL((glu_bwd_k<float, 1>), dim3(dxa_grid1d(rows * F, TPB)), dim3(TPB), 0, ST, (const float*)gu, (const float*)dout, (float*)dgu, rows, F, act);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && y && n >= 0 && ok_dtype(dtype), "dxa_act_fwd: bad args");
  if (n == 0) return DXA_OK;
  const bool vec = n % 4 == 0 && al(x, 16) && al(y, 16);
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((act_fwd_k<bf16_t, 4>), dim3(dxa_grid1d(n / 4, TPB)), dim3(TPB), 0, ST, (const bf16_t*)x, (bf16_t*)y, n, act);
    else hipLaunchKernelGGL((act_fwd_k<bf16_t, 1>), dim3(dxa_grid1d(n, TPB)), dim3(TPB), 0, ST, (const bf16_t*)x, (bf16_t*)y, n, act);
  } else {
    if (vec) hipLaunchKernelGGL((act_fwd_k<float, 4>), dim3(dxa_grid1d(n / 4, TPB)), dim3(TPB), 0, ST, (const float*)x, (float*)y, n, act);
    else hipLaunchKernelGGL((act_fwd_k<float, 1>), dim3(dxa_grid1d(n, TPB)), dim3(TPB), 0, ST, (const float*)x, (float*)y, n, act);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int act, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && dy && dx && n >= 0 && ok_dtype(dtype), "dxa_act_bwd: bad args");
  if (n == 0) return DXA_OK;
  const bool vec = n % 4 == 0 && al(x, 16) && al(dy, 16) && al(dx, 16);
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((act_bwd_k<bf16_t, 4>), dim3(dxa_grid1d(n / 4, TPB)), dim3(TPB), 0, ST, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, act);
    else hipLaunchKernelGGL((act_bwd_k<bf16_t, 1>), dim3(dxa_grid1d(n, TPB)), dim3(TPB), 0, ST, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, act);
  } else {
    if (vec) hipLaunchKernelGGL((act_bwd_k<float, 4>), dim3(dxa_grid1d(n / 4, TPB)), dim3(TPB), 0, ST, (const float*)x, (const float*)dy, (float*)dx, n, act);
    else hipLaunchKernelGGL((act_bwd_k<float, 1>), dim3(dxa_grid1d(n, TPB)), dim3(TPB), 0, ST, (const float*)x, (const float*)dy, (float*)dx, n, act);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_add(const void* a, const void* b, void* out, int64_t n, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(a && b && out && n >= 0 && ok_dtype(dtype), "dxa_add: bad args");
  if (n == 0) return DXA_OK;
  const bool vec = n % 4 == 0 && al(a, 16) && al(b, 16) && al(out, 16);
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((add_k<bf16_t, 4>), dim3(dxa_grid1d(n / 4, TPB)), dim3(TPB), 0, ST, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
    else hipLaunchKernelGGL((add_k<bf16_t, 1>), dim3(dxa_grid1d(n, TPB)), dim3(TPB), 0, ST, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
  } else {
    if (vec) hipLaunchKernelGGL((add_k<float, 4>), dim3(dxa_grid1d(n / 4, TPB)), dim3(TPB), 0, ST, (const float*)a, (const float*)b, (float*)out, n);
    else hipLaunchKernelGGL((add_k<float, 1>), dim3(dxa_grid1d(n, TPB)), dim3(TPB), 0, ST, (const float*)a, (const float*)b, (float*)out, n);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
namespace {
template <typename T, int VEC, int OP>   // OP 0: alpha*a + beta*b ; 1: a*b
__global__ __launch_bounds__(TPB) void binary_k(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t n,
                                                float alpha, float beta) {
  const int64_t total = n / VEC;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    float x[VEC], y[VEC];
    Vec<T, VEC>::ld(x, a + it * VEC);
    Vec<T, VEC>::ld(y, b + it * VEC);
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] = OP == 0 ? alpha * x[i] + beta * y[i] : x[i] * y[i];
    Vec<T, VEC>::st(o + it * VEC, x);
  }
}
template <typename T>
__global__ __launch_bounds__(TPB) void mul_rows_k(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ o,
                                                  int64_t R, int64_t Nn, int64_t C) {
  const int64_t total = R * Nn * C;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t c = it % C, r = it / (Nn * C);
    stf<T>(o + it, ldf<T>(x + it) * ldf<T>(g + r * C + c));
  }
}
template <int OP>
int launch_binary(const void* a, const void* b, void* out, int64_t n, float alpha, float beta, int dtype, dxa_stream_t stream) {
  const bool vec = n % 4 == 0 && al(a, 16) && al(b, 16) && al(out, 16);
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((binary_k<bf16_t, 4, OP>), dim3(dxa_grid1d(n / 4, TPB)), dim3(TPB), 0, ST, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n, alpha, beta);
    else hipLaunchKernelGGL((binary_k<bf16_t, 1, OP>), dim3(dxa_grid1d(n, TPB)), dim3(TPB), 0, ST, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n, alpha, beta);
  } else {
    if (vec) hipLaunchKernelGGL((binary_k<float, 4, OP>), dim3(dxa_grid1d(n / 4, TPB)), dim3(TPB), 0, ST, (const float*)a, (const float*)b, (float*)out, n, alpha, beta);
    else hipLaunchKernelGGL((binary_k<float, 1, OP>), dim3(dxa_grid1d(n, TPB)), dim3(TPB), 0, ST, (const float*)a, (const float*)b, (float*)out, n, alpha, beta);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
}  // namespace
extern "C" int dxa_axpby(const void* a, const void* b, void* out, int64_t n, float alpha, float beta, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(a && b && out && n >= 0 && ok_dtype(dtype), "dxa_axpby: bad args");
  if (n == 0) return DXA_OK;
  return launch_binary<0>(a, b, out, n, alpha, beta, dtype, stream);
}
extern "C" int dxa_mul(const void* a, const void* b, void* out, int64_t n, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(a && b && out && n >= 0 && ok_dtype(dtype), "dxa_mul: bad args");
  if (n == 0) return DXA_OK;
  return launch_binary<1>(a, b, out, n, 1.f, 1.f, dtype, stream);
}
extern "C" int dxa_mul_rows(const void* x, const void* g, void* out, int64_t R, int64_t Nn, int64_t C, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && g && out && R >= 0 && Nn >= 0 && C > 0 && ok_dtype(dtype), "dxa_mul_rows: bad args");
  if (R * Nn == 0) return DXA_OK;
  if (dtype == DXA_BF16) hipLaunchKernelGGL((mul_rows_k<bf16_t>), dim3(dxa_grid1d(R * Nn * C, TPB)), dim3(TPB), 0, ST, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)out, R, Nn, C);
  else hipLaunchKernelGGL((mul_rows_k<float>), dim3(dxa_grid1d(R * Nn * C, TPB)), dim3(TPB), 0, ST, (const float*)x, (const float*)g, (float*)out, R, Nn, C);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(src && dst && n >= 0 && ok_dtype(src_dtype) && ok_dtype(dst_dtype), "dxa_cast: bad args");
  if (n == 0) return DXA_OK;
  const bool vec = n % 4 == 0 && al(src, 16) && al(dst, 16);
  const dim3 g(dxa_grid1d(vec ? n / 4 : n, TPB));
#define CASE(TS, TD)                                                                                              \
  do {                                                                                                            \
    if (vec) hipLaunchKernelGGL((cast_k<TS, TD, 4>), g, dim3(TPB), 0, ST, (const TS*)src, (TD*)dst, n);           \
    else hipLaunchKernelGGL((cast_k<TS, TD, 1>), g, dim3(TPB), 0, ST, (const TS*)src, (TD*)dst, n);               \
  } while (0)
  if (src_dtype == DXA_F32 && dst_dtype == DXA_BF16) CASE(float, bf16_t);
  else if (src_dtype == DXA_BF16 && dst_dtype == DXA_F32) CASE(bf16_t, float);
  else if (src_dtype == DXA_F32) CASE(float, float);
  else CASE(bf16_t, bf16_t);
#undef CASE
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols,
                          int64_t cols_padded, int src_dtype, int dst_dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(src && dst && rows >= 0 && cols >= 0 && cols_padded >= cols && ldd >= cols_padded && lds >= cols &&
                    ok_dtype(src_dtype) && ok_dtype(dst_dtype), "dxa_copy2d: bad args");
  if (rows * cols_padded == 0) return DXA_OK;
  const dim3 g(dxa_grid1d(rows * cols_padded, TPB));
#define CASE(TS, TD) hipLaunchKernelGGL((copy2d_k<TS, TD>), g, dim3(TPB), 0, ST, (const TS*)src, lds, (TD*)dst, ldd, rows, cols, cols_padded)
  if (src_dtype == DXA_F32 && dst_dtype == DXA_BF16) CASE(float, bf16_t);
  else if (src_dtype == DXA_BF16 && dst_dtype == DXA_F32) CASE(bf16_t, float);
  else if (src_dtype == DXA_F32) CASE(float, float);
  else CASE(bf16_t, bf16_t);
#undef CASE
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_splice_fwd(const int64_t* plan, const void* embed, const void* img, void* out, int64_t n_rows,
                              int64_t d, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(plan && embed && out && n_rows >= 0 && d > 0 && ok_dtype(dtype), "dxa_splice_fwd: bad args");
  if (n_rows == 0) return DXA_OK;
  const bool vec = d % 4 == 0 && al(embed, 16) && al(out, 16) && (!img || al(img, 16));
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((splice_fwd_k<bf16_t, 4>), dim3(dxa_grid1d(n_rows * d / 4, TPB)), dim3(TPB), 0, ST, plan, (const bf16_t*)embed, (const bf16_t*)img, (bf16_t*)out, n_rows, d);
    else hipLaunchKernelGGL((splice_fwd_k<bf16_t, 1>), dim3(dxa_grid1d(n_rows * d, TPB)), dim3(TPB), 0, ST, plan, (const bf16_t*)embed, (const bf16_t*)img, (bf16_t*)out, n_rows, d);
  } else {
    if (vec) hipLaunchKernelGGL((splice_fwd_k<float, 4>), dim3(dxa_grid1d(n_rows * d / 4, TPB)), dim3(TPB), 0, ST, plan, (const float*)embed, (const float*)img, (float*)out, n_rows, d);
    else hipLaunchKernelGGL((splice_fwd_k<float, 1>), dim3(dxa_grid1d(n_rows * d, TPB)), dim3(TPB), 0, ST, plan, (const float*)embed, (const float*)img, (float*)out, n_rows, d);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_splice_bwd(const int64_t* plan, const void* dout, float* d_embed, void* d_img, int64_t n_rows,
                              int64_t d, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(plan && dout && n_rows >= 0 && d > 0 && ok_dtype(dtype), "dxa_splice_bwd: bad args");
  DXA_CHECK_ARG(n_rows < (1ll << 31), "dxa_splice_bwd: too many rows");
  if (n_rows == 0) return DXA_OK;
  const bool vec = d % 4 == 0 && al(dout, 16) && (!d_img || al(d_img, 16));
  const dim3 g4((unsigned)n_rows, (unsigned)dxa_cdiv(d, 1024)), g1((unsigned)n_rows, (unsigned)dxa_cdiv(d, 256));
  if (dtype == DXA_BF16) {
    if (vec) hipLaunchKernelGGL((splice_bwd_k<bf16_t, 4>), g4, dim3(256), 0, ST, plan, (const bf16_t*)dout, d_embed, (bf16_t*)d_img, n_rows, d);
    else hipLaunchKernelGGL((splice_bwd_k<bf16_t, 1>), g1, dim3(256), 0, ST, plan, (const bf16_t*)dout, d_embed, (bf16_t*)d_img, n_rows, d);
  } else {
    if (vec) hipLaunchKernelGGL((splice_bwd_k<float, 4>), g4, dim3(256), 0, ST, plan, (const float*)dout, d_embed, (float*)d_img, n_rows, d);
    else hipLaunchKernelGGL((splice_bwd_k<float, 1>), g1, dim3(256), 0, ST, plan, (const float*)dout, d_embed, (float*)d_img, n_rows, d);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_zero_rows(const int64_t* plan, float* g, int64_t n_rows, int64_t d, dxa_stream_t stream) {
  DXA_CHECK_ARG(plan && g && n_rows >= 0 && d > 0 && n_rows < (1ll << 31), "dxa_zero_rows: bad args");
  if (n_rows == 0) return DXA_OK;
  hipLaunchKernelGGL(zero_rows_k, dim3((unsigned)n_rows), dim3(256), 0, ST, plan, g, n_rows, d);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_gather_rows(const void* x, const int64_t* idx, void* out, int64_t n, int64_t d, int src_dtype,
                               int dst_dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && idx && out && n >= 0 && d > 0 && ok_dtype(src_dtype) && ok_dtype(dst_dtype), "dxa_gather_rows: bad args");
  if (n == 0) return DXA_OK;
  const dim3 g(dxa_grid1d(n * d, TPB));
#define CASE(TS, TD) hipLaunchKernelGGL((gather_rows_k<TS, TD>), g, dim3(TPB), 0, ST, (const TS*)x, idx, (TD*)out, n, d)
  if (src_dtype == DXA_F32 && dst_dtype == DXA_BF16) CASE(float, bf16_t);
  else if (src_dtype == DXA_BF16 && dst_dtype == DXA_F32) CASE(bf16_t, float);
  else if (src_dtype == DXA_F32) CASE(float, float);
  else CASE(bf16_t, bf16_t);
#undef CASE
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_scatter_rows(const void* dout, const int64_t* idx, void* dx, int64_t n, int64_t R, int64_t d,
                                int src_dtype, int dst_dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(dout && idx && dx && n >= 0 && R >= 0 && d > 0 && ok_dtype(src_dtype) && ok_dtype(dst_dtype), "dxa_scatter_rows: bad args");
  if (R == 0) return DXA_OK;
  const dim3 g(dxa_grid1d(R * d, TPB));
#define CASE(TS, TD) hipLaunchKernelGGL((scatter_rows_k<TS, TD>), g, dim3(TPB), 0, ST, (const TS*)dout, idx, (TD*)dx, n, R, d)
  if (src_dtype == DXA_F32 && dst_dtype == DXA_BF16) CASE(float, bf16_t);
  else if (src_dtype == DXA_BF16 && dst_dtype == DXA_F32) CASE(bf16_t, float);
  else if (src_dtype == DXA_F32) CASE(float, float);
  else CASE(bf16_t, bf16_t);
#undef CASE
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_im2col(const void* images, void* rows, int N, int H, int W, int P, int64_t ld, int img_dtype,
                          int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(images && rows && N >= 0 && P > 0 && H % P == 0 && W % P == 0 && ld >= 3 * P * P &&
                    ok_dtype(img_dtype) && ok_dtype(dtype), "dxa_im2col: bad args");
  const int64_t total = (int64_t)N * (H / P) * (W / P) * ld;
  if (total == 0) return DXA_OK;
  const dim3 g(dxa_grid1d(total, TPB));
#define CASE(TI, T) hipLaunchKernelGGL((im2col_k<TI, T>), g, dim3(TPB), 0, ST, (const TI*)images, (T*)rows, N, H, W, P, ld)
  if (img_dtype == DXA_F32 && dtype == DXA_BF16) CASE(float, bf16_t);
  else if (img_dtype == DXA_BF16 && dtype == DXA_F32) CASE(bf16_t, float);
  else if (img_dtype == DXA_F32) CASE(float, float);
  else CASE(bf16_t, bf16_t);
#undef CASE
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_vit_embed_fwd(const void* patch, const void* cls, const void* pos, void* x, int N, int np, int C,
                                 int dtype, int w_dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(patch && cls && pos && x && ok_dtype(dtype) && ok_dtype(w_dtype) && !(dtype == DXA_F32 && w_dtype == DXA_BF16),
                "dxa_vit_embed_fwd: bad args");
  const int64_t total = (int64_t)N * (np + 1) * C;
  if (total == 0) return DXA_OK;
  const dim3 g(dxa_grid1d(total, TPB));
  if (dtype == DXA_BF16 && w_dtype == DXA_BF16)
    hipLaunchKernelGGL((vit_embed_fwd_k<bf16_t, bf16_t>), g, dim3(TPB), 0, ST, (const bf16_t*)patch, (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)x, N, np, C);
  else if (dtype == DXA_BF16)
    hipLaunchKernelGGL((vit_embed_fwd_k<bf16_t, float>), g, dim3(TPB), 0, ST, (const bf16_t*)patch, (const float*)cls, (const float*)pos, (bf16_t*)x, N, np, C);
  else
    hipLaunchKernelGGL((vit_embed_fwd_k<float, float>), g, dim3(TPB), 0, ST, (const float*)patch, (const float*)cls, (const float*)pos, (float*)x, N, np, C);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_vit_embed_bwd(const void* dx, void* dpatch, int N, int np, int C, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(dx && dpatch && ok_dtype(dtype), "dxa_vit_embed_bwd: bad args");
  const int64_t total = (int64_t)N * np * C;
  if (total == 0) return DXA_OK;
  const dim3 g(dxa_grid1d(total, TPB));
  if (dtype == DXA_BF16) hipLaunchKernelGGL((vit_embed_bwd_k<bf16_t>), g, dim3(TPB), 0, ST, (const bf16_t*)dx, (bf16_t*)dpatch, N, np, C);
  else hipLaunchKernelGGL((vit_embed_bwd_k<float>), g, dim3(TPB), 0, ST, (const float*)dx, (float*)dpatch, N, np, C);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_qsample(const float* x0, const float* noise, const float* a, const float* s, float* xt, int64_t N,
                           int64_t per, dxa_stream_t stream) {
  DXA_CHECK_ARG(x0 && noise && a && s && xt && N >= 0 && per > 0, "dxa_qsample: bad args");
  if (N == 0) return DXA_OK;
  hipLaunchKernelGGL(qsample_k, dim3(dxa_grid1d(N * per, TPB)), dim3(TPB), 0, ST, x0, noise, a, s, xt, N, per);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_timestep_embedding(const float* t, const float* freqs, float* out, int64_t N, int half,
                                      dxa_stream_t stream) {
  DXA_CHECK_ARG(t && freqs && out && N >= 0 && half > 0, "dxa_timestep_embedding: bad args");
  if (N == 0) return DXA_OK;
  hipLaunchKernelGGL(timestep_embedding_k, dim3(dxa_grid1d(N * half, TPB)), dim3(TPB), 0, ST, t, freqs, out, N, half);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_dit_assemble_fwd(const float* xe, const float* te, const float* ze, const float* pos, float* h, int N,
                                    int T, int hd, dxa_stream_t stream) {
  DXA_CHECK_ARG(xe && te && ze && pos && h, "dxa_dit_assemble_fwd: bad args");
  const int64_t total = (int64_t)N * (T + 1) * hd;
  if (total == 0) return DXA_OK;
  hipLaunchKernelGGL(dit_assemble_fwd_k, dim3(dxa_grid1d(total, TPB)), dim3(TPB), 0, ST, xe, te, ze, pos, h, N, T, hd);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_dit_assemble_bwd(const float* dh, float* dxe, float* dc, int N, int T, int hd, dxa_stream_t stream) {
  DXA_CHECK_ARG(dh && dxe && dc, "dxa_dit_assemble_bwd: bad args");
  const int64_t total = (int64_t)N * (T + 1) * hd;
  if (total == 0) return DXA_OK;
  hipLaunchKernelGGL(dit_assemble_bwd_k, dim3(dxa_grid1d(total, TPB)), dim3(TPB), 0, ST, dh, dxe, dc, N, T, hd);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_token_drop(const float* z, const float* uncond, const uint8_t* drop, float* out, int64_t N, int64_t d,
                              dxa_stream_t stream) {
  DXA_CHECK_ARG(z && uncond && drop && out && N >= 0 && d > 0, "dxa_token_drop: bad args");
  if (N == 0) return DXA_OK;
  hipLaunchKernelGGL(token_drop_k, dim3(dxa_grid1d(N * d, TPB)), dim3(TPB), 0, ST, z, uncond, drop, out, N, d);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_token_drop_bwd(const float* dout, const uint8_t* drop, float* dz, float* duncond, int64_t N, int64_t d,
                                  int accumulate, dxa_stream_t stream) {
  DXA_CHECK_ARG(dout && drop && N >= 0 && d > 0, "dxa_token_drop_bwd: bad args");
  if (N == 0) return DXA_OK;
  hipLaunchKernelGGL(token_drop_bwd_k, dim3((unsigned)((d + TPB - 1) / TPB)), dim3(TPB), 0, ST, dout, drop, dz, duncond, N, d, accumulate);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_mse_loss(const float* pred, const float* target, float* loss, float* dpred, int64_t n, float gscale,
                            dxa_stream_t stream) {
  DXA_CHECK_ARG(pred && target && loss && n > 0, "dxa_mse_loss: bad args");
  hipLaunchKernelGGL(mse_loss_k, dim3(1), dim3(1024), 0, ST, pred, target, loss, dpred, n, gscale);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_mse_loss_rows(const float* pred, const float* target, const float* row_w, float* loss, float* dpred,
                                 int64_t rows, int64_t cols, float gscale, dxa_stream_t stream) {
  DXA_CHECK_ARG(pred && target && row_w && loss && rows > 0 && cols > 0, "dxa_mse_loss_rows: bad args");
  hipLaunchKernelGGL(mse_loss_rows_k, dim3(1), dim3(1024), 0, ST, pred, target, row_w, loss, dpred, rows, cols, gscale);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_ddim_step(float* x, const float* model_out, int64_t B, int64_t per, int use_cfg, float cfg_scale,
                             float c_recip, float c_recipm1, float ab_prev, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && model_out && B >= 0 && per > 0, "dxa_ddim_step: bad args");
  if (B == 0) return DXA_OK;
  hipLaunchKernelGGL(ddim_step_k, dim3(dxa_grid1d(B * per, TPB)), dim3(TPB), 0, ST, x, model_out, B, per, use_cfg, cfg_scale, c_recip, c_recipm1, ab_prev);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

// ------------------------------------------------------------------------------------ transpose / permute
// dst[c, r] = src[r, c] for r < R, c < C; dst columns R..Rp-1 are zero-filled (padding the contraction
// dimension of the dW GEMMs to a multiple of the K slab).  64x64 tiles through LDS, 16-byte global accesses
// on both sides (bf16: 8 elements), rows of the LDS tile padded by 2 elements against bank conflicts.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void transpose_k(const T* __restrict__ src, int64_t lds_, T* __restrict__ dst, int64_t ldd,
                                                   int64_t R, int64_t C, int64_t Rp, int vec) {
  constexpr int TS = 64;
  constexpr int EPV = 16 / sizeof(T);          // elements per 16-byte vector
  __shared__ T tile[TS][TS + 2];
  const int64_t r0 = (int64_t)blockIdx.y * TS, c0 = (int64_t)blockIdx.x * TS;
  const int tid = threadIdx.x;
  constexpr int VPR = TS / EPV;                 // vectors per tile row
  for (int v = tid; v < TS * VPR; v += 256) {
    const int rr = v / VPR, cc = (v % VPR) * EPV;
    const int64_t r = r0 + rr, c = c0 + cc;
    alignas(16) T tmp[EPV];
    if (r < R && vec && c + EPV <= C) {
      *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(src + r * lds_ + c);
    } else {
#pragma unroll
      for (int e = 0; e < EPV; ++e) tmp[e] = (r < R && c + e < C) ? src[r * lds_ + c + e] : (T)0;
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) tile[rr][cc + e] = tmp[e];
  }
  __syncthreads();
  for (int v = tid; v < TS * VPR; v += 256) {
    const int cc = v / VPR, rr = (v % VPR) * EPV;   // dst row = src col cc, dst cols = src rows rr..rr+EPV-1
    const int64_t c = c0 + cc, r = r0 + rr;
    if (c >= C || r >= Rp) continue;
    alignas(16) T tmp[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) tmp[e] = tile[rr + e][cc];   // rows >= R were loaded as zeros
    if (vec && r + EPV <= Rp) {
      *reinterpret_cast<uint4*>(dst + c * ldd + r) = *reinterpret_cast<const uint4*>(tmp);
    } else {
#pragma unroll
      for (int e = 0; e < EPV; ++e) if (r + e < Rp) dst[c * ldd + r + e] = tmp[e];
    }
  }
}
// bf16 fast path: no LDS.  A lane owns an 8x8 block: eight 16-byte row loads (a wave instruction covers 8 rows x
// 128 B, whole cache lines), an in-register 16-bit transpose with v_perm_b32, eight 16-byte stores (8 dst rows x
// 128 B per wave instruction).  Workgroup = 128 x 128 (waves 2 x 2): 256-B runs on both sides, 8 independent
// loads in flight per lane.  Needs 16-byte aligned rows, C % 8 == 0 and R_padded % 8 == 0.
__global__ __launch_bounds__(256) void transpose8_bf16_k(const bf16_t* __restrict__ src, int64_t lds_, bf16_t* __restrict__ dst,
                                                         int64_t ldd, int64_t R, int64_t C, int64_t Rp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * 128 + (wave >> 1) * 64 + (lane >> 3) * 8;
  const int64_t c0 = (int64_t)blockIdx.x * 128 + (wave & 1) * 64 + (lane & 7) * 8;
  if (c0 >= C || r0 >= Rp) return;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  u4 in[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    in[i] = (u4){0u, 0u, 0u, 0u};
    if (r0 + i < R) in[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + (r0 + i) * lds_ + c0));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      o[d] = __builtin_amdgcn_perm(in[2 * d + 1][j >> 1], in[2 * d][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
    __builtin_nontemporal_store(o, reinterpret_cast<u4*>(dst + (c0 + j) * ldd + r0));
  }
}
// [B,S,H,D] -> [B,H,S,D] (to_head=1) or back (to_head=0); 16-byte pieces
template <typename T>
__global__ __launch_bounds__(256) void permute_bshd_k(const T* __restrict__ src, T* __restrict__ dst, int B, int S, int H, int D,
                                                      int to_head) {
  constexpr int EPV = 16 / sizeof(T);
  const int dv = D / EPV;
  const int64_t total = (int64_t)B * S * H * dv;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int v = (int)(it % dv);
    const int h = (int)((it / dv) % H);
    const int s = (int)((it / ((int64_t)dv * H)) % S);
    const int b = (int)(it / ((int64_t)dv * H * S));
    const int64_t tok = (((int64_t)b * S + s) * H + h) * D + v * EPV;
    const int64_t head = (((int64_t)b * H + h) * S + s) * D + v * EPV;
    if (to_head) *reinterpret_cast<uint4*>(dst + head) = *reinterpret_cast<const uint4*>(src + tok);
    else *reinterpret_cast<uint4*>(dst + tok) = *reinterpret_cast<const uint4*>(src + head);
  }
}
}  // namespace

extern "C" int dxa_transpose(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t R, int64_t C, int64_t R_padded,
                             int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(src && dst && R >= 0 && C >= 0 && R_padded >= R && ld_src >= C && ld_dst >= R_padded && ok_dtype(dtype),
                "dxa_transpose: bad args");
  if (R_padded == 0 || C == 0) return DXA_OK;
  const size_t es = dtype == DXA_BF16 ? 2 : 4;
  const int64_t epv = 16 / es;
  const int vec = al(src, 16) && al(dst, 16) && ld_src % epv == 0 && ld_dst % epv == 0;
  if (dtype == DXA_BF16 && vec && C % 8 == 0 && R_padded % 8 == 0) {
    dim3 g8((unsigned)((C + 127) / 128), (unsigned)((R_padded + 127) / 128));
    DXA_CHECK_ARG(g8.y <= 65535, "dxa_transpose: too many row tiles");
    hipLaunchKernelGGL(transpose8_bf16_k, g8, dim3(256), 0, ST, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, R, C, R_padded);
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R_padded + 63) / 64));
  DXA_CHECK_ARG(grid.y <= 65535, "dxa_transpose: too many row tiles");
  if (dtype == DXA_BF16)
    hipLaunchKernelGGL((transpose_k<bf16_t>), grid, dim3(256), 0, ST, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, R, C, R_padded, vec);
  else
    hipLaunchKernelGGL((transpose_k<float>), grid, dim3(256), 0, ST, (const float*)src, ld_src, (float*)dst, ld_dst, R, C, R_padded, vec);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_permute_bshd(const void* src, void* dst, int B, int S, int H, int D, int to_head, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(src && dst && ok_dtype(dtype), "dxa_permute_bshd: bad args");
  const size_t es = dtype == DXA_BF16 ? 2 : 4;
  DXA_CHECK_ARG(D % (int)(16 / es) == 0 && al(src, 16) && al(dst, 16), "dxa_permute_bshd: head_dim must be a multiple of %d", (int)(16 / es));
  const int64_t total = (int64_t)B * S * H * (D / (int)(16 / es));
  if (total == 0) return DXA_OK;
  if (dtype == DXA_BF16)
    hipLaunchKernelGGL((permute_bshd_k<bf16_t>), dim3(dxa_grid1d(total, 256)), dim3(256), 0, ST, (const bf16_t*)src, (bf16_t*)dst, B, S, H, D, to_head);
  else
    hipLaunchKernelGGL((permute_bshd_k<float>), dim3(dxa_grid1d(total, 256)), dim3(256), 0, ST, (const float*)src, (float*)dst, B, S, H, D, to_head);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
