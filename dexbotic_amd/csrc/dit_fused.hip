// Persistent DiT-block kernel for the action sampler (SURVEY.md §8a rows A6/A8 at inference time).
//
// One DDIM step of the reference's sampler (dexbotic/model/cogact/action_model/dit.py:137-162, 281-311 called from
// diffusion.py ddim_sample_loop, cogact_arch.py:186-197) walks `depth` DiTBlocks over 2 x 17 rows: per block a
// LayerNorm, the qkv linear, a 17-token attention, the output linear, a LayerNorm and the two MLP linears.  As
// separate launches that is ~100 kernels of 4-14 us per step and 10 steps per action chunk, almost all of it launch
// and drain latency.  Here ONE launch runs every block: a grid of co-resident workgroups walks the phases
//     [row statistics] qkv = LN(h) Wqkv^T + b | attention | h += o Wproj^T + b |
//     [row statistics] a = gelu_tanh(LN(h) W1^T + b1) | h += a W2^T + b2
// with a device-wide barrier between them (one agent-scope counter; release/acquire fences make the few hundred KB of
// activations visible across the 8 XCD L2s).  The products are the skinny fp32 decomposition of gemm.hip: a
// workgroup owns 16 output columns, its 8 waves split K in 64-deep blocks straight from L2/HBM into exact fp32
// MFMA (v_mfma_f32_16x16x4_f32), partials folded through LDS.  LayerNorm is applied on the fly to the A operand.
// fp32 throughout, same arithmetic as the unfused kernels (held to them by tests/test_kernels_gpu.py).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MAXM = 48;     // rows (CFG batch 2 x 17 tokens = 34)
constexpr int MB = 3;        // 16-row blocks
constexpr int MAXT = 32;     // tokens per sample
constexpr int HD = 64;       // head width

struct DitP {
  float *h, *qkv, *o, *a;
  const float* const* w;     // [depth][8]: qkv_w, qkv_b, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b
  unsigned* bar;
  int M, N, T1, H, heads, I, depth;
  float eps, scale;
  int dbg;   // tuning aid: 1 = barriers only, 2 = work only (wrong results)
};

struct Smem {
  float red[8][MB][64][4];
  float mu[MAXM], rs[MAXM];
  float q[MAXT][HD], k[MAXT][HD + 1], v[MAXT][HD], p[MAXT][MAXT + 1];
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// Activations travel between workgroups on different XCDs, whose L2s are not coherent with each other: every
// activation access is an agent-scope (sc1) buffer access — stores write through, loads miss in the private caches —
// so the barrier needs no cache-wide write-back / invalidate and the weights stay cached.
constexpr int SC1 = 16;
struct Act {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ Act(float* p, size_t floats)
      : r(__builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(floats * sizeof(float)), 0x00020000)) {}
  __device__ __forceinline__ float4 ld4(size_t idx) const {
#if defined(DXA_DIT_SC1_LOADS)
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * 4), 0, SC1);
#else
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * 4), 0, 0);   // cached; the barrier invalidates
#endif
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  }
  __device__ __forceinline__ float ld1(size_t idx) const {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(idx * 4), 0, SC1));
  }
  __device__ __forceinline__ void st4(size_t idx, const float4& v) const {
    const u32x4_t u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)(idx * 4), 0, SC1);
  }
  __device__ __forceinline__ void st1(size_t idx, float v) const {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)(idx * 4), 0, SC1);
  }
};

__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned nblk, unsigned& epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += 1;
    const unsigned target = epoch * nblk;
    const unsigned arrived = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived != target)
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
#if !defined(DXA_DIT_SC1_LOADS)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // drop stale activation lines from this CU's L1 / this XCD's L2
#endif
  }
  __syncthreads();
}

// per-row mean / rstd of h: a wave owns rows wave, wave+8, ...; the row is read ONCE into registers (H <= 1024)
__device__ __forceinline__ void row_stats(const DitP& p, Smem& s, const Act& h) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int MAXR = 6;                                   // rows per wave (48 / 8)
  constexpr int MAXV = 4;                                   // float4 per lane per row (H <= 1024)
  float4 x[MAXR][MAXV];
  const int nv = (p.H + 255) / 256;
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int m = wave + 8 * r;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane * 4 + 256 * i;
      x[r][i] = (m < p.M && i < nv && c < p.H) ? h.ld4((size_t)m * p.H + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int m = wave + 8 * r;
    if (m >= p.M) break;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) sum += x[r][i].x + x[r][i].y + x[r][i].z + x[r][i].w;
    const float mean = wave_sum(sum) / (float)p.H;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane * 4 + 256 * i;
      if (i < nv && c < p.H) {
        const float d0 = x[r][i].x - mean, d1 = x[r][i].y - mean, d2 = x[r][i].z - mean, d3 = x[r][i].w - mean;
        ss += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)p.H + p.eps);
    if (lane == 0) { s.mu[m] = mean; s.rs[m] = rstd; }
  }
  __syncthreads();
}

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESADD = 2 };

// C[M, Nout] = epi(pro(A)[M, K] W[Nout, K]^T + bias); LN: pro(a) = (a - mu[m]) * rs[m]; RESADD: C += (in place).
// A and C are activations (sc1 accesses), W and bias are weights (ordinary cached loads).
template <bool LN, int EPI>
__device__ __forceinline__ void gemm_phase(const DitP& p, Smem& s, const Act& A, int lda, const float* __restrict__ W,
                                           const float* __restrict__ bias, const Act& C, int ldc, int Nout, int K) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lg = lane >> 4;
  size_t arow[MB];
  float mu[MB], rs[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = min(mb * 16 + l16, p.M - 1);
    arow[mb] = (size_t)m * lda + 4 * lg;
    mu[mb] = LN ? s.mu[m] : 0.f;
    rs[mb] = LN ? s.rs[m] : 1.f;
  }
  const int nkb = K / 64;
  for (int cb = blockIdx.x; cb * 16 < Nout; cb += gridDim.x) {
    const int n0 = cb * 16;
    const float* Wp = W + (size_t)(n0 + l16) * K + 4 * lg;
    // epilogue operands of the folding waves are requested first: they are back long before the K loop ends
    const int em = wave * 16 + l16, en = n0 + 4 * lg;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave < MB && em < p.M) {
      b4 = *reinterpret_cast<const float4*>(bias + en);
      if (EPI == EPI_RESADD) c4 = C.ld4((size_t)em * ldc + en);
    }
    f32x4_t acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // two 64-deep K blocks per trip: 32 independent 16-byte loads per lane in flight
    for (int kb = wave; kb < nkb; kb += 16) {
      const bool two = kb + 8 < nkb;
      float4 wv[2][4], av[2][MB][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k0 = (kb + 8 * u) * 64;
        if (u == 0 || two) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wv[u][j] = *reinterpret_cast<const float4*>(Wp + k0 + 16 * j);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) av[u][mb][j] = A.ld4(arow[mb] + k0 + 16 * j);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            float4 t = av[u][mb][j];
            if (LN) {
              t.x = (t.x - mu[mb]) * rs[mb]; t.y = (t.y - mu[mb]) * rs[mb];
              t.z = (t.z - mu[mb]) * rs[mb]; t.w = (t.w - mu[mb]) * rs[mb];
            }
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][j].x, t.x, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][j].y, t.y, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][j].z, t.z, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][j].w, t.w, acc[mb], 0, 0, 0);
          }
      }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      *reinterpret_cast<float4*>(s.red[wave][mb][lane]) = make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
    __syncthreads();
    if (wave < MB) {
      // wave mb folds the 8 partials of row block mb: lane holds out[m = 16 mb + l16][n0 + 4 lg + {0..3}]
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) {
        const float4 v = *reinterpret_cast<const float4*>(s.red[w8][wave][lane]);
        r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
      }
      if (em < p.M) {
        r.x += b4.x; r.y += b4.y; r.z += b4.z; r.w += b4.w;
        if (EPI == EPI_GELU) {
          r.x = act_fwd(DXA_ACT_GELU_TANH, r.x); r.y = act_fwd(DXA_ACT_GELU_TANH, r.y);
          r.z = act_fwd(DXA_ACT_GELU_TANH, r.z); r.w = act_fwd(DXA_ACT_GELU_TANH, r.w);
        } else if (EPI == EPI_RESADD) {
          r.x += c4.x; r.y += c4.y; r.z += c4.z; r.w += c4.w;
        }
        C.st4((size_t)em * ldc + en, r);
      }
    }
    __syncthreads();
  }
}

// one workgroup per (sample, head): 17 x 17 scores from LDS copies of q, k, v
__device__ __forceinline__ void attention_phase(const DitP& p, Smem& s, const Act& qkv, const Act& o) {
  const int tid = threadIdx.x, T1 = p.T1, ld = 3 * p.H;
  for (int pr = blockIdx.x; pr < p.N * p.heads; pr += gridDim.x) {
    const int n = pr / p.heads, hd = pr - n * p.heads;
    const size_t base = (size_t)n * T1 * ld + hd * HD;
    for (int e = tid; e < T1 * (HD / 4); e += 512) {
      const int i = e / (HD / 4), d = (e - i * (HD / 4)) * 4;
      const size_t r = base + (size_t)i * ld + d;
      const float4 q4 = qkv.ld4(r), k4 = qkv.ld4(r + p.H), v4 = qkv.ld4(r + 2 * p.H);
      *reinterpret_cast<float4*>(&s.q[i][d]) = q4;
      s.k[i][d] = k4.x; s.k[i][d + 1] = k4.y; s.k[i][d + 2] = k4.z; s.k[i][d + 3] = k4.w;
      *reinterpret_cast<float4*>(&s.v[i][d]) = v4;
    }
    __syncthreads();
    for (int e = tid; e < T1 * T1; e += 512) {
      const int i = e / T1, j = e - i * T1;
      float acc = 0.f;
#pragma unroll 16
      for (int d = 0; d < HD; ++d) acc += s.q[i][d] * s.k[j][d];
      s.p[i][j] = acc * p.scale;
    }
    __syncthreads();
    if (tid < T1) {
      float mx = -INFINITY;
      for (int j = 0; j < T1; ++j) mx = fmaxf(mx, s.p[tid][j]);
      float sum = 0.f;
      for (int j = 0; j < T1; ++j) { const float e = expf(s.p[tid][j] - mx); s.p[tid][j] = e; sum += e; }
      const float inv = 1.f / sum;
      for (int j = 0; j < T1; ++j) s.p[tid][j] *= inv;
    }
    __syncthreads();
    const size_t ob = (size_t)n * T1 * p.H + hd * HD;
    for (int e = tid; e < T1 * HD; e += 512) {
      const int i = e / HD, d = e - i * HD;
      float acc = 0.f;
      for (int j = 0; j < T1; ++j) acc += s.p[i][j] * s.v[j][d];
      o.st1(ob + (size_t)i * p.H + d, acc);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(512) void dit_blocks_fused_k(const DitP p) {
  __shared__ Smem s;
  unsigned epoch = 0;
  const unsigned nblk = gridDim.x;
  const Act h(p.h, (size_t)p.M * p.H), qkv(p.qkv, (size_t)p.M * 3 * p.H), o(p.o, (size_t)p.M * p.H), a(p.a, (size_t)p.M * p.I);
  const bool work = p.dbg != 1, sync = p.dbg != 2;
  for (int blk = 0; blk < p.depth; ++blk) {
    const float* const* w = p.w + blk * 8;
    if (work) row_stats(p, s, h);
    if (work) gemm_phase<true, EPI_BIAS>(p, s, h, p.H, w[0], w[1], qkv, 3 * p.H, 3 * p.H, p.H);
    if (sync) grid_sync(p.bar, nblk, epoch);
    if (work) attention_phase(p, s, qkv, o);
    if (sync) grid_sync(p.bar, nblk, epoch);
    if (work) gemm_phase<false, EPI_RESADD>(p, s, o, p.H, w[2], w[3], h, p.H, p.H, p.H);
    if (sync) grid_sync(p.bar, nblk, epoch);
    if (work) row_stats(p, s, h);
    if (work) gemm_phase<true, EPI_GELU>(p, s, h, p.H, w[4], w[5], a, p.I, p.I, p.H);
    if (sync) grid_sync(p.bar, nblk, epoch);
    if (work) gemm_phase<false, EPI_RESADD>(p, s, a, p.I, w[6], w[7], h, p.H, p.H, p.I);
    if (sync) grid_sync(p.bar, nblk, epoch);
  }
}

}  // namespace

extern "C" size_t dxa_dit_blocks_workspace(int M, int H, int I) {
  // qkv [M,3H] + o [M,H] + a [M,I] floats + the barrier counter (kept 256-byte aligned)
  if (M <= 0 || H <= 0 || I <= 0) return 0;
  return ((size_t)M * (4 * (size_t)H + I) * sizeof(float) + 255) / 256 * 256 + 256;
}

extern "C" int dxa_dit_blocks_fwd(float* h, const float* const* weights, int depth, int N, int T1, int H, int heads, int I,
                                  float eps, void* workspace, size_t workspace_bytes, dxa_stream_t stream) {
  DXA_CHECK_ARG(h && weights && workspace, "dxa_dit_blocks_fwd: null buffer");
  DXA_CHECK_ARG(depth >= 1 && N >= 1 && T1 >= 1 && heads >= 1, "dxa_dit_blocks_fwd: bad sizes");
  const int M = N * T1;
  DXA_CHECK_ARG(M <= MAXM && T1 <= MAXT, "dxa_dit_blocks_fwd: at most %d rows / %d tokens per sample (got %d / %d)", MAXM,
                MAXT, M, T1);
  DXA_CHECK_ARG(H % 64 == 0 && I % 64 == 0 && H == heads * HD && H <= 1024, "dxa_dit_blocks_fwd: needs head width 64 and H, I %% 64 == 0");
  DXA_CHECK_ARG(workspace_bytes >= dxa_dit_blocks_workspace(M, H, I), "dxa_dit_blocks_fwd: workspace too small");
  DXA_CHECK_ARG((reinterpret_cast<uintptr_t>(h) % 16) == 0 && (reinterpret_cast<uintptr_t>(workspace) % 16) == 0,
                "dxa_dit_blocks_fwd: buffers must be 16-byte aligned");
  DitP p;
  p.h = h;
  p.qkv = reinterpret_cast<float*>(workspace);
  p.o = p.qkv + (size_t)M * 3 * H;
  p.a = p.o + (size_t)M * H;
  const size_t act_bytes = ((size_t)M * (4 * (size_t)H + I) * sizeof(float) + 255) / 256 * 256;
  p.bar = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + act_bytes);
  p.w = weights;
  p.M = M; p.N = N; p.T1 = T1; p.H = H; p.heads = heads; p.I = I; p.depth = depth;
  p.eps = eps;
  p.scale = 1.f / sqrtf((float)HD);
  static const int dbg = getenv("DXA_DIT_DBG") ? atoi(getenv("DXA_DIT_DBG")) : 0;
  p.dbg = dbg;
  hipStream_t st = (hipStream_t)stream;
  DXA_CHECK_HIP(hipMemsetAsync(p.bar, 0, sizeof(unsigned), st));
  // every workgroup must be resident at once (device-wide barrier): one per 16 columns of the widest product,
  // never more than the 256 CUs can hold
  int grid = I / 16;
  if (3 * H / 16 > grid) grid = 3 * H / 16;
  if (grid > 240) grid = 240;
  static const int grid_cap = getenv("DXA_DIT_GRID") ? atoi(getenv("DXA_DIT_GRID")) : 0;   // tuning aid
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  hipLaunchKernelGGL(dit_blocks_fused_k, dim3(grid), dim3(512), 0, st, p);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
