// Device-side pieces of the MemVLA memory path (dexbotic/model/memvla/memvla_arch.py:82-127, 195-411)
// that used to run as torch glue (rand / compare / cast / divide per dropout mask; cosine_similarity + .item() per merge):
//   dxa_dropout_mask       one launch draws a whole dropout mask (entries 0 | 1/(1-p)) from a counter-based generator
//   dxa_bank_consolidate   the token-merge consolidation of one memory bank ENTIRELY on the device: cosine similarity of
//                          every neighbouring pair of entries, arg-max, merge of that pair, compaction of entries and
//                          timesteps — no host read-back, the bank's length evolves deterministically (host bookkeeping)
#include "common.h"

namespace {

constexpr int TPB = 256;

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3) x key (k0, k1) -> 4 x 32 random bits
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// mask[i] = u_i >= p ? 1 / (1 - p) : 0 with u_i uniform in [0, 1): nn.Dropout / SDPA dropout_p semantics (keep probability
// 1 - p, survivors scaled); thread t draws elements 4 t .. 4 t + 3 from counter (t, offset)
template <typename T>
__global__ __launch_bounds__(TPB) void dropout_mask_k(T* __restrict__ out, int64_t n, float p, float keep_scale, uint32_t seed_lo,
                                                      uint32_t seed_hi, uint32_t off_lo, uint32_t off_hi) {
  const int64_t quads = (n + 3) / 4;
  for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < quads; t += (int64_t)gridDim.x * TPB) {
    uint32_t c[4] = {(uint32_t)t, (uint32_t)(t >> 32), off_lo, off_hi};
    philox4x32_10(c, seed_lo, seed_hi);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = 4 * t + e;
      if (i < n) {
        const float u = (float)(c[e] >> 8) * (1.0f / 16777216.0f);      // 24 random bits: exactly representable, < 1
        stf<T>(out + i, u >= p ? keep_scale : 0.f);
      }
    }
  }
}

// sims[i] = mean over the N tokens of cos(feat[i, n, :], feat[i + 1, n, :]), i < len - 1 (F.cosine_similarity, eps 1e-8 on each
// norm, memvla_arch.py:263-275).  One workgroup per pair; a wave owns tokens wave, wave + 4, ...; fixed fold order.
template <typename T>
__global__ __launch_bounds__(TPB) void bank_sims_k(const T* __restrict__ feat, float* __restrict__ sims, int64_t N, int64_t D) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* a = feat + (int64_t)blockIdx.x * N * D;
  const T* b = a + N * D;
  float acc = 0.f;
  for (int64_t n = wave; n < N; n += 4) {
    float dot = 0.f, na = 0.f, nb = 0.f;
    for (int64_t d = lane; d < D; d += 64) {
      const float x = ldf<T>(a + n * D + d), y = ldf<T>(b + n * D + d);
      dot += x * y; na += x * x; nb += y * y;
    }
    dot = wave_sum(dot); na = wave_sum(na); nb = wave_sum(nb);
    acc += dot / (fmaxf(sqrtf(na), 1e-8f) * fmaxf(sqrtf(nb), 1e-8f));
  }
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sims[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)N;
}

// j = first arg-max of sims[0 .. len - 2] (numpy.argmax); entry j <- 0.5 (entry j + entry j + 1), entries j + 2 .. move down by
// one, same for the timesteps (memvla_arch.py:276-287).  Element e of every entry is handled by ONE thread walking the
// entries in ascending order (reads entry i + 1 before it writes entry i): in place, no barrier needed.
// fifo = 1: the oldest entry is dropped instead (memvla_arch.py:300-303): every entry moves down by one, sims is not read.
template <typename T>
__global__ __launch_bounds__(TPB) void bank_merge_k(T* __restrict__ feat, float* __restrict__ ts, const float* __restrict__ sims,
                                                    int len, int64_t E, int fifo) {
  int j = -1;
  if (!fifo) {
    j = 0;
    float best = sims[0];
    for (int i = 1; i < len - 1; ++i) {
      const float s = sims[i];
      if (s > best) { best = s; j = i; }
    }
  }
  for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < E; e += (int64_t)gridDim.x * TPB) {
    // the merged value rounded through the storage type, as 0.5 * (f_j + f_j+1) is stored by the reference
    if (j >= 0)
      stf<T>(feat + (int64_t)j * E + e, 0.5f * ldf<T>(feat + (int64_t)j * E + e) + 0.5f * ldf<T>(feat + (int64_t)(j + 1) * E + e));
    for (int i = j + 1; i < len - 1; ++i) feat[(int64_t)i * E + e] = feat[(int64_t)(i + 1) * E + e];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (j >= 0) ts[j] = 0.5f * (ts[j] + ts[j + 1]);
    for (int i = j + 1; i < len - 1; ++i) ts[i] = ts[i + 1];
  }
}

// out[r, n, c] = (x ? x[r, n, c] : 0) + alpha * g[r, c]: the timestep positional embedding added to every token of a bank entry
// (memvla_arch.py:352-360), and the broadcast of a per-entry row over the tokens (x = null: backward of the token mean)
template <typename T>
__global__ __launch_bounds__(TPB) void add_rows_k(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ o, int64_t R,
                                                  int64_t Nn, int64_t C, float alpha) {
  const int64_t total = R * Nn * C;
  for (int64_t it = (int64_t)blockIdx.x * TPB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TPB) {
    const int64_t c = it % C, r = it / (Nn * C);
    stf<T>(o + it, (x ? ldf<T>(x + it) : 0.f) + alpha * ldf<T>(g + r * C + c));
  }
}
// out[r, c] = scale * sum_n x[r, n, c] (fp32 accumulation in token order: deterministic); grid (column tiles, R)
template <typename T>
__global__ __launch_bounds__(TPB) void token_sum_k(const T* __restrict__ x, T* __restrict__ o, int64_t Nn, int64_t C, float scale) {
  const int64_t c = (int64_t)blockIdx.x * TPB + threadIdx.x, r = blockIdx.y;
  if (c >= C) return;
  const T* p = x + r * Nn * C + c;
  float s = 0.f;
  for (int64_t n = 0; n < Nn; ++n) s += ldf<T>(p + n * C);
  stf<T>(o + r * C + c, s * scale);
}

}  // namespace

extern "C" int dxa_add_rows(const void* x, const void* g, void* out, int64_t R, int64_t Nn, int64_t C, float alpha, int dtype,
                            dxa_stream_t stream) {
  DXA_CHECK_ARG(g && out && R >= 0 && Nn >= 0 && C > 0 && (dtype == DXA_F32 || dtype == DXA_BF16), "dxa_add_rows: bad args");
  if (R * Nn == 0) return DXA_OK;
  const dim3 grid(dxa_grid1d(R * Nn * C, TPB));
  if (dtype == DXA_BF16)
    hipLaunchKernelGGL((add_rows_k<bf16_t>), grid, dim3(TPB), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)out, R, Nn, C, alpha);
  else
    hipLaunchKernelGGL((add_rows_k<float>), grid, dim3(TPB), 0, (hipStream_t)stream, (const float*)x, (const float*)g, (float*)out, R, Nn, C, alpha);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_token_sum(const void* x, void* out, int64_t R, int64_t Nn, int64_t C, float scale, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && out && R >= 0 && Nn >= 0 && C > 0 && R <= 65535 && (dtype == DXA_F32 || dtype == DXA_BF16), "dxa_token_sum: bad args");
  if (R == 0) return DXA_OK;
  const dim3 grid((unsigned)((C + TPB - 1) / TPB), (unsigned)R);
  if (dtype == DXA_BF16)
    hipLaunchKernelGGL((token_sum_k<bf16_t>), grid, dim3(TPB), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, Nn, C, scale);
  else
    hipLaunchKernelGGL((token_sum_k<float>), grid, dim3(TPB), 0, (hipStream_t)stream, (const float*)x, (float*)out, Nn, C, scale);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_dropout_mask(void* out, int64_t n, float p, uint64_t seed, uint64_t offset, int dtype, dxa_stream_t stream) {
  DXA_CHECK_ARG(out && n >= 0 && p >= 0.f && p < 1.f && (dtype == DXA_F32 || dtype == DXA_BF16), "dxa_dropout_mask: bad args");
  if (n == 0) return DXA_OK;
  const float ks = 1.f / (1.f - p);
  const dim3 g(dxa_grid1d((n + 3) / 4, TPB));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DXA_BF16)
    hipLaunchKernelGGL((dropout_mask_k<bf16_t>), g, dim3(TPB), 0, st, (bf16_t*)out, n, p, ks, (uint32_t)seed, (uint32_t)(seed >> 32),
                       (uint32_t)offset, (uint32_t)(offset >> 32));
  else
    hipLaunchKernelGGL((dropout_mask_k<float>), g, dim3(TPB), 0, st, (float*)out, n, p, ks, (uint32_t)seed, (uint32_t)(seed >> 32),
                       (uint32_t)offset, (uint32_t)(offset >> 32));
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

extern "C" int dxa_bank_consolidate(void* feat, float* ts, int len, int64_t N, int64_t D, int dtype, int fifo, float* sims,
                                    dxa_stream_t stream) {
  DXA_CHECK_ARG(feat && ts && (sims || fifo) && len >= 2 && N > 0 && D > 0 && (dtype == DXA_F32 || dtype == DXA_BF16),
                "dxa_bank_consolidate: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t E = N * D;
  if (dtype == DXA_BF16) {
    if (!fifo) hipLaunchKernelGGL((bank_sims_k<bf16_t>), dim3(len - 1), dim3(TPB), 0, st, (const bf16_t*)feat, sims, N, D);
    hipLaunchKernelGGL((bank_merge_k<bf16_t>), dim3(dxa_grid1d(E, TPB, 256)), dim3(TPB), 0, st, (bf16_t*)feat, ts, sims, len, E, fifo);
  } else {
    if (!fifo) hipLaunchKernelGGL((bank_sims_k<float>), dim3(len - 1), dim3(TPB), 0, st, (const float*)feat, sims, N, D);
    hipLaunchKernelGGL((bank_merge_k<float>), dim3(dxa_grid1d(E, TPB, 256)), dim3(TPB), 0, st, (float*)feat, ts, sims, len, E, fifo);
  }
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
