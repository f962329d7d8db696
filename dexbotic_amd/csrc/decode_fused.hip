// One decoder pass of a KV-cached greedy-decode step (batch 1, ONE new token) in one persistent launch.
//
// Reference: the use_cache=True single-token pass of HF Qwen2Model that GenerationMixin.generate drives from
// DiscreteVLAForCausalLM.generate (dexbotic/model/discrete_vla/discrete_vla_arch.py:24-50, dexbotic_arch.py:429-496):
// per layer  h = RMSNorm(x); q,k,v = h W^T + b; RoPE(q, k); cache append; softmax(q K^T / sqrt(D)) V; x += o W_o^T;
//            h = RMSNorm(x); x += (silu(h W_g^T) * h W_u^T) W_d^T;   and the final RMSNorm.
//
// As separate launches that is ~340 kernels per token of which 113 are weight streams of 26-270 MB that each ramp up and
// drain (gemm_skinny_bf16_kernel: 4.6 TB/s inside the calls, 3.4 TB/s over the token: profiles/r05_decode_per_token_kernel_stats.txt).
// The step is a pure stream over 13 GB of bf16 weights with one row of activations, so here ONE grid of co-resident
// workgroups (one per CU) walks the five phases of every layer
//     [RMSNorm] qkv GEMV + bias | RoPE + cache append + attention over the cache | o GEMV + residual |
//     [RMSNorm] gate/up GEMV + SiLU * up | down GEMV + residual
// with a device-wide barrier between them (the dit_fused.hip barrier: 16 spread arrival counters, bounded spin, abort word).
// A workgroup owns N / grid output rows of a product (18 / 14 / 74 pairs / 14 at the 7B widths on 256 CUs: no tail).  Its 8
// waves split K in 512-wide blocks (block b -> wave b % 8): a lane's 16-byte load is 8 consecutive k of one weight row, the
// activation vector sits in LDS as bf16 (the rounding point of the unfused path: every activation between two ops is a bf16
// tensor there), products are v_dot2c_f32_bf16 into fp32, 16 rows are in flight per wave and their 16 lane-partials are folded
// with a halving butterfly (17 shuffles instead of 96).  Activations that cross workgroups (x, qkv, attention output, gated
// MLP activation: <= 38 KB) are agent-scope (sc1) accesses: stores write through, loads miss in the XCD-private L2s.
// Rounding points = those of the unfused path (norm.hip rmsnorm_fwd_fast_k, elementwise.hip rope_k / swiglu_fwd_k, the
// GEMM epilogue): bf16 after the norm, after bias, after each RoPE product, after silu, after the gate product, after
// the residual add.  Attention is fp32 from bf16 q / k / v (the flash kernel of the unfused path rounds the probabilities
// to bf16 for its MFMA; this one does not), so the two paths agree to bf16 rounding, not bit for bit.
#include <map>
#include <mutex>

#include "common.h"

namespace {

constexpr int SC1 = 16;            // buffer-instruction cache policy: agent scope (gfx94x / gfx95x)
constexpr unsigned NCTR = 16;      // arrival counters of the device-wide barrier
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int MAXR = 160;          // rows (a gate / up pair counts 2) a workgroup folds per pass
constexpr int ACT_MAX = 32768;     // longest activation vector (elements) staged in LDS: 64 KiB of bf16
constexpr int T_MAX = ACT_MAX / 2 - 1;   // cached keys per sequence (descriptor sizes and loop counters are 32-bit; nothing is staged per key)
constexpr int D_MAX = 256;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

struct DecP {
  const void* const* lp;          // [L][9]: ln1_w, qkv_w, qkv_b, o_w, ln2_w, gate_up_w, down_w, k_cache, v_cache
  const bf16_t* x_in;             // [d] embedding of the new token
  bf16_t* out;                    // [d] post final-norm hidden state
  const bf16_t* final_w;
  const float* cosr; const float* sinr;   // [D / 2] rotary row of the new token's position
  bf16_t *xres, *qkv, *ao, *act;  // workspace: residual stream [d], raw q|k|v [nq], attention output [Hq D], gated activation [F]
  unsigned* bar;
  int L, d, Hq, Hkv, D, F, slot, kv_lo, max_len;
  float eps, scale;
};

struct alignas(16) DSmem {
  bf16_t act[ACT_MAX];            // activation vector of the running product | the attention phase's partial softmax states
  float red[8][MAXR];
  float q[D_MAX], kn[D_MAX], vn[D_MAX];
  float wred[16];
};

// a pointer that is the same in every lane, moved to scalar registers: a buffer descriptor built from a VGPR pointer makes hipcc wrap
// EVERY buffer instruction in a readfirstlane "waterfall" loop (the table entries come out of a vector load)
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uni(p)), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ float ld_bf(__amdgpu_buffer_rsrc_t r, int idx) {      // coherent load of one bf16
  return bf2f(__builtin_amdgcn_raw_buffer_load_b16(r, idx * 2, 0, SC1));
}
__device__ __forceinline__ void st_bf(__amdgpu_buffer_rsrc_t r, int idx, float v) {
  __builtin_amdgcn_raw_buffer_store_b16(f2bf(v), r, idx * 2, 0, SC1);
}

// Tuning build (-DDXA_DEC_STAMPS=<workgroup>): s_memtime of wave 0 of one workgroup at the start of every phase, in front of its
// barrier and behind it, summed per phase (0 qkv, 1 attention, 2 o, 3 gate / up, 4 down) as [work, barrier] cycles;
// dxa_decode_debug_stamps() copies the 10 sums out.
#if defined(DXA_DEC_STAMPS)
__device__ unsigned long long g_dec_stamps[16];   // [10..13]: attention: request + RoPE | keys | merge + store
#define DEC_T(v_) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_) :: "memory"); } while (0)
#define DEC_ATT(i_, a_, b_) do { if (blockIdx.x == DXA_DEC_STAMPS && threadIdx.x == 0) g_dec_stamps[10 + (i_)] += (b_) - (a_); } while (0)
#define DEC_PHASE_BEGIN() unsigned long long t0_, t1_, t2_; DEC_T(t0_)
#define DEC_PHASE_SYNC() DEC_T(t1_)
__device__ unsigned long long g_dec_wg[5][256];     // work cycles of EVERY workgroup per phase (who are the stragglers?)
#define DEC_PHASE_END(ph_) do { DEC_T(t2_); if (threadIdx.x == 0 && blockIdx.x < 256) g_dec_wg[ph_][blockIdx.x] += t1_ - t0_; if (blockIdx.x == DXA_DEC_STAMPS && threadIdx.x == 0) { g_dec_stamps[2 * (ph_)] += t1_ - t0_; g_dec_stamps[2 * (ph_) + 1] += t2_ - t1_; } } while (0)
#else
#define DEC_T(v_) do { } while (0)
#define DEC_ATT(i_, a_, b_) do { } while (0)
#define DEC_PHASE_BEGIN() do { } while (0)
#define DEC_PHASE_SYNC() do { } while (0)
#define DEC_PHASE_END(ph_) do { } while (0)
#endif
// the barrier of dit_fused.hip (see there for the measurements behind the 16 counters)
// (round 6 A/B, profiles/r06_decode_prefetch_ab.txt: requesting the next product's first weight block in front of the barrier made
//  the token 6 % SLOWER — the 28 MB burst of 256 workgroups sits in front of the small activation loads every phase starts with;
//  the stream is bandwidth-bound, a prefetch adds no bandwidth.  Removed.)
__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned nblk, unsigned& epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  epoch += 1;
  if (threadIdx.x < 64) {
    unsigned* abortw = bar + 56;
    if (threadIdx.x == 0)
      (void)__hip_atomic_fetch_add(bar + 1024u * (blockIdx.x % NCTR + 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned g = threadIdx.x % NCTR;
    const unsigned target = epoch * ((nblk + NCTR - 1u - g) / NCTR);
    const unsigned* mine = bar + 1024u * (g + 1u);
    unsigned spins = 0;
    while (true) {
      const unsigned v = threadIdx.x < NCTR ? __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
      if (__builtin_amdgcn_ballot_w64(v < target) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 15u) == 0u) {
        if (spins >= SPIN_LIMIT && threadIdx.x == 0) __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins >= SPIN_LIMIT) break;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float block_reduce(float v, float* wred, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = wred[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) t = is_max ? fmaxf(t, wred[i]) : t + wred[i];
  return t;
}

// s.act[0, n) = src[0, n) (bf16, written by other workgroups before the last barrier)
__device__ __forceinline__ void stage_plain(DSmem& s, const bf16_t* src, int n) {
  const __amdgpu_buffer_rsrc_t r = rsrc_of(src, (size_t)n * 2);
  for (int c = threadIdx.x * 8; c < n; c += 512 * 8) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, c * 2, 0, SC1);
    *reinterpret_cast<u32x4_t*>(&s.act[c]) = v;
  }
  __syncthreads();
}

// s.act[0, n) = bf16(w * bf16(x * rstd)), rstd = rsqrt(mean(x^2) + eps): HF Qwen2RMSNorm on a bf16 row (norm.hip rounding points)
__device__ __forceinline__ void stage_norm(DSmem& s, const bf16_t* x, const bf16_t* w, int n, float eps) {
  const __amdgpu_buffer_rsrc_t r = rsrc_of(x, (size_t)n * 2);
  float ss = 0.f;
  for (int c = threadIdx.x * 8; c < n; c += 512 * 8) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, c * 2, 0, SC1);
    *reinterpret_cast<u32x4_t*>(&s.act[c]) = v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = __uint_as_float(v[e] << 16), b = __uint_as_float(v[e] & 0xffff0000u);
      ss += a * a + b * b;
    }
  }
  ss = block_reduce(ss, s.wred, false);
  const float rstd = rsqrtf(ss / (float)n + eps);
  for (int c = threadIdx.x * 8; c < n; c += 512 * 8) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&s.act[c]);
    const u32x4_t g = *reinterpret_cast<const u32x4_t*>(w + c);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = __uint_as_float(v[e] << 16), b = __uint_as_float(v[e] & 0xffff0000u);
      const float ga = __uint_as_float(g[e] << 16), gb = __uint_as_float(g[e] & 0xffff0000u);
      o[e] = pack_bf16x2(ga * rnd<bf16_t>(a * rstd), gb * rnd<bf16_t>(b * rstd));
    }
    *reinterpret_cast<u32x4_t*>(&s.act[c]) = o;
  }
  __syncthreads();
}

constexpr int RG = 16;             // weight rows in flight per wave
// This wave's share (its 512-wide K blocks) of the dot products of RG weight rows with s.act; lane-partials folded by a halving
// butterfly: on return lane l with (l & 3) == 0 holds in `out` the wave's partial of row index
// 8 bit5(l) + 4 bit4(l) + 2 bit3(l) + bit2(l).  rows[] are row numbers of W (uniform over the wave).
__device__ __forceinline__ float wave_rows(const DSmem& s, const bf16_t* __restrict__ W, int64_t ldw, int K, const int64_t (&rows)[RG],
                                           int wave, int lane) {
  float acc[RG];
#pragma unroll
  for (int i = 0; i < RG; ++i) acc[i] = 0.f;
  const int nblk = (K + 511) >> 9;
  for (int b = wave; b < nblk; b += 8) {
    const int k = (b << 9) + lane * 8;
    if (k < K) {                                      // K % 8 == 0: a lane's 8 k are all inside or all outside
      const u32x4_t hv = *reinterpret_cast<const u32x4_t*>(&s.act[k]);
      u32x4_t wv[RG];
#pragma unroll
      for (int i = 0; i < RG; ++i)
#if defined(DXA_DEC_PLAINLD)
        wv[i] = *reinterpret_cast<const u32x4_t*>(W + rows[i] * ldw + k);
#else
        wv[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(W + rows[i] * ldw + k));
#endif
#pragma unroll
      for (int i = 0; i < RG; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#if defined(DXA_DEC_FMA)
          acc[i] += __uint_as_float(wv[i][e] << 16) * __uint_as_float(hv[e] << 16);
          acc[i] += __uint_as_float(wv[i][e] & 0xffff0000u) * __uint_as_float(hv[e] & 0xffff0000u);
#else
          // (inline asm: __builtin_amdgcn_fdot2_f32_bf16 on elements of a 4 x u32 vector came out of hipcc 7.2 with ONE register for
          //  all four activation words and repeated weight words — results off by O(1), found by scripts/probes/decode_step_debug.py)
          asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[i]) : "v"(wv[i][e]), "v"(hv[e]));
#endif
        }
    }
  }
  float v8[8], v4[4], v2[2];
  {
    const bool hi = (lane & 32) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float keep = hi ? acc[8 + i] : acc[i], send = hi ? acc[i] : acc[8 + i];
      v8[i] = keep + __shfl_xor(send, 32, 64);
    }
  }
  {
    const bool hi = (lane & 16) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float keep = hi ? v8[4 + i] : v8[i], send = hi ? v8[i] : v8[4 + i];
      v4[i] = keep + __shfl_xor(send, 16, 64);
    }
  }
  {
    const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float keep = hi ? v4[2 + i] : v4[i], send = hi ? v4[i] : v4[2 + i];
      v2[i] = keep + __shfl_xor(send, 8, 64);
    }
  }
  float v1;
  {
    const bool hi = (lane & 4) != 0;
    const float keep = hi ? v2[1] : v2[0], send = hi ? v2[0] : v2[1];
    v1 = keep + __shfl_xor(send, 4, 64);
  }
  v1 += __shfl_xor(v1, 2, 64);
  v1 += __shfl_xor(v1, 1, 64);
  return v1;
}
__device__ __forceinline__ int row_of_lane(int lane) {
  return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

enum { EPI_BIAS = 0, EPI_RES = 1 };
// out[r] = epi(W[r, :] . s.act) for this workgroup's rows of an [N, K] matrix.
//   EPI_BIAS: dst[r] = bf16(acc + bias[r]);   EPI_RES: dst[r] = bf16(acc + res[r])   (dst, res: activations, sc1)
template <int EPI>
__device__ __forceinline__ void gemv_plain(DSmem& s, const bf16_t* W, int N, int K, const bf16_t* bias, const bf16_t* res, bf16_t* dst,
                                           unsigned nwg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = (int)((int64_t)N * blockIdx.x / nwg), hi = (int)((int64_t)N * (blockIdx.x + 1) / nwg);
  const __amdgpu_buffer_rsrc_t rD = rsrc_of(dst, (size_t)N * 2), rR = rsrc_of(res ? res : dst, (size_t)N * 2);
  for (int p0 = lo; p0 < hi; p0 += MAXR) {
    const int pn = min(MAXR, hi - p0);
    for (int g0 = 0; g0 < pn; g0 += RG) {
      int64_t rows[RG];
#pragma unroll
      for (int i = 0; i < RG; ++i) rows[i] = p0 + min(g0 + i, pn - 1);
      const float v = wave_rows(s, W, K, K, rows, wave, lane);
      const int ri = g0 + row_of_lane(lane);
      if ((lane & 3) == 0 && ri < pn) s.red[wave][ri] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < pn) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += s.red[w][threadIdx.x];
      const int r = p0 + threadIdx.x;
      if (EPI == EPI_BIAS) a += bias ? bf2f(bias[r]) : 0.f;
      else a += ld_bf(rR, r);
      st_bf(rD, r, a);
    }
    __syncthreads();
  }
}

// act[p] = bf16(bf16(silu(g)) * u), g = bf16(W[p, :] . h), u = bf16(W[F + p, :] . h) for this workgroup's p (HF Qwen2MLP)
__device__ __forceinline__ void gemv_glu(DSmem& s, const bf16_t* W, int F, int K, bf16_t* dst, unsigned nwg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = (int)((int64_t)F * blockIdx.x / nwg), hi = (int)((int64_t)F * (blockIdx.x + 1) / nwg);
  const __amdgpu_buffer_rsrc_t rD = rsrc_of(dst, (size_t)F * 2);
  constexpr int MAXP = MAXR / 2;
  for (int p0 = lo; p0 < hi; p0 += MAXP) {
    const int pn = min(MAXP, hi - p0);
    for (int g0 = 0; g0 < pn; g0 += RG / 2) {            // 8 pairs = 16 rows per group: slot 2 i = gate, 2 i + 1 = up
      int64_t rows[RG];
#pragma unroll
      for (int i = 0; i < RG; ++i) rows[i] = (int64_t)(i & 1) * F + p0 + min(g0 + (i >> 1), pn - 1);
      const float v = wave_rows(s, W, K, K, rows, wave, lane);
      const int ri = 2 * g0 + row_of_lane(lane);
      if ((lane & 3) == 0 && ri < 2 * pn) s.red[wave][ri] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < pn) {
      float g = 0.f, u = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { g += s.red[w][2 * threadIdx.x]; u += s.red[w][2 * threadIdx.x + 1]; }
      g = rnd<bf16_t>(g); u = rnd<bf16_t>(u);
      st_bf(rD, p0 + threadIdx.x, rnd<bf16_t>(g / (1.f + expf(-g))) * u);
    }
    __syncthreads();
  }
}

// workgroup h < Hq: RoPE of q_h and of the new key, cache append (first head of a GQA group), softmax(q K^T scale) V over the
// cached keys [kv_lo, slot) and the new one
__device__ __forceinline__ void attention_phase(const DecP& p, DSmem& s, bf16_t* kc, bf16_t* vc) {
  const int h = blockIdx.x;
  if (h >= p.Hq) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = p.D, half = D >> 1, grp = p.Hq / p.Hkv, g = h / grp;
  const int nq = (p.Hq + 2 * p.Hkv) * D;
  const __amdgpu_buffer_rsrc_t rQ = rsrc_of(p.qkv, (size_t)nq * 2);
  bf16_t* kg = kc + (int64_t)g * p.max_len * D;
  bf16_t* vg = vc + (int64_t)g * p.max_len * D;
  // ---- the first AU key groups of this wave are requested BEFORE the RoPE stage (they do not depend on q): the cache rows were
  //      last touched a token ago (HBM, cold TLB) and arrive while q / k / v come in from the qkv scratch
  const int T = p.slot - p.kv_lo;
  const int lpk = D >> 3, kpw = 64 / lpk;              // lanes per key (16 B each), keys per wave load
  const int kq = lane / lpk, dl = lane % lpk;
  constexpr int AU = 12;                                // 8 waves x 12 x (64 / lpk) keys per round: 384 at head_dim 128
  // (buffer loads: one per-lane byte offset + a scalar offset per key group, rows past the last cached key read zeros and are
  //  never folded; 24 flat addresses would take 48 VGPRs)
  const __amdgpu_buffer_rsrc_t rK = rsrc_of(kg + (int64_t)p.kv_lo * D, (size_t)T * D * 2);
  const __amdgpu_buffer_rsrc_t rV = rsrc_of(vg + (int64_t)p.kv_lo * D, (size_t)T * D * 2);
  u32x4_t kr[AU], vr[AU];
  auto request = [&](int j0) {
    const uint32_t voff = (uint32_t)(((j0 + kq) * D + dl * 8) * 2);
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      // (the group offset rides in the VECTOR offset: the descriptor's range check does not see a scalar offset)
      kr[u] = __builtin_amdgcn_raw_buffer_load_b128(rK, voff + (uint32_t)(u * kpw * D * 2), 0, 0);
      vr[u] = __builtin_amdgcn_raw_buffer_load_b128(rV, voff + (uint32_t)(u * kpw * D * 2), 0, 0);
    }
  };
  const int j_first = wave * kpw * AU, j_stride = 8 * kpw * AU;
  unsigned long long ta0 = 0, ta1 = 0, ta2 = 0, ta3 = 0;
  (void)ta0; (void)ta1; (void)ta2; (void)ta3;
  DEC_T(ta0);
  if (j_first < T) request(j_first);
  // ---- RoPE (elementwise.hip rope_k: every product rounded to bf16, cos / sin rounded to bf16)
  if (tid < D) {
    const bool isk = tid >= half;
    const int t = isk ? tid - half : tid;
    const int base = isk ? (p.Hq + g) * D : h * D;
    const float x1 = ld_bf(rQ, base + t), x2 = ld_bf(rQ, base + t + half);
    const float cc = rnd<bf16_t>(p.cosr[t]), sn = rnd<bf16_t>(p.sinr[t]);
    const float o1 = rnd<bf16_t>(rnd<bf16_t>(x1 * cc) + rnd<bf16_t>(-x2 * sn));
    const float o2 = rnd<bf16_t>(rnd<bf16_t>(x2 * cc) + rnd<bf16_t>(x1 * sn));
    float* dstv = isk ? s.kn : s.q;
    dstv[t] = o1; dstv[t + half] = o2;
  } else if (tid < 2 * D) {
    const int t = tid - D;
    s.vn[t] = ld_bf(rQ, (p.Hq + p.Hkv + g) * D + t);
  }
  __syncthreads();
  if (h % grp == 0 && tid < D) {                       // append: read by LATER launches only (this step takes it from LDS)
    kg[(int64_t)p.slot * D + tid] = f2bf(s.kn[tid]);
    vg[(int64_t)p.slot * D + tid] = f2bf(s.vn[tid]);
  }
  // ---- one pass over the cached keys [kv_lo, slot): a lane group of D / 8 lanes owns a key (16 bytes of K and of V per lane), a wave
  //      has AU such key groups of K AND V in flight at once (a cache of <= 384 keys is ONE round trip; the first form of this phase —
  //      load, reduce, next — was 9 dependent round trips for the scores and 9 more for P V: 15.7 us of a 115 us layer,
  //      profiles/r06_decode_stamps.txt) and keeps an online softmax (running max, sum, 8 output dims) per lane group; the new key
  //      joins wave 0's first group from LDS; the 8 x (64 / lpk) partial states are merged through LDS.
  DEC_T(ta1);
  float qf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] = s.q[dl * 8 + e] * p.scale;
  float m = -INFINITY, lsum = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  auto fold = [&](float sc, const float (&vf)[8]) {   // one key into the lane group's running state
    const float mn = fmaxf(m, sc);
    const float so = __expf(m - mn), pj = __expf(sc - mn);   // (m = -inf: so = 0)
    lsum = lsum * so + pj;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] * so + pj * vf[e];
    m = mn;
  };
  for (int j0 = j_first; j0 < T; j0 += j_stride) {
    float dots[AU];
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        dot += qf[2 * e] * __uint_as_float(kr[u][e] << 16) + qf[2 * e + 1] * __uint_as_float(kr[u][e] & 0xffff0000u);
      dots[u] = dot;
    }
    for (int o = 1; o < lpk; o <<= 1)                   // the AU reductions side by side (a shuffle is an LDS round trip)
#pragma unroll
      for (int u = 0; u < AU; ++u) dots[u] += __shfl_xor(dots[u], o, 64);
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      float vf[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { vf[2 * e] = __uint_as_float(vr[u][e] << 16); vf[2 * e + 1] = __uint_as_float(vr[u][e] & 0xffff0000u); }
      if (j0 + u * kpw + kq < T) fold(dots[u], vf);
    }
    if (j0 + j_stride < T) request(j0 + j_stride);
  }
  if (wave == 0 && kq == 0) {                          // the new token's own key / value
    float dot = 0.f, vf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dot += qf[e] * s.kn[dl * 8 + e]; vf[e] = s.vn[dl * 8 + e]; }
    for (int o = 1; o < lpk; o <<= 1) dot += __shfl_xor(dot, o, 64);
    fold(dot, vf);                                     // (the shuffles above stay inside the active lane group)
  }
  DEC_T(ta2);
  // ---- merge the 8 kpw partial states: maxima, sums and [8 kpw][D] outputs in the (now free) activation buffer; wave 0 turns the
  //      maxima into weights exp(m_i - M) and the total sum, then D threads add the weighted partial outputs
  float* mm = reinterpret_cast<float*>(s.act);
  float* ll = mm + 64;
  float* oo = ll + 64;
  const int np = 8 * kpw, slot_id = wave * kpw + kq;
  if (dl == 0) { mm[slot_id] = m; ll[slot_id] = lsum; }
#pragma unroll
  for (int e = 0; e < 8; ++e) oo[slot_id * D + dl * 8 + e] = acc[e];
  __syncthreads();
  if (wave == 0) {
    const float mi = lane < np ? mm[lane] : -INFINITY;
    const float M = wave_max(mi);
    const float wi = lane < np ? __expf(mi - M) : 0.f;   // (an empty group: exp(-inf) = 0)
    const float Lt = wave_sum(lane < np ? ll[lane] * wi : 0.f);
    if (lane < np) mm[lane] = wi;
    if (lane == 0) s.wred[0] = Lt;
  }
  __syncthreads();
  if (tid < D) {
    float o = 0.f;
#pragma unroll 8
    for (int i = 0; i < np; ++i) o += oo[i * D + tid] * mm[i];
    st_bf(rsrc_of(p.ao, (size_t)p.Hq * D * 2), h * D + tid, o / s.wred[0]);
  }
  DEC_T(ta3);
  DEC_ATT(0, ta0, ta1); DEC_ATT(1, ta1, ta2); DEC_ATT(2, ta2, ta3);
}

__global__ __launch_bounds__(512) void decode_step_k(const DecP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  DSmem& s = *reinterpret_cast<DSmem*>(smem_raw);
  unsigned epoch = 0;
  const unsigned nwg = gridDim.x;
  const int nq = (p.Hq + 2 * p.Hkv) * p.D;
  const int HD = p.Hq * p.D;
  for (int l = 0; l < p.L; ++l) {
    const void* const* lw = p.lp + (size_t)l * 9;
    const bf16_t* xsrc = l == 0 ? p.x_in : p.xres;
    // ---- qkv = RMSNorm(x) W_qkv^T + b
    { DEC_PHASE_BEGIN();
    stage_norm(s, xsrc, (const bf16_t*)lw[0], p.d, p.eps);
    gemv_plain<EPI_BIAS>(s, (const bf16_t*)lw[1], nq, p.d, (const bf16_t*)lw[2], nullptr, p.qkv, nwg);
    DEC_PHASE_SYNC();
    grid_sync(p.bar, nwg, epoch);
    DEC_PHASE_END(0); }
    // ---- attention over the cache
    { DEC_PHASE_BEGIN();
    attention_phase(p, s, (bf16_t*)lw[7], (bf16_t*)lw[8]);
    DEC_PHASE_SYNC();
    grid_sync(p.bar, nwg, epoch);
    DEC_PHASE_END(1); }
    // ---- x = x + o W_o^T
    { DEC_PHASE_BEGIN();
    stage_plain(s, p.ao, HD);
    gemv_plain<EPI_RES>(s, (const bf16_t*)lw[3], p.d, HD, nullptr, xsrc, p.xres, nwg);
    DEC_PHASE_SYNC();
    grid_sync(p.bar, nwg, epoch);
    DEC_PHASE_END(2); }
    // ---- a = silu(h W_g^T) * (h W_u^T), h = RMSNorm(x)
    { DEC_PHASE_BEGIN();
    stage_norm(s, p.xres, (const bf16_t*)lw[4], p.d, p.eps);
    gemv_glu(s, (const bf16_t*)lw[5], p.F, p.d, p.act, nwg);
    DEC_PHASE_SYNC();
    grid_sync(p.bar, nwg, epoch);
    DEC_PHASE_END(3); }
    // ---- x = x + a W_d^T
    { DEC_PHASE_BEGIN();
    stage_plain(s, p.act, p.F);
    gemv_plain<EPI_RES>(s, (const bf16_t*)lw[6], p.d, p.F, nullptr, p.xres, p.xres, nwg);
    DEC_PHASE_SYNC();
    grid_sync(p.bar, nwg, epoch);
    DEC_PHASE_END(4); }
  }
  if (blockIdx.x == 0) {
    stage_norm(s, p.L > 0 ? p.xres : p.x_in, p.final_w, p.d, p.eps);
    for (int c = threadIdx.x * 8; c < p.d; c += 512 * 8)
      *reinterpret_cast<u32x4_t*>(p.out + c) = *reinterpret_cast<const u32x4_t*>(&s.act[c]);
  }
  // leave the barrier state zeroed for the next launch on this stream (see dit_fused.hip: atomics, not a memset node)
  if (threadIdx.x == 0) {
    unsigned* exit_cnt = p.bar + 48;
    const unsigned outn = __hip_atomic_fetch_add(exit_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (outn == nwg) {
      for (unsigned g = 0; g < NCTR; ++g)
        __hip_atomic_store(p.bar + 1024u * (g + 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(exit_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

constexpr size_t SYNC_BYTES = 4096 * (NCTR + 1);
int get_sync_block(hipStream_t st, unsigned** out) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, unsigned*> tab;
  int dev = 0;
  DXA_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  auto it = tab.find({dev, st});
  if (it == tab.end()) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      dxa_set_error("dxa_decode_step: first use on a stream allocates its sync block and cannot happen under stream capture");
      return DXA_ERR_BAD_ARG;
    }
    unsigned* p = nullptr;
    DXA_CHECK_HIP(hipMalloc((void**)&p, SYNC_BYTES));
    DXA_CHECK_HIP(hipMemset(p, 0, SYNC_BYTES));
    DXA_CHECK_HIP(hipDeviceSynchronize());
    it = tab.emplace(std::make_pair(dev, st), p).first;
  }
  *out = it->second;
  return DXA_OK;
}
size_t ws_elems(int d, int Hq, int Hkv, int D, int F) {
  auto up = [](size_t v) { return (v + 127) / 128 * 128; };
  return up((size_t)d) + up((size_t)(Hq + 2 * Hkv) * D) + up((size_t)Hq * D) + up((size_t)F);
}

}  // namespace

extern "C" size_t dxa_decode_step_workspace(int d, int Hq, int Hkv, int D, int F) {
  if (d <= 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || F <= 0) return 0;
  return ws_elems(d, Hq, Hkv, D, F) * sizeof(bf16_t);
}

extern "C" int dxa_decode_step(const dxa_decode_desc* q, dxa_stream_t stream) {
  DXA_CHECK_ARG(q && q->layers && q->x_in && q->out && q->final_norm_w && q->cos_row && q->sin_row && q->workspace,
                "dxa_decode_step: null pointer");
  DXA_CHECK_ARG(q->n_layers >= 0 && q->d > 0 && q->Hq > 0 && q->Hkv > 0 && q->Hq % q->Hkv == 0 && q->F > 0,
                "dxa_decode_step: bad sizes");
  DXA_CHECK_ARG(q->D == 64 || q->D == 128 || q->D == 256, "dxa_decode_step: head_dim 64, 128 or 256 (got %d)", q->D);
  DXA_CHECK_ARG(q->d % 8 == 0 && q->F % 8 == 0, "dxa_decode_step: hidden and MLP widths must be multiples of 8");
  DXA_CHECK_ARG(q->d <= ACT_MAX && q->F <= ACT_MAX && q->Hq * q->D <= ACT_MAX,
                "dxa_decode_step: activation vectors of at most %d elements", ACT_MAX);
  DXA_CHECK_ARG((int64_t)2 * q->F * q->d * 2 < (1ll << 31) && (int64_t)(q->Hq + 2 * q->Hkv) * q->D * q->d * 2 < (1ll << 31),
                "dxa_decode_step: a weight matrix of 2 GiB or more (32-bit buffer offsets)");
  DXA_CHECK_ARG(q->slot >= 0 && q->slot < q->max_len && q->kv_lo >= 0 && q->kv_lo <= q->slot,
                "dxa_decode_step: cache slot %d outside [kv_lo %d, max_len %d)", q->slot, q->kv_lo, q->max_len);
  DXA_CHECK_ARG(q->slot - q->kv_lo <= T_MAX, "dxa_decode_step: at most %d cached keys", T_MAX);
  DXA_CHECK_ARG(q->workspace_bytes >= dxa_decode_step_workspace(q->d, q->Hq, q->Hkv, q->D, q->F), "dxa_decode_step: workspace too small");
  DXA_CHECK_ARG((reinterpret_cast<uintptr_t>(q->workspace) % 256) == 0 && (reinterpret_cast<uintptr_t>(q->x_in) % 16) == 0 &&
                (reinterpret_cast<uintptr_t>(q->out) % 16) == 0 && (reinterpret_cast<uintptr_t>(q->final_norm_w) % 16) == 0,
                "dxa_decode_step: workspace must be 256-byte, x_in / out / final_norm_w 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  DecP p;
  p.lp = q->layers;
  p.x_in = (const bf16_t*)q->x_in; p.out = (bf16_t*)q->out; p.final_w = (const bf16_t*)q->final_norm_w;
  p.cosr = q->cos_row; p.sinr = q->sin_row;
  auto up = [](size_t v) { return (v + 127) / 128 * 128; };
  p.xres = (bf16_t*)q->workspace;
  p.qkv = p.xres + up((size_t)q->d);
  p.ao = p.qkv + up((size_t)(q->Hq + 2 * q->Hkv) * q->D);
  p.act = p.ao + up((size_t)q->Hq * q->D);
  if (int rc = get_sync_block(st, &p.bar)) return rc;
  p.L = q->n_layers; p.d = q->d; p.Hq = q->Hq; p.Hkv = q->Hkv; p.D = q->D; p.F = q->F;
  p.slot = q->slot; p.kv_lo = q->kv_lo; p.max_len = q->max_len;
  p.eps = q->eps; p.scale = 1.f / sqrtf((float)q->D);
  // every workgroup must be resident at once (device-wide barrier): one per CU, at least one per attention head
  static int resident = 0;
  if (resident == 0) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    DXA_CHECK_HIP(hipGetDevice(&dev));
    DXA_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    DXA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_step_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)sizeof(DSmem)));
    DXA_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_step_k, 512, sizeof(DSmem)));
    DXA_CHECK_ARG(per_cu >= 1, "dxa_decode_step: the kernel does not fit on this device");
    resident = prop.multiProcessorCount;          // one workgroup per CU: the weight stream wants every CU's load queue, not more waves
  }
  int grid = resident;
  static const int grid_cap = getenv("DXA_DECODE_GRID") ? atoi(getenv("DXA_DECODE_GRID")) : 0;   // tuning aid
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  DXA_CHECK_ARG(q->Hq <= grid, "dxa_decode_step: %d attention heads need at least as many workgroups (%d)", q->Hq, grid);
  hipLaunchKernelGGL(decode_step_k, dim3(grid), dim3(512), sizeof(DSmem), st, p);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

#if defined(DXA_DEC_STAMPS)
extern "C" int dxa_decode_debug_wg(unsigned long long* out) {      // tuning build only: [5][256] work cycles per workgroup, then zeroed
  DXA_CHECK_HIP(hipDeviceSynchronize());
  DXA_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_wg), sizeof(unsigned long long) * 5 * 256));
  static unsigned long long z[5 * 256];
  DXA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dec_wg), z, sizeof(z)));
  return DXA_OK;
}
// tuning build only: [work, barrier] cycle sums per phase since the last call, then zeroed
extern "C" int dxa_decode_debug_stamps(unsigned long long* out) {
  DXA_CHECK_HIP(hipDeviceSynchronize());
  DXA_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_stamps), sizeof(unsigned long long) * 16));
  unsigned long long z[16] = {0};
  DXA_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dec_stamps), z, sizeof(z)));
  return DXA_OK;
}
#endif

// 1 if a decode launch on this stream gave up at a device-wide barrier since the last call (its output is garbage); the barrier
// state is re-armed.  Synchronises the stream.
extern "C" int dxa_decode_status(dxa_stream_t stream, int* timed_out) {
  DXA_CHECK_ARG(timed_out != nullptr, "dxa_decode_status: null output");
  hipStream_t st = (hipStream_t)stream;
  unsigned* tail = nullptr;
  if (int rc = get_sync_block(st, &tail)) return rc;
  unsigned word = 0;
  DXA_CHECK_HIP(hipMemcpyAsync(&word, tail + 56, sizeof(word), hipMemcpyDeviceToHost, st));
  DXA_CHECK_HIP(hipStreamSynchronize(st));
  *timed_out = word != 0;
  if (word != 0) {
    DXA_CHECK_HIP(hipMemsetAsync(tail, 0, SYNC_BYTES, st));
    DXA_CHECK_HIP(hipStreamSynchronize(st));
  }
  return DXA_OK;
}
