// Device-side pieces of the GEMM translation unit (gemm.hip: generic / few-row / ring / ping-pong kernels and the dispatch; also
// included by the round-4 experiment scripts/probes/gemm_w4.hip): the launch parameter block, the epilogue building blocks
// and the split-K hand-off.  Everything except GemmP has internal linkage (one copy per translation unit).
#pragma once
#include "common.h"

namespace dxa_gemm_detail {
typedef float f32x16_t __attribute__((ext_vector_type(16)));
struct GemmP {
  int64_t M, N, K;
  const char* A; int64_t lda;
  const char* B; int64_t ldb;
  char* C; int64_t ldc;
  const char* bias;
  const char* R; int64_t ldr;
  char* aux;
  const char* G; int64_t ldg;
  float alpha;
  int act, accumulate;
  int nb1, nb2;  // nb[1], nb[2]
  int64_t sA[3], sB[3], sC[3], sR[3], sG[3];
  int tm, tn;
  int vecA, vecB, vecC, vecR, vecG, vecBias;
  // ring kernel, split-K tail: tiles [0, full) run whole; the last tail_r tiles are cut into split_s K-ranges
  int full, tail_r, split_s;
  int group_m;  // ring kernel: row-tiles per group of the tile order
  char* mirror; // bf16 copy of the fp32 output (same ldc), or null
  float* sumsq; // fp32 output: per-tile sum of squares of the final C values (one float per 256x256 tile), or null
  float* ws;    // fp32 partial accumulators, tail_r * (split_s - 1) slots of 256x256
  int* flags;   // per tail tile arrival counter (self-resetting)
  // TN with a second (A2, B2) segment of K2 contraction rows (same lda / ldb): C = A^T B + A2^T B2 (ping-pong kernel only)
  const char* A2; const char* B2; int64_t K2;
  int fuse; int64_t ldaux;   // dxa_gemm_desc.fuse / ld_aux (gated-MLP epilogue: C = silu(gate) * up, aux = pre-activations)
};
}  // namespace dxa_gemm_detail

namespace {
using dxa_gemm_detail::GemmP;
using dxa_gemm_detail::f32x16_t;
typedef uint32_t u32x4n_t __attribute__((ext_vector_type(4)));   // native 16-byte vector (nontemporal builtins)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;              // bytes per LDS row (K slab)
constexpr int TILE_BYTES = BM * ROWB;  // 16 KiB per operand per buffer

template <typename T> __device__ __forceinline__ void load4(float (&o)[4], const T* p, bool vec, int n_ok);
template <> __device__ __forceinline__ void load4<float>(float (&o)[4], const float* p, bool vec, int n_ok) {
  if (vec && n_ok == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = i < n_ok ? p[i] : 0.f;
  }
}
template <> __device__ __forceinline__ void load4<bf16_t>(float (&o)[4], const bf16_t* p, bool vec, int n_ok) {
  if (vec && n_ok == 4) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = i < n_ok ? bf2f(p[i]) : 0.f;
  }
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4], bool vec, int n_ok);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4], bool vec, int n_ok) {
  if (vec && n_ok == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < n_ok) p[i] = v[i];
  }
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4], bool vec, int n_ok) {
  if (vec && n_ok == 4) {
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = o;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < n_ok) p[i] = f2bf(v[i]);
  }
}

// fused epilogue on 4 consecutive n of row m (see dxa_gemm in the header for the operation order)
template <typename TI, typename TO>
__device__ __forceinline__ void epilogue4(const GemmP& p, TO* C, TO* AUX, const TI* R, const TI* G,
                                          const float (&bv)[4], int64_t m, int64_t n, int n_ok, const float (&a)[4]) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = a[r] * p.alpha + bv[r];
  if (AUX) store4<TO>(AUX + m * p.ldc + n, v, p.vecC, n_ok);
  if (G) {
    float g[4];
    load4<TI>(g, G + m * p.ldg + n, p.vecG, n_ok);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= act_grad(p.act, g[r]);
  } else if (p.act != DXA_ACT_NONE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = act_fwd(p.act, v[r]);
  }
  if (R) {
    float rr[4];
    load4<TI>(rr, R + m * p.ldr + n, p.vecR, n_ok);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += rr[r];
  }
  TO* cp = C + m * p.ldc + n;
  if (p.accumulate) {
    float c0[4];
    load4<TO>(c0, cp, p.vecC, n_ok);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += c0[r];
  }
  store4<TO>(cp, v, p.vecC, n_ok);
}

constexpr int NUM_CU_D = 256;   // split-K scratch slots (one per CU)

// Split-K hand-off of a tail tile: every piece but the last stores its fp32 partial (lane-linear slots, sc1 write-through)
// and bumps the tile's arrival counter; the last piece waits for them and adds them in slice order.  Returns false for a
// workgroup that is done (it only contributed a partial).
template <int AI>
__device__ __forceinline__ bool tile_split_exchange(const GemmP& p, f32x16_t (&acc)[AI][2], int tid, int split_j, int split_s,
                                                    int tail_i) {
#if defined(__HIP_DEVICE_COMPILE__)
  // ---- split-K tail: partial accumulators travel through a lane-linear fp32 slot (8 KiB per wave store).
  // sc1 (agent scope) stores write through the XCD-private L2 and sc1 loads miss in it, so no cache-wide
  // write-back / invalidate is needed; 16-byte accesses keep the gatherer off the instruction-issue limit.
  if (split_s > 1) {
    constexpr int SC1 = 16;                                   // buffer-instruction cache policy bit (gfx94x/95x)
    constexpr uint32_t SLOT = 256 * 256 * 4;                  // bytes per partial
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)(NUM_CU_D * SLOT), 0x00020000);
    const uint32_t slot0 = (uint32_t)tail_i * (uint32_t)(split_s - 1) * SLOT + (uint32_t)tid * 16u;
    if (split_j < split_s - 1) {
      const uint32_t dst = slot0 + (uint32_t)split_j * SLOT;
#pragma unroll
      for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rW, dst + ((i * 2 + j) * 4 + q) * 8192, 0, SC1);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.flags + tail_i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    // gatherer: it has the highest block ids of its tile, so its partners were dispatched before it
    if (tid == 0) {
      while (__hip_atomic_load(p.flags + tail_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < split_s - 1)
        __builtin_amdgcn_s_sleep(4);
      __hip_atomic_store(p.flags + tail_i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int sj = 0; sj < split_s - 1; ++sj) {
      const uint32_t src = slot0 + (uint32_t)sj * SLOT;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rW, src + ((i * 2 + j) * 4 + q) * 8192, 0, SC1));
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][j][4 * q + c] += v[c];
          }
        if (AI == 4 && i == 1) __builtin_amdgcn_sched_barrier(0);   // at most 16 loads (64 VGPRs) in flight
      }
    }
  }

  // ---- epilogue: accumulators -> wave-private LDS slab (fp32, padded rows) -> row-contiguous global stores.
  // The MFMA layout gives each lane 4 consecutive n of 32 different rows: stored directly that is 8-byte
  // pieces at a row stride (measured: 0.6 TB/s, 58 us per tile).  Re-read from LDS, 16 lanes cover one
  // 64-column row segment, so stores (and the residual / mulgrad / accumulate reads) are full 128/256-byte
  // lines.  Two passes of 64 rows; the slab is private to the wave, so no workgroup barrier is needed (the
  // loop's last barrier already retired every ring read and LDS-DMA).
#endif
  return true;
}

// Everything after the K loop of the 256-column-tile bf16 NT kernels (ring and ping-pong main loops share it): the
// split-K hand-off of tail tiles and the fused epilogue.  acc[i][j] = 32x32 block (32-row block i of the wave's rows,
// 32-column block j of its 64 columns) in the v_mfma_f32_32x32x16 accumulator layout with swapped operands.
// SMALL = false: 17 KiB staging slab per wave at the start of LDS (the K-loop buffers are dead by then).
// SMALL = true : 4 KiB per wave above the 128 KiB of K-loop buffers (persistent ping-pong kernel: the next tile's
//                LDS-DMA pieces are already landing in those buffers while this epilogue runs).
template <typename TO, int AI, typename TE, bool SMALL = false>
__device__ __forceinline__ void tile_finish(const GemmP& p, f32x16_t (&acc)[AI][2], char* smem, int tid, int lane, int wave,
                                            int wm, int wn, int l32, int lh, int64_t m0, int64_t n0, int split_j,
                                            int split_s, int tail_i) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BMR = AI * 64;
  if (!tile_split_exchange<AI>(p, acc, tid, split_j, split_s, tail_i)) return;

  TO* C = reinterpret_cast<TO*>(p.C);
  TO* AUX = p.aux ? reinterpret_cast<TO*>(p.aux) : nullptr;
  const TE* R = reinterpret_cast<const TE*>(p.R);
  const TE* G = reinterpret_cast<const TE*>(p.G);
  const TE* bias = reinterpret_cast<const TE*>(p.bias);
  const int cr = lane >> 4, cc = (lane & 15) * 4;
  const int64_t n = n0 + wn * 64 + cc;
  const int n_ok = (int)max((int64_t)0, min((int64_t)4, p.N - n));
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias && n_ok > 0) load4<TE>(bv, bias + n, p.vecBias, n_ok);
  if constexpr (!SMALL) {
    constexpr int ROWP = 64 * 4 + 16;
    char* slab = smem + wave * (64 * ROWP);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 2 * pass + ii;
            if (i >= AI) continue;
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            *reinterpret_cast<float4*>(slab + (ii * 32 + l32) * ROWP + (j * 32 + 8 * q + 4 * lh) * 4) = v;
          }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll 4
      for (int it = 0; it < (AI == 3 && pass == 1 ? 8 : 16); ++it) {
        const int row = cr + 4 * it;
        const int64_t m = m0 + wm * (BMR / 2) + pass * 64 + row;
        const float4 v = *reinterpret_cast<const float4*>(slab + row * ROWP + cc * 4);
        if (m < p.M && n_ok > 0) {
          const float a4[4] = {v.x, v.y, v.z, v.w};
          epilogue4<TE, TO>(p, C, AUX, R, G, bv, m, n, n_ok, a4);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    // 16 rows x 64 columns of fp32 (4 KiB) per pass; instead of row padding the 16-byte unit u of row r sits at
    // u ^ r: the 8-lane groups of the ds_write_b128 (8 consecutive rows, one unit) and the 16-lane groups of the
    // ds_read_b128 (one row, 16 units / two rows, 8 + 8 units) each touch every bank once.
    char* slab = smem + 131072 + wave * 4096;
    const int r16 = l32 & 15;
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if ((l32 >> 4) == hh) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
              *reinterpret_cast<float4*>(slab + r16 * 256 + (((j * 8 + 2 * q + lh) ^ r16) << 4)) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = cr + 4 * it;
          const int64_t m = m0 + wm * (BMR / 2) + i * 32 + hh * 16 + row;
          const float4 v = *reinterpret_cast<const float4*>(slab + row * 256 + (((lane & 15) ^ row) << 4));
          if (m < p.M && n_ok > 0) {
            const float a4[4] = {v.x, v.y, v.z, v.w};
            epilogue4<TE, TO>(p, C, AUX, R, G, bv, m, n, n_ok, a4);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);   // passes stay in order: no hoisting of the next pass's loads (register pressure)
      }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// Tuning builds (scripts/ablate_gemm.sh): -DDXA_PPV=<bits> removes parts of the main loop (results are garbage, timings are
// not): 1 no LDS-DMA in the loop, 2 no ds_read, 4 no MFMA, 8 s_setprio around the MFMA clusters, 16 no stagger,
// 32 no epilogue stores.
#ifndef DXA_PPV
#define DXA_PPV 0
#endif
// origin of tile t of the grouped order (group_m row tiles x all column tiles per group)
__device__ __forceinline__ void sk_tile_origin(const GemmP& p, int t, int& m0, int& n0) {
  const int width = p.group_m * p.tn;
  const int group = t / width;
  const int first_pm = group * p.group_m;
  const int gsz = min(p.tm - first_pm, p.group_m);
  const int rem = t - group * width;
  const int pn = rem / gsz;
  m0 = (first_pm + rem - pn * gsz) * 256;
  n0 = pn * 256;
}
template <typename T> struct SkIO;      // 16-byte / 8-byte buffer accesses of CPL consecutive elements
template <> struct SkIO<float> {
  template <int CPL>
  static __device__ __forceinline__ void ld(float (&o)[CPL], __amdgpu_buffer_rsrc_t r, uint32_t off) {
    static_assert(CPL == 4, "fp32 operands come 4 columns per lane");
    const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e];
  }
  template <int CPL>
  static __device__ __forceinline__ void st(const float (&v)[CPL], __amdgpu_buffer_rsrc_t r, uint32_t off) {
    static_assert(CPL == 4, "fp32 outputs go 4 columns per lane");
    const f32x4_t o = {v[0], v[1], v[2], v[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), r, off, 0, 0);
  }
};
template <> struct SkIO<bf16_t> {
  template <int CPL>
  static __device__ __forceinline__ void ld(float (&o)[CPL], __amdgpu_buffer_rsrc_t r, uint32_t off) {
    if constexpr (CPL == 8) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(v[e] << 16); o[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
    } else {
      typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
#pragma unroll
      for (int e = 0; e < 2; ++e) { o[2 * e] = __uint_as_float(v[e] << 16); o[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
    }
  }
  template <int CPL>
  static __device__ __forceinline__ void st(const float (&v)[CPL], __amdgpu_buffer_rsrc_t r, uint32_t off) {
    static_assert(CPL == 8, "bf16 outputs go 8 columns per lane");
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
    __builtin_amdgcn_raw_buffer_store_b128(o, r, off, 0, 0);
  }
};

// fused epilogue of one 256x256 tile from the accumulators: C = alpha * acc + bias (+ residual) (+ C).  Products with an
// activation, a mulgrad operand or an aux output (ViT / projector MLPs: ~4 % of the step's FLOPs) stay on the ring kernel,
// whose epilogue carries the whole menu.
// sk_epilogue_rows: the 128 x 64 block of one wave (rows m0 + 128 wm + ..., columns n0 + 64 wn + ...); returns the lane's share of
// the sum of squares of what it stored.  sk_epilogue: the 8-wave (2 x 4) kernels' tile = one such block per wave + the fold.
template <typename TO, typename TE, int AI = 4>
__device__ __forceinline__ float sk_epilogue_rows(const GemmP& p, f32x16_t (&acc)[AI][2], char* slab, int lane, int wm, int wn,
                                                  int m0, int n0) {
  float ssq = 0.f;                                   // sum of squares of the values this lane stores (p.sumsq)
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CPL = 16 / (int)sizeof(TO);        // columns per lane: 16 bytes of output
  constexpr int LPR = 64 / CPL;                    // lanes per 64-column row (8 | 16)
  constexpr int RPI = 64 / LPR;                    // rows per wave instruction (8 | 4)
  const int l32 = lane & 31, lh = lane >> 5, r16 = l32 & 15;
  const int lr = lane / LPR, lc = (lane % LPR) * CPL;
  const int n = n0 + wn * 64 + lc;
  const int rowb = m0 + wm * (32 * AI) + lr;       // + 32 i + 16 hh + RPI it   (AI = 3: the 192-row tile's 96 x 64 block)
  const bool col_ok = n < (int)p.N;
  const uint32_t Mi = (uint32_t)p.M;
  const uint32_t esO = sizeof(TO), esE = sizeof(TE);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((p.M - 1) * p.ldc + p.N) * esO), 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.R ? p.R : p.A), 0, p.R ? (int)(((p.M - 1) * p.ldr + p.N) * esE) : 0, 0x00020000);
  const uint32_t ldcB = (uint32_t)p.ldc * esO, ldrB = (uint32_t)p.ldr * esE;
  const uint32_t offC0 = (uint32_t)rowb * ldcB + (uint32_t)n * esO;
  const uint32_t offR0 = (uint32_t)rowb * ldrB + (uint32_t)n * esE;
  float bv[CPL];
#pragma unroll
  for (int e = 0; e < CPL; ++e) bv[e] = 0.f;
  if (p.bias && col_ok) {
    const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.bias), 0, (int)(p.N * esE), 0x00020000);
    SkIO<TE>::template ld<CPL>(bv, rBias, (uint32_t)n * esE);
  }
  const bool has_R = p.R != nullptr, accum = p.accumulate != 0, has_mirror = p.mirror != nullptr;
  const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(p.mirror ? p.mirror : (char*)p.C, 0,
                                                                      p.mirror ? (int)(((p.M - 1) * p.ldc + p.N) * 2) : 0, 0x00020000);
  const float alpha = p.alpha;
  const bool want_ssq = p.sumsq != nullptr;
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if ((l32 >> 4) == hh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            *reinterpret_cast<float4*>(slab + r16 * 256 + (((j * 8 + 2 * q + lh) ^ r16) << 4)) = v;
          }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 16 / RPI; ++it) {
        const int row = lr + RPI * it;                         // row of the 16-row slab
        const uint32_t rr = (uint32_t)(32 * i + 16 * hh + RPI * it);
        const bool ok = col_ok && (uint32_t)rowb + rr < Mi;
        float v[CPL];
#pragma unroll
        for (int u = 0; u < CPL / 4; ++u) {
          const float4 t = *reinterpret_cast<const float4*>(slab + row * 256 + (((lc / 4 + u) ^ row) << 4));
          v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
        }
        const uint32_t oc = ok ? offC0 + rr * ldcB : 0x80000000u;
#pragma unroll
        for (int e = 0; e < CPL; ++e) v[e] = v[e] * alpha + bv[e];
        if (has_R) {
          float r[CPL];
          SkIO<TE>::template ld<CPL>(r, rR, ok ? offR0 + rr * ldrB : 0x80000000u);
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] += r[e];
        }
        if (accum) {
          float c0[CPL];
          SkIO<TO>::template ld<CPL>(c0, rC, oc);
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] += c0[e];
        }
#if !(DXA_PPV & 32)
        SkIO<TO>::template st<CPL>(v, rC, oc);
        if constexpr (sizeof(TO) == 2) {
          if (want_ssq && ok) {                                // bf16 gradient arena: the norm is taken over what is stored
#pragma unroll
            for (int e = 0; e < CPL; ++e) { const float t = rnd<bf16_t>(v[e]); ssq += t * t; }
          }
        }
        if constexpr (sizeof(TO) == 4) {
          if (ok) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) ssq += v[e] * v[e];
          }
          if (has_mirror) {                                    // bf16 communication copy: 8 bytes per lane, same rows
            typedef uint32_t u32x2_t_ __attribute__((ext_vector_type(2)));
            const u32x2_t_ o2 = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            __builtin_amdgcn_raw_buffer_store_b64(o2, rM, ok ? (offC0 + rr * ldcB) >> 1 : 0x80000000u, 0, 0);
          }
        }
#endif
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);   // passes stay in order: no hoisting of the next pass's work (register pressure)
    }
#endif
  return ssq;
}
// DXA_FUSE_SWIGLU epilogue of one tile (bf16 out).  The operand loads gave the wave's 64 tile columns the meaning
// [32 gate | 32 up] of the SAME 32 outputs  (n0 / 2 + 32 wn + c, c = 0..31): a lane takes gate and up of 4 outputs from the slab
// (16-byte units lq and 8 + lq of its row), rounds both to bf16 — the pre-activations as dxa_gemm would have stored them —
// and stores out = bf16(bf16(silu(g)) * u) (elementwise.hip swiglu_fwd_k, bit for bit) plus, if asked, the two pre-activation
// pieces: 8-byte stores, 8 lanes = one 64-byte row segment.
template <int AI = 4>
__device__ __forceinline__ void sk_epilogue_swiglu(const GemmP& p, f32x16_t (&acc)[AI][2], char* slab, int lane, int wm, int wn,
                                                   int m0, int n0) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int l32 = lane & 31, lh = lane >> 5, r16 = l32 & 15;
  const int lr = lane >> 3, lq = lane & 7;                 // row of the 8-row instruction, 4-output group
  const int F = (int)(p.N >> 1);
  const int nout = (n0 >> 1) + wn * 32 + lq * 4;           // first of the lane's 4 outputs
  const int rowb = m0 + wm * (32 * AI) + lr;
  const bool col_ok = nout < F;
  const uint32_t Mi = (uint32_t)p.M;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((p.M - 1) * p.ldc + F) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(p.aux ? p.aux : p.C, 0,
                                                                      p.aux ? (int)(((p.M - 1) * p.ldaux + p.N) * 2) : 0, 0x00020000);
  const uint32_t ldcB = (uint32_t)p.ldc * 2u, ldxB = (uint32_t)p.ldaux * 2u;
  const uint32_t offC0 = (uint32_t)rowb * ldcB + (uint32_t)nout * 2u;
  const uint32_t offG0 = (uint32_t)rowb * ldxB + (uint32_t)nout * 2u, offU0 = offG0 + (uint32_t)F * 2u;
  const bool has_aux = p.aux != nullptr;
  const float alpha = p.alpha;
  typedef uint32_t u32x2_t_ __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if ((l32 >> 4) == hh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            *reinterpret_cast<float4*>(slab + r16 * 256 + (((j * 8 + 2 * q + lh) ^ r16) << 4)) = v;
          }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = lr + 8 * it;                           // row of the 16-row slab
        const uint32_t rr = (uint32_t)(32 * i + 16 * hh + 8 * it);
        const bool ok = col_ok && (uint32_t)rowb + rr < Mi;
        const float4 g4 = *reinterpret_cast<const float4*>(slab + row * 256 + ((lq ^ row) << 4));
        const float4 u4 = *reinterpret_cast<const float4*>(slab + row * 256 + (((8 + lq) ^ row) << 4));
        const float g[4] = {rnd<bf16_t>(g4.x * alpha + 0.f), rnd<bf16_t>(g4.y * alpha + 0.f), rnd<bf16_t>(g4.z * alpha + 0.f), rnd<bf16_t>(g4.w * alpha + 0.f)};
        const float u[4] = {rnd<bf16_t>(u4.x * alpha + 0.f), rnd<bf16_t>(u4.y * alpha + 0.f), rnd<bf16_t>(u4.z * alpha + 0.f), rnd<bf16_t>(u4.w * alpha + 0.f)};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (fast_sigmoidf — v_exp_f32 + v_rcp_f32 — where swiglu_fwd_k has expf and an IEEE division: the SiLUs of a product are VALU
          //  work nothing overlaps in an epilogue, one workgroup per CU; against swiglu_fwd_k a result may differ by one bf16 step in a
          //  few values per million, none on the test data)
          o[e] = rnd<bf16_t>(g[e] * fast_sigmoidf(g[e])) * u[e];
        }
        const u32x2_t_ ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        __builtin_amdgcn_raw_buffer_store_b64(ov, rC, ok ? offC0 + rr * ldcB : 0x80000000u, 0, 0);
        if (has_aux) {
          const u32x2_t_ gv = {pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3])};
          const u32x2_t_ uv = {pack_bf16x2(u[0], u[1]), pack_bf16x2(u[2], u[3])};
          __builtin_amdgcn_raw_buffer_store_b64(gv, rX, ok ? offG0 + rr * ldxB : 0x80000000u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(uv, rX, ok ? offU0 + rr * ldxB : 0x80000000u, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
}
template <typename TO, typename TE, int AI = 4>
__device__ __forceinline__ void sk_epilogue(const GemmP& p, f32x16_t (&acc)[AI][2], char* slab, int lane, int wm, int wn,
                                            int m0, int n0, float* red, int tile) {
#if defined(__HIP_DEVICE_COMPILE__)
  float ssq = sk_epilogue_rows<TO, TE, AI>(p, acc, slab, lane, wm, wn, m0, n0);
  {
    // global-norm clip: this tile's share of sum(g^2), folded lane -> wave -> workgroup in a fixed order and written to
    // the tile's own slot (the host adds the slots in index order): the separate 30 GB pass over the gradient arena that
    // used to read every dW back is not needed for gradients a single product writes
    if (p.sumsq != nullptr) {                  // uniform over the workgroup
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ssq += __shfl_xor(ssq, o, 64);
      if (lane == 0) red[wm * 4 + wn] = ssq;
      __syncthreads();
      if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) t += red[w8];
        p.sumsq[tile] = t;
      }
    }
  }
#endif
}
}  // namespace
