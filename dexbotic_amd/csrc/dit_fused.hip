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

#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace {

constexpr int MAXM = 48;     // rows (CFG batch 2 x 17 tokens = 34)
constexpr int MB = 3;        // 16-row blocks
constexpr int MAXT = 32;     // tokens per sample
constexpr int HD = 64;       // head width

constexpr int SMAX = 8;      // K slices of the narrow products (proj, fc2)

struct DitP {
  float *h, *qkv, *o, *a, *part;   // part [SMAX][M][H]: partial sums of the K-sliced products
  const float* const* w;           // [depth][8]: qkv_w, qkv_b, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b
  unsigned *bar, *cnt_proj, *cnt_fc2;
  int M, N, T1, H, heads, I, depth;
  int s_proj, s_fc2;
  float eps, scale;
  int dbg;   // tuning aid (wrong results): 1 = barriers only, 2 = work only, 3 = no attention, 5 = no weight loads, 6 = no activation loads
};

struct alignas(16) Smem {
  float red[8][MB][64][4];
  float sx[8][MAXM], sxx[8][MAXM], wsum[16];   // LayerNorm: per-wave row sums / sums of squares, sum_k W[n,k]
  float q[MAXT][HD], k[MAXT][HD + 4], v[MAXT][HD], p[MAXT][MAXT + 1];
  int last;
#if defined(DXA_DIT_STAMPS)
  unsigned long long stamp[6][8], ts[8];    // tuning build: cycles per segment of each phase type, summed by wave 0 of workgroup DXA_DIT_STAMPS
#endif
};

// Tuning build (-DDXA_DIT_STAMPS=<workgroup>): s_memtime at fixed points of every phase, taken by wave 0 of one workgroup and summed
// per phase type (0 qkv, 1 attention, 2 proj, 3 fc1, 4 fc2) in LDS; dxa_dit_debug_stamps() copies the sums out.  Segments of a product
// phase: 0 entry -> operands loaded and MFMAs retired, 1 -> partials of all waves in LDS, 2 -> epilogue stores issued,
// 3 -> stores acknowledged (vmcnt(0)) and the workgroup assembled, 4 -> device-wide barrier left.
#if defined(DXA_DIT_STAMPS)
__device__ unsigned long long g_dit_stamps[6][8];
#define DIT_STAMP(s_, i_) do { if (blockIdx.x == DXA_DIT_STAMPS && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); (s_).ts[i_] = t_; } } while (0)
#define DIT_STAMP_AFTER(s_, i_, v_) do { if (blockIdx.x == DXA_DIT_STAMPS && threadIdx.x == 0) { unsigned long long t_; const unsigned d_ = __builtin_amdgcn_readfirstlane(__float_as_uint(v_)); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "s"(d_) : "memory"); (s_).ts[i_] = t_; } } while (0)
#define DIT_STAMP_FOLD(s_, ph_, n_) do { if (blockIdx.x == DXA_DIT_STAMPS && threadIdx.x == 0) { for (int i_ = 0; i_ < (n_); ++i_) (s_).stamp[ph_][i_] += (s_).ts[i_ + 1] - (s_).ts[i_]; (s_).stamp[ph_][7] += 1; } } while (0)
#else
#define DIT_STAMP(s_, i_) do { } while (0)
#define DIT_STAMP_AFTER(s_, i_, v_) do { } while (0)
#define DIT_STAMP_FOLD(s_, ph_, n_) do { } while (0)
#endif

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// Activations travel between workgroups on different XCDs, whose L2s are not coherent with each other: every
// activation access is an agent-scope (sc1) buffer access — stores write through, loads miss in the private caches —
// so the barrier needs no cache-wide write-back / invalidate and the weights stay cached.
constexpr int SC1 = 16;
#if defined(DXA_DIT_CACHED_LOADS)
constexpr int A_AUX = 0;      // tuning build: the A operand through the XCD's L2
#else
constexpr int A_AUX = SC1;
#endif
struct Act {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ Act(float* p, size_t floats)
      : r(__builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(floats * sizeof(float)), 0x00020000)) {}
  // coherent load: data written by other workgroups since the last device-wide barrier (partials, residual)
  __device__ __forceinline__ float4 ld4(size_t idx) const {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * 4), 0, SC1);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  }
  // the same by BYTE offset; an offset beyond the descriptor (0x80000000) reads zeros: branch-free loads of ragged tiles
  __device__ __forceinline__ float4 ld4_or0(uint32_t byte_off) const {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, SC1);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  }
  // cached load: data complete before the last device-wide barrier (whose acquire dropped stale lines); the 20+
  // workgroups of an XCD then share one copy in its L2 instead of each pulling the matrix from the memory side
  __device__ __forceinline__ float4 ld4c(size_t idx) const {
#if !defined(DXA_DIT_CACHED_LOADS)
    return ld4(idx);     // measured: the cached variant (+ acquire at every barrier) is 8 % slower end to end
#else
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * 4), 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
#endif
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

// Device-wide barrier over workgroups that must all be resident.  The launch is a plain one sized by the occupancy query
// (a cooperative launch adds 15-19 us per forward and enforces nothing more, MI355X_MICROARCH.md "coop-launch"), so
// co-residency holds only while nothing else occupies CUs for long: the spin is BOUNDED (~2 s).  A workgroup that gives up
// raises a sticky abort word; every later barrier of every workgroup then falls through, the launch drains in
// microseconds with a garbage result, and dxa_dit_blocks_status() reports it to the host, which re-runs the request on
// the unfused path (persistent kernel needs a watchdog).
constexpr unsigned SPIN_LIMIT = 1u << 21;
// Round 4: SIXTEEN arrival counters, one 4 KiB apart from the next (bar + 1024 (g + 1) words), workgroup b arrives at counter b % 16
// with a no-return atomic and lanes 0-15 of wave 0 poll one counter each until every counter shows its whole group.  The single
// counter + flag of rounds 1-3 cost 1.0 us + 10 ns per WORKGROUP (2.97 us at 192, 1.06 us at 8: `profiles/r04_barrier_vs_grid.txt`) —
// device-scope atomics on one address are applied one after the other at the memory side, and the flag hop is a second dependent
// round trip behind them; spread over 16 lines in 16 places the same arrivals take 1.27 us and nobody waits for a publisher
// (`scripts/probes/sync_probe.hip` modes 4 / 16: `profiles/r04_barrier_split.txt`).
constexpr unsigned NCTR = 16;
__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned nblk, unsigned& epoch, bool sleep, Smem* sst = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores have been acknowledged
  __syncthreads();
  if (sst) DIT_STAMP(*sst, 4);
  epoch += 1;                                                // every thread keeps the count (wave 0's lanes need it)
  if (threadIdx.x < 64) {
    unsigned* abortw = bar + 56;
    if (threadIdx.x == 0)
      (void)__hip_atomic_fetch_add(bar + 1024u * (blockIdx.x % NCTR + 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned g = threadIdx.x % NCTR;
    const unsigned target = epoch * ((nblk + NCTR - 1u - g) / NCTR);      // workgroups b < nblk with b % NCTR == g, `epoch` times
    const unsigned* mine = bar + 1024u * (g + 1u);
    unsigned spins = 0;
    while (true) {
      const unsigned v = threadIdx.x < NCTR ? __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
      if (__builtin_amdgcn_ballot_w64(v < target) == 0ull) break;
      if (sleep) __builtin_amdgcn_s_sleep(1);
      if ((++spins & 15u) == 0u) {            // the abort word: every 16th poll (a launch that was aborted drains in milliseconds)
        if (spins >= SPIN_LIMIT && threadIdx.x == 0) __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins >= SPIN_LIMIT) break;
      }
    }
  }
  __syncthreads();
}

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESADD = 2 };

// C[M, Nout] = epi(pro(A)[M, K] W[Nout, K]^T + bias).  A, C are activations (sc1 accesses), W, bias weights (cached).
//  * LN: pro(a) = (a - mu[m]) * rs[m] is applied in the EPILOGUE, rs[m] * (acc[m,n] - mu[m] * wsum[n]), with
//    wsum[n] = sum_k W[n,k] picked up by the product itself from an all-ones row appended to A (row M < 48) and the
//    row statistics accumulated from the very A fragments the waves load for the product (each row is covered
//    exactly once by the 8 waves): no separate pass over h, nothing waits for the statistics.
//  * S > 1 (narrow products): the work items are (16 columns) x (K slice); a slice's partial tile goes to `part`
//    through write-through stores and the LAST workgroup to arrive at the tile's counter (split-K protocol of the ring
//    GEMM) adds the S partials in slice order, the bias and the residual: deterministic, no extra device-wide barrier.
template <bool LN, int EPI, int U>
__device__ __forceinline__ void gemm_phase_u(const DitP& p, Smem& s, const Act& A, int lda, const float* __restrict__ W,
                                           const float* __restrict__ bias, const Act& C, int ldc, int Nout, int K, int S,
                                           unsigned* cnt, unsigned cnt_target, const Act& part) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: uniform branches
  const int l16 = lane & 15, lg = lane >> 4;
  // Operand loads are BRANCH-FREE buffer loads (round 4): a row that does not exist (padding of the last 16-row block, the LN
  // ones row) and a K piece past the end of this wave's share get an offset outside the descriptor, which returns zeros.  The
  // round-3 form (per-lane `if (row valid) load`) compiled to a branch around every load with an s_waitcnt vmcnt(0) after every
  // fourth one and flat loads for W: 4-5 serialised memory round trips, 20,000 cycles from the barrier to the last MFMA of a
  // K = 768 product whose MFMAs need 3,100 (profiles/r04_dit_stamps_before.txt).
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t aoff[MB];        // byte offset of this lane's 16-byte piece in its A row, or OOB
  bool ones[MB];            // the LN ones row (m == M): sum_k W[n, k] comes out of the product itself
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 16 + l16;
    aoff[mb] = m < p.M ? (uint32_t)(((size_t)m * lda + 4 * lg) * sizeof(float)) : OOB;
    ones[mb] = LN && m == p.M;
  }
  // descriptors in scalar registers (the pointers come out of the weight table: tell the compiler they are wave-uniform, or every
  // buffer load is wrapped in a waterfall loop)
  auto uniform_ptr = [](const float* q) {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t Wr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(W), 0, (int)((size_t)Nout * K * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t Br = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(bias), 0, Nout * (int)sizeof(float), 0x00020000);
  // K is walked in 32-deep pieces: piece hb of a slice goes to wave hb % 8, U pieces per wave and trip (K = 768: 24 pieces, U = 3, one
  // trip, every wave exactly three — round 3 dealt 64-deep blocks 2,2,2,2,1,1,1,1).  A piece past the end of the slice is loaded as
  // zeros and multiplied like any other: no branch sits between a load and its use, so nothing tempts the compiler to sink a load
  // behind the products of the pieces before it (it did, with a wave-uniform `continue` here).  U is picked by the caller from the
  // slice length so that dead pieces only occur in ragged shapes.
  const int ncb = Nout / 16, nhb = K / 32, per = nhb / S;
  for (int item = blockIdx.x; item < ncb * S; item += gridDim.x) {
    const int cb = item % ncb, ks = item / ncb, n0 = cb * 16;
    const int hb_lo = ks * per, hb_hi = hb_lo + per;
    const uint32_t woff = (uint32_t)(((size_t)(n0 + l16) * K + 4 * lg) * sizeof(float));
    // epilogue operands of the folding waves are requested first
    const int em = wave * 16 + l16, en = n0 + 4 * lg;
    const bool erow = wave < MB && em < p.M;
    const u32x4_t b4u = __builtin_amdgcn_raw_buffer_load_b128(Br, en * 4, 0, 0);
    const float4 b4 = make_float4(__uint_as_float(b4u[0]), __uint_as_float(b4u[1]), __uint_as_float(b4u[2]), __uint_as_float(b4u[3]));
    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_RESADD && S == 1) c4 = C.ld4_or0(erow ? (uint32_t)(((size_t)em * ldc + en) * sizeof(float)) : OOB);
    f32x4_t acc[MB];
    float sx[MB], sxx[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) { acc[mb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sx[mb] = 0.f; sxx[mb] = 0.f; }
    for (int base = hb_lo; base < hb_hi; base += 8 * U) {
      u32x4_t wv[U][2], av[U][MB][2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int hb = base + wave + 8 * u;
        const bool live = hb < hb_hi;
        const uint32_t koff = (uint32_t)hb * 128u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wv[u][j] = __builtin_amdgcn_raw_buffer_load_b128(Wr, (int)(live && p.dbg != 5 ? woff + koff + 64u * j : OOB), 0, 0);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            av[u][mb][j] = __builtin_amdgcn_raw_buffer_load_b128(A.r, (int)(live && aoff[mb] != OOB && p.dbg != 6 ? aoff[mb] + koff + 64u * j : OOB), 0, A_AUX);
        }
      }
      __builtin_amdgcn_sched_barrier(0);        // every load of the trip is in flight before the first MFMA (the scheduler otherwise
                                                // threads the loads through the MFMAs with an s_waitcnt vmcnt(0) behind each group)
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float a[MB][4];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              a[mb][c] = ones[mb] ? 1.f : __uint_as_float(av[u][mb][j][c]);
              if (LN) { sx[mb] += a[mb][c]; sxx[mb] += a[mb][c] * a[mb][c]; }
            }
          // the three accumulators alternate: a dependent MFMA is three issues (96 cycles) behind its predecessor, not back to back
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
              acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wv[u][j][c]), a[mb][c], acc[mb], 0, 0, 0);
        }
      }
    }
    DIT_STAMP_AFTER(s, 1, acc[0][0] + acc[1][0] + acc[2][0]);
    if (LN) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {       // the 4 lanes l16 + 16 lg hold disjoint columns of row mb*16 + l16
        float a = sx[mb], b = sxx[mb];
        a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
        a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
        if (lg == 0) { s.sx[wave][mb * 16 + l16] = a; s.sxx[wave][mb * 16 + l16] = b; }
      }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      *reinterpret_cast<float4*>(s.red[wave][mb][lane]) = make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
    __syncthreads();
    DIT_STAMP(s, 2);
    // wave mb folds the 8 partials of row block mb: lane holds out[m = 16 mb + l16][n0 + 4 lg + {0..3}]
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave < MB) {
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) {
        const float4 v = *reinterpret_cast<const float4*>(s.red[w8][wave][lane]);
        r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
      }
      if (LN && em == p.M) *reinterpret_cast<float4*>(&s.wsum[4 * lg]) = r;     // the ones row: sum_k W[n, k]
    }
    if (LN) __syncthreads();
    if (S == 1) {
      if (wave < MB && em < p.M) {
        if (LN) {
          const float4 ws = *reinterpret_cast<const float4*>(&s.wsum[4 * lg]);
          float tx = 0.f, txx = 0.f;
#pragma unroll
          for (int w8 = 0; w8 < 8; ++w8) { tx += s.sx[w8][em]; txx += s.sxx[w8][em]; }
          const float mu = tx / (float)K;
          const float rs = rsqrtf(fmaxf(txx / (float)K - mu * mu, 0.f) + p.eps);
          r.x = rs * (r.x - mu * ws.x); r.y = rs * (r.y - mu * ws.y);
          r.z = rs * (r.z - mu * ws.z); r.w = rs * (r.w - mu * ws.w);
        }
        r.x += b4.x; r.y += b4.y; r.z += b4.z; r.w += b4.w;
        if (EPI == EPI_GELU) {
          r.x = act_fwd(DXA_ACT_GELU_TANH, r.x); r.y = act_fwd(DXA_ACT_GELU_TANH, r.y);
          r.z = act_fwd(DXA_ACT_GELU_TANH, r.z); r.w = act_fwd(DXA_ACT_GELU_TANH, r.w);
        } else if (EPI == EPI_RESADD) {
          r.x += c4.x; r.y += c4.y; r.z += c4.z; r.w += c4.w;
        }
        C.st4((size_t)em * ldc + en, r);
      }
    } else {
      // K-sliced (never LN): publish the partial tile, the last arrival gathers
      if (wave < MB && em < p.M) part.st4(((size_t)ks * p.M + em) * Nout + en, r);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned arrived = __hip_atomic_fetch_add(cnt + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        s.last = arrived == cnt_target;
      }
      __syncthreads();
      if (s.last && wave < MB && em < p.M) {
        float4 t = b4;
        if (EPI == EPI_RESADD) {
          const float4 c0 = C.ld4((size_t)em * ldc + en);
          t.x += c0.x; t.y += c0.y; t.z += c0.z; t.w += c0.w;
        }
        float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < S; ++q) {
          const float4 v = part.ld4(((size_t)q * p.M + em) * Nout + en);
          acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
        }
        t.x += acc4.x; t.y += acc4.y; t.z += acc4.z; t.w += acc4.w;
        C.st4((size_t)em * ldc + en, t);
      }
    }
    __syncthreads();
  }
}

template <bool LN, int EPI>
__device__ __forceinline__ void gemm_phase(const DitP& p, Smem& s, const Act& A, int lda, const float* __restrict__ W,
                                           const float* __restrict__ bias, const Act& C, int ldc, int Nout, int K, int S,
                                           unsigned* cnt, unsigned cnt_target, const Act& part) {
  const int per = K / 32 / S;             // 32-deep pieces per slice, dealt to 8 waves
  if (per <= 8) gemm_phase_u<LN, EPI, 1>(p, s, A, lda, W, bias, C, ldc, Nout, K, S, cnt, cnt_target, part);
  else if (per <= 16) gemm_phase_u<LN, EPI, 2>(p, s, A, lda, W, bias, C, ldc, Nout, K, S, cnt, cnt_target, part);
  else gemm_phase_u<LN, EPI, 3>(p, s, A, lda, W, bias, C, ldc, Nout, K, S, cnt, cnt_target, part);
}

// one workgroup per (sample, head): T1 x T1 scores from LDS copies of q, k, v.  Round 4: every loop over the tokens or the head width
// is a fixed-length, fully unrolled run of 16-byte LDS reads and the softmax of a row lives in the 32 lanes that computed its
// scores — the round-3 form (runtime-length scalar loops, 17 threads doing the softmax) took 20,000 cycles per phase, as long as a
// K = 768 product (profiles/r04_dit_stamps_before.txt).
__device__ __forceinline__ void attention_phase(const DitP& p, Smem& s, const Act& qkv, const Act& o) {
  const int tid = threadIdx.x, T1 = p.T1, ld = 3 * p.H;
  constexpr uint32_t OOB = 0x80000000u;
  for (int pr = blockIdx.x; pr < p.N * p.heads; pr += gridDim.x) {
    const int n = pr / p.heads, hd = pr - n * p.heads;
    const size_t base = (size_t)n * T1 * ld + hd * HD;
    // q, k, v rows of this head: 3 x T1 x 16 pieces of 16 bytes, two per thread at most (T1 <= 32: 1536 pieces), all requested
    // before the first is used; rows T1 .. 31 of v are zero-filled (the P V loop runs over all 32 keys)
    float4 ld_[3];
#pragma unroll
    for (int r3 = 0; r3 < 3; ++r3) {
      const int e = tid + 512 * r3, which = e / (MAXT * 16), rem = e - which * (MAXT * 16), i = rem >> 4, d = (rem & 15) * 4;
      ld_[r3] = qkv.ld4_or0(i < T1 ? (uint32_t)((base + (size_t)i * ld + (size_t)which * p.H + d) * sizeof(float)) : OOB);
    }
#pragma unroll
    for (int r3 = 0; r3 < 3; ++r3) {
      const int e = tid + 512 * r3, which = e / (MAXT * 16), rem = e - which * (MAXT * 16), i = rem >> 4, d = (rem & 15) * 4;
      float* dst = which == 0 ? &s.q[i][d] : (which == 1 ? &s.k[i][d] : &s.v[i][d]);
      *reinterpret_cast<float4*>(dst) = ld_[r3];
    }
    __syncthreads();
    // scores + softmax: half-wave h = tid / 32 owns query rows h and h + 16, lane j = tid % 32 the key
    const int j = tid & 31;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int i = (tid >> 5) + 16 * pass;                      // half-wave uniform
      if (i >= T1) continue;                                      // 17 tokens: the second pass is one row of one half-wave
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 q4 = *reinterpret_cast<const float4*>(&s.q[i][d]), k4 = *reinterpret_cast<const float4*>(&s.k[j][d]);
        sc += (q4.x * k4.x + q4.y * k4.y) + (q4.z * k4.z + q4.w * k4.w);
      }
      sc = j < T1 ? sc * p.scale : -INFINITY;
      float mx = sc;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float e = j < T1 ? expf(sc - mx) : 0.f;
      float sum = e;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
      if (i < T1) s.p[i][j] = e / sum;                           // keys T1 .. 31: exact zeros
    }
    __syncthreads();
    // out[i][d .. d+3] = sum_j p[i][j] v[j][d .. d+3]: one 16-byte piece per thread
    {
      const int i = tid >> 4, d = (tid & 15) * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const int T1r = (T1 + 3) & ~3;                 // keys T1 .. T1r-1: p is an exact zero, v a zero row
#pragma unroll 4
      for (int jj = 0; jj < T1r; ++jj) {
        const float pj = s.p[i][jj];
        const float4 v4 = *reinterpret_cast<const float4*>(&s.v[jj][d]);
        acc.x += pj * v4.x; acc.y += pj * v4.y; acc.z += pj * v4.z; acc.w += pj * v4.w;
      }
      if (i < T1) o.st4((size_t)n * T1 * p.H + hd * HD + (size_t)i * p.H + d, acc);
    }
    __syncthreads();
  }
}

// the `depth` blocks of one denoising call; `base` = blocks walked by this launch before (the split-K tile counters count on)
__device__ __forceinline__ void walk_blocks(const DitP& p, Smem& s, unsigned& epoch, unsigned nblk, unsigned base, const Act& h,
                                            const Act& qkv, const Act& o, const Act& a, const Act& part) {
  const bool work = p.dbg != 1, sync = p.dbg != 2;
  for (int blk = 0; blk < p.depth; ++blk) {
    const float* const* w = p.w + blk * 8;
    DIT_STAMP(s, 0);
    if (work) gemm_phase<true, EPI_BIAS>(p, s, h, p.H, w[0], w[1], qkv, 3 * p.H, 3 * p.H, p.H, 1, nullptr, 0, part);
    DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, p.dbg != 4, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 0, 5);
    DIT_STAMP(s, 0);
    if (work && p.dbg != 3) attention_phase(p, s, qkv, o);
    DIT_STAMP(s, 1); DIT_STAMP(s, 2); DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, p.dbg != 4, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 1, 5);
    DIT_STAMP(s, 0);
    if (work) gemm_phase<false, EPI_RESADD>(p, s, o, p.H, w[2], w[3], h, p.H, p.H, p.H, p.s_proj, p.cnt_proj,
                                            (base + (unsigned)blk + 1u) * p.s_proj, part);
    DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, p.dbg != 4, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 2, 5);
    DIT_STAMP(s, 0);
    if (work) gemm_phase<true, EPI_GELU>(p, s, h, p.H, w[4], w[5], a, p.I, p.I, p.H, 1, nullptr, 0, part);
    DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, p.dbg != 4, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 3, 5);
    DIT_STAMP(s, 0);
    if (work) gemm_phase<false, EPI_RESADD>(p, s, a, p.I, w[6], w[7], h, p.H, p.H, p.I, p.s_fc2, p.cnt_fc2,
                                            (base + (unsigned)blk + 1u) * p.s_fc2, part);
    DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, p.dbg != 4, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 4, 5);
  }
}

// Leave the counters zeroed for the next launch on this stream (see the comment at the end of dit_blocks_fused_k)
__device__ __forceinline__ void leave_clean(const DitP& p, unsigned nblk) {
  if (threadIdx.x == 0) {
    unsigned* exit_cnt = p.bar + 48;
    const unsigned out = __hip_atomic_fetch_add(exit_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (out == nblk) {
      for (int i = 0; i < 64; ++i) {
        __hip_atomic_store(p.cnt_proj + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.cnt_fc2 + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (unsigned g = 0; g < NCTR; ++g)
        __hip_atomic_store(p.bar + 1024u * (g + 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(exit_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The WHOLE sampler in one launch (reference: the loop of GaussianDiffusion.ddim_sample_loop, diffusion.py:714-794, around
// DiT.forward_with_cfg, dit.py:273-311): per DDIM step
//   assemble   h = [t_emb + z_emb ; x W_x^T + b_x] + pos          (x_embedder, dit.py:106-135; the conditioning token, :281-286)
//   blocks     the `depth` DiTBlocks (walk_blocks)
//   final      eps_hat = LN(h) W_f^T + b_f on the action tokens    (FinalLayer, dit.py:165-178)
//   update     eps = u + s (c - u);  x <- ddim(x, eps)              (forward_with_cfg :294-311 + ddim_sample :626-673, eta = 0)
// The embeddings that do not depend on x are precomputed by the caller: z_emb [N, H] (constant over the steps) and t_emb
// [steps, H] (the schedule is known).  x lives in global memory (sc1 accesses: workgroup 0 writes it, everybody reads it).
struct DitSampleP {
  DitP blk;
  float* x;                  // [nb, T, A] in/out
  const float* ze;           // [N, H]
  const float* te;           // [steps, H]
  const float* pos;          // [T1, H]
  const float* xw; const float* xb;    // x_embedder: [H, A], [H]
  const float* fw; const float* fb;    // final linear: [A, H], [A]
  const float* coef;         // [steps][4]: sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, alphas_cumprod_prev, -
  float* xpp;                // [2][nb, T, A] scratch: x of the current / previous step
  float* eps;                // [M][MAXA] scratch: the network output rows of the step
  int steps, A, nb, use_cfg;
  float cfg_scale;
};
constexpr int MAXA = 8;      // action_dim (7)

__global__ __launch_bounds__(512) void dit_sample_fused_k(const DitSampleP sp) {
  __shared__ Smem s;
  __shared__ float xs[MAXM * MAXA];              // this step's x, every workgroup's own copy
  const DitP& p = sp.blk;
  unsigned epoch = 0;
  const unsigned nblk = gridDim.x;
  const int T = p.T1 - 1, A = sp.A, H = p.H, nx = sp.nb * T * A;
  const Act h(p.h, (size_t)p.M * p.H), qkv(p.qkv, (size_t)p.M * 3 * p.H), o(p.o, (size_t)p.M * p.H), a(p.a, (size_t)p.M * p.I),
      part(p.part, (size_t)SMAX * p.M * p.H), xin(sp.x, (size_t)nx), xpp(sp.xpp, (size_t)2 * nx), epsg(sp.eps, (size_t)p.M * MAXA);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // x of step `step` = ddim(x of step - 1, eps of step - 1): recomputed by EVERY workgroup into its LDS (3 x 112 coherent loads)
  // instead of one workgroup updating x and everybody waiting for one more device-wide barrier; workgroup 0 also keeps the
  // global copy (ping-pong: the others still read the previous one in this phase)
  auto load_x = [&](int step) {
    if (tid < nx) {
      float xv;
      if (step == 0) {
        xv = xin.ld1(tid);
      } else {
        const int k = tid % A, t = (tid / A) % T, n = tid / (A * T);
        float eps = epsg.ld1((size_t)(n * p.T1 + 1 + t) * MAXA + k);
        if (sp.use_cfg) {
          const float eu = epsg.ld1((size_t)((n + sp.nb) * p.T1 + 1 + t) * MAXA + k);
          eps = eu + sp.cfg_scale * (eps - eu);
        }
        const float c_recip = sp.coef[(step - 1) * 4], c_recipm1 = sp.coef[(step - 1) * 4 + 1], ab_prev = sp.coef[(step - 1) * 4 + 2];
        const float xo = xpp.ld1((size_t)((step - 1) & 1) * nx + tid);
        const float x0 = c_recip * xo - c_recipm1 * eps;
        const float eps2 = (c_recip * xo - x0) / c_recipm1;
        xv = x0 * sqrtf(ab_prev) + sqrtf(1.f - ab_prev - 0.f) * eps2;
      }
      xs[tid] = xv;
      if (blockIdx.x == 0) {
        if (step < sp.steps) xpp.st1((size_t)(step & 1) * nx + tid, xv);
        else xin.st1(tid, xv);                                   // the sample
      }
    }
    __syncthreads();
  };
  for (int step = 0; step < sp.steps; ++step) {
    load_x(step);
    // ---- assemble: every element of h once, spread over the grid
    for (int idx = blockIdx.x * 512 + tid; idx < p.M * H; idx += gridDim.x * 512) {
      const int m = idx / H, c = idx - m * H;
      const int n = m / p.T1, t = m - n * p.T1;
      float v;
      if (t == 0) {
        v = sp.te[(size_t)step * H + c] + sp.ze[(size_t)n * H + c];
      } else {
        const float* xr = xs + ((n % sp.nb) * T + (t - 1)) * A;          // CFG: both halves embed the first half of x
        v = sp.xb[c];
        for (int k = 0; k < A; ++k) v += xr[k] * sp.xw[(size_t)c * A + k];
      }
      h.st1(idx, v + sp.pos[(size_t)t * H + c]);
    }
    grid_sync(p.bar, nblk, epoch, true);
    walk_blocks(p, s, epoch, nblk, (unsigned)step * (unsigned)p.depth, h, qkv, o, a, part);
    // ---- final layer on the action tokens: one wave per row, rows spread over the workgroups
    for (int m = blockIdx.x * 8 + wave; m < p.M; m += gridDim.x * 8) {
      if (m % p.T1 == 0) continue;                                      // the conditioning token is dropped (dit.py:291)
      float v[16];                                                      // H <= 1024: 16 values per lane
      float sx = 0.f, sxx = 0.f;
      const int per = H / 64;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v[j] = 0.f;
        if (j < per) { v[j] = h.ld1((size_t)m * H + j * 64 + lane); sx += v[j]; sxx += v[j] * v[j]; }
      }
      sx = wave_sum(sx); sxx = wave_sum(sxx);
      const float mu = sx / (float)H;
      const float rs = rsqrtf(fmaxf(sxx / (float)H - mu * mu, 0.f) + p.eps);
      for (int k = 0; k < A; ++k) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (j < per) d += (v[j] - mu) * rs * sp.fw[(size_t)k * H + j * 64 + lane];
        d = wave_sum(d);
        if (lane == 0) epsg.st1((size_t)m * MAXA + k, d + sp.fb[k]);
      }
    }
    grid_sync(p.bar, nblk, epoch, true);
  }
  if (blockIdx.x == 0) load_x(sp.steps);                                  // the last update, written to x
  leave_clean(p, nblk);
}

__global__ __launch_bounds__(512) void dit_blocks_fused_k(const DitP p) {
  __shared__ Smem s;
  unsigned epoch = 0;
  const unsigned nblk = gridDim.x;
  const Act h(p.h, (size_t)p.M * p.H), qkv(p.qkv, (size_t)p.M * 3 * p.H), o(p.o, (size_t)p.M * p.H), a(p.a, (size_t)p.M * p.I),
      part(p.part, (size_t)SMAX * p.M * p.H);
#if defined(DXA_DIT_STAMPS)
  if (threadIdx.x < 48) s.stamp[threadIdx.x / 8][threadIdx.x % 8] = 0ull;
  __syncthreads();
#endif
  walk_blocks(p, s, epoch, nblk, 0u, h, qkv, o, a, part);
#if defined(DXA_DIT_STAMPS)
  __syncthreads();
  if (blockIdx.x == DXA_DIT_STAMPS && threadIdx.x < 48) g_dit_stamps[threadIdx.x / 8][threadIdx.x % 8] = s.stamp[threadIdx.x / 8][threadIdx.x % 8];
#endif
  // Leave the counters zeroed for the next launch on this stream (like the split-K flags of the ring GEMM): every
  // workgroup has passed the last barrier when it gets here, so the LAST one out may clear them.  Agent-scope atomic
  // stores, not a host-side memset: under HIP-graph replay a memset node's zeros were not reliably what the next
  // kernel's atomics saw (the sampler hung), atomics are performed at the memory side and always are.
  leave_clean(p, nblk);
}

// Exit counter, abort word and the two per-tile counter arrays (H / 16 <= 64 entries each) in the first KiB, the 16 arrival counters
// of the device-wide barrier 4 KiB apart behind it: one 68 KiB block per (device, stream), zeroed when it is created and left
// zeroed by every launch.
constexpr size_t SYNC_BYTES = 4096 * (NCTR + 1);      // the first KiB: exit counter, abort word, tile counters; then the arrival counters
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
      dxa_set_error("dxa_dit_blocks_fwd: first use on a stream allocates its sync block and cannot happen under stream "
                    "capture: run the request once eagerly on this stream first");
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
size_t act_bytes_for(int M, int H, int I) {
  return ((size_t)M * ((4 + SMAX) * (size_t)H + I) * sizeof(float) + 255) / 256 * 256;
}
int pick_slices(int ncb, int nkb, int grid) {
  for (int sl = SMAX; sl > 1; --sl)
    if (nkb % sl == 0 && ncb * sl <= grid) return sl;
  return 1;
}

}  // namespace

// 1 if a launch on this stream gave up at a device-wide barrier since the last call (its result is garbage); the sync block
// is re-armed.  Synchronises the stream: call it where the host waits for the result anyway.
extern "C" int dxa_dit_blocks_status(dxa_stream_t stream, int* timed_out) {
  DXA_CHECK_ARG(timed_out != nullptr, "dxa_dit_blocks_status: null output");
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

#if defined(DXA_DIT_STAMPS)
// tuning build only: the segment sums of the LAST dit_blocks_fwd launch, [6 phase types][8] (entry 7 = number of phases summed;
// type 5 = the perceptual attention: keys / values requested + q in LDS | scores | row softmax | P V | merge + store | barrier)
extern "C" int dxa_dit_debug_stamps(unsigned long long* out) {
  DXA_CHECK_HIP(hipDeviceSynchronize());
  DXA_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dit_stamps), sizeof(unsigned long long) * 48));
  return DXA_OK;
}
#endif

extern "C" size_t dxa_dit_blocks_workspace(int M, int H, int I) {
  // qkv [M,3H] + o [M,H] + a [M,I] + K-slice partials [8][M,H] floats
  if (M <= 0 || H <= 0 || I <= 0) return 0;
  return act_bytes_for(M, H, I);
}

namespace {
// argument checks, workspace carving and launch geometry shared by the two entry points
int setup_blocks(DitP& p, int* grid_out, float* h, const float* const* weights, int depth, int N, int T1, int H, int heads, int I,
                 float eps, void* workspace, size_t workspace_bytes, hipStream_t st, const char* who) {
  DXA_CHECK_ARG(h && weights && workspace, "%s: null buffer", who);
  DXA_CHECK_ARG(depth >= 1 && N >= 1 && T1 >= 1 && heads >= 1, "%s: bad sizes", who);
  const int M = N * T1;
  DXA_CHECK_ARG(M < MAXM && T1 <= MAXT, "%s: at most %d rows / %d tokens per sample (got %d / %d)", who, MAXM - 1, MAXT, M, T1);
  DXA_CHECK_ARG(H % 64 == 0 && I % 64 == 0 && H == heads * HD && H <= 1024, "%s: needs head width 64, H <= 1024 and H, I %% 64 == 0", who);
  DXA_CHECK_ARG(workspace_bytes >= dxa_dit_blocks_workspace(M, H, I), "%s: workspace too small", who);
  DXA_CHECK_ARG((reinterpret_cast<uintptr_t>(h) % 16) == 0 && (reinterpret_cast<uintptr_t>(workspace) % 16) == 0,
                "%s: buffers must be 16-byte aligned", who);
  p.h = h;
  p.qkv = reinterpret_cast<float*>(workspace);
  p.o = p.qkv + (size_t)M * 3 * H;
  p.a = p.o + (size_t)M * H;
  p.part = p.a + (size_t)M * I;
  unsigned* tail = nullptr;          // first use on a stream allocates: must not happen under stream capture
  if (int rc = get_sync_block(st, &tail)) return rc;
  p.bar = tail;
  p.cnt_proj = tail + 64;
  p.cnt_fc2 = tail + 128;
  p.w = weights;
  p.M = M; p.N = N; p.T1 = T1; p.H = H; p.heads = heads; p.I = I; p.depth = depth;
  p.eps = eps;
  p.scale = 1.f / sqrtf((float)HD);
  static const int dbg = getenv("DXA_DIT_DBG") ? atoi(getenv("DXA_DIT_DBG")) : 0;
  p.dbg = dbg;
  // every workgroup must be resident at once (device-wide barrier): one per 16 columns of the widest product,
  // never more than the 256 CUs can hold
  int grid = I / 16;
  if (3 * H / 16 > grid) grid = 3 * H / 16;
  // ... and never more than fit on the device at once (registers allow one 512-thread workgroup per CU)
  static int resident = 0;
  if (resident == 0) {
    int dev = 0, per_cu = 0, per_cu2 = 0;
    hipDeviceProp_t prop;
    DXA_CHECK_HIP(hipGetDevice(&dev));
    DXA_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    DXA_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dit_blocks_fused_k, 512, 0));
    DXA_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, dit_sample_fused_k, 512, 0));
    if (per_cu2 < per_cu) per_cu = per_cu2;
    resident = per_cu * prop.multiProcessorCount;
    DXA_CHECK_ARG(resident >= 1, "%s: the kernel does not fit on this device", who);
  }
  if (grid > resident) grid = resident;
  static const int grid_cap = getenv("DXA_DIT_GRID") ? atoi(getenv("DXA_DIT_GRID")) : 0;   // tuning aid
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  static const int no_slice = getenv("DXA_DIT_NO_SLICE") ? 1 : 0;
  p.s_proj = no_slice ? 1 : pick_slices(H / 16, H / 64, grid);
  p.s_fc2 = no_slice ? 1 : pick_slices(H / 16, I / 64, grid);
  static const int proj_cap = getenv("DXA_DIT_PROJ_SLICES") ? atoi(getenv("DXA_DIT_PROJ_SLICES")) : 0;      // tuning aid
  if (proj_cap > 0 && p.s_proj > proj_cap && (H / 64) % proj_cap == 0) p.s_proj = proj_cap;
  *grid_out = grid;
  return DXA_OK;
}
}  // namespace

extern "C" int dxa_dit_blocks_fwd(float* h, const float* const* weights, int depth, int N, int T1, int H, int heads, int I,
                                  float eps, void* workspace, size_t workspace_bytes, dxa_stream_t stream) {
  DitP p;
  int grid = 0;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = setup_blocks(p, &grid, h, weights, depth, N, T1, H, heads, I, eps, workspace, workspace_bytes, st, "dxa_dit_blocks_fwd"))
    return rc;
  hipLaunchKernelGGL(dit_blocks_fused_k, dim3(grid), dim3(512), 0, st, p);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

constexpr size_t SAMPLE_EXTRA = (3 * MAXM * MAXA * sizeof(float) + 255) / 256 * 256;      // x ping-pong + eps rows
extern "C" size_t dxa_dit_sample_workspace(int M, int H, int I) {
  if (M <= 0 || H <= 0 || I <= 0) return 0;
  return dxa_dit_blocks_workspace(M, H, I) + ((size_t)M * H * sizeof(float) + 255) / 256 * 256 + SAMPLE_EXTRA;   // + h + small scratch
}

extern "C" int dxa_dit_sample_fwd(float* x, const float* z_emb, const float* t_emb, const float* pos, const float* x_w,
                                  const float* x_b, const float* final_w, const float* final_b, const float* coef, int steps, int A,
                                  int nb, int use_cfg, float cfg_scale, const float* const* weights, int depth, int N, int T1, int H,
                                  int heads, int I, float eps, void* workspace, size_t workspace_bytes, dxa_stream_t stream) {
  DXA_CHECK_ARG(x && z_emb && t_emb && pos && x_w && x_b && final_w && final_b && coef && workspace, "dxa_dit_sample_fwd: null buffer");
  DXA_CHECK_ARG(steps >= 1 && A >= 1 && A <= MAXA && nb >= 1 && N == (use_cfg ? 2 * nb : nb),
                "dxa_dit_sample_fwd: needs 1 <= action_dim <= %d and N == nb (or 2 nb with guidance)", MAXA);
  const int M = N * T1;
  DXA_CHECK_ARG(workspace_bytes >= dxa_dit_sample_workspace(M, H, I), "dxa_dit_sample_fwd: workspace too small");
  DitSampleP sp;
  int grid = 0;
  hipStream_t st = (hipStream_t)stream;
  DXA_CHECK_ARG(nb * (T1 - 1) * A <= 512 && nb * (T1 - 1) * A <= MAXM * MAXA, "dxa_dit_sample_fwd: the sample has too many elements");
  float* h = reinterpret_cast<float*>(workspace);
  float* extra = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ((size_t)M * H * sizeof(float) + 255) / 256 * 256);
  void* ws = reinterpret_cast<char*>(extra) + SAMPLE_EXTRA;
  if (int rc = setup_blocks(sp.blk, &grid, h, weights, depth, N, T1, H, heads, I, eps, ws, workspace_bytes - ((char*)ws - (char*)workspace),
                            st, "dxa_dit_sample_fwd"))
    return rc;
  sp.x = x; sp.ze = z_emb; sp.te = t_emb; sp.pos = pos; sp.xw = x_w; sp.xb = x_b; sp.fw = final_w; sp.fb = final_b; sp.coef = coef;
  sp.steps = steps; sp.A = A; sp.nb = nb; sp.use_cfg = use_cfg; sp.cfg_scale = cfg_scale;
  sp.xpp = extra; sp.eps = extra + 2 * MAXM * MAXA;
  hipLaunchKernelGGL(dit_sample_fused_k, dim3(grid), dim3(512), 0, st, sp);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

// =====================================================================================================================
// bf16-operand sampler (round 4).  The reference serves action inference with the WHOLE model loaded in bfloat16
// (dexbotic/exp/cogact_exp.py:134-138: from_pretrained(torch_dtype=torch.bfloat16); inference_action, cogact_arch.py:149-204, runs
// the DiT on bf16 weights and bf16 activations — the autocast(float32) of :133 wraps the TRAINING head only).  When the product
// serves in bf16 the sampler above still multiplied in exact fp32: 1,024 MACs per 32-cycle v_mfma_f32_16x16x4_f32, 4,600 cycles of
// MFMA per SIMD in every K = 768 phase, 340 MB of fp32 weights streamed per denoising step.  This variant keeps everything that
// decides the accuracy in fp32 — residual stream h, LayerNorm statistics, attention, accumulation, GELU — and hands the MFMAs
// bf16 operands like the reference's matmuls get:
//   * weights: a packed bf16 copy (dxa_dit_bf16_pack): for 16 output columns nb and 32-deep K piece pc one contiguous 1 KiB tile
//     [16 n][32 k], so a wave's 16-byte-per-lane load IS the A operand of v_mfma_f32_16x16x32_bf16 (lane = n + 16 (k / 8)) and is
//     one contiguous KiB; 170 MB for DiT-B: it stays in the Infinity Cache between the steps.  sum_k W[n, k] (the LayerNorm fold
//     needs it) is computed once at pack time, from the rounded weights.
//   * activations that feed a product exist as bf16 tiles [K / 32][Mp rows][32 k] written by their PRODUCER (one 8-byte store per
//     lane next to the fp32 value): hb = bf16(h), ob = attention output, ab = gelu(fc1).  Consumers load 16 bytes per lane and
//     (piece, row block): half the bytes, half the load instructions, no conversion in the hot loop.
//   * LayerNorm statistics travel beside h: its producer leaves (sum, sum of squares) of every (row, 16-column tile) — 13 KB —
//     and a consumer's folding waves add the 48 tile pairs of their rows while the operands are in flight.  LN itself stays in the
//     epilogue: rs (acc - mu wsum) + bias.
// Per K = 768 phase and wave: 12 loads and 9 MFMAs of 16 cycles where the fp32 form has 24 loads and 72 MFMAs of 32 cycles.
namespace {

typedef uint32_t u32x2b_t __attribute__((ext_vector_type(2)));

struct Buf {                       // an activation / workspace array behind a buffer descriptor, agent-scope accesses by BYTE offset
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ Buf(const void* p, size_t bytes)
      : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
  __device__ __forceinline__ u32x4_t ld16(uint32_t off) const { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, SC1); }
  __device__ __forceinline__ float4 ld16f(uint32_t off) const {
    const u32x4_t v = ld16(off);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  }
  __device__ __forceinline__ float ld4f(uint32_t off) const { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, SC1)); }
  __device__ __forceinline__ void st16(uint32_t off, const float4& v) const {
    const u32x4_t u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)off, 0, SC1);
  }
  __device__ __forceinline__ void st8(uint32_t off, uint32_t a, uint32_t b) const {
    const u32x2b_t u = {a, b};
    __builtin_amdgcn_raw_buffer_store_b64(u, r, (int)off, 0, SC1);
  }
  __device__ __forceinline__ void st4f(uint32_t off, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)off, 0, SC1); }
};

struct DitBfP {
  float *h, *qkv, *stats, *part;   // h [M][H] fp32 residual stream; qkv [M][3H] fp32; stats [Mp][H/16][2]; part [S][H/16][Mp][16]
  bf16_t *hb, *ob, *ab;            // [K/32][Mp][32] bf16 operand tiles: bf16(h), attention output, gelu(fc1)
  const void* const* w;            // [depth][wstride]: qkv_wp, qkv_b, proj_wp, proj_b, fc1_wp, fc1_b, fc2_wp, fc2_b, qkv_wsum, fc1_wsum
                                   // (+ with perceptual attention, wstride 16: q_wp, q_b, q_wsum — norm3's affine folded in —, out_wp, out_b)
  const float* kv;                 // perceptual attention: the request's projected keys / values [depth][N][P][2][H] fp32 (P = 0: none)
  unsigned *bar, *cnt_proj, *cnt_fc2;
  int M, Mp, N, T1, H, heads, I, depth;
  int P, wstride;
  int s_proj, s_fc2;
  float eps, scale;
  int dbg;
};

struct BfBufs {
  Buf h, qkv, stats, part, hb, ob, ab;
  __device__ __forceinline__ BfBufs(const DitBfP& p)
      : h(p.h, (size_t)p.M * p.H * 4), qkv(p.qkv, (size_t)p.M * 3 * p.H * 4), stats(p.stats, (size_t)p.Mp * (p.H / 16) * 8),
        part(p.part, (size_t)SMAX * (p.H / 16) * p.Mp * 64), hb(p.hb, (size_t)(p.H / 32) * p.Mp * 64),
        ob(p.ob, (size_t)(p.H / 32) * p.Mp * 64), ab(p.ab, (size_t)(p.I / 32) * p.Mp * 64) {}
};

__device__ __forceinline__ uint32_t tile_off(int Mp, int m, int c) {      // byte offset of element (row m, column c) in a bf16 operand tile array
  return (uint32_t)((((c >> 5) * Mp + m) * 32 + (c & 31)) * 2);
}

// four consecutive columns c .. c+3 (c % 4 == 0, inside ONE 16-column tile ct) of row m of the residual stream: the fp32 value, its
// bf16 operand copy and — from the four lanes that hold the 16 columns of the tile (lanes `lane ^ x1`, `lane ^ x2` are the others) —
// the tile's (sum, sum of squares) for the LayerNorm of the consumers.  Every lane of the wave calls this (shuffles); `valid` gates the stores.
__device__ __forceinline__ void write_h(const DitBfP& p, const BfBufs& b, int m, int c, const float4& t, bool valid, bool first_of_tile,
                                        int x1, int x2) {
  float sx = (t.x + t.y) + (t.z + t.w), sq = (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
  sx += __shfl_xor(sx, x1, 64); sq += __shfl_xor(sq, x1, 64);
  sx += __shfl_xor(sx, x2, 64); sq += __shfl_xor(sq, x2, 64);
  if (valid) {
    b.h.st16((uint32_t)(((size_t)m * p.H + c) * 4), t);
    b.hb.st8(tile_off(p.Mp, m, c), pack_bf16x2(t.x, t.y), pack_bf16x2(t.z, t.w));
    if (first_of_tile) b.stats.st8((uint32_t)((m * (p.H / 16) + (c >> 4)) * 8), __float_as_uint(sx), __float_as_uint(sq));
  }
}

enum { BF_QKV = 0, BF_FC1 = 1, BF_RES = 2 };

// out = epi(pro(A) W^T + bias) with bf16 operands.  BF_QKV: A = hb, LayerNorm, out -> qkv fp32 [M][3H];  BF_FC1: A = hb, LayerNorm,
// gelu, out -> ab tiles;  BF_RES: A = ob / ab, S K-slices folded by the last arrival (as in gemm_phase), out -> h += ..., hb, stats.
template <int KIND, int U>
__device__ __forceinline__ void gemm_bf_u(const DitBfP& p, Smem& s, const BfBufs& b, const Buf& A, int K, const bf16_t* Wp,
                                          const float* bias, const float* wsum, int Nout, int S, unsigned* cnt, unsigned cnt_target) {
  constexpr bool LN = KIND != BF_RES;
  constexpr uint32_t OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lg = lane >> 4, Mp = p.Mp;
  uint32_t aoff[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 16 + l16;
    aoff[mb] = m < p.M ? (uint32_t)(m * 64 + lg * 16) : OOB;
  }
  auto uniform_ptr = [](const void* q) {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t Wr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Wp), 0, (int)((size_t)Nout * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t Br = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(bias), 0, Nout * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t Sr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(LN ? (const void*)wsum : (const void*)bias), 0, Nout * 4, 0x00020000);
  const int ncb = Nout / 16, npc = K / 32, per = npc / S, ntile = K / 16, nq = ntile / 2;
  for (int item = blockIdx.x; item < ncb * S; item += gridDim.x) {
    const int cb = item % ncb, ks = item / ncb, n0 = cb * 16;
    const int pc_lo = ks * per, pc_hi = pc_lo + per;
    const uint32_t woff = (uint32_t)(((size_t)cb * npc * 16 + l16) * 64 + lg * 16);
    const int em = wave * 16 + l16, en = n0 + 4 * lg;
    const bool erow = wave < MB && em < p.M;
    // epilogue operands first: bias, sum_k W, the row's LayerNorm tile sums, the residual
    const u32x4_t b4u = __builtin_amdgcn_raw_buffer_load_b128(Br, en * 4, 0, 0);
    u32x4_t ws4u = {0u, 0u, 0u, 0u};
    float4 st[8];
    if (LN) {
      ws4u = __builtin_amdgcn_raw_buffer_load_b128(Sr, en * 4, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = lg + 4 * i;
        st[i] = b.stats.ld16f(erow && q < nq ? (uint32_t)(em * ntile * 8 + q * 16) : OOB);
      }
    }
    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KIND == BF_RES && S == 1) c4 = b.h.ld16f(erow ? (uint32_t)(((size_t)em * p.H + en) * 4) : OOB);
    f32x4_t acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int base = pc_lo; base < pc_hi; base += 8 * U) {
      u32x4_t wv[U], av[U][MB];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pc = base + wave + 8 * u;
        const bool live = pc < pc_hi;
        wv[u] = __builtin_amdgcn_raw_buffer_load_b128(Wr, (int)(live && p.dbg != 5 ? woff + (uint32_t)pc * 1024u : OOB), 0, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          av[u][mb] = A.ld16(live && aoff[mb] != OOB && p.dbg != 6 ? aoff[mb] + (uint32_t)(pc * Mp) * 64u : OOB);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[u]), __builtin_bit_cast(bf16x8_t, av[u][mb]), acc[mb], 0, 0, 0);
    }
    DIT_STAMP_AFTER(s, 1, acc[0][0] + acc[1][0] + acc[2][0]);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      *reinterpret_cast<float4*>(s.red[wave][mb][lane]) = make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
    __syncthreads();
    DIT_STAMP(s, 2);
    const float4 b4 = make_float4(__uint_as_float(b4u[0]), __uint_as_float(b4u[1]), __uint_as_float(b4u[2]), __uint_as_float(b4u[3]));
    if (wave < MB) {            // wave mb folds the 8 partials of row block mb: lane holds out[m = 16 mb + l16][n0 + 4 lg + {0..3}]
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) {
        const float4 v = *reinterpret_cast<const float4*>(s.red[w8][wave][lane]);
        r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
      }
      if (S == 1) {
        if (LN) {
          float tx = 0.f, txx = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) { tx += st[i].x + st[i].z; txx += st[i].y + st[i].w; }
          tx += __shfl_xor(tx, 16, 64); txx += __shfl_xor(txx, 16, 64);
          tx += __shfl_xor(tx, 32, 64); txx += __shfl_xor(txx, 32, 64);
          const float mu = tx / (float)K;
          const float rs = rsqrtf(fmaxf(txx / (float)K - mu * mu, 0.f) + p.eps);
          r.x = rs * (r.x - mu * __uint_as_float(ws4u[0])); r.y = rs * (r.y - mu * __uint_as_float(ws4u[1]));
          r.z = rs * (r.z - mu * __uint_as_float(ws4u[2])); r.w = rs * (r.w - mu * __uint_as_float(ws4u[3]));
        }
        r.x += b4.x; r.y += b4.y; r.z += b4.z; r.w += b4.w;
        if (KIND == BF_QKV) {
          if (em < p.M) b.qkv.st16((uint32_t)(((size_t)em * Nout + en) * 4), r);
        } else if (KIND == BF_FC1) {
          r.x = act_fwd(DXA_ACT_GELU_TANH, r.x); r.y = act_fwd(DXA_ACT_GELU_TANH, r.y);
          r.z = act_fwd(DXA_ACT_GELU_TANH, r.z); r.w = act_fwd(DXA_ACT_GELU_TANH, r.w);
          if (em < p.M) b.ab.st8(tile_off(Mp, em, en), pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w));
        } else {
          r.x += c4.x; r.y += c4.y; r.z += c4.z; r.w += c4.w;
          write_h(p, b, em, en, r, em < p.M, lg == 0, 16, 32);
        }
      } else if (em < p.M) {
        b.part.st16((uint32_t)((((size_t)ks * ncb + cb) * Mp + em) * 64 + lg * 16), r);
      }
    }
    if (S > 1) {
      // (tried in round 4 and dropped: the partial tiles as 8-byte (value, tag) pairs polled by the workgroup of slice 0 instead of
      // counter + gather — the fold segment grew from 4,100 to 6,600 cycles: a poll round is a full round trip and the first one
      // always misses, profiles/r04_sampler_pairs.txt)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned arrived = __hip_atomic_fetch_add(cnt + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        s.last = arrived == cnt_target;
      }
      __syncthreads();
      if (s.last && wave < MB) {
        float4 t = b.h.ld16f(em < p.M ? (uint32_t)(((size_t)em * p.H + en) * 4) : OOB);
        float4 pq[SMAX];
#pragma unroll
        for (int q = 0; q < SMAX; ++q)
          pq[q] = b.part.ld16f(q < S && em < p.M ? (uint32_t)((((size_t)q * ncb + cb) * Mp + em) * 64 + lg * 16) : OOB);
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < SMAX; ++q) { a4.x += pq[q].x; a4.y += pq[q].y; a4.z += pq[q].z; a4.w += pq[q].w; }
        t.x += b4.x + a4.x; t.y += b4.y + a4.y; t.z += b4.z + a4.z; t.w += b4.w + a4.w;
        write_h(p, b, em, en, t, em < p.M, lg == 0, 16, 32);
      }
    }
    __syncthreads();
  }
}

template <int KIND>
__device__ __forceinline__ void gemm_bf(const DitBfP& p, Smem& s, const BfBufs& b, const Buf& A, int K, const void* Wp, const void* bias,
                                        const void* wsum, int Nout, int S, unsigned* cnt, unsigned cnt_target) {
  const int per = K / 32 / S;
  const bf16_t* W = reinterpret_cast<const bf16_t*>(Wp);
  const float *bi = reinterpret_cast<const float*>(bias), *ws = reinterpret_cast<const float*>(wsum);
  if (per <= 8) gemm_bf_u<KIND, 1>(p, s, b, A, K, W, bi, ws, Nout, S, cnt, cnt_target);
  else if (per <= 16) gemm_bf_u<KIND, 2>(p, s, b, A, K, W, bi, ws, Nout, S, cnt, cnt_target);
  else gemm_bf_u<KIND, 3>(p, s, b, A, K, W, bi, ws, Nout, S, cnt, cnt_target);
}

// attention_phase with the output written as bf16 operand tiles (ob): the A operand of the output projection
__device__ __forceinline__ void attention_bf(const DitBfP& p, Smem& s, const BfBufs& b) {
  const int tid = threadIdx.x, T1 = p.T1, ld = 3 * p.H;
  constexpr uint32_t OOB = 0x80000000u;
  for (int pr = blockIdx.x; pr < p.N * p.heads; pr += gridDim.x) {
    const int n = pr / p.heads, hd = pr - n * p.heads;
    const size_t base = (size_t)n * T1 * ld + hd * HD;
    float4 ld_[3];
    const int li = tid >> 4, ldd = (tid & 15) * 4;
#pragma unroll
    for (int r3 = 0; r3 < 3; ++r3)
      ld_[r3] = b.qkv.ld16f(li < T1 ? (uint32_t)((base + (size_t)li * ld + (size_t)r3 * p.H + ldd) * 4) : OOB);
    *reinterpret_cast<float4*>(&s.q[li][ldd]) = ld_[0];
    *reinterpret_cast<float4*>(&s.k[li][ldd]) = ld_[1];
    *reinterpret_cast<float4*>(&s.v[li][ldd]) = ld_[2];
    __syncthreads();
    const int j = tid & 31;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int i = (tid >> 5) + 16 * pass;
      if (i >= T1) continue;                                      // 17 tokens: the second pass is one row of one half-wave
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 q4 = *reinterpret_cast<const float4*>(&s.q[i][d]), k4 = *reinterpret_cast<const float4*>(&s.k[j][d]);
        sc += (q4.x * k4.x + q4.y * k4.y) + (q4.z * k4.z + q4.w * k4.w);
      }
      sc = j < T1 ? sc * p.scale : -INFINITY;
      float mx = sc;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float e = j < T1 ? expf(sc - mx) : 0.f;
      float sum = e;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
      if (i < T1) s.p[i][j] = e / sum;
    }
    __syncthreads();
    {
      const int i = tid >> 4, d = (tid & 15) * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const int T1r = (T1 + 3) & ~3;                 // keys T1 .. T1r-1: p is an exact zero, v a zero row
#pragma unroll 4
      for (int jj = 0; jj < T1r; ++jj) {
        const float pj = s.p[i][jj];
        const float4 v4 = *reinterpret_cast<const float4*>(&s.v[jj][d]);
        acc.x += pj * v4.x; acc.y += pj * v4.y; acc.z += pj * v4.z; acc.w += pj * v4.w;
      }
      if (i < T1) b.ob.st8(tile_off(p.Mp, n * T1 + i, hd * HD + d), pack_bf16x2(acc.x, acc.y), pack_bf16x2(acc.z, acc.w));
    }
    __syncthreads();
  }
}

// MemVLA's perceptual cross attention inside the sampler (nn.MultiheadAttention over the perceptual tokens, memvla/action_model/dit.py:
// 158-185): q = the block's projection of norm3(h) [M][H] fp32 (at the start of the qkv array, row stride H), keys / values = the
// request's cached projections kv [N][P][2][H] fp32 — written before the launch, read-only in it: plain cached loads.
// One workgroup per (sample, head, group of <= 8 query rows), the rows of a group at once, so a key / value row is read once per group
// (the first form — one wave per (row, head), every wave walking all P keys — re-read them T1 times through the vector L1: 77 us a
// phase, 18.5 of the sampler's 30 ms; all rows in one workgroup: 26 us, bound by the LDS reads of the query rows — every wave
// broadcasts every row; profiles/r06_memvla_sampler.txt).  Wave w owns keys [w P/8, (w+1) P/8):
//   scores   lane = (key, half of the head width): its half K row in registers (8 x 16-byte loads), the T1 query rows broadcast from
//            LDS, the two halves joined by one cross-lane add;
//   softmax  scaled scores -> LDS [key][row]; lane r then walks row r of the wave's tile for its max and sum (a cross-lane reduction
//            per row is ten dependent permutes: 17 rows of them were most of the phase), probabilities back in place;
//   P V      lane = column: a value row is one 256-byte line, T1 accumulators a lane;
//   merge    the eight waves' (max, sum, partial output) per row through LDS, flash-attention style; output -> the bf16 operand
//            tiles of the output projection (ob).
struct PerOperands { float4 kr[8]; float vr[32]; };      // a lane's half key row and its column of the wave's value rows
constexpr int PER_TP = 8;           // query rows a workgroup takes, padded to a multiple of 4
constexpr int PER_MAXT = 24;        // tokens per sample with perceptual attention (three row groups of 8)
// work item = (sample, head, group of <= 8 query rows).  The row groups of one (sample, head) read the same keys / values: they are
// dealt to workgroups whose ids are congruent mod 8 — one XCD, one L2 (workgroups go round-robin over the 8 XCDs) — and to workgroups
// that have no tile of the query projection (the first H / 16 have one): those request their keys / values while the projection runs.
__device__ __forceinline__ int per_first_item(const DitBfP& p) {
  const int shift = (p.H / 16) % (int)gridDim.x;
  return ((int)blockIdx.x + (int)gridDim.x - shift) % (int)gridDim.x;
}
__device__ __forceinline__ int per_items(const DitBfP& p) { return ((p.N * p.heads + 7) & ~7) * ((p.T1 + PER_TP - 1) / PER_TP); }
__device__ __forceinline__ bool per_decode(const DitBfP& p, int item, int& n, int& hd, int& r0, int& nr) {      // false: a hole of the padding
  const int npair = p.N * p.heads, nrg = (p.T1 + PER_TP - 1) / PER_TP, npair8 = (npair + 7) & ~7, rper = (p.T1 + nrg - 1) / nrg;   // 17 rows: 6, 6, 5
  const int pr = item % npair8, rg = item / npair8;
  if (item >= npair8 * nrg || pr >= npair) return false;
  n = pr / p.heads; hd = pr - n * p.heads; r0 = rg * rper; nr = min(rper, p.T1 - r0);
  return true;
}
__device__ __forceinline__ void per_load(const DitBfP& p, const float* __restrict__ kv, int n, int hd, PerOperands& q) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, key = lane & 31, half = lane >> 5;
  const int KW = p.P >> 3, ld = 2 * p.H;
  const float* kbase = kv + (size_t)n * p.P * ld + hd * HD;
  const float4* krow = reinterpret_cast<const float4*>(kbase + (size_t)(wave * KW + (key < KW ? key : 0)) * ld + half * 32);
#pragma unroll
  for (int c = 0; c < 8; ++c) q.kr[c] = krow[c];
  const float* vbase = kbase + p.H + (size_t)(wave * KW) * ld + lane;
#pragma unroll
  for (int jj = 0; jj < 32; ++jj) q.vr[jj] = jj < KW ? vbase[(size_t)jj * ld] : 0.f;
}

__device__ __forceinline__ void per_attention_bf(const DitBfP& p, Smem& s, const BfBufs& b, const float* __restrict__ kv,
                                                 PerOperands& ops, bool preloaded) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KW = p.P >> 3, T1 = p.T1;                           // KW = keys per wave: 8 .. 32
  // LDS: the phase owns everything in front of Smem::last — [ q: TP x 64 | (max, sum): 8 x TP x 2 | A: scores / probabilities 8 x 32 x TP,
  // then (after the P V loop) the waves' partial outputs 8 x TP x 64 ]
  float* const sq = reinterpret_cast<float*>(&s.red[0][0][0][0]);
  float* const sml = sq + PER_TP * HD;
  float* const sA = sml + 8 * PER_TP * 2;
  static_assert((PER_TP * HD + 8 * PER_TP * 2 + 8 * PER_TP * HD) * sizeof(float) <= offsetof(Smem, last), "per_attention_bf: LDS plan");
  static_assert(32 * PER_TP <= PER_TP * HD, "per_attention_bf: the probabilities fit under the partial outputs");
  float* const pw = sA + wave * 32 * PER_TP;
  const int key = lane & 31, half = lane >> 5;
  const int first = per_first_item(p);
  for (int item = first; item < per_items(p); item += gridDim.x) {
    int n, hd, r0, nr;
    if (!per_decode(p, item, n, hd, r0, nr)) continue;     // (uniform per workgroup)
    // this wave's key rows (half a row a lane) and value rows (one column a lane): requested before the barrier in front of this phase
    // (per_load in walk_blocks_bf) or, for a second item of the workgroup, first thing here — they do not depend on q
    if (!(preloaded && item == first)) per_load(p, kv, n, hd, ops);
    const bool kval = key < KW;
    const float4* kr = ops.kr;
    const float* vr = ops.vr;
    if (tid < nr * 16) {
      const int i = tid >> 4, d4 = tid & 15;
      *reinterpret_cast<float4*>(sq + i * HD + d4 * 4) = b.qkv.ld16f((uint32_t)(((size_t)(n * T1 + r0 + i) * p.H + hd * HD + d4 * 4) * 4));
    }
    __syncthreads();
    DIT_STAMP(s, 1);
    float pv[PER_TP];
#pragma unroll
    for (int i = 0; i < PER_TP; ++i) {
      pv[i] = 0.f;
      if (i < nr) {
        const float4* qr = reinterpret_cast<const float4*>(sq + i * HD + half * 32);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
          const float4 q0 = qr[c], q1 = qr[c + 1];
          a0 += (q0.x * kr[c].x + q0.y * kr[c].y) + (q0.z * kr[c].z + q0.w * kr[c].w);
          a1 += (q1.x * kr[c + 1].x + q1.y * kr[c + 1].y) + (q1.z * kr[c + 1].z + q1.w * kr[c + 1].w);
        }
        float sc = a0 + a1;
        sc += __shfl_xor(sc, 32, 64);
        pv[i] = kval ? sc * p.scale : -INFINITY;
      }
    }
    if (half == 0) {                                       // scaled scores -> LDS [key][row] (rows nr .. TP-1: zeros, and they stay zeros)
#pragma unroll
      for (int i = 0; i < PER_TP; i += 4) *reinterpret_cast<float4*>(pw + key * PER_TP + i) = make_float4(pv[i], pv[i + 1], pv[i + 2], pv[i + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    DIT_STAMP(s, 2);
    // the wave's softmax statistics: lane r takes row r of its 32 x TP score tile — 32 independent LDS reads, then registers (a
    // cross-lane reduction per row is ten dependent permutes).  Keys past KW hold -inf: probability 0.
    if (lane < nr) {
      float sv[32];
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) sv[jj] = pw[jj * PER_TP + lane];
      float mx = sv[0];
#pragma unroll
      for (int jj = 1; jj < 32; ++jj) mx = fmaxf(mx, sv[jj]);
      float sum = 0.f;
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) {
        sv[jj] = __expf(sv[jj] - mx);
        sum += sv[jj];
        pw[jj * PER_TP + lane] = sv[jj];
      }
      sml[(wave * PER_TP + lane) * 2] = mx;
      sml[(wave * PER_TP + lane) * 2 + 1] = sum;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    DIT_STAMP(s, 3);
    // ---- P V over this wave's keys: lane = column
    float o[PER_TP];
#pragma unroll
    for (int i = 0; i < PER_TP; ++i) o[i] = 0.f;
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) {
      if (jj < KW) {
        const float v = vr[jj];
#pragma unroll
        for (int i = 0; i < PER_TP; i += 4) {
          const float4 p4 = *reinterpret_cast<const float4*>(pw + jj * PER_TP + i);
          o[i] += p4.x * v; o[i + 1] += p4.y * v; o[i + 2] += p4.z * v; o[i + 3] += p4.w * v;
        }
      }
    }
    DIT_STAMP_AFTER(s, 4, o[0] + o[1]);
    __syncthreads();                                       // every wave is done with its probabilities: A becomes the partial outputs
    float* const so = sA + wave * PER_TP * HD;
#pragma unroll
    for (int i = 0; i < PER_TP; ++i)
      if (i < nr) so[i * HD + lane] = o[i];
    __syncthreads();
    if (tid < nr * 16) {                                   // merge: thread = (row, four columns)
      const int i = tid >> 4, d = (tid & 15) * 4;
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < 8; ++w) M = fmaxf(M, sml[(w * PER_TP + i) * 2]);
      float Lsum = 0.f;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const float f = __expf(sml[(w * PER_TP + i) * 2] - M);
        Lsum += f * sml[(w * PER_TP + i) * 2 + 1];
        const float4 t = *reinterpret_cast<const float4*>(sA + (w * PER_TP + i) * HD + d);
        acc.x += f * t.x; acc.y += f * t.y; acc.z += f * t.z; acc.w += f * t.w;
      }
      const float inv = 1.f / Lsum;
      b.ob.st8(tile_off(p.Mp, n * T1 + r0 + i, hd * HD + d), pack_bf16x2(acc.x * inv, acc.y * inv), pack_bf16x2(acc.z * inv, acc.w * inv));
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void walk_blocks_bf(const DitBfP& p, Smem& s, const BfBufs& b, unsigned& epoch, unsigned nblk, unsigned base) {
  const bool work = p.dbg != 1, sync = p.dbg != 2;
  for (int blk = 0; blk < p.depth; ++blk) {
    const void* const* w = p.w + blk * p.wstride;
    const bool per = p.P > 0;
    // the K-sliced output projections count arrivals per column block: with perceptual attention TWO products per block use cnt_proj
    const unsigned tgt_proj = per ? (2u * (base + (unsigned)blk) + 1u) * p.s_proj : (base + (unsigned)blk + 1u) * p.s_proj;
    DIT_STAMP(s, 0);
    if (work) gemm_bf<BF_QKV>(p, s, b, b.hb, p.H, w[0], w[1], w[8], 3 * p.H, 1, nullptr, 0);
    DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, true, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 0, 5);
    DIT_STAMP(s, 0);
    if (work && p.dbg != 3) attention_bf(p, s, b);
    DIT_STAMP(s, 1); DIT_STAMP(s, 2); DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, true, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 1, 5);
    DIT_STAMP(s, 0);
    if (work) gemm_bf<BF_RES>(p, s, b, b.ob, p.H, w[2], w[3], nullptr, p.H, p.s_proj, p.cnt_proj, tgt_proj);
    DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, true, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 2, 5);
    if (per) {
      // x + MHA(norm3 x, per, per): q = (norm3 folded into the packed q rows of in_proj) | attention over the cached keys / values | out_proj
      if (work) gemm_bf<BF_QKV>(p, s, b, b.hb, p.H, w[10], w[11], w[12], p.H, 1, nullptr, 0);
      // the keys / values of this workgroup's first attention item: in flight (or, for the workgroups without a projection tile,
      // already here) when the barrier opens — a cold 128 KB of them costs 5 us after it (profiles/r06_memvla_sampler.txt)
      const float* kvb = p.kv + (size_t)blk * p.N * p.P * 2 * p.H;
      PerOperands ops;
      bool pre = false;
      if (work && p.dbg != 3 && p.dbg != 7) {
        int n_, hd_, r0_, nr_;
        if (per_decode(p, per_first_item(p), n_, hd_, r0_, nr_)) { per_load(p, kvb, n_, hd_, ops); pre = true; }
      }
      if (sync) grid_sync(p.bar, nblk, epoch, true, &s);
      DIT_STAMP(s, 0);
      if (work && p.dbg != 3) per_attention_bf(p, s, b, kvb, ops, pre);
      DIT_STAMP(s, 5);
      if (sync) grid_sync(p.bar, nblk, epoch, true, &s);
      DIT_STAMP(s, 6); DIT_STAMP_FOLD(s, 5, 6);
      if (work) gemm_bf<BF_RES>(p, s, b, b.ob, p.H, w[13], w[14], nullptr, p.H, p.s_proj, p.cnt_proj, tgt_proj + (unsigned)p.s_proj);
      if (sync) grid_sync(p.bar, nblk, epoch, true, &s);
    }
    DIT_STAMP(s, 0);
    if (work) gemm_bf<BF_FC1>(p, s, b, b.hb, p.H, w[4], w[5], w[9], p.I, 1, nullptr, 0);
    DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, true, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 3, 5);
    DIT_STAMP(s, 0);
    if (work) gemm_bf<BF_RES>(p, s, b, b.ab, p.I, w[6], w[7], nullptr, p.H, p.s_fc2, p.cnt_fc2, (base + (unsigned)blk + 1u) * p.s_fc2);
    DIT_STAMP(s, 3);
    if (sync) grid_sync(p.bar, nblk, epoch, true, &s);
    DIT_STAMP(s, 5); DIT_STAMP_FOLD(s, 4, 5);
  }
}

struct DitSampleBfP {
  DitBfP blk;
  float* x; const float* ze; const float* te; const float* pos; const float* xw; const float* xb; const float* fw; const float* fb;
  const float* coef; float* xpp; float* eps;
  int steps, A, nb, use_cfg;
  float cfg_scale;
};

__global__ __launch_bounds__(512) void dit_sample_bf16_k(const DitSampleBfP sp) {
  __shared__ Smem s;
  __shared__ float xs[MAXM * MAXA];
  const DitBfP& p = sp.blk;
  unsigned epoch = 0;
  const unsigned nblk = gridDim.x;
  const int T = p.T1 - 1, A = sp.A, H = p.H, nx = sp.nb * T * A;
  const BfBufs b(p);
  const Act xin(sp.x, (size_t)nx), xpp(sp.xpp, (size_t)2 * nx), epsg(sp.eps, (size_t)p.M * MAXA), hact(p.h, (size_t)p.M * p.H);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if defined(DXA_DIT_STAMPS)
  if (threadIdx.x < 48) s.stamp[threadIdx.x / 8][threadIdx.x % 8] = 0ull;
  __syncthreads();
#endif
  auto load_x = [&](int step) {          // as in dit_sample_fused_k
    if (tid < nx) {
      float xv;
      if (step == 0) {
        xv = xin.ld1(tid);
      } else {
        const int k = tid % A, t = (tid / A) % T, n = tid / (A * T);
        float eps = epsg.ld1((size_t)(n * p.T1 + 1 + t) * MAXA + k);
        if (sp.use_cfg) {
          const float eu = epsg.ld1((size_t)((n + sp.nb) * p.T1 + 1 + t) * MAXA + k);
          eps = eu + sp.cfg_scale * (eps - eu);
        }
        const float c_recip = sp.coef[(step - 1) * 4], c_recipm1 = sp.coef[(step - 1) * 4 + 1], ab_prev = sp.coef[(step - 1) * 4 + 2];
        const float xo = xpp.ld1((size_t)((step - 1) & 1) * nx + tid);
        const float x0 = c_recip * xo - c_recipm1 * eps;
        const float eps2 = (c_recip * xo - x0) / c_recipm1;
        xv = x0 * sqrtf(ab_prev) + sqrtf(1.f - ab_prev - 0.f) * eps2;
      }
      xs[tid] = xv;
      if (blockIdx.x == 0) {
        if (step < sp.steps) xpp.st1((size_t)(step & 1) * nx + tid, xv);
        else xin.st1(tid, xv);
      }
    }
    __syncthreads();
  };
  const int ntile = H / 16;
  for (int step = 0; step < sp.steps; ++step) {
    load_x(step);
    // ---- assemble h = [t_emb + z_emb ; x W_x^T + b_x] + pos: four lanes per (row, 16-column tile), each four columns
    for (int g = blockIdx.x * 512 + tid; (g >> 2) < ((p.M * ntile + 127) / 128) * 128; g += gridDim.x * 512) {
      const int item = g >> 2, q = g & 3;
      const bool valid = item < p.M * ntile;
      const int m = valid ? item / ntile : 0, ct = valid ? item - m * ntile : 0, c = ct * 16 + 4 * q;
      const int n = m / p.T1, t = m - n * p.T1;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (t == 0) {
          v[e] = sp.te[(size_t)step * H + c + e] + sp.ze[(size_t)n * H + c + e];
        } else {
          const float* xr = xs + ((n % sp.nb) * T + (t - 1)) * A;
          float a = sp.xb[c + e];
          for (int k = 0; k < A; ++k) a += xr[k] * sp.xw[(size_t)(c + e) * A + k];
          v[e] = a;
        }
        v[e] += sp.pos[(size_t)t * H + c + e];
      }
      write_h(p, b, m, c, make_float4(v[0], v[1], v[2], v[3]), valid, q == 0, 1, 2);
    }
    grid_sync(p.bar, nblk, epoch, true);
    walk_blocks_bf(p, s, b, epoch, nblk, (unsigned)step * (unsigned)p.depth);
    // ---- final layer on the action tokens (fp32, from the fp32 residual stream): one wave per row
    for (int m = blockIdx.x * 8 + wave; m < p.M; m += gridDim.x * 8) {
      if (m % p.T1 == 0) continue;
      float v[16];
      float sx = 0.f, sxx = 0.f;
      const int per = H / 64;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v[j] = 0.f;
        if (j < per) { v[j] = hact.ld1((size_t)m * H + j * 64 + lane); sx += v[j]; sxx += v[j] * v[j]; }
      }
      sx = wave_sum(sx); sxx = wave_sum(sxx);
      const float mu = sx / (float)H;
      const float rs = rsqrtf(fmaxf(sxx / (float)H - mu * mu, 0.f) + p.eps);
      for (int k = 0; k < A; ++k) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (j < per) d += (v[j] - mu) * rs * sp.fw[(size_t)k * H + j * 64 + lane];
        d = wave_sum(d);
        if (lane == 0) epsg.st1((size_t)m * MAXA + k, d + sp.fb[k]);
      }
    }
    grid_sync(p.bar, nblk, epoch, true);
  }
  if (blockIdx.x == 0) load_x(sp.steps);
#if defined(DXA_DIT_STAMPS)
  __syncthreads();
  if (blockIdx.x == DXA_DIT_STAMPS && threadIdx.x < 48) g_dit_stamps[threadIdx.x / 8][threadIdx.x % 8] = s.stamp[threadIdx.x / 8][threadIdx.x % 8];
#endif
  if (threadIdx.x == 0) {                 // leave_clean
    unsigned* exit_cnt = p.bar + 48;
    const unsigned out = __hip_atomic_fetch_add(exit_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (out == nblk) {
      for (int i = 0; i < 64; ++i) {
        __hip_atomic_store(p.cnt_proj + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.cnt_fc2 + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (unsigned g = 0; g < NCTR; ++g)
        __hip_atomic_store(p.bar + 1024u * (g + 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(exit_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- weight packing: one workgroup per (block, matrix, 16 output columns); thread (r = tid / 16, t = tid % 16) walks row nb * 16 + r
// With perceptual attention (per = 1; `w` is then [depth][14]: + in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, norm3.weight,
// norm3.bias; the table [depth][16]) two more matrices per block: the QUERY rows of nn.MultiheadAttention's packed in_proj with norm3's
// affine folded in — W'[n][k] = W[n][k] gamma[k], b'[n] = b[n] + sum_k W[n][k] beta[k], so the phase's plain-LayerNorm identity
// rs (x W'^T - mu sum_k W') + b' is norm3(x) W^T + b — and out_proj.
struct PackP { const float* const* w; char* arena; const void** table; int depth, H, I, per, wst, tst; size_t per_block, off[10]; };
__global__ __launch_bounds__(256) void dit_bf16_pack_k(const PackP q) {
  const int nbq = 3 * q.H / 16, nbp = q.H / 16, nb1 = q.I / 16, nb2 = q.H / 16, nbx = q.per ? q.H / 16 : 0;
  const int per_blk = nbq + nbp + nb1 + nb2 + 2 * nbx;
  const int blk = blockIdx.x / per_blk;
  int r0 = blockIdx.x - blk * per_blk, mat, K;
  if (r0 < nbq) { mat = 0; K = q.H; } else if ((r0 -= nbq) < nbp) { mat = 1; K = q.H; } else if ((r0 -= nbp) < nb1) { mat = 2; K = q.H; }
  else if ((r0 -= nb1) < nb2) { mat = 3; K = q.I; } else if ((r0 -= nb2) < nbx) { mat = 4; K = q.H; } else { r0 -= nbx; mat = 5; K = q.H; }
  const int nb = r0, tid = threadIdx.x, r = tid >> 4, t = tid & 15, npc = K / 32;
  const int wsrc[6] = {0, 2, 4, 6, 8, 10};
  const float* W = q.w[blk * q.wst + wsrc[mat]];
  char* base = q.arena + (size_t)blk * q.per_block;
  const size_t woff[6] = {q.off[0], q.off[2], q.off[3], q.off[5], q.off[6], q.off[9]};
  bf16_t* out = reinterpret_cast<bf16_t*>(base + woff[mat]);
  float sum = 0.f, bfold = 0.f;
  const float* row = W + (size_t)(nb * 16 + r) * K;
  const float* gam = mat == 4 ? q.w[blk * q.wst + 12] : nullptr;
  const float* bet = mat == 4 ? q.w[blk * q.wst + 13] : nullptr;
  for (int k4 = t; k4 < K / 4; k4 += 16) {
    float4 v = *reinterpret_cast<const float4*>(row + 4 * k4);
    if (mat == 4) {
      const float4 g = *reinterpret_cast<const float4*>(gam + 4 * k4), be = *reinterpret_cast<const float4*>(bet + 4 * k4);
      bfold += (v.x * be.x + v.y * be.y) + (v.z * be.z + v.w * be.w);
      v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
    }
    const uint32_t lo = pack_bf16x2(v.x, v.y), hi = pack_bf16x2(v.z, v.w);
    sum += (__uint_as_float(lo << 16) + __uint_as_float(lo & 0xffff0000u)) + (__uint_as_float(hi << 16) + __uint_as_float(hi & 0xffff0000u));
    const int k = 4 * k4, pc = k >> 5, sl = k & 31;
    *reinterpret_cast<uint2*>(out + (((size_t)nb * npc + pc) * 16 + r) * 32 + sl) = make_uint2(lo, hi);
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) { sum += __shfl_xor(sum, o, 64); bfold += __shfl_xor(bfold, o, 64); }
  if (t == 0 && (mat == 0 || mat == 2)) reinterpret_cast<float*>(base + (mat == 0 ? q.off[1] : q.off[4]))[nb * 16 + r] = sum;
  if (t == 0 && mat == 4) {
    reinterpret_cast<float*>(base + q.off[7])[nb * 16 + r] = sum;
    reinterpret_cast<float*>(base + q.off[8])[nb * 16 + r] = q.w[blk * q.wst + 9][nb * 16 + r] + bfold;
  }
  if (blockIdx.x % per_blk == 0 && tid < q.tst) {
    const void* e;
    switch (tid) {
      case 0: e = base + q.off[0]; break;  case 1: e = q.w[blk * q.wst + 1]; break;
      case 2: e = base + q.off[2]; break;  case 3: e = q.w[blk * q.wst + 3]; break;
      case 4: e = base + q.off[3]; break;  case 5: e = q.w[blk * q.wst + 5]; break;
      case 6: e = base + q.off[5]; break;  case 7: e = q.w[blk * q.wst + 7]; break;
      case 8: e = base + q.off[1]; break;  case 9: e = base + q.off[4]; break;
      case 10: e = base + q.off[6]; break; case 11: e = base + q.off[8]; break;
      case 12: e = base + q.off[7]; break; case 13: e = base + q.off[9]; break;
      case 14: e = q.w[blk * q.wst + 11]; break;
      default: e = nullptr; break;
    }
    q.table[blk * q.tst + tid] = e;
  }
}

void bf16_layout(int H, int I, int per, size_t off[10], size_t* per_block) {
  size_t o = 0;
  auto put = [&](int i, size_t bytes) { off[i] = o; o += (bytes + 255) / 256 * 256; };
  put(0, (size_t)3 * H * H * 2); put(1, (size_t)3 * H * 4); put(2, (size_t)H * H * 2); put(3, (size_t)I * H * 2); put(4, (size_t)I * 4);
  put(5, (size_t)H * I * 2);
  for (int i = 6; i < 10; ++i) off[i] = 0;
  if (per) { put(6, (size_t)H * H * 2); put(7, (size_t)H * 4); put(8, (size_t)H * 4); put(9, (size_t)H * H * 2); }
  *per_block = o;
}
size_t bf_ws_bytes(int M, int H, int I, size_t off[8]) {
  const size_t Mp = (size_t)((M + 1) & ~1);
  size_t o = 0;
  auto put = [&](int i, size_t bytes) { off[i] = o; o += (bytes + 255) / 256 * 256; };
  put(0, (size_t)M * H * 4);                       // h
  put(1, (size_t)M * 3 * H * 4);                   // qkv
  put(2, Mp * (H / 16) * 8);                       // stats
  put(3, (size_t)SMAX * (H / 16) * Mp * 64);       // part
  put(4, (size_t)(H / 32) * Mp * 64);              // hb
  put(5, (size_t)(H / 32) * Mp * 64);              // ob
  put(6, (size_t)(I / 32) * Mp * 64);              // ab
  put(7, 3 * MAXM * MAXA * sizeof(float));         // x ping-pong + eps rows
  return o;
}

}  // namespace

namespace {
size_t pack_bytes(int depth, int H, int I, int per) {
  if (depth <= 0 || H <= 0 || I <= 0) return 0;
  size_t off[10], pb = 0;
  bf16_layout(H, I, per, off, &pb);
  return pb * depth;
}
int pack_launch(const char* who, const float* const* weights, int depth, int H, int I, int per, void* packed, size_t packed_bytes,
                const void** table, dxa_stream_t stream) {
  DXA_CHECK_ARG(weights && packed && table, "%s: null buffer", who);
  DXA_CHECK_ARG(depth >= 1 && H % 64 == 0 && I % 64 == 0, "%s: needs H, I %% 64 == 0", who);
  DXA_CHECK_ARG(packed_bytes >= pack_bytes(depth, H, I, per), "%s: arena too small", who);
  DXA_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) % 256) == 0, "%s: the arena must be 256-byte aligned", who);
  PackP q;
  q.w = weights; q.arena = reinterpret_cast<char*>(packed); q.table = table; q.depth = depth; q.H = H; q.I = I;
  q.per = per; q.wst = per ? 14 : 8; q.tst = per ? 16 : 10;
  bf16_layout(H, I, per, q.off, &q.per_block);
  const int per_blk = 3 * H / 16 + H / 16 + I / 16 + H / 16 + (per ? 2 * (H / 16) : 0);
  hipLaunchKernelGGL(dit_bf16_pack_k, dim3(depth * per_blk), dim3(256), 0, (hipStream_t)stream, q);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
}  // namespace

extern "C" size_t dxa_dit_bf16_pack_bytes(int depth, int H, int I) { return pack_bytes(depth, H, I, 0); }
extern "C" size_t dxa_dit_bf16_pack_per_bytes(int depth, int H, int I) { return pack_bytes(depth, H, I, 1); }

extern "C" int dxa_dit_bf16_pack(const float* const* weights, int depth, int H, int I, void* packed, size_t packed_bytes,
                                 const void** table, dxa_stream_t stream) {
  return pack_launch("dxa_dit_bf16_pack", weights, depth, H, I, 0, packed, packed_bytes, table, stream);
}

extern "C" int dxa_dit_bf16_pack_per(const float* const* weights, int depth, int H, int I, void* packed, size_t packed_bytes,
                                     const void** table, dxa_stream_t stream) {
  return pack_launch("dxa_dit_bf16_pack_per", weights, depth, H, I, 1, packed, packed_bytes, table, stream);
}

extern "C" size_t dxa_dit_sample_bf16_workspace(int M, int H, int I) {
  if (M <= 0 || H <= 0 || I <= 0) return 0;
  size_t off[8];
  return bf_ws_bytes(M, H, I, off);
}

namespace {
int sample_bf16_launch(const char* who, float* x, const float* z_emb, const float* t_emb, const float* pos, const float* x_w,
                       const float* x_b, const float* final_w, const float* final_b, const float* coef, int steps, int A,
                       int nb, int use_cfg, float cfg_scale, const void* const* packed_table, const float* per_kv, int P, int depth, int N,
                       int T1, int H, int heads, int I, float eps, void* workspace, size_t workspace_bytes, dxa_stream_t stream) {
  DXA_CHECK_ARG(P == 0 || (per_kv && P % 64 == 0 && P >= 64 && P <= 256 && T1 <= PER_MAXT),
                "%s: perceptual attention needs 64 <= P <= 256, P %% 64 == 0, at most %d tokens per sample and the keys / values", who, PER_MAXT);
  DXA_CHECK_ARG(x && z_emb && t_emb && pos && x_w && x_b && final_w && final_b && coef && workspace && packed_table, "%s: null buffer", who);
  DXA_CHECK_ARG(steps >= 1 && A >= 1 && A <= MAXA && nb >= 1 && N == (use_cfg ? 2 * nb : nb),
                "%s: needs 1 <= action_dim <= %d and N == nb (or 2 nb with guidance)", who, MAXA);
  DXA_CHECK_ARG(depth >= 1 && T1 >= 1 && heads >= 1, "%s: bad sizes", who);
  const int M = N * T1;
  DXA_CHECK_ARG(M <= MAXM && T1 <= MAXT, "%s: at most %d rows / %d tokens per sample (got %d / %d)", who, MAXM, MAXT, M, T1);
  DXA_CHECK_ARG(H % 64 == 0 && I % 64 == 0 && H == heads * HD && H <= 1024, "%s: needs head width 64, H <= 1024 and H, I %% 64 == 0", who);
  DXA_CHECK_ARG(nb * (T1 - 1) * A <= 512 && nb * (T1 - 1) * A <= MAXM * MAXA, "%s: the sample has too many elements", who);
  size_t off[8];
  DXA_CHECK_ARG(workspace_bytes >= bf_ws_bytes(M, H, I, off), "%s: workspace too small", who);
  DXA_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) % 256) == 0, "%s: the workspace must be 256-byte aligned", who);
  hipStream_t st = (hipStream_t)stream;
  DitSampleBfP sp;
  DitBfP& p = sp.blk;
  char* ws = reinterpret_cast<char*>(workspace);
  p.h = reinterpret_cast<float*>(ws + off[0]); p.qkv = reinterpret_cast<float*>(ws + off[1]); p.stats = reinterpret_cast<float*>(ws + off[2]);
  p.part = reinterpret_cast<float*>(ws + off[3]); p.hb = reinterpret_cast<bf16_t*>(ws + off[4]); p.ob = reinterpret_cast<bf16_t*>(ws + off[5]);
  p.ab = reinterpret_cast<bf16_t*>(ws + off[6]);
  float* extra = reinterpret_cast<float*>(ws + off[7]);
  unsigned* tail = nullptr;
  if (int rc = get_sync_block(st, &tail)) return rc;
  p.bar = tail; p.cnt_proj = tail + 64; p.cnt_fc2 = tail + 128;
  p.w = packed_table;
  p.kv = per_kv; p.P = P; p.wstride = P > 0 ? 16 : 10;
  p.M = M; p.Mp = (M + 1) & ~1; p.N = N; p.T1 = T1; p.H = H; p.heads = heads; p.I = I; p.depth = depth;
  p.eps = eps; p.scale = 1.f / sqrtf((float)HD);
  static const int dbg = getenv("DXA_DIT_DBG") ? atoi(getenv("DXA_DIT_DBG")) : 0;
  p.dbg = dbg;
  int grid = I / 16;
  if (3 * H / 16 > grid) grid = 3 * H / 16;
  static int resident = 0;
  if (resident == 0) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    DXA_CHECK_HIP(hipGetDevice(&dev));
    DXA_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    DXA_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dit_sample_bf16_k, 512, 0));
    resident = per_cu * prop.multiProcessorCount;
    DXA_CHECK_ARG(resident >= 1, "%s: the kernel does not fit on this device", who);
  }
  if (grid > resident) grid = resident;
  static const int grid_cap = getenv("DXA_DIT_GRID") ? atoi(getenv("DXA_DIT_GRID")) : 0;
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  static const int no_slice = getenv("DXA_DIT_NO_SLICE") ? 1 : 0;
  p.s_proj = no_slice ? 1 : pick_slices(H / 16, H / 64, grid);
  p.s_fc2 = no_slice ? 1 : pick_slices(H / 16, I / 64, grid);
  // the output projection is NOT K-sliced here: with bf16 operands a workgroup's whole K = 768 slab is 25 KB of weights + 52 KB of
  // activations (the qkv phase's load), while a sliced product pays the partial exchange — store, acknowledge, tile counter, gather by
  // the last arrival — after its MFMAs: 3.90 ms per sample with 4 slices, 3.82 with 2, 3.70 with 1 (profiles/r04_proj_slices.txt)
  static const int proj_cap = getenv("DXA_DIT_PROJ_SLICES") ? atoi(getenv("DXA_DIT_PROJ_SLICES")) : 1;
  if (proj_cap > 0 && p.s_proj > proj_cap && (H / 64) % proj_cap == 0) p.s_proj = proj_cap;
  sp.x = x; sp.ze = z_emb; sp.te = t_emb; sp.pos = pos; sp.xw = x_w; sp.xb = x_b; sp.fw = final_w; sp.fb = final_b; sp.coef = coef;
  sp.steps = steps; sp.A = A; sp.nb = nb; sp.use_cfg = use_cfg; sp.cfg_scale = cfg_scale;
  sp.xpp = extra; sp.eps = extra + 2 * MAXM * MAXA;
  hipLaunchKernelGGL(dit_sample_bf16_k, dim3(grid), dim3(512), 0, st, sp);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
}  // namespace

extern "C" int dxa_dit_sample_bf16_fwd(float* x, const float* z_emb, const float* t_emb, const float* pos, const float* x_w,
                                       const float* x_b, const float* final_w, const float* final_b, const float* coef, int steps, int A,
                                       int nb, int use_cfg, float cfg_scale, const void* const* packed_table, int depth, int N, int T1,
                                       int H, int heads, int I, float eps, void* workspace, size_t workspace_bytes, dxa_stream_t stream) {
  return sample_bf16_launch("dxa_dit_sample_bf16_fwd", x, z_emb, t_emb, pos, x_w, x_b, final_w, final_b, coef, steps, A, nb, use_cfg,
                            cfg_scale, packed_table, nullptr, 0, depth, N, T1, H, heads, I, eps, workspace, workspace_bytes, stream);
}

extern "C" int dxa_dit_sample_bf16_per_fwd(float* x, const float* z_emb, const float* t_emb, const float* pos, const float* x_w,
                                           const float* x_b, const float* final_w, const float* final_b, const float* coef, int steps,
                                           int A, int nb, int use_cfg, float cfg_scale, const void* const* packed_table,
                                           const float* per_kv, int P, int depth, int N, int T1, int H, int heads, int I, float eps,
                                           void* workspace, size_t workspace_bytes, dxa_stream_t stream) {
  DXA_CHECK_ARG(per_kv != nullptr && P > 0, "dxa_dit_sample_bf16_per_fwd: needs the perceptual keys / values");
  return sample_bf16_launch("dxa_dit_sample_bf16_per_fwd", x, z_emb, t_emb, pos, x_w, x_b, final_w, final_b, coef, steps, A, nb, use_cfg,
                            cfg_scale, packed_table, per_kv, P, depth, N, T1, H, heads, I, eps, workspace, workspace_bytes, stream);
}
