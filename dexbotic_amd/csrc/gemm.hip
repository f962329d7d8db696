// MFMA GEMM with fused epilogue for gfx950 (see include/dexbotic_amd.h :: dxa_gemm).
//
// Tile 128x128 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 4x4 MFMA 16x16 tiles),
// K-slab of 128 BYTES per row per operand (64 bf16 / 32 fp32) double-buffered in LDS (64 KiB), so
// two workgroups share a CU.  LDS rows are 8 x 16-byte chunks, XOR-swizzled by (row & 7): the
// ds_read_b128 fragment reads of a 16-lane group land on 16 distinct 16-B slots (conflict free).
//
//   k-contiguous operands (A of NT/NN, B of NT)  : 16-B global loads, straight 16-B LDS stores.
//   k-strided operands  (A of TN, B of NN/TN)     : each thread loads 4 rows x (8|4) k, transposes
//                                                  in registers and stores four 16-B chunks.
// MFMA: bf16 -> v_mfma_f32_16x16x32_bf16 (8 k per lane);  fp32 -> 4 x v_mfma_f32_16x16x4_f32 fed from
// one 16-B LDS read (lane group g holds k = 4g..4g+3 of a 16-k block; step j uses element j of both
// operands, i.e. a k-permutation shared by A and B — the sum over the block is unchanged and is an
// exact fp32 fma chain).
// Operands are fed swapped (MFMA "A" = B-tile rows = n, MFMA "B" = A-tile rows = m) so that a lane's
// four accumulator registers are four CONSECUTIVE n for one m: the epilogue does 8/16-byte stores.
// Workgroup ids are remapped XCD-aware (8 XCDs, private L2s): each XCD walks a contiguous range of
// a grouped (8 tile-rows) ordering so neighbouring tiles share A/B panels in that XCD's L2.
#include "common.h"
#include "gemm_common.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace {

// (typedefs, GemmP, load4 / store4 / epilogue4, the split-K hand-off, tile_finish and sk_epilogue live in gemm_common.h)

template <typename T> struct EltTraits;
template <> struct EltTraits<float> { static constexpr int EPC = 4; };   // elements per 16-B chunk
template <> struct EltTraits<bf16_t> { static constexpr int EPC = 8; };

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }

// ---- k-contiguous staging: thread -> (chunk c = tid&7, rows (tid>>3)+32*i) --------------------------
template <typename T, int NR>
__device__ __forceinline__ void load_kc(uint4 (&r)[4], const T* __restrict__ base, int64_t ld, int64_t row0,
                                        int64_t rows, int64_t k0, int64_t K, int vec, int tid) {
  constexpr int EPC = EltTraits<T>::EPC;
  const int c = tid & 7;
  const int64_t k = k0 + (int64_t)c * EPC;
#pragma unroll
  for (int i = 0; i < NR / 32; ++i) {
    const int64_t row = row0 + (tid >> 3) + 32 * i;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < rows && k < K) {
      const T* p = base + row * ld + k;
      if (vec && k + EPC <= K) {
        v = *reinterpret_cast<const uint4*>(p);
      } else {
        alignas(16) T tmp[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) tmp[e] = (k + e < K) ? p[e] : (T)0;
        v = *reinterpret_cast<const uint4*>(tmp);
      }
    }
    r[i] = v;
  }
}
template <int NR>
__device__ __forceinline__ void store_kc(const uint4 (&r)[4], char* tile, int tid) {
  const int c = tid & 7;
#pragma unroll
  for (int i = 0; i < NR / 32; ++i) {
    const int row = (tid >> 3) + 32 * i;
    *reinterpret_cast<uint4*>(tile + lds_off(row, c)) = r[i];
  }
}

// ---- k-strided staging: element (row, k) at base[k*ld + row]; thread -> (chunk kg = tid&7, rows 4*(tid>>3)+q)
template <typename T> struct KsRegs;
template <> struct KsRegs<bf16_t> { uint2 v[8]; };   // v[j] = 4 rows for k_j
template <> struct KsRegs<float> { uint4 v[4]; };

template <typename T, int NR>
__device__ __forceinline__ void load_ks(KsRegs<T>& r, const T* __restrict__ base, int64_t ld, int64_t row0,
                                        int64_t rows, int64_t k0, int64_t K, int vec, int tid) {
  constexpr int EPC = EltTraits<T>::EPC;
  const int kg = tid & 7;
  const int64_t rr = row0 + 4 * (tid >> 3);
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    const int64_t k = k0 + kg * EPC + j;
    alignas(16) T tmp[4] = {(T)0, (T)0, (T)0, (T)0};
    if (k < K && rr < rows && 4 * (tid >> 3) < NR) {
      const T* p = base + k * ld + rr;
      if (vec && rr + 3 < rows) {
        if constexpr (sizeof(T) == 2) {
          *reinterpret_cast<uint2*>(tmp) = *reinterpret_cast<const uint2*>(p);
        } else {
          *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(p);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) tmp[q] = (rr + q < rows) ? p[q] : (T)0;
      }
    }
    if constexpr (sizeof(T) == 2) {
      r.v[j] = *reinterpret_cast<const uint2*>(tmp);
    } else {
      r.v[j] = *reinterpret_cast<const uint4*>(tmp);
    }
  }
}
__device__ __forceinline__ uint32_t half_of(const uint2& v, int q) {
  const uint32_t w = (q & 2) ? v.y : v.x;
  return (q & 1) ? (w >> 16) : (w & 0xffffu);
}
template <int NR>
__device__ __forceinline__ void store_ks(const KsRegs<bf16_t>& r, char* tile, int tid) {
  const int kg = tid & 7;
  if (4 * (tid >> 3) >= NR) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * (tid >> 3) + q;
    uint4 o;
    o.x = half_of(r.v[0], q) | (half_of(r.v[1], q) << 16);
    o.y = half_of(r.v[2], q) | (half_of(r.v[3], q) << 16);
    o.z = half_of(r.v[4], q) | (half_of(r.v[5], q) << 16);
    o.w = half_of(r.v[6], q) | (half_of(r.v[7], q) << 16);
    *reinterpret_cast<uint4*>(tile + lds_off(row, kg)) = o;
  }
}
__device__ __forceinline__ uint32_t comp_of(const uint4& v, int q) {
  return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w));
}
template <int NR>
__device__ __forceinline__ void store_ks(const KsRegs<float>& r, char* tile, int tid) {
  const int kg = tid & 7;
  if (4 * (tid >> 3) >= NR) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * (tid >> 3) + q;
    uint4 o = make_uint4(comp_of(r.v[0], q), comp_of(r.v[1], q), comp_of(r.v[2], q), comp_of(r.v[3], q));
    *reinterpret_cast<uint4*>(tile + lds_off(row, kg)) = o;
  }
}

template <typename T, bool KS, int NR> struct Stager;
template <typename T, int NR> struct Stager<T, false, NR> {
  uint4 r[4];
  __device__ __forceinline__ void load(const T* base, int64_t ld, int64_t row0, int64_t rows, int64_t k0,
                                       int64_t K, int vec, int tid) {
    load_kc<T, NR>(r, base, ld, row0, rows, k0, K, vec, tid);
  }
  __device__ __forceinline__ void store(char* tile, int tid) { store_kc<NR>(r, tile, tid); }
};
template <typename T, int NR> struct Stager<T, true, NR> {
  KsRegs<T> r;
  __device__ __forceinline__ void load(const T* base, int64_t ld, int64_t row0, int64_t rows, int64_t k0,
                                       int64_t K, int vec, int tid) {
    load_ks<T, NR>(r, base, ld, row0, rows, k0, K, vec, tid);
  }
  __device__ __forceinline__ void store(char* tile, int tid) { store_ks<NR>(r, tile, tid); }
};

template <typename T>
__device__ __forceinline__ void mma_step(f32x4_t& acc, const uint4& first, const uint4& second);
template <>
__device__ __forceinline__ void mma_step<bf16_t>(f32x4_t& acc, const uint4& first, const uint4& second) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, first),
                                                __builtin_bit_cast(bf16x8_t, second), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<float>(f32x4_t& acc, const uint4& first, const uint4& second) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(first.x), __uint_as_float(second.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(first.y), __uint_as_float(second.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(first.z), __uint_as_float(second.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(first.w), __uint_as_float(second.w), acc, 0, 0, 0);
}


// TM = 16x16 fragments per wave per dimension: 4 -> 128x128 tile (default), 2 -> 64x64 tile (small problems that
// would otherwise leave most of the 256 CUs idle, e.g. the fp32 DiT head GEMMs with M = 1088)
template <typename TI, typename TO, bool A_KS, bool B_KS, int TM>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
  constexpr int BT = 32 * TM;                 // tile rows / cols
  constexpr int TILE_B = BT * ROWB;           // bytes per operand per buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buf][A,B][TILE_B]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, lg = lane >> 4;

  // ---- XCD-aware, grouped tile order --------------------------------------------------------------
  const int nt = p.tm * p.tn;
  int bid = blockIdx.x;
  {
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP_M = 8;
  const int width = GROUP_M * p.tn;
  const int group = bid / width;
  const int first_pm = group * GROUP_M;
  const int gsz = min(p.tm - first_pm, GROUP_M);
  const int pm = first_pm + (bid % width) % gsz;
  const int pn = (bid % width) / gsz;
  const int64_t m0 = (int64_t)pm * BT, n0 = (int64_t)pn * BT;

  // ---- batch offsets ----------------------------------------------------------------------------------
  const int z = blockIdx.z;
  const int b2 = z % p.nb2, b1 = (z / p.nb2) % p.nb1, b0 = z / (p.nb2 * p.nb1);
  const TI* A = reinterpret_cast<const TI*>(p.A) + b0 * p.sA[0] + b1 * p.sA[1] + b2 * p.sA[2];
  const TI* B = reinterpret_cast<const TI*>(p.B) + b0 * p.sB[0] + b1 * p.sB[1] + b2 * p.sB[2];
  const int64_t offC = b0 * p.sC[0] + b1 * p.sC[1] + b2 * p.sC[2];

  constexpr int EPC = EltTraits<TI>::EPC;
  constexpr int BK = 8 * EPC;
  const int64_t nk = (p.K + BK - 1) / BK;

  f32x4_t acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  Stager<TI, A_KS, BT> sa;
  Stager<TI, B_KS, BT> sb;
  sa.load(A, p.lda, m0, p.M, 0, p.K, p.vecA, tid);
  sb.load(B, p.ldb, n0, p.N, 0, p.K, p.vecB, tid);
  sa.store(smem, tid);
  sb.store(smem + TILE_B, tid);
  __syncthreads();

  for (int64_t kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    char* As = smem + cur * 2 * TILE_B;
    char* Bs = As + TILE_B;
    const bool more = kt + 1 < nk;
    if (more) {
      sa.load(A, p.lda, m0, p.M, (kt + 1) * BK, p.K, p.vecA, tid);
      sb.load(B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, p.vecB, tid);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 af[TM], bfr[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        af[i] = *reinterpret_cast<const uint4*>(As + lds_off(wm * 16 * TM + i * 16 + l16, 4 * s + lg));
        bfr[i] = *reinterpret_cast<const uint4*>(Bs + lds_off(wn * 16 * TM + i * 16 + l16, 4 * s + lg));
      }
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TM; ++ni) mma_step<TI>(acc[mi][ni], bfr[ni], af[mi]);
    }
    if (more) {
      char* An = smem + (cur ^ 1) * 2 * TILE_B;
      sa.store(An, tid);
      sb.store(An + TILE_B, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds, per (mi,ni), m = .. + l16 and n = .. + 4*lg + {0..3} ---------------------
  TO* C = reinterpret_cast<TO*>(p.C) + offC;
  TO* AUX = p.aux ? reinterpret_cast<TO*>(p.aux) + offC : nullptr;
  const TI* R = p.R ? reinterpret_cast<const TI*>(p.R) + b0 * p.sR[0] + b1 * p.sR[1] + b2 * p.sR[2] : nullptr;
  const TI* G = p.G ? reinterpret_cast<const TI*>(p.G) + b0 * p.sG[0] + b1 * p.sG[1] + b2 * p.sG[2] : nullptr;
  const TI* bias = reinterpret_cast<const TI*>(p.bias);
#pragma unroll
  for (int ni = 0; ni < TM; ++ni) {
    const int64_t n = n0 + wn * 16 * TM + ni * 16 + 4 * lg;
    if (n >= p.N) continue;
    const int n_ok = (int)min((int64_t)4, p.N - n);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) load4<TI>(bv, bias + n, p.vecBias, n_ok);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int64_t m = m0 + wm * 16 * TM + mi * 16 + l16;
      if (m >= p.M) continue;
      const float a4[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      epilogue4<TI, TO>(p, C, AUX, R, G, bv, m, n, n_ok, a4);
    }
  }
}

// =====================================================================================================
// Few-row bf16 NT kernel: 128x128 tiles, 4 waves, LDS-DMA ring.  The batch-1 action request runs every linear of the
// ViT (2 x 257 tokens) and of the decoder prefill (S = 543) on a few hundred rows: 256-row tiles waste up to 29 % of
// their MFMA work on padding (543 = 2 x 256 + 31) and leave most of the 256 CUs without a tile (3 x 18 for qkv, 3 x 14
// for o_proj).  Here a workgroup of 4 waves (2 x 2, 64 x 64 per wave = 2 x 2 v_mfma_f32_32x32x16_bf16 blocks) owns a
// 128 x 128 tile: 5 x 36 = 180 tiles for qkv.  K tile = 64 (whole 128-byte lines), operands staged by LDS-DMA into a ring
// of FOUR 32 KiB stages (three K tiles = 96 KiB in flight: with one wave per SIMD nothing else hides the L2 latency),
// same LDS image as the ping-pong kernel (16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7), swizzle applied on
// the SOURCE address).  One barrier per K tile; inside a tile the fragment reads of k-step s+1 are in flight under the
// MFMAs of k-step s (counted lgkmcnt).  The epilogue runs from the registers through epilogue4 (bias / activation / aux /
// mulgrad / residual / accumulate: the whole menu).  Tiles are ordered row-tile fastest inside an XCD's contiguous run, so
// the row tiles that share a weight panel hit the same L2.
// Requirements: K % 64 == 0, 16-byte aligned A/B rows, no batching, operands < 2 GiB.
// =====================================================================================================
// WS (wave-specialised, 8 waves): waves 4-7 do nothing but issue the LDS-DMA of the ring, waves 0-3 nothing but fragment reads
// and MFMAs.  With one wave per SIMD a K tile costs its wave ~0.4 us of DMA issue (8 instructions at 60-180 issue cycles each)
// PLUS ~0.36 us of reads and MFMAs, back to back: 0.8 us per K tile measured (qkv at M = 543: 45 us for 56 K tiles).  With a
// loader wave and a compute wave on every SIMD the two run side by side and a K tile costs the longer of them.
// TE: element type of bias / residual / mulgrad (float for the split-bf16 fp32 products of the action head, dxa_gemm_desc.epi_f32)
template <typename TO, int NS, int WS, typename TE = bf16_t>
__global__ __launch_bounds__(256 + 64 * WS) void gemm_nt_t128_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 32768, A_ST = 16384;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = WS ? wave_id >= 4 : true;          // issues the DMA
  const bool worker = WS ? wave_id < 4 : true;           // reads fragments, runs the MFMAs and the epilogue
  const int wave = wave_id & 3;                           // index of a compute wave
  constexpr int NLW = WS ? WS : 4;                        // waves that share a K tile's 32 DMA instructions
  constexpr int DPW = 16 / NLW;                           // ... A and B instructions each per loader wave and K tile
  const int lw = WS ? (wave_id - 4 < 0 ? 0 : wave_id - 4) : wave_id;
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, lh = lane >> 5;
  const int nt = p.tm * p.tn;
  // split-K (few tiles, deep K: ViT fc2 / o_proj on 514 rows): block ids [j * nt, (j + 1) * nt) are K slice j of every tile; the
  // slices 0 .. S-2 leave their fp32 partial in a lane-linear 64 KiB slot (write-through) and count themselves in, slice S-1 —
  // the highest block ids of a tile, dispatched after its partners — waits for them, adds the partials in slice order and
  // runs the epilogue: deterministic, same protocol as the tail split of the 256-row kernels
  const int split_s = p.split_s > 1 ? p.split_s : 1;
  const int split_j = (int)blockIdx.x / nt;
  int bid = (int)blockIdx.x - split_j * nt;
  {
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int pm = bid % p.tm, pn = bid / p.tm;
  const int m0 = pm * 128, n0 = pn * 128;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.A), 0, (int)(((p.M - 1) * p.lda + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, (int)(((p.N - 1) * p.ldb + p.K) * 2), 0x00020000);
  // DMA instruction j of a wave covers tile rows (4 wave + j) * 8 .. + 7: lane -> (row = + lane / 8, LDS chunk slot = lane % 8,
  // source chunk = slot ^ ((row >> 1) & 7))
  uint32_t voA[DPW], voB[DPW];
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    const int row = (lw * DPW + j) * 8 + (lane >> 3);
    const uint32_t ch = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    const int ga = m0 + row, gb = n0 + row;
    voA[j] = ga < (int)p.M ? (uint32_t)ga * (uint32_t)p.lda * 2u + ch : 0x80000000u;
    voB[j] = gb < (int)p.N ? (uint32_t)gb * (uint32_t)p.ldb * 2u + ch : 0x80000000u;
  }
  const int nk_tot = (int)(p.K >> 6);
  const int k_lo = split_j * nk_tot / split_s;
  const int nk = (split_j + 1) * nk_tot / split_s - k_lo;      // K tiles of this workgroup (>= 1)
#define T1_DMA(slot, t)                                                                                                  \
  do {                                                                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < DPW; ++j_) {                                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t*)(smem + (slot) * STAGE + (lw * DPW + j_) * 1024), 16, voA[j_], (k_lo + (t)) * 128, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t*)(smem + (slot) * STAGE + A_ST + (lw * DPW + j_) * 1024), 16, voB[j_], (k_lo + (t)) * 128, 0, 0); \
    }                                                                                                                    \
  } while (0)
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int sw = (l32 >> 1) & 7;
  const uint32_t ya = (lds0 + (wm * 64 + l32) * 128) | (uint32_t)((lh ^ sw) << 4);
  const uint32_t yb = (lds0 + A_ST + (wn * 64 + l32) * 128) | (uint32_t)((lh ^ sw) << 4);
#define T1_READ(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define T1_SB() __builtin_amdgcn_sched_barrier(0)
  // prologue: three K tiles in flight
  if (loader) {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nk) T1_DMA(s, s);
  }
  for (int t = 0; t < nk; ++t) {
    const int rem = nk - 1 - t;                  // K tiles after this one (up to NS - 2 of them already requested)
    if (loader) {
      // (2 DPW instructions per K tile and loader wave still in flight for each of the tiles after this one)
      if (NS >= 4 && rem >= 2) { if (DPW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
      else if (NS >= 3 && rem >= 1) { if (DPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    T1_SB();
    __builtin_amdgcn_s_barrier();                // tile t landed for every wave; every wave is done reading tile t - 1
    T1_SB();
    if (loader && t + NS - 1 < nk) T1_DMA((t + NS - 1) % NS, t + NS - 1);
    T1_SB();
    if (!worker) continue;
    const uint32_t so = (uint32_t)((t % NS) * STAGE);
    const uint32_t a_s = ya + so, b_s = yb + so;
    u32x4_t af[2][2], bf[2][2];                  // [k-step parity][block]
    T1_READ(af[0][0], a_s, 0); T1_READ(af[0][1], a_s, 4096); T1_READ(bf[0][0], b_s, 0); T1_READ(bf[0][1], b_s, 4096);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks & 1, nx = c ^ 1;
      if (ks < 3) {
        const uint32_t a_k = a_s ^ (uint32_t)((ks + 1) << 5), b_k = b_s ^ (uint32_t)((ks + 1) << 5);
        T1_READ(af[nx][0], a_k, 0); T1_READ(af[nx][1], a_k, 4096); T1_READ(bf[nx][0], b_k, 0); T1_READ(bf[nx][1], b_k, 4096);
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      T1_SB();
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bf[c][j]), __builtin_bit_cast(bf16x8_t, af[c][i]), acc[i][j], 0, 0, 0);
      T1_SB();
    }
  }
#undef T1_DMA
#undef T1_READ
#undef T1_SB
  if (split_s > 1) {
    constexpr int SC1 = 16;
    constexpr uint32_t SLOT = 128 * 128 * 4;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)(1024u * SLOT), 0x00020000);
    const uint32_t slot0 = (uint32_t)bid * (uint32_t)(split_s - 1) * SLOT + (uint32_t)(tid & 255) * 16u;
    if (split_j < split_s - 1) {
      if (worker)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rW,
                                                   slot0 + (uint32_t)split_j * SLOT + ((i * 2 + j) * 4 + q) * 4096, 0, SC1);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.flags + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      while (__hip_atomic_load(p.flags + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < split_s - 1)
        __builtin_amdgcn_s_sleep(4);
      __hip_atomic_store(p.flags + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (worker)
    for (int sj = 0; sj < split_s - 1; ++sj) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
                rW, slot0 + (uint32_t)sj * SLOT + ((i * 2 + j) * 4 + q) * 4096, 0, SC1));
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][j][4 * q + c] += v[c];
          }
    }
  }
  if (!worker) return;
  // ---- epilogue from the registers: lane (l32, lh) holds, per block (i, j), row 32 i + l32 and columns 32 j + 8 q + 4 lh + {0..3}
  TO* C = reinterpret_cast<TO*>(p.C);
  TO* AUX = p.aux ? reinterpret_cast<TO*>(p.aux) : nullptr;
  const TE* R = reinterpret_cast<const TE*>(p.R);
  const TE* G = reinterpret_cast<const TE*>(p.G);
  const TE* bias = reinterpret_cast<const TE*>(p.bias);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t n = n0 + wn * 64 + 32 * j + 8 * q + 4 * lh;
      if (n >= p.N) continue;
      const int n_ok = (int)min((int64_t)4, p.N - n);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) load4<TE>(bv, bias + n, p.vecBias, n_ok);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wm * 64 + 32 * i + l32;
        if (m >= p.M) continue;
        const float a4[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        epilogue4<TE, TO>(p, C, AUX, R, G, bv, m, n, n_ok, a4);
      }
    }
#endif
}
template __global__ void gemm_nt_t128_kernel<bf16_t, 2, 0>(const GemmP);
template __global__ void gemm_nt_t128_kernel<float, 2, 0>(const GemmP);
template __global__ void gemm_nt_t128_kernel<bf16_t, 4, 0>(const GemmP);
template __global__ void gemm_nt_t128_kernel<float, 4, 0>(const GemmP);
template __global__ void gemm_nt_t128_kernel<bf16_t, 4, 4>(const GemmP);
template __global__ void gemm_nt_t128_kernel<float, 4, 4>(const GemmP);
template __global__ void gemm_nt_t128_kernel<bf16_t, 4, 8>(const GemmP);
template __global__ void gemm_nt_t128_kernel<float, 4, 8>(const GemmP);
template __global__ void gemm_nt_t128_kernel<float, 4, 4, float>(const GemmP);

// =====================================================================================================
// Few-row NN product (M <= 8): out[M, N] = A[M, K] W[K, N] with W as it lies — the dX of a linear layer applied to a handful
// of tokens (MemVLA's one-token cognition stream through its retrieval blocks, memvla_arch.py:194-216: dy [1, 14336] against a
// [14336, 3584] weight).  It is a stream over W: thread -> 8 consecutive output columns (16-byte loads of W rows), the K range is
// cut into `ksplit` slices so that a few hundred workgroups share the stream, per-slice fp32 partials go to the split-K scratch
// and a second launch adds them in slice order (deterministic).  The 64x64-tile kernel it replaces spent 93 us per call.
// =====================================================================================================
constexpr int GV_MAXM = 8;
template <int M_>
__global__ __launch_bounds__(256) void gemv_nn_stage1_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W,
                                                        int64_t ldb, float* __restrict__ part, int M, int64_t N, int64_t K,
                                                        int kper) {
  __shared__ float as[GV_MAXM][512];
  const int ks = blockIdx.y;
  const int64_t k0 = (int64_t)ks * kper, k1 = min(K, k0 + kper);
  for (int i = threadIdx.x; i < M_ * kper; i += 256) {
    const int m = i / kper, kk = i - m * kper;
    as[m][kk] = (m < M && k0 + kk < k1) ? bf2f(A[(int64_t)m * lda + k0 + kk]) : 0.f;
  }
  __syncthreads();
  const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (n >= N) return;
  float acc[M_][8];
#pragma unroll
  for (int m = 0; m < M_; ++m)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[m][e] = 0.f;
  const bf16_t* wp = W + k0 * ldb + n;
  const int nkk = (int)(k1 - k0);
  int kk = 0;
  for (; kk + 4 <= nkk; kk += 4) {                     // four independent 16-byte loads in flight
    float w[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) Vec<bf16_t, 8>::ld(w[u], wp + (int64_t)(kk + u) * ldb);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int m = 0; m < M_; ++m) {
        const float a = as[m][kk + u];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[m][e] += a * w[u][e];
      }
  }
  for (; kk < nkk; ++kk) {
    float w[8];
    Vec<bf16_t, 8>::ld(w, wp + (int64_t)kk * ldb);
#pragma unroll
    for (int m = 0; m < M_; ++m) {
      const float a = as[m][kk];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[m][e] += a * w[e];
    }
  }
#pragma unroll
  for (int m = 0; m < M_; ++m)
    if (m < M) {
      float* pp = part + ((int64_t)ks * M + m) * N + n;
      *reinterpret_cast<float4*>(pp) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
      *reinterpret_cast<float4*>(pp + 4) = make_float4(acc[m][4], acc[m][5], acc[m][6], acc[m][7]);
    }
}
template <typename TO>
__global__ __launch_bounds__(256) void gemv_nn_stage2_k(const float* __restrict__ part, TO* __restrict__ C, int64_t ldc, int M,
                                                        int64_t N, int ksplit, float alpha) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  const int64_t m = i / N, n = i - m * N;
  float s = 0.f;
  for (int ks = 0; ks < ksplit; ++ks) s += part[((int64_t)ks * M + m) * N + n];
  stf<TO>(C + m * ldc + n, s * alpha);
}

// =====================================================================================================
// Skinny fp32 NT kernel (M <= 64 rows): the DiT head at inference time is ~50 linears per DDIM step on 36 rows
// (2 x 18 tokens), i.e. a stream over each weight matrix with almost no arithmetic.  The tiled kernel puts such a
// problem on N/64 workgroups that each walk all of K serially (measured 44 us per call).  Here a workgroup owns
// 16 output columns, its 8 waves split K eight ways (64-k blocks, round robin) reading W and A straight from
// L2/HBM into MFMA operands (exact fp32 v_mfma_f32_16x16x4_f32, same k permutation as mma_step<float>), the
// eight partial accumulators are folded through LDS and the first MB waves run the usual fused epilogue.
// Requirements: K % 64 == 0, 16-byte aligned A/B rows, no batching.
// =====================================================================================================
template <int MB>
__global__ __launch_bounds__(512) void gemm_skinny_f32_kernel(const GemmP p) {
  __shared__ float red[8][MB][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lg = lane >> 4;
  // workgroup = (16-column tile, K range): split_s ranges per tile when few tiles meet a deep K (a DiT-L fc2 is 64 tiles
  // of K 4096: 64 CUs would each walk 8 dependent load rounds, 30 us); ranges of one tile have ascending block ids
  const int split_s = p.split_s > 1 ? p.split_s : 1;
  const int ntile = (int)gridDim.x / split_s;
  const int tile = (int)blockIdx.x % ntile, split_j = (int)blockIdx.x / ntile;
  const int64_t n0 = (int64_t)tile * 16;
  const float* W = reinterpret_cast<const float*>(p.B) + min(n0 + l16, p.N - 1) * p.ldb + 4 * lg;
  const float* A[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    A[mb] = reinterpret_cast<const float*>(p.A) + min((int64_t)mb * 16 + l16, p.M - 1) * p.lda + 4 * lg;
  f32x4_t acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nkb = (int)(p.K / 64);
  const int kb_lo = split_j * nkb / split_s, kb_hi = (split_j + 1) * nkb / split_s;
  auto ld = [&](int kb, float4 (&wv)[4], float4 (&av)[MB][4]) {
    const int k0 = kb * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = *reinterpret_cast<const float4*>(W + k0 + 16 * j);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) av[mb][j] = *reinterpret_cast<const float4*>(A[mb] + k0 + 16 * j);
  };
  auto mma = [&](const float4 (&wv)[4], const float4 (&av)[MB][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j].x, av[mb][j].x, acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j].y, av[mb][j].y, acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j].z, av[mb][j].z, acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j].w, av[mb][j].w, acc[mb], 0, 0, 0);
      }
  };
  // the wave's 64-k blocks: every 8th of the range (two blocks' loads in flight at once measured the same: the rounds are
  // bound by the 32-cycle fp32 MFMAs and the A re-reads, not by the round trip)
  for (int kb = kb_lo + wave; kb < kb_hi; kb += 8) {
    float4 w0[4], a0[MB][4];
    ld(kb, w0, a0);
    mma(w0, a0);
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    *reinterpret_cast<float4*>(red[wave][mb][lane]) = make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
  __syncthreads();
  // wave mb folds the 8 partials of row block mb: lane holds out[m = 16 mb + l16][n0 + 4 lg + {0..3}]
  const bool worker = wave < MB;
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  if (worker) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(red[w][wave][lane]);
      a4[0] += v.x; a4[1] += v.y; a4[2] += v.z; a4[3] += v.w;
    }
  }
  if (split_s > 1) {
    // split-K protocol of the ring kernel: ranges 0 .. S-2 leave their partial tile in the scratch (write-through stores) and
    // count themselves in; the last range — dispatched after its partners — waits for the count, adds the partials in range
    // order (deterministic) and runs the epilogue
    constexpr int SC1 = 16;
    constexpr uint32_t SLOT = MB * 64 * 16;                  // bytes per partial tile
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)(256u * 7u * 4u * 64u * 16u), 0x00020000);
    const uint32_t slot0 = (uint32_t)tile * (uint32_t)(split_s - 1) * SLOT + (uint32_t)(wave * 64 + lane) * 16u;
    if (split_j < split_s - 1) {
      if (worker) {
        const f32x4_t v = {a4[0], a4[1], a4[2], a4[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rW, slot0 + (uint32_t)split_j * SLOT, 0, SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.flags + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      while (__hip_atomic_load(p.flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < split_s - 1)
        __builtin_amdgcn_s_sleep(2);
      __hip_atomic_store(p.flags + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (worker) {
      float own[4] = {a4[0], a4[1], a4[2], a4[3]};
      a4[0] = a4[1] = a4[2] = a4[3] = 0.f;
      for (int sj = 0; sj < split_s - 1; ++sj) {            // K ranges in ascending order, this workgroup's (the last) at the end
        const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rW, slot0 + (uint32_t)sj * SLOT, 0, SC1));
        a4[0] += v[0]; a4[1] += v[1]; a4[2] += v[2]; a4[3] += v[3];
      }
      a4[0] += own[0]; a4[1] += own[1]; a4[2] += own[2]; a4[3] += own[3];
    }
  }
  if (!worker) return;
  const int64_t m = (int64_t)wave * 16 + l16, n = n0 + 4 * lg;
  if (m >= p.M || n >= p.N) return;
  const int n_ok = (int)min((int64_t)4, p.N - n);
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  const float* bias = reinterpret_cast<const float*>(p.bias);
  if (bias) load4<float>(bv, bias + n, p.vecBias, n_ok);
  epilogue4<float, float>(p, reinterpret_cast<float*>(p.C), reinterpret_cast<float*>(p.aux),
                          reinterpret_cast<const float*>(p.R), reinterpret_cast<const float*>(p.G), bv, m, n, n_ok, a4);
}

// bf16 twin of the skinny kernel (M <= 64): KV-cached decode (M = batch) and other few-row products are a pure
// stream over the weight matrix.  Same decomposition — 16 output columns per workgroup, 8 waves splitting K in
// 64-k blocks — with v_mfma_f32_16x16x32_bf16: a lane's 16-byte load IS its MFMA operand (8 consecutive k).
template <int MB, typename TO, int U = 2>
__global__ __launch_bounds__(512) void gemm_skinny_bf16_kernel(const GemmP p) {
  __shared__ float red[8][MB][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lg = lane >> 4;
  // (16-column tile, K range) per workgroup, as in the fp32 kernel: few tiles over a deep K are cut across workgroups
  const int split_s = p.split_s > 1 ? p.split_s : 1;
  const int ntile = (int)gridDim.x / split_s;
  const int tile = (int)blockIdx.x % ntile, split_j = (int)blockIdx.x / ntile;
  const int64_t n0 = (int64_t)tile * 16;
  const bf16_t* W = reinterpret_cast<const bf16_t*>(p.B) + min(n0 + l16, p.N - 1) * p.ldb + 8 * lg;
  const bf16_t* A[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    A[mb] = reinterpret_cast<const bf16_t*>(p.A) + min((int64_t)mb * 16 + l16, p.M - 1) * p.lda + 8 * lg;
  f32x4_t acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nkb = (int)(p.K / 64);
  const int kb_lo = split_j * nkb / split_s, kb_hi = (split_j + 1) * nkb / split_s;
  // U blocks of 64 k in flight per wave, written out by hand: the compiler REFUSES "#pragma unroll" on this loop (runtime trip
  // count; -Wpass-failed), so until round 5 a wave had ONE block = 2 KiB of W in flight, and the products with 224 column tiles
  // (o_proj, down_proj of the decode step: one 8-wave workgroup per CU) 16 KiB per CU against the ~50 KiB that 6 TB/s need.
  // Blocks past the range are zeros (0 * a adds nothing); the order in which a wave adds its blocks is unchanged.
  for (int kb = kb_lo + wave; kb < kb_hi; kb += 8 * U) {
    u32x4n_t wv[U][2];
    uint4 av[U][MB][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k0 = (kb + 8 * u) * 64;
      if (kb + 8 * u < kb_hi) {
#pragma unroll
        for (int j = 0; j < 2; ++j) wv[u][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4n_t*>(W + k0 + 32 * j));
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int j = 0; j < 2; ++j) av[u][mb][j] = *reinterpret_cast<const uint4*>(A[mb] + k0 + 32 * j);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wv[u][j] = (u32x4n_t){0u, 0u, 0u, 0u};
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) av[u][mb][j] = make_uint4(0, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[u][j]),
                                                            __builtin_bit_cast(bf16x8_t, av[u][mb][j]), acc[mb], 0, 0, 0);
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    *reinterpret_cast<float4*>(red[wave][mb][lane]) = make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
  __syncthreads();
  const bool worker = wave < MB;
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  if (worker) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(red[w][wave][lane]);
      a4[0] += v.x; a4[1] += v.y; a4[2] += v.z; a4[3] += v.w;
    }
  }
  if (split_s > 1) {                                         // the ordered gather of gemm_skinny_f32_kernel
    constexpr int SC1 = 16;
    constexpr uint32_t SLOT = MB * 64 * 16;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)(256u * 7u * 4u * 64u * 16u), 0x00020000);
    const uint32_t slot0 = (uint32_t)tile * (uint32_t)(split_s - 1) * SLOT + (uint32_t)(wave * 64 + lane) * 16u;
    if (split_j < split_s - 1) {
      if (worker) {
        const f32x4_t v = {a4[0], a4[1], a4[2], a4[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rW, slot0 + (uint32_t)split_j * SLOT, 0, SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.flags + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      while (__hip_atomic_load(p.flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < split_s - 1)
        __builtin_amdgcn_s_sleep(2);
      __hip_atomic_store(p.flags + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (worker) {
      const float own[4] = {a4[0], a4[1], a4[2], a4[3]};
      a4[0] = a4[1] = a4[2] = a4[3] = 0.f;
      for (int sj = 0; sj < split_s - 1; ++sj) {
        const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rW, slot0 + (uint32_t)sj * SLOT, 0, SC1));
        a4[0] += v[0]; a4[1] += v[1]; a4[2] += v[2]; a4[3] += v[3];
      }
      a4[0] += own[0]; a4[1] += own[1]; a4[2] += own[2]; a4[3] += own[3];
    }
  }
  if (!worker) return;
  const int64_t m = (int64_t)wave * 16 + l16, n = n0 + 4 * lg;
  if (m >= p.M || n >= p.N) return;
  const int n_ok = (int)min((int64_t)4, p.N - n);
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(p.bias);
  if (bias) load4<bf16_t>(bv, bias + n, p.vecBias, n_ok);
  epilogue4<bf16_t, TO>(p, reinterpret_cast<TO*>(p.C), reinterpret_cast<TO*>(p.aux), reinterpret_cast<const bf16_t*>(p.R),
                        reinterpret_cast<const bf16_t*>(p.G), bv, m, n, n_ok, a4);
}


// =====================================================================================================
// Fast path: bf16 NT "ring" kernel — 256x256 tile, 512 threads, LDS-DMA ring, ONE barrier per K slab.
//
// Measured on MI355X (8192^3, ablation builds of the staggered kernel): an s_barrier over 8 waves costs
// ~100-200 cycles, so its 8 barriers per 64-deep slab cap that skeleton near 1.4 PF even with no loads.
// Here the K slab is 32 deep (64-byte LDS rows) and LDS holds a ring of FOUR slabs (4 x 32 KiB):
//   * LDS-DMA runs two slabs ahead: during slab t a wave issues its 4 pieces of slab t+3 and, before the
//     barrier that ends slab t, retires its pieces of slab t+2 (s_waitcnt vmcnt(4)).  After that barrier
//     slabs <= t+2 are complete for every wave, slab t-1's buffer is free for slab t+3.
//   * each wave software-pipelines itself at k-step (16) granularity with two fragment register sets:
//       lgkmcnt(0) | 2 MFMA(ks0) | 6 ds_read ks1 -> R1 | 6 MFMA(ks0) + 4 LDS-DMA |
//       lgkmcnt(0) | 2 MFMA(ks1) | 6 ds_read ks0 of slab t+1 -> R0 | 6 MFMA(ks1) | vmcnt | s_barrier
//     so fragment reads and DMA issue sit in the 32-cycle gaps of the wave's own MFMAs and the two waves
//     of a SIMD need no role split.
//   * 64-byte rows: chunk ^ ((row>>2)&3) makes the 16-lane groups of ds_read_b128 conflict free for the
//     32-row fragments (4 tile rows per 256-B bank row).
// Requirements: K % 32 == 0, 16-byte aligned A/B rows, no batching, operands < 2 GiB.
// =====================================================================================================
// AI = 32-row A blocks per wave: 4 -> 256-row tiles, 3 -> 192-row tiles (few-row products such as the B=1 prefill,
// M = 543: 3 x 192 wastes 6% of the MFMA work, 3 x 256 wastes 29%).
template <typename TO, int AI, typename TE>
__global__ __launch_bounds__(512) void gemm_nt_ring_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_ST = 256 * 64, STAGE = 2 * A_ST;   // 16 KiB + 16 KiB per slab
  constexpr int BMR = AI * 64;                       // tile rows; A has BMR / 16 DMA pieces per slab
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l32 = lane & 31, lh = lane >> 5;

  // tiles [0, full): one workgroup each, block ids dealt round-robin to the 8 XCDs are remapped so every XCD
  // walks a contiguous run of tiles.  Tail tiles (the partial last round of the 256 CUs) are cut along K into
  // split_s workgroups each; the last split gathers the others' fp32 partials and runs the epilogue.
  int bid = blockIdx.x;
  int split_j = 0, split_s = 1, tail_i = 0;
  if (bid < p.full) {
    const int q = p.full >> 3, r = p.full & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  } else {
    const int idx = bid - p.full;
    tail_i = idx % p.tail_r;
    split_j = idx / p.tail_r;
    split_s = p.split_s;
    bid = p.full + tail_i;
  }
  const int GROUP_M = p.group_m;
  const int width = GROUP_M * p.tn;
  const int group = bid / width;
  const int first_pm = group * GROUP_M;
  const int gsz = min(p.tm - first_pm, GROUP_M);
  const int pm = first_pm + (bid % width) % gsz;
  const int pn = (bid % width) / gsz;
  const int64_t m0 = (int64_t)pm * BMR, n0 = (int64_t)pn * 256;
  const bool has_a = AI == 4 || wave < 6;              // 192-row tiles: waves 6,7 carry B pieces only

  const uint32_t bytesA = (uint32_t)(((p.M - 1) * p.lda + p.K) * 2);
  const uint32_t bytesB = (uint32_t)(((p.N - 1) * p.ldb + p.K) * 2);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.A), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, bytesB, 0x00020000);
  // LDS-DMA piece pi (1 KiB) = tile rows 16pi..16pi+15 x 4 chunks; lane -> (row = 16pi + lane/4, slot = lane%4)
  uint32_t offA[2], offB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const int64_t ga = m0 + row, gb = n0 + row;
    offA[j] = ga < p.M ? (uint32_t)((ga * p.lda) * 2 + chunk * 16) : 0x80000000u;
    offB[j] = gb < p.N ? (uint32_t)((gb * p.ldb) * 2 + chunk * 16) : 0x80000000u;
  }
  const int nk_tot = (int)(p.K / 32);
  const int k_lo = (int)((int64_t)split_j * nk_tot / split_s);
  const int nk = (int)((int64_t)(split_j + 1) * nk_tot / split_s) - k_lo;   // slabs of this workgroup
#define DMA_A(slab, j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t*)(smem + ((slab) & 3) * STAGE + (wave * 2 + (j)) * 1024), 16, offA[j], (k_lo + (slab)) * 64, 0, 0)
#define DMA_B(slab, j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t*)(smem + ((slab) & 3) * STAGE + A_ST + (wave * 2 + (j)) * 1024), 16, offB[j], (k_lo + (slab)) * 64, 0, 0)
#define DMA_SLAB(slab) do { if (has_a) { DMA_A(slab, 0); DMA_A(slab, 1); } DMA_B(slab, 0); DMA_B(slab, 1); } while (0)
#define WAIT_PREV_SLAB() do { if (has_a) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } while (0)

  f32x16_t acc[AI][2];
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: slabs 0,1,2 in flight; slabs 0 and 1 complete before the loop
  DMA_SLAB(0);
  if (nk > 1) DMA_SLAB(1);
  if (nk > 2) {
    DMA_SLAB(2);
    WAIT_PREV_SLAB();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // fragment registers: R0 = k-step 0 of the current slab, R1 = k-step 1
  u32x4_t a0[AI], b0[2], a1[AI], b1[2];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int sw = (l32 >> 2) & 3;
  const uint32_t abase0 = lds0 + (wm * (BMR / 2) + l32) * 64 + (((0 + lh) ^ sw) << 4);
  const uint32_t abase1 = lds0 + (wm * (BMR / 2) + l32) * 64 + (((2 + lh) ^ sw) << 4);
  const uint32_t bbase0 = lds0 + A_ST + (wn * 64 + l32) * 64 + (((0 + lh) ^ sw) << 4);
  const uint32_t bbase1 = lds0 + A_ST + (wn * 64 + l32) * 64 + (((2 + lh) ^ sw) << 4);
#if defined(DXA_ABL) && DXA_ABL == 3
#define DS_READ(dst, addr, imm) asm volatile("" : "+v"(dst) : "v"(addr))
#else
#define DS_READ(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#endif
#define RD_SET(A_, B_, abase, bbase, soff)                                                      \
  do {                                                                                          \
    const uint32_t ax_ = (abase) + (soff), bx_ = (bbase) + (soff);                              \
    DS_READ(A_[0], ax_, 0); DS_READ(A_[1], ax_, 2048); DS_READ(A_[2], ax_, 4096);          \
    if constexpr (AI > 3) DS_READ(A_[3], ax_, 6144);                                            \
    DS_READ(B_[0], bx_, 0); DS_READ(B_[1], bx_, 2048);                                          \
  } while (0)
#if defined(DXA_ABL) && DXA_ABL == 4
#define MFMA1(A_, B_, i, j) asm volatile("" : "+v"(acc[i][j]) : "v"(A_[i]), "v"(B_[j]))
#else
#define MFMA1(A_, B_, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, B_[j]), __builtin_bit_cast(bf16x8_t, A_[i]), acc[i][j], 0, 0, 0)
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)

  RD_SET(a0, b0, abase0, bbase0, 0u);       // k-step 0 of slab 0
  for (int t = 0; t < nk; ++t) {
    const uint32_t soff = (uint32_t)((t & 3) * STAGE), soff_n = (uint32_t)(((t + 1) & 3) * STAGE);
#if defined(DXA_ABL) && DXA_ABL == 1
    const bool dma = false;
#else
    const bool dma = t + 3 < nk;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SB();
    MFMA1(a0, b0, 0, 0); MFMA1(a0, b0, 0, 1);
    SB();
    RD_SET(a1, b1, abase1, bbase1, soff);                 // k-step 1 of slab t
    SB();
    MFMA1(a0, b0, 1, 0); MFMA1(a0, b0, 1, 1);
    SB();
    if (dma && has_a) { DMA_A(t + 3, 0); DMA_A(t + 3, 1); }
    SB();
    MFMA1(a0, b0, 2, 0); MFMA1(a0, b0, 2, 1);
    SB();
    if (dma) { DMA_B(t + 3, 0); DMA_B(t + 3, 1); }
    SB();
    if constexpr (AI > 3) { MFMA1(a0, b0, 3, 0); MFMA1(a0, b0, 3, 1); }
    SB();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SB();
    MFMA1(a1, b1, 0, 0); MFMA1(a1, b1, 0, 1);
    SB();
    if (t + 1 < nk) RD_SET(a0, b0, abase0, bbase0, soff_n);   // k-step 0 of slab t+1 (complete since barrier t)
    SB();
    MFMA1(a1, b1, 1, 0); MFMA1(a1, b1, 1, 1);
    MFMA1(a1, b1, 2, 0); MFMA1(a1, b1, 2, 1);
    if constexpr (AI > 3) { MFMA1(a1, b1, 3, 0); MFMA1(a1, b1, 3, 1); }
    SB();
    if (dma) WAIT_PREV_SLAB();                                   // my pieces of slab t+2 have landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(defined(DXA_ABL) && DXA_ABL == 2)
    __builtin_amdgcn_s_barrier();
#endif
    SB();
  }
#undef DMA_A
#undef DMA_B
#undef DMA_SLAB
#undef WAIT_PREV_SLAB
#undef DS_READ
#undef RD_SET
#undef MFMA1
#undef SB

  tile_finish<TO, AI, TE>(p, acc, smem, tid, lane, wave, wm, wn, l32, lh, m0, n0, split_j, split_s, tail_i);
#endif  // __HIP_DEVICE_COMPILE__
}
template __global__ void gemm_nt_ring_kernel<bf16_t, 4, bf16_t>(const GemmP);
template __global__ void gemm_nt_ring_kernel<float, 4, bf16_t>(const GemmP);
template __global__ void gemm_nt_ring_kernel<bf16_t, 3, bf16_t>(const GemmP);
template __global__ void gemm_nt_ring_kernel<float, 3, bf16_t>(const GemmP);
template __global__ void gemm_nt_ring_kernel<float, 4, float>(const GemmP);   // fp32 epilogue operands (bf16x3 products)
template __global__ void gemm_nt_ring_kernel<float, 3, float>(const GemmP);

// =====================================================================================================
// Fast path 2: bf16 NT "ping-pong" kernel — 256x256 tile, K tile 64 (whole 128-byte lines), 8 waves as
// 2 (M) x 4 (N), two LDS buffers of 64 KiB.
//
// What bounded the ring kernel (PMC + ablations): its L2->LDS feed moved 64-byte row segments and
// every wave interleaved its own LDS reads / DMA issue with its own MFMAs, so the two waves of a SIMD competed
// instead of complementing each other (49 % MFMA-busy).  Here the two wave groups (waves 0-3 / 4-7; wave w and
// w+4 share a SIMD) run ONE barrier interval apart and alternate roles:
//     group 0:       | M0 | C0 | M1 | C1 | ...          M = memory cluster: ds_read_b128 fragment loads of the
//     group 1:  | -- | M0 | C0 | M1 | C1 | ...              quadrant computed next + 2 LDS-DMA instructions + counted vmcnt
//                                                        C = compute cluster: 8 x v_mfma_f32_32x32x16_bf16 (256 clk)
// so in every interval each SIMD has one wave feeding the matrix pipe and one wave on the LDS / L2 side
// (measured: the stagger alone is worth 13 %; s_setprio around the clusters nothing).  A K tile is 4 phases = the 4
// quadrants (64 rows x 32 cols) of the wave's 128 x 64 output:
//     phase 0: reads A0,B0 -> Q(A0,B0) | 1: reads B1 -> Q(A0,B1) | 2: reads A1 -> Q(A1,B1) | 3: reads B0 -> Q(A1,B0)
// (A0/A1 = 64-row halves of the wave's rows, B0/B1 = 32-column halves of its columns): 24 ds_read_b128 per wave per
// K tile for 32 MFMAs.  The next K tile arrives as 4 DMA "pieces" of 16 KiB, one per phase (each piece = that half for
// ALL waves: 2 x 1 KiB instructions per wave).
// LDS image: rows of 128 bytes (64 k), 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7): the 16-lane
// groups of ds_read_b128 over a 32-row fragment column hit 16 distinct 16-byte slots of the 256-byte bank row.  The DMA
// writes LDS lane-linearly (8 rows x 128 B per instruction), so the swizzle is applied to the per-lane SOURCE chunk;
// each row is still fetched as one whole 128-byte line.
// Ablations on 8192^3 (scripts/ablate_gemm.sh): everything 832 us; without LDS-DMA 574 (1.92 PF/s); without MFMA 607;
// without ds_read 628; 2 phases per K tile instead of 4 (half the barriers) 819 — the L2->LDS feed (~27 B/clk/CU
// achieved, 32 needed at the MFMA peak for a 256x256 tile) and its interference with the fragment reads are what is
// left, not the barriers.
// Requirements: K % 64 == 0, 16-byte aligned A/B rows, no batching, operands < 2 GiB.
// =====================================================================================================

// LEAN = the epilogue is C = alpha * acc + bias (+ residual) (+ C) over whole 16-byte accesses (sk_epilogue); otherwise
// the generic epilogue of tile_finish (activation / mulgrad / aux / ragged edges).
// A_KS / B_KS = that operand is K-STRIDED: element (row r of the tile, contraction index k) lives at base[k * ld + r] —
// B of the NN product dX = dY W (W is [out, in] row-major, the contraction runs over `out`), A and B of the TN product
// dW = dY^T X (both activations are [tokens, features]).  Such an operand is staged as it lies in memory, one piece =
// [64 k][128 tile rows of that half] with 256-byte LDS rows (4 k-rows per DMA instruction, every 128 / 64 contiguous
// source bytes a whole / half line), and the MFMA fragment (8 consecutive k of ONE tile row per lane) is gathered with
// two ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 rows] block and lane i receives the 4 k of row i.  The four
// k-rows of such a block would share banks (256-byte pitch): 64-byte chunk q of k-row t sits at chunk q ^ (t & 3).
// No transposed copy of anything is ever made: the W^T shadow arena and the per-GEMM activation transposes are gone.
// K need not be a multiple of 64 when every operand is k-strided (rows past K are out of the buffer's range: zeros).
// FUSE = 1 (DXA_FUSE_SWIGLU, bf16 NT lean only): B = [gate ; up] of a gated MLP; the tile's 256 B rows are 128 gate rows and the 128
// up rows of the same outputs — per wave column wn: 32 gate rows (its block j = 0) and 32 up rows (j = 1) — picked by the DMA source
// addresses, and the epilogue stores silu(gate) * up (sk_epilogue_swiglu).  Main loop, tiles, K order: unchanged.
template <typename TO, typename TE, bool LEAN, bool A_KS, bool B_KS, int FUSE = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int REG = 256 * 128;            // one operand of one K tile: 256 rows x 128 B
  constexpr int BUF = 2 * REG;              // A | B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l32 = lane & 31, lh = lane >> 5;

  // tiles [0, full): one workgroup each, block ids dealt round-robin to the 8 XCDs are remapped so every XCD
  // walks a contiguous run of tiles.  Tail tiles (the partial last round of the 256 CUs) are cut along K into
  // split_s workgroups each; the last split gathers the others' fp32 partials and runs the epilogue.
  int bid = blockIdx.x;
  int split_j = 0, split_s = 1, tail_i = 0;
  if (bid < p.full) {
    const int q = p.full >> 3, r = p.full & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  } else {
    const int idx = bid - p.full;
    tail_i = idx % p.tail_r;
    split_j = idx / p.tail_r;
    split_s = p.split_s;
    bid = p.full + tail_i;
  }
  int m0i, n0i;
  sk_tile_origin(p, bid, m0i, n0i);
  const int64_t m0 = m0i, n0 = n0i;

  const uint32_t bytesA = (uint32_t)((A_KS ? (p.K - 1) * p.lda + p.M : (p.M - 1) * p.lda + p.K) * 2);
  const uint32_t bytesB = (uint32_t)((B_KS ? (p.K - 1) * p.ldb + p.N : (p.N - 1) * p.ldb + p.K) * 2);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.A), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, bytesB, 0x00020000);
  // DMA instruction (h, j) of a wave covers 8 consecutive tile rows; g = 2*wave + j in 0..15 selects the 8-row group:
  //   A half h: rows (g & 7) * 8 + (g >> 3) * 128 + 64 h   (the A-h rows of both wave rows wm)
  //   B half h: rows (g >> 2) * 64 + (g & 3) * 8 + 32 h     (the B-h columns of all four wave columns wn)
  // lane -> (row = row0 + lane / 8, LDS chunk slot = lane % 8, source chunk = slot ^ ((row >> 1) & 7))
#define PP_A_ROW0(h, j) ((((wave * 2 + (j)) & 7) * 8) + (((wave * 2 + (j)) >> 3) * 128) + 64 * (h))
#define PP_B_ROW0(h, j) ((((wave * 2 + (j)) >> 2) * 64) + (((wave * 2 + (j)) & 3) * 8) + 32 * (h))
  // k-strided operand: DMA instruction (h, j) of a wave covers k-rows 4 (2 wave + j) .. +3 of piece h; lane -> (k-row =
  // + lane / 16, LDS slot s = lane % 16 of the 256-byte row, source slot = s ^ ((k-row & 3) << 2)); source slot sigma ->
  // tile row: A: (sigma >> 3) * 128 + 64 h + 8 (sigma & 7) (two 128-byte segments: the A-h rows of wm = 0, 1)
  //           B: (sigma >> 2) * 64 + 32 h + 8 (sigma & 3)  (four 64-byte segments: the B-h columns of wn = 0..3)
  uint32_t voA[2][2], voB[2][2];
  {
    const int lrow = lane >> 3, lslot = lane & 7;
    const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int krow = (wave * 2 + j) * 4 + (lane >> 4);
        const int sig = (lane & 15) ^ ((krow & 3) << 2);
        if constexpr (A_KS) {
          const int ga = m0i + (sig >> 3) * 128 + 64 * h + 8 * (sig & 7);
          voA[h][j] = ga < (int)p.M ? (uint32_t)krow * lda2 + (uint32_t)ga * 2u : 0x80000000u;
        } else {
          const int ra = PP_A_ROW0(h, j) + lrow, ga = m0i + ra;
          voA[h][j] = ga < (int)p.M ? (uint32_t)ga * lda2 + (uint32_t)((lslot ^ ((ra >> 1) & 7)) << 4) : 0x80000000u;
        }
        if constexpr (B_KS) {
          const int gb = n0i + (sig >> 2) * 64 + 32 * h + 8 * (sig & 3);
          voB[h][j] = gb < (int)p.N ? (uint32_t)krow * ldb2 + (uint32_t)gb * 2u : 0x80000000u;
        } else if constexpr (FUSE == 1) {
          const int rb = PP_B_ROW0(h, j) + lrow, half_n = (int)(p.N >> 1);
          const int lc = (n0i >> 1) + (rb >> 6) * 32 + (rb & 31), gb = ((rb >> 5) & 1) * half_n + lc;
          voB[h][j] = lc < half_n ? (uint32_t)gb * ldb2 + (uint32_t)((lslot ^ ((rb >> 1) & 7)) << 4) : 0x80000000u;
        } else {
          const int rb = PP_B_ROW0(h, j) + lrow, gb = n0i + rb;
          voB[h][j] = gb < (int)p.N ? (uint32_t)gb * ldb2 + (uint32_t)((lslot ^ ((rb >> 1) & 7)) << 4) : 0x80000000u;
        }
      }
  }
  // TN with a second segment (p.K2 > 0): K tiles [0, nk1) come from (A, B), tiles [nk1, nk1 + nk2) from (A2, B2); the last
  // tile of segment 1 is zero-filled past K by its descriptor's range check, like the last tile of any TN product
  const int nk1 = (int)((p.K + 63) >> 6);
  const int nk_tot = nk1 + ((A_KS && B_KS) ? (int)((p.K2 + 63) >> 6) : 0);
  const bool seg2 = (A_KS && B_KS) && p.K2 > 0;
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(seg2 ? p.A2 : p.A), 0, seg2 ? (uint32_t)(((p.K2 - 1) * p.lda + p.M) * 2) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(seg2 ? p.B2 : p.B), 0, seg2 ? (uint32_t)(((p.K2 - 1) * p.ldb + p.N) * 2) : 0u, 0x00020000);
  const int k_lo = split_j * nk_tot / split_s;
  const int nk = (split_j + 1) * nk_tot / split_s - k_lo;   // K tiles of this workgroup (>= 1)
  // K-contiguous operand: the K tile is a byte offset inside the row (scalar offset of the instruction).  K-strided: it is
  // 64 rows further down; that goes into the VECTOR offset so that the descriptor's range check sees it (rows >= K: zeros)
  const uint32_t ktileA = A_KS ? 64u * (uint32_t)p.lda * 2u : 0u, ktileB = B_KS ? 64u * (uint32_t)p.ldb * 2u : 0u;
#define PP_DMA(rsrc, vo, ldsoff, buf, soff)                                                                             \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(smem + (buf) * BUF + (ldsoff)), 16, vo, soff, 0, 0)
#if (DXA_PPV & 1)
#define PP_LOOP(x) do { } while (0)
#else
#define PP_LOOP(x) x
#endif
#define PP_DMA_A1(h, j, buf, tile)                                                                          \
  do {                                                                                                      \
    if constexpr (A_KS && B_KS) {                                                                           \
      const int kt_ = k_lo + (tile);                                                                        \
      if (kt_ < nk1) PP_DMA(rA, voA[h][j] + (uint32_t)kt_ * ktileA, (h) * 16384 + (wave * 2 + (j)) * 1024, buf, 0);          \
      else PP_DMA(rA2, voA[h][j] + (uint32_t)(kt_ - nk1) * ktileA, (h) * 16384 + (wave * 2 + (j)) * 1024, buf, 0);           \
    } else if constexpr (A_KS) {                                                                            \
      const uint32_t kb_ = (uint32_t)(k_lo + (tile)) * ktileA;                                              \
      PP_DMA(rA, voA[h][j] + kb_, (h) * 16384 + (wave * 2 + (j)) * 1024, buf, 0);                           \
    } else {                                                                                                \
      PP_DMA(rA, voA[h][j], PP_A_ROW0(h, j) * 128, buf, (k_lo + (tile)) * 128);                             \
    }                                                                                                       \
  } while (0)
#define PP_DMA_B1(h, j, buf, tile)                                                                          \
  do {                                                                                                      \
    if constexpr (A_KS && B_KS) {                                                                           \
      const int kt_ = k_lo + (tile);                                                                        \
      if (kt_ < nk1) PP_DMA(rB, voB[h][j] + (uint32_t)kt_ * ktileB, REG + (h) * 16384 + (wave * 2 + (j)) * 1024, buf, 0);    \
      else PP_DMA(rB2, voB[h][j] + (uint32_t)(kt_ - nk1) * ktileB, REG + (h) * 16384 + (wave * 2 + (j)) * 1024, buf, 0);     \
    } else if constexpr (B_KS) {                                                                            \
      const uint32_t kb_ = (uint32_t)(k_lo + (tile)) * ktileB;                                              \
      PP_DMA(rB, voB[h][j] + kb_, REG + (h) * 16384 + (wave * 2 + (j)) * 1024, buf, 0);                     \
    } else {                                                                                                \
      PP_DMA(rB, voB[h][j], REG + PP_B_ROW0(h, j) * 128, buf, (k_lo + (tile)) * 128);                       \
    }                                                                                                       \
  } while (0)
#define PP_DMA_A(h, buf, tile) do { PP_DMA_A1(h, 0, buf, tile); PP_DMA_A1(h, 1, buf, tile); } while (0)
#define PP_DMA_B(h, buf, tile) do { PP_DMA_B1(h, 0, buf, tile); PP_DMA_B1(h, 1, buf, tile); } while (0)

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addresses: A block i (32 rows) of k-step ks: row = wm*128 + 32 i + l32, chunk c = 2 ks + lh at slot
  // c ^ sw.  2 ks + lh = (ks << 1) ^ lh, the row base has zero bits below 128 and buffer 1 starts at bit 16, so
  // address = ya ^ (ks << 5) ^ (cur << 16) with ONE per-lane register ya = row base | ((lh ^ sw) << 4) per operand.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int sw = (l32 >> 1) & 7;
  // k-strided operand: lane (i = lane % 16, half-group (lane >> 4) & 1, lh) passes the address of 4 consecutive tile rows
  // (8 bytes) of k-row t = 8 lh + (i >> 2) [+ 16 ks + 4 r as immediate] of its [4 k][16 rows] block and receives row i's
  // 4 k; the wave's 64-byte chunk of the 256-byte piece row is q = 2 wm + ii (A) / wn (B), stored at q ^ (t & 3).
  const int i16 = lane & 15, tq = (i16 >> 2) & 3, tk = 8 * lh + (i16 >> 2);
  const uint32_t ks_lane = (uint32_t)(tk * 256 + ((lane >> 4) & 1) * 32 + (i16 & 3) * 8);
  const uint32_t ya = A_KS ? lds0 + ks_lane + (uint32_t)(((wm * 2) ^ tq) << 6)
                           : (lds0 + (wm * 128 + l32) * 128) | (uint32_t)((lh ^ sw) << 4);
  const uint32_t yb = B_KS ? lds0 + REG + ks_lane + (uint32_t)((wn ^ tq) << 6)
                           : (lds0 + REG + (wn * 64 + l32) * 128) | (uint32_t)((lh ^ sw) << 4);
  u32x4_t af[2][4], bq[2][4];            // B fragments of both halves stay in registers: B0 serves phases 0 and 3
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#if (DXA_PPV & 2)
#define PP_READ(dst, addr, imm) asm volatile("" : "+v"(dst) : "v"(addr))
#define PP_READ_TR(dst, addr, imm) asm volatile("" : "+v"(dst) : "v"(addr))
#else
#define PP_READ(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define PP_READ_TR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#endif
  // one fragment (8 k of one tile row per lane) of a k-strided operand: k 0..3 | k 4..7
#define PP_FRAG_TR(dst, addr, imm)                                                       \
  do {                                                                                   \
    u32x2_t lo_ = {0u, 0u}, hi_ = {0u, 0u};                                              \
    PP_READ_TR(lo_, addr, imm); PP_READ_TR(hi_, addr, (imm) + 1024);                     \
    dst = (u32x4_t){lo_[0], lo_[1], hi_[0], hi_[1]};                                     \
  } while (0)
#define PP_RD_A(cur, h)                                                                                     \
  do {                                                                                                      \
    if constexpr (A_KS) {                                                                                   \
      const uint32_t a0_ = ya ^ (uint32_t)((cur) * BUF + (h) * 16384), a1_ = a0_ ^ 64u;                     \
      PP_FRAG_TR(af[0][0], a0_, 0); PP_FRAG_TR(af[1][0], a1_, 0);                                           \
      PP_FRAG_TR(af[0][1], a0_, 4096); PP_FRAG_TR(af[1][1], a1_, 4096);                                     \
      PP_FRAG_TR(af[0][2], a0_, 8192); PP_FRAG_TR(af[1][2], a1_, 8192);                                     \
      PP_FRAG_TR(af[0][3], a0_, 12288); PP_FRAG_TR(af[1][3], a1_, 12288);                                   \
    } else {                                                                                                \
      const uint32_t a0_ = ya ^ (uint32_t)((cur) * BUF), a1_ = ya ^ (uint32_t)((cur) * BUF + 32),           \
                     a2_ = ya ^ (uint32_t)((cur) * BUF + 64), a3_ = ya ^ (uint32_t)((cur) * BUF + 96);      \
      PP_READ(af[0][0], a0_, (2 * (h)) * 4096); PP_READ(af[1][0], a0_, (2 * (h) + 1) * 4096);              \
      PP_READ(af[0][1], a1_, (2 * (h)) * 4096); PP_READ(af[1][1], a1_, (2 * (h) + 1) * 4096);              \
      PP_READ(af[0][2], a2_, (2 * (h)) * 4096); PP_READ(af[1][2], a2_, (2 * (h) + 1) * 4096);              \
      PP_READ(af[0][3], a3_, (2 * (h)) * 4096); PP_READ(af[1][3], a3_, (2 * (h) + 1) * 4096);              \
    }                                                                                                       \
  } while (0)
#define PP_RD_B(cur, j)                                                                                     \
  do {                                                                                                      \
    if constexpr (B_KS) {                                                                                   \
      const uint32_t b0_ = yb ^ (uint32_t)((cur) * BUF + (j) * 16384);                                      \
      PP_FRAG_TR(bq[j][0], b0_, 0); PP_FRAG_TR(bq[j][1], b0_, 4096);                                        \
      PP_FRAG_TR(bq[j][2], b0_, 8192); PP_FRAG_TR(bq[j][3], b0_, 12288);                                    \
    } else {                                                                                                \
      const uint32_t b0_ = yb ^ (uint32_t)((cur) * BUF), b1_ = yb ^ (uint32_t)((cur) * BUF + 32),           \
                     b2_ = yb ^ (uint32_t)((cur) * BUF + 64), b3_ = yb ^ (uint32_t)((cur) * BUF + 96);      \
      PP_READ(bq[j][0], b0_, (j) * 4096); PP_READ(bq[j][1], b1_, (j) * 4096);                               \
      PP_READ(bq[j][2], b2_, (j) * 4096); PP_READ(bq[j][3], b3_, (j) * 4096);                               \
    }                                                                                                       \
  } while (0)
#if (DXA_PPV & 4)
#define PP_MFMA2(ii, ks, i, j, jr) asm volatile("" : "+v"(acc[i][j]) : "v"(bq[jr][ks]), "v"(af[ii][ks]))
#else
#define PP_MFMA2(ii, ks, i, j, jr) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bq[jr][ks]), __builtin_bit_cast(bf16x8_t, af[ii][ks]), acc[i][j], 0, 0, 0)
#endif
#define PP_MFMA(ii, ks, i, j) PP_MFMA2(ii, ks, i, j, j)
#define PP_SB() __builtin_amdgcn_sched_barrier(0)
#define PP_BAR() do { PP_SB(); __builtin_amdgcn_s_barrier(); PP_SB(); } while (0)
  // compute cluster of quadrant (A half h, B half j): the two accumulators alternate so dependent MFMAs are 2 apart
#define PP_COMPUTE(h, j)                                                                          \
  do {                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    PP_SB();                                                                                      \
    PP_MFMA(0, 0, 2 * (h), j); PP_MFMA(1, 0, 2 * (h) + 1, j);                                     \
    PP_MFMA(0, 1, 2 * (h), j); PP_MFMA(1, 1, 2 * (h) + 1, j);                                     \
    PP_MFMA(0, 2, 2 * (h), j); PP_MFMA(1, 2, 2 * (h) + 1, j);                                     \
    PP_MFMA(0, 3, 2 * (h), j); PP_MFMA(1, 3, 2 * (h) + 1, j);                                     \
    PP_SB();                                                                                      \
  } while (0)
  // k-step ks of A half h / B half j alone (round 4: the fragment reads ride in the gaps of the PREVIOUS compute cluster's MFMAs)
#define PP_RDK_A(cur, h, ks)                                                                                \
  do {                                                                                                      \
    if constexpr (A_KS) {                                                                                   \
      const uint32_t a0_ = ya ^ (uint32_t)((cur) * BUF + (h) * 16384), a1_ = a0_ ^ 64u;                     \
      PP_FRAG_TR(af[0][ks], a0_, 4096 * (ks)); PP_FRAG_TR(af[1][ks], a1_, 4096 * (ks));                     \
    } else {                                                                                                \
      const uint32_t a_ = ya ^ (uint32_t)((cur) * BUF + 32 * (ks));                                         \
      PP_READ(af[0][ks], a_, (2 * (h)) * 4096); PP_READ(af[1][ks], a_, (2 * (h) + 1) * 4096);              \
    }                                                                                                       \
  } while (0)
#define PP_RDK_B(cur, j, ks)                                                                                \
  do {                                                                                                      \
    if constexpr (B_KS) {                                                                                   \
      const uint32_t b0_ = yb ^ (uint32_t)((cur) * BUF + (j) * 16384);                                      \
      PP_FRAG_TR(bq[j][ks], b0_, 4096 * (ks));                                                              \
    } else {                                                                                                \
      const uint32_t b_ = yb ^ (uint32_t)((cur) * BUF + 32 * (ks));                                         \
      PP_READ(bq[j][ks], b_, (j) * 4096);                                                                   \
    }                                                                                                       \
  } while (0)
  // the same with the register set named apart from the B half (DXA_PPR=2: the two sets swap roles every K tile)
#define PP_RDK_B2(cur, j, jr, ks)                                                                           \
  do {                                                                                                      \
    if constexpr (B_KS) {                                                                                   \
      const uint32_t b0_ = yb ^ (uint32_t)((cur) * BUF + (j) * 16384);                                      \
      PP_FRAG_TR(bq[jr][ks], b0_, 4096 * (ks));                                                             \
    } else {                                                                                                \
      const uint32_t b_ = yb ^ (uint32_t)((cur) * BUF + 32 * (ks));                                         \
      PP_READ(bq[jr][ks], b_, (j) * 4096);                                                                  \
    }                                                                                                       \
  } while (0)
#define PP_COMPUTE_R2(h, j, jr, R)                                                                \
  do {                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    PP_SB();                                                                                      \
    PP_MFMA2(0, 0, 2 * (h), j, jr); PP_MFMA2(1, 0, 2 * (h) + 1, j, jr); PP_SB(); R(0); PP_SB();   \
    PP_MFMA2(0, 1, 2 * (h), j, jr); PP_MFMA2(1, 1, 2 * (h) + 1, j, jr); PP_SB(); R(1); PP_SB();   \
    PP_MFMA2(0, 2, 2 * (h), j, jr); PP_MFMA2(1, 2, 2 * (h) + 1, j, jr); PP_SB(); R(2); PP_SB();   \
    PP_MFMA2(0, 3, 2 * (h), j, jr); PP_MFMA2(1, 3, 2 * (h) + 1, j, jr); PP_SB(); R(3); PP_SB();   \
  } while (0)
  // compute cluster with the NEXT cluster's fragment reads in its gaps: after the two MFMAs of k-step ks their operand registers are
  // dead, R(ks) refills them (the data lands tens of cycles after the MFMAs have read their sources)
#define PP_COMPUTE_R(h, j, R)                                                                     \
  do {                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    PP_SB();                                                                                      \
    PP_MFMA(0, 0, 2 * (h), j); PP_MFMA(1, 0, 2 * (h) + 1, j); PP_SB(); R(0); PP_SB();             \
    PP_MFMA(0, 1, 2 * (h), j); PP_MFMA(1, 1, 2 * (h) + 1, j); PP_SB(); R(1); PP_SB();             \
    PP_MFMA(0, 2, 2 * (h), j); PP_MFMA(1, 2, 2 * (h) + 1, j); PP_SB(); R(2); PP_SB();             \
    PP_MFMA(0, 3, 2 * (h), j); PP_MFMA(1, 3, 2 * (h) + 1, j); PP_SB(); R(3); PP_SB();             \
  } while (0)
#define PP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  // one K tile in buffer `cur = t & 1`.  LDS-DMA runs SIX pieces (96 KiB) ahead of the reads, in the two 64 KiB buffers
  // alone: the B0 fragments stay in registers from phase 0 to phase 3, so every piece is read from LDS in exactly one
  // memory cluster (A0, B0: M0; B1: M1; A1: M2; M3 reads nothing) and its bytes are free again two phases later.  The global
  // issue order is A0(0) B0(0) B1(0) A1(0) A0(1) B0(1) | B1(t+1) A1(t+1) A0(t+2) B0(t+2) in phases 0..3 of tile t: each
  // piece is issued 5-6 phases (~1.5 K tiles, > 1.5 us) before its read instead of 2-3 — the loaded L2 -> LDS latency no
  // longer stalls the MFMA clusters.  A piece read in M_q is waited for at the END of M_{q-1} by every wave (vmcnt(8): the
  // four younger pieces stay in flight); both groups' M_{q-1} end before the interval in which the first M_q starts, so
  // wait + barrier order the DMA before every read.  Reads of M_q have completed for BOTH staggered groups two phases on
  // (group 1's reads of M_q retire in barrier interval 2q+2, group 0 issues M_{q+2} in interval 2q+4): a piece may be
  // overwritten from phase q+2 onward — A0(t+2) in phase 2, B0(t+2) in phase 3 of tile t (read in M0 of tile t), B1(t+1) /
  // A1(t+1) in phases 0 / 1 of tile t (read in M1 / M2 of tile t-1).
#if (DXA_PPV & 64)   /* tuning variant: LDS-DMA issue ahead of the fragment reads of the same memory cluster */
#define PP_M(reads, dma) do { dma; PP_SB(); reads; PP_SB(); } while (0)
#else
#define PP_M(reads, dma) do { reads; PP_SB(); dma; PP_SB(); } while (0)
#endif
  // The eight LDS-DMA instructions a wave issues per K tile, in their (fixed) order: slots 0-1 = B1(t+1), 2-3 = A1(t+1) into
  // the other buffer, 4-5 = A0(t+2), 6-7 = B0(t+2) into this one.  They are dealt to the four memory clusters as PPD0..PPD3
  // instructions: M0 already carries 12 fragment reads, M1 4, M2 8 and M3 none, and a memory cluster longer than the other
  // group's compute cluster (8 MFMAs = 256 cycles) stalls the MFMA pipe — so the DMA issue goes where the reads are few
  // (default 1, 2, 2, 3; DXA_PPD=2222 is the even deal).  Slots 4-7 overwrite bytes read in M0 of this tile: not before M2.
#ifndef DXA_PPD
#define DXA_PPD 1223
#endif
  constexpr int PPD0 = DXA_PPD / 1000, PPD1 = DXA_PPD / 100 % 10, PPD2 = DXA_PPD / 10 % 10, PPD3 = DXA_PPD % 10;
  static_assert(PPD0 + PPD1 + PPD2 + PPD3 == 8 && PPD0 + PPD1 <= 4, "eight LDS-DMA instructions per K tile; slots 4-7 from phase 2 on");
#define PP_SLOT(k, cur, t)                                                                                  \
  do {                                                                                                      \
    if ((k) < 4 ? more1 : more2) {                                                                          \
      if ((k) == 0) PP_LOOP(PP_DMA_B1(1, 0, (cur) ^ 1, (t) + 1));                                           \
      else if ((k) == 1) PP_LOOP(PP_DMA_B1(1, 1, (cur) ^ 1, (t) + 1));                                      \
      else if ((k) == 2) PP_LOOP(PP_DMA_A1(1, 0, (cur) ^ 1, (t) + 1));                                      \
      else if ((k) == 3) PP_LOOP(PP_DMA_A1(1, 1, (cur) ^ 1, (t) + 1));                                      \
      else if ((k) == 4) PP_LOOP(PP_DMA_A1(0, 0, cur, (t) + 2));                                            \
      else if ((k) == 5) PP_LOOP(PP_DMA_A1(0, 1, cur, (t) + 2));                                            \
      else if ((k) == 6) PP_LOOP(PP_DMA_B1(0, 0, cur, (t) + 2));                                            \
      else PP_LOOP(PP_DMA_B1(0, 1, cur, (t) + 2));                                                          \
    }                                                                                                       \
  } while (0)
#define PP_SLOTS(from, to, cur, t) do { _Pragma("unroll") for (int k_ = (from); k_ < (to); ++k_) PP_SLOT(k_, cur, t); } while (0)
#define PP_VMCNT_N(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  // tuning variant DXA_PPV & 128: s_memtime after every barrier of a K tile (8 stamps, read once per tile where no LDS read is
  // outstanding); the eight barrier-to-barrier intervals are summed per wave and stored over the first bytes of C
#if (DXA_PPV & 128)
  uint64_t ts_[8];
  uint32_t iv_[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, ts_last_ = 0u;
#define PP_STAMP(i) asm volatile("s_memtime %0" : "=s"(ts_[i]))
#define PP_STAMPS_FOLD()                                                                                    \
  do {                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    PP_SB();                                                                                                \
    if (ts_last_ != 0u) iv_[7] += (uint32_t)ts_[0] - ts_last_;                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 7; ++i_) iv_[i_] += (uint32_t)ts_[i_ + 1] - (uint32_t)ts_[i_];  \
    ts_last_ = (uint32_t)ts_[7];                                                                            \
    PP_SB();                                                                                                \
  } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#define PP_STAMPS_FOLD() do { } while (0)
#endif
#ifndef DXA_PPR
#define DXA_PPR 1
#endif
#if DXA_PPR == 2
  // Round 4 schedule, second form (DXA_PPR=2, default).  DXA_PPR=1 below moved every fragment read into the gaps of the compute cluster
  // BEFORE the one that needs it; its stamps (profiles/r04_pp_stamps_after.txt) then showed the barrier intervals that hold a C3 —
  // 12 reads in its gaps (24 ds_read_b64_tr_b16 for TN): A0(t+1) and B0(t+1) — at 360 / 410 cycles (TN 427 / 500) against 270-340 for
  // the others, while C2 carried none: A1 and B0 are both still live in C2, so nothing could be refilled there.  Here the two B
  // register sets swap roles every K tile (the loop is unrolled by two, so the set index is a compile-time constant): in tile t set
  // s0 = t & 1 holds B0(t) and set s1 = s0 ^ 1 receives B1(t) in C0; C2 (A1 x B1) consumes s1 k-step by k-step and refills it with
  // B0(t+1) — which is then already where tile t+1 (s0' = s1) expects it — and C3 (A1 x B0, set s0) refills only the A registers
  // with A0(t+1).  Reads per compute cluster 4 / 8 / 4 / 8 instead of 4 / 8 / 0 / 12.
  //   C0 (A0 x B0) reads B1(t);  C1 (A0 x B1) reads A1(t);  C2 (A1 x B1) reads B0(t+1);  C3 (A1 x B0) reads A0(t+1).
  // B0(t+1) is needed one phase earlier than before, so the LDS-DMA order swaps A0 and B0:  M0 A1(t+1) -> other buffer, M1 B0(t+2),
  // M2 A0(t+2), M3 B1(t+2) -> this buffer (the bytes they overwrite were read in C1(t-1), C2(t-1), C3(t-1), C0(t): at least one full
  // barrier interval before, for both groups).  In-order issue per wave (2 instructions each):
  //   ... A1(t) | B0(t+1) | A0(t+1) | B1(t+1) | A1(t+1) | B0(t+2) | A0(t+2) | B1(t+2) ...
  // A piece read in C_q is waited for at the end of M_{q-1} by every wave: always the piece issued four phases earlier = vmcnt(8).
#define PP_R2_B1(ks) PP_RDK_B2(cur_, 1, cur_ ^ 1, ks)
#define PP_R2_A1(ks) PP_RDK_A(cur_, 1, ks)
#define PP_R2_B0N(ks) do { if (more1) { PP_RDK_B2(cur_ ^ 1, 0, cur_ ^ 1, ks); } } while (0)
#define PP_R2_A0N(ks) do { if (more1) { PP_RDK_A(cur_ ^ 1, 0, ks); } } while (0)
#define PP_TILE(cur, t)                                                                                     \
  do {                                                                                                      \
    constexpr int cur_ = (cur);                                                                             \
    const bool more1 = (t) + 1 < nk, more2 = (t) + 2 < nk;                                                  \
    /* phase 0 */                                                                                           \
    if (more1) { PP_LOOP(PP_DMA_A(1, cur_ ^ 1, (t) + 1)); PP_SB(); PP_VMCNT(8); } else { PP_VMCNT(0); }     \
    PP_BAR(); PP_STAMP(0); PP_COMPUTE_R2(0, 0, cur_, PP_R2_B1); PP_BAR(); PP_STAMP(1);                      \
    /* phase 1 */                                                                                           \
    if (more2) { PP_LOOP(PP_DMA_B(0, cur_, (t) + 2)); PP_SB(); PP_VMCNT(8); } else if (more1) { PP_VMCNT(6); } \
    PP_BAR(); PP_STAMP(2); PP_COMPUTE_R2(0, 1, cur_ ^ 1, PP_R2_A1); PP_BAR(); PP_STAMP(3);                  \
    /* phase 2 */                                                                                           \
    if (more2) { PP_LOOP(PP_DMA_A(0, cur_, (t) + 2)); PP_SB(); PP_VMCNT(8); } else if (more1) { PP_VMCNT(4); } \
    PP_BAR(); PP_STAMP(4); PP_COMPUTE_R2(1, 1, cur_ ^ 1, PP_R2_B0N); PP_BAR(); PP_STAMP(5);                 \
    /* phase 3 */                                                                                           \
    if (more2) { PP_LOOP(PP_DMA_B(1, cur_, (t) + 2)); PP_SB(); PP_VMCNT(8); } else if (more1) { PP_VMCNT(2); } \
    PP_BAR(); PP_STAMP(6); PP_COMPUTE_R2(1, 0, cur_, PP_R2_A0N); PP_BAR(); PP_STAMP(7);                     \
    PP_STAMPS_FOLD();                                                                                       \
  } while (0)
#elif DXA_PPR
  // Round 4 schedule (DXA_PPR=1, default): NO fragment read is left in a memory cluster.  The stamp build (DXA_PPV=128,
  // profiles/r04_pp_stamps_before.txt) showed the two barrier intervals in which a group runs M0 — 12 ds_read_b128 (24
  // ds_read_b64_tr_b16 when both operands are k-strided) — at 440 cycles (TN 540) against 285 for the other six: the compute
  // cluster of the other group finishes its 256 cycles of MFMAs and waits.  Now every fragment is read in the gaps of the compute
  // cluster BEFORE the one that needs it, into the operand registers its MFMAs have just consumed:
  //   C0 (A0 x B0) reads B1(t);  C1 (A0 x B1) reads A1(t);  C2 (A1 x B1) reads nothing;  C3 (A1 x B0) reads A0(t+1), B0(t+1).
  // A piece read in C_q by one group is read a barrier interval earlier than the other group's M_q ends, so it is waited for at the
  // end of M_{q-1} by every wave (vmcnt + the barrier order it before every read), and the LDS-DMA issue moves a phase earlier with
  // it: M0 A1(t+1) -> other buffer, M1 A0(t+2), M2 B0(t+2), M3 B1(t+2) -> this buffer (the bytes they overwrite were read in
  // C1(t-1), C3(t-1), C3(t-1), C0(t): at least one full barrier interval before, for both groups).  In-order issue per wave:
  //   ... A1(t) | A0(t+1) | B0(t+1) | B1(t+1) | A1(t+1) | A0(t+2) | B0(t+2) | B1(t+2) ...   (2 instructions each)
  // end of M0: A1(t) landed = vmcnt(8);  end of M2: A0(t+1), B0(t+1) = vmcnt(8);  end of M3: B1(t+1) = vmcnt(8)  (4 phases ahead).
#define PP_R_B1(ks) PP_RDK_B(cur_, 1, ks)
#define PP_R_A1(ks) PP_RDK_A(cur_, 1, ks)
#define PP_R_NEXT(ks) do { if (more1) { PP_RDK_A(cur_ ^ 1, 0, ks); PP_RDK_B(cur_ ^ 1, 0, ks); } } while (0)
#define PP_TILE(cur, t)                                                                                     \
  do {                                                                                                      \
    constexpr int cur_ = (cur);                                                                             \
    const bool more1 = (t) + 1 < nk, more2 = (t) + 2 < nk;                                                  \
    /* phase 0 */                                                                                           \
    if (more1) { PP_LOOP(PP_DMA_A(1, cur_ ^ 1, (t) + 1)); PP_SB(); PP_VMCNT(8); } else { PP_VMCNT(0); }     \
    PP_BAR(); PP_STAMP(0); PP_COMPUTE_R(0, 0, PP_R_B1); PP_BAR(); PP_STAMP(1);                              \
    /* phase 1 */                                                                                           \
    if (more2) { PP_LOOP(PP_DMA_A(0, cur_, (t) + 2)); PP_SB(); }                                            \
    PP_BAR(); PP_STAMP(2); PP_COMPUTE_R(0, 1, PP_R_A1); PP_BAR(); PP_STAMP(3);                              \
    /* phase 2 */                                                                                           \
    if (more2) { PP_LOOP(PP_DMA_B(0, cur_, (t) + 2)); PP_SB(); PP_VMCNT(8); } else if (more1) { PP_VMCNT(4); } \
    PP_BAR(); PP_STAMP(4); PP_COMPUTE(1, 1); PP_BAR(); PP_STAMP(5);                                         \
    /* phase 3 */                                                                                           \
    if (more2) { PP_LOOP(PP_DMA_B(1, cur_, (t) + 2)); PP_SB(); PP_VMCNT(8); } else if (more1) { PP_VMCNT(2); } \
    PP_BAR(); PP_STAMP(6); PP_COMPUTE_R(1, 0, PP_R_NEXT); PP_BAR(); PP_STAMP(7);                            \
    PP_STAMPS_FOLD();                                                                                       \
  } while (0)
#else
#define PP_TILE(cur, t)                                                                                     \
  do {                                                                                                      \
    const bool more1 = (t) + 1 < nk, more2 = (t) + 2 < nk;                                                  \
    /* phase 0 */                                                                                           \
    PP_M(PP_RD_A(cur, 0); PP_RD_B(cur, 0), PP_SLOTS(0, PPD0, cur, t));                                      \
    /* B1(t) landed: younger = A1(t) (2) + A0, B0 of t+1 (4) + this tile's slots so far */                  \
    if (more1) { PP_VMCNT_N(6 + PPD0); } else { PP_VMCNT(2); }                                              \
    PP_BAR(); PP_STAMP(0); PP_COMPUTE(0, 0); PP_BAR(); PP_STAMP(1);                                         \
    /* phase 1 */                                                                                           \
    PP_M(PP_RD_B(cur, 1), PP_SLOTS(PPD0, PPD0 + PPD1, cur, t));                                             \
    if (more1) { PP_VMCNT_N(4 + PPD0 + PPD1); } else { PP_VMCNT(0); }     /* A1(t) landed */               \
    PP_BAR(); PP_STAMP(2); PP_COMPUTE(0, 1); PP_BAR(); PP_STAMP(3);                                         \
    /* phase 2 */                                                                                           \
    PP_M(PP_RD_A(cur, 1), PP_SLOTS(PPD0 + PPD1, PPD0 + PPD1 + PPD2, cur, t));                               \
    PP_BAR(); PP_STAMP(4); PP_COMPUTE(1, 1); PP_BAR(); PP_STAMP(5);                                         \
    /* phase 3: B0 fragments are still in bq[0] */                                                          \
    PP_M(, PP_SLOTS(PPD0 + PPD1 + PPD2, 8, cur, t));                                                        \
    if (more2) { PP_VMCNT(8); } else if (more1) { PP_VMCNT(4); }      /* A0(t+1), B0(t+1) landed */         \
    PP_BAR(); PP_STAMP(6); PP_COMPUTE(1, 0); PP_BAR(); PP_STAMP(7);                                         \
    PP_STAMPS_FOLD();                                                                                       \
  } while (0)
#endif

#if DXA_PPR
  // ---- prologue: the four pieces of tile 0 and A0, B0, B1 of tile 1 in the steady state's issue order; A0(0), B0(0), B1(0) landed
  //      for every wave before the first read; tile 0's A0 / B0 fragments are read here (C3 of "tile -1")
  PP_DMA_A(0, 0, 0); PP_DMA_B(0, 0, 0); PP_DMA_B(1, 0, 0); PP_DMA_A(1, 0, 0);
#if DXA_PPR == 2
  if (nk > 1) { PP_DMA_B(0, 1, 1); PP_DMA_A(0, 1, 1); PP_DMA_B(1, 1, 1); }
#else
  if (nk > 1) { PP_DMA_A(0, 1, 1); PP_DMA_B(0, 1, 1); PP_DMA_B(1, 1, 1); }
#endif
  PP_SB();
  if (nk > 1) { PP_VMCNT(8); } else { PP_VMCNT(2); }
  PP_BAR();
  PP_RD_A(0, 0); PP_RD_B(0, 0);
  PP_SB();
#else
  // ---- prologue: the four pieces of tile 0 and the first two of tile 1; A0(0) and B0(0) landed for every wave before
  //      the first read
  PP_DMA_A(0, 0, 0); PP_DMA_B(0, 0, 0); PP_DMA_B(1, 0, 0); PP_DMA_A(1, 0, 0);
  if (nk > 1) { PP_DMA_A(0, 1, 1); PP_DMA_B(0, 1, 1); }
  PP_SB();
  if (nk > 1) { PP_VMCNT(8); } else { PP_VMCNT(4); }
  PP_BAR();
#endif
  if (!(DXA_PPV & 16) && wm == 1) PP_BAR();    // group 1 runs one barrier interval behind group 0
  for (int t = 0; t < nk; t += 2) {
    PP_TILE(0, t);
    if (t + 1 < nk) PP_TILE(1, t + 1);
  }
  if (!(DXA_PPV & 16) && wm == 0) PP_BAR();    // every wave has now passed the same number of barriers
#if (DXA_PPV & 128)
  const uint32_t iv0_ = iv_[0], iv1_ = iv_[1], iv2_ = iv_[2], iv3_ = iv_[3], iv4_ = iv_[4], iv5_ = iv_[5], iv6_ = iv_[6], iv7_ = iv_[7];
#endif
#undef PP_A_ROW0
#undef PP_B_ROW0
#undef PP_DMA
#undef PP_LOOP
#undef PP_DMA_A
#undef PP_DMA_B
#undef PP_READ
#undef PP_READ_TR
#undef PP_FRAG_TR
#undef PP_RD_A
#undef PP_RD_B
#undef PP_MFMA
#undef PP_SB
#undef PP_BAR
#undef PP_COMPUTE
#undef PP_VMCNT
#undef PP_TILE
#undef PP_SLOT
#undef PP_SLOTS
#undef PP_VMCNT_N
#undef PP_DMA_A1
#undef PP_DMA_B1
#undef PP_M
#undef PP_STAMP
#undef PP_STAMPS_FOLD
#undef PP_RDK_A
#undef PP_RDK_B
#undef PP_COMPUTE_R
#undef PP_COMPUTE_R2
#undef PP_RDK_B2
#undef PP_MFMA2
#if DXA_PPR == 2
#undef PP_R2_B1
#undef PP_R2_A1
#undef PP_R2_B0N
#undef PP_R2_A0N
#elif DXA_PPR
#undef PP_R_B1
#undef PP_R_A1
#undef PP_R_NEXT
#endif

  if constexpr (LEAN) {
    if (!tile_split_exchange<4>(p, acc, tid, split_j, split_s, tail_i)) return;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FUSE == 1) sk_epilogue_swiglu<4>(p, acc, smem + wave * 4096, lane, wm, wn, m0i, n0i);
    else sk_epilogue<TO, TE>(p, acc, smem + wave * 4096, lane, wm, wn, m0i, n0i, reinterpret_cast<float*>(smem + 8 * 4096), bid);
#if (DXA_PPV & 128)
    __syncthreads();
    if (m0i == 0 && n0i == 0 && lane == 0) {          // the first tile's waves: [wave][8 intervals + K tiles] over the head of C
      uint32_t* o_ = reinterpret_cast<uint32_t*>(p.C) + wave * 16;
      o_[0] = iv0_; o_[1] = iv1_; o_[2] = iv2_; o_[3] = iv3_; o_[4] = iv4_; o_[5] = iv5_; o_[6] = iv6_; o_[7] = iv7_;
      o_[8] = (uint32_t)nk;
    }
#endif
  } else {
    tile_finish<TO, 4, TE>(p, acc, smem, tid, lane, wave, wm, wn, l32, lh, m0, n0, split_j, split_s, tail_i);
  }
#endif  // __HIP_DEVICE_COMPILE__
}
// NT: every forward linear, dX against an explicit W^T (fp32 head through bf16x3); NN: dX = dY W; TN: dW = dY^T X
template __global__ void gemm_pp_kernel<bf16_t, bf16_t, true, false, false>(const GemmP);
template __global__ void gemm_pp_kernel<float, bf16_t, true, false, false>(const GemmP);
template __global__ void gemm_pp_kernel<float, float, true, false, false>(const GemmP);
template __global__ void gemm_pp_kernel<bf16_t, bf16_t, false, false, false>(const GemmP);
template __global__ void gemm_pp_kernel<float, bf16_t, false, false, false>(const GemmP);
template __global__ void gemm_pp_kernel<float, float, false, false, false>(const GemmP);
template __global__ void gemm_pp_kernel<bf16_t, bf16_t, true, false, true>(const GemmP);
template __global__ void gemm_pp_kernel<bf16_t, bf16_t, false, false, true>(const GemmP);
template __global__ void gemm_pp_kernel<float, bf16_t, true, false, true>(const GemmP);
template __global__ void gemm_pp_kernel<float, bf16_t, true, true, true>(const GemmP);
template __global__ void gemm_pp_kernel<bf16_t, bf16_t, true, true, true>(const GemmP);
template __global__ void gemm_pp_kernel<bf16_t, bf16_t, true, false, false, 1>(const GemmP);     // gate / up + SiLU * up

// =====================================================================================================
// Fast path 2b (round 4): the ping-pong schedule on a 192-row tile — bf16 NT, K % 64 == 0, both operands K-contiguous.
//
// Row counts that pad badly to 256 (the one-request prefill: 543 rows = 2.1 tiles of 256, 2.8 of 192) ran on the round-1 ring
// kernel with 192-row tiles, whose eight waves all do their own LDS-DMA, fragment reads and MFMAs in lockstep (one barrier per
// 32-deep slab): 0.80-0.86 PF/s where the ping-pong kernel reaches 1.24-1.41 on its 256-row tiles.  This is that kernel's schedule
// — two wave groups one barrier interval apart alternating memory and compute clusters, fragment reads in the gaps of the
// compute cluster before the one that needs them — for 8 waves as 2 (M) x 4 (N) with 96 x 64 per wave:
//   K tile = 3 phases; phase q: compute cluster C_q = A block q (32 rows) x (B0, B1) x 4 k-steps = 8 MFMAs.
//   Both B halves stay in registers for the whole K tile (two sets, swapped per tile), the A block registers are double-buffered.
//   reads in the gaps:  C0: A1(t);   C1: A2(t), B0(t+1);   C2: A0(t+1), B1(t+1)            (4 / 8 / 8 ds_read_b128)
//   LDS-DMA pieces per K tile: A0, A1, A2 (64 rows = 8 KiB: 1 instruction per wave) and B0, B1 (128 rows: 2 per wave) into two
//   buffers of 56 KiB;  issue  M0: A2(t+1), B0(t+2);  M1: A0(t+2), B1(t+2);  M2: A1(t+2) — each exactly two phases after the read of
//   the bytes it overwrites — and every piece is waited for by every wave at the end of the memory cluster BEFORE the compute
//   cluster that reads it: always vmcnt(7) in the steady state (7 instructions per wave and K tile).
// Same MFMA (v_mfma_f32_32x32x16_bf16), same K order, same epilogue (tile_finish<.., 3, ..>) as the ring kernel: bit-identical.
// =====================================================================================================
#if defined(DXA_PP3_STAMPS)      // tuning build: s_memtime at entry / first operands landed / main loop done / epilogue done of ONE workgroup
__device__ unsigned long long g_pp3_stamps[4];
#define P3_STAMP(i) do { if (blockIdx.x == DXA_PP3_STAMPS && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_pp3_stamps[i] = t_; } } while (0)
#else
#define P3_STAMP(i) do { } while (0)
#endif
template <typename TO, typename TE, bool LEAN, int FUSE = 0>
__global__ __launch_bounds__(512) void gemm_pp3_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  P3_STAMP(0);
  constexpr int AREG = 192 * 128, BREG = 256 * 128, BUFSZ = AREG + BREG;      // 24 KiB + 32 KiB per K tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l32 = lane & 31, lh = lane >> 5;
  int bid = blockIdx.x;
  int split_j = 0, split_s = 1, tail_i = 0;
  if (bid < p.full) {
    const int q = p.full >> 3, r = p.full & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  } else {
    const int idx = bid - p.full;
    tail_i = idx % p.tail_r;
    split_j = idx / p.tail_r;
    split_s = p.split_s;
    bid = p.full + tail_i;
  }
  const int width = p.group_m * p.tn;
  const int group = bid / width;
  const int first_pm = group * p.group_m;
  const int gsz = min(p.tm - first_pm, p.group_m);
  const int pm = first_pm + (bid % width) % gsz;
  const int pn = (bid % width) / gsz;
  const int64_t m0 = (int64_t)pm * 192, n0 = (int64_t)pn * 256;

  const uint32_t bytesA = (uint32_t)(((p.M - 1) * p.lda + p.K) * 2);
  const uint32_t bytesB = (uint32_t)(((p.N - 1) * p.ldb + p.K) * 2);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.A), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, bytesB, 0x00020000);
  // LDS-DMA: an instruction covers 8 tile rows (lane -> row0 + lane / 8, LDS slot lane % 8, source chunk slot ^ ((row >> 1) & 7)).
  //   A block g: this wave's instruction = rows (wave >> 2) * 96 + 32 g + (wave & 3) * 8 ..+7   (both wave rows' block g: 64 rows)
  //   B half h : instructions j = 0, 1 = rows ((2 wave + j) >> 2) * 64 + ((2 wave + j) & 3) * 8 + 32 h ..+7   (as the 256-row kernel)
#define P3_A_ROW0(g) ((wave >> 2) * 96 + 32 * (g) + (wave & 3) * 8)
#define P3_B_ROW0(h, j) ((((wave * 2 + (j)) >> 2) * 64) + (((wave * 2 + (j)) & 3) * 8) + 32 * (h))
  uint32_t voA[3], voB[2][2];
  {
    const int lrow = lane >> 3, lslot = lane & 7;
    const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int ra = P3_A_ROW0(g) + lrow;
      const int64_t ga = m0 + ra;
      voA[g] = ga < p.M ? (uint32_t)ga * lda2 + (uint32_t)((lslot ^ ((ra >> 1) & 7)) << 4) : 0x80000000u;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rb = P3_B_ROW0(h, j) + lrow;
        if constexpr (FUSE == 1) {          // [gate ; up]: see gemm_pp_kernel
          const int half_n = (int)(p.N >> 1);
          const int lc = (int)(n0 >> 1) + (rb >> 6) * 32 + (rb & 31), gb = ((rb >> 5) & 1) * half_n + lc;
          voB[h][j] = lc < half_n ? (uint32_t)gb * ldb2 + (uint32_t)((lslot ^ ((rb >> 1) & 7)) << 4) : 0x80000000u;
        } else {
          const int64_t gb = n0 + rb;
          voB[h][j] = gb < p.N ? (uint32_t)gb * ldb2 + (uint32_t)((lslot ^ ((rb >> 1) & 7)) << 4) : 0x80000000u;
        }
      }
  }
  const int nk_tot = (int)(p.K / 64);
  const int k_lo = split_j * nk_tot / split_s;
  const int nk = (split_j + 1) * nk_tot / split_s - k_lo;   // K tiles of this workgroup (>= 1)
#define P3_DMA(rsrc, vo, ldsoff, buf, tile) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(smem + (buf) * BUFSZ + (ldsoff)), 16, vo, (k_lo + (tile)) * 128, 0, 0)
#define P3_DMA_A(g, buf, tile) P3_DMA(rA, voA[g], P3_A_ROW0(g) * 128, buf, tile)
#define P3_DMA_B(h, buf, tile) do { P3_DMA(rB, voB[h][0], AREG + P3_B_ROW0(h, 0) * 128, buf, tile); P3_DMA(rB, voB[h][1], AREG + P3_B_ROW0(h, 1) * 128, buf, tile); } while (0)

  f32x16_t acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row base | ((lh ^ sw) << 4), XOR (ks << 5) per k-step; one base per buffer (56 KiB is no power of two)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int sw = (l32 >> 1) & 7;
  const uint32_t ya0 = (lds0 + (wm * 96 + l32) * 128) | (uint32_t)((lh ^ sw) << 4), ya1 = ya0 + BUFSZ;
  const uint32_t yb0 = (lds0 + AREG + (wn * 64 + l32) * 128) | (uint32_t)((lh ^ sw) << 4), yb1 = yb0 + BUFSZ;
  u32x4_t af[2][4], bq[2][2][4];
#define P3_READ(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define P3_RD_A(buf, g, set, ks) P3_READ(af[set][ks], ((buf) ? ya1 : ya0) ^ (uint32_t)(32 * (ks)), (g) * 4096)
#define P3_RD_B(buf, h, bset, ks) P3_READ(bq[bset][h][ks], ((buf) ? yb1 : yb0) ^ (uint32_t)(32 * (ks)), (h) * 4096)
#define P3_MFMA(g, set, bset, j, ks) acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bq[bset][j][ks]), __builtin_bit_cast(bf16x8_t, af[set][ks]), acc[g][j], 0, 0, 0)
#define P3_SB() __builtin_amdgcn_sched_barrier(0)
#define P3_BAR() do { P3_SB(); __builtin_amdgcn_s_barrier(); P3_SB(); } while (0)
  // compute cluster of A block g (registers af[set]) against both B halves (registers bq[bset]); R(ks) rides in the gap after k-step ks
#define P3_COMPUTE(g, set, bset, R)                                                                 \
  do {                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
    P3_SB();                                                                                        \
    P3_MFMA(g, set, bset, 0, 0); P3_MFMA(g, set, bset, 1, 0); P3_SB(); R(0); P3_SB();               \
    P3_MFMA(g, set, bset, 0, 1); P3_MFMA(g, set, bset, 1, 1); P3_SB(); R(1); P3_SB();               \
    P3_MFMA(g, set, bset, 0, 2); P3_MFMA(g, set, bset, 1, 2); P3_SB(); R(2); P3_SB();               \
    P3_MFMA(g, set, bset, 0, 3); P3_MFMA(g, set, bset, 1, 3); P3_SB(); R(3); P3_SB();               \
  } while (0)
#define P3_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  // One K tile in buffer cur = t & 1.  A blocks alternate between the two register sets phase by phase (3 phases per tile: the
  // parity flips per tile, hence `par`), the B sets per tile.
  //   tile with par = 0: A0 in af[0], A1 -> af[1], A2 -> af[0], A0(t+1) -> af[1];   par = 1: the mirror image
#define P3_TILE(cur, t)                                                                                                  \
  do {                                                                                                                   \
    constexpr int cur_ = (cur), s0_ = (cur), s1_ = (cur) ^ 1;        /* af set of A0 / of A1; B set of this tile = cur */    \
    const bool more1 = (t) + 1 < nk, more2 = (t) + 2 < nk;                                                               \
    /* phase 0 */                                                                                                        \
    if (more1) P3_DMA_A(2, cur_ ^ 1, (t) + 1);                                                                           \
    if (more2) { P3_DMA_B(0, cur_, (t) + 2); P3_SB(); P3_VMCNT(7); } else if (more1) { P3_SB(); P3_VMCNT(5); } else { P3_VMCNT(0); } \
    P3_BAR();                                                                                                            \
    P3_COMPUTE(0, s0_, cur_, P3_R_A1);                                                                                   \
    P3_BAR();                                                                                                            \
    /* phase 1 */                                                                                                        \
    if (more2) { P3_DMA_A(0, cur_, (t) + 2); P3_DMA_B(1, cur_, (t) + 2); P3_SB(); P3_VMCNT(7); } else if (more1) { P3_VMCNT(2); } \
    P3_BAR();                                                                                                            \
    P3_COMPUTE(1, s1_, cur_, P3_R_A2B0);                                                                                 \
    P3_BAR();                                                                                                            \
    /* phase 2 */                                                                                                        \
    if (more2) { P3_DMA_A(1, cur_, (t) + 2); P3_SB(); P3_VMCNT(7); } else if (more1) { P3_VMCNT(1); }                    \
    P3_BAR();                                                                                                            \
    P3_COMPUTE(2, s0_, cur_, P3_R_A0B1);                                                                                 \
    P3_BAR();                                                                                                            \
  } while (0)
#define P3_R_A1(ks) P3_RD_A(cur_, 1, s1_, ks)
#define P3_R_A2B0(ks) do { P3_RD_A(cur_, 2, s0_, ks); if (more1) P3_RD_B(cur_ ^ 1, 0, cur_ ^ 1, ks); } while (0)
#define P3_R_A0B1(ks) do { if (more1) { P3_RD_A(cur_ ^ 1, 0, s1_, ks); P3_RD_B(cur_ ^ 1, 1, cur_ ^ 1, ks); } } while (0)

  // ---- prologue: tile 0 and what the steady state would have issued for tile 1 before M0(0), in its order:
  //      B0(0) A0(0) B1(0) | A1(0) | A2(0) B0(1) | A0(1) B1(1) | A1(1);   B0, A0, B1, A1 of tile 0 landed for every wave before the first read
  P3_DMA_B(0, 0, 0); P3_DMA_A(0, 0, 0); P3_DMA_B(1, 0, 0); P3_DMA_A(1, 0, 0); P3_DMA_A(2, 0, 0);
  if (nk > 1) { P3_DMA_B(0, 1, 1); P3_DMA_A(0, 1, 1); P3_DMA_B(1, 1, 1); P3_DMA_A(1, 1, 1); }
  P3_SB();
  if (nk > 1) { P3_VMCNT(7); } else { P3_VMCNT(1); }
  P3_BAR();
  P3_STAMP(1);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { P3_RD_A(0, 0, 0, ks); P3_RD_B(0, 0, 0, ks); P3_RD_B(0, 1, 0, ks); }
  P3_SB();
  if (wm == 1) P3_BAR();            // group 1 runs one barrier interval behind group 0
  for (int t = 0; t < nk; t += 2) {
    P3_TILE(0, t);
    if (t + 1 < nk) P3_TILE(1, t + 1);
  }
  if (wm == 0) P3_BAR();            // every wave has now passed the same number of barriers
  P3_STAMP(2);
#undef P3_A_ROW0
#undef P3_B_ROW0
#undef P3_DMA
#undef P3_DMA_A
#undef P3_DMA_B
#undef P3_READ
#undef P3_RD_A
#undef P3_RD_B
#undef P3_MFMA
#undef P3_SB
#undef P3_BAR
#undef P3_COMPUTE
#undef P3_VMCNT
#undef P3_TILE
#undef P3_R_A1
#undef P3_R_A2B0
#undef P3_R_A0B1
  if constexpr (LEAN) {          // C = alpha acc + bias (+ residual) (+ C) over whole 16-byte accesses: the 256-row kernel's epilogue
    if (!tile_split_exchange<3>(p, acc, tid, split_j, split_s, tail_i)) return;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FUSE == 1) sk_epilogue_swiglu<3>(p, acc, smem + wave * 4096, lane, wm, wn, (int)m0, (int)n0);
    else sk_epilogue<TO, TE, 3>(p, acc, smem + wave * 4096, lane, wm, wn, (int)m0, (int)n0, reinterpret_cast<float*>(smem + 8 * 4096), bid);
  } else {
    tile_finish<TO, 3, TE>(p, acc, smem, tid, lane, wave, wm, wn, l32, lh, m0, n0, split_j, split_s, tail_i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P3_STAMP(3);
#endif  // __HIP_DEVICE_COMPILE__
}
#undef P3_STAMP
template __global__ void gemm_pp3_kernel<bf16_t, bf16_t, true>(const GemmP);
template __global__ void gemm_pp3_kernel<float, bf16_t, true>(const GemmP);
template __global__ void gemm_pp3_kernel<bf16_t, bf16_t, false>(const GemmP);
template __global__ void gemm_pp3_kernel<float, bf16_t, false>(const GemmP);
template __global__ void gemm_pp3_kernel<float, float, true>(const GemmP);      // fp32 epilogue operands (bf16x3 products of the fp32 head)
template __global__ void gemm_pp3_kernel<float, float, false>(const GemmP);
template __global__ void gemm_pp3_kernel<bf16_t, bf16_t, true, 1>(const GemmP);                  // gate / up + SiLU * up

// x = hi + lo (two bf16): dst[r] = [hi | hi | lo] (side 0) or [hi | lo | hi] (side 1), 4 elements per thread
__global__ __launch_bounds__(256) void split3_k(const float* __restrict__ src, int64_t ld, bf16_t* __restrict__ dst,
                                                int64_t rows, int64_t cols, int side) {
  const int64_t c4 = cols >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * c4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c4, c = (i - r * c4) * 4;
    float x[4];
    Vec<float, 4>::ld(x, src + r * ld + c);
    float hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = bf2f(f2bf(x[e]));
      lo[e] = x[e] - hi[e];
    }
    bf16_t* d = dst + r * 3 * cols + c;
    Vec<bf16_t, 4>::st(d, hi);
    Vec<bf16_t, 4>::st(d + cols, side == 0 ? hi : lo);
    Vec<bf16_t, 4>::st(d + 2 * cols, side == 0 ? lo : hi);
  }
}

template <typename TI, typename TO, int TM>
int launch(const GemmP& p, int layout, dim3 grid, hipStream_t st) {
  constexpr size_t LDS = 4 * (32 * TM) * ROWB;
  switch (layout) {
    case DXA_NT: hipLaunchKernelGGL((gemm_kernel<TI, TO, false, false, TM>), grid, dim3(256), LDS, st, p); break;
    case DXA_NN: hipLaunchKernelGGL((gemm_kernel<TI, TO, false, true, TM>), grid, dim3(256), LDS, st, p); break;
    case DXA_TN: hipLaunchKernelGGL((gemm_kernel<TI, TO, true, true, TM>), grid, dim3(256), LDS, st, p); break;
    default: return DXA_ERR_BAD_ARG;
  }
  return DXA_OK;
}

// Split-K scratch of the ring kernel: < 256 slots of 256 KiB + 256 counters per (device, stream), allocated on
// first use (so the first dxa_gemm on a stream must not run under stream capture) and kept for the process.
constexpr int NUM_CU = 256;
static_assert(NUM_CU == NUM_CU_D, "split-K scratch sizing");
struct SplitWs { float* ws; int* flags; };
int get_split_ws(hipStream_t st, SplitWs* out) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SplitWs> tab;
  int dev = 0;
  DXA_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  auto it = tab.find({dev, st});
  if (it == tab.end()) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      dxa_set_error("dxa_gemm: first split-K product on a stream allocates its scratch and cannot happen under stream "
                    "capture: run the request once eagerly on this stream first");
      return DXA_ERR_BAD_ARG;
    }
    char* base = nullptr;
    const size_t ws_bytes = (size_t)NUM_CU * 256 * 256 * 4;
    DXA_CHECK_HIP(hipMalloc((void**)&base, ws_bytes + NUM_CU * sizeof(int)));
    DXA_CHECK_HIP(hipMemset(base + ws_bytes, 0, NUM_CU * sizeof(int)));
    DXA_CHECK_HIP(hipDeviceSynchronize());
    it = tab.emplace(std::make_pair(dev, st), SplitWs{(float*)base, (int*)(base + ws_bytes)}).first;
  }
  *out = it->second;
  return DXA_OK;
}

// persistent kernels launch one workgroup per compute unit
inline int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
    else
      n = NUM_CU;
  }
  return n;
}
inline bool skinny_off_g() {
  static const bool off = getenv("DXA_GEMM_NO_SKINNY") != nullptr;
  return off;
}
inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline bool strides_mult(const int64_t s[3], int64_t m) { return s[0] % m == 0 && s[1] % m == 0 && s[2] % m == 0; }

}  // namespace

namespace {
int gemm_dispatch(const dxa_gemm_desc* d, dxa_stream_t stream, bool* mirrored, bool* summed);

// sum of squares of C [M, N] (fp32, leading dimension ldc) into `slots` per-"tile" partials for products whose kernel has no
// sum-of-squares epilogue: workgroup b folds rows b, b + slots, ... in a fixed order, so every slot is written
template <typename T>
__global__ __launch_bounds__(256) void sumsq_rows_k(const T* __restrict__ C, int64_t ldc, int64_t M, int64_t N,
                                                    float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t r = blockIdx.x; r < M; r += gridDim.x) {
    const T* row = C + r * ldc;
    for (int64_t c = threadIdx.x; c < N; c += 256) { const float v = ldf<T>(row + c); s += v * v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
}
extern "C" int64_t dxa_gemm_sumsq_slots(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  return ((M + 255) / 256) * ((N + 255) / 256);
}
extern "C" int dxa_gemm(const dxa_gemm_desc* d, dxa_stream_t stream) {
  bool mirrored = false, summed = false;
  if (int rc = gemm_dispatch(d, stream, &mirrored, &summed)) return rc;
  if (d->mirror && !mirrored && d->M > 0 && d->N > 0) {   // kernels without the mirror epilogue: one narrow copy pass
    if (int rc = dxa_copy2d(d->C, d->ldc, d->mirror, d->ldc, d->M, d->N, d->N, DXA_F32, DXA_BF16, stream)) return rc;
  }
  if (d->sumsq && !summed && d->M > 0 && d->N > 0) {      // ... and without the sum-of-squares epilogue: one read of C
    const dim3 sgrid((unsigned)dxa_gemm_sumsq_slots(d->M, d->N));
    if (d->out_dtype == DXA_F32)
      hipLaunchKernelGGL(sumsq_rows_k<float>, sgrid, dim3(256), 0, (hipStream_t)stream, (const float*)d->C, d->ldc, d->M, d->N, d->sumsq);
    else
      hipLaunchKernelGGL(sumsq_rows_k<bf16_t>, sgrid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d->C, d->ldc, d->M, d->N, d->sumsq);
    DXA_CHECK_LAUNCH();
  }
  return DXA_OK;
}
namespace {
int gemm_dispatch(const dxa_gemm_desc* d, dxa_stream_t stream, bool* mirrored, bool* summed) {
  DXA_CHECK_ARG(d != nullptr, "dxa_gemm: null desc");
  DXA_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0, "dxa_gemm: negative dims");
  DXA_CHECK_ARG(d->layout >= DXA_NT && d->layout <= DXA_TN, "dxa_gemm: bad layout %d", d->layout);
  DXA_CHECK_ARG(d->in_dtype == DXA_F32 || d->in_dtype == DXA_BF16, "dxa_gemm: bad in_dtype %d", d->in_dtype);
  DXA_CHECK_ARG(d->out_dtype == d->in_dtype || d->out_dtype == DXA_F32,
                "dxa_gemm: out_dtype must equal in_dtype or be fp32");
  DXA_CHECK_ARG(d->nb[0] >= 1 && d->nb[1] >= 1 && d->nb[2] >= 1, "dxa_gemm: batch extents must be >= 1");
  if (d->M == 0 || d->N == 0) return DXA_OK;
  DXA_CHECK_ARG(d->A && d->B && d->C, "dxa_gemm: null operand");
  const int64_t nbatch = (int64_t)d->nb[0] * d->nb[1] * d->nb[2];
  DXA_CHECK_ARG(nbatch <= 65535, "dxa_gemm: too many batches (%lld)", (long long)nbatch);

  const size_t es = d->in_dtype == DXA_BF16 ? 2 : 4, os = d->out_dtype == DXA_BF16 ? 2 : 4;
  const int64_t epc = 16 / es;
  // ---- a contraction length that is not a multiple of 64 (SigLIP-So400m's 4304-wide MLP: fc2 forward, fc1 dX) would send a
  //      large bf16 NT / NN product to the generic kernel (3x slower): K = K0 + K1 with K0 = K - K % 64 on the MFMA fast path
  //      (bias / residual epilogue applied there) and the < 64-deep tail accumulated on top by the generic kernel.  With a bf16
  //      C the partial result is rounded once more than a single pass would (one extra bf16 rounding of the output).
  static const bool ksplit_off = getenv("DXA_GEMM_NO_KTAIL") != nullptr;
  if (!ksplit_off && d->in_dtype == DXA_BF16 && nbatch == 1 && d->layout != DXA_TN && d->K % 64 != 0 && d->K >= 1024 &&
      d->M >= 256 && d->N >= 256 && d->act == DXA_ACT_NONE && !d->aux_out && !d->mulgrad && !d->sumsq && !d->mirror && !d->epi_f32) {
    const int64_t K0 = d->K - d->K % 64;
    dxa_gemm_desc head = *d, tail = *d;
    head.K = K0;
    tail.K = d->K - K0;
    tail.accumulate = 1; tail.bias = nullptr; tail.residual = nullptr;
    tail.A = (const char*)d->A + K0 * (int64_t)es;
    tail.B = (const char*)d->B + (d->layout == DXA_NT ? K0 : K0 * d->ldb) * (int64_t)es;
    bool m2 = false, s2 = false;
    if (int rc = gemm_dispatch(&head, stream, &m2, &s2)) return rc;
    return gemm_dispatch(&tail, stream, &m2, &s2);
  }
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.A = (const char*)d->A; p.lda = d->lda;
  p.B = (const char*)d->B; p.ldb = d->ldb;
  p.C = (char*)d->C; p.ldc = d->ldc;
  p.bias = (const char*)d->bias;
  p.R = (const char*)d->residual; p.ldr = d->ldr;
  p.aux = (char*)d->aux_out;
  p.G = (const char*)d->mulgrad; p.ldg = d->ldg;
  p.alpha = d->alpha; p.act = d->act; p.accumulate = d->accumulate;
  p.fuse = d->fuse; p.ldaux = d->ld_aux;
  if (d->fuse != DXA_FUSE_NONE) {
    DXA_CHECK_ARG(d->fuse == DXA_FUSE_SWIGLU, "dxa_gemm: unknown fuse mode %d", d->fuse);
    const int64_t F_ = d->N / 2;
    DXA_CHECK_ARG(d->layout == DXA_NT && d->in_dtype == DXA_BF16 && d->out_dtype == DXA_BF16 && nbatch == 1 && d->N % 2 == 0 &&
                  F_ % 8 == 0 && d->K % 64 == 0 && d->K >= 64 && d->M >= 129 && F_ >= 128 && !d->bias && !d->residual && !d->mulgrad &&
                  d->act == DXA_ACT_NONE && !d->accumulate && !d->mirror && !d->sumsq && !d->epi_f32 && d->K2 == 0,
                  "dxa_gemm: DXA_FUSE_SWIGLU is a plain bf16 NT product of the MFMA fast path (M >= 129, K %% 64 == 0, F %% 8 == 0)");
    DXA_CHECK_ARG(aligned_to(d->A, 16) && aligned_to(d->B, 16) && aligned_to(d->C, 8) && d->lda % 8 == 0 && d->ldb % 8 == 0 &&
                  d->ldc % 4 == 0 && d->ldc >= F_ && (!d->aux_out || (aligned_to(d->aux_out, 8) && d->ld_aux % 4 == 0 && d->ld_aux >= d->N)) &&
                  ((d->M - 1) * d->ld_aux + d->N) * 2 < (1ll << 31) && ((d->M - 1) * d->lda + d->K) * 2 < (1ll << 31) &&
                  ((d->N - 1) * d->ldb + d->K) * 2 < (1ll << 31),
                  "dxa_gemm: DXA_FUSE_SWIGLU needs 16-byte aligned operand rows, 8-byte aligned output rows and operands < 2 GiB");
  }
  DXA_CHECK_ARG(!d->mirror || (d->out_dtype == DXA_F32 && nbatch == 1), "dxa_gemm: mirror needs an fp32, unbatched output");
  DXA_CHECK_ARG(!d->sumsq || nbatch == 1, "dxa_gemm: sumsq needs an unbatched output");
  p.nb1 = d->nb[1]; p.nb2 = d->nb[2];
  for (int i = 0; i < 3; ++i) {
    p.sA[i] = d->sA[i]; p.sB[i] = d->sB[i]; p.sC[i] = d->sC[i]; p.sR[i] = d->sR[i]; p.sG[i] = d->sG[i];
  }
  p.tm = dxa_cdiv(d->M, BM);
  p.tn = dxa_cdiv(d->N, BN);
  const bool a_ks = d->layout == DXA_TN, b_ks = d->layout != DXA_NT;
  // vector path: k-contiguous needs 16-B aligned rows; k-strided loads 4 consecutive rows (4*es bytes)
  p.vecA = a_ks ? (aligned_to(d->A, 4 * es) && d->lda % 4 == 0 && strides_mult(d->sA, 4))
                : (aligned_to(d->A, 16) && d->lda % epc == 0 && strides_mult(d->sA, epc));
  p.vecB = b_ks ? (aligned_to(d->B, 4 * es) && d->ldb % 4 == 0 && strides_mult(d->sB, 4))
                : (aligned_to(d->B, 16) && d->ldb % epc == 0 && strides_mult(d->sB, epc));
  p.vecC = aligned_to(d->C, 4 * os) && d->ldc % 4 == 0 && strides_mult(d->sC, 4) &&
           (!d->aux_out || aligned_to(d->aux_out, 4 * os));
  const size_t ees = d->epi_f32 ? 4 : es;   // element size of bias / residual / mulgrad
  DXA_CHECK_ARG(!d->epi_f32 || (d->in_dtype == DXA_BF16 && d->out_dtype == DXA_F32), "dxa_gemm: epi_f32 needs bf16 in, fp32 out");
  p.vecR = d->residual && aligned_to(d->residual, 4 * ees) && d->ldr % 4 == 0 && strides_mult(d->sR, 4);
  p.vecG = d->mulgrad && aligned_to(d->mulgrad, 4 * ees) && d->ldg % 4 == 0 && strides_mult(d->sG, 4);
  p.vecBias = d->bias && aligned_to(d->bias, 4 * ees);

  hipStream_t st = (hipStream_t)stream;
  // ---- fast paths (bf16, no batching, 16-byte aligned rows, operands < 2 GiB): NT with K % 32 == 0 (ring) / K % 64 == 0
  //      (ping-pong); NN and TN (k-strided operands staged as they lie, ping-pong kernel only): NN needs K % 64 == 0 for its
  //      k-contiguous A, TN takes any K
  const int64_t bytesA = (a_ks ? (d->K - 1) * d->lda + d->M : (d->M - 1) * d->lda + d->K) * 2;
  const int64_t bytesB = (b_ks ? (d->K - 1) * d->ldb + d->N : (d->N - 1) * d->ldb + d->K) * 2;
  static const bool fast_off = getenv("DXA_GEMM_NO_FAST") != nullptr;
  static const bool ks_off = getenv("DXA_GEMM_NO_KS") != nullptr;
  const bool ks_layout = d->layout != DXA_NT;
  const int64_t cpl = 16 / (int64_t)os;
  // the lean epilogue: C = alpha * acc + bias (+ residual) (+ C) made of whole 16-byte accesses
  const bool lean_ok = d->N % cpl == 0 && d->ldc % cpl == 0 && aligned_to(d->C, 16) &&
                       ((d->M - 1) * d->ldc + d->N) * (int64_t)os < (1ll << 31) && !d->aux_out && !d->mulgrad &&
                       d->act == DXA_ACT_NONE && (!d->bias || aligned_to(d->bias, cpl * ees)) &&
                       (!d->residual || (aligned_to(d->residual, cpl * ees) && d->ldr % cpl == 0 &&
                                         ((d->M - 1) * d->ldr + d->N) * (int64_t)ees < (1ll << 31)));
  const bool ks_ok = (d->layout == DXA_NN ? (d->out_dtype == DXA_BF16 || lean_ok) : lean_ok) && !d->epi_f32 && ks_layout && !ks_off && aligned_to(d->A, 16) && aligned_to(d->B, 16) && d->lda % 8 == 0 && d->ldb % 8 == 0 &&
                     (d->layout == DXA_TN || d->K % 64 == 0) && d->K >= 64;
  // ---- few-row NT products (batch-1 prefill, ViT on a couple of images): 128x128 tiles fill the chip where 256-row tiles
  //      cannot; any epilogue of the bf16 menu
  static const bool t128_off = getenv("DXA_GEMM_NO_T128") != nullptr;
  static const int t128_max_m = getenv("DXA_GEMM_T128_MAX_M") ? atoi(getenv("DXA_GEMM_T128_MAX_M")) : 1024;
  // measured (scripts/gemm_bench.py pre, M = 543 / 514): wins where its tiles fit one round of the 256 CUs and K is short
  // (qkv 45 vs 50 us, o_proj 44 vs 52, ViT fc1 20 vs 31); loses to the 192-row ring kernel + split-K tail on wide N or deep K
  // (gate_up 211 vs 181, down 181 vs 120): an LDS-DMA instruction costs its wave 60-180 issue cycles, and a 128x128 tile
  // needs twice as many of them per MFMA as a 256x256 tile
  const int64_t t128_tiles = (int64_t)dxa_cdiv(d->M, 128) * dxa_cdiv(d->N, 128);
  static const bool t128_all = getenv("DXA_GEMM_T128_NS") != nullptr;     // tuning: every admissible shape
  // (epi_f32 = the split-bf16 fp32 products of the action head: M = 64 x 17 rows, K' = 3K up to 9216 — their 128x128 tiles fill
  //  54-216 CUs where the 192-row ring kernel's fill 18-72)
  // opt-in (DXA_GEMM_T128_F32EPI=1): parity-green, but the step measured the same with it (255.6 vs 256.2 ms on one box)
  static const bool t128_f32epi = getenv("DXA_GEMM_T128_F32EPI") && atoi(getenv("DXA_GEMM_T128_F32EPI")) != 0;
  if (!fast_off && !t128_off && d->fuse == DXA_FUSE_NONE && d->layout == DXA_NT && d->in_dtype == DXA_BF16 && nbatch == 1 && (!d->epi_f32 || t128_f32epi) &&
      d->M >= 64 && d->M <= (d->epi_f32 ? 2048 : t128_max_m) && d->N >= 64 && d->K >= 64 && d->K % 64 == 0 && p.vecA && p.vecB &&
      bytesA < (1ll << 31) && bytesB < (1ll << 31) &&
      (t128_all || (t128_tiles >= 32 && t128_tiles <= NUM_CU && (d->K <= 4096 || d->epi_f32)))) {
    p.tm = dxa_cdiv(d->M, 128);
    p.tn = dxa_cdiv(d->N, 128);
    // few tiles over a deep K: cut K so that every CU gets a workgroup (>= 8 K tiles of 64 per slice, <= 4 slices)
    static const bool t128_nosplit = getenv("DXA_GEMM_NO_SPLIT") != nullptr;
    int t_split = 1;
    // (120 tiles x K 1024 measured slower cut in two: 21.2 vs 18.7 us; up to half the CUs over a deep K it pays: the decoder's
    //  qkv / o_proj at 287 rows are 108 / 84 tiles of K 3584, each CU's feed rate being the bound)
    static const int t128_deepk = getenv("DXA_GEMM_T128_DEEPK") ? atoi(getenv("DXA_GEMM_T128_DEEPK")) : 2048;
    if (!t128_nosplit && (t128_tiles <= NUM_CU / 4 || (t128_tiles <= NUM_CU / 2 && t128_deepk > 0 && d->K >= t128_deepk)))
      t_split = (int)std::min<int64_t>(std::min<int64_t>(4, NUM_CU / t128_tiles), (d->K / 64) / 8);
    if (t_split >= 2) {
      SplitWs w;
      if (int rc = get_split_ws(st, &w)) return rc;
      p.split_s = t_split; p.ws = w.ws; p.flags = w.flags;
    } else {
      p.split_s = 1;
    }
    dim3 tgrid((unsigned)(p.tm * p.tn * p.split_s));
    // two stages (64 KiB): two workgroups share a CU and hide each other's LDS-DMA issue and barriers; four stages when a CU
    // gets one workgroup anyway
    static const int force_ns = getenv("DXA_GEMM_T128_NS") ? atoi(getenv("DXA_GEMM_T128_NS")) : 0;
    const int ns = force_ns ? force_ns : (p.tm * p.tn > NUM_CU ? 2 : 4);
#define LAUNCH_T128(TO_, NS_, WS_)                                                                                   \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_t128_kernel<TO_, NS_, WS_>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, NS_ * 32768);                            \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_nt_t128_kernel<TO_, NS_, WS_>), tgrid, dim3(256 + 64 * WS_), NS_ * 32768, st, p);       \
  } while (0)
    // wave-specialised build: DXA_GEMM_T128_WS = number of dedicated loader waves (0: every wave loads and computes, 4, 8)
    static const int t128_ws = getenv("DXA_GEMM_T128_WS") ? atoi(getenv("DXA_GEMM_T128_WS")) : 4;
    if (d->epi_f32) {
      static bool attr_f32 = false;
      if (!attr_f32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_t128_kernel<float, 4, 4, float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
        attr_f32 = true;
      }
      hipLaunchKernelGGL((gemm_nt_t128_kernel<float, 4, 4, float>), tgrid, dim3(512), 4 * 32768, st, p);
    } else
    if (ns == 2) { if (d->out_dtype == DXA_BF16) LAUNCH_T128(bf16_t, 2, 0); else LAUNCH_T128(float, 2, 0); }
    else if (t128_ws == 8) { if (d->out_dtype == DXA_BF16) LAUNCH_T128(bf16_t, 4, 8); else LAUNCH_T128(float, 4, 8); }
    else if (t128_ws == 4) { if (d->out_dtype == DXA_BF16) LAUNCH_T128(bf16_t, 4, 4); else LAUNCH_T128(float, 4, 4); }
    else { if (d->out_dtype == DXA_BF16) LAUNCH_T128(bf16_t, 4, 0); else LAUNCH_T128(float, 4, 0); }
#undef LAUNCH_T128
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  // ---- a second (A2, B2) segment (TN: dW over two micro-batches in one pass over C): the ping-pong kernel contracts both;
  //      every other path runs the two products one after the other, the second accumulating (mirror / sumsq from the second)
  if (d->K2 > 0) {
    DXA_CHECK_ARG(d->layout == DXA_TN && d->A2 && d->B2 && nbatch == 1, "dxa_gemm: A2 / B2 / K2 need an unbatched TN product");
    const int64_t bytesA2 = ((d->K2 - 1) * d->lda + d->M) * 2, bytesB2 = ((d->K2 - 1) * d->ldb + d->N) * 2;
    const bool seg_fast = !fast_off && d->in_dtype == DXA_BF16 && ks_ok && d->M >= 64 && d->N >= 64 &&
                          (int64_t)d->M * d->N >= 128 * 128 && bytesA < (1ll << 31) && bytesB < (1ll << 31) &&
                          bytesA2 < (1ll << 31) && bytesB2 < (1ll << 31) && aligned_to(d->A2, 16) && aligned_to(d->B2, 16);
    if (!seg_fast) {
      dxa_gemm_desc first = *d, second = *d;
      first.A2 = first.B2 = nullptr; first.K2 = 0; first.mirror = nullptr; first.sumsq = nullptr;
      second.A = d->A2; second.B = d->B2; second.K = d->K2; second.A2 = second.B2 = nullptr; second.K2 = 0;
      second.accumulate = 1; second.bias = nullptr; second.residual = nullptr;
      bool m1 = false, s1 = false;
      if (int rc = gemm_dispatch(&first, stream, &m1, &s1)) return rc;
      return gemm_dispatch(&second, stream, mirrored, summed);
    }
    p.A2 = (const char*)d->A2; p.B2 = (const char*)d->B2; p.K2 = d->K2;
  }
  if (!fast_off && d->in_dtype == DXA_BF16 && nbatch == 1 &&
      ((d->layout == DXA_NT && d->K >= 32 && d->K % 32 == 0 && p.vecA && p.vecB) || ks_ok) &&
      d->M >= 64 && d->N >= 64 && (int64_t)d->M * d->N >= 128 * 128 && bytesA < (1ll << 31) && bytesB < (1ll << 31)) {
    // 192-row tiles when they trim the padded row count by more than 8% (they run ~6% below the 256-row tile's rate)
    static const int force_ai = getenv("DXA_GEMM_RING_AI") ? atoi(getenv("DXA_GEMM_RING_AI")) : 0;
    const int64_t pad256 = (int64_t)dxa_cdiv(d->M, 256) * 256, pad192 = (int64_t)dxa_cdiv(d->M, 192) * 192;
    const int ai = ks_layout ? 4 : (force_ai ? force_ai : (pad192 * 27 < pad256 * 25 ? 3 : 4));
    p.tm = dxa_cdiv(d->M, ai * 64);
    p.tn = dxa_cdiv(d->N, 256);
    const int nt = p.tm * p.tn, nk_tot = (int)((d->K + d->K2 + 31) / 32);
    p.full = nt; p.tail_r = 0; p.split_s = 1;
    static const int group_m = getenv("DXA_GEMM_GROUP_M") ? atoi(getenv("DXA_GEMM_GROUP_M")) : 4;
    p.group_m = group_m;
    static const bool split_off = getenv("DXA_GEMM_NO_SPLIT") != nullptr;
    const int tail = nt % NUM_CU;
    if (!split_off && tail > 0) {
      // the last round would leave NUM_CU - tail CUs idle: cut its tiles along K (>= 16 slabs per piece)
      // (measured: each fp32 partial slot costs ~0.25 us of write-through traffic, so short K does not pay)
      static const int min_nk = getenv("DXA_SPLIT_MIN_NK") ? atoi(getenv("DXA_SPLIT_MIN_NK")) : 64;
      // (min_piece: K slabs of 32 per piece.  16 — rounds 1 - 6 — cut the tail tiles of the step's products and the fp32 head's dW products
      //  into up to 6 - 7 pieces of 512-deep K; at >= 32 slabs (1024-deep pieces, <= 3 of them at K = 3584) the step is 1.8 - 2.6 ms
      //  shorter, MemVLA's 2.8 ms, the 287-row request's gate/up 126 -> 119 us: a piece's fp32 partial costs its write-through and read-back
      //  whoever adds it up (profiles/r06_split_dist.txt).  24: -1.0 ms, 48: -1.3, 64: level, no split at all: +7.5 — profiles/r06_split_knobs.txt)
      static const int min_piece = getenv("DXA_SPLIT_MIN_PIECE") ? atoi(getenv("DXA_SPLIT_MIN_PIECE")) : 32;
      static const int max_split = getenv("DXA_SPLIT_MAX") ? atoi(getenv("DXA_SPLIT_MAX")) : 8;
      const int s = nk_tot >= min_nk ? std::min(std::min(NUM_CU / tail, max_split), nk_tot / min_piece) : 1;
      if (s >= 2) {
        SplitWs w;
        if (int rc = get_split_ws(st, &w)) return rc;
        p.full = nt - tail; p.tail_r = tail; p.split_s = s; p.ws = w.ws; p.flags = w.flags;
      }
    }
    dim3 fgrid((unsigned)(p.full + p.tail_r * p.split_s));
    constexpr int RING_LDS = 139264;   // max(4 x 32 KiB ring, 8 waves x 64 x 272 B epilogue slabs)
#define LAUNCH_RING(TO_, AI_, TE_)                                                                              \
  do {                                                                                                          \
    static bool attr_set = false;                                                                               \
    if (!attr_set) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_ring_kernel<TO_, AI_, TE_>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);                          \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL((gemm_nt_ring_kernel<TO_, AI_, TE_>), fgrid, dim3(512), RING_LDS, st, p);                \
  } while (0)
    // ---- ping-pong main loop: 256-row tiles, K % 64 == 0; the lean epilogue when it is made of whole 16-byte accesses
    static const bool pp_off = getenv("DXA_GEMM_NO_PP") != nullptr;
    static const bool pp3_on = !(getenv("DXA_GEMM_PP3") && atoi(getenv("DXA_GEMM_PP3")) == 0);
    const bool pp = ks_layout || (!pp_off && ai == 4 && d->K % 64 == 0);
    const bool lean = pp && lean_ok;
#define LAUNCH_PP(TO_, TE_, LEAN_, AKS_, BKS_)                                                                  \
  do {                                                                                                          \
    static bool attr_set = false;                                                                               \
    if (!attr_set) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<TO_, TE_, LEAN_, AKS_, BKS_>),    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);                          \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL((gemm_pp_kernel<TO_, TE_, LEAN_, AKS_, BKS_>), fgrid, dim3(512), RING_LDS, st, p);       \
  } while (0)
    if (lean) { p.mirror = (char*)d->mirror; *mirrored = d->mirror != nullptr; }
    if (lean) { p.sumsq = d->sumsq; *summed = d->sumsq != nullptr; }
    if (d->layout == DXA_NN) {          // dX = dY W: bf16 out (lean, or with the activation-gradient epilogue), fp32 out lean
      if (d->out_dtype == DXA_BF16) { if (lean) LAUNCH_PP(bf16_t, bf16_t, true, false, true); else LAUNCH_PP(bf16_t, bf16_t, false, false, true); }
      else LAUNCH_PP(float, bf16_t, true, false, true);
    } else if (d->layout == DXA_TN) {   // dW = dY^T X: fp32 (accumulating) or bf16 out, plain epilogue
      if (d->out_dtype == DXA_BF16) LAUNCH_PP(bf16_t, bf16_t, true, true, true); else LAUNCH_PP(float, bf16_t, true, true, true);
    } else if (d->fuse == DXA_FUSE_SWIGLU && ai == 4) {
#define LAUNCH_PPF(FUSE_)                                                                                       \
  do {                                                                                                          \
    static bool attr_set = false;                                                                               \
    if (!attr_set) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<bf16_t, bf16_t, true, false, false, FUSE_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);                          \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL((gemm_pp_kernel<bf16_t, bf16_t, true, false, false, FUSE_>), fgrid, dim3(512), RING_LDS, st, p); \
  } while (0)
      LAUNCH_PPF(1);
#undef LAUNCH_PPF
    } else if (d->fuse == DXA_FUSE_SWIGLU) {                   // 192-row tiles
      static bool attr_f = false;
      if (!attr_f) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp3_kernel<bf16_t, bf16_t, true, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);
        attr_f = true;
      }
      hipLaunchKernelGGL((gemm_pp3_kernel<bf16_t, bf16_t, true, 1>), fgrid, dim3(512), RING_LDS, st, p);
    } else if (pp && lean) {
      if (d->epi_f32) LAUNCH_PP(float, float, true, false, false);
      else if (d->out_dtype == DXA_BF16) LAUNCH_PP(bf16_t, bf16_t, true, false, false);
      else LAUNCH_PP(float, bf16_t, true, false, false);
    } else if (pp) {
      if (d->epi_f32) LAUNCH_PP(float, float, false, false, false);
      else if (d->out_dtype == DXA_BF16) LAUNCH_PP(bf16_t, bf16_t, false, false, false);
      else LAUNCH_PP(float, bf16_t, false, false, false);
    }
#undef LAUNCH_PP
    // ---- 192-row tiles on the ping-pong schedule (round 4): K % 64 == 0; DXA_GEMM_PP3=0: the ring kernel
    else if (ai == 3 && pp3_on && d->K % 64 == 0 && d->layout == DXA_NT) {
#define LAUNCH_PP3(TO_, TE_, LEAN_)                                                                             \
  do {                                                                                                          \
    static bool attr_set = false;                                                                               \
    if (!attr_set) {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp3_kernel<TO_, TE_, LEAN_>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);                          \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL((gemm_pp3_kernel<TO_, TE_, LEAN_>), fgrid, dim3(512), RING_LDS, st, p);                  \
  } while (0)
      if (d->epi_f32) { if (lean_ok) LAUNCH_PP3(float, float, true); else LAUNCH_PP3(float, float, false); }
      else if (lean_ok) { if (d->out_dtype == DXA_BF16) LAUNCH_PP3(bf16_t, bf16_t, true); else LAUNCH_PP3(float, bf16_t, true); }
      else { if (d->out_dtype == DXA_BF16) LAUNCH_PP3(bf16_t, bf16_t, false); else LAUNCH_PP3(float, bf16_t, false); }
#undef LAUNCH_PP3
    }
    else if (d->epi_f32) { if (ai == 3) LAUNCH_RING(float, 3, float); else LAUNCH_RING(float, 4, float); }
    else if (ai == 3) { if (d->out_dtype == DXA_BF16) LAUNCH_RING(bf16_t, 3, bf16_t); else LAUNCH_RING(float, 3, bf16_t); }
    else { if (d->out_dtype == DXA_BF16) LAUNCH_RING(bf16_t, 4, bf16_t); else LAUNCH_RING(float, 4, bf16_t); }
#undef LAUNCH_RING
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  DXA_CHECK_ARG(d->fuse == DXA_FUSE_NONE, "dxa_gemm: fuse is only implemented on the bf16 NT MFMA fast path (DXA_GEMM_NO_FAST set?)");
  DXA_CHECK_ARG(!d->epi_f32, "dxa_gemm: epi_f32 is only implemented on the bf16 NT fast path (K %% 32 == 0, M, N >= 64, "
                              "16-byte aligned rows, no batching)");
  // ---- few-row NN (dX of a linear on <= 8 tokens): a stream over W as it lies, K cut into slices, partials in the split scratch
  static const bool gemv_off = getenv("DXA_GEMM_NO_GEMV") != nullptr;
  if (!gemv_off && d->layout == DXA_NN && d->in_dtype == DXA_BF16 && nbatch == 1 && d->M <= GV_MAXM && d->K >= 64 &&
      d->N >= 64 && d->N % 8 == 0 && d->ldb % 8 == 0 && aligned_to(d->B, 16) && !d->bias && !d->residual && !d->aux_out &&
      !d->mulgrad && d->act == DXA_ACT_NONE && !d->accumulate && !d->epi_f32) {
    const int nbn = dxa_cdiv(d->N, 2048);
    int ksplit = (int)std::min<int64_t>(std::min<int64_t>(64, d->K / 32), dxa_cdiv(512, nbn));
    if (ksplit < 1) ksplit = 1;
    int kper = dxa_cdiv(d->K, ksplit);
    if (kper > 512) { kper = 512; ksplit = dxa_cdiv(d->K, kper); }
    if ((size_t)ksplit * d->M * d->N * sizeof(float) <= (size_t)NUM_CU * 256 * 256 * 4) {
      SplitWs w;
      if (int rc = get_split_ws(st, &w)) return rc;
      dim3 g1((unsigned)nbn, (unsigned)ksplit);
      const bf16_t* A_ = (const bf16_t*)d->A;
      const bf16_t* W_ = (const bf16_t*)d->B;
#define LAUNCH_GV(M_) hipLaunchKernelGGL((gemv_nn_stage1_k<M_>), g1, dim3(256), 0, st, A_, d->lda, W_, d->ldb, w.ws, (int)d->M, d->N, d->K, kper)
      if (d->M <= 1) LAUNCH_GV(1); else if (d->M <= 2) LAUNCH_GV(2); else if (d->M <= 4) LAUNCH_GV(4); else LAUNCH_GV(8);
#undef LAUNCH_GV
      dim3 g2((unsigned)dxa_cdiv(d->M * d->N, 256));
      if (d->out_dtype == DXA_BF16)
        hipLaunchKernelGGL(gemv_nn_stage2_k<bf16_t>, g2, dim3(256), 0, st, w.ws, (bf16_t*)d->C, d->ldc, (int)d->M, d->N, ksplit, d->alpha);
      else
        hipLaunchKernelGGL(gemv_nn_stage2_k<float>, g2, dim3(256), 0, st, w.ws, (float*)d->C, d->ldc, (int)d->M, d->N, ksplit, d->alpha);
      DXA_CHECK_LAUNCH();
      return DXA_OK;
    }
  }
  // ---- skinny bf16 path: M <= 64 (KV-cached decode, few-row products): a stream over the weights
  if (!skinny_off_g() && d->layout == DXA_NT && d->in_dtype == DXA_BF16 && nbatch == 1 && d->M <= 64 && d->K >= 64 &&
      d->K % 64 == 0 && p.vecA && p.vecB) {
    // few column tiles (<= 128) over a deep K: K cut across workgroups (>= 8 blocks of 64 k per range), as for the fp32 twin
    static const int skb_target = getenv("DXA_SKINNY_TARGET") ? atoi(getenv("DXA_SKINNY_TARGET")) : 256;
    static const int skb_maxtiles = getenv("DXA_SKINNY_BF16_MAXTILES") ? atoi(getenv("DXA_SKINNY_BF16_MAXTILES")) : NUM_CU / 2;
    static const int skb_unroll = getenv("DXA_SKINNY_BF16_UNROLL") ? atoi(getenv("DXA_SKINNY_BF16_UNROLL")) : 4;
    const int64_t skb_tiles = dxa_cdiv(d->N, 16);
    int skb_split = 1;
    if (skb_target > 0 && skb_tiles <= skb_maxtiles)
      skb_split = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, (d->K / 64) / 8), dxa_cdiv(skb_target, skb_tiles)));
    p.split_s = 1;
    if (skb_split >= 2) {
      SplitWs w;
      if (int rc = get_split_ws(st, &w)) return rc;
      p.split_s = skb_split; p.ws = w.ws; p.flags = w.flags;
    }
    dim3 sgrid((unsigned)(skb_tiles * p.split_s));
    const int mb = dxa_cdiv(d->M, 16);
#define LAUNCH_SK(MB_)                                                                                              \
  do {                                                                                                              \
    if (d->out_dtype == DXA_BF16) hipLaunchKernelGGL((gemm_skinny_bf16_kernel<MB_, bf16_t>), sgrid, dim3(512), 0, st, p); \
    else hipLaunchKernelGGL((gemm_skinny_bf16_kernel<MB_, float>), sgrid, dim3(512), 0, st, p);                   \
  } while (0)
    if (mb == 1 && skb_unroll != 2) {            // one row (the decode step): 4 blocks in flight; DXA_SKINNY_BF16_UNROLL=1: rounds 2-4
#define LAUNCH_SK1(U_)                                                                                              \
  do {                                                                                                              \
    if (d->out_dtype == DXA_BF16) hipLaunchKernelGGL((gemm_skinny_bf16_kernel<1, bf16_t, U_>), sgrid, dim3(512), 0, st, p); \
    else hipLaunchKernelGGL((gemm_skinny_bf16_kernel<1, float, U_>), sgrid, dim3(512), 0, st, p);                 \
  } while (0)
      if (skb_unroll == 1) LAUNCH_SK1(1); else LAUNCH_SK1(4);
#undef LAUNCH_SK1
    } else if (mb == 1) LAUNCH_SK(1); else if (mb == 2) LAUNCH_SK(2); else if (mb == 3) LAUNCH_SK(3); else LAUNCH_SK(4);
#undef LAUNCH_SK
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  // ---- skinny fp32 path: M <= 64 (DiT head at inference), weights streamed by N/16 workgroups of 8 K-splitting waves
  if (!skinny_off_g() && d->layout == DXA_NT && d->in_dtype == DXA_F32 && d->out_dtype == DXA_F32 && nbatch == 1 &&
      d->M <= 64 && d->K >= 64 && d->K % 64 == 0 && p.vecA && p.vecB) {
    // few column tiles (<= 128) over a deep K: cut K so that ~256 workgroups share the exact-fp32 MFMAs (32 cycles apiece: 64
    // workgroups walking K 4096 are MFMA-bound, DiT-L fc2 33 -> 19 us cut in four); each of a range's 8 waves keeps >= 1 block of
    // 64 k.  192+ tiles measured slower cut (15.1 vs 12.1 us)
    static const int sk_target = getenv("DXA_SKINNY_TARGET") ? atoi(getenv("DXA_SKINNY_TARGET")) : 256;
    static const int sk_minkb = getenv("DXA_SKINNY_MINKB") ? std::max(1, atoi(getenv("DXA_SKINNY_MINKB"))) : 8;
    const int64_t sk_tiles = dxa_cdiv(d->N, 16);
    int sk_split = 1;
    if (sk_target > 0 && sk_tiles <= NUM_CU / 2)
      sk_split = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, (d->K / 64) / sk_minkb), dxa_cdiv(sk_target, sk_tiles)));
    p.split_s = 1;
    if (sk_split >= 2) {
      SplitWs w;
      if (int rc = get_split_ws(st, &w)) return rc;
      p.split_s = sk_split; p.ws = w.ws; p.flags = w.flags;
    }
    dim3 sgrid((unsigned)(sk_tiles * p.split_s));
    switch (dxa_cdiv(d->M, 16)) {
      case 1: hipLaunchKernelGGL((gemm_skinny_f32_kernel<1>), sgrid, dim3(512), 0, st, p); break;
      case 2: hipLaunchKernelGGL((gemm_skinny_f32_kernel<2>), sgrid, dim3(512), 0, st, p); break;
      case 3: hipLaunchKernelGGL((gemm_skinny_f32_kernel<3>), sgrid, dim3(512), 0, st, p); break;
      default: hipLaunchKernelGGL((gemm_skinny_f32_kernel<4>), sgrid, dim3(512), 0, st, p); break;
    }
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  // generic kernel: 128x128 tiles, or 64x64 when 128-tiles would occupy less than ~one wave of the 256 CUs
  const bool small = (int64_t)p.tm * p.tn * nbatch < 192;
  if (small) {
    p.tm = dxa_cdiv(d->M, 64);
    p.tn = dxa_cdiv(d->N, 64);
  }
  dim3 grid((unsigned)(p.tm * p.tn), 1, (unsigned)nbatch);
  int rc;
  if (d->in_dtype == DXA_BF16) {
    if (small) rc = d->out_dtype == DXA_BF16 ? launch<bf16_t, bf16_t, 2>(p, d->layout, grid, st) : launch<bf16_t, float, 2>(p, d->layout, grid, st);
    else rc = d->out_dtype == DXA_BF16 ? launch<bf16_t, bf16_t, 4>(p, d->layout, grid, st) : launch<bf16_t, float, 4>(p, d->layout, grid, st);
  } else {
    rc = small ? launch<float, float, 2>(p, d->layout, grid, st) : launch<float, float, 4>(p, d->layout, grid, st);
  }
  if (rc != DXA_OK) return rc;
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
}  // namespace

namespace {
// split3 of the TRANSPOSE in one pass: src [R, C] fp32 -> dst [C, 3 Rp] bf16, dst[c, j Rp + r] = part j of src[r, c] (r < R; the
// padding columns R .. Rp - 1 are zero).  What transpose_k + split3_k did in two launches through an fp32 intermediate for the
// dX = dY W and dW = dY^T X products of the fp32 heads (they run as NT products of the transposed operands).  32 x 32 tiles
// through LDS (33-float rows: no bank conflicts), 256 threads = 32 x 8.
__global__ __launch_bounds__(256) void split3_t_k(const float* __restrict__ src, int64_t ld, bf16_t* __restrict__ dst, int64_t R,
                                                  int64_t C, int64_t Rp, int side) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? src[r * ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < Rp) {
      const float x = tile[tx][ty + 8 * i];
      const bf16_t hb = f2bf(x);
      const bf16_t lb = f2bf(x - bf2f(hb));
      bf16_t* d = dst + c * 3 * Rp + r;
      d[0] = hb;
      d[Rp] = side == 0 ? hb : lb;
      d[2 * Rp] = side == 0 ? lb : hb;
    }
  }
}
}  // namespace
extern "C" int dxa_split3_t(const float* src, int64_t ld, void* dst, int64_t R, int64_t C, int64_t Rp, int side,
                            dxa_stream_t stream) {
  DXA_CHECK_ARG(R >= 0 && C >= 0 && Rp >= R && ld >= C && (side == 0 || side == 1), "dxa_split3_t: bad arguments");
  if (Rp == 0 || C == 0) return DXA_OK;
  DXA_CHECK_ARG(src && dst, "dxa_split3_t: null buffer");
  const dim3 grid((unsigned)((C + 31) / 32), (unsigned)((Rp + 31) / 32));
  DXA_CHECK_ARG(grid.y <= 65535, "dxa_split3_t: too many rows");
  hipLaunchKernelGGL(split3_t_k, grid, dim3(256), 0, (hipStream_t)stream, src, ld, (bf16_t*)dst, R, C, Rp, side);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
namespace {
// both operand splits of one bf16x3 product in one launch: blocks [0, a.nblocks) write operand a, the rest operand b, each with the
// body of split3_k (grid-stride over its own blocks) or of split3_t_k (its 32 x 32 tiles flattened)
struct S3Op { const float* src; int64_t ld; bf16_t* dst; int64_t rows, cols, pad; int side, transposed; unsigned nblocks, tiles_x; };
__global__ __launch_bounds__(256) void split3_pair_k(const S3Op a, const S3Op b) {
  __shared__ float tile[32][33];
  const bool first = blockIdx.x < a.nblocks;
  const S3Op& o = first ? a : b;
  const unsigned bid = first ? blockIdx.x : blockIdx.x - a.nblocks;
  if (!o.transposed) {
    const int64_t c4 = o.cols >> 2;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < o.rows * c4; i += (int64_t)o.nblocks * 256) {
      const int64_t r = i / c4, c = (i - r * c4) * 4;
      float x[4];
      Vec<float, 4>::ld(x, o.src + r * o.ld + c);
      float hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = bf2f(f2bf(x[e]));
        lo[e] = x[e] - hi[e];
      }
      bf16_t* d = o.dst + r * 3 * o.cols + c;
      Vec<bf16_t, 4>::st(d, hi);
      Vec<bf16_t, 4>::st(d + o.cols, o.side == 0 ? hi : lo);
      Vec<bf16_t, 4>::st(d + 2 * o.cols, o.side == 0 ? lo : hi);
    }
    return;
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t R = o.rows, C_ = o.cols, Rp = o.pad;
  const int64_t r0 = (int64_t)(bid / o.tiles_x) * 32, c0 = (int64_t)(bid % o.tiles_x) * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C_) ? o.src[r * o.ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C_ && r < Rp) {
      const float x = tile[tx][ty + 8 * i];
      const bf16_t hb = f2bf(x);
      const bf16_t lb = f2bf(x - bf2f(hb));
      bf16_t* d = o.dst + c * 3 * Rp + r;
      d[0] = hb;
      d[Rp] = o.side == 0 ? hb : lb;
      d[2 * Rp] = o.side == 0 ? lb : hb;
    }
  }
}
int s3_fill(const dxa_split3_op* u, S3Op* o) {
  DXA_CHECK_ARG(u && u->rows >= 0 && u->cols >= 0 && (u->side == 0 || u->side == 1) && u->ld >= u->cols, "dxa_split3_pair: bad operand");
  o->src = u->src; o->ld = u->ld; o->dst = (bf16_t*)u->dst; o->rows = u->rows; o->cols = u->cols; o->pad = u->pad; o->side = u->side;
  o->transposed = u->transposed != 0; o->nblocks = 0; o->tiles_x = 1;
  if (u->rows == 0 || u->cols == 0) return DXA_OK;
  DXA_CHECK_ARG(u->src && u->dst, "dxa_split3_pair: null buffer");
  if (o->transposed) {
    DXA_CHECK_ARG(u->pad >= u->rows, "dxa_split3_pair: pad < rows");
    o->tiles_x = (unsigned)((u->cols + 31) / 32);
    const int64_t nb = (int64_t)o->tiles_x * ((u->pad + 31) / 32);
    DXA_CHECK_ARG(nb < (1ll << 30), "dxa_split3_pair: operand too large");
    o->nblocks = (unsigned)nb;
  } else {
    DXA_CHECK_ARG(u->cols % 4 == 0 && u->ld % 4 == 0 && (reinterpret_cast<uintptr_t>(u->src) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(u->dst) % 8) == 0,
                  "dxa_split3_pair: cols and ld must be multiples of 4 and the buffers 16-byte aligned");
    o->nblocks = (unsigned)dxa_grid1d(u->rows * (u->cols / 4), 256);
  }
  return DXA_OK;
}
}  // namespace
extern "C" int dxa_split3_pair(const dxa_split3_op* a, const dxa_split3_op* b, dxa_stream_t stream) {
  S3Op oa, ob;
  if (int rc = s3_fill(a, &oa)) return rc;
  if (int rc = s3_fill(b, &ob)) return rc;
  if (oa.nblocks + ob.nblocks == 0) return DXA_OK;
  hipLaunchKernelGGL(split3_pair_k, dim3(oa.nblocks + ob.nblocks), dim3(256), 0, (hipStream_t)stream, oa, ob);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
extern "C" int dxa_split3(const float* src, int64_t ld, void* dst, int64_t rows, int64_t cols, int side, dxa_stream_t stream) {
  DXA_CHECK_ARG(rows >= 0 && cols >= 0 && (side == 0 || side == 1), "dxa_split3: bad arguments");
  if (rows == 0 || cols == 0) return DXA_OK;
  DXA_CHECK_ARG(src && dst, "dxa_split3: null buffer");
  DXA_CHECK_ARG(cols % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(src) % 16) == 0 &&
                    (reinterpret_cast<uintptr_t>(dst) % 8) == 0,
                "dxa_split3: cols and ld must be multiples of 4 and the buffers 16-byte aligned");
  hipLaunchKernelGGL(split3_k, dim3(dxa_grid1d(rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, src, ld,
                     (bf16_t*)dst, rows, cols, side);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

#if defined(DXA_PP3_STAMPS)
extern "C" int dxa_gemm_debug_pp3_stamps(unsigned long long* out) {
  DXA_CHECK_HIP(hipDeviceSynchronize());
  DXA_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp3_stamps), sizeof(unsigned long long) * 4));
  return DXA_OK;
}
#endif
