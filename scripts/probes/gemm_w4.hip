// ROUND-4 EXPERIMENT, NOT PART OF THE PRODUCT (moved out of dexbotic_amd/csrc in round 5).  A 4-wave 128x128-per-wave kernel, parity-green
// and bit-identical to the ping-pong kernel, that lost its A/B on every layout (3-8 % slower: profiles/r04_w4_*.txt; MFMA busy 0.663
// against 0.718 — one wave per SIMD pays the full issue cost of its own LDS-DMA instructions).  To run it again: copy it back next to
// gemm.hip, add it to dexbotic_amd/build.py SOURCES and restore the DXA_GEMM_W4 hook in gemm_dispatch (git history, round 4).
// bf16 "w4" GEMM kernel for gfx950: 256x256 tile, K tile 64, FOUR waves (2 x 2), each wave owns 128 x 128 of the tile.
//
// Why a second large-tile kernel.  The 8-wave ping-pong kernel (gemm.hip, gemm_pp_kernel) feeds LDS
// with LDS-DMA (buffer_load ... lds); an LDS-DMA instruction costs the issuing wave 60-180 cycles of issue, eight of them per
// wave per K tile sit in the memory clusters, and those clusters — not the MFMA clusters — set the phase length (MFMA busy
// 0.55-0.67).  Its 128 x 64 wave tile also needs 0.75 fragment reads per MFMA.  Here:
//   * a wave owns 128 x 128 (16 accumulators of 32x32 = 256 registers, in the AGPR half of the 512-register file; one wave
//     per SIMD): a k-step of 16 is 4 A + 4 B fragments (8 x ds_read_b128) for 16 MFMAs = 0.5 reads per MFMA;
//   * the feed is REGISTER-staged: 16-byte buffer loads into 16 x 4 VGPRs one K tile ahead, ds_write_b128 into the other
//     LDS buffer — a load costs its wave one issue slot and a write ~13 cycles, all of which fit in the 32-cycle gaps
//     between the wave's own v_mfma_f32_32x32x16_bf16 (<= 5 single-issue instructions hide per gap on a one-wave SIMD);
//   * ONE barrier per K tile (2048 MFMA cycles), placed inside the last k-step's MFMAs.
// Every staging register is re-loaded (tile t+2) right after it has been written to LDS (tile t+1): each load has one whole
// K tile (~1 us) to land, the wave keeps 16 KiB in flight.  The instruction stream is the same for every K tile: past the
// last tile the loads re-read the last tile (K-contiguous operand: clamped tile index) or fall outside the buffer
// descriptor's range (K-strided operand: zeros), and the redundant LDS image they produce is never read.
//
// LDS (2 buffers x [A 32 KiB | B 32 KiB]):
//   K-contiguous operand (A of NT / NN, B of NT): [256 rows][128 B], 16-byte chunk c of row r at c ^ ((r >> 1) & 7) — the
//     16-lane groups of ds_read_b128 over a 32-row fragment hit 16 distinct 16-byte slots (same image as the ping-pong kernel,
//     the swizzle applied by the ds_write address here);
//   K-strided operand (B of NN, A and B of TN; staged as it lies, no transposed copy): two pieces [64 k][128 rows] with
//     256-byte rows, 64-byte chunk q of k-row t at q ^ (t & 3); a fragment (8 consecutive k of one row per lane) is two
//     ds_read_b64_tr_b16 (a 16-lane group reads a [4 k][16 rows] block, lane i receives the 4 k of row i).
// Operands are fed swapped (MFMA "A" = B-tile rows) so that a lane's 4 accumulator registers are 4 consecutive n of one m;
// the epilogue is the ping-pong kernel's (sk_epilogue_rows, 128 x 64 per call, twice per wave), as are the XCD-aware grouped
// tile order and the split-K hand-off of the tail tiles.
// Requirements (checked by the dispatch in gemm.hip): bf16 operands, K % 64 == 0 for a K-contiguous operand, 16-byte aligned
// rows, no batching, operands < 2 GiB, the lean epilogue (C = alpha acc + bias (+ R) (+ C), whole 16-byte accesses).
#include "gemm_common.h"

// Tuning instantiations (NT, bf16 out only; env DXA_GEMM_W4V picks one, results are garbage, timings are not): V bits
// 1 no LDS-DMA in the loop, 4 no fragment reads, 32 no s_barrier in the loop (keeps the operand statistics: a fair clock).

namespace {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

constexpr int W4_OP = 32768;          // one operand of one K tile = one ring slot
constexpr int W4_LDS = 5 * W4_OP;     // ring of five slots = all 160 KiB (the epilogue slabs reuse them)

// Split-K hand-off of a tail tile (same protocol and slot size as tile_split_exchange in gemm_common.h; 256 threads, 16
// accumulator blocks per lane): every piece but the last stores its fp32 partial with write-through (sc1) 16-byte stores and
// bumps the tile's arrival counter; the last piece waits for them and adds them in slice order.
__device__ __forceinline__ bool w4_split_exchange(const GemmP& p, f32x16_t (&acc)[2][4][2], int tid, int split_j, int split_s,
                                                  int tail_i) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (split_s > 1) {
    constexpr int SC1 = 16;
    constexpr uint32_t SLOT = 256 * 256 * 4;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)(NUM_CU_D * SLOT), 0x00020000);
    const uint32_t slot0 = (uint32_t)tail_i * (uint32_t)(split_s - 1) * SLOT + (uint32_t)tid * 16u;
    if (split_j < split_s - 1) {
      const uint32_t dst = slot0 + (uint32_t)split_j * SLOT;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4_t v = {acc[h][i][j][4 * q], acc[h][i][j][4 * q + 1], acc[h][i][j][4 * q + 2], acc[h][i][j][4 * q + 3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rW,
                                                     dst + ((((h * 4 + i) * 2 + j) * 4 + q) * 4096), 0, SC1);
            }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.flags + tail_i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    if (tid == 0) {
      while (__hip_atomic_load(p.flags + tail_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < split_s - 1)
        __builtin_amdgcn_s_sleep(4);
      __hip_atomic_store(p.flags + tail_i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int sj = 0; sj < split_s - 1; ++sj) {
      const uint32_t src = slot0 + (uint32_t)sj * SLOT;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x4_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              v[q] = __builtin_bit_cast(
                  f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rW, src + ((((h * 4 + i) * 2 + j) * 4 + q) * 4096), 0, SC1));
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[h][i][j][4 * q + c] += v[q][c];
          }
          __builtin_amdgcn_sched_barrier(0);   // 8 loads (32 VGPRs) in flight: the accumulators live in the AGPR half
        }
    }
  }
#endif
  return true;
}

template <typename TO, typename TE, bool A_KS, bool B_KS, int DXA_W4V, int SCH>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const GemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, lh = lane >> 5;

  // tiles [0, full): one workgroup each, block ids dealt round-robin to the 8 XCDs are remapped so every XCD walks a
  // contiguous run of tiles; tail tiles are cut along K into split_s workgroups each (the last one gathers)
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

  const uint32_t bytesA = (uint32_t)((A_KS ? (p.K - 1) * p.lda + p.M : (p.M - 1) * p.lda + p.K) * 2);
  const uint32_t bytesB = (uint32_t)((B_KS ? (p.K - 1) * p.ldb + p.N : (p.N - 1) * p.ldb + p.K) * 2);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.A), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, bytesB, 0x00020000);
  const int nk_tot = (int)((p.K + 63) >> 6);
  const int k_lo = split_j * nk_tot / split_s;
  const int nk = (split_j + 1) * nk_tot / split_s - k_lo;   // K tiles of this workgroup (>= 1)

  // ---- feed: LDS-DMA (buffer_load ... lds), 1 KiB per wave instruction written lane-linearly, so the LDS swizzle is applied
  //      to the per-lane SOURCE address.  A wave issues 8 + 8 of the 32 + 32 pieces of a K tile:
  //   K-contiguous: piece j of wave w = tile rows 64 w + 8 j .. + 7; lane -> (row + lane / 8, LDS chunk slot lane % 8, source
  //                 chunk = slot ^ ((row >> 1) & 7)): every row is still fetched as one whole 128-byte line
  //   K-strided   : piece j of wave w = k-rows 32 (w & 1) + 4 j .. + 3 of LDS piece (w >> 1); lane -> (k-row + lane / 16, LDS
  //                 16-byte slot s = lane % 16, source slot = s ^ ((k-row & 3) << 2) = tile rows 8 slot .. + 7 of the piece)
  // rows / columns outside the matrix: offset 0x80000000 (outside the descriptor's range: zeros); the K tile goes into the
  // scalar offset for a K-contiguous operand and into the VECTOR offset for a K-strided one (so that k-rows past K are
  // range-checked: the scalar offset is not)
  uint32_t voA[8], voB[8];
  {
    const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;
    const int kq = lane >> 4, sig = (lane & 15) ^ (kq << 2);
    if constexpr (A_KS) {
      const int ga = m0i + (wave >> 1) * 128 + 8 * sig;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        voA[j] = ga < (int)p.M ? (uint32_t)(k_lo * 64 + (wave & 1) * 32 + 4 * j + kq) * lda2 + (uint32_t)ga * 2u : 0x80000000u;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ga = m0i + 64 * wave + 8 * j + (lane >> 3);
        voA[j] = ga < (int)p.M ? (uint32_t)ga * lda2 + (uint32_t)(((lane & 7) ^ (kq | ((j & 1) << 2))) << 4) : 0x80000000u;
      }
    }
    if constexpr (B_KS) {
      const int gb = n0i + (wave >> 1) * 128 + 8 * sig;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        voB[j] = gb < (int)p.N ? (uint32_t)(k_lo * 64 + (wave & 1) * 32 + 4 * j + kq) * ldb2 + (uint32_t)gb * 2u : 0x80000000u;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int gb = n0i + 64 * wave + 8 * j + (lane >> 3);
        voB[j] = gb < (int)p.N ? (uint32_t)gb * ldb2 + (uint32_t)(((lane & 7) ^ (kq | ((j & 1) << 2))) << 4) : 0x80000000u;
      }
    }
  }
  const uint32_t ktileA = A_KS ? 64u * (uint32_t)p.lda * 2u : 0u, ktileB = B_KS ? 64u * (uint32_t)p.ldb * 2u : 0u;
  // piece j of this wave inside the operand's 32 KiB: K-contiguous rows 64 w + 8 j (128 B each); K-strided LDS piece w >> 1,
  // k-rows 32 (w & 1) + 4 j (256 B each)
  const int pieceA0 = A_KS ? (wave >> 1) * 16384 + (wave & 1) * 8192 : wave * 8192;
  const int pieceB0 = B_KS ? (wave >> 1) * 16384 + (wave & 1) * 8192 : wave * 8192;

  // ---- fragment read addresses
  //   K-contiguous: block b (32 rows) of k-step ks: row = 128 w_ + 32 b + l32, chunk 2 ks + lh at slot ^ sw:
  //                 (ya ^ (ks << 5)) + 4096 b with ya = row base | ((lh ^ sw) << 4)
  //   K-strided   : lane (i = lane % 16, half-group (lane >> 4) & 1, lh) passes the address of 4 consecutive tile rows (8 bytes)
  //                 of k-row 8 lh + (i >> 2) [+ 16 ks + 4 r as immediate] and receives row i's 4 k; the block's 64-byte chunk b
  //                 sits at b ^ (k-row & 3): (yb ^ (b << 6)) + 4096 ks + 1024 r
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int sw = (l32 >> 1) & 7;
  const int i16 = lane & 15;
  const uint32_t ks_lane = (uint32_t)((8 * lh + (i16 >> 2)) * 256 + (((i16 >> 2) & 3) << 6) + ((lane >> 4) & 1) * 32 + (i16 & 3) * 8);
  const uint32_t ya = lds0 + (A_KS ? (uint32_t)(wm * 16384) + ks_lane : (uint32_t)((wm * 128 + l32) * 128) | (uint32_t)((lh ^ sw) << 4));
  const uint32_t yb = lds0 + (B_KS ? (uint32_t)(wn * 16384) + ks_lane : (uint32_t)((wn * 128 + l32) * 128) | (uint32_t)((lh ^ sw) << 4));

  f32x16_t acc[2][4][2];                 // [64-column half][32-row block][32-column block of the half]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;
  u32x4_t fa[2][4], fb[2][4];            // fragments of two k-steps
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#define W4_PIN() __builtin_amdgcn_sched_barrier(0)
  // Ring of five 32 KiB slots: operand-tile u = 2 t + (0: A | 1: B) lives in slot u % 5.  The slot byte offsets of the current
  // tile (oA, oB), of the previous one (oAp, oBp) and of the next one (oA1, oB1) are wave-uniform integers rotated once per
  // tile: ONE loop body serves every tile (no five-fold unrolling), fragment addresses = per-lane base + slot offset.
  uint32_t oA = 0, oB = W4_OP, oAp = 3 * W4_OP, oBp = 4 * W4_OP, oA1 = 2 * W4_OP, oB1 = 3 * W4_OP;
  // per-lane fragment bases without the slot: K-contiguous indexed by k-step, K-strided by 32-row block
  uint32_t yak[4], ybk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    yak[i] = A_KS ? (ya ^ (uint32_t)(i << 6)) : (ya ^ (uint32_t)(i << 5));
    ybk[i] = B_KS ? (yb ^ (uint32_t)(i << 6)) : (yb ^ (uint32_t)(i << 5));
  }
  // LDS-DMA of piece j of K tile `tile` (relative to k_lo) into the slot at byte offset `off`
#define W4_DMA(rsrc, vo, so, ldsoff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(smem + (ldsoff)), 16, vo, so, 0, 0)
#define W4_DMA_A(j, off, tile)                                                                                       \
  do {                                                                                                               \
    if (!(DXA_W4V & 1)) {                                                                                            \
      if constexpr (A_KS) W4_DMA(rA, voA[j] + (uint32_t)(tile) * ktileA, 0, (off) + pieceA0 + (j) * 1024);           \
      else W4_DMA(rA, voA[j], (k_lo + min((tile), nk - 1)) * 128, (off) + pieceA0 + (j) * 1024);                     \
    }                                                                                                                \
  } while (0)
#define W4_DMA_B(j, off, tile)                                                                                       \
  do {                                                                                                               \
    if (!(DXA_W4V & 1)) {                                                                                            \
      if constexpr (B_KS) W4_DMA(rB, voB[j] + (uint32_t)(tile) * ktileB, 0, (off) + pieceB0 + (j) * 1024);           \
      else W4_DMA(rB, voB[j], (k_lo + min((tile), nk - 1)) * 128, (off) + pieceB0 + (j) * 1024);                     \
    }                                                                                                                \
  } while (0)
  // fragment reads (inline asm: the compiler must not order them against the DMA it tracks; waits are placed by hand)
#define W4_RD128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define W4_RDTR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define W4_TR(dst, addr, imm)                                                                                        \
  do {                                                                                                               \
    u32x2_t lo_ = {0u, 0u}, hi_ = {0u, 0u};                                                                          \
    W4_RDTR(lo_, addr, imm); W4_RDTR(hi_, addr, (imm) + 1024);                                                       \
    dst = (u32x4_t){lo_[0], lo_[1], hi_[0], hi_[1]};                                                                 \
  } while (0)
  // block b of k-step ks of the operand whose slot offset is `off` into register set F
#define W4_RDA(b, ks, off, F)                                                                                        \
  do {                                                                                                               \
    if (!(DXA_W4V & 4)) {                                                                                            \
      if constexpr (A_KS) W4_TR(fa[F][b], yak[b] + (off), (ks) * 4096);                                              \
      else W4_RD128(fa[F][b], yak[ks] + (off), (b) * 4096);                                                          \
    }                                                                                                                \
  } while (0)
#define W4_RDB(b, ks, off, F)                                                                                        \
  do {                                                                                                               \
    if (!(DXA_W4V & 4)) {                                                                                            \
      if constexpr (B_KS) W4_TR(fb[F][b], ybk[b] + (off), (ks) * 4096);                                              \
      else W4_RD128(fb[F][b], ybk[ks] + (off), (b) * 4096);                                                          \
    }                                                                                                                \
  } while (0)
  // read op n (0..7) of a k-step: A0 B0 A1 B1 A2 B2 A3 B3
#define W4_READ(n, ks, offa, offb, F) do { if ((n) & 1) W4_RDB((n) >> 1, ks, offb, F); else W4_RDA((n) >> 1, ks, offa, F); } while (0)
  // MFMA n (0..15) of a k-step from register set F: row block n & 3, column block n >> 2
#define W4_MFMA(n, F)                                                                                                \
  acc[(n) >> 3][(n) & 3][((n) >> 2) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                  \
      __builtin_bit_cast(bf16x8_t, fb[F][(n) >> 2]), __builtin_bit_cast(bf16x8_t, fa[F][(n) & 3]),                    \
      acc[(n) >> 3][(n) & 3][((n) >> 2) & 1], 0, 0, 0)
#define W4_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); W4_PIN(); } while (0)
  // One K tile t (operands in slots oA, oB).  Every k-step opens with lgkmcnt(0) (its fragments were requested >= 8 MFMAs
  // earlier) and reads the NEXT k-step's fragments behind its first 8 MFMAs; k-step 3: NB MFMAs, vmcnt(8), s_barrier (every
  // wave's reads of A(t), B(t) are retired and every wave's pieces of tile t+1 have landed), then the reads of k-step 0 of tile
  // t+1.  In front of that barrier a wave's outstanding pieces are ... B(t+1) A(t+2): vmcnt(8) leaves A(t+2) in flight.
  // Where the 16 DMA pieces of a tile go (SCH):
  //   0: right behind the barrier — B(t+2) into A(t)'s slot with the reads of k-step 3, A(t+3) into B(t)'s slot behind them
  //      (4 pieces) and in k-step 0 of tile t+1 (4 pieces): B runs one whole tile, A two tiles ahead of its barrier;
  //   1: away from the fragment reads (an LDS-DMA issued beside ds_reads costs its wave 100-185 cycles, alone 25-60): B(t+1)
  //      behind MFMAs 8..15 of k-step 0 into A(t-1)'s slot, A(t+2) behind MFMAs 8..15 of k-step 1 into B(t-1)'s slot: B runs
  //      ~0.7, A ~1.5 tiles ahead.
#define W4_KSTEP(ks, t)                                                                                              \
  do {                                                                                                               \
    W4_LGKM0();                                                                                                      \
    _Pragma("unroll") for (int n_ = 0; n_ < 16; ++n_) {                                                              \
      W4_MFMA(n_, (ks) & 1);                                                                                         \
      if (n_ < 8) W4_READ(n_, (ks) + 1, oA, oB, ((ks) + 1) & 1);                                                     \
      else if (SCH == 0 && (ks) == 0 && n_ < 12) W4_DMA_A(n_ - 4, oBp, (t) + 2);                                     \
      else if (SCH == 1 && (ks) == 0) W4_DMA_B(n_ - 8, oAp, (t) + 1);                                                \
      else if (SCH == 1 && (ks) == 1) W4_DMA_A(n_ - 8, oBp, (t) + 2);                                                \
      W4_PIN();                                                                                                      \
    }                                                                                                                \
  } while (0)
#define W4_TILE(t)                                                                                                   \
  do {                                                                                                               \
    W4_KSTEP(0, t);                                                                                                  \
    W4_KSTEP(1, t);                                                                                                  \
    W4_KSTEP(2, t);                                                                                                  \
    W4_LGKM0();                                                                                                      \
    _Pragma("unroll") for (int n_ = 0; n_ < 16; ++n_) {                                                              \
      W4_MFMA(n_, 1);                                                                                                \
      if (n_ == W4_NB - 1) {                                                                                         \
        W4_PIN();                                                                                                    \
        if (!(DXA_W4V & 64)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                        \
        if (!(DXA_W4V & 32)) __builtin_amdgcn_s_barrier();                                                           \
        asm volatile("" ::: "memory");                                                                               \
      }                                                                                                              \
      if (n_ >= W4_NB && n_ < W4_NB + 8) W4_READ(n_ - W4_NB, 0, oA1, oB1, 0);                                        \
      if (SCH == 0 && n_ >= W4_NB && n_ < W4_NB + 8) W4_DMA_B(n_ - W4_NB, oA, (t) + 2);                              \
      if (SCH == 0 && n_ >= W4_NB + 8) W4_DMA_A(n_ - W4_NB - 8, oB, (t) + 3);                                        \
      W4_PIN();                                                                                                      \
    }                                                                                                                \
  } while (0)
  constexpr int W4_NB = 4;               // MFMAs of k-step 3 ahead of the barrier (16 - NB - 8 = 4 A pieces fit behind it)
  static_assert(W4_NB == 4, "the piece split 12 + 4 assumes NB = 4");

  // ---- prologue.  SCH 0: A(0) B(0) A(1) B(1) and the first 4 pieces of A(2) requested (36 per wave), vmcnt(20): A(0), B(0)
  //      landed.  SCH 1: A(0) B(0) A(1) (24 per wave), vmcnt(8).  Then k-step 0 of tile 0 is read.
#pragma unroll
  for (int j = 0; j < 8; ++j) W4_DMA_A(j, 0 * W4_OP, 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) W4_DMA_B(j, 1 * W4_OP, 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) W4_DMA_A(j, 2 * W4_OP, 1);
  if constexpr (SCH == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) W4_DMA_B(j, 3 * W4_OP, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) W4_DMA_A(j, 4 * W4_OP, 2);
  }
  W4_PIN();
  if constexpr (SCH == 0) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  W4_PIN();
#pragma unroll
  for (int n = 0; n < 8; ++n) W4_READ(n, 0, oA, oB, 0);
  W4_PIN();
  // rotate the ring: tile t+1's slots become current; tile t+2 sits two slots further (mod 5)
#define W4_ROTATE()                                                                                                  \
  do {                                                                                                               \
    oAp = oA; oBp = oB; oA = oA1; oB = oB1;                                                                          \
    oA1 = oA + 2 * W4_OP; if (oA1 >= 5 * W4_OP) oA1 -= 5 * W4_OP;                                                    \
    oB1 = oB + 2 * W4_OP; if (oB1 >= 5 * W4_OP) oB1 -= 5 * W4_OP;                                                    \
  } while (0)
  for (int t = 0; t < nk; t += 2) {      // (two tiles per trip: with one, hipcc permutes the accumulator registers on the back edge)
    W4_TILE(t);
    W4_ROTATE();
    if (t + 1 < nk) {
      W4_TILE(t + 1);
      W4_ROTATE();
    }
  }
#undef W4_ROTATE
  W4_LGKM0();                            // the (unused) fragments of the tile past the end
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and its DMA pieces, before the slots become epilogue slabs
#undef W4_TILE
#undef W4_KSTEP
#undef W4_LGKM0
#undef W4_MFMA
#undef W4_READ
#undef W4_RDA
#undef W4_RDB
#undef W4_TR
#undef W4_RDTR
#undef W4_RD128
#undef W4_DMA_A
#undef W4_DMA_B
#undef W4_DMA
#undef W4_PIN
  // every wave is past its last fragment read and its last DMA piece before the slots become epilogue slabs
  __syncthreads();
  if (!w4_split_exchange(p, acc, tid, split_j, split_s, tail_i)) return;
  __builtin_amdgcn_sched_barrier(0);
  float ssq = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
    ssq += sk_epilogue_rows<TO, TE>(p, acc[h], smem + wave * 4096, lane, wm, 2 * wn + h, m0i, n0i);
  if (p.sumsq != nullptr) {                    // uniform: this tile's share of sum(g^2), folded in a fixed order
    float* red = reinterpret_cast<float*>(smem + 4 * 4096);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssq += __shfl_xor(ssq, o, 64);
    if (lane == 0) red[wave] = ssq;
    __syncthreads();
    if (tid == 0) p.sumsq[bid] = (red[0] + red[1]) + (red[2] + red[3]);
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <typename TO, typename TE, bool A_KS, bool B_KS, int V = 0, int SCH = 0>
int w4_launch_one(const GemmP& p, dim3 grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<TO, TE, A_KS, B_KS, V, SCH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_w4_kernel<TO, TE, A_KS, B_KS, V, SCH>), grid, dim3(256), W4_LDS, st, p);
  return 0;
}

}  // namespace

namespace dxa_gemm_detail {
// layout: DXA_NT / DXA_NN / DXA_TN; out_f32: fp32 C (else bf16); epi_f32: fp32 bias / residual (NT, fp32 C only).
// The caller has filled p (tile counts for 256 x 256 tiles, full / tail_r / split_s / ws / flags, group_m, mirror, sumsq).
int gemm_w4_launch(const GemmP& p, int layout, bool out_f32, bool epi_f32, hipStream_t st) {
  const dim3 grid((unsigned)(p.full + p.tail_r * p.split_s));
  if (layout == DXA_NT) {
    if (epi_f32) return w4_launch_one<float, float, false, false>(p, grid, st);
    if (!out_f32) {                      // tuning instantiations
      const char* e = getenv("DXA_GEMM_W4V");
      const int v = e ? atoi(e) : 0;
      switch (v) {
        case 0: break;
        case 1: return w4_launch_one<bf16_t, bf16_t, false, false, 1>(p, grid, st);
        case 4: return w4_launch_one<bf16_t, bf16_t, false, false, 4>(p, grid, st);
        case 32: return w4_launch_one<bf16_t, bf16_t, false, false, 32>(p, grid, st);
        case 64: return w4_launch_one<bf16_t, bf16_t, false, false, 64>(p, grid, st);
        case 96: return w4_launch_one<bf16_t, bf16_t, false, false, 96>(p, grid, st);
        case 100: return w4_launch_one<bf16_t, bf16_t, false, false, 0, 1>(p, grid, st);      // schedule 1
        case 132: return w4_launch_one<bf16_t, bf16_t, false, false, 32, 1>(p, grid, st);     // schedule 1, no barrier
        default: break;
      }
    }
    return out_f32 ? w4_launch_one<float, bf16_t, false, false>(p, grid, st) : w4_launch_one<bf16_t, bf16_t, false, false>(p, grid, st);
  }
  if (layout == DXA_NN)
    return out_f32 ? w4_launch_one<float, bf16_t, false, true>(p, grid, st) : w4_launch_one<bf16_t, bf16_t, false, true>(p, grid, st);
  return out_f32 ? w4_launch_one<float, bf16_t, true, true>(p, grid, st) : w4_launch_one<bf16_t, bf16_t, true, true>(p, grid, st);
}
}  // namespace dxa_gemm_detail
