// Attention for the CogACT path (see include/dexbotic_amd.h :: dxa_attn_*).
//
// FORWARD, bf16, head_dim 64/128 — fused flash kernel:
//   workgroup = 4 waves = 64 query rows (16 per wave); K/V tiles of 64 keys staged in LDS
//   (K row-major [key][d], V transposed [d][key] through an in-register 4x8 transpose), both XOR
//   swizzled so the ds_read_b128 / ds_read_b64 fragment reads are bank-conflict free.
//   "Swapped" products keep everything lane-local:
//     S^T = K Q^T  -> lane (q = lane&15, g = lane>>4) holds 16 scores of ONE query (keys 16n+4g+r):
//                     the row max / row sum are 15 local ops + two wavefront shuffles (xor 16, 32);
//     O^T = V^T P^T-> the bf16 P fragment of the PV MFMA is exactly the lane's own registers (a
//                     k-slot permutation shared with the V^T fragment), so P never touches LDS;
//                     the running rescale of O is one scalar per lane.
//   fp32 online softmax, fp32 accumulation, saves log-sum-exp for the backward.
// FORWARD, generic (any dtype / head_dim): one wavefront per query row, fp32 math, scores in LDS.
// BACKWARD: recompute P = exp(scale*QK^T - lse) and form dQ/dK/dV with five batched MFMA GEMMs
//   (dxa_gemm: NT scores, NT dP, NN dQ, TN dK, TN dV; fp32 scores/dP) plus three HBM-bound kernels.
//   With 288 GB of HBM the [B,H,Sq,Sk] probability slab of one layer (74 MB at B16/S287) is cheap;
//   the GQA group is folded into the GEMM row index so K/V gradients need no separate reduction.
#include "common.h"

namespace {

struct AttnP {
  int B, Hq, Hkv, Sq, Sk, D, causal;
  float scale;
  const char* q; int64_t q_sb, q_sh, q_ss;
  const char* k; int64_t k_sb, k_sh, k_ss;
  const char* v; int64_t v_sb, v_sh, v_ss;
  char* o; int64_t o_sb, o_sh, o_ss;
  float* lse;
  const int32_t* kv_start;
  const int32_t* kv_end;
  const int32_t* q_limit;      // [B,Sq] or NULL: query i sees keys j < q_limit[b,i] (block-prefix masks, pi0)
  const uint8_t* key_valid;    // [B,Sk] or NULL: key j takes part at all (padding / missing camera)
  const void* drop_mask;       // [B,Hq,Sq,Sk] or NULL: attention dropout, P <- P * mask (0 or 1 / (1 - p)) after the softmax
  // flash forward over key ranges (few queries against a long cache): grid z = B * nsplit, workgroup (b, sp) sees keys
  // [sp * split_len, (sp + 1) * split_len) only and writes the softmax-normalised partial of its range + its log-sum-exp to
  // o / lse AS IF the batch were B * nsplit (strides of a [B * nsplit, Hq, Sq, D] buffer); attn_split_combine_k folds them
  int nsplit, split_len;
};

// ------------------------------------------------------------------------------------ generic forward
template <typename T, int VEC>
__global__ __launch_bounds__(256) void attn_fwd_generic_k(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* qs = sm + (size_t)wave * (p.D + p.Sk);
  float* ps = qs + p.D;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  const int64_t total = (int64_t)p.B * p.Hq * p.Sq;
  if (row >= total) return;
  const int i = (int)(row % p.Sq);
  const int h = (int)((row / p.Sq) % p.Hq);
  const int b = (int)(row / ((int64_t)p.Sq * p.Hq));
  const int hk = h / (p.Hq / p.Hkv);
  const T* q = reinterpret_cast<const T*>(p.q) + b * p.q_sb + h * p.q_sh + (int64_t)i * p.q_ss;
  const T* k = reinterpret_cast<const T*>(p.k) + b * p.k_sb + hk * p.k_sh;
  const T* v = reinterpret_cast<const T*>(p.v) + b * p.v_sb + hk * p.v_sh;
  T* o = reinterpret_cast<T*>(p.o) + b * p.o_sb + h * p.o_sh + (int64_t)i * p.o_ss;
  for (int d = lane; d < p.D; d += 64) qs[d] = ldf<T>(q + d);
  int j0 = p.kv_start ? p.kv_start[b] : 0;
  int j1 = p.kv_end ? p.kv_end[b] : p.Sk;
  if (p.causal) j1 = min(j1, i + (p.Sk - p.Sq) + 1);
  if (p.q_limit) j1 = min(j1, p.q_limit[(int64_t)b * p.Sq + i]);
  const uint8_t* kvld = p.key_valid ? p.key_valid + (int64_t)b * p.Sk : nullptr;
  j0 = max(j0, 0);
  j1 = min(j1, p.Sk);
  // wave-private LDS: same-wave write->read needs only the LDS counter, __syncthreads not required,
  // but lanes read values written by other lanes: make it visible
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  float mx = -INFINITY;
  for (int j = j0 + lane; j < j1; j += 64) {
    const T* kr = k + (int64_t)j * p.k_ss;
    float acc = 0.f;
    for (int d = 0; d < p.D; d += VEC) {
      float kv[VEC];
      Vec<T, VEC>::ld(kv, kr + d);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc += qs[d + e] * kv[e];
    }
    acc = (kvld && !kvld[j]) ? -INFINITY : acc * p.scale;
    ps[j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = j0 + lane; j < j1; j += 64) {
    const float e = mx == -INFINITY ? 0.f : expf(ps[j] - mx);
    ps[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const bool any = sum > 0.f;
  const float inv = any ? 1.f / sum : 0.f;
  // attention dropout (torch SDPA: dropout on the attention weights AFTER the softmax; the normaliser is untouched)
  const T* dm = p.drop_mask ? reinterpret_cast<const T*>(p.drop_mask) + row * p.Sk : nullptr;
  for (int j = j0 + lane; j < j1; j += 64) {
    float pj = rnd<T>(ps[j] * inv);
    if (dm) pj = rnd<T>(pj * ldf<T>(dm + j));
    ps[j] = pj;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < p.D; d += 64) {
    float acc = 0.f;
    for (int j = j0; j < j1; ++j) acc += ps[j] * ldf<T>(v + (int64_t)j * p.v_ss + d);
    stf<T>(o + d, acc);
  }
  if (lane == 0 && p.lse) p.lse[row] = any ? mx + logf(sum) : 0.f;
}

// ------------------------------------------------------------------------------- small fp32 forward
// The fp32 attentions of the diffusion heads are tiny (DiT: 17-18 tokens over 16 heads; MemVLA's perceptual cross attention:
// 17 queries over 256 keys) and run hundreds of times per sampled action; one wave per query row (attn_fwd_generic_k) walks
// the keys and then the values serially — 60 us a call at those sizes.  Here a workgroup owns 16 queries of one (b, head):
// S = Q K^T by exact fp32 MFMA (v_mfma_f32_16x16x4_f32; the four waves take the 16-key tiles round robin) into LDS, a masked
// row softmax in LDS (same formulas as the generic kernel), O = P V by MFMA again with the waves taking the 16-column tiles
// of the head.  K rows and Q rows are read as 16-byte vectors (k permutation shared by both operands, as in gemm.hip's
// mma_step<float>); V is d-contiguous while MFMA wants the contraction index per lane, so V is read as dwords, 64-byte runs
// per 16 lanes.  Requirements: fp32, D % 16 == 0 (instantiated 32/64/96/128), no dropout mask, 16-byte aligned q/k/o rows.
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_small_f32_k(const AttnP p, const int srow) {
  extern __shared__ __attribute__((aligned(16))) float S[];             // [16][srow], srow = 16 * ceil(Sk / 16) + 4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
  const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (p.Hq / p.Hkv);
  const float* qb = reinterpret_cast<const float*>(p.q) + b * p.q_sb + h * p.q_sh;
  const float* kb = reinterpret_cast<const float*>(p.k) + b * p.k_sb + hk * p.k_sh;
  const float* vb = reinterpret_cast<const float*>(p.v) + b * p.v_sb + hk * p.v_sh;
  float* ob = reinterpret_cast<float*>(p.o) + b * p.o_sb + h * p.o_sh;
  const int nkt = (p.Sk + 15) / 16;
  {
    const float* qr = qb + (int64_t)min(q0 + l16, p.Sq - 1) * p.q_ss + 4 * lg;
    float4 qf[D / 16];
#pragma unroll
    for (int t = 0; t < D / 16; ++t) qf[t] = *reinterpret_cast<const float4*>(qr + 16 * t);
    // four key tiles a round: their 4 x D/16 loads are all in flight before the first MFMA
    for (int kt0 = wave; kt0 < nkt; kt0 += 16) {
      float4 kf[4][D / 16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* kr = kb + (int64_t)min((kt0 + 4 * u) * 16 + l16, p.Sk - 1) * p.k_ss + 4 * lg;
#pragma unroll
        for (int t = 0; t < D / 16; ++t) kf[u][t] = *reinterpret_cast<const float4*>(kr + 16 * t);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kt = kt0 + 4 * u;
        if (kt < nkt) {
          f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < D / 16; ++t) {
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][t].x, qf[t].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][t].y, qf[t].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][t].z, qf[t].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][t].w, qf[t].w, s, 0, 0, 0);
          }
          // lane holds S[query l16][key 16 kt + 4 lg + {0..3}]
          *reinterpret_cast<float4*>(&S[l16 * srow + kt * 16 + 4 * lg]) = make_float4(s[0], s[1], s[2], s[3]);
        }
      }
    }
  }
  __syncthreads();
  const int npad = nkt * 16;
  for (int r = wave * 4; r < wave * 4 + 4; ++r) {            // wave w: the softmax of query rows 4w .. 4w+3
    const int i = q0 + r;
    float* ps = S + r * srow;
    if (i >= p.Sq) {
      for (int j = lane; j < npad; j += 64) ps[j] = 0.f;
      continue;
    }
    int j0 = p.kv_start ? p.kv_start[b] : 0;
    int j1 = p.kv_end ? p.kv_end[b] : p.Sk;
    if (p.causal) j1 = min(j1, i + (p.Sk - p.Sq) + 1);
    if (p.q_limit) j1 = min(j1, p.q_limit[(int64_t)b * p.Sq + i]);
    const uint8_t* kvld = p.key_valid ? p.key_valid + (int64_t)b * p.Sk : nullptr;
    j0 = max(j0, 0);
    j1 = min(j1, p.Sk);
    float mx = -INFINITY;
    for (int j = j0 + lane; j < j1; j += 64) {
      const float a = (kvld && !kvld[j]) ? -INFINITY : ps[j] * p.scale;
      ps[j] = a;
      mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = j0 + lane; j < j1; j += 64) {
      const float e = mx == -INFINITY ? 0.f : expf(ps[j] - mx);
      ps[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const bool any = sum > 0.f;
    const float inv = any ? 1.f / sum : 0.f;
    for (int j = lane; j < npad; j += 64) ps[j] = (j >= j0 && j < j1) ? ps[j] * inv : 0.f;
    if (lane == 0 && p.lse) p.lse[((int64_t)b * p.Hq + h) * p.Sq + i] = any ? mx + logf(sum) : 0.f;
  }
  __syncthreads();
  for (int dt = wave; dt < D / 16; dt += 4) {                // wave w: head columns 16 dt .. 16 dt + 15
    const float* vc = vb + dt * 16 + l16;
    f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < nkt; t0 += 4) {                    // four 16-key blocks a round: 16 value loads in flight
      float4 pf[4];
      float vv[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = min(t0 + u, nkt - 1);
        const int kbase = 16 * t + 4 * lg;                   // this lane's four keys of the 16-key block
        pf[u] = t0 + u < nkt ? *reinterpret_cast<const float4*>(&S[l16 * srow + 16 * t + 4 * lg]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[u][e] = kbase + e < p.Sk ? vc[(int64_t)(kbase + e) * p.v_ss] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[u][0], pf[u].x, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[u][1], pf[u].y, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[u][2], pf[u].z, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[u][3], pf[u].w, o, 0, 0, 0);
      }
    }
    // lane holds O[query l16][column 16 dt + 4 lg + {0..3}]
    if (q0 + l16 < p.Sq)
      *reinterpret_cast<float4*>(ob + (int64_t)(q0 + l16) * p.o_ss + dt * 16 + 4 * lg) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
constexpr float LOG2E = 1.4426950408889634f;

// -------------------------------------------------------------------------------------- flash forward
// NW waves of 16 queries each share one staged K/V tile: 4 (64 queries per workgroup) or 8 (128 queries: the tile's staging is
// amortised over twice the MFMA work and every SIMD holds two waves that hide each other's LDS reads and softmax — head_dim 256,
// whose 64 accumulator + 32 query registers leave room for only one 4-wave workgroup per CU)
// DV < D (round 5): a head width the tiles are not built for (SigLIP-So400m: 72) runs on the next tile width with only
// its DV real columns loaded (the chunks past them are zeros in registers / LDS, never fetched), the k-steps and output
// blocks past DV skipped (72: 3 of 4 k-steps of QK^T, 5 of 8 output blocks of PV) and only DV columns stored — no padded
// copies of q / k / v / o in HBM.  DV % 8 == 0.
template <int D, int NW, int DV = D>
__global__ __launch_bounds__(64 * NW) void attn_fwd_flash_k(const AttnP p) {
  static_assert(DV % 8 == 0 && DV <= D && DV > D / 2, "valid head width: a multiple of 8 in (D/2, D]");
  constexpr int DSN = (DV + 31) / 32;   // k-steps of 32 that hold real columns
  constexpr int DIN = (DV + 15) / 16;   // 16-column output blocks that hold real columns
  constexpr int NT = 64 * NW;
  constexpr int NCH = D / 8;          // 16-B chunks per K row
  constexpr int KROW = D * 2;         // bytes per K row
  constexpr int KT_BYTES = 64 * KROW; // K tile
  constexpr int VT_BYTES = D * 128;   // V^T tile: D rows x 64 keys x 2 B
  __shared__ __attribute__((aligned(16))) char smem[KT_BYTES + VT_BYTES];
  char* Ks = smem;
  char* Vs = smem + KT_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lg = lane >> 4;
  const int bz = blockIdx.z, h = blockIdx.y;
  const int b = p.nsplit > 1 ? bz / p.nsplit : bz;             // key-range splits: (batch, range) on grid z
  const int sp = bz - b * p.nsplit;
  const int hk = h / (p.Hq / p.Hkv);
  // (heaviest causal query tile first measured 1 % SLOWER here — thousands of short workgroups balance themselves:
  //  profiles/r05_attn_order.txt; the dK/dV kernel's grid order is what matters)
  const int q0 = blockIdx.x * (16 * NW);
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(p.q) + b * p.q_sb + h * p.q_sh;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(p.k) + b * p.k_sb + hk * p.k_sh;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(p.v) + b * p.v_sb + hk * p.v_sh;

  // Q fragments: MFMA second operand, lane (q = l16, k-group lg) holds d = 32*ds + 8*lg .. +7
  const int qi = q0 + wave * 16 + l16;
  uint4 qf[DSN];
#pragma unroll
  for (int ds = 0; ds < DSN; ++ds) {
    qf[ds] = make_uint4(0, 0, 0, 0);
    if (qi < p.Sq && 32 * ds + 8 * lg < DV) qf[ds] = *reinterpret_cast<const uint4*>(qb + (int64_t)qi * p.q_ss + 32 * ds + 8 * lg);
  }
  f32x4_t oacc[DIN];
#pragma unroll
  for (int i = 0; i < DIN; ++i) oacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float sc2 = p.scale * LOG2E;   // softmax in base 2: exp(x) = 2^(x log2 e), one v_exp_f32 per score

  int j_lo = p.kv_start ? p.kv_start[b] : 0;
  int j_hi = p.kv_end ? p.kv_end[b] : p.Sk;
  j_lo = max(j_lo, 0);
  j_hi = min(j_hi, p.Sk);
  if (p.nsplit > 1) {
    j_lo = max(j_lo, sp * p.split_len);
    j_hi = min(j_hi, (sp + 1) * p.split_len);
  }
  const int coff = p.Sk - p.Sq;
  int blk_hi = j_hi;  // exclusive key bound for the whole workgroup
  if (p.causal) blk_hi = min(blk_hi, min(q0 + 16 * NW - 1, p.Sq - 1) + coff + 1);
  int my_hi = p.causal ? min(j_hi, qi + coff + 1) : j_hi;  // exclusive bound for this lane's query
  if (p.q_limit) my_hi = min(my_hi, qi < p.Sq ? p.q_limit[(int64_t)b * p.Sq + qi] : 0);   // block-prefix mask (pi0)
  const uint8_t* kvld = p.key_valid ? p.key_valid + (int64_t)b * p.Sk : nullptr;
  const int t_lo = j_lo / 64, t_hi = (blk_hi + 63) / 64;

  // K / V tiles travel global -> registers -> LDS; the NEXT tile's global loads are issued right after this tile has been
  // written to LDS, so they are in flight under this tile's MFMAs and softmax instead of in front of them
  constexpr int RPI = NT / NCH;            // K rows per pass
  constexpr int KPT = 64 / RPI;            // K chunks (16 B) per thread
  constexpr int VWI = (8 * (D / 4) + NT - 1) / NT;   // V work items (8 keys x 4 d) per thread
  uint4 kreg[KPT];
  uint2 vreg[VWI][8];
  auto load_tile = [&](int kt) {
    const int key0 = kt * 64;
    const int c = tid % NCH;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int key = key0 + tid / NCH + RPI * i;
      kreg[i] = make_uint4(0, 0, 0, 0);
      if (key < p.Sk && c * 8 < DV) kreg[i] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * p.k_ss + c * 8);
    }
#pragma unroll
    for (int w = 0; w < VWI; ++w) {
      const int wi = tid + NT * w;
      const int kc = wi & 7, dg = wi >> 3;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = key0 + kc * 8 + e;
        vreg[w][e] = make_uint2(0, 0);
        if (wi < 8 * (D / 4) && dg * 4 < DV && key < p.Sk) vreg[w][e] = *reinterpret_cast<const uint2*>(vb + (int64_t)key * p.v_ss + dg * 4);
      }
    }
  };
  if (t_lo < t_hi) load_tile(t_lo);
  for (int kt = t_lo; kt < t_hi; ++kt) {
    const int key0 = kt * 64;
    __syncthreads();  // previous tile fully consumed
    // ---- stage K tile: [64][D] row-major, chunk ^= row & (NCH-1)
    {
      const int c = tid % NCH;
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const int r = tid / NCH + RPI * i;
        *reinterpret_cast<uint4*>(Ks + r * KROW + ((c ^ (r & (NCH - 1))) << 4)) = kreg[i];
      }
    }
    // ---- stage V^T tile: work item (kc = key chunk of 8, dg = group of 4 d) transposes 8x4 -> 4x8
#pragma unroll
    for (int w = 0; w < VWI; ++w) {
      const int wi = tid + NT * w;
      if (wi >= 8 * (D / 4)) break;
      const int kc = wi & 7, dg = wi >> 3;
      const uint2* vv = vreg[w];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d = dg * 4 + qd;
        uint32_t w4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const uint32_t lo = (qd & 2) ? vv[2 * m].y : vv[2 * m].x;
          const uint32_t hi = (qd & 2) ? vv[2 * m + 1].y : vv[2 * m + 1].x;
          const uint32_t lo16 = (qd & 1) ? (lo >> 16) : (lo & 0xffffu);
          const uint32_t hi16 = (qd & 1) ? (hi >> 16) : (hi & 0xffffu);
          w4[m] = lo16 | (hi16 << 16);
        }
        *reinterpret_cast<uint4*>(Vs + d * 128 + ((kc ^ ((d >> 1) & 7)) << 4)) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
      }
    }
    __syncthreads();
    if (kt + 1 < t_hi) load_tile(kt + 1);

    // ---- S^T = K Q^T : sacc[n][r] = score(query l16, key key0 + 16n + 4lg + r)
    f32x4_t sacc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      sacc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const int r = n * 16 + l16;
#pragma unroll
      for (int ds = 0; ds < DSN; ++ds) {
        const int c = 4 * ds + lg;
        const uint4 kf = *reinterpret_cast<const uint4*>(Ks + r * KROW + ((c ^ (r & (NCH - 1))) << 4));
        sacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf),
                                                          __builtin_bit_cast(bf16x8_t, qf[ds]), sacc[n], 0, 0, 0);
      }
    }
    // ---- online softmax for this lane's query
    // key validity of the 64 keys of this tile as one 64-bit wave-uniform mask: lane j reads key_valid[key0 + j] (one coalesced
    // 64-byte access) and the ballot spreads it, instead of 16 scattered byte loads per lane
    unsigned long long kmask = ~0ull;
    if (kvld) {
      const int kj = key0 + lane;
      kmask = __ballot(kj < p.Sk && kvld[kj] != 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + 16 * n + 4 * lg + r;
        const bool vis = key >= j_lo && key < my_hi && ((kmask >> (16 * n + 4 * lg + r)) & 1ull);
        const float s = vis ? sacc[n][r] * sc2 : -INFINITY;          // scores in log2 units
        sacc[n][r] = s;
        tmax = fmaxf(tmax, s);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float alpha = 1.f;
    if (m_new > -INFINITY) alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // m_run = -inf -> 0
    float psum = 0.f;
    uint32_t pk[8];  // bf16 pairs: pk[2*kb2 + ...] see below
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e[r] = (m_new > -INFINITY) ? __builtin_amdgcn_exp2f(sacc[n][r] - m_new) : 0.f;  // 2^-inf = 0 for masked keys
        psum += e[r];
      }
      pk[2 * n] = pack_bf16(e[0], e[1]);
      pk[2 * n + 1] = pack_bf16(e[2], e[3]);
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DIN; ++i) {
      oacc[i][0] *= alpha; oacc[i][1] *= alpha; oacc[i][2] *= alpha; oacc[i][3] *= alpha;
    }
    // ---- O^T += V^T P^T.  k-slot (lg, e) of 32-key block kb2 <-> key 32*kb2 + 16*(e>>2) + 4*lg + (e&3):
    //      P fragment = {sacc[2kb2][0..3], sacc[2kb2+1][0..3]} of this lane (already in pk);
    //      V^T fragment of row d: two 8-byte reads at keys 32kb2+4lg and 32kb2+16+4lg.
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      const uint4 pf = make_uint4(pk[4 * kb2], pk[4 * kb2 + 1], pk[4 * kb2 + 2], pk[4 * kb2 + 3]);
#pragma unroll
      for (int di = 0; di < DIN; ++di) {
        const int d = di * 16 + l16;
        const int sw = (d >> 1) & 7;
        const int k1 = 32 * kb2 + 4 * lg, k2 = k1 + 16;
        const uint2 v1 = *reinterpret_cast<const uint2*>(Vs + d * 128 + ((((k1 >> 3)) ^ sw) << 4) + (k1 & 7) * 2);
        const uint2 v2 = *reinterpret_cast<const uint2*>(Vs + d * 128 + ((((k2 >> 3)) ^ sw) << 4) + (k2 & 7) * 2);
        const uint4 vf = make_uint4(v1.x, v1.y, v2.x, v2.y);
        oacc[di] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf),
                                                           __builtin_bit_cast(bf16x8_t, pf), oacc[di], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: lane holds O[query l16][d = 16di + 4lg + r]
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (qi < p.Sq) {
    bf16_t* orow = reinterpret_cast<bf16_t*>(p.o) + bz * p.o_sb + h * p.o_sh + (int64_t)qi * p.o_ss;
#pragma unroll
    for (int di = 0; di < DIN; ++di) {
      uint2 ov;
      ov.x = pack_bf16(oacc[di][0] * inv, oacc[di][1] * inv);
      ov.y = pack_bf16(oacc[di][2] * inv, oacc[di][3] * inv);
      if (di * 16 + 4 * lg < DV) *reinterpret_cast<uint2*>(orow + di * 16 + 4 * lg) = ov;
    }
    // (a key range in which this query sees nothing: -inf, so that the fold gives it no weight; unsplit: 0 like the other kernels)
    if (lg == 0 && p.lse)
      p.lse[((int64_t)bz * p.Hq + h) * p.Sq + qi] = l_tot > 0.f ? m_run * 0.6931471805599453f + logf(l_tot)
                                                                : (p.nsplit > 1 ? -INFINITY : 0.f);
  }
}

// fold of the key-range partials: o = sum_s w_s o_s / sum_s w_s with w_s = exp(lse_s - max lse), lse = max + log sum w_s.
// One wave per (b, h, query) row; partial o rows are bf16 [B * nsplit, Hq, Sq, D] contiguous, partial lse fp32 [B * nsplit, Hq, Sq].
template <int D>
__global__ __launch_bounds__(256) void attn_split_combine_k(const bf16_t* __restrict__ op, const float* __restrict__ lp, char* o,
                                                            int64_t o_sb, int64_t o_sh, int64_t o_ss, float* __restrict__ lse,
                                                            int B, int Hq, int Sq, int nsplit) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Hq * Sq) return;
  const int i = (int)(row % Sq), h = (int)((row / Sq) % Hq), b = (int)(row / ((int64_t)Sq * Hq));
  constexpr int EPL = D / 64;                                 // elements per lane
  constexpr int NSB = 8;                                      // ranges per batch of loads
  float acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
  float mx = -INFINITY;
  const int64_t pr0 = ((int64_t)b * nsplit * Hq + h) * Sq + i, prs = (int64_t)Hq * Sq;    // partial row of range s: pr0 + s * prs
  for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, lp[pr0 + s * prs]);
  float wsum = 0.f;
  for (int s0 = 0; s0 < nsplit; s0 += NSB) {                  // NSB ranges' rows in flight at once (they are independent loads)
    float w[NSB], v[NSB][EPL];
#pragma unroll
    for (int u = 0; u < NSB; ++u) {
      const int s = min(s0 + u, nsplit - 1);
      w[u] = (s0 + u < nsplit && mx != -INFINITY) ? expf(lp[pr0 + s * prs] - mx) : 0.f;
      const bf16_t* src = op + (pr0 + s * prs) * D + lane * EPL;
#pragma unroll
      for (int e = 0; e < EPL; ++e) v[u][e] = bf2f(src[e]);
    }
#pragma unroll
    for (int u = 0; u < NSB; ++u) {                           // range order: deterministic
      wsum += w[u];
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[e] += w[u] * v[u][e];
    }
  }
  const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
  bf16_t* dst = reinterpret_cast<bf16_t*>(o) + b * o_sb + h * o_sh + (int64_t)i * o_ss + lane * EPL;
#pragma unroll
  for (int e = 0; e < EPL; ++e) dst[e] = f2bf(acc[e] * inv);
  if (lane == 0) lse[row] = wsum > 0.f ? mx + logf(wsum) : 0.f;
}

// -------------------------------------------------------------------------------- backward helpers
// P[b,h,i,j] = visible ? exp(scale*S - lse) : 0   (S fp32 -> P dtype T)
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_k(const float* __restrict__ S, const float* __restrict__ lse,
                                                    T* __restrict__ P, int B, int H, int Sq, int Sk, float scale, int causal,
                                                    const int32_t* __restrict__ kv_start, const int32_t* __restrict__ kv_end,
                                                    const int32_t* __restrict__ q_limit, const uint8_t* __restrict__ key_valid) {
  const int64_t total = (int64_t)B * H * Sq * Sk;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int j = (int)(it % Sk);
    const int64_t row = it / Sk;
    const int i = (int)(row % Sq);
    const int b = (int)(row / ((int64_t)Sq * H));
    int j0 = kv_start ? kv_start[b] : 0, j1 = kv_end ? kv_end[b] : Sk;
    if (causal) j1 = min(j1, i + (Sk - Sq) + 1);
    if (q_limit) j1 = min(j1, q_limit[(int64_t)b * Sq + i]);
    const bool vis = j >= j0 && j < j1 && (!key_valid || key_valid[(int64_t)b * Sk + j]);
    stf<T>(P + it, vis ? expf(S[it] * scale - lse[row]) : 0.f);
  }
}
// delta[b,h,i] = sum_d dO*O ; one wave per row
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_k(const T* __restrict__ dO, int64_t do_sb, int64_t do_sh, int64_t do_ss,
                                                    const T* __restrict__ O, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                                                    float* __restrict__ delta, int B, int H, int Sq, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * H * Sq) return;
  const int i = (int)(row % Sq);
  const int h = (int)((row / Sq) % H);
  const int b = (int)(row / ((int64_t)Sq * H));
  const T* a = dO + b * do_sb + h * do_sh + (int64_t)i * do_ss;
  const T* c = O + b * o_sb + h * o_sh + (int64_t)i * o_ss;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += ldf<T>(a + d) * ldf<T>(c + d);
  s = wave_sum(s);
  if (lane == 0) delta[row] = s;
}
// dS = P * (dP - delta) * scale   (dP fp32 -> dS dtype T)
// with attention dropout (mask m): dP <- dP * m before the softmax backward, and P is overwritten with P * m, the matrix
// dV = (P * m)^T dO needs (delta = rowsum(dO * O) already equals rowsum(P * m * dP))
template <typename T>
__global__ __launch_bounds__(256) void attn_ds_k(T* __restrict__ P, const float* __restrict__ dP,
                                                 const float* __restrict__ delta, T* __restrict__ dS, int64_t rows, int Sk, float scale,
                                                 const T* __restrict__ mask) {
  const int64_t total = rows * Sk;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int64_t row = it / Sk;
    const float pv = ldf<T>(P + it);
    const float m = mask ? ldf<T>(mask + it) : 1.f;
    stf<T>(dS + it, pv * (dP[it] * m - delta[row]) * scale);
    if (mask) stf<T>(P + it, pv * m);
  }
}

// P[row, :] = softmax(scale * S[row, :]) over the visible keys (dtype T, rounded like the generic kernel: P then P * mask),
// lse[row] = log-sum-exp; one wave per row of a MATERIALISED score slab (forward of the non-flash path at sizes where the
// one-wave-per-row kernel would walk hundreds of keys serially: fp32 serving of pi0, retrieval attention with dropout)
template <typename T>
__global__ __launch_bounds__(256) void attn_softmax_rows_k(const float* __restrict__ S, T* __restrict__ P, float* __restrict__ lse,
                                                           int B, int H, int Sq, int Sk, float scale, int causal,
                                                           const int32_t* __restrict__ kv_start, const int32_t* __restrict__ kv_end,
                                                           const int32_t* __restrict__ q_limit, const uint8_t* __restrict__ key_valid,
                                                           const T* __restrict__ drop) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * H * Sq) return;
  const int i = (int)(row % Sq);
  const int b = (int)(row / ((int64_t)Sq * H));
  int j0 = kv_start ? kv_start[b] : 0, j1 = kv_end ? kv_end[b] : Sk;
  if (causal) j1 = min(j1, i + (Sk - Sq) + 1);
  if (q_limit) j1 = min(j1, q_limit[(int64_t)b * Sq + i]);
  j0 = max(j0, 0);
  j1 = min(j1, Sk);
  const uint8_t* kvld = key_valid ? key_valid + (int64_t)b * Sk : nullptr;
  const float* s = S + row * Sk;
  T* pr = P + row * Sk;
  float mx = -INFINITY;
  for (int j = j0 + lane; j < j1; j += 64)
    if (!kvld || kvld[j]) mx = fmaxf(mx, s[j] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = j0 + lane; j < j1; j += 64)
    if (!kvld || kvld[j]) sum += expf(s[j] * scale - mx);
  sum = wave_sum(sum);
  const bool any = sum > 0.f;
  const float inv = any ? 1.f / sum : 0.f;
  const T* dm = drop ? drop + row * Sk : nullptr;
  for (int j = lane; j < Sk; j += 64) {
    float pj = 0.f;
    if (j >= j0 && j < j1 && (!kvld || kvld[j]) && any) {
      pj = rnd<T>(expf(s[j] * scale - mx) * inv);
      if (dm) pj = rnd<T>(pj * ldf<T>(dm + j));
    }
    stf<T>(pr + j, pj);
  }
  if (lane == 0) lse[row] = any ? mx + logf(sum) : 0.f;
}

// ------------------------------------------------------------------------------ fused flash backward (bf16)
// 64-row tiles of Q/dO (or K/V) are staged ROW-major in LDS ([row][d], 16-byte chunks XOR-swizzled by row) and
// feed both MFMA operand shapes:
//   * "row" fragments (k = d): one 16-byte read per lane                         -> S = Q K^T, dP = dO V^T
//   * "column" fragments (k = tile row): two ds_read_b64_tr_b16 per lane.  Within a 16-lane group lane i passes
//     the address of 4 contiguous d of row (i>>2) and receives column i of that [4 rows][16 d] block, i.e. 4
//     consecutive tile rows of ONE d — the MFMA operand of dV^T += dO^T P, dK^T += Q^T dS, dQ^T += K^T dS^T
//     (measured semantics: scripts/probes/tr_read_probe.hip).
// The k-slot order of those products is (lg,e) <-> row 32*kb2 + 16*(e>>2) + 4*lg + (e&3), which makes a lane's
// own 16 score registers the other operand.  P is recomputed from the saved log-sum-exp; nothing S x S touches
// HBM.  The next tile's global loads are issued before the current tile's MFMAs (register prefetch).
//   dQ kernel  : workgroup = 64 queries of one head, waves own 16 queries; loops over K/V tiles:
//                S^T = K Q^T, dP^T = V dO^T (lane = one query), dS^T = P^T (dP^T - delta) scale, dQ^T += K^T dS^T
//   dKV kernel : workgroup = 64 keys of one kv head, waves own 16 keys; loops over the G query heads of the
//                group and over query tiles: S = Q K^T, dP = dO V^T (lane = one key), dV^T += dO^T P, dK^T += Q^T dS
template <int D, int NT = 256> struct FlashTile {
  static constexpr int NCH = D / 8;             // 16-byte chunks per row
  static constexpr int ROW = D * 2;             // bytes per row
  static constexpr int RM_BYTES = 64 * ROW;
  static constexpr int SWZ = (NCH < 16 ? NCH : 16) - 1;   // chunk ^= row & SWZ (16 rows x 16-byte chunks span the 64 banks)
  static constexpr int RPI = NT / NCH;          // rows covered by one pass of the NT threads
  static constexpr int NPASS = 64 / RPI;        // 16-byte loads per thread per tile
};
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t tile_reg_t __attribute__((ext_vector_type(4)));   // 16 bytes of a staged tile row

// global -> registers: thread (c = tid % NCH, r = tid / NCH + RPI*i) holds chunk c of row r.  Rows past the end
// are clamped to the last row (finite data; the score mask zeroes whatever they produce) — no divergent loads.
template <int D, int NT = 256, int DV = D>
__device__ __forceinline__ void tile_gload(tile_reg_t* __restrict__ reg, const bf16_t* base, int64_t stride, int row0,
                                           int nrows, int tid) {
  using FT = FlashTile<D, NT>;
  const bf16_t* src = base + (tid % FT::NCH) * 8;
  const bool real = DV == D || (tid % FT::NCH) * 8 < DV;     // chunks past the head's DV real columns: zeros, never fetched
#pragma unroll
  for (int i = 0; i < FT::NPASS; ++i) {
    const int r = min(row0 + tid / FT::NCH + FT::RPI * i, nrows - 1);
    reg[i] = (tile_reg_t){0u, 0u, 0u, 0u};
    if (real) reg[i] = *reinterpret_cast<const tile_reg_t*>(src + (int64_t)r * stride);
  }
}
template <int D, int NT = 256>
__device__ __forceinline__ void tile_sstore(char* dst, const tile_reg_t* __restrict__ reg, int tid) {
  using FT = FlashTile<D, NT>;
  const int c = tid % FT::NCH, r0 = tid / FT::NCH;
  // RPI is a multiple of NCH's swizzle period only when RPI >= NCH; the row's low bits are those of r0 then
  char* d0 = dst + r0 * FT::ROW;
#pragma unroll
  for (int i = 0; i < FT::NPASS; ++i) {
    const int r = r0 + FT::RPI * i;
    *reinterpret_cast<tile_reg_t*>(d0 + FT::RPI * i * FT::ROW + ((c ^ (r & FT::SWZ)) << 4)) = reg[i];
  }
}
// Per-lane byte offsets into a swizzled row-major tile, split so that everything that varies inside the MFMA
// loops is a compile-time immediate:
//   row fragment (tile row 16n + l16, d = 32ds + 8lg ..)      : row[ds] + n * 16 * ROW
//   column fragment (d = 16di + l16, rows 32kb2 + 4lg + {0..3}) : col[di] + kb2 * 32 * ROW, and + 16 * ROW
template <int D> struct FragAddr {
  using FT = FlashTile<D>;
  uint32_t row[D / 32], col[D / 16];
  __device__ __forceinline__ FragAddr(int l16, int lg) {
    const int xr = l16 & FT::SWZ;
#pragma unroll
    for (int ds = 0; ds < D / 32; ++ds) row[ds] = l16 * FT::ROW + ((((ds ^ (xr >> 2)) << 2) | (lg ^ (xr & 3))) << 4);
    const int rl = 4 * lg + (l16 >> 2), xc = rl & FT::SWZ, b = (l16 & 3) >> 1;
#pragma unroll
    for (int di = 0; di < D / 16; ++di)
      col[di] = rl * FT::ROW + ((((di ^ (xc >> 1)) << 1) | (b ^ (xc & 1))) << 4) + (l16 & 1) * 8;
  }
};
__device__ __forceinline__ uint4 lds_frag(const char* p) { return *reinterpret_cast<const uint4*>(p); }
template <int D>
__device__ __forceinline__ uint4 lds_frag_col(const char* p) {
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
  const s16x4_t v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 16 * FlashTile<D>::ROW));
  const uint2 u1 = __builtin_bit_cast(uint2, v1), u2 = __builtin_bit_cast(uint2, v2);
  return make_uint4(u1.x, u1.y, u2.x, u2.y);
}

// The flash forward on the BACKWARD's tile machinery (round 5).  attn_fwd_flash_k above transposes every V tile through registers
// (8-byte global loads of 8 different rows per lane, ~100 shift / mask operations per thread and tile) and addresses its K
// fragments through an XOR that the compiler cannot fold into immediates; at head_dim 256 that is 256 registers + 24 spilled,
// and the scratch reloads wait on the same counter as the next tile's prefetch.  The dQ kernel below does MORE matrix work per
// tile (S, dP, dQ: 96 MFMAs against this kernel's 64) in less time, because its tiles are staged row-major as they lie in HBM
// (16-byte loads, no shuffling) and the transposed operand comes out of LDS with ds_read_b64_tr_b16.  Same structure here:
// S^T = K Q^T from row fragments of K, O^T += V^T P^T from COLUMN fragments of the row-major V tile; prologue, masks, online
// softmax, key-range splits and epilogue are attn_fwd_flash_k's, statement for statement.
template <int D, int NW, int DV = D>
__global__ __launch_bounds__(64 * NW) void attn_fwd_tr_k(const AttnP p) {
  static_assert(DV % 8 == 0 && DV <= D && DV > D / 2, "valid head width: a multiple of 8 in (D/2, D]");
  constexpr int DSN = (DV + 31) / 32;   // k-steps of 32 that hold real columns
  constexpr int DIN = (DV + 15) / 16;   // 16-column output blocks that hold real columns
  constexpr int NT = 64 * NW;
  using FT = FlashTile<D, NT>;
  __shared__ __attribute__((aligned(16))) char smem[2 * FT::RM_BYTES];
  char* Ks = smem;
  char* Vs = smem + FT::RM_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lg = lane >> 4;
  const FragAddr<D> fa(l16, lg);
  const int bz = blockIdx.z, h = blockIdx.y;
  const int b = p.nsplit > 1 ? bz / p.nsplit : bz;             // key-range splits: (batch, range) on grid z
  const int sp = bz - b * p.nsplit;
  const int hk = h / (p.Hq / p.Hkv);
  const int q0 = blockIdx.x * (16 * NW);
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(p.q) + b * p.q_sb + h * p.q_sh;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(p.k) + b * p.k_sb + hk * p.k_sh;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(p.v) + b * p.v_sb + hk * p.v_sh;

  // Q fragments: MFMA second operand, lane (q = l16, k-group lg) holds d = 32*ds + 8*lg .. +7
  const int qi = q0 + wave * 16 + l16;
  uint4 qf[DSN];
#pragma unroll
  for (int ds = 0; ds < DSN; ++ds) {
    qf[ds] = make_uint4(0, 0, 0, 0);
    if (qi < p.Sq && 32 * ds + 8 * lg < DV) qf[ds] = *reinterpret_cast<const uint4*>(qb + (int64_t)qi * p.q_ss + 32 * ds + 8 * lg);
  }
  f32x4_t oacc[DIN];
#pragma unroll
  for (int i = 0; i < DIN; ++i) oacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float sc2 = p.scale * LOG2E;   // softmax in base 2: exp(x) = 2^(x log2 e), one v_exp_f32 per score

  int j_lo = p.kv_start ? p.kv_start[b] : 0;
  int j_hi = p.kv_end ? p.kv_end[b] : p.Sk;
  j_lo = max(j_lo, 0);
  j_hi = min(j_hi, p.Sk);
  if (p.nsplit > 1) {
    j_lo = max(j_lo, sp * p.split_len);
    j_hi = min(j_hi, (sp + 1) * p.split_len);
  }
  const int coff = p.Sk - p.Sq;
  int blk_hi = j_hi;  // exclusive key bound for the whole workgroup
  if (p.causal) blk_hi = min(blk_hi, min(q0 + 16 * NW - 1, p.Sq - 1) + coff + 1);
  int my_hi = p.causal ? min(j_hi, qi + coff + 1) : j_hi;  // exclusive bound for this lane's query
  if (p.q_limit) my_hi = min(my_hi, qi < p.Sq ? p.q_limit[(int64_t)b * p.Sq + qi] : 0);   // block-prefix mask (pi0)
  const uint8_t* kvld = p.key_valid ? p.key_valid + (int64_t)b * p.Sk : nullptr;
  const int t_lo = j_lo / 64, t_hi = (blk_hi + 63) / 64;

  tile_reg_t rk[FT::NPASS], rv[FT::NPASS];
  if (t_lo < t_hi) {
    tile_gload<D, NT, DV>(rk, kb, p.k_ss, t_lo * 64, p.Sk, tid);
    tile_gload<D, NT, DV>(rv, vb, p.v_ss, t_lo * 64, p.Sk, tid);
  }
  for (int kt = t_lo; kt < t_hi; ++kt) {
    const int key0 = kt * 64;
    __syncthreads();                       // previous tile fully consumed
    tile_sstore<D, NT>(Ks, rk, tid);
    tile_sstore<D, NT>(Vs, rv, tid);
    __syncthreads();
    if (kt + 1 < t_hi) {                   // next tile's loads fly during this tile's MFMAs
      tile_gload<D, NT, DV>(rk, kb, p.k_ss, key0 + 64, p.Sk, tid);
      tile_gload<D, NT, DV>(rv, vb, p.v_ss, key0 + 64, p.Sk, tid);
    }
    // ---- S^T = K Q^T : sacc[n][r] = score(query l16, key key0 + 16n + 4lg + r)
    f32x4_t sacc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      sacc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < DSN; ++ds) {
        const uint4 kf = lds_frag(Ks + fa.row[ds] + n * 16 * FT::ROW);
        sacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf),
                                                          __builtin_bit_cast(bf16x8_t, qf[ds]), sacc[n], 0, 0, 0);
      }
    }
    // ---- online softmax for this lane's query
    // key validity of the 64 keys of this tile as one 64-bit wave-uniform mask: lane j reads key_valid[key0 + j] (one coalesced
    // 64-byte access) and the ballot spreads it, instead of 16 scattered byte loads per lane
    unsigned long long kmask = ~0ull;
    if (kvld) {
      const int kj = key0 + lane;
      kmask = __ballot(kj < p.Sk && kvld[kj] != 0);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + 16 * n + 4 * lg + r;
        const bool vis = key >= j_lo && key < my_hi && ((kmask >> (16 * n + 4 * lg + r)) & 1ull);
        const float s = vis ? sacc[n][r] * sc2 : -INFINITY;          // scores in log2 units
        sacc[n][r] = s;
        tmax = fmaxf(tmax, s);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float alpha = 1.f;
    if (m_new > -INFINITY) alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // m_run = -inf -> 0
    float psum = 0.f;
    uint32_t pk[8];  // bf16 pairs: pk[2*kb2 + ...] see below
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e[r] = (m_new > -INFINITY) ? __builtin_amdgcn_exp2f(sacc[n][r] - m_new) : 0.f;  // 2^-inf = 0 for masked keys
        psum += e[r];
      }
      pk[2 * n] = pack_bf16(e[0], e[1]);
      pk[2 * n + 1] = pack_bf16(e[2], e[3]);
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DIN; ++i) {
      oacc[i][0] *= alpha; oacc[i][1] *= alpha; oacc[i][2] *= alpha; oacc[i][3] *= alpha;
    }
    // ---- O^T += V^T P^T: k-slot (lg, e) of 32-key block kb2 <-> key 32*kb2 + 16*(e>>2) + 4*lg + (e&3), the order in which
    //      lds_frag_col hands out the tile rows, so that this lane's own packed scores are the other operand
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      const uint4 pf = make_uint4(pk[4 * kb2], pk[4 * kb2 + 1], pk[4 * kb2 + 2], pk[4 * kb2 + 3]);
#pragma unroll
      for (int di = 0; di < DIN; ++di) {
        const uint4 vtf = lds_frag_col<D>(Vs + fa.col[di] + kb2 * 32 * FT::ROW);
        oacc[di] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vtf),
                                                           __builtin_bit_cast(bf16x8_t, pf), oacc[di], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: lane holds O[query l16][d = 16di + 4lg + r]
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (qi < p.Sq) {
    bf16_t* orow = reinterpret_cast<bf16_t*>(p.o) + bz * p.o_sb + h * p.o_sh + (int64_t)qi * p.o_ss;
#pragma unroll
    for (int di = 0; di < DIN; ++di) {
      uint2 ov;
      ov.x = pack_bf16(oacc[di][0] * inv, oacc[di][1] * inv);
      ov.y = pack_bf16(oacc[di][2] * inv, oacc[di][3] * inv);
      if (di * 16 + 4 * lg < DV) *reinterpret_cast<uint2*>(orow + di * 16 + 4 * lg) = ov;
    }
    // (a key range in which this query sees nothing: -inf, so that the fold gives it no weight; unsplit: 0 like the other kernels)
    if (lg == 0 && p.lse)
      p.lse[((int64_t)bz * p.Hq + h) * p.Sq + qi] = l_tot > 0.f ? m_run * 0.6931471805599453f + logf(l_tot)
                                                                : (p.nsplit > 1 ? -INFINITY : 0.f);
  }
}


struct AttnBwdP {
  AttnP f;                 // forward tensors (o unused here) + lse
  float* delta;            // [B,Hq,Sq] rowsum(dO * O): written by the dQ kernel, read by the dK/dV kernel
  const char* d_o; int64_t do_sb, do_sh, do_ss;
  char* dq; int64_t dq_sb, dq_sh, dq_ss;
  char* dk; int64_t dk_sb, dk_sh, dk_ss;
  char* dv; int64_t dv_sb, dv_sh, dv_ss;
};

// fp32 attention BACKWARD of the diffusion heads (17 query tokens per sample, 12 - 16 heads of 64, 17 keys — or MemVLA's 256 perceptual
// keys under the 4 x 17 queries of a sample's diffusion repeats: DiT.forward's timm Attention, dexbotic/model/cogact/action_model/
// dit.py:137-162 under ActionModel.loss, action_models.py:102-125; memvla/action_model/dit.py:136-185): the generic path below is five
// batched exact-fp32 products and three elementwise launches per attention call (8 launches of 25 - 35 us for 18 K FLOPs per head).
// Here ONE workgroup per (sample, head) walks queries and keys in chunks of 32 through LDS: delta_i = rowsum(dO_i * O_i) from the forward's
// output, then per key chunk, per query chunk: S = q k^T, P = exp(scale S - lse), dP = dO v^T, dS = P (dP - delta) scale;
// dQ += dS k (registers, all queries), dK += dS^T q, dV += P^T dO (registers, this key chunk), written when the key chunk is done.
// Plain fp32 FMAs.  No masks, no dropout, G = 1, Sq <= 96, D <= 64 (D % 4 == 0): 34 KiB of LDS.
constexpr int SB_MAXT = 32, SB_MAXD = 64, SB_MAXQ = 96;
// NT threads: 256 for one query chunk (4 workgroups a CU when samples x heads >= 1024), 1024 for the merged repeats (samples x heads = 256:
// one workgroup a CU — the same 16 waves a CU either way; the products read every operand from LDS and need the waves to hide it).
template <int NT>
__global__ __launch_bounds__(NT) void attn_bwd_small_f32_k(const AttnBwdP bp) {
  const AttnP& p = bp.f;
  __shared__ float sq[SB_MAXT][SB_MAXD + 1], sk[SB_MAXT][SB_MAXD + 1], sv[SB_MAXT][SB_MAXD + 1], sdo[SB_MAXT][SB_MAXD + 1];
  __shared__ float sP[SB_MAXT][SB_MAXT + 1], sD[SB_MAXT][SB_MAXT + 1], sdelta[SB_MAXQ];
  const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  const int Sq = p.Sq, Sk = p.Sk, D = p.D, D4 = D >> 2;
  const float* q = reinterpret_cast<const float*>(p.q) + b * p.q_sb + h * p.q_sh;
  const float* k = reinterpret_cast<const float*>(p.k) + b * p.k_sb + h * p.k_sh;
  const float* v = reinterpret_cast<const float*>(p.v) + b * p.v_sb + h * p.v_sh;
  const float* d_o = reinterpret_cast<const float*>(bp.d_o) + b * bp.do_sb + h * bp.do_sh;
  float* dq = reinterpret_cast<float*>(bp.dq) + b * bp.dq_sb + h * bp.dq_sh;
  float* dk = reinterpret_cast<float*>(bp.dk) + b * bp.dk_sb + h * bp.dk_sh;
  float* dv = reinterpret_cast<float*>(bp.dv) + b * bp.dv_sb + h * bp.dv_sh;
  const float* lse = p.lse + ((int64_t)b * p.Hq + h) * Sq;
  auto load_q = [&](int i0, int ni) {                  // queries i0 .. i0 + ni - 1 (and their dO rows) into sq / sdo
    for (int it = tid; it < ni * D4; it += NT) {
      const int i = it / D4, d = (it - i * D4) * 4;
      const float4 a = *reinterpret_cast<const float4*>(q + (int64_t)(i0 + i) * p.q_ss + d);
      const float4 g = *reinterpret_cast<const float4*>(d_o + (int64_t)(i0 + i) * bp.do_ss + d);
      sq[i][d] = a.x; sq[i][d + 1] = a.y; sq[i][d + 2] = a.z; sq[i][d + 3] = a.w;
      sdo[i][d] = g.x; sdo[i][d + 1] = g.y; sdo[i][d + 2] = g.z; sdo[i][d + 3] = g.w;
    }
  };
  auto load_kv = [&](int j0, int nj) {                 // keys j0 .. j0 + nj - 1 into sk / sv
    for (int it = tid; it < nj * D4; it += NT) {
      const int j = it / D4, d = (it - j * D4) * 4;
      const float4 a = *reinterpret_cast<const float4*>(k + (int64_t)(j0 + j) * p.k_ss + d);
      const float4 g = *reinterpret_cast<const float4*>(v + (int64_t)(j0 + j) * p.v_ss + d);
      sk[j][d] = a.x; sk[j][d + 1] = a.y; sk[j][d + 2] = a.z; sk[j][d + 3] = a.w;
      sv[j][d] = g.x; sv[j][d + 1] = g.y; sv[j][d + 2] = g.z; sv[j][d + 3] = g.w;
    }
  };
  auto probs = [&](int i0, int ni, int nj) {           // sP = P, sD = dP of the (query chunk, key chunk) in LDS
    for (int it = tid; it < ni * nj; it += NT) {
      const int i = it / nj, j = it - i * nj;
      float s_ = 0.f, dp = 0.f;
      for (int d = 0; d < D; ++d) { s_ += sq[i][d] * sk[j][d]; dp += sdo[i][d] * sv[j][d]; }
      sP[i][j] = expf(s_ * p.scale - lse[i0 + i]);
      sD[i][j] = dp;
    }
  };
  // delta_i = rowsum(dO_i * O_i) from the forward's output (rounds 1 - 6a recomputed it as sum_j P_ij dP_ij in a first pass over every
  // (query chunk, key chunk): 45 % of the kernel): four lanes per query row
  const float* o_ = reinterpret_cast<const float*>(p.o) + b * p.o_sb + h * p.o_sh;
  for (int it = tid; it < ((Sq * 4 + 63) & ~63); it += NT) {
    const int i = it >> 2, c = it & 3;
    float t = 0.f;
    if (i < Sq)
      for (int d = c * 4; d < D; d += 16) {
        const float4 g = *reinterpret_cast<const float4*>(d_o + (int64_t)i * bp.do_ss + d);
        const float4 y = *reinterpret_cast<const float4*>(o_ + (int64_t)i * p.o_ss + d);
        t += (g.x * y.x + g.y * y.y) + (g.z * y.z + g.w * y.w);
      }
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    if (i < Sq && c == 0) sdelta[i] = t;
  }
  const bool one_q = Sq <= SB_MAXT;                              // a single query chunk stays in LDS across the key chunks
  // ---- dQ of every query in registers (element it = tid + NT r of [Sq, D]); dK / dV of the key chunk in registers
  constexpr int QR = SB_MAXQ * SB_MAXD / NT, KR = SB_MAXT * SB_MAXD / NT;
  float accq[QR];
#pragma unroll
  for (int r = 0; r < QR; ++r) accq[r] = 0.f;
  for (int j0 = 0; j0 < Sk; j0 += SB_MAXT) {
    const int nj = min(SB_MAXT, Sk - j0);
    float acck[KR], accv[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { acck[r] = 0.f; accv[r] = 0.f; }
    __syncthreads();
    load_kv(j0, nj);
    for (int i0 = 0; i0 < Sq; i0 += SB_MAXT) {
      const int ni = min(SB_MAXT, Sq - i0);
      __syncthreads();
      if (!one_q || j0 == 0) load_q(i0, ni);
      __syncthreads();
      probs(i0, ni, nj);
      __syncthreads();
      for (int it = tid; it < ni * nj; it += NT) {
        const int i = it / nj, j = it - i * nj;
        sD[i][j] = sP[i][j] * (sD[i][j] - sdelta[i0 + i]) * p.scale;          // dS
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < QR; ++r) {
        const int it = tid + NT * r;
        const int i = it / D - i0, d = it % D;
        if (it < Sq * D && i >= 0 && i < ni) {
          float a = accq[r];
          for (int j = 0; j < nj; ++j) a += sD[i][j] * sk[j][d];
          accq[r] = a;
        }
      }
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        const int it = tid + NT * r;
        if (it < nj * D) {
          const int j = it / D, d = it - j * D;
          float a = acck[r], c = accv[r];
          for (int i = 0; i < ni; ++i) { a += sD[i][j] * sq[i][d]; c += sP[i][j] * sdo[i][d]; }
          acck[r] = a; accv[r] = c;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const int it = tid + NT * r;
      if (it < nj * D) {
        const int j = it / D, d = it - j * D;
        dk[(int64_t)(j0 + j) * bp.dk_ss + d] = acck[r];
        dv[(int64_t)(j0 + j) * bp.dv_ss + d] = accv[r];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < QR; ++r) {
    const int it = tid + NT * r;
    if (it < Sq * D) dq[(int64_t)(it / D) * bp.dq_ss + (it % D)] = accq[r];
  }
}

// The same backward for MORE than 32 queries per (sample, head) — MemVLA's perceptual attention: the 4 x 17 rows of a sample's diffusion
// repeats over its 256 perceptual keys, 16 x 16 (sample, head) pairs, one workgroup of 1024 threads each.  attn_bwd_small_f32_k reads two
// LDS words per FMA (one thread per output element): 63 MB through a compute unit's LDS per workgroup = the 237 us it took at 128 B/clk.
// Here every thread keeps a small block of outputs in registers and reads its operands as 16-byte pieces of rows padded to 68 floats:
//   all (<= 96) queries and their dO rows stay in LDS, keys / values come in chunks of 64;
//   S, dP:   thread (ti, tj) -> rows ti + 32 r (r < 3) x keys tj, tj + 32: 10 ds_read_b128 per 48 FMAs; P and dS written once;
//   then     threads 0 .. 511:    dQ rows ti + 32 r x one 4-column piece (registers across the key chunks): 4 reads per 12 FMAs,
//            threads 512 .. 1023: dK, dV of keys tj, tj + 32 x one 4-column piece:                          6 reads per 16 FMAs.
// 21 MB of LDS reads per workgroup instead of 63.  The sums run in the same order (d, j, i ascending) as in attn_bwd_small_f32_k:
// bit-identical results.  139 KiB of dynamic LDS.
constexpr int S2_QP = 96, S2_KC = 64, S2_LD = 68;
constexpr int S2_LDS_BYTES = ((4 * S2_QP + 2 * S2_KC) * S2_LD + 2 * S2_QP) * 4;
__global__ __launch_bounds__(1024) void attn_bwd_small2_f32_k(const AttnBwdP bp) {
  const AttnP& p = bp.f;
  extern __shared__ __attribute__((aligned(16))) float s2_mem[];
  typedef float row_t[S2_LD];
  row_t* sq = reinterpret_cast<row_t*>(s2_mem);
  row_t* sdo = sq + S2_QP;
  row_t* sP = sdo + S2_QP;
  row_t* sD = sP + S2_QP;
  row_t* sk = sD + S2_QP;
  row_t* sv = sk + S2_KC;
  float* sdelta = reinterpret_cast<float*>(sv + S2_KC);
  float* slse = sdelta + S2_QP;
  const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  const int Sq = p.Sq, Sk = p.Sk, D = p.D, D4 = D >> 2;
  const int nrb = (Sq + 31) >> 5;                                // row blocks of 32 queries (<= 3)
  const float* q = reinterpret_cast<const float*>(p.q) + b * p.q_sb + h * p.q_sh;
  const float* k = reinterpret_cast<const float*>(p.k) + b * p.k_sb + h * p.k_sh;
  const float* v = reinterpret_cast<const float*>(p.v) + b * p.v_sb + h * p.v_sh;
  const float* o_ = reinterpret_cast<const float*>(p.o) + b * p.o_sb + h * p.o_sh;
  const float* d_o = reinterpret_cast<const float*>(bp.d_o) + b * bp.do_sb + h * bp.do_sh;
  float* dq = reinterpret_cast<float*>(bp.dq) + b * bp.dq_sb + h * bp.dq_sh;
  float* dk = reinterpret_cast<float*>(bp.dk) + b * bp.dk_sb + h * bp.dk_sh;
  float* dv = reinterpret_cast<float*>(bp.dv) + b * bp.dv_sb + h * bp.dv_sh;
  const float* lse = p.lse + ((int64_t)b * p.Hq + h) * Sq;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // queries, their dO rows (zero rows up to the next multiple of 32), lse, delta_i = rowsum(dO_i * O_i)
  for (int it = tid; it < nrb * 32 * D4; it += 1024) {
    const int i = it / D4, d = (it - i * D4) * 4;
    float4 a = z4, g = z4;
    if (i < Sq) {
      a = *reinterpret_cast<const float4*>(q + (int64_t)i * p.q_ss + d);
      g = *reinterpret_cast<const float4*>(d_o + (int64_t)i * bp.do_ss + d);
    }
    *reinterpret_cast<float4*>(&sq[i][d]) = a;
    *reinterpret_cast<float4*>(&sdo[i][d]) = g;
  }
  for (int it = tid; it < ((Sq * 4 + 63) & ~63); it += 1024) {
    const int i = it >> 2, c = it & 3;
    float t = 0.f;
    if (i < Sq)
      for (int d = c * 4; d < D; d += 16) {
        const float4 g = *reinterpret_cast<const float4*>(d_o + (int64_t)i * bp.do_ss + d);
        const float4 y = *reinterpret_cast<const float4*>(o_ + (int64_t)i * p.o_ss + d);
        t += (g.x * y.x + g.y * y.y) + (g.z * y.z + g.w * y.w);
      }
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    if (i < Sq && c == 0) { sdelta[i] = t; slse[i] = lse[i]; }
  }
  const int ti = tid >> 5, tj = tid & 31, ti0 = ti & ~1;         // S / dP role
  const bool q_role = tid < 512;                                 // dQ role | dK, dV role
  const int t2 = tid & 511, tr = t2 >> 4, tq = (t2 & 15) * 4;    // row (query block row | key) and first column of the 4-column piece
  const bool col_ok = tq < D;
  float4 accq[3] = {z4, z4, z4};
  for (int j0 = 0; j0 < Sk; j0 += S2_KC) {
    const int nj = min(S2_KC, Sk - j0);
    __syncthreads();                                             // the previous chunk's readers are done (first pass: q / dO / delta visible)
    for (int it = tid; it < S2_KC * D4; it += 1024) {
      const int j = it / D4, d = (it - j * D4) * 4;
      float4 a = z4, g = z4;
      if (j < nj) {
        a = *reinterpret_cast<const float4*>(k + (int64_t)(j0 + j) * p.k_ss + d);
        g = *reinterpret_cast<const float4*>(v + (int64_t)(j0 + j) * p.v_ss + d);
      }
      *reinterpret_cast<float4*>(&sk[j][d]) = a;
      *reinterpret_cast<float4*>(&sv[j][d]) = g;
    }
    __syncthreads();
    {
      float sa[3][2], da[3][2];
#pragma unroll
      for (int r = 0; r < 3; ++r) { sa[r][0] = sa[r][1] = da[r][0] = da[r][1] = 0.f; }
      for (int d = 0; d < D; d += 4) {
        const float4 k0 = *reinterpret_cast<const float4*>(&sk[tj][d]), k1 = *reinterpret_cast<const float4*>(&sk[tj + 32][d]);
        const float4 v0 = *reinterpret_cast<const float4*>(&sv[tj][d]), v1 = *reinterpret_cast<const float4*>(&sv[tj + 32][d]);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (32 * r + ti0 < Sq) {                               // (uniform over the wave: its two rows ti0, ti0 + 1 of this block)
            const float4 a = *reinterpret_cast<const float4*>(&sq[ti + 32 * r][d]);
            const float4 g = *reinterpret_cast<const float4*>(&sdo[ti + 32 * r][d]);
            sa[r][0] += a.x * k0.x; sa[r][0] += a.y * k0.y; sa[r][0] += a.z * k0.z; sa[r][0] += a.w * k0.w;
            sa[r][1] += a.x * k1.x; sa[r][1] += a.y * k1.y; sa[r][1] += a.z * k1.z; sa[r][1] += a.w * k1.w;
            da[r][0] += g.x * v0.x; da[r][0] += g.y * v0.y; da[r][0] += g.z * v0.z; da[r][0] += g.w * v0.w;
            da[r][1] += g.x * v1.x; da[r][1] += g.y * v1.y; da[r][1] += g.z * v1.z; da[r][1] += g.w * v1.w;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (32 * r + ti0 < Sq) {
          const int i = ti + 32 * r;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int j = tj + 32 * c;
            float pr = 0.f, ds = 0.f;
            if (i < Sq && j < nj) {
              pr = expf(sa[r][c] * p.scale - slse[i]);
              ds = pr * (da[r][c] - sdelta[i]) * p.scale;
            }
            sP[i][j] = pr;
            sD[i][j] = ds;
          }
        }
      }
    }
    __syncthreads();
    if (q_role) {
      if (col_ok) {
        for (int j = 0; j < nj; ++j) {
          const float4 kv = *reinterpret_cast<const float4*>(&sk[j][tq]);
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            if (32 * r + tr < Sq) {
              const float a = sD[tr + 32 * r][j];
              accq[r].x += a * kv.x; accq[r].y += a * kv.y; accq[r].z += a * kv.z; accq[r].w += a * kv.w;
            }
          }
        }
      }
    } else if (col_ok) {
      float4 ak[2] = {z4, z4}, av[2] = {z4, z4};
      for (int i = 0; i < Sq; ++i) {
        const float4 qv = *reinterpret_cast<const float4*>(&sq[i][tq]);
        const float4 gv = *reinterpret_cast<const float4*>(&sdo[i][tq]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float ds = sD[i][tr + 32 * c], pr = sP[i][tr + 32 * c];
          ak[c].x += ds * qv.x; ak[c].y += ds * qv.y; ak[c].z += ds * qv.z; ak[c].w += ds * qv.w;
          av[c].x += pr * gv.x; av[c].y += pr * gv.y; av[c].z += pr * gv.z; av[c].w += pr * gv.w;
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int j = tr + 32 * c;
        if (j < nj) {
          *reinterpret_cast<float4*>(dk + (int64_t)(j0 + j) * bp.dk_ss + tq) = ak[c];
          *reinterpret_cast<float4*>(dv + (int64_t)(j0 + j) * bp.dv_ss + tq) = av[c];
        }
      }
    }
  }
  if (q_role && col_ok) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int i = tr + 32 * r;
      if (r < nrb && i < Sq) *reinterpret_cast<float4*>(dq + (int64_t)i * bp.dq_ss + tq) = accq[r];
    }
  }
}

template <int D, int NW, int DV = D>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_k(const AttnBwdP bp) {
  const AttnP& p = bp.f;
  constexpr int DSN = (DV + 31) / 32, DIN = (DV + 15) / 16;     // k-steps / output blocks with real columns (attn_fwd_flash_k)
  constexpr int NT = 64 * NW;                  // NW waves of 16 queries share a staged K/V tile (see attn_fwd_flash_k)
  using FT = FlashTile<D, NT>;
  __shared__ __attribute__((aligned(16))) char smem[2 * FT::RM_BYTES];
  char* Ks = smem;
  char* Vs = smem + FT::RM_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lg = lane >> 4;
  const FragAddr<D> fa(l16, lg);
  const int b = blockIdx.z, h = blockIdx.y;
  const int hk = h / (p.Hq / p.Hkv);
  const int q0 = blockIdx.x * (16 * NW);
  const int qi = q0 + wave * 16 + l16;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(p.q) + b * p.q_sb + h * p.q_sh;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(p.k) + b * p.k_sb + hk * p.k_sh;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(p.v) + b * p.v_sb + hk * p.v_sh;
  const bf16_t* dob = reinterpret_cast<const bf16_t*>(bp.d_o) + b * bp.do_sb + h * bp.do_sh;
  const bf16_t* ob = reinterpret_cast<const bf16_t*>(p.o) + b * p.o_sb + h * p.o_sh;
  uint4 qf[DSN], dof[DSN];
  float dlt = 0.f;   // delta = rowsum(dO * O): this lane's 8-wide slices, folded over the 4 lane groups below
#pragma unroll
  for (int ds = 0; ds < DSN; ++ds) {
    qf[ds] = dof[ds] = make_uint4(0, 0, 0, 0);
    if (qi < p.Sq && 32 * ds + 8 * lg < DV) {
      qf[ds] = *reinterpret_cast<const uint4*>(qb + (int64_t)qi * p.q_ss + 32 * ds + 8 * lg);
      dof[ds] = *reinterpret_cast<const uint4*>(dob + (int64_t)qi * bp.do_ss + 32 * ds + 8 * lg);
      const uint2 o0 = *reinterpret_cast<const uint2*>(ob + (int64_t)qi * p.o_ss + 32 * ds + 8 * lg);
      const uint2 o1 = *reinterpret_cast<const uint2*>(ob + (int64_t)qi * p.o_ss + 32 * ds + 8 * lg + 4);
      const uint32_t ow[4] = {o0.x, o0.y, o1.x, o1.y}, dw[4] = {dof[ds].x, dof[ds].y, dof[ds].z, dof[ds].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        dlt += __uint_as_float(ow[e] << 16) * __uint_as_float(dw[e] << 16) +
               __uint_as_float(ow[e] & 0xffff0000u) * __uint_as_float(dw[e] & 0xffff0000u);
    }
  }
  dlt += __shfl_xor(dlt, 16, 64);
  dlt += __shfl_xor(dlt, 32, 64);
  const int64_t rowid = ((int64_t)b * p.Hq + h) * p.Sq + qi;
  if (qi < p.Sq && lg == 0) bp.delta[rowid] = dlt;               // the dK/dV kernel (launched next) reads it
  const float lse2 = (qi < p.Sq ? p.lse[rowid] : 0.f) * LOG2E;   // exp(x) = 2^(x log2 e): one v_exp_f32
  const float sc2 = p.scale * LOG2E;
  f32x4_t acc[DIN];
#pragma unroll
  for (int i = 0; i < DIN; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int j_lo = p.kv_start ? p.kv_start[b] : 0;
  int j_hi = p.kv_end ? p.kv_end[b] : p.Sk;
  j_lo = max(j_lo, 0);
  j_hi = min(j_hi, p.Sk);
  const int coff = p.Sk - p.Sq;
  int blk_hi = j_hi;
  if (p.causal) blk_hi = min(blk_hi, min(q0 + 16 * NW - 1, p.Sq - 1) + coff + 1);
  int my_hi = p.causal ? min(j_hi, qi + coff + 1) : j_hi;
  if (p.q_limit) my_hi = min(my_hi, qi < p.Sq ? p.q_limit[(int64_t)b * p.Sq + qi] : 0);   // block-prefix mask (pi0)
  const uint8_t* kvld = p.key_valid ? p.key_valid + (int64_t)b * p.Sk : nullptr;
  const int t_lo = j_lo / 64, t_hi = (blk_hi + 63) / 64;
  tile_reg_t rk[FT::NPASS], rv[FT::NPASS];
  if (t_lo < t_hi) {
    tile_gload<D, NT, DV>(rk, kb, p.k_ss, t_lo * 64, p.Sk, tid);
    tile_gload<D, NT, DV>(rv, vb, p.v_ss, t_lo * 64, p.Sk, tid);
  }
  for (int kt = t_lo; kt < t_hi; ++kt) {
    const int key0 = kt * 64;
    __syncthreads();                       // previous tile fully consumed
    tile_sstore<D, NT>(Ks, rk, tid);
    tile_sstore<D, NT>(Vs, rv, tid);
    __syncthreads();
    if (kt + 1 < t_hi) {                   // next tile's loads fly during this tile's MFMAs
      tile_gload<D, NT, DV>(rk, kb, p.k_ss, key0 + 64, p.Sk, tid);
      tile_gload<D, NT, DV>(rv, vb, p.v_ss, key0 + 64, p.Sk, tid);
    }
    unsigned long long kmask = ~0ull;       // validity of this tile's 64 keys: one coalesced byte load per lane + ballot
    if (kvld) {
      const int kj = key0 + lane;
      kmask = __ballot(kj < p.Sk && kvld[kj] != 0);
    }
    uint32_t dsp[8];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < DSN; ++ds) {
        const uint4 kf = lds_frag(Ks + fa.row[ds] + n * 16 * FT::ROW);
        const uint4 vf = lds_frag(Vs + fa.row[ds] + n * 16 * FT::ROW);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[ds]), s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, dof[ds]), dp, 0, 0, 0);
      }
      float dsv[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int key = key0 + 16 * n + 4 * lg + rr;
        const bool vis = key >= j_lo && key < my_hi && ((kmask >> (16 * n + 4 * lg + rr)) & 1ull);
        const float pr = vis ? __builtin_amdgcn_exp2f(s[rr] * sc2 - lse2) : 0.f;
        dsv[rr] = pr * (dp[rr] - dlt) * p.scale;
      }
      dsp[2 * n] = pack_bf16(dsv[0], dsv[1]);
      dsp[2 * n + 1] = pack_bf16(dsv[2], dsv[3]);
    }
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      const uint4 dsf = make_uint4(dsp[4 * kb2], dsp[4 * kb2 + 1], dsp[4 * kb2 + 2], dsp[4 * kb2 + 3]);
#pragma unroll
      for (int di = 0; di < DIN; ++di) {
        const uint4 ktf = lds_frag_col<D>(Ks + fa.col[di] + kb2 * 32 * FT::ROW);
        acc[di] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ktf), __builtin_bit_cast(bf16x8_t, dsf), acc[di], 0, 0, 0);
      }
    }
  }
  if (qi < p.Sq) {
    bf16_t* row = reinterpret_cast<bf16_t*>(bp.dq) + b * bp.dq_sb + h * bp.dq_sh + (int64_t)qi * bp.dq_ss;
#pragma unroll
    for (int di = 0; di < DIN; ++di) {
      uint2 ov;
      ov.x = pack_bf16(acc[di][0], acc[di][1]);
      ov.y = pack_bf16(acc[di][2], acc[di][3]);
      if (di * 16 + 4 * lg < DV) *reinterpret_cast<uint2*>(row + di * 16 + 4 * lg) = ov;
    }
  }
}

// NW = 8 (head_dim 256, round 5): TWO waves per 16 keys.  Both compute the scores S and dP of their keys against the whole query
// tile (the contraction over d cannot be cut without a cross-wave sum per tile), each accumulates HALF of the d range of dK / dV:
// 64 accumulator registers per wave instead of 128, so that the workgroup's 8 waves fit 256 registers each — two waves per SIMD
// that hide each other's LDS reads and exponentials — and the next Q / dO tile's global loads (16 registers per tile at 512
// threads) fit in front of this tile's MFMAs again (PF).  Price: S and dP are computed twice (96 MFMAs per wave and tile instead
// of 128, on twice the waves).  With NW = 4 at head_dim 256 the 424 registers of a wave leave one wave per SIMD.
template <int D, int DV = D, int NW = 4, bool PF = (D <= 128)>
__global__ __launch_bounds__(64 * NW, (D <= 128 && NW == 4 ? 2 : 1)) void attn_bwd_dkv_k(const AttnBwdP bp) {
  const AttnP& p = bp.f;
  constexpr int DSN = (DV + 31) / 32, DIN = (DV + 15) / 16;     // k-steps / output blocks with real columns (attn_fwd_flash_k)
  constexpr int NT = 64 * NW, HS = NW / 4;                      // threads; waves that share 16 keys (each owns DIN / HS blocks of d)
  static_assert((NW == 4 || NW == 8) && DIN % HS == 0, "4 waves, or 8 with the d range of dK / dV cut in two");
  constexpr int DIW = DIN / HS;
  using FT = FlashTile<D, NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];       // 2 tiles + 3 x 64 floats (65 KiB at D = 256)
  char* Qs = smem;
  char* Os = smem + FT::RM_BYTES;
  float (*stat)[64] = reinterpret_cast<float (*)[64]>(smem + 2 * FT::RM_BYTES);   // lse*log2e, delta, per-query key limit
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lg = lane >> 4;
  const FragAddr<D> fa(l16, lg);
  // grid (kv head, batch, key tile): the key tile is the SLOWEST dimension of the dispatch order.  Under the causal mask key tile 0
  // is seen by every query tile and the last one by one (G x 5, 4, .. 1 iterations at S = 287): dealt tile-fastest, as until round
  // 5, a compute unit that drew a light workgroup first got a heavy one behind it (makespan 42 units for 26 of work); heaviest
  // tiles first, the light ones fill in behind them (35)
#if defined(DXA_ATTN_OLD_ORDER)        // tuning build: the grid of rounds 1-4 (scripts/build_variant.sh, profiles/r05_attn_order.txt)
  const int b = blockIdx.z, hk = blockIdx.y;
  const int key0 = blockIdx.x * 64;
#else
  const int b = blockIdx.y, hk = blockIdx.x;
  const int key0 = blockIdx.z * 64;
#endif
  const int G = p.Hq / p.Hkv;
  const int key = key0 + (HS == 1 ? wave : (wave & 3)) * 16 + l16;
  const int di0 = HS == 1 ? 0 : (wave >> 2) * DIW;              // first 16-column block of dK / dV this wave accumulates
  uint32_t colw[DIW];                                           // fa.col of the wave's own blocks (static indices in the loops)
#pragma unroll
  for (int i = 0; i < DIW; ++i) colw[i] = (HS == 2 && (wave >> 2)) ? fa.col[(DIN - DIW) + i] : fa.col[i];
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(p.k) + b * p.k_sb + hk * p.k_sh;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(p.v) + b * p.v_sb + hk * p.v_sh;
  uint4 kf[DSN], vf[DSN];
#pragma unroll
  for (int ds = 0; ds < DSN; ++ds) {
    kf[ds] = vf[ds] = make_uint4(0, 0, 0, 0);
    if (key < p.Sk && 32 * ds + 8 * lg < DV) {
      kf[ds] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * p.k_ss + 32 * ds + 8 * lg);
      vf[ds] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * p.v_ss + 32 * ds + 8 * lg);
    }
  }
  f32x4_t dka[DIW], dva[DIW];
#pragma unroll
  for (int i = 0; i < DIW; ++i) { dka[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dva[i] = dka[i]; }
  int j_lo = p.kv_start ? p.kv_start[b] : 0;
  int j_hi = p.kv_end ? p.kv_end[b] : p.Sk;
  j_lo = max(j_lo, 0);
  j_hi = min(j_hi, p.Sk);
  const int coff = p.Sk - p.Sq;
  const bool key_ok = key >= j_lo && key < j_hi && (!p.key_valid || (key < p.Sk && p.key_valid[(int64_t)b * p.Sk + key]));
  const float sc2 = p.scale * LOG2E;
  // queries that can see any key of this workgroup: q >= key0 - coff (causal)
  const int qt_lo = p.causal ? max(0, key0 - coff) / 64 : 0;
  const int nqt = (p.Sq + 63) / 64 - qt_lo;
  const bool blk_live = key0 < j_hi && key0 + 64 > j_lo;
  const int total = (blk_live && nqt > 0) ? G * nqt : 0;     // iterations: (query head of the group, query tile)

  tile_reg_t rq[FT::NPASS], ro[FT::NPASS];
  float rs = 0.f;
  auto gload = [&](int it) {
    const int h = hk * G + it / nqt, q0 = (qt_lo + it % nqt) * 64;
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(p.q) + b * p.q_sb + h * p.q_sh;
    const bf16_t* dob = reinterpret_cast<const bf16_t*>(bp.d_o) + b * bp.do_sb + h * bp.do_sh;
    tile_gload<D, NT, DV>(rq, qb, p.q_ss, q0, p.Sq, tid);
    tile_gload<D, NT, DV>(ro, dob, bp.do_ss, q0, p.Sq, tid);
    if (tid < 128) {
      const int q = q0 + (tid & 63);
      const float* src = (tid < 64 ? p.lse : bp.delta) + ((int64_t)b * p.Hq + h) * p.Sq;
      rs = q < p.Sq ? src[q] : 0.f;
      if (tid < 64) rs *= LOG2E;
    } else if (tid < 192) {                // keys j < limit are visible to query q (INT_MAX without the mask)
      const int q = q0 + (tid & 63);
      rs = (p.q_limit && q < p.Sq) ? (float)p.q_limit[(int64_t)b * p.Sq + q] : 3.0e9f;
    }
  };
  constexpr bool PREFETCH = PF;           // 4 waves at D = 256: the accumulators leave no room for a second tile in registers
  if (PREFETCH && total > 0) gload(0);
  for (int it = 0; it < total; ++it) {
    const int q0 = (qt_lo + it % nqt) * 64;
    __syncthreads();                       // previous tile fully consumed
    if (!PREFETCH) gload(it);
    tile_sstore<D, NT>(Qs, rq, tid);
    tile_sstore<D, NT>(Os, ro, tid);
    if (tid < 192) stat[tid >> 6][tid & 63] = rs;
    __syncthreads();
    if (PREFETCH && it + 1 < total) gload(it + 1);     // next tile's loads fly during this tile's MFMAs
    uint32_t pp[8], dsp[8];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < DSN; ++ds) {
        const uint4 qfr = lds_frag(Qs + fa.row[ds] + n * 16 * FT::ROW);
        const uint4 ofr = lds_frag(Os + fa.row[ds] + n * 16 * FT::ROW);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qfr), __builtin_bit_cast(bf16x8_t, kf[ds]), s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ofr), __builtin_bit_cast(bf16x8_t, vf[ds]), dp, 0, 0, 0);
      }
      // lane holds its key against queries q0 + 16n + 4lg + {0..3}
      const int ql = 16 * n + 4 * lg;
      const float4 lse4 = *reinterpret_cast<const float4*>(&stat[0][ql]);
      const float4 dl4 = *reinterpret_cast<const float4*>(&stat[1][ql]);
      const float4 lm4 = *reinterpret_cast<const float4*>(&stat[2][ql]);
      const float lse[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dl[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
      const float lim[4] = {lm4.x, lm4.y, lm4.z, lm4.w};
      float pv[4], dsv[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int q = q0 + ql + rr;
        const bool vis = key_ok && q < p.Sq && (!p.causal || key <= q + coff) && (float)key < lim[rr];
        pv[rr] = vis ? __builtin_amdgcn_exp2f(s[rr] * sc2 - lse[rr]) : 0.f;      // stat[0] holds lse * log2(e)
        dsv[rr] = pv[rr] * (dp[rr] - dl[rr]) * p.scale;
      }
      pp[2 * n] = pack_bf16(pv[0], pv[1]);
      pp[2 * n + 1] = pack_bf16(pv[2], pv[3]);
      dsp[2 * n] = pack_bf16(dsv[0], dsv[1]);
      dsp[2 * n + 1] = pack_bf16(dsv[2], dsv[3]);
    }
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      const uint4 pf = make_uint4(pp[4 * kb2], pp[4 * kb2 + 1], pp[4 * kb2 + 2], pp[4 * kb2 + 3]);
      const uint4 dsf = make_uint4(dsp[4 * kb2], dsp[4 * kb2 + 1], dsp[4 * kb2 + 2], dsp[4 * kb2 + 3]);
#pragma unroll
      for (int di = 0; di < DIW; ++di) {
        const uint4 otf = lds_frag_col<D>(Os + colw[di] + kb2 * 32 * FT::ROW);
        const uint4 qtf = lds_frag_col<D>(Qs + colw[di] + kb2 * 32 * FT::ROW);
        dva[di] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, otf), __builtin_bit_cast(bf16x8_t, pf), dva[di], 0, 0, 0);
        dka[di] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qtf), __builtin_bit_cast(bf16x8_t, dsf), dka[di], 0, 0, 0);
      }
    }
  }
  if (key < p.Sk) {
    bf16_t* krow = reinterpret_cast<bf16_t*>(bp.dk) + b * bp.dk_sb + hk * bp.dk_sh + (int64_t)key * bp.dk_ss;
    bf16_t* vrow = reinterpret_cast<bf16_t*>(bp.dv) + b * bp.dv_sb + hk * bp.dv_sh + (int64_t)key * bp.dv_ss;
#pragma unroll
    for (int i = 0; i < DIW; ++i) {
      const int dc = (di0 + i) * 16 + 4 * lg;
      uint2 ok, ov;
      ok.x = pack_bf16(dka[i][0], dka[i][1]); ok.y = pack_bf16(dka[i][2], dka[i][3]);
      ov.x = pack_bf16(dva[i][0], dva[i][1]); ov.y = pack_bf16(dva[i][2], dva[i][3]);
      if (dc < DV) {
        *reinterpret_cast<uint2*>(krow + dc) = ok;
        *reinterpret_cast<uint2*>(vrow + dc) = ov;
      }
    }
  }
}

inline bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

AttnP make_params(const dxa_attn_desc* d) {
  AttnP p;
  p.B = d->B; p.Hq = d->Hq; p.Hkv = d->Hkv; p.Sq = d->Sq; p.Sk = d->Sk; p.D = d->D; p.causal = d->causal;
  p.scale = d->scale;
  p.q = (const char*)d->q; p.q_sb = d->q_sb; p.q_sh = d->q_sh; p.q_ss = d->q_ss;
  p.k = (const char*)d->k; p.k_sb = d->k_sb; p.k_sh = d->k_sh; p.k_ss = d->k_ss;
  p.v = (const char*)d->v; p.v_sb = d->v_sb; p.v_sh = d->v_sh; p.v_ss = d->v_ss;
  p.o = (char*)d->o; p.o_sb = d->o_sb; p.o_sh = d->o_sh; p.o_ss = d->o_ss;
  p.lse = d->lse; p.kv_start = d->kv_start; p.kv_end = d->kv_end;
  p.q_limit = d->q_limit; p.key_valid = d->key_valid; p.drop_mask = d->drop_mask;
  p.nsplit = 1; p.split_len = 0;
  return p;
}

int check_common(const dxa_attn_desc* d, const char* who) {
  if (!d) { dxa_set_error("%s: null desc", who); return DXA_ERR_BAD_ARG; }
  if (!(d->dtype == DXA_F32 || d->dtype == DXA_BF16)) { dxa_set_error("%s: bad dtype", who); return DXA_ERR_BAD_ARG; }
  if (d->B < 0 || d->Hq <= 0 || d->Hkv <= 0 || d->Hq % d->Hkv != 0 || d->Sq < 0 || d->Sk < 0 || d->D <= 0) {
    dxa_set_error("%s: bad shape B=%d Hq=%d Hkv=%d Sq=%d Sk=%d D=%d", who, d->B, d->Hq, d->Hkv, d->Sq, d->Sk, d->D);
    return DXA_ERR_BAD_ARG;
  }
  if (!d->q || !d->k || !d->v || !d->o || !d->lse) { dxa_set_error("%s: null tensor", who); return DXA_ERR_BAD_ARG; }
  return DXA_OK;
}

}  // namespace

// head widths of the fused flash kernels: the tile widths themselves and 72 (SigLIP-So400m, siglip_encoder.py:49-52 of the
// reference: 1152 / 16 heads), which runs on the 128-wide tiles with its 72 real columns only (attn_fwd_flash_k, DV)
static bool flash_head_dim(int D) { return D == 64 || D == 72 || D == 128 || D == 256; }
static bool fwd_flash_ok(const dxa_attn_desc* d) {
  const bool strides8 = d->q_ss % 8 == 0 && d->k_ss % 8 == 0 && d->q_sb % 8 == 0 && d->q_sh % 8 == 0 &&
                        d->k_sb % 8 == 0 && d->k_sh % 8 == 0 && d->v_ss % 4 == 0 && d->v_sb % 4 == 0 &&
                        d->v_sh % 4 == 0 && d->o_ss % 4 == 0 && d->o_sb % 4 == 0 && d->o_sh % 4 == 0;
  return !d->force_generic && !d->drop_mask && d->dtype == DXA_BF16 && flash_head_dim(d->D) && strides8 &&
         al(d->q, 16) && al(d->k, 16) && al(d->v, 8) && al(d->o, 8) && d->B <= 65535 && d->Hq <= 65535;
}

// 8-wave workgroups (128 queries share a staged K/V tile) in the forward and dQ kernels: head_dim 256 with at least 256 queries,
// and head width 72 on the 128-wide tiles (SigLIP: half of every staged tile row is padding, so amortising the staging over twice
// the queries pays: forward 75.3 -> 58.1 us, dQ + dK/dV 184.8 -> 168.1 at 48 x 16 heads x 256) when the mask is not causal and
// 128-query tiles waste no more rows than 64-query tiles.  The decoder (head_dim 128, causal, S = 287) and CLIP (64) measured
// equal or slower with 8 waves (profiles/r05_attn_nw_probe.txt) and stay on 4.
static bool wide_tiles(const dxa_attn_desc* d) {
  if (d->D == 256) return d->Sq >= 256;
  if (d->D == 72) return !d->causal && d->Sq >= 128 && (d->Sq + 127) / 128 * 128 <= (d->Sq + 63) / 64 * 64;
  return false;
}

// the forward on row-major V tiles (attn_fwd_tr_k: the default since round 5, same results bit for bit — profiles/r05_attn_variants.txt:
// head_dim 256 344 -> 198 us, 128 65 -> 52, 64 37.5 -> 27.7) loads V rows 16 bytes at a time; a V that is only 8-byte aligned
// stays on attn_fwd_flash_k (DXA_ATTN_FWD_TR=0 sends everything there: A/B and the fallback's tests)
static bool fwd_tr_ok(const dxa_attn_desc* d) {
  static const int on = getenv("DXA_ATTN_FWD_TR") ? atoi(getenv("DXA_ATTN_FWD_TR")) : 1;
  return on && d->v_ss % 8 == 0 && d->v_sb % 8 == 0 && d->v_sh % 8 == 0 && al(d->v, 16);
}

// fp32 head-sized attention (attn_fwd_small_f32_k): the [16][Sk] score slab of a workgroup has to fit the LDS
static bool fwd_small_f32_ok(const dxa_attn_desc* d) {
  static const bool off = getenv("DXA_ATTN_NO_SMALL") != nullptr;
  auto s4 = [](int64_t a, int64_t b, int64_t c) { return a % 4 == 0 && b % 4 == 0 && c % 4 == 0; };
  return !off && !d->force_generic && !d->drop_mask && d->dtype == DXA_F32 && (d->D == 32 || d->D == 64 || d->D == 96 || d->D == 128) &&
         d->Sk <= 2048 && d->B <= 65535 && d->Hq <= 65535 && s4(d->q_sb, d->q_sh, d->q_ss) && s4(d->k_sb, d->k_sh, d->k_ss) &&
         s4(d->o_sb, d->o_sh, d->o_ss) && al(d->q, 16) && al(d->k, 16) && al(d->o, 16);
}

// non-flash forward at sizes where one wave per query row (attn_fwd_generic_k) would walk hundreds of keys serially:
// materialise S = Q K^T with the batched MFMA GEMM, one masked row softmax, O = P V with the batched GEMM again
static bool fwd_materialise(const dxa_attn_desc* d) {
  if (fwd_flash_ok(d) || d->force_generic == 1) return false;          // force_generic 1: the one-wave-per-row kernel
  const double n = (double)d->B * d->Hq * d->Sq * d->Sk;
  if (n * 4 > (double)(1ull << 31)) return false;
  return d->force_generic == 2 || (d->Sq >= 32 && d->Sk >= 128);       // force_generic 2 (tests): this path at any size
}

// Few queries against a long key cache (a KV-cached denoising or decode step: pi0's 17 suffix queries x 8 heads over 833 keys are
// EIGHT workgroups of the flash kernel, each walking 13 key tiles in sequence — 70 us): the keys are cut into ranges, one
// workgroup per (query tile, head, batch, range), and the ranges' normalised partials are folded by their log-sum-exps.
static int fwd_flash_splits(const dxa_attn_desc* d) {
  static const int off = getenv("DXA_ATTN_NO_KSPLIT") != nullptr;
  if (off || !fwd_flash_ok(d) || d->Sk < 512 || d->D == 72) return 1;      // (a 300-key decode step measured the same cut or whole: 4.63 vs 4.57 ms/token)
  const int64_t wgs = (int64_t)((d->Sq + 63) / 64) * d->Hq * d->B;
  if (wgs > 64) return 1;
  const int tiles = (d->Sk + 63) / 64;
  const int want = (int)std::min<int64_t>(tiles, std::max<int64_t>(1, 256 / wgs));
  const int len = ((tiles + want - 1) / want) * 64;            // keys per range, whole tiles
  const int n = (d->Sk + len - 1) / len;
  return n >= 2 && (int64_t)d->B * n <= 65535 ? n : 1;
}

extern "C" size_t dxa_attn_fwd_workspace(const dxa_attn_desc* d) {
  if (!d || d->B <= 0 || d->Sq <= 0 || d->Sk <= 0 || d->Hq <= 0 || d->Hkv <= 0 || d->Hq % d->Hkv) return 0;
  if (const int ns = fwd_flash_splits(d); ns > 1) {
    const size_t rows = (size_t)d->B * ns * d->Hq * d->Sq;
    return align_up(rows * d->D * 2, 256) + align_up(rows * 4, 256);
  }
  if (!fwd_materialise(d)) return 0;
  const size_t n = (size_t)d->B * d->Hq * d->Sq * d->Sk;
  return align_up(n * 4, 256) + align_up(n * (d->dtype == DXA_BF16 ? 2 : 4), 256);
}

extern "C" int dxa_attn_fwd_ws(const dxa_attn_desc* d, void* workspace, size_t workspace_bytes, dxa_stream_t stream) {
  if (int rc = check_common(d, "dxa_attn_fwd_ws")) return rc;
  if (d->B == 0 || d->Sq == 0) return DXA_OK;
  const size_t need = dxa_attn_fwd_workspace(d);
  if (need == 0 || d->Sk == 0) return dxa_attn_fwd(d, stream);
  DXA_CHECK_ARG(workspace && workspace_bytes >= need, "dxa_attn_fwd_ws: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  if (const int ns = fwd_flash_splits(d); ns > 1) {
    const size_t rows = (size_t)d->B * ns * d->Hq * d->Sq;
    bf16_t* op = (bf16_t*)workspace;
    float* lp = (float*)((char*)workspace + align_up(rows * d->D * 2, 256));
    AttnP p = make_params(d);
    const int tiles = (d->Sk + 63) / 64;
    p.nsplit = ns;
    p.split_len = ((tiles + ns - 1) / ns) * 64;
    p.o = (char*)op; p.o_sb = (int64_t)d->Hq * d->Sq * d->D; p.o_sh = (int64_t)d->Sq * d->D; p.o_ss = d->D;
    p.lse = lp;
    dim3 grid((unsigned)((d->Sq + 63) / 64), (unsigned)d->Hq, (unsigned)(d->B * ns));
    if (fwd_tr_ok(d)) {
      if (d->D == 256) hipLaunchKernelGGL((attn_fwd_tr_k<256, 4>), grid, dim3(256), 0, st, p);
      else if (d->D == 128) hipLaunchKernelGGL((attn_fwd_tr_k<128, 4>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_fwd_tr_k<64, 4>), grid, dim3(256), 0, st, p);
    } else if (d->D == 256) hipLaunchKernelGGL((attn_fwd_flash_k<256, 4>), grid, dim3(256), 0, st, p);
    else if (d->D == 128) hipLaunchKernelGGL((attn_fwd_flash_k<128, 4>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_flash_k<64, 4>), grid, dim3(256), 0, st, p);
    dim3 cgrid((unsigned)(((int64_t)d->B * d->Hq * d->Sq + 3) / 4));
    if (d->D == 256) hipLaunchKernelGGL((attn_split_combine_k<256>), cgrid, dim3(256), 0, st, op, lp, (char*)d->o, d->o_sb, d->o_sh, d->o_ss, d->lse, d->B, d->Hq, d->Sq, ns);
    else if (d->D == 128) hipLaunchKernelGGL((attn_split_combine_k<128>), cgrid, dim3(256), 0, st, op, lp, (char*)d->o, d->o_sb, d->o_sh, d->o_ss, d->lse, d->B, d->Hq, d->Sq, ns);
    else hipLaunchKernelGGL((attn_split_combine_k<64>), cgrid, dim3(256), 0, st, op, lp, (char*)d->o, d->o_sb, d->o_sh, d->o_ss, d->lse, d->B, d->Hq, d->Sq, ns);
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  const int G = d->Hq / d->Hkv;
  const size_t n = (size_t)d->B * d->Hq * d->Sq * d->Sk;
  const int64_t SqSk = (int64_t)d->Sq * d->Sk;
  float* Sf = (float*)workspace;
  void* P = (char*)workspace + align_up(n * 4, 256);
  dxa_gemm_desc g;
  memset(&g, 0, sizeof(g));
  g.layout = DXA_NT; g.in_dtype = d->dtype; g.out_dtype = DXA_F32; g.act = DXA_ACT_NONE; g.alpha = 1.f;
  g.nb[0] = d->B; g.nb[1] = d->Hkv; g.nb[2] = G;
  g.M = d->Sq; g.N = d->Sk; g.K = d->D;
  g.A = d->q; g.lda = d->q_ss; g.sA[0] = d->q_sb; g.sA[1] = d->q_sh * G; g.sA[2] = d->q_sh;
  g.B = d->k; g.ldb = d->k_ss; g.sB[0] = d->k_sb; g.sB[1] = d->k_sh; g.sB[2] = 0;
  g.C = Sf; g.ldc = d->Sk; g.sC[0] = (int64_t)d->Hq * SqSk; g.sC[1] = (int64_t)G * SqSk; g.sC[2] = SqSk;
  int rc;
  if ((rc = dxa_gemm(&g, stream))) return rc;
  {
    const int64_t rows = (int64_t)d->B * d->Hq * d->Sq;
    dim3 grid((unsigned)((rows + 3) / 4));
    if (d->dtype == DXA_BF16)
      hipLaunchKernelGGL((attn_softmax_rows_k<bf16_t>), grid, dim3(256), 0, st, Sf, (bf16_t*)P, d->lse, d->B, d->Hq, d->Sq, d->Sk, d->scale,
                         d->causal, d->kv_start, d->kv_end, d->q_limit, d->key_valid, (const bf16_t*)d->drop_mask);
    else
      hipLaunchKernelGGL((attn_softmax_rows_k<float>), grid, dim3(256), 0, st, Sf, (float*)P, d->lse, d->B, d->Hq, d->Sq, d->Sk, d->scale,
                         d->causal, d->kv_start, d->kv_end, d->q_limit, d->key_valid, (const float*)d->drop_mask);
    DXA_CHECK_LAUNCH();
  }
  memset(&g, 0, sizeof(g));
  g.layout = DXA_NN; g.in_dtype = d->dtype; g.out_dtype = d->dtype; g.act = DXA_ACT_NONE; g.alpha = 1.f;
  g.nb[0] = d->B; g.nb[1] = d->Hkv; g.nb[2] = G;
  g.M = d->Sq; g.N = d->D; g.K = d->Sk;
  g.A = P; g.lda = d->Sk; g.sA[0] = (int64_t)d->Hq * SqSk; g.sA[1] = (int64_t)G * SqSk; g.sA[2] = SqSk;
  g.B = d->v; g.ldb = d->v_ss; g.sB[0] = d->v_sb; g.sB[1] = d->v_sh; g.sB[2] = 0;
  g.C = d->o; g.ldc = d->o_ss; g.sC[0] = d->o_sb; g.sC[1] = d->o_sh * G; g.sC[2] = d->o_sh;
  return dxa_gemm(&g, stream);
}

extern "C" int dxa_attn_fwd(const dxa_attn_desc* d, dxa_stream_t stream) {
  if (int rc = check_common(d, "dxa_attn_fwd")) return rc;
  if (d->B == 0 || d->Sq == 0) return DXA_OK;
  hipStream_t st = (hipStream_t)stream;
  const AttnP p = make_params(d);
  const bool flash_ok = fwd_flash_ok(d);
  if (flash_ok) {
    // 8-wave workgroups (128 queries per staged K/V tile) for head_dim 256 once there are enough queries to fill the chip that way
    static const int nw_env = getenv("DXA_ATTN_FWD_NW") ? atoi(getenv("DXA_ATTN_FWD_NW")) : 0;
    const int64_t wgs8 = (int64_t)((d->Sq + 127) / 128) * d->Hq * d->B;
    const int nw = nw_env ? nw_env : (wide_tiles(d) && wgs8 >= 256 ? 8 : 4);
    dim3 grid((unsigned)((d->Sq + 16 * nw - 1) / (16 * nw)), (unsigned)d->Hq, (unsigned)d->B);
    if (fwd_tr_ok(d)) {
#define LAUNCH_TR(D_, DV_)                                                                            \
  do {                                                                                                 \
    if (nw == 8) hipLaunchKernelGGL((attn_fwd_tr_k<D_, 8, DV_>), grid, dim3(512), 0, st, p);           \
    else hipLaunchKernelGGL((attn_fwd_tr_k<D_, 4, DV_>), grid, dim3(256), 0, st, p);                   \
  } while (0)
      if (d->D == 256) LAUNCH_TR(256, 256); else if (d->D == 128) LAUNCH_TR(128, 128);
      else if (d->D == 72) LAUNCH_TR(128, 72); else LAUNCH_TR(64, 64);
#undef LAUNCH_TR
    } else if (nw == 8) {
      if (d->D == 256) hipLaunchKernelGGL((attn_fwd_flash_k<256, 8>), grid, dim3(512), 0, st, p);
      else if (d->D == 128) hipLaunchKernelGGL((attn_fwd_flash_k<128, 8>), grid, dim3(512), 0, st, p);
      else if (d->D == 72) hipLaunchKernelGGL((attn_fwd_flash_k<128, 8, 72>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((attn_fwd_flash_k<64, 8>), grid, dim3(512), 0, st, p);
    } else {
      if (d->D == 256) hipLaunchKernelGGL((attn_fwd_flash_k<256, 4>), grid, dim3(256), 0, st, p);
      else if (d->D == 128) hipLaunchKernelGGL((attn_fwd_flash_k<128, 4>), grid, dim3(256), 0, st, p);
      else if (d->D == 72) hipLaunchKernelGGL((attn_fwd_flash_k<128, 4, 72>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_fwd_flash_k<64, 4>), grid, dim3(256), 0, st, p);
    }
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  if (fwd_small_f32_ok(d)) {
    const int srow = 16 * ((d->Sk + 15) / 16) + 4;
    const size_t lds_s = (size_t)16 * srow * sizeof(float);
    dim3 grid((unsigned)((d->Sq + 15) / 16), (unsigned)d->Hq, (unsigned)d->B);
#define LAUNCH_SMALL(D_)                                                                                          \
  do {                                                                                                            \
    static size_t attr_ = 48 * 1024;                                                                              \
    if (lds_s > attr_) {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_small_f32_k<D_>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (2048 + 4) * 4);                 \
      attr_ = 16 * (2048 + 4) * 4;                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL((attn_fwd_small_f32_k<D_>), grid, dim3(256), lds_s, st, p, srow);                          \
  } while (0)
    if (d->D == 32) LAUNCH_SMALL(32); else if (d->D == 64) LAUNCH_SMALL(64); else if (d->D == 96) LAUNCH_SMALL(96); else LAUNCH_SMALL(128);
#undef LAUNCH_SMALL
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  const size_t lds = 4 * (size_t)(d->D + d->Sk) * sizeof(float);
  DXA_CHECK_ARG(lds <= 160 * 1024, "dxa_attn_fwd: generic kernel supports Sk+D <= 10240 (got %d)", d->Sk + d->D);
  const int64_t rows = (int64_t)d->B * d->Hq * d->Sq;
  dim3 grid((unsigned)((rows + 3) / 4));
  const size_t es = d->dtype == DXA_BF16 ? 2 : 4;
  const bool vec = d->D % 4 == 0 && d->k_ss % 4 == 0 && d->k_sb % 4 == 0 && d->k_sh % 4 == 0 && al(d->k, 4 * es);
#define LAUNCH_GENERIC(T, V)                                                                                      \
  do {                                                                                                            \
    if (lds > 48 * 1024)                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_generic_k<T, V>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
    hipLaunchKernelGGL((attn_fwd_generic_k<T, V>), grid, dim3(256), lds, st, p);                                  \
  } while (0)
  if (d->dtype == DXA_BF16) { if (vec) LAUNCH_GENERIC(bf16_t, 4); else LAUNCH_GENERIC(bf16_t, 1); }
  else { if (vec) LAUNCH_GENERIC(float, 4); else LAUNCH_GENERIC(float, 1); }
#undef LAUNCH_GENERIC
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}

// fused flash backward eligibility: bf16, D in {64,128}, 16-byte aligned rows of q/k/v/dO, 8-byte rows of dq/dk/dv
static bool bwd_flash_ok(const dxa_attn_desc* d) {
  auto s8 = [](int64_t a, int64_t b, int64_t c) { return a % 8 == 0 && b % 8 == 0 && c % 8 == 0; };
  auto s4 = [](int64_t a, int64_t b, int64_t c) { return a % 4 == 0 && b % 4 == 0 && c % 4 == 0; };
  return !d->force_generic && !d->drop_mask && d->dtype == DXA_BF16 && flash_head_dim(d->D) && d->B <= 65535 && d->Hq <= 65535 &&
         s8(d->q_sb, d->q_sh, d->q_ss) && s8(d->k_sb, d->k_sh, d->k_ss) && s8(d->v_sb, d->v_sh, d->v_ss) &&
         s8(d->do_sb, d->do_sh, d->do_ss) && s4(d->o_sb, d->o_sh, d->o_ss) && s4(d->dq_sb, d->dq_sh, d->dq_ss) &&
         s4(d->dk_sb, d->dk_sh, d->dk_ss) && s4(d->dv_sb, d->dv_sh, d->dv_ss) && al(d->q, 16) && al(d->k, 16) &&
         al(d->v, 16) && al(d->d_o, 16) && al(d->o, 8) && al(d->dq, 8) && al(d->dk, 8) && al(d->dv, 8);
}

extern "C" size_t dxa_attn_bwd_workspace(const dxa_attn_desc* d) {
  if (!d) return 0;
  if (bwd_flash_ok(d)) return align_up((size_t)d->B * d->Hq * d->Sq * 4, 256);
  const size_t n = (size_t)d->B * d->Hq * d->Sq * d->Sk;
  const size_t es = d->dtype == DXA_BF16 ? 2 : 4;
  return align_up(n * 4, 256) + 2 * align_up(n * es, 256) + align_up((size_t)d->B * d->Hq * d->Sq * 4, 256);
}

extern "C" int dxa_attn_bwd(const dxa_attn_desc* d, void* workspace, size_t workspace_bytes, dxa_stream_t stream) {
  if (int rc = check_common(d, "dxa_attn_bwd")) return rc;
  DXA_CHECK_ARG(d->d_o && d->dq && d->dk && d->dv, "dxa_attn_bwd: null gradient tensor");
  DXA_CHECK_ARG(workspace && workspace_bytes >= dxa_attn_bwd_workspace(d), "dxa_attn_bwd: workspace too small");
  if (d->B == 0 || d->Sq == 0 || d->Sk == 0) return DXA_OK;
  if (bwd_flash_ok(d)) {
    hipStream_t st = (hipStream_t)stream;
    float* delta = (float*)workspace;
    AttnBwdP bp;
    bp.f = make_params(d);
    bp.delta = delta;
    bp.d_o = (const char*)d->d_o; bp.do_sb = d->do_sb; bp.do_sh = d->do_sh; bp.do_ss = d->do_ss;
    bp.dq = (char*)d->dq; bp.dq_sb = d->dq_sb; bp.dq_sh = d->dq_sh; bp.dq_ss = d->dq_ss;
    bp.dk = (char*)d->dk; bp.dk_sb = d->dk_sb; bp.dk_sh = d->dk_sh; bp.dk_ss = d->dk_ss;
    bp.dv = (char*)d->dv; bp.dv_sb = d->dv_sb; bp.dv_sh = d->dv_sh; bp.dv_ss = d->dv_ss;
    static const int nwq_env = getenv("DXA_ATTN_DQ_NW") ? atoi(getenv("DXA_ATTN_DQ_NW")) : 0;
    const int64_t wgs8 = (int64_t)((d->Sq + 127) / 128) * d->Hq * d->B;
    const int nwq = nwq_env ? nwq_env : (wide_tiles(d) && wgs8 >= 256 ? 8 : 4);
    dim3 gq((unsigned)((d->Sq + 16 * nwq - 1) / (16 * nwq)), (unsigned)d->Hq, (unsigned)d->B);
#if defined(DXA_ATTN_OLD_ORDER)
    dim3 gk((unsigned)((d->Sk + 63) / 64), (unsigned)d->Hkv, (unsigned)d->B);
#else
    dim3 gk((unsigned)d->Hkv, (unsigned)d->B, (unsigned)((d->Sk + 63) / 64));
#endif
    // head_dim 256, dK / dV: 0 = 4 waves, no register prefetch (rounds 2-4); 1 = 4 waves + prefetch (416 registers); 2 = 8 waves,
    // the d range of dK / dV cut over wave pairs, + prefetch (256 registers, 7 spilled); 3 = 8 waves without the prefetch (248
    // registers): dQ + dK/dV at B 16 x 8 heads x 816 keys 735 / 753 / 714 / 685 us (profiles/r05_attn_variants.txt; same results
    // bit for bit) -> 3
    static const int dkv256 = getenv("DXA_ATTN_DKV256") ? atoi(getenv("DXA_ATTN_DKV256")) : 3;
#define LAUNCH_DKV(D_, DV_, NW_, PF_)                                                                             \
  do {                                                                                                            \
    constexpr int lds_ = 2 * FlashTile<D_>::RM_BYTES + 3 * 64 * (int)sizeof(float);                               \
    static bool attr_ = false;                                                                                    \
    if (!attr_ && lds_ > 48 * 1024) {                                                                             \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_k<D_, DV_, NW_, PF_>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_);                                \
      attr_ = true;                                                                                               \
    }                                                                                                             \
    hipLaunchKernelGGL((attn_bwd_dkv_k<D_, DV_, NW_, PF_>), gk, dim3(64 * NW_), lds_, st, bp);                    \
  } while (0)
#define LAUNCH_BWD(D_, DV_)                                                                                      \
  do {                                                                                                            \
    if (nwq == 8) hipLaunchKernelGGL((attn_bwd_dq_k<D_, 8, DV_>), gq, dim3(512), 0, st, bp);                      \
    else hipLaunchKernelGGL((attn_bwd_dq_k<D_, 4, DV_>), gq, dim3(256), 0, st, bp);                               \
  } while (0)
    if (d->D == 256) {
      LAUNCH_BWD(256, 256);
      if (dkv256 == 2) LAUNCH_DKV(256, 256, 8, true);
      else if (dkv256 == 3) LAUNCH_DKV(256, 256, 8, false);
      else if (dkv256 == 1) LAUNCH_DKV(256, 256, 4, true);
      else LAUNCH_DKV(256, 256, 4, false);
    } else if (d->D == 128) {
      // the 8-wave cut at head_dim 128 (182 registers without spills against 256 + 2 spilled): dQ + dK/dV of the decoder layer
      // 177.5 -> 172.5 and 175.2 -> 166.5 us in two sessions, the one-request shape 194.5 -> 186 (profiles/r05_attn_variants.txt,
      // r05_attn_nw_probe.txt; same results bit for bit); DXA_ATTN_DKV128=0: the 4-wave kernel of rounds 1-4
      static const int dkv128 = getenv("DXA_ATTN_DKV128") ? atoi(getenv("DXA_ATTN_DKV128")) : 2;
      LAUNCH_BWD(128, 128);
      if (dkv128 == 2) LAUNCH_DKV(128, 128, 8, true); else LAUNCH_DKV(128, 128, 4, true);
    }
    else if (d->D == 72) { LAUNCH_BWD(128, 72); LAUNCH_DKV(128, 72, 4, true); }
    else { LAUNCH_BWD(64, 64); LAUNCH_DKV(64, 64, 4, true); }
#undef LAUNCH_BWD
#undef LAUNCH_DKV
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  // head-sized fp32 attention without masks (the DiT heads' training backward): one launch (DXA_ATTN_NO_SMALL_BWD: the generic path)
  static const bool small_bwd_off = getenv("DXA_ATTN_NO_SMALL_BWD") != nullptr;
  if (!small_bwd_off && !d->force_generic && d->dtype == DXA_F32 && d->Hq == d->Hkv && !d->causal && !d->kv_start && !d->kv_end &&
      !d->q_limit && !d->key_valid && !d->drop_mask && d->Sq <= SB_MAXQ && d->Sk <= 4096 && d->D <= SB_MAXD && d->D % 4 == 0 &&
      d->lse && d->Hq <= 65535 && d->B <= 65535 &&
      ((uintptr_t)d->q % 16 == 0) && ((uintptr_t)d->k % 16 == 0) && ((uintptr_t)d->v % 16 == 0) && ((uintptr_t)d->d_o % 16 == 0) &&
      d->q_ss % 4 == 0 && d->k_ss % 4 == 0 && d->v_ss % 4 == 0 && d->do_ss % 4 == 0 && d->q_sh % 4 == 0 && d->k_sh % 4 == 0 &&
      d->v_sh % 4 == 0 && d->do_sh % 4 == 0 && d->q_sb % 4 == 0 && d->k_sb % 4 == 0 && d->v_sb % 4 == 0 && d->do_sb % 4 == 0 &&
      d->o && ((uintptr_t)d->o % 16 == 0) && d->o_ss % 4 == 0 && d->o_sh % 4 == 0 && d->o_sb % 4 == 0) {
    AttnBwdP bp;
    bp.f = make_params(d);
    bp.delta = nullptr;
    bp.d_o = (const char*)d->d_o; bp.do_sb = d->do_sb; bp.do_sh = d->do_sh; bp.do_ss = d->do_ss;
    bp.dq = (char*)d->dq; bp.dq_sb = d->dq_sb; bp.dq_sh = d->dq_sh; bp.dq_ss = d->dq_ss;
    bp.dk = (char*)d->dk; bp.dk_sb = d->dk_sb; bp.dk_sh = d->dk_sh; bp.dk_ss = d->dk_ss;
    bp.dv = (char*)d->dv; bp.dv_sb = d->dv_sb; bp.dv_sh = d->dv_sh; bp.dv_ss = d->dv_ss;
    // more than one chunk of queries: the register-blocked kernel (16-byte aligned dq / dk / dv rows; DXA_ATTN_SMALL_BWD_V1=1: the
    // one-element-per-thread kernel at 1024 threads, rounds 6a-b)
    static const bool v1_only = getenv("DXA_ATTN_SMALL_BWD_V1") != nullptr;
    const bool v2_ok = !v1_only && ((uintptr_t)d->dq % 16 == 0) && ((uintptr_t)d->dk % 16 == 0) && ((uintptr_t)d->dv % 16 == 0) &&
                       d->dq_ss % 4 == 0 && d->dk_ss % 4 == 0 && d->dv_ss % 4 == 0 && d->dq_sh % 4 == 0 && d->dk_sh % 4 == 0 &&
                       d->dv_sh % 4 == 0 && d->dq_sb % 4 == 0 && d->dk_sb % 4 == 0 && d->dv_sb % 4 == 0;
    if (d->Sq > SB_MAXT && v2_ok) {
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_small2_f32_k), hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS_BYTES);
        attr_set = true;
      }
      hipLaunchKernelGGL(attn_bwd_small2_f32_k, dim3((unsigned)d->Hq, (unsigned)d->B), dim3(1024), S2_LDS_BYTES, (hipStream_t)stream, bp);
    } else if (d->Sq > SB_MAXT) hipLaunchKernelGGL(attn_bwd_small_f32_k<1024>, dim3((unsigned)d->Hq, (unsigned)d->B), dim3(1024), 0, (hipStream_t)stream, bp);
    else hipLaunchKernelGGL(attn_bwd_small_f32_k<256>, dim3((unsigned)d->Hq, (unsigned)d->B), dim3(256), 0, (hipStream_t)stream, bp);
    DXA_CHECK_LAUNCH();
    return DXA_OK;
  }
  const int G = d->Hq / d->Hkv;
  if (G > 1) {
    if (d->q_sh != (int64_t)d->Sq * d->q_ss || d->do_sh != (int64_t)d->Sq * d->do_ss) {
      dxa_set_error("dxa_attn_bwd: GQA needs head-major q and dO (sh == Sq*ss) to fold the group into GEMM rows");
      return DXA_ERR_UNSUPPORTED;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)d->B * d->Hq * d->Sq * d->Sk;
  const size_t es = d->dtype == DXA_BF16 ? 2 : 4;
  char* w = (char*)workspace;
  float* Sf = (float*)w; w += align_up(n * 4, 256);
  void* P = w; w += align_up(n * es, 256);
  void* dS = w; w += align_up(n * es, 256);
  float* delta = (float*)w;
  const int64_t SqSk = (int64_t)d->Sq * d->Sk;

  dxa_gemm_desc g;
  auto reset = [&](int layout, int out_dtype) {
    memset(&g, 0, sizeof(g));
    g.layout = layout; g.in_dtype = d->dtype; g.out_dtype = out_dtype; g.act = DXA_ACT_NONE; g.alpha = 1.f;
    g.nb[0] = d->B; g.nb[1] = d->Hkv; g.nb[2] = G;
  };
  int rc;
  // 1. S = Q K^T (fp32 out), batch (b, hkv, g)
  reset(DXA_NT, DXA_F32);
  g.M = d->Sq; g.N = d->Sk; g.K = d->D;
  g.A = d->q; g.lda = d->q_ss; g.sA[0] = d->q_sb; g.sA[1] = d->q_sh * G; g.sA[2] = d->q_sh;
  g.B = d->k; g.ldb = d->k_ss; g.sB[0] = d->k_sb; g.sB[1] = d->k_sh; g.sB[2] = 0;
  g.C = Sf; g.ldc = d->Sk; g.sC[0] = (int64_t)d->Hq * SqSk; g.sC[1] = (int64_t)G * SqSk; g.sC[2] = SqSk;
  if ((rc = dxa_gemm(&g, stream))) return rc;
  // 2. P = exp(scale*S - lse)
  {
    dim3 grid(dxa_grid1d((int64_t)n, 256));
    if (d->dtype == DXA_BF16)
      hipLaunchKernelGGL((attn_probs_k<bf16_t>), grid, dim3(256), 0, st, Sf, d->lse, (bf16_t*)P, d->B, d->Hq, d->Sq, d->Sk, d->scale, d->causal, d->kv_start, d->kv_end, d->q_limit, d->key_valid);
    else
      hipLaunchKernelGGL((attn_probs_k<float>), grid, dim3(256), 0, st, Sf, d->lse, (float*)P, d->B, d->Hq, d->Sq, d->Sk, d->scale, d->causal, d->kv_start, d->kv_end, d->q_limit, d->key_valid);
  }
  // 3. delta = rowsum(dO * O)
  {
    const int64_t rows = (int64_t)d->B * d->Hq * d->Sq;
    dim3 grid((unsigned)((rows + 3) / 4));
    if (d->dtype == DXA_BF16)
      hipLaunchKernelGGL((attn_delta_k<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)d->d_o, d->do_sb, d->do_sh, d->do_ss, (const bf16_t*)d->o, d->o_sb, d->o_sh, d->o_ss, delta, d->B, d->Hq, d->Sq, d->D);
    else
      hipLaunchKernelGGL((attn_delta_k<float>), grid, dim3(256), 0, st, (const float*)d->d_o, d->do_sb, d->do_sh, d->do_ss, (const float*)d->o, d->o_sb, d->o_sh, d->o_ss, delta, d->B, d->Hq, d->Sq, d->D);
  }
  // 4. dP = dO V^T (fp32 out, reuses the score slab)
  reset(DXA_NT, DXA_F32);
  g.M = d->Sq; g.N = d->Sk; g.K = d->D;
  g.A = d->d_o; g.lda = d->do_ss; g.sA[0] = d->do_sb; g.sA[1] = d->do_sh * G; g.sA[2] = d->do_sh;
  g.B = d->v; g.ldb = d->v_ss; g.sB[0] = d->v_sb; g.sB[1] = d->v_sh; g.sB[2] = 0;
  g.C = Sf; g.ldc = d->Sk; g.sC[0] = (int64_t)d->Hq * SqSk; g.sC[1] = (int64_t)G * SqSk; g.sC[2] = SqSk;
  if ((rc = dxa_gemm(&g, stream))) return rc;
  // 5. dS = P * (dP - delta) * scale
  {
    dim3 grid(dxa_grid1d((int64_t)n, 256));
    const int64_t rows = (int64_t)d->B * d->Hq * d->Sq;
    if (d->dtype == DXA_BF16)
      hipLaunchKernelGGL((attn_ds_k<bf16_t>), grid, dim3(256), 0, st, (bf16_t*)P, Sf, delta, (bf16_t*)dS, rows, d->Sk, d->scale, (const bf16_t*)d->drop_mask);
    else
      hipLaunchKernelGGL((attn_ds_k<float>), grid, dim3(256), 0, st, (float*)P, Sf, delta, (float*)dS, rows, d->Sk, d->scale, (const float*)d->drop_mask);
  }
  // 6. dQ = dS K  (NN), batch (b, hkv, g)
  reset(DXA_NN, d->dtype);
  g.M = d->Sq; g.N = d->D; g.K = d->Sk;
  g.A = dS; g.lda = d->Sk; g.sA[0] = (int64_t)d->Hq * SqSk; g.sA[1] = (int64_t)G * SqSk; g.sA[2] = SqSk;
  g.B = d->k; g.ldb = d->k_ss; g.sB[0] = d->k_sb; g.sB[1] = d->k_sh; g.sB[2] = 0;
  g.C = d->dq; g.ldc = d->dq_ss; g.sC[0] = d->dq_sb; g.sC[1] = d->dq_sh * G; g.sC[2] = d->dq_sh;
  if ((rc = dxa_gemm(&g, stream))) return rc;
  // 7. dK = dS^T Q (TN) and 8. dV = P^T dO (TN): contraction over (g, i) = G*Sq rows, batch (b, hkv)
  reset(DXA_TN, d->dtype);
  g.nb[2] = 1;
  g.M = d->Sk; g.N = d->D; g.K = (int64_t)G * d->Sq;
  g.A = dS; g.lda = d->Sk; g.sA[0] = (int64_t)d->Hq * SqSk; g.sA[1] = (int64_t)G * SqSk;
  g.B = d->q; g.ldb = d->q_ss; g.sB[0] = d->q_sb; g.sB[1] = d->q_sh * G;
  g.C = d->dk; g.ldc = d->dk_ss; g.sC[0] = d->dk_sb; g.sC[1] = d->dk_sh;
  if ((rc = dxa_gemm(&g, stream))) return rc;
  g.A = P;
  g.B = d->d_o; g.ldb = d->do_ss; g.sB[0] = d->do_sb; g.sB[1] = d->do_sh * G;
  g.C = d->dv; g.ldc = d->dv_ss; g.sC[0] = d->dv_sb; g.sC[1] = d->dv_sh;
  if ((rc = dxa_gemm(&g, stream))) return rc;
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
