/*
 * dexbotic_amd.h — C ABI of libdexbotic_amd.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * DB-CogACT VLA forward / training hot path (SURVEY.md §8a rows A1–A14).
 *
 * The reference (dexmal/dexbotic v0.2.0) is 100 % Python and has NO FFI boundary of its own
 * (SURVEY.md §0, §8b): every FLOP on the path is executed by torch / transformers / timm underneath
 * the reference's nn.Module classes.  This header therefore declares the boundary a reference
 * maintainer would bind *underneath* those classes; each entry point cites the reference call site
 * (file:line, relative to the reference root; "HF:" = site-packages/transformers/models) whose
 * third-party arithmetic it replaces.  INTEGRATION.md shows the ctypes binding + module swap.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *   - every call is asynchronous on the caller's HIP stream (`stream` = hipStream_t) and never
 *     synchronises, allocates or frees; scratch is passed in by the caller;
 *   - return 0 on success, <0 on error (dxa_status); dxa_last_error() gives a thread-local message;
 *   - dtype codes: DXA_F32 (float) / DXA_BF16 (bfloat16 bits, round-to-nearest-even);
 *   - re-entrant from several host threads / streams; no global mutable state.
 */
#ifndef DEXBOTIC_AMD_H_
#define DEXBOTIC_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dxa_stream_t; /* hipStream_t */

enum dxa_status { DXA_OK = 0, DXA_ERR_BAD_ARG = -1, DXA_ERR_UNSUPPORTED = -2, DXA_ERR_HIP = -3 };
enum dxa_dtype { DXA_F32 = 0, DXA_BF16 = 1 };
enum dxa_act {
  DXA_ACT_NONE = 0,
  DXA_ACT_GELU_ERF = 1,   /* nn.GELU()            — mm_projector/builder.py:71-79          */
  DXA_ACT_GELU_TANH = 2,  /* nn.GELU("tanh")      — cogact/action_model/dit.py:151         */
  DXA_ACT_QUICK_GELU = 3, /* x*sigmoid(1.702x)    — HF:clip/modeling_clip.py (CLIPMLP)     */
  DXA_ACT_SILU = 4,       /* x*sigmoid(x)         — HF:qwen2/modeling_qwen2.py:35-48, dit.py:30 */
  DXA_ACT_RELU = 5,       /* nn.ReLU              — memvla_arch.py:139-155 (BottleneckSE)            */
  DXA_ACT_SIGMOID = 6     /* torch.sigmoid        — memvla_arch.py:143,178 (SE gate, GateFusion)     */
};
/* operand layouts of dxa_gemm: which operand has the contraction index contiguous in memory */
enum dxa_layout {
  DXA_NT = 0, /* C[m,n] = sum_k A[m,k] B[n,k]   y = x W^T      (forward of every nn.Linear)        */
  DXA_NN = 1, /* C[m,n] = sum_k A[m,k] B[k,n]   dx = dy W      (input gradient)                     */
  DXA_TN = 2  /* C[m,n] = sum_k A[k,m] B[k,n]   dW = dy^T x    (weight gradient)                    */
};

const char* dxa_last_error(void);
int dxa_version(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue (MFMA: v_mfma_f32_16x16x32_bf16 for bf16, v_mfma_f32_16x16x4_f32 for
 * exact-fp32).  Replaces torch F.linear / matmul + the adjacent elementwise ops at:
 *   HF:qwen2/modeling_qwen2.py:35-48,105-235 (q/k/v/o, gate/up/down), HF:clip/modeling_clip.py:259-384,
 *   mm_projector/builder.py:71-79, cogact/action_model/dit.py:22-64,137-178 (timm Attention/Mlp),
 *   HF:clip/modeling_clip.py:149-155 (patch-embed conv as a GEMM over im2col rows),
 *   and the autograd-generated dX / dW matmuls of all of them.
 *   v = alpha * sum_k(...) ; v += bias[n] ; if aux_out: aux_out[m,n] = v ; v = act(v) ;
 *   if mulgrad: v *= act'(mulgrad[m,n]) (act given by `act`, v NOT activated in that case) ;
 *   v += residual[m,n] ; if accumulate: v += C[m,n] ; C[m,n] = v.
 * Batched over (nb0, nb1, nb2) with element strides per operand (0 = broadcast).
 * Any M/N/K and any leading dimension are accepted; 16-byte aligned rows take the vector path.
 * ---------------------------------------------------------------------------------------------- */
enum dxa_fuse { DXA_FUSE_NONE = 0, DXA_FUSE_SWIGLU = 1 };
typedef struct dxa_gemm_desc {
  int32_t layout;    /* dxa_layout */
  int32_t in_dtype;  /* dtype of A, B, bias, residual, mulgrad */
  int32_t out_dtype; /* dtype of C and aux_out */
  int32_t act;       /* dxa_act */
  int64_t M, N, K;
  const void* A;
  int64_t lda;
  const void* B;
  int64_t ldb;
  void* C;
  int64_t ldc;
  const void* bias;     /* [N] or NULL */
  const void* residual; /* [M,N] (ld = ldr, batch strides = sR*) or NULL */
  int64_t ldr;
  void* aux_out; /* pre-activation copy [M,N] (ld = ldc, batch strides as C) or NULL */
  const void* mulgrad; /* [M,N] pre-activation saved by the forward (ld = ldg) or NULL */
  int64_t ldg;
  float alpha;
  int32_t accumulate; /* C += result */
  int32_t nb[3];      /* batch extents (>=1) */
  int64_t sA[3], sB[3], sC[3], sR[3], sG[3]; /* batch strides in elements */
  int32_t epi_f32;    /* 1: bias / residual / mulgrad are fp32 although A, B are bf16 (needs out_dtype fp32 and the
                         shapes of the bf16 NT fast path) — the epilogue of a split-bf16 fp32 product, see dxa_split3 */
  void* mirror;       /* fp32 output only, or NULL: a bf16 copy of the final C (after accumulate), same ldc — the
                         communication copy of a weight gradient that the data-parallel reducer exchanges instead of the
                         fp32 values (the reference's DeepSpeed bf16 run reduces bf16 gradients, script/deepspeed/zero2.json);
                         written by the dW product's own epilogue, so no cast pass over the gradient arena exists */
  float* sumsq;       /* unbatched output, or NULL: dxa_gemm_sumsq_slots(M, N) floats, ALL of them written: partial sums of
                         squares of the final C as stored (after accumulate; bf16 output: of the rounded values) whose total,
                         added in index order, is sum(C^2) — the
                         weight gradient's share of the global-norm clip (torch.nn.utils.clip_grad_norm_ in the reference's
                         Trainer, dexbotic/exp/base_exp.py:250 max_grad_norm), produced by the dW product's own epilogue
                         instead of a pass that reads the gradient back; deterministic (fixed fold order per slot) */
  const void* A2;     /* TN only, or NULL: a SECOND pair of operands A2 [K2, M] (ld = lda), B2 [K2, N] (ld = ldb) contracted into */
  const void* B2;     /* the same product: C = A^T B + A2^T B2 in ONE pass over C.  The weight gradient of a linear layer under  */
  int64_t K2;         /* gradient accumulation (the reference recipe: 8 episodes x 2 steps, dexbotic/exp/cogact_exp.py:41-46)    */
                      /* is dY1^T X1 + dY2^T X2: HF accumulates it with a read-modify-write of the fp32 gradient per micro-      */
                      /* batch; here the first micro-batch's (dY, X) are kept and the last one writes dW once                     */
  int32_t fuse;       /* dxa_fuse (0 = none).  DXA_FUSE_SWIGLU: B is the [gate ; up] matrix of a gated MLP, [N = 2 F, K] with the F gate
                         rows first (HF Qwen2MLP gate_proj / up_proj, qwen2/modeling_qwen2.py:35-48 under cogact_arch.py:97-106);
                         C [M, F] (ld = ldc) = silu(gate) * up computed in the product's own epilogue from the rounded
                         pre-activations — bit-identical to dxa_gemm + dxa_swiglu_fwd — and aux_out, if given, the pre-activations
                         [M, 2 F] (ld = ld_aux) the backward needs.  A tile takes its 128 gate and its 128 up columns of the SAME
                         128 outputs (the weight rows are picked by the operand loads: nothing is permuted in memory).
                         bf16 NT products of the MFMA fast path only (M >= 129, K % 64 == 0, F % 8 == 0, no bias / residual /
                         activation / accumulate): anything else is DXA_ERR_BAD_ARG */
  int64_t ld_aux;     /* leading dimension of aux_out when fuse != 0 */
} dxa_gemm_desc;
int dxa_gemm(const dxa_gemm_desc* d, dxa_stream_t stream);
int64_t dxa_gemm_sumsq_slots(int64_t M, int64_t N);

/* fp32 product on the bf16 MFMA path ("bf16x3", the counterpart of the TF32 matmuls the reference's trainer enables
 * with tf32=True, dexbotic/exp/base_exp.py:254, for the fp32 action head under autocast(float32),
 * dexbotic/model/cogact/cogact_arch.py:133): x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits), and
 *   A B^T ~= A_hi B_hi^T + A_hi B_lo^T + A_lo B_hi^T = [A_hi | A_hi | A_lo] [B_hi | B_lo | B_hi]^T,
 * i.e. ONE bf16 NT product with K' = 3K accumulated in fp32 inside the MFMA.  dxa_split3 writes that operand:
 * dst[r, 0:K | K:2K | 2K:3K] = (hi, hi, lo) for side 0 (A) and (hi, lo, hi) for side 1 (B); dst is bf16 [rows, 3*cols]. */
int dxa_split3(const float* src, int64_t ld, void* dst, int64_t rows, int64_t cols, int side, dxa_stream_t stream);
/* the same operand of the TRANSPOSE in one pass: src [R, C] fp32 -> dst [C, 3 Rp] bf16 (columns R..Rp-1 of every part zero):
 * the dX = dY W and dW = dY^T X products of the fp32 heads run as NT products of transposed operands (the backward of the
 * reference's autocast(float32) head, cogact_arch.py:133 / dit.py) */
int dxa_split3_t(const float* src, int64_t ld, void* dst, int64_t R, int64_t C, int64_t Rp, int side, dxa_stream_t stream);
/* BOTH operands of one such product in ONE launch (each as dxa_split3 or dxa_split3_t would write it): a product of the fp32 heads is
 * two operand splits + the MFMA launch, and at 1088 rows a split is a few microseconds of work behind a launch's fixed cost — 1,014 of
 * MemVLA's and 288 of DB-CogACT's launches per step were operand splits.  transposed == 0: rows x cols of src -> dst [rows, 3 cols]
 * (pad unused); transposed != 0: src [rows, cols] -> dst [cols, 3 pad], pad >= rows (dxa_split3_t's R, C, Rp). */
typedef struct dxa_split3_op {
  const float* src;
  int64_t ld;
  void* dst;
  int64_t rows, cols, pad;
  int32_t side;       /* 0: (hi, hi, lo) — the A operand; 1: (hi, lo, hi) — the B operand */
  int32_t transposed;
} dxa_split3_op;
int dxa_split3_pair(const dxa_split3_op* a, const dxa_split3_op* b, dxa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation.  w/b are [cols] in w_dtype (NULL = no affine).
 * RMSNorm follows HF:qwen2/modeling_qwen2.py:238-253: y = w * round_to_dtype(x * rsqrt(mean(x^2)+eps)).
 * LayerNorm follows torch.nn.LayerNorm as used by HF:clip/modeling_clip.py (eps 1e-5, affine) and
 * cogact/action_model/dit.py:143-147,170 (eps 1e-6, no affine).  Statistics in fp32, saved for bwd.
 * Backward writes dx (+ `residual` when given: the gradient that bypassed the normalised sub-block, same shape and
 * dtype as dx) and per-block partial dw/db into `partial` ([nblk, 2*cols] fp32; nblk returned by
 * dxa_norm_bwd_blocks(rows)); reduce them with dxa_colsum.
 * ---------------------------------------------------------------------------------------------- */
int dxa_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int64_t cols,
                    float eps, int dtype, int w_dtype, dxa_stream_t stream);
int dxa_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx,
                    const void* residual, float* partial_dw, int64_t rows, int64_t cols, int dtype, int w_dtype,
                    dxa_stream_t stream);
int dxa_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd,
                      int64_t rows, int64_t cols, float eps, int dtype, int w_dtype, dxa_stream_t stream);
int dxa_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                      void* dx, const void* residual, float* partial_dwdb, int64_t rows, int64_t cols, int dtype,
                      int w_dtype, dxa_stream_t stream);
int dxa_norm_bwd_blocks(int64_t rows);
/* out[c] (+)= sum_r x[r*ld + c]   (bias gradients, norm-weight gradients, pos-emb gradients).
 * Deterministic two-stage reduction; scratch >= min(64, ceil(rows/32)) * cols floats. */
int dxa_colsum(const void* x, int64_t ld, float* out, int64_t rows, int64_t cols, int dtype,
               int accumulate, float* scratch, size_t scratch_bytes, dxa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * RoPE (rotate-half, HF:qwen2/modeling_qwen2.py:107-140) fused with the QKV split into head-major
 * [B,H,S,D] tensors.  cos/sin: [n_pos, D/2] fp32 tables built on the host exactly like
 * Qwen2RotaryEmbedding (:52-104); pos: [B*S] int32 row index into the tables (NULL -> s).
 * dxa_rope_split : qkv [B*S, (Hq+2Hkv)*D] token-major  ->  q [B,Hq,S,D], k [B,Hkv,S,D], v [B,Hkv,S,D]
 * dxa_rope_merge : inverse rotation (backward): dq,dk,dv head-major -> dqkv token-major
 * ---------------------------------------------------------------------------------------------- */
int dxa_rope_split(const void* qkv, void* q, void* k, void* v, const float* cos_t, const float* sin_t,
                   const int32_t* pos, int B, int S, int Hq, int Hkv, int D, int dtype,
                   dxa_stream_t stream);
int dxa_rope_merge(const void* dq, const void* dk, const void* dv, void* dqkv, const float* cos_t,
                   const float* sin_t, const int32_t* pos, int B, int S, int Hq, int Hkv, int D,
                   int dtype, dxa_stream_t stream);
/* The same with head-major tensors that hold S_cap >= S positions per (batch, head): this call's S tokens are positions
 * s0 .. s0 + S - 1 of them (several calls fill / read ONE q, k, v — the two experts of pi0's mixture layer, pi0_arch.py:130-216,
 * whose tokens share one attention call: no concatenation / slicing copies around it). */
int dxa_rope_split_at(const void* qkv, void* q, void* k, void* v, const float* cos_t, const float* sin_t, const int32_t* pos, int B,
                      int S, int Hq, int Hkv, int D, int S_cap, int s0, int dtype, dxa_stream_t stream);
int dxa_rope_merge_at(const void* dq, const void* dk, const void* dv, void* dqkv, const float* cos_t, const float* sin_t,
                      const int32_t* pos, int B, int S, int Hq, int Hkv, int D, int S_cap, int s0, int dtype, dxa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Attention.  Replaces torch SDPA at HF:qwen2/modeling_qwen2.py:143-235 (causal + key padding, GQA),
 * HF:clip/modeling_clip.py:259-330 (full), timm Attention in dit.py:145-149 (full, 17 tokens).
 * Element (b,h,s,d) of q/k/v/o lives at base + b*sb + h*sh + s*ss + d (element strides).
 * softmax in fp32; keys j visible to query i iff kv_start[b] <= j < kv_end[b] and (!causal || j <= i).
 * Forward: fused flash kernel (bf16, D in {64,72,128,256} — 72 = SigLIP-So400m's head width, run on the 128-wide tiles with
 * its 72 real columns only: K/V tiles staged in LDS, MFMA QK^T and PV,
 * wavefront-shuffle softmax reductions) or a generic fp32-accumulate kernel (any dtype / D).
 * Saves lse[b,h,i] = log sum_j exp(scale*s_ij) for the backward.
 * Backward: dq/dk/dv from (q,k,v,o,do,lse); `workspace` must hold dxa_attn_bwd_workspace(desc) bytes.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dxa_attn_desc {
  int32_t dtype;
  int32_t B, Hq, Hkv, Sq, Sk, D;
  int32_t causal;
  float scale;
  const void* q; int64_t q_sb, q_sh, q_ss;
  const void* k; int64_t k_sb, k_sh, k_ss;
  const void* v; int64_t v_sb, v_sh, v_ss;
  void* o;       int64_t o_sb, o_sh, o_ss;
  float* lse;               /* [B,Hq,Sq] */
  const int32_t* kv_start;  /* [B] or NULL (=0)  */
  const int32_t* kv_end;    /* [B] or NULL (=Sk) */
  /* backward only */
  const void* d_o; int64_t do_sb, do_sh, do_ss;
  void* dq; int64_t dq_sb, dq_sh, dq_ss;
  void* dk; int64_t dk_sb, dk_sh, dk_ss;
  void* dv; int64_t dv_sb, dv_sh, dv_ss;
  int32_t force_generic;    /* testing: 1 = bypass the MFMA kernel (one wave per query row); 2 = bypass it and take the
                               materialised path of dxa_attn_fwd_ws at any size */
  /* block-prefix masks of the pi0 mixture-of-transformers attention (dexbotic/model/pi0/pi0_arch.py:22-33):
   * cumsum(ar_mask) is non-decreasing, so "cumsum[j] <= cumsum[i]" is a per-query key count; input_mask is a
   * per-key validity.  Either may be NULL.  With one of them set the generic kernels run. */
  const int32_t* q_limit;   /* [B,Sq]: query i attends keys j < q_limit[b,i] */
  const uint8_t* key_valid; /* [B,Sk]: 0 = key j is padding / a missing camera */
  /* attention dropout as torch SDPA applies it (dropout_p on the weights after the softmax): the retrieval blocks of
   * MemVLA, dexbotic/model/memvla/memvla_arch.py:120-123.  [B,Hq,Sq,Sk] contiguous in the attention dtype, entries 0 or
   * 1/(1-p), drawn by the caller; NULL = no dropout.  Runs the generic kernels. */
  const void* drop_mask;
} dxa_attn_desc;
int dxa_attn_fwd(const dxa_attn_desc* d, dxa_stream_t stream);
/* Forward with caller-provided scratch.  Problems the fused bf16 flash kernel does not take (fp32 — how the reference serves
 * pi0: pi0_exp.py:347-353, eager_attention_forward of pi0_arch.py:185-191 — or attention dropout, memvla_arch.py:120-123) and
 * that are large (Sq >= 32, Sk >= 128) are computed like the reference's eager path: S = Q K^T (batched MFMA GEMM, fp32 scores),
 * one masked row softmax (+ dropout mask), O = P V (batched GEMM).  dxa_attn_fwd_workspace() = bytes of scratch that path
 * needs (0: dxa_attn_fwd_ws behaves exactly like dxa_attn_fwd). */
size_t dxa_attn_fwd_workspace(const dxa_attn_desc* d);
int dxa_attn_fwd_ws(const dxa_attn_desc* d, void* workspace, size_t workspace_bytes, dxa_stream_t stream);
size_t dxa_attn_bwd_workspace(const dxa_attn_desc* d);
int dxa_attn_bwd(const dxa_attn_desc* d, void* workspace, size_t workspace_bytes, dxa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / data-movement kernels (HBM-bound; 16-byte vector access).
 * ---------------------------------------------------------------------------------------------- */
/* SwiGLU, HF:qwen2/modeling_qwen2.py:46-48.  gu = [rows, 2F]: gate = cols [0,F), up = cols [F,2F). */
int dxa_swiglu_fwd(const void* gu, void* out, int64_t rows, int64_t F, int dtype, dxa_stream_t stream);
int dxa_swiglu_bwd(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t F, int dtype,
                   dxa_stream_t stream);
/* gated MLP with any gate activation: out = act(gate) * up — Gemma's GeGLU (act = DXA_ACT_GELU_TANH), HF gemma/
 * modeling_gemma.py GemmaMLP as called from dexbotic/model/pi0/pi0_arch.py:205-208.  Same packing as SwiGLU. */
int dxa_glu_fwd(const void* gu, void* out, int64_t rows, int64_t F, int act, int dtype, dxa_stream_t stream);
int dxa_glu_bwd(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t F, int act, int dtype,
                dxa_stream_t stream);
/* out = alpha*a + beta*b, out = a*b, out[r, n, :] = x[r, n, :] * g[r, :] — the residual / gating arithmetic of the
 * MemVLA memory modules (memvla_arch.py:160-187: SE channel gate, GateFusion lerp) */
int dxa_axpby(const void* a, const void* b, void* out, int64_t n, float alpha, float beta, int dtype, dxa_stream_t stream);
int dxa_mul(const void* a, const void* b, void* out, int64_t n, int dtype, dxa_stream_t stream);
int dxa_mul_rows(const void* x, const void* g, void* out, int64_t R, int64_t Nn, int64_t C, int dtype, dxa_stream_t stream);
int dxa_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, dxa_stream_t stream);
int dxa_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int act, int dtype, dxa_stream_t stream);
/* out = a + b (same dtype) */
int dxa_add(const void* a, const void* b, void* out, int64_t n, int dtype, dxa_stream_t stream);
int dxa_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, dxa_stream_t stream);
/* dst[r, 0:cols] = src[r, 0:cols] with independent leading dimensions (padding columns zeroed) */
int dxa_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols,
               int64_t cols_padded, int src_dtype, int dst_dtype, dxa_stream_t stream);

/* dst[c, r] = src[r, c] (r < R, c < C); dst columns R..R_padded-1 are zero.  Feeds the weight-gradient GEMMs
 * (dW = dY^T X) and the transposed bf16 weight shadows (dX = dY W) as NT products into the fast MFMA path. */
int dxa_transpose(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t R, int64_t C,
                  int64_t R_padded, int dtype, dxa_stream_t stream);
/* [B,S,H,D] <-> [B,H,S,D] (to_head = 1: token-major -> head-major) */
int dxa_permute_bshd(const void* src, void* dst, int B, int S, int H, int D, int to_head, int dtype,
                     dxa_stream_t stream);

/* Token splice (dexbotic_arch.py:182-373).  plan[b*S+s] >= 0: token id; <= -1: image row -1-plan;
 * INT64_MIN: zero padding.  Forward gathers embed_tokens rows / image-feature rows into
 * inputs_embeds [B*S, d].  Backward scatters: image rows plain-stored (each used once, untouched rows
 * must be pre-zeroed), token rows added into the fp32 embedding gradient — without atomics: duplicates of a token id
 * are summed in ascending row order by the workgroup of the first occurrence, so the result is bitwise reproducible.
 * dxa_zero_rows zeroes the gradient rows of the token entries of a (previous) plan: the sparse re-zero of the dense
 * nn.Embedding gradient (dexbotic_arch.py:224-228 embeds through nn.Embedding; its .grad is dense). */
int dxa_splice_fwd(const int64_t* plan, const void* embed, const void* img, void* out, int64_t n_rows,
                   int64_t d, int dtype, dxa_stream_t stream);
int dxa_splice_bwd(const int64_t* plan, const void* dout, float* d_embed, void* d_img, int64_t n_rows,
                   int64_t d, int dtype, dxa_stream_t stream);
int dxa_zero_rows(const int64_t* plan, float* g, int64_t n_rows, int64_t d, dxa_stream_t stream);
/* out[i,:] = x[idx[i],:]   (cognition token, cogact_arch.py:110-120) and its scatter-add backward */
int dxa_gather_rows(const void* x, const int64_t* idx, void* out, int64_t n, int64_t d, int src_dtype,
                    int dst_dtype, dxa_stream_t stream);
/* dx [R,d] is written completely: row r = sum_{i: idx[i]==r} dout[i], every other row = 0 */
int dxa_scatter_rows(const void* dout, const int64_t* idx, void* dx, int64_t n, int64_t R, int64_t d,
                     int src_dtype, int dst_dtype, dxa_stream_t stream);

/* Patch embedding front-end (HF:clip/modeling_clip.py:138-218).
 * im2col: images [N,3,H,W] (img_dtype) -> rows [N*g*g, ld] (dtype), column order (c, py, px) = the
 * flattened Conv2d weight [C, 3*P*P]; columns >= 3*P*P are zero.
 * vit_embed: x[n,0,:] = cls + pos[0]; x[n,1+p,:] = patch[n,p,:] + pos[1+p]. */
int dxa_im2col(const void* images, void* rows, int N, int H, int W, int P, int64_t ld, int img_dtype,
               int dtype, dxa_stream_t stream);
int dxa_vit_embed_fwd(const void* patch, const void* cls, const void* pos, void* x, int N, int np, int C,
                      int dtype, int w_dtype, dxa_stream_t stream);
int dxa_vit_embed_bwd(const void* dx, void* dpatch, int N, int np, int C, int dtype, dxa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Diffusion action expert glue (fp32).  cogact/action_model/{action_models,dit,diffusion}.py.
 * ---------------------------------------------------------------------------------------------- */
/* x_t = a[n]*x0 + s[n]*noise   (diffusion.py:308-326; a,s gathered from the float64 tables on host) */
int dxa_qsample(const float* x0, const float* noise, const float* a, const float* s, float* xt,
                int64_t N, int64_t per, dxa_stream_t stream);
/* sinusoidal timestep embedding out[n] = [cos(t_n f) | sin(t_n f)], f = freqs[half] built on the host
 * exactly as dit.py:45-49 does (exp(-ln(10000) k/half) in fp32) */
int dxa_timestep_embedding(const float* t, const float* freqs, float* out, int64_t N, int half,
                           dxa_stream_t stream);
/* h[n,0,:] = te[n]+ze[n]+pos[0]; h[n,1+t,:] = xe[n,t]+pos[1+t]  (dit.py:281-286) and its backward */
int dxa_dit_assemble_fwd(const float* xe, const float* te, const float* ze, const float* pos, float* h,
                         int N, int T, int hd, dxa_stream_t stream);
int dxa_dit_assemble_bwd(const float* dh, float* dxe, float* dc, int N, int T, int hd, dxa_stream_t stream);
/* z_out[n] = drop[n] ? uncond : z[n]   (LabelEmbedder.token_drop, dit.py:80-96); bwd masks dz */
int dxa_token_drop(const float* z, const float* uncond, const uint8_t* drop, float* out, int64_t N,
                   int64_t d, dxa_stream_t stream);
/* dz[n] = drop[n] ? 0 : dout[n] (dz may be NULL); duncond[c] (+)= sum over dropped n of dout[n,c] */
int dxa_token_drop_bwd(const float* dout, const uint8_t* drop, float* dz, float* duncond, int64_t N,
                       int64_t d, int accumulate, dxa_stream_t stream);
/* ---- MemVLA memory path, device side (dexbotic/model/memvla/memvla_arch.py) ----------------------------------------
 * out[i] = u_i >= p ? 1/(1-p) : 0, u_i uniform in [0,1) from Philox4x32-10 keyed by `seed`, counter (i / 4, offset): one
 * launch per dropout mask of the retrieval blocks (SDPA's dropout_p on the attention weights, memvla_arch.py:120-123, and
 * the two nn.Dropout of the FFN, :99-105).  Statistically the reference's draw, not bit-for-bit torch's generator. */
int dxa_dropout_mask(void* out, int64_t n, float p, uint64_t seed, uint64_t offset, int dtype, dxa_stream_t stream);
/* token-merge consolidation of ONE memory bank on the device (_consolidate_with_token_merge, memvla_arch.py:263-287):
 * feat [len, N, D] (dtype), ts [len] fp32, len >= 2 entries oldest first.  sims[i] = mean_n cos(feat[i,n,:], feat[i+1,n,:])
 * (eps 1e-8 per norm), j = first arg-max; feat[j] <- 0.5 (feat[j] + feat[j+1]), ts[j] likewise, the later entries move down:
 * afterwards the first len - 1 entries are the bank.  `sims`: len - 1 floats of scratch.  No host read-back. */
int dxa_bank_consolidate(void* feat, float* ts, int len, int64_t N, int64_t D, int dtype, int fifo, float* sims,
                         dxa_stream_t stream);      /* fifo = 1: drop the oldest entry instead (memvla_arch.py:300-303) */
/* out[r,n,:] = (x ? x[r,n,:] : 0) + alpha g[r,:]: the timestep embedding added to every token of a bank entry
 * (memvla_arch.py:352-360); x = NULL: broadcast of a row over the tokens (backward of the token mean, :139-141) */
int dxa_add_rows(const void* x, const void* g, void* out, int64_t R, int64_t Nn, int64_t C, float alpha, int dtype,
                 dxa_stream_t stream);
/* out[r,:] = scale * sum_n x[r,n,:] (fp32 accumulation, token order): AdaptiveAvgPool2d(1) of BottleneckSE
 * (memvla_arch.py:139-141) and the gradient of a row broadcast over the tokens */
int dxa_token_sum(const void* x, void* out, int64_t R, int64_t Nn, int64_t C, float scale, int dtype, dxa_stream_t stream);
/* loss = mean((pred-target)^2) (action_models.py:119-121); dpred = 2 (pred-target) / n * gscale */
int dxa_mse_loss(const float* pred, const float* target, float* loss, float* dpred, int64_t n,
                 float gscale, dxa_stream_t stream);
/* sample-weighted variant (HybridCogACT co-training, dexbotic/model/cogact/hybrid_cogact_arch.py:168-173):
 * loss = sum_r w[r] * mean_c((pred-target)[r,c]^2) / (sum_r w[r] + 1e-6); dpred = d loss / d pred * gscale */
int dxa_mse_loss_rows(const float* pred, const float* target, const float* row_w, float* loss, float* dpred,
                      int64_t rows, int64_t cols, float gscale, dxa_stream_t stream);
/* One DDIM(eta=0) update with classifier-free guidance (dit.py:294-311, diffusion.py:626-673):
 * eps = eu + s (ec - eu) with ec = model_out[0:B], eu = model_out[B:2B] (cfg) or eps = model_out;
 * x0 = c_recip x - c_recipm1 eps ; eps' = (c_recip x - x0)/c_recipm1 ; x <- sqrt(ab_prev) x0 +
 * sqrt(1-ab_prev) eps'.  x is [B or 2B, per]; with cfg both halves receive the same result. */
int dxa_ddim_step(float* x, const float* model_out, int64_t B, int64_t per, int use_cfg, float cfg_scale,
                  float c_recip, float c_recipm1, float ab_prev, dxa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer (trainer.py:25-36,88-124 -> torch.optim.AdamW + clip_grad_norm_(1.0)) over a flat arena.
 * chunk table (device): chunk c covers elements [start[c], start[c]+len[c]) and uses group grp[c].
 * Per group: lr, weight_decay.  clip_coef: device float multiplied into every gradient.
 * shadow (optional): bf16 copy of the updated parameters written in the same pass.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dxa_adamw_desc {
  float* p; const void* g; float* m; float* v;
  uint16_t* shadow;            /* or NULL */
  const int64_t* chunk_start; const int32_t* chunk_len; const int32_t* chunk_grp; int32_t n_chunks;
  float lr[8]; float wd[8];
  float beta1, beta2, eps;
  float bc1, bc2;              /* 1-beta1^t, 1-beta2^t */
  const float* clip_coef;      /* device scalar or NULL (=1) */
  int32_t g_dtype;             /* DXA_F32: g is the fp32 gradient arena; DXA_BF16: g is its bf16 communication copy (the
                                  data-parallel run averages bf16 gradients like the reference's DeepSpeed bf16 config,
                                  script/deepspeed/zero2.json, and the optimizer reads the averaged copy directly) */
  uint8_t* chunk_state;        /* or NULL.  One byte per chunk, device memory, kept by the caller across steps: 0 = ordinary chunk;
                                  1 = chunk of a sparsely touched table (nn.Embedding weight: dexbotic_arch.py:182-373 scatters
                                  gradients into <= B * S_text rows per step) that has never seen a non-zero gradient: its m and v
                                  are exactly 0, so while its gradient is all-zero and its group's weight decay is 0 torch's AdamW
                                  leaves p, m, v bit-for-bit unchanged — the kernel reads g only and returns; the first non-zero
                                  gradient turns the byte into 2 = ordinary from then on */
  const int64_t* chunk_mv_start; /* or NULL (m, v laid out like p: element chunk_start[c] + i).  Sharded optimizer state — the
                                  reference's default DeepSpeed ZeRO config partitions the optimizer state over the data-parallel
                                  ranks (dexbotic/exp/base_exp.py:229, script/deepspeed/zero3.json:17-25): a rank keeps m / v only for
                                  the arena ranges it owns, packed back to back; chunk c's moments then start at element
                                  chunk_mv_start[c] of m and v while p, g and shadow keep the arena offset chunk_start[c] */
} dxa_adamw_desc;
int dxa_adamw(const dxa_adamw_desc* d, dxa_stream_t stream);
/* out[0] = sum x^2 over n fp32 / bf16 elements (deterministic two-stage; scratch >= 4096 doubles) */
int dxa_sumsq(const void* x, int64_t n, int dtype, double* scratch, float* out, int accumulate, dxa_stream_t stream);
/* the same over n_ranges <= 4096 short slices base[starts[i] .. starts[i] + lens[i]) (device arrays, element units): the
 * slots of a gradient bucket that no dW epilogue accounts for (dxa_gemm_desc.sumsq), in one launch */
int dxa_sumsq_ranges(const void* base, int dtype, const int64_t* starts, const int64_t* lens, int n_ranges, double* scratch,
                     float* out, int accumulate, dxa_stream_t stream);
/* out[0] (+)= sum of n floats: one workgroup, fixed order, double accumulation (folds dxa_gemm_desc.sumsq partials) */
int dxa_sum_f32(const float* x, int64_t n, float* out, int accumulate, dxa_stream_t stream);
/* norm = sqrt(sumsq); coef = min(1, max_norm/(norm+1e-6))  (torch.nn.utils.clip_grad_norm_) */
int dxa_clip_coef(const float* sumsq, float max_norm, float* norm_out, float* coef_out, dxa_stream_t stream);
/* the same for an arena that holds gradient / grad_scale (data parallelism with SUM collectives leaves world x the mean,
 * grad_scale = 1 / world; replaces DDP's gradient mean + clip_grad_norm_, dexbotic/exp/trainer.py:110,121-122):
 * norm = grad_scale * sqrt(sumsq); coef = grad_scale * min(1, max_norm/(norm+1e-6)) — dxa_adamw multiplies every
 * gradient by coef, so the update is the mean gradient's */
int dxa_clip_coef_scaled(const float* sumsq, float max_norm, float grad_scale, float* norm_out, float* coef_out,
                         dxa_stream_t stream);
int dxa_scale(float* x, int64_t n, float s, dxa_stream_t stream);
/* x *= s[0] with s a DEVICE scalar (upstream loss gradient; no host sync) */
int dxa_scale_dev(float* x, int64_t n, const float* s, dxa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LM head: causal-LM cross-entropy and greedy token choice.
 * dxa_cross_entropy_fwd/bwd: HF ForCausalLMLoss (transformers/loss/loss_utils.py) as called from
 *   dexbotic/model/dexbotic_arch.py:483-488 on logits = lm_head(hidden_states[-1]).  `labels[r]` is the label of
 *   row r AFTER the one-position shift; rows with labels == ignore_index contribute nothing.
 *   fwd: row_loss[r] = logsumexp(logits[r]) - logits[r, label] (fp32 math on fp32 or bf16 logits), lse[r] saved.
 *        loss = sum(row_loss) / #non-ignored rows is taken by the caller (dxa_colsum).
 *   bwd: dlogits[r, v] = (exp(logits[r, v] - lse[r]) - [v == label]) * gscale[0] * scale  (gscale: device scalar
 *        = upstream gradient or NULL for 1; scale = 1 / #non-ignored rows).  dlogits may alias logits.
 * dxa_argmax_rows: out[r] = first index of the row maximum — torch.argmax, the greedy choice inside
 *   GenerationMixin.generate(do_sample=False) that DiscreteVLAForCausalLM decodes with
 *   (dexbotic/model/discrete_vla/discrete_vla_arch.py:33-41).  Integer result, must equal the reference's.
 * ---------------------------------------------------------------------------------------------- */
int dxa_cross_entropy_fwd(const void* logits, int64_t ld, const int64_t* labels, float* row_loss, float* lse,
                          int64_t rows, int64_t V, int64_t ignore_index, int dtype, dxa_stream_t stream);
int dxa_cross_entropy_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* lse,
                          const float* gscale, float scale, void* dlogits, int64_t ldd, int64_t rows, int64_t V,
                          int64_t ignore_index, int dtype, dxa_stream_t stream);
int dxa_argmax_rows(const void* x, int64_t ld, int64_t* out, int64_t rows, int64_t cols, int dtype,
                    dxa_stream_t stream);

/* ---- device-side image preprocessing ------------------------------------------------
 * Replaces, for uint8 RGB frames already on the device, what the reference does per frame on the host with
 * Pillow + the CLIP image processor:
 *   dexbotic/data/dataset/rgb_preprocess.py:13-28   PreprocessRGB.__call__   (training data path)
 *   dexbotic/data/dataset/rgb_preprocess.py:30-44   expand2square
 *   dexbotic/model/dexbotic_arch.py:498-529         process_images           (inference server path)
 * i.e. [expand to square with a constant colour] -> Image.resize(BICUBIC) -> center crop -> x * rescale ->
 * (x - mean) / std -> CHW.  The resize is Pillow's 8-bit two-pass resampler (Pillow 12.2.0
 * src/libImaging/Resample.c; 22-bit fixed-point taps, uint8 image between the passes); the uint8 result
 * (out_u8) is bit-exact with Pillow's, the float result equals the processor's float32 arithmetic.
 *
 * dxa_resample_ksize / dxa_resample_coeffs: HOST functions, the tap tables of one axis (precompute_coeffs +
 *   normalize_coeffs_8bpc): bounds[out_size][2] = (first source index, tap count), kk[out_size][ksize].
 *   The caller copies them to the device once per (in_size, out_size) and passes the device copies below.
 * dxa_image_preprocess: n frames of identical h x w.  `pad` != 0 centres the frame in a max(h,w) square of colour
 *   bg.  res_h x res_w is the size after the resize of that (padded) frame; a pass that does not change the size
 *   has NULL tables (Pillow skips it too).  The crop window [crop_top, +out_h) x [crop_left, +out_w) of the
 *   resized frame is what is produced.  tmp is a uint8 scratch of n * rows * out_w * 3 bytes, where
 *   [row0, row0 + rows) are the (padded) source rows the vertical taps of the cropped output rows touch
 *   (row0 = 0, rows = padded height is always valid).
 * ---------------------------------------------------------------------------------------------- */
#define DXA_FILTER_BICUBIC 3 /* PIL.Image.BICUBIC */
typedef struct dxa_image_desc {
  const void* src;        /* uint8 [n, h, w, 3] RGB */
  int n, h, w;
  int pad;                /* expand2square */
  unsigned char bg[4];    /* r, g, b, unused */
  int res_h, res_w;
  int crop_top, crop_left, out_h, out_w;
  int row0, rows;
  const int32_t *hb, *hk; /* horizontal bounds / taps (device) or NULL */
  int hks;
  const int32_t *vb, *vk; /* vertical */
  int vks;
  void* tmp;              /* uint8 scratch */
  void* out;              /* [n, 3, out_h, out_w] of out_dtype (DXA_F32 or DXA_BF16) */
  int out_dtype;
  void* out_u8;           /* optional uint8 [n, out_h, out_w, 3]: the resized + cropped frame */
  double rescale;         /* 1/255 */
  float mean[3], std[3];
} dxa_image_desc;
int dxa_resample_ksize(int in_size, int out_size);
int dxa_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk);
int dxa_image_preprocess(const dxa_image_desc* d, dxa_stream_t stream);

/* ---- persistent DiT blocks for the action sampler ---------------------------------------------------------------
 * All `depth` DiTBlocks of one denoising call (dexbotic/model/cogact/action_model/dit.py:137-162, walked by DiT.forward
 * :281-293 inside every DDIM step of cogact_arch.py:186-197) in ONE launch: h [N*T1, H] (fp32, in place) goes through
 *   h += proj(attn(qkv(LN(h)))) ; h += fc2(gelu_tanh(fc1(LN(h))))      (LayerNorm without affine, eps; head width 64)
 * `weights` is a DEVICE array of depth*8 fp32 pointers: qkv_w [3H,H], qkv_b, proj_w [H,H], proj_b, fc1_w [I,H], fc1_b,
 * fc2_w [H,I], fc2_b per block.  Limits: N*T1 <= 47 rows, T1 <= 32, H = heads*64 <= 1024, I a multiple of 64 (the CFG
 * batch of one request: 2 x 17 rows); anything else runs block by block on the ordinary kernels.  The workspace holds
 * the qkv / attention / MLP activations and the K-slice partials; the barrier counters live in a library-owned block per
 * (device, stream) that every launch leaves zeroed (first use on a stream allocates it: not under stream capture). */
size_t dxa_dit_blocks_workspace(int M, int H, int I);
int dxa_dit_blocks_fwd(float* h, const float* const* weights, int depth, int N, int T1, int H, int heads, int I, float eps,
                       void* workspace, size_t workspace_bytes, dxa_stream_t stream);
/* The fused launch spins on device-wide barriers and therefore needs all its workgroups resident at once; the spin is
 * bounded (~2 s).  *timed_out = 1 if a launch on this stream gave up since the last call (its output is garbage: re-run
 * the request unfused); the barrier state is re-armed.  Synchronises the stream. */
/* The WHOLE DDIM sampler of a single request in one persistent launch (the loop of GaussianDiffusion.ddim_sample_loop,
 * diffusion.py:714-794, over DiT.forward_with_cfg, dit.py:273-311; eta = 0, epsilon prediction, clip_denoised = False): per step
 * x_embedder + conditioning token + positions, the `depth` DiTBlocks, FinalLayer on the action tokens, classifier-free guidance
 * eps = u + s (c - u) and the DDIM update, `steps` times, device-wide barriers in between (same co-residency contract and
 * watchdog as dxa_dit_blocks_fwd).  x [nb, T1-1, A] fp32 in/out; z_emb [N, H] = z_embedder output (N = 2 nb with guidance:
 * [cond; uncond]); t_emb [steps, H] = t_embedder output per step in execution order; coef [steps][4] = sqrt_recip_alphas_cumprod,
 * sqrt_recipm1_alphas_cumprod, alphas_cumprod_prev, 0 per step in execution order; weights as for dxa_dit_blocks_fwd. */
size_t dxa_dit_sample_workspace(int M, int H, int I);
int dxa_dit_sample_fwd(float* x, const float* z_emb, const float* t_emb, const float* pos, const float* x_w, const float* x_b,
                       const float* final_w, const float* final_b, const float* coef, int steps, int A, int nb, int use_cfg,
                       float cfg_scale, const float* const* weights, int depth, int N, int T1, int H, int heads, int I, float eps,
                       void* workspace, size_t workspace_bytes, dxa_stream_t stream);
int dxa_dit_blocks_status(dxa_stream_t stream, int* timed_out);

/* The same sampler with bf16 MFMA operands, for a model SERVED in bfloat16 — how the reference serves action inference
 * (dexbotic/exp/cogact_exp.py:134-138 loads the whole model, DiT head included, with torch_dtype=torch.bfloat16 and
 * inference_action, cogact_arch.py:149-204, runs it without the training forward's autocast(float32) of :133).  Residual stream,
 * LayerNorm statistics, attention, GELU and accumulation stay fp32; the products read bf16 weights and bf16 copies of their
 * input activations (what every nn.Linear of the reference's bf16 head reads).
 *   dxa_dit_bf16_pack: `weights` as for dxa_dit_blocks_fwd (device array of depth*8 fp32 pointers) -> `packed`, an arena of
 *   dxa_dit_bf16_pack_bytes(depth, H, I) bytes (256-byte aligned) holding, per block, the four matrices rounded to bf16 in the
 *   MFMA operand order ([16 output columns][32 k] tiles of 1 KiB) and sum_k W[n, k] of the two LayerNorm-fed matrices; and `table`,
 *   a DEVICE array of depth*10 pointers (packed qkv_w, qkv_b, packed proj_w, proj_b, packed fc1_w, fc1_b, packed fc2_w, fc2_b,
 *   qkv_wsum, fc1_wsum) that dxa_dit_sample_bf16_fwd takes as `packed_table`.  Re-pack when the fp32 weights change.
 *   dxa_dit_sample_bf16_fwd: arguments, limits (N*T1 <= 48 rows), co-residency contract and watchdog (dxa_dit_blocks_status) as
 *   dxa_dit_sample_fwd; the workspace (dxa_dit_sample_bf16_workspace bytes) must be 256-byte aligned. */
size_t dxa_dit_bf16_pack_bytes(int depth, int H, int I);
int dxa_dit_bf16_pack(const float* const* weights, int depth, int H, int I, void* packed, size_t packed_bytes, const void** table,
                      dxa_stream_t stream);
size_t dxa_dit_sample_bf16_workspace(int M, int H, int I);
int dxa_dit_sample_bf16_fwd(float* x, const float* z_emb, const float* t_emb, const float* pos, const float* x_w, const float* x_b,
                            const float* final_w, const float* final_b, const float* coef, int steps, int A, int nb, int use_cfg,
                            float cfg_scale, const void* const* packed_table, int depth, int N, int T1, int H, int heads, int I,
                            float eps, void* workspace, size_t workspace_bytes, dxa_stream_t stream);

/* The bf16 sampler for MemVLA's DiT with perceptual attention (dexbotic/model/memvla/action_model/dit.py:136-185: every block is
 * x + attn(norm1 x); x + MHA(norm3 x, per, per); x + mlp(norm2 x), and MemVLAForCausalLM.inference_action, memvla_arch.py:666-746,
 * samples with it): three more phases per block in the same persistent launch — the query projection (the q rows of
 * nn.MultiheadAttention's packed in_proj, norm3's affine folded into the packed copy), attention of each (row, head) over the
 * request's P perceptual keys / values, out_proj + residual.
 *   dxa_dit_bf16_pack_per: `weights` = device array of depth*14 fp32 pointers (the 8 of dxa_dit_bf16_pack, then per_attn.in_proj_weight
 *   [3H, H], per_attn.in_proj_bias, per_attn.out_proj.weight, per_attn.out_proj.bias, norm3.weight, norm3.bias); `table` = device array
 *   of depth*16 pointers; arena of dxa_dit_bf16_pack_per_bytes bytes.
 *   dxa_dit_sample_bf16_per_fwd: + `per_kv` [depth][N][P][2][H] fp32, the keys / values of every block projected from the request's
 *   perceptual tokens by rows [H:3H] of in_proj (they do not depend on the DDIM step; the caller computes them once per request),
 *   64 <= P <= 256, P % 64 == 0.  Everything else as dxa_dit_sample_bf16_fwd. */
size_t dxa_dit_bf16_pack_per_bytes(int depth, int H, int I);
int dxa_dit_bf16_pack_per(const float* const* weights, int depth, int H, int I, void* packed, size_t packed_bytes, const void** table,
                          dxa_stream_t stream);
int dxa_dit_sample_bf16_per_fwd(float* x, const float* z_emb, const float* t_emb, const float* pos, const float* x_w, const float* x_b,
                                const float* final_w, const float* final_b, const float* coef, int steps, int A, int nb, int use_cfg,
                                float cfg_scale, const void* const* packed_table, const float* per_kv, int P, int depth, int N, int T1,
                                int H, int heads, int I, float eps, void* workspace, size_t workspace_bytes, dxa_stream_t stream);

/* ---- KV-cached decode step in one persistent launch (csrc/decode_fused.hip) -------------------------------------------------
 * ONE new token of ONE sequence through every decoder layer and the final RMSNorm — the use_cache=True single-token pass of HF
 * Qwen2Model that GenerationMixin.generate drives for the discrete-action policies (dexbotic/model/discrete_vla/
 * discrete_vla_arch.py:24-50, dexbotic/model/dexbotic_arch.py:429-496): a grid of co-resident workgroups walks
 * [RMSNorm] qkv + bias | RoPE + cache append + attention | o + residual | [RMSNorm] gate / up + SiLU * up | down + residual
 * per layer with device-wide barriers in between (co-residency contract and watchdog of dxa_dit_blocks_fwd; dxa_decode_status
 * reports a launch that gave up).  bf16 weights, bf16 rounding points of the unfused kernels, fp32 accumulation.
 *   layers: DEVICE array of n_layers * 9 pointers — input_layernorm.weight [d], q|k|v weight [(Hq + 2 Hkv) D, d], q|k|v bias,
 *           o_proj.weight [d, Hq D], post_attention_layernorm.weight [d], gate|up weight [2 F, d], down_proj.weight [d, F] (all
 *           bf16, 16-byte aligned), key cache and value cache of the layer [Hkv, max_len, D] bf16 (post-RoPE keys).
 *   x_in [d] bf16: embedding of the new token; out [d] bf16: hidden state after the final norm (final_norm_w [d] bf16).
 *   cos_row / sin_row [D / 2] fp32: the rotary table row of the token's position.  slot: cache position the new key / value are
 *   written to; the token attends to the cached positions [kv_lo, slot) and itself.  workspace: dxa_decode_step_workspace bytes,
 *   256-byte aligned. */
typedef struct dxa_decode_desc {
  const void* const* layers;
  const void* x_in; void* out; const void* final_norm_w;
  const float* cos_row; const float* sin_row;
  void* workspace; size_t workspace_bytes;
  int32_t n_layers, d, Hq, Hkv, D, F, slot, kv_lo, max_len;
  float eps;
} dxa_decode_desc;
size_t dxa_decode_step_workspace(int d, int Hq, int Hkv, int D, int F);
int dxa_decode_step(const dxa_decode_desc* desc, dxa_stream_t stream);
int dxa_decode_status(dxa_stream_t stream, int* timed_out);

#ifdef __cplusplus
}
#endif
#endif /* DEXBOTIC_AMD_H_ */
