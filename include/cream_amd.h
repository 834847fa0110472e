/*
 * cream_amd.h — C ABI of libcream_amd.so (MI355X / gfx950 hot path of microsoft/Cream).
 *
 * Every entry point is `extern "C"`, takes plain device/host pointers, sizes and a
 * `hipStream_t` passed as `void*`, returns 0 on success or a negative CREAM_ERR_* code,
 * never throws, never synchronises, never allocates device memory.  No torch types.
 *
 * Each declaration cites the reference interface (file:line under the reference
 * checkout) that it replaces.  INTEGRATION.md shows the binding a maintainer of the
 * reference would add on top of this header.
 */
#ifndef CREAM_AMD_H_
#define CREAM_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ---------------------------------------------------------------- */
#define CREAM_OK               0
#define CREAM_ERR_BAD_ARG     -1  /* null pointer, negative size, unsupported stride    */
#define CREAM_ERR_BAD_DTYPE   -2  /* dtype enum not supported by this entry point        */
#define CREAM_ERR_LAUNCH      -3  /* hipLaunchKernel reported an error (see hipGetLastError) */
#define CREAM_ERR_TOO_LARGE   -4  /* shape exceeds what the kernel family supports       */

/* ---- element types (the "dtype" argument) ------------------------------------------ */
#define CREAM_F32   0
#define CREAM_F16   1
#define CREAM_BF16  2   /* not in the reference (AT_DISPATCH_FLOATING_TYPES_AND_HALF only) */
#define CREAM_F64   3

/* Version handshake.  Replaces `version()` of the pybind module `rpe_index_cpp`
 * (iRPE/DeiT-with-iRPE/rpe_ops/rpe_index.cpp:126-128; asserted == "1.2.0" at
 * rpe_ops/rpe_index.py:5-8).  Returns a static string. */
const char* cream_version(void);

/* Build tag of this library ("gfx950;hipcc ...") for logs. */
const char* cream_build_info(void);

/* ---- rpe_index: bucketed relative-position gather / scatter ----------------------- */

/* Y[b,h,i,j] = in[b,h,i, idx[i,j]]          (device pointers)
 * Replaces rpe_index_forward_gpu + rpe_index_forward_gpu_kernel
 * (iRPE/DeiT-with-iRPE/rpe_ops/rpe_index_cuda.cu:24-40,54-94).
 *   y    : (B,H,Lq,Lk) contiguous, caller-allocated
 *   in   : (B,H,Lq,nb) with ELEMENT strides s0..s3 (the reference passes
 *          input.strides(), rpe_index_cuda.cu:76,91 — iRPE hands over a transposed
 *          view, irpe.py:639-642)
 *   idx  : (Lq,Lk) contiguous int32 in [0,nb)  (not range-checked, as in the reference)
 * Pure data movement: results are bit-exact for every dtype. */
int cream_rpe_index_fwd(void* y, const void* in, const int32_t* idx,
                        int B, int H, int Lq, int Lk, int nb,
                        int64_t s0, int64_t s1, int64_t s2, int64_t s3,
                        int dtype, void* stream);

/* gin[b,h,i,u] (+)= sum_{j : idx[i,j]==u} gout[b,h,i,j]     (device pointers)
 * Replaces rpe_index_backward_gpu + rpe_index_backward_gpu_kernel
 * (rpe_index_cuda.cu:42-52,96-140).  accumulate != 0: `gin` is updated in place exactly
 * like the reference (whose caller zero-fills it, rpe_ops/rpe_index.py:51);
 * accumulate == 0: `gin` is overwritten and need not be initialised (saves the memset
 * and one read of gin).  All three tensors contiguous.  Unlike the reference's global atomics the summation order is FIXED:
 * gin_initial + g[j=0] + g[j=1] + ... in ascending j — bit-reproducible, and
 * bit-identical to a sequential CPU loop for f32/f64.  f16/bf16 accumulate in f32
 * and round once. */
int cream_rpe_index_bwd(void* gin, const void* gout, const int32_t* idx,
                        int B, int H, int Lq, int Lk, int nb,
                        int dtype, int accumulate, void* stream);

/* Host (CPU-tensor) entry points — the reference module also exports forward_cpu /
 * backward_cpu (rpe_index.cpp:8-73, 82-124) and BASELINE config 1 runs on them.
 * These are a separate implementation for HOST pointers, multi-threaded with
 * std::thread over (b,h,i) rows; they are never used for device tensors.
 * Same contracts as above (host fwd requires contiguous `in`, like the reference's
 * input.contiguous() at rpe_index.cpp:28). */
int cream_rpe_index_fwd_host(void* y, const void* in, const int32_t* idx,
                             int B, int H, int Lq, int Lk, int nb, int dtype);
int cream_rpe_index_bwd_host(void* gin, const void* gout, const int32_t* idx,
                             int B, int H, int Lq, int Lk, int nb, int dtype);

/* ---- fused attention with AutoFormer's 2-D relative position bias -------------------- */

/* Padded token count used by the side buffers below: N rounded up to a multiple of 32. */
int cream_attn_rpe2d_padded_len(int N);

/* Number of partial table-gradient blocks cream_attn_rpe2d_bwd writes for a (B, H) problem on the current device:
 * one per persistent workgroup of its dK/dV kernel, min(B*H, compute units). */
int cream_attn_rpe2d_dtab_parts(int B, int H);

/* Which backward runs for the AutoFormer geometry (N = 197, 14 x 14 grid, max_relative_position 14) in bf16:
 * 0 = the two-launch backward; 1 = the one-pass kernel (csrc/attn_rpe2d_bwd1.hpp: K, V, Q, dO of one (b, h) whole in LDS, P / dS
 * exchanged between the query-tile and key-tile owners through LDS; the side buffers dlt / qe / de / delta are not used as
 * such — the first 32 KB of dlt carry the bf16 operand images of the tables); 2 = the same pass with the two roles on separate
 * waves (csrc/attn_rpe2d_bwd2.hpp: 7 query-tile owners + 5 waves sharing the 14 key-side jobs, the table gradients accumulated in
 * registers over all items of a workgroup and written once) — the DEFAULT.  dq / dk / dv of modes 1 and 2 are bit-identical; the
 * per-workgroup table-gradient partials of mode 2 are the sums mode 0 forms (one fp32 chain per workgroup).  onepass < 0 only
 * queries.  Returns the previous setting; the initial one comes from CREAM_ATTN_BWD1 in the environment (default 2).
 * (What autograd derives for multihead_super.py:135-154 in every mode; a switch for same-box A/B measurements.) */
int cream_attn_rpe2d_bwd_mode(int onepass);

/* The attention core of AttentionSuper.forward between the qkv and proj GEMMs
 * (AutoFormer/model/module/multihead_super.py:135-154) with the relative position
 * embeddings of RelativePosition2D_super.forward (multihead_super.py:40-66) folded in:
 *     A[i,j] = scale * ( q_i.k_j + q_i.(Tkv[iv[i,j]] + Tkh[ih[i,j]]) )
 *     P = softmax_j(A)          (attention dropout 0 — the supernet recipe's value; cream_attn_rpe2d_fwd_drop below takes p > 0)
 *     O_i = sum_j P[i,j] * ( v_j + Tvv[iv[i,j]] + Tvh[ih[i,j]] )
 * for every (b, h); head_dim is 64 (supernet_transformer.py:243).  iv/ih are the
 * reference's index matrices for a gh x gw token grid behind one class token
 * (N = gh*gw + 1, bucket 0 for the class row/column, clamp(+-mr) + mr + 1 otherwise); they
 * are generated in-kernel from (gh, gw, mr), never read from memory.
 *   q, k, v : element (b, n, h, d) at  ptr[b*sb + n*sn + h*sh + d]  (strides in elements;
 *             the three may alias one (B, N, 3, H, 64) qkv buffer), dtype bf16 or f32
 *   tkv,tkh,tvv,tvh : (2*mr+2, 64) fp32 tables, row stride ldt elements
 *   out     : (B, N, H, 64) contiguous, same dtype as q
 *   lse     : (B, H, N) fp32       log-sum-exp of the scaled logits   (saved for backward)
 *   sp      : (B, H, 64, NP) same dtype as q: bucket sums S'^T, rows 0..31 vertical table,
 *             32..63 horizontal table, NP = cream_attn_rpe2d_padded_len(N)  (for backward; NULL: not
 *             written — a forward that no backward follows)
 * Limits: N <= 256, gh + gw + 1 <= 32, 2*mr + 2 <= 32, 16-byte aligned rows
 * (CREAM_ERR_TOO_LARGE / CREAM_ERR_BAD_ARG otherwise).  dtype: CREAM_BF16 (bf16 MFMA, fp32
 * accumulate and softmax) or CREAM_F32 (fp32 MFMA, exact fp32 products). */
int cream_attn_rpe2d_fwd(void* out, float* lse, void* sp,
                         const void* q, const void* k, const void* v,
                         int64_t sb, int64_t sn, int64_t sh,
                         const float* tkv, const float* tkh, const float* tvv, const float* tvh,
                         int ldt, int B, int H, int N, int gh, int gw, int mr,
                         float scale, int dtype, void* stream);

/* Backward of cream_attn_rpe2d_fwd (what autograd derives for multihead_super.py:135-154 and
 * the index_put of RelativePosition2D_super's table lookup, multihead_super.py:64):
 *   dq, dk, dv : gradients, element (b, n, h, d) at ptr[b*dsb + n*dsn + h*dsh + d] (may alias
 *                one (B, N, 3, H, 64) buffer), same dtype as q
 *   dtab       : (cream_attn_rpe2d_dtab_parts(B, H), 4, 32, 64) fp32, partial gradients of [tkv, tkh, tvv, tvh]
 *                (rows >= 2*mr+2 are zero): the tables are shared by every (b, h), each persistent workgroup of the
 *                dK/dV kernel accumulates over its items and writes one block; the caller sums over the first
 *                axis — a fixed-order reduction instead of atomics
 *   dlt, qe, de, delta : scratch handed from the first to the second launch:
 *                dlt (B,H,64,NP) and qe, de (B,H,NP,32) in q's dtype, delta (B,H,NP) fp32
 *   dout, out  : (B, N, H, 64) contiguous, q's dtype;  lse, sp: as written by the forward
 * Two kernels are enqueued on `stream` (dQ; then dK, dV and the table gradients).  No
 * atomics: results are bit-reproducible run to run. */
int cream_attn_rpe2d_bwd(void* dq, void* dk, void* dv, int64_t dsb, int64_t dsn, int64_t dsh,
                         float* dtab, void* dlt, void* qe, void* de, float* delta,
                         const void* dout, const void* out, const float* lse, const void* sp,
                         const void* q, const void* k, const void* v,
                         int64_t sb, int64_t sn, int64_t sh,
                         const float* tkv, const float* tkh, const float* tvv, const float* tvh,
                         int ldt, int B, int H, int N, int gh, int gw, int mr,
                         float scale, int dtype, void* stream);

/* bf16 OPERAND IMAGES of the four tables (the rows of RelativePosition2D_super's embeddings_table_v / _h,
 * multihead_super.py:22-26, as the matrix cores read them): cream_attn_rpe2d_table_image_bytes() bytes (32 KB),
 *   [key rows (64 u' x 64 d) | key^T (64 d x 64 u') | value rows | value^T],  u' = bucket of the vertical table (0..31)
 *   or 32 + bucket of the horizontal table, rows >= 2*mr+2 zero.
 * cream_attn_rpe2d_table_images builds them in one small launch; cream_adamw_step keeps them current for free when the
 * caller registers them as the bf16 copies of the four table parameters (mir = rows, mir_t = transposed at ld 64:
 * cream_amd/autoformer/block.py: BlockOperands) — the layout is exactly that of the GEMM operand copies.
 * The *_img entry points are cream_attn_rpe2d_fwd / _bwd with such an image (timg; NULL = the plain entry points):
 *   - forward, AutoFormer geometry (N = 197, 14 x 14 grid, mr = 14) in bf16: runs csrc/attn_rpe2d_fwd2.hpp — 32-key tiles with
 *     an online (lazy-maximum) softmax, two 7-wave half-workgroups per CU in ping-pong (one multiplies while the other does
 *     the element-wise / memory work), K / V by LDS-DMA; without an image (or with cream_attn_rpe2d_fwd_mode(0)) the
 *     whole-row-block kernel attn_rpe2d_fwd14 runs.  Same outputs up to the rounding of the softmax numerators' reference
 *     point (both within the bf16 bounds of tests/test_attn_gpu.py).
 *   - backward (one-pass kernel): reads the image instead of building one into dlt with an extra launch.
 * cream_attn_rpe2d_fwd_mode(1 | 0): as above; < 0 queries; returns the previous value; initial: CREAM_ATTN_FWD2, else 1. */
int64_t cream_attn_rpe2d_table_image_bytes(void);
int cream_attn_rpe2d_table_images(void* img, const float* tkv, const float* tkh, const float* tvv, const float* tvh, int ldt,
                                  int mr, void* stream);
int cream_attn_rpe2d_fwd_mode(int fwd2);
int cream_attn_rpe2d_fwd_img(void* out, float* lse, void* sp,
                             const void* q, const void* k, const void* v,
                             int64_t sb, int64_t sn, int64_t sh,
                             const float* tkv, const float* tkh, const float* tvv, const float* tvh,
                             int ldt, const void* timg, int B, int H, int N, int gh, int gw, int mr,
                             float scale, int dtype, void* stream);
int cream_attn_rpe2d_bwd_img(void* dq, void* dk, void* dv, int64_t dsb, int64_t dsn, int64_t dsh,
                             float* dtab, void* dlt, void* qe, void* de, float* delta,
                             const void* dout, const void* out, const float* lse, const void* sp,
                             const void* q, const void* k, const void* v,
                             int64_t sb, int64_t sn, int64_t sh,
                             const float* tkv, const float* tkh, const float* tvv, const float* tvh,
                             int ldt, const void* timg, int B, int H, int N, int gh, int gw, int mr,
                             float scale, int dtype, void* stream);

/* The same two entry points with ATTENTION DROPOUT between the softmax and the two value-side products
 * (multihead_super.py:145 `attn = self.attn_drop(attn)`): P[i,j] -> keep[i,j] P[i,j] / (1 - dropout_p) for the P.V product and
 * the value-side bucket sums alike; lse is that of the undropped softmax.  keep is a pure function of (dropout_seed, b * H + h,
 * i, j) — a 32-bit counter-based mix (csrc/attn_common.hpp drop_keep; cream_amd/irpe_fused.py dropout_keep_mask restates it),
 * kept iff hash >= round(dropout_p 2^32) — so the backward regenerates the forward's mask from the same (dropout_p,
 * dropout_seed): no N x N mask exists in memory, and a run is reproducible from its seeds.  0 <= dropout_p < 1
 * (CREAM_ERR_BAD_ARG otherwise); dropout_p == 0 IS cream_attn_rpe2d_fwd_img / _bwd_img.  With dropout_p > 0 every geometry and
 * dtype runs the tile-streamed forward and the two-launch backward (whose score tiles are in registers where the mask applies);
 * timg is not used then. */
int cream_attn_rpe2d_fwd_drop(void* out, float* lse, void* sp,
                              const void* q, const void* k, const void* v,
                              int64_t sb, int64_t sn, int64_t sh,
                              const float* tkv, const float* tkh, const float* tvv, const float* tvh,
                              int ldt, const void* timg, int B, int H, int N, int gh, int gw, int mr,
                              float scale, float dropout_p, uint32_t dropout_seed, int dtype, void* stream);
int cream_attn_rpe2d_bwd_drop(void* dq, void* dk, void* dv, int64_t dsb, int64_t dsn, int64_t dsh,
                              float* dtab, void* dlt, void* qe, void* de, float* delta,
                              const void* dout, const void* out, const float* lse, const void* sp,
                              const void* q, const void* k, const void* v,
                              int64_t sb, int64_t sn, int64_t sh,
                              const float* tkv, const float* tkh, const float* tvv, const float* tvh,
                              int ldt, const void* timg, int B, int H, int N, int gh, int gw, int mr,
                              float scale, float dropout_p, uint32_t dropout_seed, int dtype, void* stream);

/* ---- the two ends of the supernet around the block stack (csrc/stem_tail.hip) --------------------
 * Reference: Vision_TransformerSuper.forward_features, AutoFormer/model/supernet_transformer.py:147-172
 * (patch embedding embedding_super.py:27-40, class token + position embedding :150-155, final LayerNorm and
 * token mean :166-170 for pre_norm / gp models). */

/* patches (B*gh*gw, C*ph*pw) bf16 = unfold(img (B, C, H, W) fp32): the GEMM operand of the stride = kernel conv. */
int cream_im2patch(void* patches, const float* img, int B, int C, int H, int W, int ph, int pw, void* stream);

/* x0 (B, N, E) fp32: row 0 = cls[:E] + pos[0, :E], row n = y[b, n-1, :] (bf16, (B, N-1, E)) + pos[n, :E];
 * pos may be NULL (abs_pos off); ld_pos = row stride of the position embedding (its super width). */
int cream_stem_assemble(float* x0, const void* y, const float* cls, const float* pos, int64_t ld_pos, int B, int N, int E,
                        void* stream);

/* dy (B, N-1, E) bf16 = dx0[:, 1:, :]; psum (cream_stem_bwd_chunks(B), N, E) fp32 = sums of dx0 over chunks of 16
 * images (position-embedding / class-token gradient = their sum, fixed order). */
int cream_stem_bwd_chunks(int B);
int cream_stem_bwd(void* dy, float* psum, const float* dx0, int B, int N, int E, void* stream);

/* pooled (B, E) fp32 = mean over tokens 1.. of LayerNorm(x1 + sample_scale[b] * f) (f bf16 may be NULL);
 * also writes xm (B, E) = the token mean before the affine map (-> gamma gradient), mean / rstd (B*N) and
 * uses part (B, cream_tail_chunks(N), E) as scratch.  Two launches. */
int cream_tail_chunks(int N);
int cream_tail_fwd(float* pooled, float* xm, float* part, float* mean, float* rstd, const float* x1, const void* f,
                   const float* sample_scale, const float* gamma, const float* beta, int B, int N, int E, float eps,
                   void* stream);

/* Backward of cream_tail_fwd given g = d loss / d pooled (B, E): dx (B*N, E) fp32 residual-stream gradient,
 * dx_scaled = bf16(sample_scale[b] * dx) (the gradient of f) and partial (cream_ln_partials(), E) = column
 * sums of dx_scaled (bias gradient of the projection that produced f).  gamma / beta gradients are
 * sum_b g * xm and sum_b g (caller). */
int cream_tail_bwd(float* dx, void* dx_scaled, float* partial, const float* g, const float* x1, const void* f,
                   const float* mean, const float* rstd, const float* gamma, const float* sample_scale, int B, int N,
                   int E, void* stream);

/* ---- fused attention with iRPE (contextual mode) on queries, keys and values ------------------
 * Reference: RPEAttention.forward (iRPE/DeiT-with-iRPE/rpe_vision_transformer.py:68-97) between the
 * qkv and proj linears, with iRPE.forward_rpe_transpose / forward_rpe_no_transpose
 * (irpe.py:585-687) and the rpe_index gather (rpe_ops/rpe_index_cuda.cu:24-40) folded in — see
 * csrc/irpe_attn.hip for the formulas.  bf16 operands, fp32 accumulation and softmax, head_dim 64,
 * at most 64 buckets.  A table pointer (wq / wk / wv) that is NULL switches that term off. */
typedef struct cream_irpe_attn_desc {
    const void *q, *k, *v;          /* bf16; element (b, n, h, :) at ptr[b*sb + n*sn + h*sh]            */
    int64_t sb, sn, sh;
    void* out;                      /* (B, L, H, 64) bf16                                               */
    float* lse;                     /* (B, H, L)   log-sum-exp of the logits (fwd: out, bwd: in)        */
    void* sv;                       /* (B, H, NP, 64) bf16 value-side bucket sums (fwd: out, needed with wv) */
    const float *wq, *wk, *wv;      /* lookup_table_weight of rpe_q / rpe_k (H', 64, nb) and rpe_v (H', nb, 64) */
    int64_t wq_hs, wk_hs, wv_hs;    /* element stride between heads, 0 for shared_head                   */
    const uint8_t *idq, *idk, *idv;       /* (NP, NP) query-major bucket ids: idk[i][j] = bucket_k[i][j],
                                             idv[i][j] = bucket_v[i][j], idq[i][j] = bucket_q[j][i]      */
    const uint8_t *idq_t, *idk_t, *idv_t; /* the same matrices transposed (key-major), backward only     */
    int B, H, L, NP, nb;            /* NP = cream_irpe_padded_len(L)                                     */
    float scale;
    /* backward only */
    const void* dout;               /* (B, L, H, 64) bf16                                               */
    void *dq, *dk, *dv;             /* bf16; element (b, n, h, :) at ptr[b*dsb + n*dsn + h*dsh]          */
    int64_t dsb, dsn, dsh;
    float* delta;                   /* (B, H, NP) scratch                                               */
    void *lkg, *gg;                 /* (B, H, NP, 64) bf16 scratch (needed with wk / wv)                 */
    void *dlk, *dlq;                /* (B, H, NP, 64) bf16 out: bucket gradients of rpe_k / rpe_q lookups
                                       (dlq also serves as scratch inside the call: the rpe_q lookup rows of the
                                       keys live there between the pre-pass and the end of the second launch) */
    /* bias mode of rpe_q / rpe_k (irpe.py:622-624: lookup_table_bias (H', nb), no dependence on q / k): give the
     * table here INSTEAD of wq / wk; its gradient is the sum of the dlq / dlk rows over batch and tokens (caller) */
    const float *bq, *bk;
    int64_t bq_hs, bk_hs;
    /* causal != 0: key j takes part for query i only if j <= i — the additive upper-triangular -inf mask of the text
     * towers (TinyCLIP/src/open_clip/model.py:756-762) without an (L, L) tensor; 0: full attention */
    int32_t causal, reserved;
    /* attention dropout (rpe_vision_transformer.py:86, multihead_super.py:145 `attn = self.attn_drop(attn)`):
     * P[i,j] -> keep[i,j] P[i,j] / (1 - dropout_p) after the softmax, for the value product and the value-side
     * bucket sums alike; keep is a pure function of (dropout_seed, b, h, i, j) — the SAME seed must be given to
     * cream_irpe_attn_bwd.  0 <= dropout_p < 1 (0: none); L <= 65535 with dropout. */
    float dropout_p;
    uint32_t dropout_seed;
} cream_irpe_attn_desc;

int cream_irpe_padded_len(int L);

/* int32 (Lq, Lk) bucket ids (irpe.py:523-583) -> zero-padded uint8 (NP, NP) holding 2 * id (the byte
 * offset of the bucket in a bf16 lookup row; ids < 64); transpose != 0 converts the transposed table (dst row j
 * = column j of ids).  Done once per (table, device) by the caller.  The matrix is an OPAQUE operand of the two
 * calls below: inside every 32-byte group of a row (one streamed tile of 32 partners) the bytes are in the order
 * the lanes of the kernels read them (byte 16 g + r = partner (r & 3) + 8 (r >> 2) + 4 g of the tile), so that a
 * lane's 16 ids of a tile are one 16-byte load. */
int cream_irpe_bucket_bytes(uint8_t* dst, const int32_t* ids, int Lq, int Lk, int NP, int transpose, void* stream);

/* out, lse (and sv) from q, k, v.  One launch. */
int cream_irpe_attn_fwd(const cream_irpe_attn_desc* d, void* stream);

/* dq, dk, dv and the bucket-gradient rows dlk, dlq from dout (+ out, lse of the forward).  Two launches
 * (queries own lanes; keys own lanes) behind a small pre-pass when wq is given ((k * scale) Wq rows of all keys);
 * no global atomics. */
int cream_irpe_attn_bwd(const cream_irpe_attn_desc* d, void* stream);

/* Per-(b,h) table gradient  out[b*H+h][a][c] = mul * sum_n X[b,n,h][a] * Y[b,n,h][c]  (64 x 64 fp32):
 *   d lookup_table_weight(rpe_k) = (scale q)^T dlk, (rpe_q) = (scale k)^T dlq, (rpe_v) = sv^T dout;
 * the caller sums over b (and h for shared tables).  X, Y bf16 with element strides (sb, sn, sh). */
int cream_irpe_table_grad(float* out, const void* x, int64_t xsb, int64_t xsn, int64_t xsh, const void* y, int64_t ysb,
                          int64_t ysn, int64_t ysh, int B, int H, int L, float mul, void* stream);

/* ---- HBM-bound passes of one supernet transformer block ------------------------------------
 * Reference: TransformerEncoderLayer.forward, AutoFormer/model/supernet_transformer.py:251-287
 * (pre-norm block), LayerNormSuper.forward (model/module/layernorm_super.py:26-37), gelu in
 * fp32 (supernet_transformer.py:14-16), DropPath (model/utils.py:68-98).  The residual stream
 * is fp32 (as under the reference's autocast), GEMM operands are bf16.  All matrices are
 * row-major and contiguous with leading dimension = their width.  No atomics anywhere. */

/* y(bf16, M x E) = LayerNorm(x(f32))*gamma + beta over the last dim; mean/rstd (M) saved. */
int cream_ln_fwd(void* y, float* mean, float* rstd, const float* x, const float* gamma,
                 const float* beta, int M, int E, float eps, void* stream);

/* Residual add fused with the following LayerNorm (supernet_transformer.py:266-275: the
 * attention branch's  x = residual + drop_path(x)  and the  ffn_layer_norm  that reads it):
 * xsum(f32) = x(f32) + sample_scale[row / rows_per_sample] * res(bf16)  (scale may be NULL = 1),
 * y(bf16) = LayerNorm(xsum), mean/rstd of xsum saved. */
int cream_add_ln_fwd(float* xsum, void* y, float* mean, float* rstd, const float* x, const void* res,
                     const float* sample_scale, int rows_per_sample, const float* gamma,
                     const float* beta, int M, int E, float eps, void* stream);

/* Number of row slabs ln_bwd reduces over: partial must hold cream_ln_partials()*3*E floats. */
int cream_ln_partials(void);

/* dx(f32) = dres + dLayerNorm(dy(bf16)); if dx_scaled != NULL also writes
 * bf16(dx * sample_scale[row / rows_per_sample]) (sample_scale may be NULL = 1): the gradient
 * entering the previous residual branch with its drop-path scale.  dres may be NULL.
 * partial[p][0][:] / partial[p][1][:] = per-slab sums for dgamma / dbeta; partial[p][2][:] =
 * column sums of the bf16 values written to dx_scaled (= bias gradient of the projection
 * whose output gradient dx_scaled is; zeros when dx_scaled is NULL).  The caller adds the
 * slabs (cream_grad_finalize): fixed order. */
int cream_ln_bwd(float* dx, void* dx_scaled, float* partial, const void* dy, const float* x,
                 const float* mean, const float* rstd, const float* gamma, const float* dres,
                 const float* sample_scale, int rows_per_sample, int M, int E, void* stream);

/* g = gelu(h) / dh = dg * gelu'(h): exact erf GELU evaluated in fp32, bf16 storage; n % 8 == 0 */
int cream_gelu_fwd(void* g, const void* h, int64_t n, void* stream);
int cream_gelu_bwd(void* dh, const void* dg, const void* h, int64_t n, void* stream);

/* out(f32) = x(f32) + sample_scale[i / per_sample] * y(bf16)   (flat index i; scale may be NULL) */
int cream_residual_add(float* out, const float* x, const void* y, const float* sample_scale,
                       int64_t n, int64_t per_sample, void* stream);

/* out(bf16) = sample_scale[i / per_sample] * x(f32) */
int cream_scale_cast(void* out, const float* x, const float* sample_scale, int64_t n,
                     int64_t per_sample, void* stream);

/* Bias gradients: partial[s][c] = sum of a(bf16, M x C)[r][c] over the rows of slab s
 * (cream_colsum_slabs(M) slabs of 512 rows); the caller adds the slabs. */
int cream_colsum_slabs(int M);
int cream_colsum(float* partial, const void* a, int M, int C, void* stream);

/* The same passes with the bias gradient riding along (what autograd's sum-to-size of
 * F.linear's bias produces, Linear_super.py:52-54 / qkv_super.py:49-51), 128-row slabs:
 * partial[s][c] = sum over the rows of slab s of the bf16-ROUNDED values written (so the result
 * equals cream_colsum of the output), cream_colsum128_slabs(M) slabs, C % 8 == 0.
 *   cream_gelu_bwd_colsum    dh(bf16, M x C) = dg * gelu'(h)                    (fc1 bias)
 *   cream_scale_cast_colsum  out(bf16, M x C) = sample_scale[row / rows_per_sample] * x(f32)
 *                                                                                  (fc2 bias)
 *   cream_colsum128          column sums only (qkv bias: dqkv leaves the attention kernels) */
int cream_colsum128_slabs(int M);
int cream_colsum128(float* partial, const void* a, int M, int C, void* stream);
int cream_gelu_bwd_colsum(void* dh, float* partial, const void* dg, const void* h, int M, int C,
                          void* stream);
int cream_scale_cast_colsum(void* out, float* partial, const float* x, const float* sample_scale,
                            int rows_per_sample, int M, int C, void* stream);

/* ---- dense projections of the weight-entangled Linear layers ---------------------------------
 * LinearSuper.forward / qkv_super.forward = F.linear(x, W[:out, :in], b[:out])
 * (AutoFormer/model/module/Linear_super.py:38-54,71-81; qkv_super.py:45-55,72-83), the erf-GELU of
 * the Mlp (supernet_transformer.py:14-16,275-285) and their autograd adjoints, on hand-written MFMA
 * kernels (csrc/gemm_mfma.hpp: direct-to-LDS operand loads, v_mfma_f32_32x32x16_bf16, fused
 * epilogues).  bf16 operands, fp32 accumulation, bf16 results.  The active block of a super weight is
 * read in place: `w` points at element [0][0] of the operand copy, ld = its row stride; N = active
 * out-features, K = active in-features.  Activations are contiguous (row stride = width).
 * N % 8 == K % 8 == ld % 8 == 0, 16-byte aligned pointers.
 *   cream_linear_fwd          out(M x N) = x(M x K) . W(N x K)^T + bias(N)          (bias may be NULL)
 *   cream_linear_fwd_seg      the same with the rows of W in up to 3 segments of `nseg` rows,
 *                             `nseg_stride` elements apart: the de-interleaved [q | k | v] copy of
 *                             the qkv super weight (row n of the product = row n % nseg of segment
 *                             n / nseg) — replaces the torch.cat row gather of qkv_super.py:75
 *   cream_linear_gelu_fwd     with h = bf16(x . W^T + bias): g = gelu(float(h)) and gp = gelu'(float(h)), both
 *                             bf16, in one pass (fc1 + the fp32 gelu of supernet_transformer.py:14-16, on the
 *                             bf16-rounded h exactly as under the reference's autocast; the derivative shares
 *                             Phi and the exponential with the value and is what the backward needs of h);
 *                             gp == NULL: only g is written (a forward without a backward: evaluation sweeps
 *                             of supernet_engine.py:114-160, TinyCLIP's frozen teacher) — half the output bytes
 *   cream_linear_dgrad        dx(M x K) = dy(M x N) . W(N x K); `wt` is the TRANSPOSED copy W^T
 *                             (K rows, N contiguous, row stride ldwt)
 *   cream_linear_dgrad_seg    the same with the contraction index in segments of `kseg` (the
 *                             transposed [q | k | v] copies, `kseg_stride` elements apart; kseg % 64 == 0)
 *   cream_linear_dgrad_mul    dh(M x K) = (dy . W) * factor (bf16, M x K: the saved gelu'(h)) and the column
 *                             sums of dh per 128-row slab: colsum_parts[slab][K] (fc2 dgrad + GELU backward +
 *                             fc1 bias gradient partials, cream_gemm_rows_per_colsum_slab() rows per slab)
 *   cream_linear_wgrad_parts  parts[s](N x K) fp32 = dy_s^T x_s over S slices of the M rows (slices of
 *                             whole 64-row steps, as even as possible; added by cream_grad_finalize);
 *                             bias_parts[s](N) = column sums of dy_s — the bias gradient rides on the
 *                             same kernel (NULL: not wanted).  cream_linear_wgrad_splits gives the S
 *                             this library uses for a problem (~1 workgroup per CU, at most 16). */
int cream_gemm_rows_per_colsum_slab(void);

/* Tile choice of the NT products (Linear_super.py:38-54, qkv_super.py:45-55 and their dgrads): where the 256 x 256 macro
 * tile (8 waves, 128 KB of LDS, one persistent workgroup per CU) replaces the 128-wide tiles (two to three workgroups per CU).
 * 0 = nowhere; 1 = wide outputs (N >= 960: qkv / fc1 forward, fc2 dgrad); 2 = long contractions (K >= 1152: fc2 forward,
 * fc1 dgrad, qkv dgrad at embed dim >= 384) — the default: 6-14 % faster there with cold operands; 3 = K >= 960.
 * on < 0 only queries; returns the previous setting; the initial one comes from CREAM_GEMM_NT256 in the environment.
 * Results are identical in every mode (same products, same epilogues). */
int cream_gemm_nt256(int on);
/* Schedule of the NT products (same reference functions): the counted-vmcnt, phase-interleaved kernel of csrc/gemm_nt8.hpp
 * (256 x 256 tiles, 8 waves in two staggered rows, a whole K-tile of LDS-DMA requests in flight across every barrier, K-tile
 * stream continuous over output tiles, barrier-free per-wave epilogue).  0 = never (the kernels above), 1 = wherever its
 * limits allow (31-bit element offsets), 2 = plain and bias products everywhere, the GELU / x gelu' epilogues only on long
 * contractions (K >= 1024); 3 = plain and bias products whose 256-wide column tiles carry at most 1/8 padding; 4 = the forward
 * products (bias epilogue) under the same padding bound — the DEFAULT (no weight-gradient stream runs beside the forward) —
 * plus every product whose output AND contraction are at least 640 wide (DeiT-base / CLIP ViT-B blocks; no AutoFormer shape).
 * mode < 0 only queries; returns the previous setting; the initial one comes from CREAM_GEMM_NT8 in the environment.
 * Forward, bias, bias + GELU results are identical to the other kernels'.  cream_linear_dgrad_mul: when this kernel serves it
 * (modes 1 and 2) dy . W is rounded to bf16 before the multiplication by gelu' (as the reference's two operators do); the
 * two-stage kernels (modes 0, 3, 4 — the default) multiply the fp32 accumulator and round once — the results differ by at
 * most one bf16 rounding of the product. */
/* Round 6: 256 x (E / 2) tiles (column tiles that divide N) for the plain / bias products whose output is 320 / 384 / 448 wide
 * (csrc/gemm_mfma.hip: launch_nt_half).  1 = on, 0 = off, < 0 queries; returns the previous value; initial: CREAM_GEMM_NTHALF. */
int cream_gemm_nthalf(int on);
int cream_gemm_nt8(int mode);
/* Round 6: the two-stage NT kernels with their tile epilogue taken off the memory counters (csrc/gemm_mfma.hpp, "OPT"):
 * bit 0 = LDS-DMA as asm, LDS-only epilogue barriers, side inputs (bias, the x gelu' factor rows) requested under the first
 * K-step, counted vmcnt for the first K-step behind an epilogue — on the 128-wide tiles of the forward products (bias, bias +
 * GELU); bit 3 (8) = also on the backward's (plain store, x gelu'); bit 2 (4) = also on the 256-wide macro tiles; bit 1 = gelu(h) /
 * gelu'(h) of cream_linear_gelu_fwd from a 16 KB LDS table over the bf16 values of h (filled by the same function the direct
 * path evaluates).  Default 3, < 0 queries; returns the previous value; initial: CREAM_GEMM_NTOPT.  Results are bit-identical in every mode
 * (tests/test_block_gpu.py::test_nt_epilogue_variants_are_bit_identical; the table exhaustively over all bf16 h, with the one
 * exception |h| < 2^-125, where gelu(h) differs by less than 1.2e-38). */
int cream_gemm_ntopt(int mode);
/* The second half of a two-workgroups-per-CU NT grid starts n x 64 clocks late (A/B switch; default 0); < 0 queries. */
int cream_gemm_stagger(int n);
int cream_linear_fwd(void* out, const void* x, const void* w, const void* bias, int M, int N, int K,
                     int64_t ldw, void* stream);
int cream_linear_fwd_seg(void* out, const void* x, const void* w, const void* bias, int M, int N, int K,
                         int64_t ldw, int nseg, int64_t nseg_stride, void* stream);
int cream_linear_gelu_fwd(void* gp, void* g, const void* x, const void* w, const void* bias, int M, int N,
                          int K, int64_t ldw, void* stream);
/* The same with N rounded up to a multiple of 8: outputs of columns n >= Nvalid are written as zeros. */
int cream_linear_gelu_fwd_pad(void* gp, void* g, const void* x, const void* w, const void* bias, int M, int N, int Nvalid,
                              int K, int64_t ldw, void* stream);
int cream_linear_dgrad(void* dx, const void* dy, const void* wt, int M, int N, int K, int64_t ldwt,
                       void* stream);
int cream_linear_dgrad_seg(void* dx, const void* dy, const void* wt, int M, int N, int K, int64_t ldwt,
                           int kseg, int64_t kseg_stride, void* stream);
int cream_linear_dgrad_mul(void* dh, float* colsum_parts, const void* dy, const void* wt, const void* factor,
                             int M, int N, int K, int64_t ldwt, void* stream);
int cream_linear_wgrad_splits(int M, int N, int K);
int cream_linear_wgrad_parts(float* parts, float* bias_parts, const void* dy, const void* x, int M, int N,
                             int K, int S, void* stream);
/* The same with the partial tiles written as bf16 [S][N][K] (half the partial traffic: 113 instead of 226 MB per
 * transformer block through HBM, written here and read by cream_grad_finalize with src_bf16 = 1).  Every split's sum over
 * its ~M / S tokens is rounded to bf16 once — what torch.autocast's backward does to the WHOLE gradient of the bf16 weight
 * copy (Linear_super.py:71-81 under autocast); the bias partials stay fp32. */
int cream_linear_wgrad_parts_bf16(void* parts_bf16, float* bias_parts, const void* dy, const void* x, int M, int N,
                                  int K, int S, void* stream);
/* The split count for a weight gradient with bf16 partial tiles (Linear_super.py:71-81, qkv_super.py:72-83 backward).
 * cream_linear_wgrad_parts_bf16 called with THIS split count runs the macro-tile kernel of csrc/gemm_tn8.hpp (256 x 256 tile on
 * the phase-interleaved loop, one 8-wave workgroup per CU, token slices for half the CUs — the weight gradients share the chip
 * with the main chain and every slice is another partial tile through HBM); bias partials ride along.  With any other S, or with
 * fp32 partials (cream_linear_wgrad_parts), the 128 x 128 kernel runs.  Same partial-tile layout and rounding either way.
 * cream_gemm_tn8(0 | 1 | 2): never / problems of at least six 256 x 256 tiles / every problem (default); < 0 queries.
 * Changing it changes cream_linear_wgrad_splits_bf16 and with it the backward workspace layout (cream_block_layout_epoch). */
int cream_linear_wgrad_splits_bf16(int M, int N, int K);
int cream_gemm_tn8(int mode);

/* ---- parameter update + operand copies ------------------------------------------------------
 * torch.optim.AdamW as created by timm's create_optimizer (AutoFormer/supernet_train.py:294-296;
 * decoupled weight decay, bias-corrected moments) over EVERY tensor of the model in one launch, which
 * also writes the bf16 operand copies the GEMM kernels read: `mir` = the tensor as (rows x cols) bf16
 * with row stride ld_mir, `mir_t` = its transpose (cols x rows, row stride ld_mir_t) or NULL;
 * deinterleave != 0 (the qkv super weight, qkv_super.py:75): row 3 i + j of the tensor is row i of
 * part j — parts seg_stride (seg_stride_t) elements apart in the copies.  g == NULL or update == 0:
 * copies only (initialisation, checkpoint load).  The job table and the prefix sums of
 * cream_param_job_tiles(rows, cols) over the jobs (njobs + 1 entries) live in DEVICE memory. */
typedef struct cream_param_job {
    float* p;              /* fp32 master tensor viewed as (rows x cols), row stride ld          */
    const float* g;        /* gradient (same layout) or NULL                                    */
    float *m, *v;          /* exp_avg, exp_avg_sq                                               */
    void *mir, *mir_t;     /* bf16 copies or NULL                                               */
    int64_t ld, ld_mir, ld_mir_t, seg_stride, seg_stride_t;
    int32_t rows, cols, deinterleave;
    float weight_decay;
} cream_param_job;
int cream_param_job_tiles(int rows, int cols);
int cream_adamw_step(const cream_param_job* jobs_dev, const int32_t* first_tile_dev, int njobs, int total_tiles,
                     int update, double lr, double beta1, double beta2, double eps, int64_t step, void* stream);

/* Gradient finalisation for the weight-entangled parameters: every tensor of a block gets
 *     dst[map(r)*ld + c] += sum_p src[p*pstride + r*cols + c]          r < rows, c < cols
 * in ONE launch — dst is the ACTIVE SLICE W[:out, :in] of the fp32 super-weight gradient
 * (ld = super width; what autograd's backward of the slicing in Linear_super.py:71-81 and
 * qkv_super.py:72-83 accumulates), src are partial sums produced upstream (split-K weight
 * gradient GEMMs, per-slab column sums, per-(batch, head) table gradients).  interleave = Q > 0:
 * the rows of src are grouped [q | k | v] (Q each) and row r goes to super row
 * 3*(r % Q) + r / Q — the adjoint of qkv_super's row gather (qkv_super.py:75).  Parts are
 * added in a fixed tree: bit-reproducible.  cols, ld, pstride % 4 == 0. */
typedef struct cream_grad_job {
    float* dst;
    const void* src;
    int64_t ld;
    int64_t pstride;       /* elements between consecutive parts */
    int32_t nparts, rows, cols;
    int32_t interleave;
    int32_t src_bf16;      /* partials are bf16 (1) or fp32 (0) */
    int32_t overwrite;     /* 0: dst += sum of the parts; 1: dst = sum (a gradient that did not exist yet: the caller allocates it
                              uninitialised instead of zero-filling it; the job must cover all of dst) */
} cream_grad_job;
#define CREAM_MAX_GRAD_JOBS 24
int cream_grad_finalize(const cream_grad_job* jobs, int njobs, void* stream);

/* ---- the uint8 -> normalised float input transform of a batch, on the device (csrc/image_transform.hip) -----------------
 * AutoFormer/lib/datasets.py:189-220 (`build_transform`): eval = Resize(int(256 / 224 * input_size), bicubic) -> CenterCrop ->
 * ToTensor -> Normalize; train = timm's RandomResizedCropAndInterpolation(bicubic) -> RandomHorizontalFlip -> [RandAugment: host
 * side] -> ToTensor -> Normalize.  On the reference's PIL images every resize is Pillow's Image.resize(size, BICUBIC); the kernels
 * restate that 8-bit algorithm (libImaging/Resample.c, 22-bit fixed-point coefficients from doubles, horizontal pass then vertical
 * pass with a uint8 intermediate) integer for integer, then torchvision's x / 255 and (x - mean) / std in float32: byte-exact
 * against Pillow, bit-exact against torch's CPU ops (tests/test_image_transform_gpu.py).
 * One image: the decoded frame (HWC uint8 RGB at pixels + offset, row_stride bytes per row), the box F.crop takes, the size F.resize
 * gives the crop, the (win_top, win_left) + (out_h, out_w) window of the resized image that becomes the output (CenterCrop; (0, 0) with
 * resized == out for the training crop) and the mirror flag of RandomHorizontalFlip.
 * cream_image_batch_plan (host only, no device work) validates the B descriptors, fills row0 / nrows (the box rows the window's
 * vertical pass reads) and tmp_off, and returns the workspace size in bytes (>= 0) or an error code (< 0): CREAM_ERR_TOO_LARGE when
 * a crop is wider than 4776 pixels or shrinks by more than (40960 / (4 out_w) - 1) / 4 (x 11 at out_w = 224).
 * cream_image_batch_transform takes the planned descriptors twice — the host array (validated again, sizes the grids) and a copy
 * of it in device memory (the caller uploads it as it likes: one async copy from pinned memory) — and enqueues three launches (coefficient tables, horizontal pass, vertical pass + float tail) on
 * `stream`: out (B, 3, out_h, out_w) fp32.  pixels: 4-byte aligned, pixels_bytes a multiple of 4 (rows are staged with aligned
 * 4-byte loads); out, workspace: 16-byte aligned; out_w % 4 == 0, out_w <= 1024. */
typedef struct cream_image_desc {
    int64_t offset;                                 /* byte offset of the frame in `pixels` */
    int32_t height, width, row_stride;              /* decoded frame; row_stride >= 3 * width bytes */
    int32_t box_top, box_left, box_h, box_w;        /* F.crop */
    int32_t resized_h, resized_w;                   /* F.resize of the crop */
    int32_t win_top, win_left;                      /* window of the resized image */
    int32_t flip;                                   /* != 0: mirrored */
    int32_t row0, nrows;                            /* filled by cream_image_batch_plan */
    int64_t tmp_off;                                /* filled by cream_image_batch_plan */
    /* RandomErasing of the training recipe (timm, mode 'pixel': lib/datasets.py:199-201 re_prob / re_mode / re_count), applied to
     * the NORMALISED output: elements [:, erase_top : + erase_h, erase_left : + erase_w] become standard-normal noise, a pure
     * function of (erase_seed, channel, row, column) (a counter-based hash through Box-Muller); erase_h == 0: no erasing.  The box
     * is drawn on the host (autoformer/data.py: random_erasing_params, timm's order of draws); the noise stream is this library's
     * own (no two implementations share one). */
    int32_t erase_top, erase_left, erase_h, erase_w;
    uint32_t erase_seed;
    int32_t reserved;
} cream_image_desc;
int64_t cream_image_batch_plan(cream_image_desc* descs, int B, int out_h, int out_w);
int cream_image_batch_transform(float* out, const uint8_t* pixels, int64_t pixels_bytes, const cream_image_desc* descs,
                                const cream_image_desc* descs_dev, int B, int out_h, int out_w, const float* mean,
                                const float* stdev, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- batch-mode Mixup / CutMix + smoothed soft targets, one launch (csrc/mixup.hip) -----------------
 * `samples, targets = mixup_fn(samples, targets)` of the step body (AutoFormer/supernet_engine.py:52-53; timm.data.Mixup as
 * constructed at supernet_train.py:245-251: mode 'batch' — third-party, not vendored, restated; parity unpinned by the reference,
 * pinned against the host restatement cream_amd/autoformer/data.py).  In place on x (B, Cimg, H, W) fp32, B even, W % 4 == 0:
 *   use_cutmix == 0:  x_b <- lam x_b + (1 - lam) x_{B-1-b}
 *   use_cutmix != 0:  x_b[:, yl:yh, xl:xh] <- x_{B-1-b}[:, yl:yh, xl:xh]   (the caller passes lam corrected for the clipped box)
 *   lam == 1: images untouched.
 * y (B, num_classes) fp32 <- lam onehot_s(target_b) + (1 - lam) onehot_s(target_{B-1-b}), onehot_s = label_smoothing / C off,
 * 1 - label_smoothing + label_smoothing / C on; target (B) int64 class indices.  lambda and the box are drawn by the caller
 * (numpy's global RNG in timm's order: data.Mixup). */
int cream_mixup_cutmix(float* x, float* y, const int64_t* target, int B, int Cimg, int H, int W, int num_classes, float lam,
                       int use_cutmix, int yl, int yh, int xl, int xh, float label_smoothing, void* stream);

/* ---- soft-target cross entropy (loss rows + logit gradient in one launch) --------------------------
 * criterion(outputs, targets) of the step body (AutoFormer/supernet_engine.py:60-66) with timm's
 * SoftTargetCrossEntropy (third-party, not vendored: restated from its published definition —
 * loss = mean_b sum_c -t[b,c] log_softmax(x[b,:])[c]; parity unpinned, as for the framework formulation it replaces):
 *   loss_rows[b] = sum_c t[b,c] (logsumexp(x[b,:]) - x[b,c])                  (the caller takes the mean)
 *   dlogits[b,c] = (softmax(x[b,:])[c] * sum_c t[b,c] - t[b,c]) * grad_scale   f32 (grad_scale = 1 / B for the mean)
 * logits (B x C) bf16 or f32 (logits_dtype), target (B x C) f32, C <= 2048; softmax statistics in fp32. */
int cream_soft_ce(float* loss_rows, float* dlogits, const void* logits, const float* target, int B, int C, int logits_dtype,
                  float grad_scale, void* stream);

/* ---- fp32-I/O instantiations (parity mode) -------------------------------------------------------
 * The same operators with fp32 tensors on both sides, for the "within 1e-3 of the reference PyTorch-CPU
 * forward / backward" bar: exact-fp32 matrix-core products (v_mfma_f32_32x32x2_f32; gfx950 has no TF32)
 * and the fp32 instantiation of the LayerNorm kernels.
 *
 * cream_linear_f32_fwd   y (M x N) = x (M x K, row stride ldx) . W[wmap(n), :K]^T + bias[:N]
 *                        = LinearSuper.forward / qkv_super.forward (Linear_super.py:38-54, :71-81; qkv_super.py:72-83)
 * cream_linear_f32_dgrad dx (M x K) = dy (M x N) . W[wmap(n), :K]
 * cream_linear_f32_wgrad dW[wmap(n), :K] = dy^T . x  (rows of dw, stride lddw; other rows untouched);
 *                        dbias (N) = column sums of dy in ascending row order, or NULL
 * wmap(r) = (r % seg) * step + r / seg for seg > 0 (qkv super weight: seg = Q, step = 3 — rows 3 i + j,
 * qkv_super.py:75), identity for seg == 0.  w: fp32 super weight, row stride ldw.
 * cream_ln_f32_fwd / _bwd = LayerNormSuper.forward (layernorm_super.py:33-37) and its autograd backward:
 * partial (cream_ln_partials() x 3 x E): planes 0 / 1 = per-slab dgamma / dbeta sums (plane 2 zero). */
int cream_linear_f32_fwd(float* y, const float* x, const float* w, const float* bias, int M, int N, int K, int64_t ldx,
                         int64_t ldw, int seg, int step, void* stream);
int cream_linear_f32_dgrad(float* dx, const float* dy, const float* w, int M, int N, int K, int64_t ldw, int seg, int step,
                           void* stream);
int cream_linear_f32_wgrad(float* dw, float* dbias, const float* dy, const float* x, int M, int N, int K, int64_t ldx,
                           int64_t lddw, int seg, int step, void* stream);
/* cream_bmm_f32  C_z (M x N) = A_z (M x K) . B_z (K x N) for the nb0 x nb1 batch items z = (z0, z1): the attention
 *   products of the iRPE parity mode on the same exact-fp32 matrix-core kernel — q k^T and P v of RPEAttention.forward
 *   (iRPE/DeiT-with-iRPE/rpe_vision_transformer.py:76, :88), the lookup products of irpe.py:641-644 / :683-687 and their
 *   autograd products — instead of the framework's batched matmul (the vendor library).
 *   *_strides[4] = {row, column, z0, z1} ELEMENT strides of each operand (rows of A = m, columns = k; rows of B = k,
 *   columns = n): transposed views, the head-interleaved (B, L, 3, H, d) qkv layout and broadcast operands (batch
 *   stride 0 on A or B) need no copies.  C must not alias itself (no zero stride on an extent > 1).  nb0 * nb1 <= 65535. */
int cream_bmm_f32(float* c, const float* a, const float* b, int M, int N, int K, const int64_t* a_strides,
                  const int64_t* b_strides, const int64_t* c_strides, int nb0, int nb1, void* stream);
int cream_ln_f32_fwd(float* y, float* mean, float* rstd, const float* x, const float* gamma, const float* beta,
                     int M, int E, float eps, void* stream);
int cream_ln_f32_bwd(float* dx, float* partial, const float* dy, const float* x, const float* mean, const float* rstd,
                     const float* gamma, int M, int E, void* stream);

/* ---- active-slice gradient messages (data-parallel exchange) ------------------------------------
 * DistributedDataParallel (AutoFormer/supernet_train.py:286-289) all-reduces the FULL gradient of every
 * super weight; a sampled sub-network only writes W.grad[:rows, :cols] (Linear_super.py:71-81,
 * qkv_super.py:72-83) and every rank samples the same one (supernet_engine.py:36).  cream_slices_copy
 * gathers the active slices of one bucket into a contiguous fp32 message (to_packed != 0) or scatters
 * the reduced message back (to_packed == 0): slice i = full[r * ld + c], r < rows, c < cols, stored
 * row-major at packed + packed_off.  The job table lives in DEVICE memory (built once per bucket and
 * configuration); max_rows = the largest `rows` of the table (sizes the grid). */
typedef struct cream_slice_job {
    float* full;
    int64_t ld;
    int64_t packed_off;
    int32_t rows, cols;
} cream_slice_job;
int cream_slices_copy(const cream_slice_job* jobs_dev, int njobs, float* packed, int max_rows, int to_packed, void* stream);

/* ---- one transformer block, sequenced natively ---------------------------------------------
 * TransformerEncoderLayer.forward (AutoFormer/model/supernet_transformer.py:251-287) and its
 * autograd backward as ONE call per direction: the kernels are the ones declared above, enqueued
 * on `stream` (and, in backward, the weight-gradient GEMMs + the gradient finalisation on
 * `side_stream` behind events).  bf16 throughput mode: fp32 residual stream, bf16 GEMM operands.
 * The caller owns all memory: one flat workspace per call (sizes/offsets from *_workspace). */
typedef struct cream_block_desc {
    int32_t B, N, E, H, F;        /* batch, tokens, embed dim, heads (head dim 64), mlp hidden     */
    int32_t gh, gw, mr;           /* token grid (N = gh*gw + 1) and max_relative_position          */
    int32_t F_valid, inference;     /* inference != 0: a forward without a backward (supernet_engine.py:114-160 evaluate, the
                                       evolution search): tensors that only the backward reads (gelu'(h)) are not written.
                                       F_valid > 0: F is the sampled hidden width rounded UP to a multiple of 8 and only the
                                       first F_valid hidden units exist (the fc1 epilogue writes zeros for the rest, which
                                       makes every product over the padded columns vanish); 0: F itself is exact */
    float eps1, eps2, attn_scale;
    float reserved_f;             /* MUST be 0: every field of this struct is meaningful or reserved-as-zero — zero-initialise the
                                     struct (memset / `= {0}`) before filling it; `inference` above was a reserved field until r3 */
    /* bf16 operand copies of the SUPER weights (written by cream_adamw_step), read in place:
     * w* = (out x in) row stride ld_*, w*_t = transposed (in x out) row stride ld_*_t;
     * qkv: three de-interleaved parts [q | k | v], seg_qkv / seg_qkv_t elements apart */
    const void *wqkv, *wqkv_t, *bqkv;
    const void *wproj, *wproj_t, *bproj;
    const void *w1, *w1_t, *b1;
    const void *w2, *w2_t, *b2;
    int64_t ld_qkv, ld_qkv_t, seg_qkv, seg_qkv_t, ld_proj, ld_proj_t, ld_w1, ld_w1_t, ld_w2, ld_w2_t;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;     /* attn_layer_norm / ffn_layer_norm (fp32)     */
    const float *tkv, *tkh, *tvv, *tvh;             /* rel_pos_embed_{k,v}.embeddings_table_{v,h}  */
    int64_t ldt;
    const void* timg;             /* bf16 operand images of the four tables (cream_attn_rpe2d_table_images layout, kept current
                                     by cream_adamw_step) or NULL: with them the forward runs the ping-pong kernel and the
                                     backward saves its image launch (round 6; NULL = the behaviour before) */
} cream_block_desc;

/* fp32 gradients of the block's parameters (super shapes; accumulated into, never overwritten) */
typedef struct cream_block_grads {
    float *wqkv, *bqkv, *wproj, *bproj, *w1, *b1, *w2, *b2;
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *tkv, *tkh, *tvv, *tvh;
    int64_t ld_qkv, ld_proj, ld_w1, ld_w2, ldt;
} cream_block_grads;

/* Bytes of the forward workspace (kept by the caller for the backward); *off_x = block input after
 * a pending residual add (valid only if pend_f was given), *off_x1 / *off_f = the two tensors whose
 * sum x1 + s2*f is the block output — left PENDING for the next consumer (the next block's first
 * LayerNorm pass, or cream_residual_add). */
int64_t cream_block_fwd_workspace(const cream_block_desc* d, int64_t* off_x, int64_t* off_x1, int64_t* off_f);
/* x_in (M x E) fp32; pend_f (bf16, M x E) / pend_scale (B, may be NULL): pending branch of the
 * previous block, added on the fly (NULL: none); dp1: per-sample drop-path scale of the attention
 * branch (B) or NULL. */
int cream_block_fwd(const cream_block_desc* d, void* ws, const float* x_in, const void* pend_f,
                    const float* pend_scale, const float* dp1, void* stream);

/* Backward.  fws: the forward workspace; x: the block input (x_in, or fws + off_x if the forward
 * had a pending branch); dx2 (fp32) gradient of the block output; df (bf16) = s2 * dx2 = gradient
 * of the fc2 output and pb2 its column-sum partials (pb2_parts parts, pb2_pstride floats apart) —
 * both produced by whoever consumed the block's output.  Outputs in ws: dx (fp32, *off_dx) and,
 * if want_prev, df_prev = bf16(prev_scale * dx) (*off_df_prev) with its column sums as plane 2 of
 * the (cream_ln_partials() x 3 x E) partials at *off_pl1 — i.e. (df, pb2) of the previous block.
 * Parameter gradients are complete when `side_stream` has drained.
 * Ordering contract of df: when a call returns df_prev (want_prev) and the NEXT cream_block_bwd call on the same thread and
 * the same two streams is handed exactly that buffer as df, the library does not order the side stream behind it again (it
 * already is: one marker packet less per block).  The caller must therefore pass that df_prev on UNTOUCHED — a caller that
 * rewrites it on `stream` between the two calls, or produces another df at the same address, sets CREAM_REUSE_ORDER=0 in
 * the environment (always order).  The record is consumed by the next call on the thread whatever it is given. */
int64_t cream_block_bwd_workspace(const cream_block_desc* d, int64_t* off_dx, int64_t* off_df_prev, int64_t* off_pl1);
int cream_block_bwd(const cream_block_desc* d, const cream_block_grads* g, const void* fws, const float* x,
                    void* ws, const float* dx2, const void* df, const float* pb2, int pb2_parts,
                    int64_t pb2_pstride, const float* dp1, const float* prev_scale, int want_prev,
                    void* stream, void* side_stream);


/* cream_block_bwd writes the split-K partial tiles of the four weight gradients as bf16 (1) or fp32 (0); returns the
 * previous setting; on < 0 only queries.  Initial value: CREAM_WGRAD_BF16 in the environment, else 1. */
int cream_block_wgrad_bf16(int on);

/* Workspace-layout rule.  The byte size and the internal offsets of the backward workspace depend on the split counts of the
 * four weight gradients (cream_linear_wgrad_splits_bf16), i.e. on cream_gemm_tn8() and on the device's CU count: these must NOT
 * change between cream_block_bwd_workspace and the cream_block_bwd call that uses a workspace of that size (cream_block_bwd
 * recomputes the layout; a workspace sized under another mode is overrun).  cream_block_layout_epoch() returns a counter that
 * moves whenever such a switch changes value: callers that cache workspace sizes key the cache on it (cream_amd/autoformer/
 * block.py:_ws_layout does). */
int cream_block_layout_epoch(void);

/* CU budget of the persistent kernels.  Every hot kernel of the library launches one or two persistent workgroups per compute
 * unit and takes most of its LDS, so a collective kernel launched beside them (RCCL's per-step gradient all-reduce,
 * AutoFormer/supernet_train.py:286-289) only runs once a grid drains.  cream_cu_reserve(R) makes every grid of the library size
 * itself for (CUs - R) rounded down to a multiple of 8 (one share per XCD); the caller tells RCCL to use at most R channels
 * (cream_amd/comm.py: init_distributed exports NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS before the process group is created).
 * R < 0 queries; returns the previous reserve; moves cream_block_layout_epoch (the weight-gradient token slices follow the count).
 * cream_cu_count(): what the grids are sized for now. */
int cream_cu_reserve(int reserve);
int cream_cu_count(void);

/* Optional in-step kernel timing of the two calls above (measurement aid; no reference counterpart — the
 * reference's step is timed by `MetricLogger`, AutoFormer/lib/utils.py:58-170, at step granularity).
 * While enabled, every launch of cream_block_fwd / cream_block_bwd is bracketed by a pair of HIP events
 * recorded on the stream it is launched on (main or side), so durations are those of the real two-stream
 * step.  cream_block_prof_collect waits for the recorded events and returns, per kernel family
 * (cream_block_prof_kinds() of them, names from cream_block_prof_name), the summed duration in ms, the
 * launch count and the summed ALGORITHMIC flops / bytes; it empties the record list. */
int cream_block_prof_enable(int on);
int cream_block_prof_kinds(void);
const char* cream_block_prof_name(int kind);
int cream_block_prof_collect(double* total_ms, int64_t* launches, double* flops, double* bytes);

#ifdef __cplusplus
}  /* extern "C" */
#endif
#endif  /* CREAM_AMD_H_ */
