/*
 * oq_hip.h -- C ABI of the MI355X (gfx950) OmniQuant calibration hot path.
 *
 * The reference (SuperVan-Young/OmniQuant) has no FFI: its hot path is PyTorch eager ops inside three
 * nn.Modules.  This header is the boundary a maintainer binds instead of those op sequences; every
 * entry point cites the reference lines it replaces.  All pointers are DEVICE pointers owned by the
 * caller (PyTorch-ROCm allocations), all calls are asynchronous on `stream` (a hipStream_t passed as
 * void*), never allocate, never synchronise, and are hipGraph-capturable.  Return 0 on success or a
 * negative OQ_E_* code; the message for the last error of the calling thread is oq_last_error().
 *
 * dtype codes: OQ_F32 = 0, OQ_F16 = 1, OQ_BF16 = 2.
 * "seg" is the quantisation segment along the last dim: a whole row (per-channel / per-token),
 * a weight group (group_size) or a head (head_dim).
 */
#ifndef OQ_HIP_H
#define OQ_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OQ_F32 0
#define OQ_F16 1
#define OQ_BF16 2

#define OQ_OK 0
#define OQ_E_ARG (-1)      /* bad argument (shape, alignment, dtype) */
#define OQ_E_LAUNCH (-2)   /* HIP launch failure */
#define OQ_E_UNSUPPORTED (-3)

int oq_version(void);
const char* oq_last_error(void);

/* ---- UniformAffineQuantizer: dynamic min/max + LWC + fake quant ---------------------------------
 * Replaces quantize/quantizer.py:84-105 (fake_quant), :122-147 (per_token_dynamic_calibration),
 * :15-19 (round_ste) and, for weights, the LET re-parameterisation that feeds it:
 * models/transformation.py:24-69 (W*col_mul, /row_div, *row_mul and the W@shift bias term).
 *
 *   x[r,c]   = ((w[r,c] * col_mul[c]) / row_div[r]) * row_mul[r]          (each factor optional)
 *   per segment:  hi = max x, lo = min x;  hi' = sigmoid(up)*hi, lo' = sigmoid(low)*lo  (LWC, optional)
 *   asymmetric:   s = (hi'-lo')/(2^n-1),  z = rne(clamp(-lo'/s, +-1e4))
 *   symmetric:    s = clamp(max(|hi'|,|lo'|)/(2^(n-1)-1), 1e-5, 1e4),  z = 2^(n-1)-1
 *   y = (clamp(rne(x/s)+z, 0, 2^n-1) - z) * s
 *   wshift[r] = sum_c w[r,c]*shift[c]                                      (optional by-product)
 * Outputs: y [rows,cols] (y_dtype); scale, zp, xmin, xmax [rows*cols/seg] f32 (all required; xmin/xmax feed oq_fakequant_bwd).
 * Requirements: cols % seg == 0, seg % 8 == 0, and seg/8 a power of two when seg <= 512.
 */
int oq_fakequant_fwd(const void* w, int w_dtype, int64_t rows, int64_t cols, int64_t seg, int nbits, int symmetric,
                     const float* col_mul, const float* row_div, const float* row_mul, const float* shift,
                     const float* up, const float* low,
                     void* y, int y_dtype, float* scale, float* zp, float* xmin, float* xmax, float* wshift,
                     void* codes, float* csum, void* stream);
/* Integer side channel of the quantiser forwards (optional; `codes` and `csum` both NULL or both given): next to y the
 * kernels store the GRID CODES of quantize/quantizer.py:93-99 (x_int after the clamp, 0 .. 2^nbits - 1) as int8 [rows, cols]
 * -- plain codes for nbits <= 7, code - 128 for nbits = 8 -- and per row the sum of the stored codes (float, exact; NaN for a
 * row whose scale is 0 / NaN, the reference's all-NaN row of quirk Q1).  With scale / zp that is the exact factorisation
 * y = (code - zp) * scale, which oq_gemm_i8 contracts on the int8 MFMA.  Available for whole-row segments on grids of at
 * most 8 bits when oq_fakequant_codes_supported(cols, seg, nbits, let) is 1 (let: any of col_mul / row_div / row_mul / shift). */
int64_t oq_fakequant_codes_supported(int64_t cols, int64_t seg, int nbits, int let);

/* Backward of the above (closed form of the autograd graph the reference builds).
 *   g [rows,cols] (g_dtype) = dL/dy ; g_wshift [rows] = dL/d wshift (optional)
 * Outputs (each optional, f32): g_up, g_low [rows*cols/seg]; gx [rows,cols] (gx_dtype; only when w itself
 * needs a gradient, i.e. activation quantizers); g_col_mul [cols], g_shift [cols]; g_row_div, g_row_mul [rows].
 * The column sums (g_col_mul, g_shift) are reduced deterministically through `workspace` (f32, at least
 * oq_fakequant_bwd_workspace(rows, cols) floats; only needed when one of the two is requested).
 */
int oq_fakequant_bwd(const void* w, int w_dtype, int64_t rows, int64_t cols, int64_t seg, int nbits, int symmetric,
                     const float* col_mul, const float* row_div, const float* row_mul, const float* shift,
                     const float* up, const float* low, const float* xmin, const float* xmax,
                     const void* g, int g_dtype, const float* g_wshift,
                     float* g_up, float* g_low, void* gx, int gx_dtype,
                     float* g_col_mul, float* g_shift, float* g_row_div, float* g_row_mul,
                     float* workspace, int64_t workspace_floats, void* stream);
int64_t oq_fakequant_bwd_workspace(int64_t rows, int64_t cols);

/* Several weight matrices per call: what models/int_llama_layer.py:279-307 does matrix after matrix (smooth_*_temporary +
 * weight_quantizer for q, k, v, o; quantize/quantizer.py:108-147 and models/transformation.py:24-69 and their autograd).
 * Same arguments as oq_fakequant_fwd / oq_fakequant_bwd, one struct per matrix; same results as calling those in order.
 * When all n (<= 4) problems are LET weights of one row length and one dtype pair the work is ONE launch per direction
 * plus one launch for all column reductions; otherwise the matrices are processed one by one. */
typedef struct {
    const void* w; int w_dtype; int64_t rows, cols, seg; int nbits, symmetric;
    const float *col_mul, *row_div, *row_mul, *shift, *up, *low;
    void* y; int y_dtype;
    float *scale, *zp, *xmin, *xmax, *wshift;
    void* codes; float* csum;
} oq_fakequant_fwd_args;
typedef struct {
    const void* w; int w_dtype; int64_t rows, cols, seg; int nbits, symmetric;
    const float *col_mul, *row_div, *row_mul, *shift, *up, *low, *xmin, *xmax;
    const void* g; int g_dtype;
    const float* g_wshift;
    float *g_up, *g_low;
    void* gx; int gx_dtype;
    float *g_col_mul, *g_shift, *g_row_div, *g_row_mul, *workspace;
    int64_t workspace_floats;
} oq_fakequant_bwd_args;
int oq_fakequant_fwd_multi(const oq_fakequant_fwd_args* a, int n, void* stream);
int oq_fakequant_bwd_multi(const oq_fakequant_bwd_args* a, int n, void* stream);

/* ---- QuantLinear / QuantMatMul GEMM (quantize/int_linear.py:62 F.linear; quantize/int_matmul.py:41-43
 * torch.matmul / torch.bmm, and their autograd: dgrad + wgrad) -------------------------------------------
 *   C[b][m][n] = alpha * sum_k A(b,m,k) * B(b,n,k) + bias[n]
 *   A(b,m,k) = a[bo*sa_o + bi*sa_i + (a_kc ? m*lda + k : k*lda + m)]   (b = bo*batch_i + bi)
 *   B(b,n,k) = bm[bo*sb_o + bi*sb_i + (b_kc ? n*ldb + k : k*ldb + n)]
 *   C        = c[bo*sc_o + bi*sc_i + m*ldc + n]
 * tri_mode (causal attention; rows/cols/k are token positions of ONE square [T,T] problem per batch):
 *   0 dense;  1 skip output tiles entirely above the diagonal (n0 >= m0 + tile_m): scores, dP;
 *   2 contraction restricted to k < m0 + tile_m (P@V, dS@K);  3 contraction restricted to k >= m0 (dS^T@Q, P^T@dO).
 *   Modes 2/3 rely on the masked part of the [T,T] operand being ZERO inside the diagonal 256-block (what
 *   oq_softmax_fwd/bwd write with causal=1); everything beyond that block is never read.
 * addend (optional, out_dtype, same leading dimension and batch strides as c, may alias c): c = alpha*A.B + bias + addend
 *   -- the residual add of the block (models/int_llama_layer.py:246,264) or a gradient accumulation fused into the store.
 * in_dtype OQ_BF16: bf16 operands on v_mfma_f32_16x16x32_bf16, f32 accumulate.
 * in_dtype OQ_F32 : exact f32 on v_mfma_f32_16x16x4_f32 (parity mode).
 * out_dtype OQ_F32 or OQ_BF16.  Alignment: contiguous dims and leading dims multiples of 8 elements.
 */
int oq_gemm(const void* a, const void* bm, void* c, const float* bias, const void* addend,
            int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
            int a_kc, int b_kc, int in_dtype, int out_dtype, float alpha,
            int64_t batch_o, int64_t batch_i, int64_t sa_o, int64_t sa_i, int64_t sb_o, int64_t sb_i,
            int64_t sc_o, int64_t sc_i, int tri_mode, void* stream);

/* oq_gemm with a caller-owned workspace.  oq_gemm_workspace() gives the number of bytes that lets the kernel split the
 * contraction of this problem (0: nothing to gain, oq_gemm_ws then behaves exactly like oq_gemm).  Launches whose tile count is
 * a little more than a whole number of rounds of the 256 CUs (LLaMA-2-13B: N = 5120 -> 320 tiles of 256 x 128 = 1.25 rounds)
 * are run as S x tiles work items over 1/S of the contraction each, writing fp32 partial outputs [S][M][N] into the
 * workspace; a second launch adds the parts in index order and applies alpha, bias, addend and the output rounding
 * (deterministic; not bit-identical to the unsplit launch, whose accumulation order it changes). */
int64_t oq_gemm_workspace(int64_t M, int64_t N, int64_t K, int in_dtype, int64_t batch, int tri_mode);
int oq_gemm_ws(const void* a, const void* bm, void* c, const float* bias, const void* addend,
               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
               int a_kc, int b_kc, int in_dtype, int out_dtype, float alpha,
               int64_t batch_o, int64_t batch_i, int64_t sa_o, int64_t sa_i, int64_t sb_o, int64_t sb_i,
               int64_t sc_o, int64_t sc_i, int tri_mode, void* workspace, int64_t workspace_bytes, void* stream);

/* Integer-exact fprop of the fake-quant Linear (quantize/int_linear.py:59-62 with both operands on
 * UniformAffineQuantizer grids, quantize/quantizer.py:84-105):
 *   c[m][n] = a_scale[m] * b_scale[n] * sum_k (qa[m][k] - a_zp[m]) * (qb[n][k] - b_zp[n]) + bias[n] (+ addend[m][n])
 * a_codes [M][lda], b_codes [N][ldb]: int8, the grid codes as the quantiser entry points write them into their `codes`
 * output (plain codes; code - 128 on 8-bit grids); `*_csum[r]` = sum over k of the STORED codes of row r, as float.  The contraction runs on
 * v_mfma_i32_16x16x64_i8 into int32 (exact); scales, the three zero-point terms, bias and addend are applied in fp32 in the
 * epilogue.  out_dtype OQ_F32 / OQ_BF16; addend (addend_dtype OQ_F32 / OQ_BF16, independent of the output's) has the
 * output's leading dimension; c_wide (may be NULL): a second copy of the un-rounded fp32 result, same leading dimension --
 * the hidden state's side channel in front of the next 4-bit rounding decision while c is its bf16 copy.  Fast path:
 * K % 128 == 0, N % 8 == 0, N >= 128, 16-byte aligned rows; anything else runs a plain one-thread-per-output kernel (same
 * arithmetic). */
int oq_gemm_i8(const void* a_codes, const void* b_codes, void* c, float* c_wide, const float* bias, const void* addend,
               int addend_dtype, const float* a_scale, const float* a_zp, const float* a_csum,
               const float* b_scale, const float* b_zp, const float* b_csum,
               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
               int a_bits, int b_bits, int out_dtype, void* stream);

/* The weight-gradient products of one backward pass, dW_i[N_i][K_i] = dY_i[T][N_i]^T X_i[T][K_i] (autograd of F.linear at
 * quantize/int_linear.py:62, bf16), as ONE launch of the 256x256x32 kernel: separate launches each end in a partial round of
 * the CUs, together their tiles fill whole rounds (LLaMA-7B: 12.06 rounds instead of 3 + 6 + 1 + 3; when the excess is a
 * single tile row / column of one item -- here 16 tiles of down_proj -- that strip is computed with a split contraction:
 * fp32 partials in `workspace` (oq_wgrad_group_workspace bytes, may be NULL: no peeling), fixed-order reduce).  Results equal
 * those of n oq_gemm calls bit for bit outside a peeled strip, to fp32 summation order inside it.  Items that do not fit the
 * kernel (ragged shapes, < 256 rows / columns) make the call fall back to n single launches.  `items`: array of n structs
 * { const void* gy; const void* x; void* gw; int64_t N, K, T, ld_gy, ld_x, ld_gw; }. */
int64_t oq_wgrad_group_workspace(const void* items, int n);
int oq_wgrad_group(const void* items, int n, void* workspace, int64_t workspace_bytes, void* stream);

/* column sums: out[n] = sum_m x[m,n]  (bias gradients).  out f32, overwritten.  Deterministic (slab partials in the
 * workspace, added in slab order; no atomics).  cols % 8 == 0. */
int64_t oq_colsum_workspace(int64_t rows, int64_t cols);
int oq_colsum(const void* x, int dtype, int64_t rows, int64_t cols, float* out, float* workspace,
              int64_t workspace_floats, void* stream);

/* ---- OmniLlamaRMSNorm / OmniLayerNorm (quantize/omni_norm.py:26-34, :52-63) -----------------------------
 * rms:  y = w * x * rsqrt(mean(x^2)+eps) (+ b);   layer: y = w * (x-mean)/sqrt(var+eps) + b.   w,b f32.
 * bwd: gx [rows,cols]; gw, gb [cols] f32 (gb may be NULL), reduced deterministically through `workspace`
 * (f32, at least oq_norm_bwd_workspace(rows, cols) floats).
 */
int oq_norm_fwd(const void* x, int dtype, int64_t rows, int64_t cols, const float* w, const float* b, float eps,
                int is_layernorm, void* y, float* rstd, float* mean, void* stream);
int oq_norm_bwd(const void* x, const void* gy, int dtype, int64_t rows, int64_t cols, const float* w,
                const float* rstd, const float* mean, int is_layernorm, void* gx, float* gw, float* gb,
                const void* gx_addend /* optional [rows,cols], x's dtype: gx += it (residual-path gradient) */,
                float* workspace, int64_t workspace_floats, void* stream);
int64_t oq_norm_bwd_workspace(int64_t rows, int64_t cols);

/* ---- block glue (models/int_llama_layer.py:124-125 RoPE, :44-45 SiLU*up, :153-163 mask+softmax;
 *      models/int_opt_layer.py:151-170; quantize/omniquant.py:220-222 MSE) ---------------------------------
 * rope: x [T, heads, hd] in place-capable; cos/sin f32 [T, hd] already gathered by position_ids.
 *       inverse!=0 applies the transposed rotation (backward).
 * softmax: p = softmax(max(s*alpha + mask[row % mask_rows], lowest)) over the last dim, f32 math.
 *       rows = batch*heads*Tq; mask f32 [mask_rows, cols] or NULL.
 *       causal != 0 (requires Tq == cols, mask NULL): row t = row % cols attends to columns <= t only; columns
 *       t+1 .. roundup(t+1,256)-1 are written as 0 and the rest of the row is neither read nor written.
 */
int oq_rope(const void* x, void* y, int dtype, int64_t T, int64_t heads, int64_t hd, const float* cos, const float* sin,
            int inverse, void* stream);
int oq_silu_mul_fwd(const void* gate, const void* up, void* y, int dtype, int64_t n, void* stream);
int oq_silu_mul_bwd(const void* gate, const void* up, const void* gy, void* ggate, void* gup, int dtype, int64_t n,
                    void* stream);
/* The same two on [rows, cols] problems whose gate / up (and ggate / gup) rows are `ld` elements apart (column blocks of a
 * stacked gate/up GEMM's buffer); y / gy dense.  cols and ld multiples of 8. */
int oq_silu_mul_fwd_2d(const void* gate, const void* up, void* y, int dtype, int64_t rows, int64_t cols, int64_t ld, void* stream);
int oq_silu_mul_bwd_2d(const void* gate, const void* up, const void* gy, void* ggate, void* gup, int dtype, int64_t rows,
                       int64_t cols, int64_t ld, void* stream);
/* norm -> per-token fake quant in one kernel per direction: y = fake_quant(norm(x)) with OmniLlamaRMSNorm (is_layernorm 0) or
 * OmniLayerNorm (1) semantics (quantize/omni_norm.py:26-34,52-63) and the dynamic per-token asymmetric quantiser of the
 * QuantLinear behind it (quantize/int_linear.py:59-60, quantize/quantizer.py:84-147).  rows x cols, cols = 512 .. 8192
 * (multiple of 8), dtype OQ_BF16 or OQ_F32 for x / y / g / gx alike; w, b (b may be NULL) f32 [cols].  The forward also
 * writes rstd, mean (layernorm) and the quantiser's scale, zp, xmin, xmax [rows]; the backward takes g (+ optional g2, g3:
 * dL/dy arriving in pieces, one per consumer of y -- the q/k/v or gate/up projections -- summed in that order while loading), writes
 * gx = dL/dx (+ gx_addend, the gradient that reached x on the residual path) and the column sums gw, gb (gb may be NULL)
 * through a workspace of oq_norm_quant_bwd_workspace(rows, cols) floats (deterministic two-stage reduction). */
int64_t oq_norm_quant_supported(int dtype, int64_t cols);
int64_t oq_norm_quant_bwd_workspace(int64_t rows, int64_t cols);
/* y_dtype / g_dtype: the dtype of y resp. of g, g2, g3, gx_addend and gx -- equal to x's, or OQ_BF16 with an OQ_F32 x (a
 * hidden state kept in fp32 in front of the 4-bit rounding decision).  codes / csum: see oq_fakequant_fwd. */
int oq_norm_quant_fwd(const void* x, int dtype, int64_t rows, int64_t cols, const float* w, const float* b, float eps,
                      int is_layernorm, int nbits, void* y, int y_dtype, float* rstd, float* mean, float* scale, float* zp,
                      float* xmin, float* xmax, void* codes, float* csum, void* stream);
int oq_norm_quant_bwd(const void* x, const void* g, const void* g2, const void* g3, int dtype, int g_dtype, int64_t rows, int64_t cols,
                      const float* w, const float* b,
                      const float* rstd, const float* mean, int is_layernorm, int nbits, const float* xmin, const float* xmax,
                      void* gx, float* gw, float* gb, const void* gx_addend, float* workspace, int64_t workspace_floats,
                      void* stream);
/* RoPE + head-wise fake quant of a projection output in one kernel per direction (head_dim 128): x [rows = bs*T, nh, 128]
 * (OQ_BF16 or OQ_F32) -> y = fake_quant_per_(token, head)(x * cos + rotate_half(x) * sin), cos / sin [T, 128] f32 already
 * gathered by position (both NULL: no rotation, the v path).  Replaces apply_rotary_pos_emb + qkt_matmul.quant_x1/x2 /
 * pv_matmul.quant_x2 (models/int_llama_layer.py:124-125,140-143,161 -> quantize/quantizer.py:84-147).  scale, zp, xmin,
 * xmax: [rows*nh] f32; the backward needs xmin / xmax and writes gx = dL/dx (g_dtype) from g = dL/dy. */
int64_t oq_rope_quant_supported(int dtype, int hd);
int oq_rope_quant_fwd(const void* x, int x_dtype, int64_t rows, int64_t T, int nh, int hd, const float* cos, const float* sin,
                      int nbits, void* y, int y_dtype, float* scale, float* zp, float* xmin, float* xmax, void* stream);
int oq_rope_quant_bwd(const void* x, int x_dtype, int64_t rows, int64_t T, int nh, int hd, const float* cos, const float* sin,
                      int nbits, const float* xmin, const float* xmax, const void* g, int g_dtype, void* gx, void* stream);

/* oq_qkv_rope_quant_fwd/bwd: the same for q, k and v in ONE launch per direction.  x (and gx) is the merged projection
 * output [rows, nhq + nhk + nhv, 128] (the three GEMMs write column blocks of one buffer); q and k heads are rotated, v
 * heads are not (models/int_llama_layer.py:124-125); the quantised tensors / their gradients stay three contiguous
 * [rows, nh_i, 128] buffers as the attention kernels expect.  scale / zp / xmin / xmax: [rows * (nhq + nhk + nhv)].
 * nbits == 16 (these two entry points only) is the identity grid of quantize/quantizer.py:109-110: rotate and split only
 * (weight-only configurations run q | k | v as one stacked GEMM too); scale / zp / xmin / xmax may then be NULL.
 * out_grid != 0 (nbits <= 8): yq / yk / yv receive the GRID COORDINATES clamp(round(x / s) + z, 0, 2^nbits - 1) - z of
 * quantize/quantizer.py:93-100 -- small integers, exact in bf16 -- instead of the values coordinate * s; together with
 * `scale` they are the operands of oq_attn_fwd_grid / oq_attn_bwd_grid. */
int oq_qkv_rope_quant_fwd(const void* x, int x_dtype, int64_t rows, int64_t T, int nhq, int nhk, int nhv, int hd,
                          const float* cos, const float* sin, int nbits, void* yq, void* yk, void* yv, int y_dtype,
                          int out_grid, float* scale, float* zp, float* xmin, float* xmax, void* stream);
int oq_qkv_rope_quant_bwd(const void* x, int x_dtype, int64_t rows, int64_t T, int nhq, int nhk, int nhv, int hd,
                          const float* cos, const float* sin, int nbits, const float* xmin, const float* xmax,
                          const void* gq, const void* gk, const void* gv, int g_dtype, void* gx, void* stream);
/* Fused producer + quantiser for the down_proj input of the LLaMA MLP (models/int_llama_layer.py:44-45 followed by the
 * act_quantizer call of quantize/int_linear.py:59-60): y = per-token fake_quant(silu(gate) * up), rows x cols, cols =
 * 512 .. 32768 (multiple of 8).  The product reaches the quantiser in fp32 and never goes through memory.  The backward
 * takes g = dL/dy and the forward's xmin / xmax and writes dL/dgate, dL/dup (straight-through rounding, clip mask and the
 * amax / amin tie terms of quantize/quantizer.py:122-147 included).  dtype OQ_BF16 or OQ_F32 (all tensors alike).
 * ld: row stride (elements, multiple of 8, >= cols; 0 = cols) of gate / up and of ggate / gup -- gate | up may be the two
 * column blocks of the ONE buffer a stacked gate/up GEMM writes, and their gradients the column blocks of the buffer the
 * stacked dgrad / wgrad GEMMs read; y and g are always dense [rows, cols]. */
/* dtype: gate / up; y_dtype / g_dtype: y resp. g, ggate, gup -- equal to dtype, or OQ_BF16 with OQ_F32 gate / up (projection
 * outputs kept in fp32 in front of the rounding decision).  codes / csum: see oq_fakequant_fwd. */
int oq_silu_mul_quant_fwd(const void* gate, const void* up, int dtype, int64_t rows, int64_t cols, int64_t ld, int nbits, void* y,
                          int y_dtype, float* scale, float* zp, float* xmin, float* xmax, void* codes, float* csum, void* stream);
int oq_silu_mul_quant_bwd(const void* gate, const void* up, const void* g, int dtype, int g_dtype, int64_t rows, int64_t cols,
                          int64_t ld, int nbits, const float* xmin, const float* xmax, void* ggate, void* gup, void* stream);
/* Grouped-query attention, backward of repeat_kv (models/int_llama_layer.py:136-141): y[g][d] = sum_r x[g][r][d] for
 * x [groups, rep, hd] -> y [groups, hd] (groups = rows * kv heads; fp32 sum in head order, one rounding), for one or two
 * tensors (dK and dV; x1 / y1 may be NULL) in one launch.  hd multiple of 8, dtype OQ_BF16 or OQ_F32. */
int oq_group_sum(const void* x0, void* y0, const void* x1, void* y1, int dtype, int64_t groups, int rep, int64_t hd, void* stream);
int oq_relu_fwd(const void* x, void* y, int dtype, int64_t n, void* stream);
int oq_relu_bwd(const void* x, const void* gy, void* gx, int dtype, int64_t n, void* stream);
int oq_softmax_fwd(const void* s, void* p, int dtype, int64_t rows, int64_t cols, float alpha, const float* mask,
                   int64_t mask_rows, int causal, void* stream);
int oq_softmax_bwd(const void* p, const void* gp, void* gs, int dtype, int64_t rows, int64_t cols, float alpha,
                   int causal, void* stream);
/* ---- activation statistics for the LET initialisation (generate_act_scale_shift.py:25-94), computed from the
 *      teacher pass's activations instead of a separate offline pre-pass.  x [nsamp, rows, cols]:
 *      scale[c] = running max over samples of max_t |x[s,t,c]|;  shift[c] = (max_t x + min_t x)/2 of the first sample,
 *      then 0.99*shift + 0.01*(max_t x + min_t x)/2 for every further sample, applied in sample order.  `seen` = number
 *      of samples already folded into scale/shift (0: they are overwritten). ------------------------------------- */
int64_t oq_act_stats_workspace(int64_t nsamp, int64_t cols);
int oq_act_stats(const void* x, int dtype, int64_t nsamp, int64_t rows, int64_t cols, float* scale, float* shift,
                 int64_t seen, float* workspace, int64_t workspace_floats, void* stream);

/* ---- real-quant packing (quantize/omniquant.py:255-278 -> AutoGPTQ qlinear_cuda.QuantLinear.pack; the library is not
 *      vendored in the reference: parity unpinned, see csrc/oq_pack.hip).  w [out,in] is the folded fake-quant weight,
 *      scales / zeros [out, in/group] f32 (weight_quantizer.scales / .zeros viewed [out,-1]).  Outputs (int32):
 *      qweight [in/32*bits, out], qzeros [in/group, out/32*bits] (zeros - 1, AutoGPTQ convention). ----------------- */
int oq_pack_weights(const void* w, int dtype, int64_t out, int64_t in, int64_t group, int bits, const float* scales,
                    const float* zeros, int32_t* qweight, int32_t* qzeros, void* stream);

/* ---- fused causal attention (models/int_llama_layer.py:143-163 and its autograd; bf16, head_dim 128, exact
 *      causal mask, T == 128 or T % 256 == 0 -- oq_attn_supported() says whether a problem qualifies; everything else keeps using
 *      oq_gemm + oq_softmax_*).  q, o, go [bs,T,nh,hd]; k, v [bs,T,nkv,hd] (q/k/v already fake-quantised by the caller,
 *      as quant_x1/quant_x2 do in the reference); lse [bs,nh,T] f32 (log2 domain) is written by fwd and read by bwd.
 *      bwd writes gk, gv [bs,T,nh,hd] (per q-head: the caller sums GQA groups), dsum [bs,nh,T] f32 scratch and
 *      ds_t = dS^T [bs,nh,T(key),T(query)] bf16 with zeros above the diagonal inside each 256-block, so that
 *      dQ = dS K is one oq_gemm(a=ds_t, a_kc=0, b=k, b_kc=0, tri_mode=2).  No atomics: results are deterministic. */
int64_t oq_attn_supported(int dtype, int64_t T, int hd, int causal);
int oq_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int64_t bs, int64_t T,
                int nh, int nkv, int hd, float scale, int causal, void* stream);
int oq_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* go, const float* lse,
                float* dsum, void* ds_t, void* gk, void* gv, int dtype, int64_t bs, int64_t T, int nh, int nkv, int hd,
                float scale, int causal, void* stream);
/* The same attention (models/int_llama_layer.py:140-163) on the head quantisers' INTEGER GRID: nq / nk / nv (bf16) hold the
 * grid coordinates code - zero_point of quant_x1 / quant_x2 (oq_qkv_rope_quant_fwd with out_grid), sq / sk / sv the
 * per-(token, head) scales, element [(b*T + t) * ld_s + head] (three offsets into the one [rows, nhq + nhk + nhv] vector
 * that call writes).  The fake-quantised values of quantize/quantizer.py:100 are coordinate * scale; here the products
 * Q K^T and P V contract the coordinates -- exactly, small integers accumulated in f32 -- and the scales are applied in f32
 * to the score / probability, so that q, k, v are never rounded to 16 bits; the probabilities meet the V coordinates as a
 * bf16 high + low pair (two MFMAs).  o (bf16) and, when o32 != NULL, the same result in f32 are written.  The backward
 * returns gk, gv and (through ds_t and one oq_gemm against the K COORDINATES) dQ as gradients with respect to the
 * fake-quantised VALUES, exactly as oq_attn_bwd does; with o32 (the forward's f32 output) D = rowsum(dO * O) is taken from it. */
int oq_attn_fwd_grid(const void* nq, const void* nk, const void* nv, const float* sq, const float* sk, const float* sv,
                     int64_t ld_s, void* o, float* o32, float* lse, int64_t bs, int64_t T, int nh, int nkv, int hd,
                     float scale, int causal, void* stream);
int oq_attn_bwd_grid(const void* nq, const void* nk, const void* nv, const float* sq, const float* sk, const float* sv,
                     int64_t ld_s, const void* o, const float* o32, const void* go, const float* lse, float* dsum, void* ds_t,
                     void* gk, void* gv, int64_t bs, int64_t T, int nh, int nkv, int hd, float scale, int causal, void* stream);
/* loss[0] = mean((out-t1)^2) (+ mean((out-t2)^2) if t2); g = dloss/dout * gscale.  `loss` is a buffer of
 * 1 + OQ_MSE_MAX_BLOCKS floats: loss[0] receives the result, loss[1..] is scratch for the per-workgroup partial sums
 * that a second tiny kernel adds in a fixed order (no atomics: the reported loss is bit-reproducible). */
#define OQ_MSE_MAX_BLOCKS 1024
/* out_dtype: dtype of `out` -- the targets' (`dtype`), or OQ_F32 next to bf16 targets / gradient (the block's output arriving
 * through its un-rounded side channel, see oq_gemm_i8's c_wide). */
int oq_mse_fwd_bwd(const void* out, int out_dtype, const void* t1, const void* t2, int dtype, int64_t n, float gscale,
                   float* loss, void* g, void* stream);
/* y = a + b (residual) and y = a * s (OPT query scaling) */
int oq_add(const void* a, const void* b, void* y, int dtype, int64_t n, void* stream);
int oq_scale(const void* a, float s, void* y, int dtype, int64_t n, void* stream);

/* oq_copy_samples: n (1..3) device-to-device copies of bytes_each bytes (a multiple of 16) in ONE launch: the current
 * calibration sample and its teacher output(s) -- quant_inps[j], fp_inps[j] (, fp_inps_2[j]) of quantize/omniquant.py:216-219
 * -- into the fixed buffers the replayed hipGraph of the step reads. */
int oq_copy_samples(int n, const void* src0, void* dst0, const void* src1, void* dst1, const void* src2, void* dst2,
                    int64_t bytes_each, void* stream);

/* ---- optimiser (utils.py:11-24,32-46; torch.optim.AdamW at quantize/omniquant.py:207-208;
 *      models/transformation.py:5-20 truncate_number) ----------------------------------------------------
 * Learnables of a block live in ONE flat f32 arena: [0,n_let) LET params (lr_let), [n_let,n) LWC (lr_lwc).
 * oq_gradnorm: norm_out[0] = ||g||_2 (f32), norm_out[1] = 1.0 if every element is finite else 0.0; workspace: 2048 floats.
 * oq_adamw: torch.optim.AdamW update with bias correction for `step` (read from step_ptr[0], f32 counter
 *   kept on device so the call is graph-replayable; incremented by the kernel when the step is applied).
 *   The update is skipped (GradScaler semantics, utils.py:42-43) when norm[1]==0.
 * oq_truncate: |x|<thr -> sign(x)*thr in place.
 */
int oq_gradnorm(const float* g, int64_t n, float* norm_out, float* workspace, void* stream);
int oq_adamw(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_let, float lr_let, float lr_lwc,
             float beta1, float beta2, float eps, float wd, float* step_ptr, const float* norm, void* stream);

/* oq_adamw_step: the whole optimiser step of quantize/omniquant.py:226-229 (loss_scaler -> grad norm, skip on
 * non-finite, AdamW) in three launches: grad norm (two stages; the second writes norm_out[0..1] and advances step_ptr[0]
 * when the gradients are finite), then AdamW (as oq_adamw, reading the advanced counter) which also applies truncate_number
 * (models/transformation.py:5-20, threshold truncate_thr) to the first n_truncate parameters -- the LET scales, what the
 * reference does at the top of the NEXT step (models/int_llama_layer.py:281-284) -- and clears g when zero_grads != 0
 * (the optimizer.zero_grad() of quantize/omniquant.py:225, moved behind the update).  workspace: 2048 floats.
 * step_log (optional, 1 + 2 * step_log_len floats): the per-step record the reference reads back with loss.item() /
 * norm.cpu() every step (quantize/omniquant.py:223-231) kept on the device instead -- step_log[0] counts the steps logged
 * since the host last zeroed it, entry k is (loss[0], gradient norm) of step k (written while k < step_log_len; `loss` may be
 * NULL).  The host reads the log once per epoch. */
int oq_adamw_step(float* p, float* g, float* m, float* v, int64_t n, int64_t n_let, int64_t n_truncate, float truncate_thr,
                  int zero_grads, float lr_let, float lr_lwc, float beta1, float beta2, float eps, float wd, float* step_ptr,
                  float* norm_out, float* workspace, const float* loss, float* step_log, int64_t step_log_len, void* stream);
int oq_truncate(float* x, int64_t n, float thr, void* stream);
/* dst[e][i] = sum_k src[e*OQ_SUM_MAX_SRC + k][i], k < nsrc[e], i < n[e], for e < entries, in one launch and in a fixed
 * order.  Gathers the partial gradients several backward kernels produce for one shared LET parameter into the
 * optimiser's gradient arena (what autograd's accumulation does with ~25 tiny adds behind
 * quantize/omniquant.py:226).  The pointer tables are host arrays (copied into the kernel arguments). */
#define OQ_SUM_MAX_ENTRIES 16
#define OQ_SUM_MAX_SRC 6
int oq_sum_vectors(int entries, float* const* dst, const int64_t* n, const int* nsrc, const float* const* src,
                   void* stream);

/* LET vector algebra of one block (norm weights/biases and the bias side of smooth_ln_fcs / smooth_fc_fc /
 * smooth_q_k, models/transformation.py:24-69) in one launch; all vectors f32 [n] (n = hidden size):
 *   ln1_tw = ln1_w/s1, ln1_tb = (ln1_b-h1)/s1 (ln1_b NULL: (-1*h1)/s1), same for ln2 with s3/h3;
 *   b_q = (bq0+ws_q)/t, b_k = (bk0+ws_k)*t, b_v = ((bv0+ws_v)-h2)/s2, b_o = bo0+ws_o   (bias NULL: ws alone)
 * s1/h1 = qkv scale/shift, s2/h2 = out, s3/h3 = fc1, t = qkt, ws_* = W@shift from oq_fakequant_fwd.
 * _bwd takes the gradients of the eight outputs and returns those of the seven LET vectors and of the four ws. */
int oq_let_vectors_fwd(int64_t n, const float* s1, const float* h1, const float* s2, const float* h2,
                       const float* s3, const float* h3, const float* t, const float* ln1_w, const float* ln1_b,
                       const float* ln2_w, const float* ln2_b, const float* ws_q, const float* ws_k, const float* ws_v,
                       const float* ws_o, const float* bq0, const float* bk0, const float* bv0, const float* bo0,
                       float* ln1_tw, float* ln1_tb, float* ln2_tw, float* ln2_tb, float* b_q, float* b_k, float* b_v,
                       float* b_o, void* stream);
int oq_let_vectors_bwd(int64_t n, const float* s1, const float* h1, const float* s2, const float* h2,
                       const float* s3, const float* h3, const float* t, const float* ln1_w, const float* ln1_b,
                       const float* ln2_w, const float* ln2_b, const float* ws_q, const float* ws_k, const float* ws_v,
                       const float* ws_o, const float* bq0, const float* bk0, const float* bv0, const float* bo0,
                       const float* g_ln1_tw, const float* g_ln1_tb, const float* g_ln2_tw, const float* g_ln2_tb,
                       const float* g_b_q, const float* g_b_k, const float* g_b_v, const float* g_b_o,
                       float* g_s1, float* g_h1, float* g_s2, float* g_h2, float* g_s3, float* g_h3, float* g_t,
                       float* g_ws_q, float* g_ws_k, float* g_ws_v, float* g_ws_o, void* stream);

/* dtype conversion with round-to-nearest-even */
int oq_cast(const void* x, int src_dtype, void* y, int dst_dtype, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
