// Optimiser step for the flat learnable arena of one block (gfx950).
// Replaces utils.py:11-24 (grad norm), utils.py:32-46 (GradScaler-style skip on non-finite grads) and the
// torch.optim.AdamW step configured at quantize/omniquant.py:207-208, plus truncate_number
// (models/transformation.py:5-20).  The reference launches ~100-190 tiny kernels for this; here it is two
// launches over one contiguous f32 buffer, with the step counter and the norm kept on device so the whole
// sample-step replays from a hipGraph without host round trips.
#include "oq_common.h"

namespace {

__global__ void __launch_bounds__(256) gradnorm_partial_kernel(const float* g, int64_t n, float* ws) {
    __shared__ float red[2][4];
    float ss = 0.f, bad = 0.f;
    // 16-byte loads, four in flight per thread (a grouped 70B block has 27 M learnables: 107 MB of gradients per step);
    // the arena is 16-byte aligned (a fresh allocation) and the tail is walked element-wise
    const int64_t n4 = (reinterpret_cast<uintptr_t>(g) & 15u) == 0 ? n / 4 : 0;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const f32x4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ss += a[k] * a[k]; ss += b[k] * b[k]; ss += c[k] * c[k]; ss += d[k] * d[k];
            if (!(fabsf(a[k]) <= 3.4028234663852886e38f) || !(fabsf(b[k]) <= 3.4028234663852886e38f) ||
                !(fabsf(c[k]) <= 3.4028234663852886e38f) || !(fabsf(d[k]) <= 3.4028234663852886e38f)) bad = 1.f;   // NaN or inf
        }
    }
    for (; i < n4; i += stride) {
        const f32x4 a = g4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ss += a[k] * a[k];
            if (!(fabsf(a[k]) <= 3.4028234663852886e38f)) bad = 1.f;
        }
    }
    for (int64_t j = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += stride) {
        const float v = g[j];
        ss += v * v;
        if (!(fabsf(v) <= 3.4028234663852886e38f)) bad = 1.f;
    }
    ss = wave_sum(ss);
    bad = wave_max(bad);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[0][wid] = ss; red[1][wid] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ws[2 * blockIdx.x + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    }
}

// deterministic second stage (fixed order)
__global__ void __launch_bounds__(64) gradnorm_final_kernel(const float* ws, int nblocks, float* out) {
    float ss = 0.f, bad = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) { ss += ws[2 * i]; bad = fmaxf(bad, ws[2 * i + 1]); }
    ss = wave_sum(ss);
    bad = wave_max(bad);
    if (threadIdx.x == 0) {
        out[0] = sqrtf(ss);
        out[1] = (bad == 0.f && ss == ss && ss <= 3.4028234663852886e38f) ? 1.f : 0.f;
    }
}

__global__ void __launch_bounds__(256) adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_let,
                                                    float lr_let, float lr_lwc, float b1, float b2, float eps, float wd,
                                                    const float* step_ptr, const float* norm) {
    if (norm && norm[1] == 0.f) return;   // non-finite gradients: skip the step (GradScaler semantics)
    const float step = step_ptr[0] + 1.f;
    const float bc1 = 1.f - powf(b1, step);
    const float bc2 = 1.f - powf(b2, step);
    const float bc2_sqrt = sqrtf(bc2);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float lr = i < n_let ? lr_let : lr_lwc;
        const float gi = g[i];
        float pi = p[i];
        pi = pi * (1.f - lr * wd);
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * (1.f - b1);                 // lerp_, as torch
        vi = vi * b2 + (1.f - b2) * (gi * gi);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// ---- fused step: three launches (grad-norm partials; its single-workgroup second stage, which also advances the step
// counter; AdamW + truncate_number of the LET scales + clearing of the gradient arena) instead of the six of
// oq_gradnorm + oq_adamw + oq_truncate + a fill.  (Finishing the reduction in the first launch through a "last
// workgroup" ticket was measured and dropped: the agent-scope fences it needs write back and invalidate the L2 of every
// XCD -- the step became 20 % slower.)
__global__ void __launch_bounds__(64) gradnorm_final_step_kernel(const float* ws, int nblocks, float* out, float* step_ptr,
                                                                 const float* loss, float* log, int64_t log_len) {
    float ss = 0.f, bad = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) { ss += ws[2 * i]; bad = fmaxf(bad, ws[2 * i + 1]); }
    ss = wave_sum(ss);
    bad = wave_max(bad);
    if (threadIdx.x == 0) {
        const bool finite = bad == 0.f && ss == ss && ss <= 3.4028234663852886e38f;
        out[0] = sqrtf(ss);
        out[1] = finite ? 1.f : 0.f;
        if (finite) step_ptr[0] += 1.f;       // the AdamW launch behind this one reads the advanced counter
        if (log) {
            // step log: log[0] = number of steps logged since the host last reset it, then (loss, grad norm) pairs -- what the
            // reference fetches with loss.item() every step (quantize/omniquant.py:223-231) stays on the device until the
            // epoch's single read-back
            const int64_t cur = (int64_t)log[0];
            if (cur < log_len) {
                log[1 + 2 * cur] = loss ? loss[0] : 0.f;
                log[2 + 2 * cur] = out[0];
            }
            log[0] = (float)(cur + 1);
        }
    }
}

__global__ void __launch_bounds__(256) adamw_fused_kernel(float* p, float* g, float* m, float* v, int64_t n, int64_t n_let,
                                                          int64_t n_trunc, float thr, int zero_grads, float lr_let, float lr_lwc,
                                                          float b1, float b2, float eps, float wd, const float* step_ptr,
                                                          const float* norm) {
    const bool skip = norm[1] == 0.f;     // non-finite gradients: no update (GradScaler semantics), gradients still cleared
    const float step = step_ptr[0];       // already advanced by gradnorm_final_step_kernel
    const float bc1 = 1.f - powf(b1, step);
    const float bc2 = 1.f - powf(b2, step);
    const float bc2_sqrt = sqrtf(bc2);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!skip) {
            const float lr = i < n_let ? lr_let : lr_lwc;
            const float gi = g[i];
            float pi = p[i];
            pi = pi * (1.f - lr * wd);
            float mi = m[i], vi = v[i];
            mi = mi + (gi - mi) * (1.f - b1);                 // lerp_, as torch
            vi = vi * b2 + (1.f - b2) * (gi * gi);
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pi = pi - (lr / bc1) * (mi / denom);
            // truncate_number (models/transformation.py:5-20) of the LET scales: what the NEXT step would do first
            if (i < n_trunc && fabsf(pi) < thr) pi = pi > 0.f ? thr : (pi < 0.f ? -thr : 0.f);
            p[i] = pi; m[i] = mi; v[i] = vi;
        }
        if (zero_grads) g[i] = 0.f;
    }
}

__global__ void step_inc_kernel(float* step_ptr, const float* norm) {
    if (norm && norm[1] == 0.f) return;
    step_ptr[0] += 1.f;
}

__global__ void truncate_kernel(float* x, int64_t n, float thr) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (fabsf(v) < thr) x[i] = (v > 0.f ? thr : (v < 0.f ? -thr : 0.f));
    }
}

constexpr int GN_BLOCKS = 1024;
}  // namespace

extern "C" int oq_gradnorm(const float* g, int64_t n, float* norm_out, float* workspace, void* stream) {
    OQ_CHECK_ARG(g && norm_out && workspace && n > 0, "oq_gradnorm: bad args (workspace needs %d floats)", 2 * GN_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = (n + 255) / 256;
    nb = nb > GN_BLOCKS ? GN_BLOCKS : nb;
    hipLaunchKernelGGL(gradnorm_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, workspace);
    hipLaunchKernelGGL(gradnorm_final_kernel, dim3(1), dim3(64), 0, st, workspace, (int)nb, norm_out);
    OQ_CHECK_LAUNCH("oq_gradnorm");
    return OQ_OK;
}

extern "C" int oq_adamw(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_let, float lr_let,
                        float lr_lwc, float beta1, float beta2, float eps, float wd, float* step_ptr, const float* norm,
                        void* stream) {
    OQ_CHECK_ARG(p && g && m && v && step_ptr && n > 0 && n_let >= 0 && n_let <= n, "oq_adamw: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = (n + 255) / 256;
    nb = nb > 2048 ? 2048 : nb;
    hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, st, p, g, m, v, n, n_let, lr_let, lr_lwc, beta1, beta2, eps, wd,
                       (const float*)step_ptr, norm);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step_ptr, norm);
    OQ_CHECK_LAUNCH("oq_adamw");
    return OQ_OK;
}

extern "C" int oq_adamw_step(float* p, float* g, float* m, float* v, int64_t n, int64_t n_let, int64_t n_truncate,
                             float truncate_thr, int zero_grads, float lr_let, float lr_lwc, float beta1, float beta2, float eps,
                             float wd, float* step_ptr, float* norm_out, float* workspace, const float* loss, float* step_log,
                             int64_t step_log_len, void* stream) {
    OQ_CHECK_ARG(p && g && m && v && step_ptr && norm_out && workspace && n > 0 && n_let >= 0 && n_let <= n &&
                 n_truncate >= 0 && n_truncate <= n, "oq_adamw_step: bad args (workspace needs %d floats)", 2 * GN_BLOCKS);
    OQ_CHECK_ARG(!step_log || step_log_len > 0, "oq_adamw_step: step log of %lld entries", (long long)step_log_len);
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = (n + 255) / 256;
    const int64_t gb = nb > GN_BLOCKS ? GN_BLOCKS : nb;
    hipLaunchKernelGGL(gradnorm_partial_kernel, dim3(gb), dim3(256), 0, st, (const float*)g, n, workspace);
    hipLaunchKernelGGL(gradnorm_final_step_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, (int)gb, norm_out, step_ptr,
                       loss, step_log, step_log_len);
    nb = nb > 2048 ? 2048 : nb;
    hipLaunchKernelGGL(adamw_fused_kernel, dim3(nb), dim3(256), 0, st, p, g, m, v, n, n_let, n_truncate, truncate_thr, zero_grads,
                       lr_let, lr_lwc, beta1, beta2, eps, wd, (const float*)step_ptr, (const float*)norm_out);
    OQ_CHECK_LAUNCH("oq_adamw_step");
    return OQ_OK;
}

extern "C" int oq_truncate(float* x, int64_t n, float thr, void* stream) {
    OQ_CHECK_ARG(x && n > 0, "oq_truncate: bad args");
    int64_t nb = (n + 255) / 256;
    nb = nb > 1024 ? 1024 : nb;
    hipLaunchKernelGGL(truncate_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, thr);
    OQ_CHECK_LAUNCH("oq_truncate");
    return OQ_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// dst_e[i] = sum_k src_{e,k}[i] for up to OQ_SUM_MAX_ENTRIES vectors in ONE launch: gathers the partial gradients
// that several backward kernels produced for the same shared LET parameter (q/k/v/norm all touch the qkv scale)
// straight into the optimiser's gradient arena.  Replaces the ~25 tiny `add` launches autograd would issue for
// that accumulation (quantize/omniquant.py:226 loss_scaler -> loss.backward()); fixed summation order.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct SumV {
    float* dst[OQ_SUM_MAX_ENTRIES];
    const float* src[OQ_SUM_MAX_ENTRIES][OQ_SUM_MAX_SRC];
    int64_t n[OQ_SUM_MAX_ENTRIES];
    int nsrc[OQ_SUM_MAX_ENTRIES];
};

__global__ void __launch_bounds__(256) sum_vectors_kernel(SumV p) {
    const int e = blockIdx.y;
    const int64_t n = p.n[e];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float a = 0.f;
        for (int k = 0; k < p.nsrc[e]; ++k) a += p.src[e][k][i];
        p.dst[e][i] = a;
    }
}
}  // namespace

extern "C" int oq_sum_vectors(int entries, float* const* dst, const int64_t* n, const int* nsrc, const float* const* src,
                              void* stream) {
    OQ_CHECK_ARG(entries > 0 && entries <= OQ_SUM_MAX_ENTRIES && dst && n && nsrc && src, "oq_sum_vectors: %d entries (max %d)",
                 entries, OQ_SUM_MAX_ENTRIES);
    SumV p{};
    int64_t nmax = 0;
    for (int e = 0; e < entries; ++e) {
        OQ_CHECK_ARG(dst[e] && n[e] > 0 && nsrc[e] > 0 && nsrc[e] <= OQ_SUM_MAX_SRC, "oq_sum_vectors: entry %d: n=%lld nsrc=%d", e,
                     (long long)n[e], nsrc[e]);
        p.dst[e] = dst[e];
        p.n[e] = n[e];
        p.nsrc[e] = nsrc[e];
        for (int k = 0; k < nsrc[e]; ++k) {
            OQ_CHECK_ARG(src[e * OQ_SUM_MAX_SRC + k], "oq_sum_vectors: entry %d source %d is NULL", e, k);
            p.src[e][k] = src[e * OQ_SUM_MAX_SRC + k];
        }
        nmax = n[e] > nmax ? n[e] : nmax;
    }
    int64_t gx = (nmax + 255) / 256;
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(sum_vectors_kernel, dim3((unsigned)gx, (unsigned)entries), dim3(256), 0, (hipStream_t)stream, p);
    OQ_CHECK_LAUNCH("oq_sum_vectors");
    return OQ_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// LET vector algebra of one block in ONE launch (forward) + ONE launch (backward).
// Replaces the [hidden]-sized eager ops of models/transformation.py:24-69 (norm weight/bias re-parameterisation and
// the bias side of smooth_ln_fcs / smooth_fc_fc / smooth_q_k):
//   ln1_tw = ln1_w / s1            ln1_tb = (ln1_b - h1) / s1      (ln1_b NULL: (-1*h1)/s1)
//   ln2_tw = ln2_w / s3            ln2_tb = (ln2_b - h3) / s3
//   b_q = (bq0 + ws_q) / t         b_k = (bk0 + ws_k) * t          (bias NULL: ws alone)
//   b_v = ((bv0 + ws_v) - h2) / s2 b_o = bo0 + ws_o
// with s1/h1 = qkv scale/shift, s2/h2 = out, s3/h3 = fc1, t = qkt, ws_* = W @ shift from oq_fakequant_fwd.
// The reference launches ~10 tiny kernels for this and ~25 more in autograd; each costs ~4-5 us inside the graph.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct LetV {
    int64_t n;
    const float *s1, *h1, *s2, *h2, *s3, *h3, *t;
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const float *ws_q, *ws_k, *ws_v, *ws_o, *bq0, *bk0, *bv0, *bo0;
    float *ln1_tw, *ln1_tb, *ln2_tw, *ln2_tb, *b_q, *b_k, *b_v, *b_o;                 // fwd outputs / bwd: incoming grads
    float *g_s1, *g_h1, *g_s2, *g_h2, *g_s3, *g_h3, *g_t, *g_ws_q, *g_ws_k, *g_ws_v, *g_ws_o;   // bwd outputs
};

__global__ void __launch_bounds__(256) letvec_fwd_kernel(LetV p) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s1 = p.s1[i], h1 = p.h1[i], s2 = p.s2[i], h2 = p.h2[i], s3 = p.s3[i], h3 = p.h3[i], t = p.t[i];
        p.ln1_tw[i] = p.ln1_w[i] / s1;
        p.ln1_tb[i] = p.ln1_b ? (p.ln1_b[i] - h1) / s1 : (-1.f * h1) / s1;
        p.ln2_tw[i] = p.ln2_w[i] / s3;
        p.ln2_tb[i] = p.ln2_b ? (p.ln2_b[i] - h3) / s3 : (-1.f * h3) / s3;
        const float aq = p.bq0 ? p.bq0[i] + p.ws_q[i] : p.ws_q[i];
        const float ak = p.bk0 ? p.bk0[i] + p.ws_k[i] : p.ws_k[i];
        const float av = p.bv0 ? p.bv0[i] + p.ws_v[i] : p.ws_v[i];
        p.b_q[i] = aq / t;
        p.b_k[i] = ak * t;
        p.b_v[i] = (av - h2) / s2;
        p.b_o[i] = p.bo0 ? p.bo0[i] + p.ws_o[i] : p.ws_o[i];
    }
}

// the struct's forward-output slots carry the INCOMING gradients here
__global__ void __launch_bounds__(256) letvec_bwd_kernel(LetV p) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s1 = p.s1[i], h1 = p.h1[i], s2 = p.s2[i], h2 = p.h2[i], s3 = p.s3[i], h3 = p.h3[i], t = p.t[i];
        const float g1w = p.ln1_tw[i], g1b = p.ln1_tb[i], g2w = p.ln2_tw[i], g2b = p.ln2_tb[i];
        const float gq = p.b_q[i], gk = p.b_k[i], gv = p.b_v[i], go = p.b_o[i];
        const float n1 = p.ln1_b ? p.ln1_b[i] - h1 : -1.f * h1;
        const float n3 = p.ln2_b ? p.ln2_b[i] - h3 : -1.f * h3;
        p.g_s1[i] = -(g1w * p.ln1_w[i] + g1b * n1) / (s1 * s1);
        p.g_h1[i] = -g1b / s1;
        p.g_s3[i] = -(g2w * p.ln2_w[i] + g2b * n3) / (s3 * s3);
        p.g_h3[i] = -g2b / s3;
        const float aq = p.bq0 ? p.bq0[i] + p.ws_q[i] : p.ws_q[i];
        const float ak = p.bk0 ? p.bk0[i] + p.ws_k[i] : p.ws_k[i];
        const float av = p.bv0 ? p.bv0[i] + p.ws_v[i] : p.ws_v[i];
        p.g_t[i] = -gq * aq / (t * t) + gk * ak;
        p.g_ws_q[i] = gq / t;
        p.g_ws_k[i] = gk * t;
        p.g_s2[i] = -gv * (av - h2) / (s2 * s2);
        p.g_h2[i] = -gv / s2;
        p.g_ws_v[i] = gv / s2;
        p.g_ws_o[i] = go;
    }
}
}  // namespace

extern "C" int oq_let_vectors_fwd(int64_t n, const float* s1, const float* h1, const float* s2, const float* h2,
                                  const float* s3, const float* h3, const float* t, const float* ln1_w,
                                  const float* ln1_b, const float* ln2_w, const float* ln2_b, const float* ws_q,
                                  const float* ws_k, const float* ws_v, const float* ws_o, const float* bq0,
                                  const float* bk0, const float* bv0, const float* bo0, float* ln1_tw, float* ln1_tb,
                                  float* ln2_tw, float* ln2_tb, float* b_q, float* b_k, float* b_v, float* b_o,
                                  void* stream) {
    OQ_CHECK_ARG(n > 0 && s1 && h1 && s2 && h2 && s3 && h3 && t && ln1_w && ln2_w && ws_q && ws_k && ws_v && ws_o &&
                     ln1_tw && ln1_tb && ln2_tw && ln2_tb && b_q && b_k && b_v && b_o,
                 "oq_let_vectors_fwd: null pointer");
    LetV p{};
    p.n = n; p.s1 = s1; p.h1 = h1; p.s2 = s2; p.h2 = h2; p.s3 = s3; p.h3 = h3; p.t = t;
    p.ln1_w = ln1_w; p.ln1_b = ln1_b; p.ln2_w = ln2_w; p.ln2_b = ln2_b;
    p.ws_q = ws_q; p.ws_k = ws_k; p.ws_v = ws_v; p.ws_o = ws_o; p.bq0 = bq0; p.bk0 = bk0; p.bv0 = bv0; p.bo0 = bo0;
    p.ln1_tw = ln1_tw; p.ln1_tb = ln1_tb; p.ln2_tw = ln2_tw; p.ln2_tb = ln2_tb; p.b_q = b_q; p.b_k = b_k; p.b_v = b_v; p.b_o = b_o;
    int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(letvec_fwd_kernel, dim3(nb > 256 ? 256 : nb), dim3(256), 0, (hipStream_t)stream, p);
    OQ_CHECK_LAUNCH("oq_let_vectors_fwd");
    return OQ_OK;
}

extern "C" int oq_let_vectors_bwd(int64_t n, const float* s1, const float* h1, const float* s2, const float* h2,
                                  const float* s3, const float* h3, const float* t, const float* ln1_w,
                                  const float* ln1_b, const float* ln2_w, const float* ln2_b, const float* ws_q,
                                  const float* ws_k, const float* ws_v, const float* ws_o, const float* bq0,
                                  const float* bk0, const float* bv0, const float* bo0, const float* g_ln1_tw,
                                  const float* g_ln1_tb, const float* g_ln2_tw, const float* g_ln2_tb,
                                  const float* g_b_q, const float* g_b_k, const float* g_b_v, const float* g_b_o,
                                  float* g_s1, float* g_h1, float* g_s2, float* g_h2, float* g_s3, float* g_h3,
                                  float* g_t, float* g_ws_q, float* g_ws_k, float* g_ws_v, float* g_ws_o, void* stream) {
    OQ_CHECK_ARG(n > 0 && s1 && h1 && s2 && h2 && s3 && h3 && t && ln1_w && ln2_w && ws_q && ws_k && ws_v && ws_o &&
                     g_ln1_tw && g_ln1_tb && g_ln2_tw && g_ln2_tb && g_b_q && g_b_k && g_b_v && g_b_o && g_s1 && g_h1 &&
                     g_s2 && g_h2 && g_s3 && g_h3 && g_t && g_ws_q && g_ws_k && g_ws_v && g_ws_o,
                 "oq_let_vectors_bwd: null pointer");
    LetV p{};
    p.n = n; p.s1 = s1; p.h1 = h1; p.s2 = s2; p.h2 = h2; p.s3 = s3; p.h3 = h3; p.t = t;
    p.ln1_w = ln1_w; p.ln1_b = ln1_b; p.ln2_w = ln2_w; p.ln2_b = ln2_b;
    p.ws_q = ws_q; p.ws_k = ws_k; p.ws_v = ws_v; p.ws_o = ws_o; p.bq0 = bq0; p.bk0 = bk0; p.bv0 = bv0; p.bo0 = bo0;
    p.ln1_tw = const_cast<float*>(g_ln1_tw); p.ln1_tb = const_cast<float*>(g_ln1_tb);
    p.ln2_tw = const_cast<float*>(g_ln2_tw); p.ln2_tb = const_cast<float*>(g_ln2_tb);
    p.b_q = const_cast<float*>(g_b_q); p.b_k = const_cast<float*>(g_b_k); p.b_v = const_cast<float*>(g_b_v);
    p.b_o = const_cast<float*>(g_b_o);
    p.g_s1 = g_s1; p.g_h1 = g_h1; p.g_s2 = g_s2; p.g_h2 = g_h2; p.g_s3 = g_s3; p.g_h3 = g_h3; p.g_t = g_t;
    p.g_ws_q = g_ws_q; p.g_ws_k = g_ws_k; p.g_ws_v = g_ws_v; p.g_ws_o = g_ws_o;
    int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(letvec_bwd_kernel, dim3(nb > 256 ? 256 : nb), dim3(256), 0, (hipStream_t)stream, p);
    OQ_CHECK_LAUNCH("oq_let_vectors_bwd");
    return OQ_OK;
}
