// OmniLlamaRMSNorm / OmniLayerNorm forward + backward for gfx950 (quantize/omni_norm.py:26-34,:52-63).
// One workgroup per row (row kept in registers, <= 4 chunks of 8 per lane), f32 statistics.
// Backward column sums (gw, gb) are accumulated in registers over the rows a workgroup walks, written as one partial
// row per workgroup and finished by norm_colreduce_kernel in a fixed order (no atomics).
#include "oq_common.h"
#include "oq_quant_dev.h"

namespace {
constexpr int MAXCH = 4;

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    if (nw == 1) return v;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float a = red[0];
    for (int k = 1; k < nw; ++k) a += red[k];
    return a;
}

template <typename T, int NCH>
__global__ void __launch_bounds__(1024) norm_fwd_kernel(const T* x, int64_t rows, int64_t cols, const float* w,
                                                        const float* b, float eps, int ln, T* y, float* rstd_out,
                                                        float* mean_out) {
    __shared__ float red[16];
    const int t = threadIdx.x, BT = blockDim.x;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        float v[NCH][8];
        bool valid[NCH];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int64_t c0 = ((int64_t)j * BT + t) * 8;
            valid[j] = c0 < cols;
            if (valid[j]) {
                Vec8<T>::load(x + r * cols + c0, v[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += ln ? v[j][i] : v[j][i] * v[j][i];
            }
        }
        s = block_sum(s, red);
        float mean = 0.f, var;
        if (ln) {
            mean = s / (float)cols;
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                if (valid[j]) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float d = v[j][i] - mean; s2 += d * d; }
                }
            var = block_sum(s2, red) / (float)cols;
        } else {
            var = s / (float)cols;
        }
        const float rstd = rsqrtf(var + eps);
        if (t == 0) {
            rstd_out[r] = rstd;
            if (mean_out) mean_out[r] = mean;
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (!valid[j]) continue;
            const int64_t c0 = ((int64_t)j * BT + t) * 8;
            float o[8], wv[8], bv[8];
            Vec8<float>::load(w + c0, wv);               // 32-byte vector reads (were 16 scalar loads per chunk)
            if (b) {
                Vec8<float>::load(b + c0, bv);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (v[j][i] - mean) * rstd;
                o[i] = wv[i] * xh + bv[i];
            }
            Vec8<T>::store(y + r * cols + c0, o);
        }
    }
}

template <typename T, int NCH>
__global__ void __launch_bounds__(512) norm_bwd_kernel(const T* x, const T* gy, int64_t rows, int64_t cols,
                                                       const float* w, const float* rstd_in, const float* mean_in,
                                                       int ln, T* gx, float* ws, int want_b, const T* gx_add) {
    __shared__ float red[16];
    const int t = threadIdx.x, BT = blockDim.x;
    float aw[NCH][8], ab[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) { aw[j][i] = 0.f; ab[j][i] = 0.f; }
    float wv[NCH][8];                       // the norm weight of this thread's columns: loaded once, as vectors
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int64_t c0 = ((int64_t)j * BT + t) * 8;
        if (c0 < cols) {
            Vec8<float>::load(w + c0, wv[j]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[j][i] = 0.f;
        }
    }
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float rstd = rstd_in[r];
        const float mean = (ln && mean_in) ? mean_in[r] : 0.f;
        float xh[NCH][8], gh[NCH][8];   // normalised x, and gy*w
        bool valid[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int64_t c0 = ((int64_t)j * BT + t) * 8;
            valid[j] = c0 < cols;
            if (valid[j]) {
                float g[8];
                Vec8<T>::load(x + r * cols + c0, xh[j]);
                Vec8<T>::load(gy + r * cols + c0, g);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xh[j][i] = (xh[j][i] - mean) * rstd;
                    aw[j][i] += g[i] * xh[j][i];
                    ab[j][i] += g[i];
                    gh[j][i] = g[i] * wv[j][i];
                    s1 += gh[j][i];
                    s2 += gh[j][i] * xh[j][i];
                }
            }
        }
        s2 = block_sum(s2, red) / (float)cols;
        s1 = ln ? block_sum(s1, red) / (float)cols : 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (!valid[j]) continue;
            const int64_t c0 = ((int64_t)j * BT + t) * 8;
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (gh[j][i] - s1 - xh[j][i] * s2) * rstd;
            if (gx_add) {          // gradient that reached x on the residual path: added here instead of by a separate launch
                float ga[8];
                Vec8<T>::load(gx_add + r * cols + c0, ga);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += ga[i];
            }
            Vec8<T>::store(gx + r * cols + c0, o);
        }
    }
    float* pw = ws + (int64_t)blockIdx.x * cols;
    float* pb = ws + ((int64_t)gridDim.x + blockIdx.x) * cols;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int64_t c0 = ((int64_t)j * BT + t) * 8;
        if (c0 < cols) {
            Vec8<float>::store(pw + c0, aw[j]);
            if (want_b) Vec8<float>::store(pb + c0, ab[j]);
        }
    }
}

__global__ void __launch_bounds__(256) norm_colreduce_kernel(const float* part, int nblocks, int64_t cols, float* out0,
                                                             float* out1) {
    const float* src = part + (int64_t)blockIdx.y * nblocks * cols;
    float* out = blockIdx.y == 0 ? out0 : out1;
    if (!out) return;
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t c = (int64_t)blockIdx.x * 16 + cl;
    __shared__ float red[16][17];
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    if (c < cols) {          // 32 independent loads in flight per lane (see colreduce_kernel in oq_quant.hip)
        for (int base = 0; base < nblocks; base += 512) {
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int b = base + rg + 16 * k;
                v[k] = b < nblocks ? src[(int64_t)b * cols + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k & 7] += v[k];
        }
    }
    red[rg][cl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (rg == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        out[c] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// norm -> per-token fake quant in one kernel per direction
// ---------------------------------------------------------------------------------------------------
// y = fake_quant_per_token(norm(x)): OmniLlamaRMSNorm / OmniLayerNorm (quantize/omni_norm.py:26-34,52-63) followed by
// the act_quantizer of the first QuantLinear behind it (quantize/int_linear.py:59-60 -> quantize/quantizer.py:84-147).
// The normalised row never goes through memory and reaches the quantiser in fp32 (unfused it was stored in bf16 right
// in front of the 4-bit rounding decision).  Geometry of oq_rowq.hip: nw waves per row (2 for hidden sizes up to 4096),
// up to 8 chunks of 8 per lane, one LDS exchange per row-wide reduction; the norm weight / bias live in LDS.
// Backward: recomputes h = norm(x), applies the quantiser's closed-form gradient (clip mask, straight-through rounding,
// amax / amin tie terms) to get dL/dh, then the norm backward; gw / gb column sums accumulate in registers (lane ->
// column ownership is fixed) and leave as one partial row per workgroup (norm_colreduce_kernel finishes).
struct NQ {
    const void* x;
    void* y;
    const float *w, *b;
    int64_t rows, cols;
    float eps;
    int ln, nbits;
    float inv_q;
    float *rstd, *mean, *scale, *zp, *xmin, *xmax;
    const void* g;
    const void* g2;      // optional further gradients of the same output (sibling consumers): summed while loading
    const void* g3;
    void* gx;
    const void* gx_add;
    float* ws;
    int want_b;
    int8_t* codes;       // integer side channel of the forward (see FQ::codes / csum in oq_quant_dev.h); optional
    float* csum;
};

template <typename TIN, typename TOUT, int CH, bool LN>
__global__ void __launch_bounds__(512) normq_fwd_kernel(NQ p, int nw, int chn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int rpb = wpb / nw;
    const int rslot = wid / nw, wsub = wid - rslot * nw;
    const int K = (int)p.cols;
    const int nchunks = K >> 3;
    float* w_s = smem;
    float* b_s = smem + K;
    float* red = smem + 2 * K;
    const float Q = (float)((1 << p.nbits) - 1);
    for (int i = threadIdx.x * 4; i < K; i += blockDim.x * 4) {
        *reinterpret_cast<f32x4*>(w_s + i) = *reinterpret_cast<const f32x4*>(p.w + i);
        *reinterpret_cast<f32x4*>(b_s + i) = p.b ? *reinterpret_cast<const f32x4*>(p.b + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    bool valid[CH];
    int cc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = (j * nw + wsub) * 64 + lane;
        valid[j] = c < nchunks;
        cc[j] = (valid[j] ? c : nchunks - 1) * 8;
    }
    const TIN* xbase = reinterpret_cast<const TIN*>(p.x);
    TOUT* ybase = reinterpret_cast<TOUT*>(p.y);
    const float invK = 1.f / (float)K;
    int par = 0;
    // integer side channel: the row's code sum rides in a free slot of the NEXT row's first exchange (several waves per row)
    const bool want_codes = p.codes != nullptr;
    const bool eight = p.nbits == 8;
    int64_t pend_r = -1;
    float pend_v = 0.f;
    for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < p.rows; r0 += (int64_t)gridDim.x * rpb) {
        int64_t r = r0 + rslot;
        if (r >= p.rows) r = p.rows - 1;
        float v[CH][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (j < chn) {
                Vec8<TIN>::load(xbase + r * K + cc[j], v[j]);
                float sc = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) sc = LN ? sc + v[j][i] : fmaf(v[j][i], v[j][i], sc);
                s += valid[j] ? sc : 0.f;       // surplus lanes hold a copy of the row's last chunk
            }
        const int op0[4] = {0, 0, 0, 0};
        float e4[4] = {wave_sum(s), pend_v, 0.f, 0.f};
        if (nw > 1) {
            row_exchange(red, par, wid, rslot, nw, lane, e4, op0);
            if (want_codes && pend_r >= 0 && wsub == 0) p.csum[pend_r] = e4[1];      // the previous row's code sum
        }
        float mean = 0.f, var;
        if (LN) {
            mean = e4[0] * invK;
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (j < chn) {
                    float sc = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float d = v[j][i] - mean; sc = fmaf(d, d, sc); }
                    s2 += valid[j] ? sc : 0.f;
                }
            float f4[4] = {wave_sum(s2), 0.f, 0.f, 0.f};
            if (nw > 1) row_exchange(red, par, wid, rslot, nw, lane, f4, op0);
            var = f4[0] * invK;
        } else {
            var = e4[0] * invK;
        }
        const float rstd = rsqrtf(var + p.eps);
        float hi = -INFINITY, lo = INFINITY;
        uint64_t nanm = 0;
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (j < chn) {
                float wv[8], bv[8];
                Vec8<float>::load(w_s + cc[j], wv);
                Vec8<float>::load(b_s + cc[j], bv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = LN ? (v[j][i] - mean) * rstd : v[j][i] * rstd;
                    v[j][i] = wv[i] * xh + bv[i];
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    hi = vmax3(hi, v[j][i], v[j][i + 1]);
                    lo = vmin3(lo, v[j][i], v[j][i + 1]);
                    nanm |= __builtin_amdgcn_fcmpf(v[j][i], v[j][i + 1], 8);       // FCMP_UNO: either one is NaN
                }
            }
        float m4[4] = {wave_max(hi), wave_min(lo), nanm != 0 ? 1.f : 0.f, 0.f};
        if (nw > 1) {
            const int op[4] = {1, 2, 1, 0};
            row_exchange(red, par, wid, rslot, nw, lane, m4, op);
        }
        hi = m4[0]; lo = m4[1];
        const float bad = m4[2];
        if (bad != 0.f) { hi = NAN; lo = NAN; }
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, false, 0.f, 0.f, p.nbits, 0, p.inv_q, &inv_s);
        const bool regular = q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f && bad == 0.f;
        float csl = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (j < chn) {
                float yv[8];
                if (regular) {
                    float qv[8];
                    float cs8 = eight ? -1024.f : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float tq;
                        const float rq = rne_div(v[j][i], q.s, inv_s, &tq);
                        qv[i] = __builtin_amdgcn_fmed3f(rq + q.z, 0.f, Q);
                        yv[i] = (qv[i] - q.z) * q.s;
                        cs8 += qv[i];
                    }
                    if (want_codes) {
                        oq_store_codes8(p.codes + r * K + cc[j], qv, eight);
                        csl += valid[j] ? cs8 : 0.f;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float t = rne_ste(v[j][i] / q.s) + q.z;
                        t = (t != t) ? t : fminf(fmaxf(t, 0.f), Q);
                        yv[i] = (t - q.z) * q.s;
                    }
                    if (want_codes) {
                        const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        oq_store_codes8(p.codes + r * K + cc[j], zero8, false);
                    }
                }
                Vec8<TOUT>::store(ybase + r * K + cc[j], yv);
            }
        if (wsub == 0) {
            p.rstd[r] = rstd;
            if (p.mean) p.mean[r] = mean;
            p.scale[r] = q.s;
            p.zp[r] = q.z;
            p.xmin[r] = lo;
            p.xmax[r] = hi;
        }
        if (want_codes) {
            // a row whose scale is 0 / NaN is all-NaN in the reference (quirk Q1): a NaN code sum makes the integer GEMM say so
            const float cs = regular ? wave_sum(csl) : NAN;
            if (nw > 1) { pend_r = r; pend_v = cs; }
            else p.csum[r] = cs;
        }
    }
    if (want_codes && nw > 1) {       // flush the last row's code sum
        const int op0[4] = {0, 0, 0, 0};
        float e4[4] = {0.f, pend_v, 0.f, 0.f};
        row_exchange(red, par, wid, rslot, nw, lane, e4, op0);
        if (pend_r >= 0 && wsub == 0) p.csum[pend_r] = e4[1];
    }
}

template <typename TIN, typename TG, int CH, bool LN>
__global__ void __launch_bounds__(512, CH <= 2 ? 2 : 1) normq_bwd_kernel(NQ p, int nw, int chn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int rpb = wpb / nw;
    const int rslot = wid / nw, wsub = wid - rslot * nw;
    const int K = (int)p.cols;
    const int nchunks = K >> 3;
    float* w_s = smem;
    float* b_s = smem + K;
    float* red = smem + 2 * K;           // 64 floats
    float* fold = red + 64;              // [wpb][chn*512] for the final cross-wave sum of the column accumulators
    const float Q = (float)((1 << p.nbits) - 1);
    for (int i = threadIdx.x * 4; i < K; i += blockDim.x * 4) {
        *reinterpret_cast<f32x4*>(w_s + i) = *reinterpret_cast<const f32x4*>(p.w + i);
        *reinterpret_cast<f32x4*>(b_s + i) = p.b ? *reinterpret_cast<const f32x4*>(p.b + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    bool valid[CH];
    int cc[CH];
    float aw[CH][8], ab[CH][8];          // column accumulators; those of surplus lanes are never stored
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = (j * nw + wsub) * 64 + lane;
        valid[j] = c < nchunks;
        cc[j] = (valid[j] ? c : nchunks - 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { aw[j][i] = 0.f; ab[j][i] = 0.f; }
    }
    const TIN* xbase = reinterpret_cast<const TIN*>(p.x);
    const TG* gbase = reinterpret_cast<const TG*>(p.g);
    const TG* g2base = reinterpret_cast<const TG*>(p.g2);
    const TG* g3base = reinterpret_cast<const TG*>(p.g3);
    const TG* abase = reinterpret_cast<const TG*>(p.gx_add);
    TG* gxbase = reinterpret_cast<TG*>(p.gx);
    const float invK = 1.f / (float)K;
    int par = 0;
    for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < p.rows; r0 += (int64_t)gridDim.x * rpb) {
        const bool livew = r0 + rslot < p.rows;     // wave-uniform; the waves of a row past the end only keep the barriers company
        const int64_t r = livew ? r0 + rslot : p.rows - 1;
        const float rstd = p.rstd[r];
        const float mean = (LN && p.mean) ? p.mean[r] : 0.f;
        const float hi = p.xmax[r], lo = p.xmin[r];
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, false, 0.f, 0.f, p.nbits, 0, p.inv_q, &inv_s);
        // scale == 0 (quirk Q1): round_ste turns h / 0 = +-inf into NaN; a NaN zero-point inside round(t) + z gives the same
        // all-NaN row without a per-element select ((r - t) + t == r for every finite t)
        const float zr = q.s == 0.f ? NAN : q.z;
        float xh[CH][8], gh[CH][8];
        float gs = 0.f;
        int whi = 0, wlo = 0;
        uint32_t tieflag = 0;
        Raw8<TG> ga[CH];                 // the addend of the result (residual branch) is fetched with the row, not after it
        if (livew && abase) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (j < chn) ga[j].load(abase + r * K + cc[j]);
        }
        if (livew) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (j < chn) {
                    float G[8], wv[8], bv[8], hv[8];
                    Vec8<TIN>::load(xbase + r * K + cc[j], xh[j]);
                    Vec8<TG>::load(gbase + r * K + cc[j], G);
                    if (g2base) {       // dL/dy arrives in pieces (one per consumer of y): added here, in a fixed order
                        float G2[8];
                        Vec8<TG>::load(g2base + r * K + cc[j], G2);
#pragma unroll
                        for (int i = 0; i < 8; ++i) G[i] += G2[i];
                    }
                    if (g3base) {
                        float G3[8];
                        Vec8<TG>::load(g3base + r * K + cc[j], G3);
#pragma unroll
                        for (int i = 0; i < 8; ++i) G[i] += G3[i];
                    }
                    Vec8<float>::load(w_s + cc[j], wv);
                    Vec8<float>::load(b_s + cc[j], bv);
                    float gsc = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        xh[j][i] = LN ? (xh[j][i] - mean) * rstd : xh[j][i] * rstd;
                        const float h = wv[i] * xh[j][i] + bv[i];
                        hv[i] = h;
                        const float tq = h * inv_s;
                        const float u = rintf(tq) + zr;
                        const float qv = __builtin_amdgcn_fmed3f(u, 0.f, Q);
                        gh[j][i] = qv == u ? G[i] : 0.f;
                        gsc = fmaf(G[i], qv - q.z, gsc);
                        gsc = fmaf(-gh[j][i], tq, gsc);
                    }
                    gs += valid[j] ? gsc : 0.f;         // surplus lanes hold a copy of the row's last chunk
                    float cmx = hv[0], cmn = hv[0];
#pragma unroll
                    for (int i = 1; i < 7; i += 2) {
                        cmx = vmax3(cmx, hv[i], hv[i + 1]);
                        cmn = vmin3(cmn, hv[i], hv[i + 1]);
                    }
                    cmx = vmax(cmx, hv[7]);
                    cmn = vmin(cmn, hv[7]);
                    const uint64_t hit = __builtin_amdgcn_ballot_w64(valid[j] && (cmx == hi || cmn == lo));
                    if (hit != 0) {      // this wave holds an amax / amin element of the row in this chunk (rare)
                        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid[j]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            whi += __builtin_popcountll(__builtin_amdgcn_fcmpf(hv[i], hi, 1) & vmask);      // FCMP_OEQ
                            wlo += __builtin_popcountll(__builtin_amdgcn_fcmpf(hv[i], lo, 1) & vmask);
                        }
                        tieflag |= 1u << j;
                    }
                }
        }
        const int op0[4] = {0, 0, 0, 0};
        float e4[4] = {wave_sum(gs), (float)whi, (float)wlo, 0.f};
        if (nw > 1) row_exchange(red, par, wid, rslot, nw, lane, e4, op0);
        if (tieflag != 0) {
            const float tie_hi = (e4[0] / Q) / e4[1], tie_lo = -(e4[0] / Q) / e4[2];
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (j < chn && ((tieflag >> j) & 1u)) {
                    float wv[8], bv[8];
                    Vec8<float>::load(w_s + cc[j], wv);
                    Vec8<float>::load(b_s + cc[j], bv);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float h = wv[i] * xh[j][i] + bv[i];
                        if (h == hi) gh[j][i] += tie_hi;
                        if (h == lo) gh[j][i] += tie_lo;
                    }
                }
        }
        // ---- norm backward on dL/dh = gh ----
        float s1 = 0.f, s2 = 0.f;
        if (livew) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (j < chn) {
                    float wv[8];
                    Vec8<float>::load(w_s + cc[j], wv);
                    float s1c = 0.f, s2c = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        aw[j][i] = fmaf(gh[j][i], xh[j][i], aw[j][i]);
                        ab[j][i] += gh[j][i];
                        gh[j][i] = gh[j][i] * wv[i];
                        s1c += gh[j][i];
                        s2c = fmaf(gh[j][i], xh[j][i], s2c);
                    }
                    s1 += valid[j] ? s1c : 0.f;
                    s2 += valid[j] ? s2c : 0.f;
                }
        }
        float f4[4] = {wave_sum(s1), wave_sum(s2), 0.f, 0.f};
        if (nw > 1) row_exchange(red, par, wid, rslot, nw, lane, f4, op0);
        const float m1 = LN ? f4[0] * invK : 0.f, m2 = f4[1] * invK;
        if (livew) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (j < chn && valid[j]) {      // surplus lanes hold a copy of the last chunk: they must not store
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = LN ? (gh[j][i] - m1 - xh[j][i] * m2) * rstd : (gh[j][i] - xh[j][i] * m2) * rstd;
                    if (abase) {
                        float gav[8];
                        ga[j].unpack(gav);
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] += gav[i];
                    }
                    Vec8<TG>::store(gxbase + r * K + cc[j], o);
                }
        }
    }
    // ---- column sums: waves that own the same columns (same wsub) are added in slot order, one partial row per workgroup
    const int slab = chn * 512;
    float* pw = p.ws + (int64_t)blockIdx.x * K;
    float* pb = p.ws + ((int64_t)gridDim.x + blockIdx.x) * K;
    for (int pass = 0; pass < (p.want_b ? 2 : 1); ++pass) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (j < chn) {
#pragma unroll
                for (int i = 0; i < 8; ++i) fold[wid * slab + (j * 8 + i) * 64 + lane] = pass == 0 ? aw[j][i] : ab[j][i];
            }
        __syncthreads();
        if (rslot == 0) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (j < chn && valid[j]) {
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float a = 0.f;
                        for (int rs = 0; rs < rpb; ++rs) a += fold[(rs * nw + wsub) * slab + (j * 8 + i) * 64 + lane];
                        o[i] = a;
                    }
                    Vec8<float>::store((pass == 0 ? pw : pb) + cc[j], o);
                }
        }
    }
}

constexpr int64_t NORM_BWD_BLOCKS = 512;

int norm_threads(int64_t cols) {
    const int64_t lanes = (cols + 7) / 8;
    if (lanes <= 64) return 64;
    if (lanes <= 128) return 128;
    if (lanes <= 256) return 256;
    if (cols <= 8 * 512 * MAXCH) return 512;
    return 1024;
}
}  // namespace

extern "C" int oq_norm_fwd(const void* x, int dtype, int64_t rows, int64_t cols, const float* w, const float* b,
                           float eps, int is_layernorm, void* y, float* rstd, float* mean, void* stream) {
    OQ_CHECK_ARG(x && y && w && rstd, "oq_norm_fwd: null pointer");
    OQ_CHECK_ARG(oq_aligned16(w) && oq_aligned16(b), "oq_norm_fwd: weight / bias must be 16-byte aligned");
    OQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 8 * 1024 * MAXCH, "oq_norm_fwd: cols %lld", (long long)cols);
    OQ_CHECK_ARG(!is_layernorm || mean, "oq_norm_fwd: layernorm needs mean buffer");
    const int bt = norm_threads(cols);
    const int64_t grid = rows < 8192 ? rows : 8192;
    hipStream_t st = (hipStream_t)stream;
    // chunks of 8 elements per thread: 1, 2 or 4 (a template parameter: the row lives in registers, and sizing every
    // instantiation for 4 chunks cost half the occupancy at hidden sizes <= 4096)
    const int64_t per = ((cols + 7) / 8 + bt - 1) / bt;
    const int nch = per <= 1 ? 1 : (per <= 2 ? 2 : 4);
#define NORM_FWD(T_, N_) hipLaunchKernelGGL((norm_fwd_kernel<T_, N_>), dim3(grid), dim3(bt), 0, st, (const T_*)x, rows, cols, w, b, eps, is_layernorm, (T_*)y, rstd, mean)
    if (dtype == OQ_F32) {
        if (nch == 1) NORM_FWD(float, 1); else if (nch == 2) NORM_FWD(float, 2); else NORM_FWD(float, 4);
    } else if (dtype == OQ_BF16) {
        if (nch == 1) NORM_FWD(bf16_t, 1); else if (nch == 2) NORM_FWD(bf16_t, 2); else NORM_FWD(bf16_t, 4);
    } else {
        oq_set_error("oq_norm_fwd: dtype %d", dtype);
        return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_norm_fwd");
    return OQ_OK;
}

extern "C" int oq_norm_bwd(const void* x, const void* gy, int dtype, int64_t rows, int64_t cols, const float* w,
                           const float* rstd, const float* mean, int is_layernorm, void* gx, float* gw, float* gb,
                           const void* gx_addend, float* workspace, int64_t workspace_floats, void* stream) {
    OQ_CHECK_ARG(x && gy && gx && w && rstd, "oq_norm_bwd: null pointer");
    OQ_CHECK_ARG(oq_aligned16(w), "oq_norm_bwd: weight must be 16-byte aligned");
    OQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 8 * 512 * MAXCH, "oq_norm_bwd: cols %lld", (long long)cols);
    int bt = norm_threads(cols);
    if (bt > 512) bt = 512;
    const int64_t grid = rows < NORM_BWD_BLOCKS ? rows : NORM_BWD_BLOCKS;
    OQ_CHECK_ARG(gw && workspace && workspace_floats >= 2 * grid * cols, "oq_norm_bwd: gw and a workspace of %lld floats are required",
                 (long long)(2 * grid * cols));
    hipStream_t st = (hipStream_t)stream;
    const int64_t per = ((cols + 7) / 8 + bt - 1) / bt;
    const int nch = per <= 1 ? 1 : (per <= 2 ? 2 : 4);
#define NORM_BWD(T_, N_) hipLaunchKernelGGL((norm_bwd_kernel<T_, N_>), dim3(grid), dim3(bt), 0, st, (const T_*)x, (const T_*)gy, rows, cols, w, rstd, mean, is_layernorm, (T_*)gx, workspace, gb ? 1 : 0, (const T_*)gx_addend)
    if (dtype == OQ_F32) {
        if (nch == 1) NORM_BWD(float, 1); else if (nch == 2) NORM_BWD(float, 2); else NORM_BWD(float, 4);
    } else if (dtype == OQ_BF16) {
        if (nch == 1) NORM_BWD(bf16_t, 1); else if (nch == 2) NORM_BWD(bf16_t, 2); else NORM_BWD(bf16_t, 4);
    } else {
        oq_set_error("oq_norm_bwd: dtype %d", dtype);
        return OQ_E_UNSUPPORTED;
    }
    if (dtype == OQ_F32 || dtype == OQ_BF16) {
        const dim3 rg((unsigned)((cols + 15) / 16), gb ? 2 : 1);
        hipLaunchKernelGGL(norm_colreduce_kernel, rg, dim3(256), 0, st, workspace, (int)grid, cols, gw, gb);
    }
    OQ_CHECK_LAUNCH("oq_norm_bwd");
    return OQ_OK;
}

extern "C" int64_t oq_norm_bwd_workspace(int64_t rows, int64_t cols) {
    const int64_t grid = rows < NORM_BWD_BLOCKS ? rows : NORM_BWD_BLOCKS;
    return 2 * grid * cols;
}

extern "C" int64_t oq_norm_quant_supported(int dtype, int64_t cols) {
    RowGeo g;
    return (dtype == OQ_BF16 || dtype == OQ_F32) && row_geo(cols, 0, &g) && cols <= 8192;
}

// Backward geometry.  Rows of >= 4096 elements: 4 waves per row (2 chunks per lane at 4096: 134 VGPRs instead of 208) in
// workgroups of 8 waves, ONE workgroup per CU -- what a workgroup pays once (staging weight / bias in LDS, folding and
// writing its partial column sums) outweighs the extra workgroups' parallelism at the 2048 rows of a calibration sample:
// in-step 43.4 + 7.9 us (2 waves per row, 512 workgroups of 4 waves) -> 38.6 + 5.3 us per launch (tools/_nq_sweep.sh).
static bool normq_bwd_geo(int64_t cols, RowGeo* g) {
    const int nw_env = (int)env_i("OQ_NORMQ_BWD_NW", 0);
    if (!row_geo(cols, nw_env > 0 ? nw_env : (cols >= 4096 ? 4 : 0), g)) return false;
    int wpb = (int)env_i("OQ_NORMQ_BWD_WPB", 0);            // waves per workgroup (rows per workgroup = wpb / nw)
    if (wpb == 0 && g->nw == 4) wpb = 8;
    if (wpb >= g->nw && wpb <= 8 && wpb % g->nw == 0) g->wpb = wpb;
    return true;
}

static int64_t normq_bwd_grid(int64_t rows, const RowGeo& g) {
    const int rpb = g.wpb / g.nw;
    const int64_t need = (rows + rpb - 1) / rpb;
    const int64_t cap = env_i("OQ_NORMQ_BWD_BLOCKS", g.wpb == 8 ? n_cus() : 512);
    return need < cap ? need : cap;
}

extern "C" int64_t oq_norm_quant_bwd_workspace(int64_t rows, int64_t cols) {
    RowGeo g;
    if (!normq_bwd_geo(cols, &g)) return 0;
    return 2 * normq_bwd_grid(rows, g) * cols;
}

extern "C" int oq_norm_quant_fwd(const void* x, int dtype, int64_t rows, int64_t cols, const float* w, const float* b, float eps,
                                 int is_layernorm, int nbits, void* y, int y_dtype, float* rstd, float* mean, float* scale,
                                 float* zp, float* xmin, float* xmax, void* codes, float* csum, void* stream) {
    OQ_CHECK_ARG(x && y && w && rstd && scale && zp && xmin && xmax, "oq_norm_quant_fwd: null pointer");
    OQ_CHECK_ARG((codes == nullptr) == (csum == nullptr) && (!codes || nbits <= 8),
                 "oq_norm_quant_fwd: codes and csum go together and need a grid of at most 8 bits");
    OQ_CHECK_ARG(y_dtype == dtype || (dtype == OQ_F32 && y_dtype == OQ_BF16),
                 "oq_norm_quant_fwd: output dtype %d for input dtype %d (same, or bf16 from f32)", y_dtype, dtype);
    OQ_CHECK_ARG(oq_aligned16(x) && oq_aligned16(y) && oq_aligned16(w) && oq_aligned16(b), "oq_norm_quant_fwd: 16-byte alignment");
    OQ_CHECK_ARG(rows > 0 && nbits >= 2 && nbits < 16, "oq_norm_quant_fwd: rows %lld, bitwidth %d", (long long)rows, nbits);
    OQ_CHECK_ARG(!is_layernorm || mean, "oq_norm_quant_fwd: layernorm needs the mean buffer");
    RowGeo g;
    if (!oq_norm_quant_supported(dtype, cols) || !row_geo(cols, (int)env_i("OQ_NORMQ_FWD_NW", 0), &g)) {
        oq_set_error("oq_norm_quant_fwd: dtype %d / %lld columns unsupported (bf16 or f32, 512 .. 8192, multiple of 8)", dtype, (long long)cols);
        return OQ_E_UNSUPPORTED;
    }
    NQ p{};
    p.x = x; p.y = y; p.w = w; p.b = b; p.rows = rows; p.cols = cols; p.eps = eps; p.ln = is_layernorm; p.nbits = nbits;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.rstd = rstd; p.mean = mean; p.scale = scale; p.zp = zp; p.xmin = xmin; p.xmax = xmax;
    p.codes = (int8_t*)codes; p.csum = csum;
    const int rpb = g.wpb / g.nw;
    const int64_t need = (rows + rpb - 1) / rpb, cap = (int64_t)n_cus() * 8;
    const dim3 grid((unsigned)(need < cap ? need : cap)), blk((unsigned)(g.wpb * 64));
    const size_t smem = sizeof(float) * (2 * cols + 64);
    hipStream_t st = (hipStream_t)stream;
#define NQ_FWD(T_, TY_, LN_) do { if (g.chn <= 4) hipLaunchKernelGGL((normq_fwd_kernel<T_, TY_, 4, LN_>), grid, blk, smem, st, p, g.nw, g.chn); \
                             else hipLaunchKernelGGL((normq_fwd_kernel<T_, TY_, 8, LN_>), grid, blk, smem, st, p, g.nw, g.chn); } while (0)
    if (dtype == OQ_BF16) { if (is_layernorm) NQ_FWD(bf16_t, bf16_t, true); else NQ_FWD(bf16_t, bf16_t, false); }
    else if (y_dtype == OQ_BF16) { if (is_layernorm) NQ_FWD(float, bf16_t, true); else NQ_FWD(float, bf16_t, false); }
    else { if (is_layernorm) NQ_FWD(float, float, true); else NQ_FWD(float, float, false); }
    OQ_CHECK_LAUNCH("oq_norm_quant_fwd");
    return OQ_OK;
}

extern "C" int oq_norm_quant_bwd(const void* x, const void* g_, const void* g2, const void* g3, int dtype, int g_dtype, int64_t rows, int64_t cols, const float* w,
                                 const float* b, const float* rstd, const float* mean, int is_layernorm, int nbits,
                                 const float* xmin, const float* xmax, void* gx, float* gw, float* gb, const void* gx_addend,
                                 float* workspace, int64_t workspace_floats, void* stream) {
    OQ_CHECK_ARG(x && g_ && gx && w && rstd && xmin && xmax && gw, "oq_norm_quant_bwd: null pointer");
    OQ_CHECK_ARG(oq_aligned16(x) && oq_aligned16(g_) && oq_aligned16(gx) && oq_aligned16(w) && oq_aligned16(b) && oq_aligned16(gx_addend),
                 "oq_norm_quant_bwd: 16-byte alignment");
    OQ_CHECK_ARG(rows > 0 && nbits >= 2 && nbits < 16, "oq_norm_quant_bwd: rows %lld, bitwidth %d", (long long)rows, nbits);
    OQ_CHECK_ARG(g_dtype == dtype || (dtype == OQ_F32 && g_dtype == OQ_BF16),
                 "oq_norm_quant_bwd: gradient dtype %d for input dtype %d (same, or bf16 with an f32 input)", g_dtype, dtype);
    RowGeo g;
    if (!oq_norm_quant_supported(dtype, cols) || !normq_bwd_geo(cols, &g)) {
        oq_set_error("oq_norm_quant_bwd: dtype %d / %lld columns unsupported", dtype, (long long)cols);
        return OQ_E_UNSUPPORTED;
    }
    const int64_t nblk = normq_bwd_grid(rows, g);
    OQ_CHECK_ARG(workspace && workspace_floats >= 2 * nblk * cols, "oq_norm_quant_bwd: workspace of %lld floats needed",
                 (long long)(2 * nblk * cols));
    NQ p{};
    OQ_CHECK_ARG(oq_aligned16(g2) && oq_aligned16(g3) && (g2 || !g3), "oq_norm_quant_bwd: g2 / g3 alignment (g3 needs g2)");
    p.x = x; p.g = g_; p.g2 = g2; p.g3 = g3; p.gx = gx; p.gx_add = gx_addend; p.w = w; p.b = b; p.rows = rows; p.cols = cols; p.ln = is_layernorm;
    p.nbits = nbits; p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.rstd = const_cast<float*>(rstd); p.mean = const_cast<float*>(mean);
    p.xmin = const_cast<float*>(xmin); p.xmax = const_cast<float*>(xmax); p.ws = workspace; p.want_b = gb ? 1 : 0;
    const dim3 grid((unsigned)nblk), blk((unsigned)(g.wpb * 64));
    const size_t smem = sizeof(float) * (2 * cols + 64 + (size_t)g.wpb * g.chn * 512);
    hipStream_t st = (hipStream_t)stream;
#define NQ_BWD_K(T_, TG_, C_, LN_) ((const void*)normq_bwd_kernel<T_, TG_, C_, LN_>)
#define NQ_BWD_SEL(T_, TG_) (g.chn <= 2 ? (is_layernorm ? NQ_BWD_K(T_, TG_, 2, true) : NQ_BWD_K(T_, TG_, 2, false))  \
                      : g.chn <= 4 ? (is_layernorm ? NQ_BWD_K(T_, TG_, 4, true) : NQ_BWD_K(T_, TG_, 4, false))  \
                                   : (is_layernorm ? NQ_BWD_K(T_, TG_, 8, true) : NQ_BWD_K(T_, TG_, 8, false)))
    const void* k = dtype == OQ_BF16 ? NQ_BWD_SEL(bf16_t, bf16_t) : (g_dtype == OQ_BF16 ? NQ_BWD_SEL(float, bf16_t) : NQ_BWD_SEL(float, float));
    if (smem > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
        oq_set_error("oq_norm_quant_bwd: cannot reserve %zu bytes of LDS", smem);
        return OQ_E_LAUNCH;
    }
    {
        int nw_ = g.nw, chn_ = g.chn;
        void* args[] = {&p, &nw_, &chn_};
        if (hipLaunchKernel(k, grid, blk, args, smem, st) != hipSuccess) {
            oq_set_error("oq_norm_quant_bwd: launch failed");
            return OQ_E_LAUNCH;
        }
    }
    const dim3 rg((unsigned)((cols + 15) / 16), gb ? 2 : 1);
    hipLaunchKernelGGL(norm_colreduce_kernel, rg, dim3(256), 0, st, workspace, (int)nblk, cols, gw, gb);
    OQ_CHECK_LAUNCH("oq_norm_quant_bwd");
    return OQ_OK;
}
