// RoPE + head-wise fake quant in one kernel per direction (q / k), and the plain head-wise fake quant of v.
//
// Replaces, for head_dim 128: apply_rotary_pos_emb (models/int_llama_layer.py:124-125; rotate_half formula of
// transformers 4.31) followed by qkt_matmul.quant_x1 / quant_x2 and pv_matmul.quant_x2 (models/int_llama_layer.py:
// 140-143,161; quantize/int_matmul.py:31-39 -> quantize/quantizer.py:84-147 per (token, head) over head_dim).
//
// Unfused, the rotated tensor makes a 2 x 16.8 MB round trip per direction and tensor and is rounded to bf16 right in
// front of a 4-bit rounding decision.  Here the projection's output (bf16 or fp32) is rotated in registers and
// quantised at once; the backward recomputes the rotation, applies the quantiser's closed-form gradient
// (oq_quant_dev.h arithmetic: same scale / zero-point / rounding as every other quantiser kernel) and the transposed
// rotation.  A (token, head) segment is 4 lanes x 32 elements: lane l owns the 8-element chunks l, l+4, l+8, l+12, so
// a load instruction reads 64 contiguous bytes per segment, the rotation partner (element e +- 64 = chunk j +- 8) is
// in the SAME lane's registers (no second load, no cross-lane traffic), min / max / gradient sums are 2-step DPP
// reductions inside a quad, and the per-segment arithmetic (scale, zero-point, tie terms: ~150 instructions) is
// amortised over 32 elements per lane instead of 8 (the first version used 16 lanes x 8 elements and spent more
// instructions on the segment constants than on the elements).  Sixteen segments per wave instruction.
#include "oq_common.h"
#include "oq_quant_dev.h"

namespace {

struct RQ {
    const void* x;      // [rows, nh, 128]
    void* y;
    const float* cs;    // [T, 128] or NULL (no rotation: v)
    const float* sn;
    int64_t rows;       // bs * T
    int64_t T;
    int nh, nbits;
    float inv_q;
    float *scale, *zp, *xmin, *xmax;      // [rows * nh]
    const void* g;
    void* gx;
    // merged q / k / v call: x (and gx) is [rows, nh, 128] with nh = nh0 + nh1 + nh2 heads; heads [0, nh0) are written to
    // y (read from g), [nh0, nh0 + nh1) to y1 (g1), the rest to y2 (g2); the first `nrope` heads are rotated
    void *y1, *y2;
    const void *g1, *g2;
    int nh0, nh1, nrope;
    int grid;           // forward: store the grid coordinates clamp(round(x/s) + z, 0, Q) - z instead of the values (x s)
};

struct HeadSel {        // which output / gradient tensor a head belongs to, and its index there
    int t, hl, nht;
};
__device__ __forceinline__ HeadSel head_sel(const RQ& p, int h) {
    HeadSel o;
    if (h < p.nh0) { o.t = 0; o.hl = h; o.nht = p.nh0; }
    else if (h < p.nh0 + p.nh1) { o.t = 1; o.hl = h - p.nh0; o.nht = p.nh1; }
    else { o.t = 2; o.hl = h - p.nh0 - p.nh1; o.nht = p.nh - p.nh0 - p.nh1; }
    return o;
}

constexpr int HD = 128, LPS = 4, NC = 4;      // lanes per segment, chunks of 8 per lane

// rotated (or plain) values of lane l's chunks; chunk c of lane l covers elements (c*4 + l)*8 .. +8
template <typename TIN>
__device__ __forceinline__ void rotated(const RQ& p, int64_t seg, int l, float (&x)[NC][8]) {
    const TIN* px = reinterpret_cast<const TIN*>(p.x) + seg * HD;
#pragma unroll
    for (int c = 0; c < NC; ++c) Vec8<TIN>::load(px + (c * 4 + l) * 8, x[c]);
    if (p.cs && (int)(seg % p.nh) < p.nrope) {
        const int64_t t = (seg / p.nh) % p.T;
        float r[NC][8];
#pragma unroll
        for (int c = 0; c < 2; ++c) {                   // cos / sin repeat on the two halves of the head
            float cv[8], sv[8];
            Vec8<float>::load(p.cs + t * HD + (c * 4 + l) * 8, cv);
            Vec8<float>::load(p.sn + t * HD + (c * 4 + l) * 8, sv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {               // rotate_half = cat(-x2, x1)
                r[c][i] = x[c][i] * cv[i] + (-x[c + 2][i]) * sv[i];
                r[c + 2][i] = x[c + 2][i] * cv[i] + x[c][i] * sv[i];
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[c][i] = r[c][i];
    }
}

template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) ropeq_fwd_kernel(RQ p) {
    const int lane = threadIdx.x & 63, l = lane & 3;
    const int64_t nseg = p.rows * p.nh;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float Q = (float)((1 << p.nbits) - 1);
    TOUT* ybase = reinterpret_cast<TOUT*>(p.y);
    for (int64_t s0 = wave * 16; s0 < nseg; s0 += nwaves * 16) {
        int64_t seg = s0 + (lane >> 2);
        if (seg >= nseg) seg = nseg - 1;                // surplus quads redo the last segment (same values stored again)
        float x[NC][8];
        rotated<TIN>(p, seg, l, x);
        const HeadSel hs = head_sel(p, (int)(seg % p.nh));
        TOUT* yseg = (hs.t == 0 ? ybase : reinterpret_cast<TOUT*>(hs.t == 1 ? p.y1 : p.y2)) + ((seg / p.nh) * hs.nht + hs.hl) * HD;
        if (p.nbits >= 16) {                            // identity grid (weight-only configurations): rotate and split only
#pragma unroll
            for (int c = 0; c < NC; ++c) Vec8<TOUT>::store(yseg + (c * 4 + l) * 8, x[c]);
            continue;
        }
        float hi = -INFINITY, lo = INFINITY;
        uint64_t nanm = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                hi = vmax3(hi, x[c][i], x[c][i + 1]);
                lo = vmin3(lo, x[c][i], x[c][i + 1]);
                nanm |= __builtin_amdgcn_fcmpf(x[c][i], x[c][i + 1], 8);      // FCMP_UNO: either one is NaN
            }
        float bad = ((nanm >> lane) & 1) ? 1.f : 0.f;
        hi = wave_max(hi, LPS);
        lo = wave_min(lo, LPS);
        bad = wave_max(bad, LPS);
        if (bad != 0.f) { hi = NAN; lo = NAN; }
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, false, 0.f, 0.f, p.nbits, 0, p.inv_q, &inv_s);
        const bool regular = q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f && bad == 0.f;
        const float os = p.grid ? 1.f : q.s;        // (an irregular segment is NaN either way)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float yv[8];
            if (regular) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float tq;
                    const float rq = rne_div(x[c][i], q.s, inv_s, &tq);
                    yv[i] = (__builtin_amdgcn_fmed3f(rq + q.z, 0.f, Q) - q.z) * os;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = rne_ste(x[c][i] / q.s) + q.z;
                    v = (v != v) ? v : fminf(fmaxf(v, 0.f), Q);
                    yv[i] = (v - q.z) * (p.grid ? (q.s != q.s ? NAN : 1.f) : q.s);
                }
            }
            Vec8<TOUT>::store(yseg + (c * 4 + l) * 8, yv);
        }
        if (l == 0) {
            p.scale[seg] = q.s;
            p.zp[seg] = q.z;
            p.xmin[seg] = lo;
            p.xmax[seg] = hi;
        }
    }
}

template <typename TIN, typename TG>
__global__ void __launch_bounds__(256) ropeq_bwd_kernel(RQ p) {
    const int lane = threadIdx.x & 63, l = lane & 3;
    const int64_t nseg = p.rows * p.nh;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float Q = (float)((1 << p.nbits) - 1);
    const TG* gbase = reinterpret_cast<const TG*>(p.g);
    TG* gxbase = reinterpret_cast<TG*>(p.gx);
    for (int64_t s0 = wave * 16; s0 < nseg; s0 += nwaves * 16) {
        int64_t seg = s0 + (lane >> 2);
        if (seg >= nseg) seg = nseg - 1;
        float x[NC][8], gin[NC][8];
        const bool ident = p.nbits >= 16;               // identity grid: dL/dx is the transposed rotation of dL/dy
        if (!ident) rotated<TIN>(p, seg, l, x);
        const float hi = ident ? 0.f : p.xmax[seg], lo = ident ? 0.f : p.xmin[seg];
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, false, 0.f, 0.f, ident ? 8 : p.nbits, 0, p.inv_q, &inv_s);
        // scale == 0 (quirk Q1): round_ste turns x / 0 = +-inf into NaN; a NaN zero-point inside round(t) + z gives the same
        // all-NaN segment without a per-element select ((r - t) + t == r for every finite t)
        const float zr = q.s == 0.f ? NAN : q.z;
        float gs = 0.f, chi = 0.f, clo = 0.f;
        const int head = (int)(seg % p.nh);
        const HeadSel hs = head_sel(p, head);
        const TG* gseg = (hs.t == 0 ? gbase : reinterpret_cast<const TG*>(hs.t == 1 ? p.g1 : p.g2)) + ((seg / p.nh) * hs.nht + hs.hl) * HD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float G[8];
            Vec8<TG>::load(gseg + (c * 4 + l) * 8, G);
            if (ident) {
#pragma unroll
                for (int i = 0; i < 8; ++i) gin[c][i] = G[i];
                continue;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float tq = x[c][i] * inv_s;
                const float u = rintf(tq) + zr;
                const float qv = __builtin_amdgcn_fmed3f(u, 0.f, Q);
                gin[c][i] = qv == u ? G[i] : 0.f;
                gs = fmaf(G[i], qv - q.z, gs);
                gs = fmaf(-gin[c][i], tq, gs);
                chi += x[c][i] == hi ? 1.f : 0.f;
                clo += x[c][i] == lo ? 1.f : 0.f;
            }
        }
        if (!ident) {
            gs = wave_sum(gs, LPS);
            chi = wave_sum(chi, LPS);
            clo = wave_sum(clo, LPS);
            const float tie_hi = (gs / Q) / chi, tie_lo = -(gs / Q) / clo;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (x[c][i] == hi) gin[c][i] += tie_hi;
                    if (x[c][i] == lo) gin[c][i] += tie_lo;
                }
        }
        if (p.cs && head < p.nrope) {
            // transposed rotation: gx_e = g_e cos_e - sgn_e * g_partner * sin_e   (cos / sin are equal on both halves)
            const int64_t t = (seg / p.nh) % p.T;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float cv[8], sv[8];
                Vec8<float>::load(p.cs + t * HD + (c * 4 + l) * 8, cv);
                Vec8<float>::load(p.sn + t * HD + (c * 4 + l) * 8, sv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float ga = gin[c][i], gb = gin[c + 2][i];
                    gin[c][i] = ga * cv[i] + gb * sv[i];              // first half: sgn = -1, partner = second half
                    gin[c + 2][i] = gb * cv[i] + (-ga) * sv[i];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) Vec8<TG>::store(gxbase + seg * HD + (c * 4 + l) * 8, gin[c]);
    }
}

int check(const char* fn, int64_t rows, int64_t T, int nh, int hd, int nbits, const float* cs, const float* sn, bool ident_ok = false) {
    OQ_CHECK_ARG(rows > 0 && T > 0 && nh > 0 && rows % T == 0, "%s: rows %lld must be a positive multiple of T %lld", fn,
                 (long long)rows, (long long)T);
    if (hd != HD) {
        oq_set_error("%s: head_dim %d unsupported (only 128)", fn, hd);
        return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_ARG(nbits >= 2 && (nbits < 16 || (ident_ok && nbits == 16)), "%s: bitwidth %d", fn, nbits);
    OQ_CHECK_ARG((cs == nullptr) == (sn == nullptr), "%s: cos / sin must both be given or both NULL", fn);
    return OQ_OK;
}

unsigned rq_grid(int64_t nseg) {
    const int64_t need = (nseg + 63) / 64;          // 16 segments per wave, 4 waves per workgroup
    return (unsigned)(need < 16384 ? need : 16384);
}

}  // namespace

extern "C" int64_t oq_rope_quant_supported(int dtype, int hd) { return (dtype == OQ_BF16 || dtype == OQ_F32) && hd == HD; }

extern "C" int oq_rope_quant_fwd(const void* x, int x_dtype, int64_t rows, int64_t T, int nh, int hd, const float* cos,
                                 const float* sin, int nbits, void* y, int y_dtype, float* scale, float* zp, float* xmin,
                                 float* xmax, void* stream) {
    const int rc = check("oq_rope_quant_fwd", rows, T, nh, hd, nbits, cos, sin);
    if (rc) return rc;
    OQ_CHECK_ARG(x && y && scale && zp && xmin && xmax, "oq_rope_quant_fwd: null pointer");
    OQ_CHECK_ARG(oq_aligned16(x) && oq_aligned16(y) && oq_aligned16(cos) && oq_aligned16(sin), "oq_rope_quant_fwd: 16-byte alignment");
    RQ p{};
    p.x = x; p.y = y; p.cs = cos; p.sn = sin; p.rows = rows; p.T = T; p.nh = nh; p.nbits = nbits;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.scale = scale; p.zp = zp; p.xmin = xmin; p.xmax = xmax;
    p.nh0 = nh; p.nh1 = 0; p.nrope = cos ? nh : 0;
    const dim3 grid(rq_grid(rows * nh)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype * 3 + y_dtype) {
        case OQ_BF16 * 3 + OQ_BF16: hipLaunchKernelGGL((ropeq_fwd_kernel<bf16_t, bf16_t>), grid, blk, 0, st, p); break;
        case OQ_F32 * 3 + OQ_BF16: hipLaunchKernelGGL((ropeq_fwd_kernel<float, bf16_t>), grid, blk, 0, st, p); break;
        case OQ_F32 * 3 + OQ_F32: hipLaunchKernelGGL((ropeq_fwd_kernel<float, float>), grid, blk, 0, st, p); break;
        default:
            oq_set_error("oq_rope_quant_fwd: unsupported dtype pair in=%d out=%d", x_dtype, y_dtype);
            return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_rope_quant_fwd");
    return OQ_OK;
}

extern "C" int oq_rope_quant_bwd(const void* x, int x_dtype, int64_t rows, int64_t T, int nh, int hd, const float* cos,
                                 const float* sin, int nbits, const float* xmin, const float* xmax, const void* g,
                                 int g_dtype, void* gx, void* stream) {
    const int rc = check("oq_rope_quant_bwd", rows, T, nh, hd, nbits, cos, sin);
    if (rc) return rc;
    OQ_CHECK_ARG(x && g && gx && xmin && xmax, "oq_rope_quant_bwd: null pointer");
    OQ_CHECK_ARG(oq_aligned16(x) && oq_aligned16(g) && oq_aligned16(gx), "oq_rope_quant_bwd: 16-byte alignment");
    RQ p{};
    p.x = x; p.cs = cos; p.sn = sin; p.rows = rows; p.T = T; p.nh = nh; p.nbits = nbits;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.xmin = const_cast<float*>(xmin); p.xmax = const_cast<float*>(xmax); p.g = g; p.gx = gx;
    p.nh0 = nh; p.nh1 = 0; p.nrope = cos ? nh : 0;
    const dim3 grid(rq_grid(rows * nh)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype * 3 + g_dtype) {
        case OQ_BF16 * 3 + OQ_BF16: hipLaunchKernelGGL((ropeq_bwd_kernel<bf16_t, bf16_t>), grid, blk, 0, st, p); break;
        case OQ_F32 * 3 + OQ_BF16: hipLaunchKernelGGL((ropeq_bwd_kernel<float, bf16_t>), grid, blk, 0, st, p); break;
        case OQ_F32 * 3 + OQ_F32: hipLaunchKernelGGL((ropeq_bwd_kernel<float, float>), grid, blk, 0, st, p); break;
        default:
            oq_set_error("oq_rope_quant_bwd: unsupported dtype pair x=%d g=%d", x_dtype, g_dtype);
            return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_rope_quant_bwd");
    return OQ_OK;
}

// ---- q, k and v in one launch per direction (same arithmetic as three oq_rope_quant_* calls: q and k rotated, v not) ------
extern "C" int oq_qkv_rope_quant_fwd(const void* x, int x_dtype, int64_t rows, int64_t T, int nhq, int nhk, int nhv, int hd,
                                     const float* cos, const float* sin, int nbits, void* yq, void* yk, void* yv, int y_dtype,
                                     int out_grid, float* scale, float* zp, float* xmin, float* xmax, void* stream) {
    OQ_CHECK_ARG(nhq > 0 && nhk > 0 && nhv > 0, "oq_qkv_rope_quant_fwd: head counts %d / %d / %d", nhq, nhk, nhv);
    const int nh = nhq + nhk + nhv;
    const int rc = check("oq_qkv_rope_quant_fwd", rows, T, nh, hd, nbits, cos, sin, true);
    if (rc) return rc;
    OQ_CHECK_ARG(x && yq && yk && yv && cos && sin && (nbits == 16 || (scale && zp && xmin && xmax)), "oq_qkv_rope_quant_fwd: null pointer");
    OQ_CHECK_ARG(!out_grid || nbits <= 8, "oq_qkv_rope_quant_fwd: grid coordinates are exact in bf16 up to 8 bits, not %d", nbits);
    OQ_CHECK_ARG(oq_aligned16(x) && oq_aligned16(yq) && oq_aligned16(yk) && oq_aligned16(yv) && oq_aligned16(cos) && oq_aligned16(sin),
                 "oq_qkv_rope_quant_fwd: 16-byte alignment");
    RQ p{};
    p.x = x; p.y = yq; p.y1 = yk; p.y2 = yv; p.cs = cos; p.sn = sin; p.rows = rows; p.T = T; p.nh = nh; p.nbits = nbits;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.scale = scale; p.zp = zp; p.xmin = xmin; p.xmax = xmax;
    p.nh0 = nhq; p.nh1 = nhk; p.nrope = nhq + nhk;
    p.grid = out_grid ? 1 : 0;
    const dim3 grid(rq_grid(rows * nh)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype * 3 + y_dtype) {
        case OQ_BF16 * 3 + OQ_BF16: hipLaunchKernelGGL((ropeq_fwd_kernel<bf16_t, bf16_t>), grid, blk, 0, st, p); break;
        case OQ_F32 * 3 + OQ_BF16: hipLaunchKernelGGL((ropeq_fwd_kernel<float, bf16_t>), grid, blk, 0, st, p); break;
        case OQ_F32 * 3 + OQ_F32: hipLaunchKernelGGL((ropeq_fwd_kernel<float, float>), grid, blk, 0, st, p); break;
        default:
            oq_set_error("oq_qkv_rope_quant_fwd: unsupported dtype pair in=%d out=%d", x_dtype, y_dtype);
            return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_qkv_rope_quant_fwd");
    return OQ_OK;
}

extern "C" int oq_qkv_rope_quant_bwd(const void* x, int x_dtype, int64_t rows, int64_t T, int nhq, int nhk, int nhv, int hd,
                                     const float* cos, const float* sin, int nbits, const float* xmin, const float* xmax,
                                     const void* gq, const void* gk, const void* gv, int g_dtype, void* gx, void* stream) {
    OQ_CHECK_ARG(nhq > 0 && nhk > 0 && nhv > 0, "oq_qkv_rope_quant_bwd: head counts %d / %d / %d", nhq, nhk, nhv);
    const int nh = nhq + nhk + nhv;
    const int rc = check("oq_qkv_rope_quant_bwd", rows, T, nh, hd, nbits, cos, sin, true);
    if (rc) return rc;
    OQ_CHECK_ARG(x && gq && gk && gv && gx && cos && sin && (nbits == 16 || (xmin && xmax)), "oq_qkv_rope_quant_bwd: null pointer");
    OQ_CHECK_ARG(oq_aligned16(x) && oq_aligned16(gq) && oq_aligned16(gk) && oq_aligned16(gv) && oq_aligned16(gx),
                 "oq_qkv_rope_quant_bwd: 16-byte alignment");
    RQ p{};
    p.x = x; p.cs = cos; p.sn = sin; p.rows = rows; p.T = T; p.nh = nh; p.nbits = nbits;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.xmin = const_cast<float*>(xmin); p.xmax = const_cast<float*>(xmax); p.g = gq; p.g1 = gk; p.g2 = gv; p.gx = gx;
    p.nh0 = nhq; p.nh1 = nhk; p.nrope = nhq + nhk;
    const dim3 grid(rq_grid(rows * nh)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype * 3 + g_dtype) {
        case OQ_BF16 * 3 + OQ_BF16: hipLaunchKernelGGL((ropeq_bwd_kernel<bf16_t, bf16_t>), grid, blk, 0, st, p); break;
        case OQ_F32 * 3 + OQ_BF16: hipLaunchKernelGGL((ropeq_bwd_kernel<float, bf16_t>), grid, blk, 0, st, p); break;
        case OQ_F32 * 3 + OQ_F32: hipLaunchKernelGGL((ropeq_bwd_kernel<float, float>), grid, blk, 0, st, p); break;
        default:
            oq_set_error("oq_qkv_rope_quant_bwd: unsupported dtype pair x=%d g=%d", x_dtype, g_dtype);
            return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_qkv_rope_quant_bwd");
    return OQ_OK;
}
