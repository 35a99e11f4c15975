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
// rotation.  A (token, head) segment is 16 lanes x 8 elements: its min / max and gradient sums are 4-step DPP
// reductions, the rotation partner (element e +- 64) sits 8 lanes away -- loaded as a second 16-byte vector in the
// forward, fetched with one DPP row rotation per element in the backward.  Four segments per wave instruction.
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
};

constexpr int HD = 128, LPS = 16;        // lanes per segment

__device__ __forceinline__ float ror8(float v) {      // value of the lane 8 places around inside the 16-lane row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}

template <typename TIN>
__device__ __forceinline__ void rotated(const RQ& p, int64_t seg, int l, float (&x)[8]) {
    const TIN* px = reinterpret_cast<const TIN*>(p.x) + seg * HD;
    Vec8<TIN>::load(px + l * 8, x);
    if (p.cs) {
        float xp[8], c[8], s[8];
        Vec8<TIN>::load(px + (l ^ 8) * 8, xp);
        const int64_t t = (seg / p.nh) % p.T;
        Vec8<float>::load(p.cs + t * HD + l * 8, c);
        Vec8<float>::load(p.sn + t * HD + l * 8, s);
        const float sgn = l < 8 ? -1.f : 1.f;           // rotate_half = cat(-x2, x1)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * c[i] + (sgn * xp[i]) * s[i];
    }
}

template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) ropeq_fwd_kernel(RQ p) {
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int64_t nseg = p.rows * p.nh;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float Q = (float)((1 << p.nbits) - 1);
    TOUT* ybase = reinterpret_cast<TOUT*>(p.y);
    for (int64_t s0 = wave * 4; s0 < nseg; s0 += nwaves * 4) {
        int64_t seg = s0 + (lane >> 4);
        if (seg >= nseg) seg = nseg - 1;                // surplus groups redo the last segment (same values stored again)
        float x[8];
        rotated<TIN>(p, seg, l, x);
        float hi = -INFINITY, lo = INFINITY, bad = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            hi = vmax(hi, x[i]);
            lo = vmin(lo, x[i]);
            if (x[i] != x[i]) bad = 1.f;
        }
        hi = wave_max(hi, LPS);
        lo = wave_min(lo, LPS);
        bad = wave_max(bad, LPS);
        if (bad != 0.f) { hi = NAN; lo = NAN; }
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, false, 0.f, 0.f, p.nbits, 0, p.inv_q, &inv_s);
        float yv[8];
        if (q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f && bad == 0.f) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float tq;
                const float rq = rne_div(x[i], q.s, inv_s, &tq);
                yv[i] = (__builtin_amdgcn_fmed3f(rq + q.z, 0.f, Q) - q.z) * q.s;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = rne_ste(x[i] / q.s) + q.z;
                v = (v != v) ? v : fminf(fmaxf(v, 0.f), Q);
                yv[i] = (v - q.z) * q.s;
            }
        }
        Vec8<TOUT>::store(ybase + seg * HD + l * 8, yv);
        p.scale[seg] = q.s;       // the 16 lanes of a segment store the same value
        p.zp[seg] = q.z;
        p.xmin[seg] = lo;
        p.xmax[seg] = hi;
    }
}

template <typename TIN, typename TG>
__global__ void __launch_bounds__(256) ropeq_bwd_kernel(RQ p) {
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int64_t nseg = p.rows * p.nh;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float Q = (float)((1 << p.nbits) - 1);
    const TG* gbase = reinterpret_cast<const TG*>(p.g);
    TG* gxbase = reinterpret_cast<TG*>(p.gx);
    for (int64_t s0 = wave * 4; s0 < nseg; s0 += nwaves * 4) {
        int64_t seg = s0 + (lane >> 4);
        if (seg >= nseg) seg = nseg - 1;
        float x[8], G[8];
        rotated<TIN>(p, seg, l, x);
        Vec8<TG>::load(gbase + seg * HD + l * 8, G);
        const float hi = p.xmax[seg], lo = p.xmin[seg];
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, false, 0.f, 0.f, p.nbits, 0, p.inv_q, &inv_s);
        const bool regular = q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f;
        float gs = 0.f, chi = 0.f, clo = 0.f, gin[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float tq = x[i] * inv_s;
            const float u = (regular ? rintf(tq) : rne_ste(tq)) + q.z;
            const float qv = __builtin_amdgcn_fmed3f(u, 0.f, Q);
            const bool in = qv == u;
            gs = fmaf(G[i], (qv - q.z) - (in ? tq : 0.f), gs);
            chi += x[i] == hi ? 1.f : 0.f;
            clo += x[i] == lo ? 1.f : 0.f;
            gin[i] = in ? G[i] : 0.f;
        }
        gs = wave_sum(gs, LPS);
        chi = wave_sum(chi, LPS);
        clo = wave_sum(clo, LPS);
        const float tie_hi = (gs / Q) / chi, tie_lo = -(gs / Q) / clo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (x[i] == hi) gin[i] += tie_hi;
            if (x[i] == lo) gin[i] += tie_lo;
        }
        if (p.cs) {
            // transposed rotation: gx_e = g_e cos_e - sgn_e * g_partner * sin_e   (cos / sin are equal on both halves)
            float c[8], s[8];
            const int64_t t = (seg / p.nh) % p.T;
            Vec8<float>::load(p.cs + t * HD + l * 8, c);
            Vec8<float>::load(p.sn + t * HD + l * 8, s);
            const float sgn = l < 8 ? -1.f : 1.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gp = ror8(gin[i]);
                gin[i] = gin[i] * c[i] + (-(sgn * gp)) * s[i];
            }
        }
        Vec8<TG>::store(gxbase + seg * HD + l * 8, gin);
    }
}

int check(const char* fn, int64_t rows, int64_t T, int nh, int hd, int nbits, const float* cs, const float* sn) {
    OQ_CHECK_ARG(rows > 0 && T > 0 && nh > 0 && rows % T == 0, "%s: rows %lld must be a positive multiple of T %lld", fn,
                 (long long)rows, (long long)T);
    if (hd != HD) {
        oq_set_error("%s: head_dim %d unsupported (only 128)", fn, hd);
        return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_ARG(nbits >= 2 && nbits < 16, "%s: bitwidth %d", fn, nbits);
    OQ_CHECK_ARG((cs == nullptr) == (sn == nullptr), "%s: cos / sin must both be given or both NULL", fn);
    return OQ_OK;
}

unsigned rq_grid(int64_t nseg) {
    const int64_t need = (nseg + 15) / 16;          // 4 segments per wave, 4 waves per workgroup
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
