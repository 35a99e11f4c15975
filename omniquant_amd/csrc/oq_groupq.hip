// UniformAffineQuantizer on GROUPED weights without LET (W3A16g128, W2A16g64, ...): lanes-per-segment kernels.
//
// Replaces quantize/quantizer.py:84-147 (group reshape :123-129, per-group min / max, LWC clipping, scale / zero-point,
// round-clamp-dequant) and the part of its autograd the calibration needs for FROZEN weights: dL/d upbound_factor and
// dL/d lowbound_factor per group (SURVEY.md 8a closed form; no input gradient, so no amax / amin tie terms).
//
// Why a third kernel family: the segment kernels of oq_quant.hip give a group of 64 / 128 elements to 8 / 16 lanes with 8
// elements each, so the per-group arithmetic -- two sigmoids, the Markstein scale, the zero-point: ~150 instructions -- is
// repeated by every lane for 8 elements of payload (profiles/r3_step_breakdown_llama-2-70b-w2a16g64.txt: 2.3 ms backward +
// 1.0 ms forward per 70B block step, 1.5 / 3.4 TB/s).  Here a group belongs to LPS = group / 32 lanes with 32 elements each
// (the geometry of oq_ropeq.hip): lane l owns the 8-element chunks l, l + LPS, l + 2 LPS, l + 3 LPS, so one load instruction
// reads LPS x 16 contiguous bytes per group and 64 / LPS groups per wave; min / max / gradient sums are log2(LPS) DPP steps
// inside the group's lanes; nothing is exchanged through LDS and no wave waits for another.  A weight matrix is one flat
// array of groups (cols % group == 0, row-major: groups are contiguous).  Same helper arithmetic as the other families
// (oq_quant_dev.h): forward values are bit-identical.
#include <stdlib.h>
#include "oq_common.h"
#include "oq_quant_dev.h"

namespace {

constexpr int GNC = 4;      // chunks of 8 per lane

template <typename TIN, typename TOUT, int LPS>
__global__ void __launch_bounds__(256) gq_fwd_kernel(FQ p, int64_t nseg) {
    constexpr int SPW = 64 / LPS;                   // groups per wave
    const int lane = threadIdx.x & 63, l = lane & (LPS - 1);
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float Q = (float)((1 << p.nbits) - 1);
    const int seg = LPS * 32;
    const TIN* wbase = reinterpret_cast<const TIN*>(p.w);
    TOUT* ybase = reinterpret_cast<TOUT*>(p.y);
    const bool lwc = p.up != nullptr;
    for (int64_t s0 = wave * SPW; s0 < nseg; s0 += nwaves * SPW) {
        int64_t s = s0 + lane / LPS;
        if (s >= nseg) s = nseg - 1;                // surplus lane groups redo the last group (same values stored again)
        const TIN* px = wbase + s * seg;
        Raw8<TIN> raw[GNC];
#pragma unroll
        for (int c = 0; c < GNC; ++c) raw[c].load(px + (c * LPS + l) * 8);
        const float upl = lwc ? p.up[s] : 0.f, lowl = lwc ? p.low[s] : 0.f;
        float x[GNC][8];
        float hi = -INFINITY, lo = INFINITY;
        uint64_t nanm = 0;
#pragma unroll
        for (int c = 0; c < GNC; ++c) {
            raw[c].unpack(x[c]);
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                hi = vmax3(hi, x[c][i], x[c][i + 1]);
                lo = vmin3(lo, x[c][i], x[c][i + 1]);
                nanm |= __builtin_amdgcn_fcmpf(x[c][i], x[c][i + 1], 8);      // FCMP_UNO: either one is NaN
            }
        }
        float bad = ((nanm >> lane) & 1) ? 1.f : 0.f;
        hi = wave_max(hi, LPS);
        lo = wave_min(lo, LPS);
        bad = wave_max(bad, LPS);
        if (bad != 0.f) { hi = NAN; lo = NAN; }     // torch.amax / amin propagate NaN
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, lwc, upl, lowl, p.nbits, p.symmetric, p.inv_q, &inv_s);
        const bool regular = q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f && bad == 0.f;
        TOUT* py = ybase + s * seg;
#pragma unroll
        for (int c = 0; c < GNC; ++c) {
            float yv[8];
            if (regular) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float tq;
                    const float rq = rne_div(x[c][i], q.s, inv_s, &tq);
                    yv[i] = (__builtin_amdgcn_fmed3f(rq + q.z, 0.f, Q) - q.z) * q.s;
                }
            } else {                                 // constant group (scale 0, quirk Q1), inf / NaN inputs: the reference's op sequence
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = rne_ste(x[c][i] / q.s) + q.z;
                    v = (v != v) ? v : fminf(fmaxf(v, 0.f), Q);
                    yv[i] = (v - q.z) * q.s;
                }
            }
            Vec8<TOUT>::store(py + (c * LPS + l) * 8, yv);
        }
        if (l == 0) {
            p.scale[s] = q.s;
            p.zp[s] = q.z;
            p.xmin[s] = lo;
            p.xmax[s] = hi;
        }
    }
}

template <typename TIN, typename TG, int LPS>
__global__ void __launch_bounds__(256) gq_bwd_kernel(FQ p, int64_t nseg) {
    constexpr int SPW = 64 / LPS;
    const int lane = threadIdx.x & 63, l = lane & (LPS - 1);
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float Q = (float)((1 << p.nbits) - 1);
    const int seg = LPS * 32;
    const TIN* wbase = reinterpret_cast<const TIN*>(p.w);
    const TG* gbase = reinterpret_cast<const TG*>(p.g);
    const bool lwc = p.up != nullptr;
    for (int64_t s0 = wave * SPW; s0 < nseg; s0 += nwaves * SPW) {
        int64_t s = s0 + lane / LPS;
        if (s >= nseg) s = nseg - 1;
        Raw8<TIN> rw[GNC];
        Raw8<TG> rg[GNC];
#pragma unroll
        for (int c = 0; c < GNC; ++c) {
            rw[c].load(wbase + s * seg + (c * LPS + l) * 8);
            rg[c].load(gbase + s * seg + (c * LPS + l) * 8);
        }
        const float hi = p.xmax[s], lo = p.xmin[s];
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, lwc, lwc ? p.up[s] : 0.f, lwc ? p.low[s] : 0.f, p.nbits, p.symmetric, p.inv_q, &inv_s);
        // scale == 0 (quirk Q1): round_ste turns x / 0 = +-inf into NaN; a NaN zero-point inside round(t) + z gives the same
        // all-NaN group without a per-element select ((r - t) + t == r for every finite t)
        const float z = q.z, zr = q.s == 0.f ? NAN : z;
        float gs = 0.f;
#pragma unroll
        for (int c = 0; c < GNC; ++c) {
            float w[8], G[8];
            rw[c].unpack(w);
            rg[c].unpack(G);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float tq = w[i] * inv_s;
                const float u = rintf(tq) + zr;
                const float qv = __builtin_amdgcn_fmed3f(u, 0.f, Q);
                const float gi = qv == u ? G[i] : 0.f;              // inside [0, Q] (false for NaN)
                gs = fmaf(G[i], qv - z, gs);                        // d y / d s = (q - z) - m * x / s
                gs = fmaf(-gi, tq, gs);
            }
        }
        gs = wave_sum(gs, LPS);
        float ds_dhs, ds_dls;
        if (p.symmetric) {
            const float lvl = (float)((1 << (p.nbits - 1)) - 1);
            const float hs = q.su * q.hi, ls = q.sl * q.lo;
            const float a = fabsf(hs), b = fabsf(ls);
            const float raw = fmaxf(a, b) / lvl;
            const float pass = (raw >= 1e-5f && raw <= 1e4f) ? 1.f : 0.f;
            const float sh = hs > 0.f ? 1.f : (hs < 0.f ? -1.f : 0.f);
            const float sg = ls > 0.f ? 1.f : (ls < 0.f ? -1.f : 0.f);
            const float wa = a > b ? 1.f : (a == b ? 0.5f : 0.f);
            ds_dhs = pass * wa * sh / lvl;
            ds_dls = pass * (1.f - wa) * sg / lvl;
        } else {
            ds_dhs = 1.f / Q;
            ds_dls = -1.f / Q;
        }
        if (l == 0) {
            if (p.g_up) p.g_up[s] = (gs * ds_dhs) * q.hi * q.su * (1.f - q.su);
            if (p.g_low) p.g_low[s] = (gs * ds_dls) * q.lo * q.sl * (1.f - q.sl);
        }
    }
}

int64_t gq_env(const char* n, int64_t d) {
    const char* v = getenv(n);
    return v ? atoll(v) : d;
}

unsigned gq_grid(int64_t nseg, int lps) {
    const int64_t spw = 64 / lps;
    const int64_t need = (nseg + spw * 4 - 1) / (spw * 4);          // 4 waves per workgroup
    const int64_t cap = gq_env("OQ_GROUPQ_BLOCKS", 16384);
    return (unsigned)(need < cap ? need : cap);
}

}  // namespace

// groups of 32 * 2^k elements (32 .. 512), no LET, whole matrix = a flat array of groups
bool oq_groupq_eligible(const FQ& p, bool backward) {
    if (gq_env("OQ_GROUPQ", 1) == 0) return false;
    if (p.seg >= p.cols || p.seg < 32 || p.seg > 512 || p.seg % 32 != 0 || p.cols % p.seg != 0) return false;
    const int64_t lps = p.seg / 32;
    if (lps & (lps - 1)) return false;
    if (p.col_mul || p.row_div || p.row_mul || p.shift || p.wshift || p.codes) return false;
    if (p.nbits >= 16) return false;
    if (backward && (p.gx || p.g_col_mul || p.g_shift || p.g_row_div || p.g_row_mul || p.g_wshift)) return false;
    return true;
}

#define GQ_LPS_SWITCH(KERNEL, T0, T1)                                                                         \
    switch (lps) {                                                                                            \
        case 1: hipLaunchKernelGGL((KERNEL<T0, T1, 1>), grid, dim3(256), 0, st, p, nseg); break;              \
        case 2: hipLaunchKernelGGL((KERNEL<T0, T1, 2>), grid, dim3(256), 0, st, p, nseg); break;              \
        case 4: hipLaunchKernelGGL((KERNEL<T0, T1, 4>), grid, dim3(256), 0, st, p, nseg); break;              \
        case 8: hipLaunchKernelGGL((KERNEL<T0, T1, 8>), grid, dim3(256), 0, st, p, nseg); break;              \
        default: hipLaunchKernelGGL((KERNEL<T0, T1, 16>), grid, dim3(256), 0, st, p, nseg); break;            \
    }

// Return OQ_OK when launched, 1 when the dtype pair is not theirs (the caller runs the segment kernels), negative on error.
int oq_groupq_fwd(const FQ& p, int w_dtype, int y_dtype, void* stream) {
    const int lps = (int)(p.seg / 32);
    const int64_t nseg = p.rows * (p.cols / p.seg);
    const dim3 grid(gq_grid(nseg, lps));
    hipStream_t st = (hipStream_t)stream;
    switch (w_dtype * 3 + y_dtype) {
        case OQ_F32 * 3 + OQ_F32: GQ_LPS_SWITCH(gq_fwd_kernel, float, float); break;
        case OQ_F32 * 3 + OQ_BF16: GQ_LPS_SWITCH(gq_fwd_kernel, float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: GQ_LPS_SWITCH(gq_fwd_kernel, f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: GQ_LPS_SWITCH(gq_fwd_kernel, f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: GQ_LPS_SWITCH(gq_fwd_kernel, bf16_t, bf16_t); break;
        default: return 1;
    }
    OQ_CHECK_LAUNCH("oq_fakequant_fwd(groupq)");
    return OQ_OK;
}

int oq_groupq_bwd(const FQ& p, int w_dtype, int g_dtype, void* stream) {
    const int lps = (int)(p.seg / 32);
    const int64_t nseg = p.rows * (p.cols / p.seg);
    const dim3 grid(gq_grid(nseg, lps));
    hipStream_t st = (hipStream_t)stream;
    switch (w_dtype * 3 + g_dtype) {
        case OQ_F32 * 3 + OQ_F32: GQ_LPS_SWITCH(gq_bwd_kernel, float, float); break;
        case OQ_F32 * 3 + OQ_BF16: GQ_LPS_SWITCH(gq_bwd_kernel, float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: GQ_LPS_SWITCH(gq_bwd_kernel, f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: GQ_LPS_SWITCH(gq_bwd_kernel, f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: GQ_LPS_SWITCH(gq_bwd_kernel, bf16_t, bf16_t); break;
        default: return 1;
    }
    OQ_CHECK_LAUNCH("oq_fakequant_bwd(groupq)");
    return OQ_OK;
}
