// UniformAffineQuantizer, whole-row segments (per-channel weights, per-token activations), wave-per-row kernels.
//
// Replaces the same reference lines as oq_quant.hip (quantize/quantizer.py:84-147 + the weight side of
// models/transformation.py:24-69) for the case "one segment = one row" with rows of 512 .. 32768 elements.
//
// Why a second kernel family: the segment kernels of oq_quant.hip give a row to a whole WORKGROUP (16 elements per lane,
// two workgroup barriers per row).  rocprofv3 (profiles/r2_quant_sq_before.txt) showed them neither HBM- nor
// VALU-throughput-bound: 250 VGPRs -> 2 waves per SIMD, 44-67 VALU instructions per element (per-row work repeated by
// every wave for 16 elements), 41-44 % of the wave time in s_waitcnt / s_barrier.  Here a row belongs to ONE wave (or 2,
// 4, 8 waves for rows longer than 4096): 32-64 elements per lane amortise the per-row arithmetic, waves never wait for
// each other (rows <= 4096) and every wave keeps 8-16 KB of loads in flight.
//   * LET column vectors (col_mul, shift) live in LDS, staged once per workgroup;
//   * the backward's column gradients (g_col_mul, g_shift) are accumulated in LDS: each wave updates its OWN
//     accumulator slab with 16-byte LDS reads/writes (one wave per address, program order: deterministic), the slabs of
//     a workgroup are summed in a fixed order at the end and written as one partial row per workgroup (finished by
//     colreduce_kernel);
//   * the straight-through terms through max / min touch only the chunks that hold an element equal to the row max /
//     min: a scalar flag per chunk (from the tie-count compares) selects them, everything else is written in the main pass.
// Arithmetic (scale, zero-point, rounding with the exact-division fallback, tie sharing) is shared with oq_quant.hip
// through oq_quant_dev.h, so both families give identical values.
#include <stdlib.h>
#include "oq_common.h"
#include "oq_quant_dev.h"

namespace {

// acc[plane][chunk slot][4] += v[4 * plane .. ]: a wave's private accumulator slab, updated with plain 16-byte LDS reads
// and writes (only the owning wave touches it, in program order: deterministic).  The two planes keep consecutive lanes on
// consecutive 16-byte slots, which is conflict-free for ds_read_b128 / ds_write_b128.  (ds_add_f32 was tried first: one
// LDS atomic per element and vector made the kernel 4x SLOWER than the register-accumulator version it replaced.)
__device__ __forceinline__ void slab_add(float* slab, int plane_stride, int slot, const float (&v)[8]) {
    f32x4* p0 = reinterpret_cast<f32x4*>(slab) + slot;
    f32x4* p1 = reinterpret_cast<f32x4*>(slab + plane_stride) + slot;
    f32x4 a = *p0, b = *p1;
    a += f32x4{v[0], v[1], v[2], v[3]};
    b += f32x4{v[4], v[5], v[6], v[7]};
    *p0 = a;
    *p1 = b;
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// PRO = 1: the row is produced on the fly as silu(w) * w2 (QuantLlamaMLP: act_fn(gate_proj(x)) * up_proj(x),
// models/int_llama_layer.py:44-45) -- the product never goes through memory and reaches the quantiser in fp32.
__device__ __forceinline__ float silu_f(float g, float* sg_out) {
    const float sg = sigmoidf_(g);
    *sg_out = sg;
    return g * sg;
}

template <typename TIN, typename TOUT, bool LET, int CH, int PRO = 0>
__global__ void __launch_bounds__(512) rowq_fwd_kernel(FQ p, int nw, int chn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int rpb = wpb / nw;
    const int rslot = wid / nw, wsub = wid - rslot * nw;
    const int K = (int)p.cols;
    const int nchunks = K >> 3;
    float* cm_s = smem;
    float* sh_s = smem + K;
    float* red = smem + (LET ? 2 * K : 0);
    const float Q = (float)((1 << p.nbits) - 1);
    if constexpr (LET) {
        for (int i = threadIdx.x * 4; i < K; i += blockDim.x * 4) {
            *reinterpret_cast<f32x4*>(cm_s + i) = p.col_mul ? *reinterpret_cast<const f32x4*>(p.col_mul + i) : f32x4{1.f, 1.f, 1.f, 1.f};
            *reinterpret_cast<f32x4*>(sh_s + i) = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
    }
    bool valid[CH];
    int cc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = (j * nw + wsub) * 64 + lane;
        valid[j] = c < nchunks;
        cc[j] = (valid[j] ? c : nchunks - 1) * 8;     // surplus lanes redo the last chunk (benign duplicate stores)
    }
    const TIN* wbase = reinterpret_cast<const TIN*>(p.w);
    const int64_t LW = (PRO == 1 && p.ldw) ? p.ldw : (int64_t)K;      // row stride of w / w2 (gate | up column blocks of one buffer)
    TOUT* ybase = reinterpret_cast<TOUT*>(p.y);
    const bool lwc = p.up != nullptr;
    int par = 0;
    // integer side channel (FQ::codes / csum): the row's code sum needs one more row-wide reduction; with several waves per row
    // it rides in the free slot of the NEXT row's min / max exchange (no extra barrier per row), flushed once after the loop
    // (LET rows use the exchange's fourth slot for w @ shift: their code sum takes an exchange of its own -- the row-group
    // kernels serve the LET weights of the benchmarked widths, this path only the narrow and the very wide ones)
    const bool want_codes = p.codes != nullptr;
    const bool eight = p.nbits == 8;
    int64_t pend_r = -1;
    float pend_v = 0.f;
    for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < p.rows; r0 += (int64_t)gridDim.x * rpb) {
        int64_t r = r0 + rslot;
        if (r >= p.rows) r = p.rows - 1;              // surplus waves redo the last row: same values stored again
        const TIN* wrow = wbase + r * LW;
        Raw8<TIN> raw[CH];
        Raw8<TIN> raw2[PRO ? CH : 1];
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (j < chn) {
                raw[j].load(wrow + cc[j]);
                if constexpr (PRO == 1) raw2[j].load(reinterpret_cast<const TIN*>(p.w2) + r * LW + cc[j]);
            }
        const float upl = lwc ? p.up[r] : 0.f, lowl = lwc ? p.low[r] : 0.f;
        const float rd = (LET && p.row_div) ? p.row_div[r] : 1.f;
        const float rm = (LET && p.row_mul) ? p.row_mul[r] : 1.f;
        const float inv_rd = 1.f / rd;
        float x[CH][8];
        float hi = -INFINITY, lo = INFINITY, dot = 0.f;
        uint64_t nanm = 0;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (j < chn) {
                raw[j].unpack(x[j]);
                if constexpr (PRO == 1) {
                    float u[8], sg;
                    raw2[j].unpack(u);
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[j][i] = silu_f(x[j][i], &sg) * u[i];
                }
                if constexpr (LET) {
                    float cm[8], sh[8];
                    Vec8<float>::load(cm_s + cc[j], cm);
                    Vec8<float>::load(sh_s + cc[j], sh);
                    const float lv = valid[j] ? 1.f : 0.f;     // surplus lanes must not add to w @ shift
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        dot = fmaf(x[j][i] * lv, sh[i], dot);
                        float v = x[j][i] * cm[i];
                        if (p.row_div) v = div_nr(v, rd, inv_rd);
                        if (p.row_mul) v = v * rm;
                        x[j][i] = v;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    hi = vmax(hi, x[j][i]);
                    lo = vmin(lo, x[j][i]);
                    nanm |= __builtin_amdgcn_fcmpf(x[j][i], x[j][i], 8);       // FCMP_UNO
                }
            }
        }
        float v4[4] = {wave_max(hi), wave_min(lo), nanm != 0 ? 1.f : 0.f, LET ? wave_sum(dot) : pend_v};
        if (nw > 1) {
            const int op[4] = {1, 2, 1, 0};
            row_exchange(red, par, wid, rslot, nw, lane, v4, op);
            if (!LET && want_codes && pend_r >= 0 && wsub == 0) p.csum[pend_r] = v4[3];     // the previous row's code sum
        }
        hi = v4[0]; lo = v4[1];
        const float bad = v4[2];
        dot = v4[3];
        if (bad != 0.f) { hi = NAN; lo = NAN; }
        float inv_s = 0.f;
        const QP q = make_qp(hi, lo, lwc, upl, lowl, p.nbits, p.symmetric, p.inv_q, &inv_s);
        const bool regular = q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f && bad == 0.f;     // wave-uniform
        TOUT* yrow = ybase + r * K;
        float csl = 0.f;                              // this lane's share of the row's code sum
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (j < chn) {
                float yv[8];
                if (p.nbits >= 16) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) yv[i] = x[j][i];
                } else if (regular && want_codes) {
                    float rq[8], qv[8];
                    uint64_t susp = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float tq = x[j][i] * inv_s;
                        rq[i] = rintf(tq);
                        susp |= __builtin_amdgcn_fcmpf(fabsf(tq - rq[i]), fmaf(-4e-7f, fabsf(tq), 0.5f), 2);   // FCMP_OGT
                    }
                    if (susp != 0) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float tq = x[j][i] * inv_s;
                            if (fabsf(tq - rq[i]) > fmaf(-4e-7f, fabsf(tq), 0.5f)) rq[i] = rintf(x[j][i] / q.s);
                        }
                    }
                    float cs8 = eight ? -1024.f : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        qv[i] = __builtin_amdgcn_fmed3f(rq[i] + q.z, 0.f, Q);
                        yv[i] = (qv[i] - q.z) * q.s;
                        cs8 += qv[i];
                    }
                    oq_store_codes8(p.codes + r * K + cc[j], qv, eight);
                    csl += valid[j] ? cs8 : 0.f;      // surplus lanes redo the row's last chunk
                } else if (regular) {
                    float rq[8];
                    uint64_t susp = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float tq = x[j][i] * inv_s;
                        rq[i] = rintf(tq);
                        susp |= __builtin_amdgcn_fcmpf(fabsf(tq - rq[i]), fmaf(-4e-7f, fabsf(tq), 0.5f), 2);   // FCMP_OGT
                    }
                    if (susp != 0) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float tq = x[j][i] * inv_s;
                            if (fabsf(tq - rq[i]) > fmaf(-4e-7f, fabsf(tq), 0.5f)) rq[i] = rintf(x[j][i] / q.s);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) yv[i] = (__builtin_amdgcn_fmed3f(rq[i] + q.z, 0.f, Q) - q.z) * q.s;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float v = rne_ste(x[j][i] / q.s) + q.z;
                        v = (v != v) ? v : fminf(fmaxf(v, 0.f), Q);
                        yv[i] = (v - q.z) * q.s;
                    }
                }
                Vec8<TOUT>::store(yrow + cc[j], yv);
                if (want_codes && !(regular && p.nbits < 16)) {
                    const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    oq_store_codes8(p.codes + r * K + cc[j], zero8, false);
                }
            }
        }
        if (wsub == 0) {      // every lane stores the same value (no divergent branch around the stores)
            p.scale[r] = q.s;
            p.zp[r] = q.z;
            p.xmin[r] = lo;
            p.xmax[r] = hi;
            if (LET && p.wshift) p.wshift[r] = dot;
        }
        if (want_codes) {
            // a row whose scale is 0 / NaN is all-NaN in the reference (quirk Q1): a NaN code sum makes the integer GEMM say so
            const float cs = (regular && p.nbits < 16) ? wave_sum(csl) : NAN;
            if (nw == 1) {
                p.csum[r] = cs;
            } else if (LET) {
                float c4[4] = {0.f, 0.f, 0.f, cs};
                const int opc[4] = {1, 2, 1, 0};
                row_exchange(red, par, wid, rslot, nw, lane, c4, opc);
                if (wsub == 0) p.csum[r] = c4[3];
            } else {
                pend_r = r;
                pend_v = cs;
            }
        }
    }
    if (!LET && want_codes && nw > 1) {       // flush the last row's code sum (every wave of the workgroup walked the same number of rows)
        float v4[4] = {0.f, 0.f, 0.f, pend_v};
        const int op[4] = {1, 2, 1, 0};
        row_exchange(red, par, wid, rslot, nw, lane, v4, op);
        if (pend_r >= 0 && wsub == 0) p.csum[pend_r] = v4[3];
    }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
template <typename TIN, typename TG, bool LET, int CH, int PRO = 0>
__global__ void __launch_bounds__(512) rowq_bwd_kernel(FQ p, int nw, int chn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int rpb = wpb / nw;
    const int rslot = wid / nw, wsub = wid - rslot * nw;
    const int K = (int)p.cols;
    const int nchunks = K >> 3;
    const float Q = (float)((1 << p.nbits) - 1);
    const bool need_cm = LET && p.g_col_mul, need_sh = LET && p.g_shift;
    const bool need_row = LET && (p.g_row_div || p.g_row_mul);
    const bool need_col = need_cm || need_sh;
    const bool ident = p.nbits >= 16;
    const bool need_tie = (p.gx || need_cm) && !ident;
    const int acc_stride = chn * 512;                       // floats per accumulator slab (one wave, one vector)
    float* cm_s = smem;
    float* acc = smem + (LET ? K : 0);
    float* red = acc + (need_col ? wpb * 2 * acc_stride : 0);
    if constexpr (LET) {
        for (int i = threadIdx.x * 4; i < K; i += blockDim.x * 4)
            *reinterpret_cast<f32x4*>(cm_s + i) = p.col_mul ? *reinterpret_cast<const f32x4*>(p.col_mul + i) : f32x4{1.f, 1.f, 1.f, 1.f};
        if (need_col)
            for (int i = threadIdx.x; i < wpb * 2 * acc_stride; i += blockDim.x) acc[i] = 0.f;
        __syncthreads();
    }
    float* acc_cm = acc + wid * 2 * acc_stride;             // this wave's own slabs
    float* acc_sh = acc_cm + acc_stride;
    bool valid[CH];
    int cc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = (j * nw + wsub) * 64 + lane;
        valid[j] = c < nchunks;
        cc[j] = (valid[j] ? c : nchunks - 1) * 8;
    }
    const TIN* wbase = reinterpret_cast<const TIN*>(p.w);
    const TG* gbase = reinterpret_cast<const TG*>(p.g);
    TG* gxbase = reinterpret_cast<TG*>(p.gx);
    TG* gx2base = reinterpret_cast<TG*>(p.gx2);
    const TIN* w2base = reinterpret_cast<const TIN*>(p.w2);
    const int64_t LW = (PRO == 1 && p.ldw) ? p.ldw : (int64_t)K;      // row stride of w / w2 and of gx / gx2
    // PRO = 1: x = silu(w) * w2; dL/dx (gin) becomes dL/dw = gin * w2 * silu'(w) -> gx and dL/dw2 = gin * silu(w) -> gx2
    auto store_grads = [&](int64_t off, const float (&gin)[8], const float (&wv)[8], const float (&uv)[8]) {
        if constexpr (PRO == 1) {
            float og[8], ou[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float sg;
                const float sl = silu_f(wv[i], &sg);
                ou[i] = gin[i] * sl;
                og[i] = gin[i] * uv[i] * (sg * (1.f + wv[i] * (1.f - sg)));
            }
            Vec8<TG>::store(gxbase + off, og);
            Vec8<TG>::store(gx2base + off, ou);
        } else {
            Vec8<TG>::store(gxbase + off, gin);
        }
    };
    const bool lwc = p.up != nullptr;
    int par = 0;
    for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < p.rows; r0 += (int64_t)gridDim.x * rpb) {
        const bool livew = r0 + rslot < p.rows;             // wave-uniform; dead waves only keep the barriers company
        const int64_t r = livew ? r0 + rslot : p.rows - 1;
        const TIN* wrow = wbase + r * LW;
        const TG* grow = gbase + r * K;
        float gs = 0.f, arm = 0.f;
        int whi = 0, wlo = 0;
        uint32_t tieflag = 0;                               // bit j: chunk j of this wave holds an element == max or min
        float hi = 0.f, lo = 0.f, inv_s = 0.f, rd = 1.f, rm = 1.f, gws = 0.f, inv_rd = 1.f, rmrd = 1.f;
        QP q;
        q.s = 1.f; q.z = 0.f; q.su = q.sl = 1.f; q.hi = q.lo = 0.f;
        if (livew) {
            Raw8<TIN> rw[CH];
            Raw8<TIN> ru[PRO ? CH : 1];
            Raw8<TG> rg[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (j < chn) {
                    rw[j].load(wrow + cc[j]);
                    rg[j].load(grow + cc[j]);
                    if constexpr (PRO == 1) ru[j].load(w2base + r * LW + cc[j]);
                }
            hi = p.xmax[r];
            lo = p.xmin[r];
            rd = (LET && p.row_div) ? p.row_div[r] : 1.f;
            rm = (LET && p.row_mul) ? p.row_mul[r] : 1.f;
            gws = (LET && p.g_wshift) ? p.g_wshift[r] : 0.f;
            inv_rd = 1.f / rd;
            rmrd = rm * inv_rd;
            q = make_qp(hi, lo, lwc, lwc ? p.up[r] : 0.f, lwc ? p.low[r] : 0.f, p.nbits, p.symmetric, p.inv_q, &inv_s);
            const float z = q.z;
            // scale == 0 (quirk Q1): round_ste turns x / 0 = +-inf into NaN; a NaN zero-point inside round(t) + z gives the
            // same all-NaN row without a per-element select ((r - t) + t == r for every finite t)
            const float zr = q.s == 0.f ? NAN : z;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j < chn) {
                    float w[8], G[8], gin[8], uu[8], xs[8], xv[8];
                    rw[j].unpack(w);
                    rg[j].unpack(G);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { xs[i] = w[i]; uu[i] = 0.f; }
                    if constexpr (PRO == 1) {
                        ru[j].unpack(uu);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { float sg; xs[i] = silu_f(w[i], &sg) * uu[i]; }
                    }
                    const float lv = valid[j] ? 1.f : 0.f;
                    float cm[8], ccm[8], csh[8];
                    if constexpr (LET) Vec8<float>::load(cm_s + cc[j], cm);
                    float gsc = 0.f, armc = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float v = xs[i], a2 = xs[i];
                        if constexpr (LET) {
                            v = v * cm[i];
                            a2 = v;
                            if (p.row_div) v = div_nr(v, rd, inv_rd);
                            if (p.row_mul) v = v * rm;
                        }
                        xv[i] = v;
                        const float tq = v * inv_s;
                        const float u = rintf(tq) + zr;
                        const float qv = __builtin_amdgcn_fmed3f(u, 0.f, Q);
                        const bool in = ident || qv == u;                       // inside [0, Q] (false for NaN)
                        gin[i] = in ? G[i] : 0.f;
                        gsc = fmaf(G[i], qv - z, gsc);
                        gsc = fmaf(-gin[i], tq, gsc);
                        if constexpr (LET) {
                            const float gi = gin[i] * lv;                       // surplus lanes contribute nothing
                            if (need_row) armc = fmaf(gi, a2 * inv_rd, armc);   // b = (w*cm)/rd, x = b*rm
                            ccm[i] = (gi * rmrd) * w[i];
                            csh[i] = (gws * lv) * w[i];
                        }
                    }
                    gs += valid[j] ? gsc : 0.f;                                 // surplus lanes re-read the last chunk
                    arm += valid[j] ? armc : 0.f;
                    float cmx = xv[0], cmn = xv[0];
#pragma unroll
                    for (int i = 1; i < 7; i += 2) {
                        cmx = vmax3(cmx, xv[i], xv[i + 1]);
                        cmn = vmin3(cmn, xv[i], xv[i + 1]);
                    }
                    cmx = vmax(cmx, xv[7]);
                    cmn = vmin(cmn, xv[7]);
                    const uint64_t hit = __builtin_amdgcn_ballot_w64(valid[j] && (cmx == hi || cmn == lo));
                    if (hit != 0) {          // this wave holds an amax / amin element of the row in this chunk (rare)
                        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid[j]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            whi += __builtin_popcountll(__builtin_amdgcn_fcmpf(xv[i], hi, 1) & vmask);      // FCMP_OEQ
                            wlo += __builtin_popcountll(__builtin_amdgcn_fcmpf(xv[i], lo, 1) & vmask);
                        }
                        tieflag |= 1u << j;
                    }
                    if constexpr (LET) {
                        if (need_cm) slab_add(acc_cm, acc_stride / 2, j * 64 + lane, ccm);
                        if (need_sh) slab_add(acc_sh, acc_stride / 2, j * 64 + lane, csh);
                    }
                    if (p.gx) store_grads(r * LW + cc[j], gin, w, uu);         // tie chunks are rewritten below
                }
            }
        }
        float v4[4] = {wave_sum(gs), (float)whi, (float)wlo, LET ? wave_sum(arm) : 0.f};
        if (nw > 1) {
            const int op[4] = {0, 0, 0, 0};
            row_exchange(red, par, wid, rslot, nw, lane, v4, op);
        }
        if (!livew) continue;
        gs = ident ? 0.f : v4[0];
        const float nhi = v4[1], nlo = v4[2];
        arm = v4[3];
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
        const float g_hs = gs * ds_dhs, g_ls = gs * ds_dls;
        if (wsub == 0) {
            if (p.g_up) p.g_up[r] = g_hs * q.hi * q.su * (1.f - q.su);
            if (p.g_low) p.g_low[r] = g_ls * q.lo * q.sl * (1.f - q.sl);
            if constexpr (LET) {
                if (need_row) {
                    const float tot = arm + (g_hs * q.su * q.hi + g_ls * q.sl * q.lo) / rm;
                    if (p.g_row_mul) p.g_row_mul[r] = tot;
                    if (p.g_row_div) p.g_row_div[r] = -rmrd * tot;
                }
            }
        }
        if (need_tie && tieflag != 0) {
            // straight-through terms of amax / amin: only the chunks holding an element equal to the row max / min
            const float tie_hi = g_hs * q.su / nhi, tie_lo = g_ls * q.sl / nlo;
            const float z = q.z;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j < chn && ((tieflag >> j) & 1u)) {
                    float w[8], G[8], gin[8], ccm[8], uu[8], xs[8];
                    Vec8<TIN>::load(wrow + cc[j], w);
                    Vec8<TG>::load(grow + cc[j], G);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { xs[i] = w[i]; uu[i] = 0.f; }
                    if constexpr (PRO == 1) {
                        Vec8<TIN>::load(w2base + r * LW + cc[j], uu);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { float sg; xs[i] = silu_f(w[i], &sg) * uu[i]; }
                    }
                    float cm[8];
                    if constexpr (LET) Vec8<float>::load(cm_s + cc[j], cm);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float v = xs[i];
                        if constexpr (LET) {
                            v = v * cm[i];
                            if (p.row_div) v = div_nr(v, rd, inv_rd);
                            if (p.row_mul) v = v * rm;
                        }
                        const float tq = v * inv_s;
                        const float u = rintf(tq) + (q.s == 0.f ? NAN : z);
                        const bool in = __builtin_amdgcn_fmed3f(u, 0.f, Q) == u;
                        float tt = 0.f;
                        if (v == hi) tt += tie_hi;
                        if (v == lo) tt += tie_lo;
                        gin[i] = (in ? G[i] : 0.f) + tt;
                        ccm[i] = valid[j] ? (tt * rmrd) * w[i] : 0.f;
                    }
                    if constexpr (LET) {
                        if (need_cm) slab_add(acc_cm, acc_stride / 2, j * 64 + lane, ccm);
                    }
                    if (p.gx) store_grads(r * LW + cc[j], gin, w, uu);
                }
            }
        }
    }
    if constexpr (LET) {
        if (need_col) {
            // sum the workgroup's slabs in wave order and write ONE partial row per workgroup (colreduce_kernel finishes)
            __syncthreads();
            float* wcm = p.ws + (int64_t)blockIdx.x * K;
            float* wsh = p.ws + ((int64_t)gridDim.x + blockIdx.x) * K;
            for (int e = threadIdx.x; e < K; e += blockDim.x) {
                const int c = e >> 3, i = e & 7;
                const int qd = c >> 6, ln = c & 63;
                const int ws_ = qd % nw, j = qd / nw;
                const int slot = (i >> 2) * (acc_stride / 2) + (j * 64 + ln) * 4 + (i & 3);    // [plane][chunk slot][4]
                float s0 = 0.f, s1 = 0.f;
                for (int rs = 0; rs < rpb; ++rs) {
                    const float* a = acc + (rs * nw + ws_) * 2 * acc_stride;
                    s0 += a[slot];
                    s1 += a[acc_stride + slot];
                }
                if (p.g_col_mul) wcm[e] = s0;
                if (p.g_shift) wsh[e] = s1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LET weights (col_mul / row_div / row_mul / shift + their gradients): row-GROUP kernels
// ---------------------------------------------------------------------------------------------------
// A workgroup of 4 waves walks groups of 4 rows.  Lane -> column ownership is fixed (wave w, lane l own the chunks
// (j*4 + w)*64 + l), so col_mul / shift sit in registers and the column gradients accumulate in registers across all
// rows of the workgroup (written once, as one partial row per workgroup, at the end).  What the segment kernels pay per
// ROW and per WAVE -- the sigmoids, scale / zero-point, the reductions' tail and two barriers -- is paid once per row:
// wave w finalises row w of the group, the others read the result from LDS; 2 (forward) / 3 (backward) barriers per
// group.  The forward walks groups of 2 rows when rows are <= 4096 elements (fewer registers, more workgroups); the
// backward walks groups of 4 with the next row -- and the next group's row constants -- prefetched.  Element loops carry no
// per-element selects: row_div / row_mul presence is a template parameter (a uniform `if` inside an unrolled loop is
// compiled to v_cndmask per element), amax / amin ties are found per 8-element chunk with v_max3 / v_min3.
constexpr int RG = 4;
#ifndef LETQ_FWD_WPE
#define LETQ_FWD_WPE 3      // min waves per SIMD the register allocator must leave room for
#endif
#ifndef LETQ_BWD_WPE
#define LETQ_BWD_WPE 3
#endif

struct RowQ {       // per-row constants handed from the finalising wave to everybody (one LDS broadcast read each)
    float s, z, inv_s, regular, hi, lo, rd, rm, inv_rd, rmrd, gws, live;
};

// MODE: bit 0 = row_div present, bit 1 = row_mul present (compile-time: no per-element selects for them)
// RGT rows per group: 4, or 2 for matrices of few row groups (half the registers per wave, twice the workgroups)
template <int RGT>
struct LetqFwdSmem {        // declared once per KERNEL (a body instantiated per row mode would otherwise get one copy each)
    float part[4][RGT][4];
    float qps[RGT][4];
    float csp[4][RGT];      // per-wave code sums of the group's rows (integer side channel)
};

template <typename TIN, typename TOUT, int CH, int MODE, int RGT>
__device__ __forceinline__ void letq_fwd_body(const FQ& p, const int bid, const int nb, LetqFwdSmem<RGT>& sm) {
    constexpr bool has_rd = (MODE & 1) != 0, has_rm = (MODE & 2) != 0;
    float (&part)[4][RGT][4] = sm.part;
    float (&qps)[RGT][4] = sm.qps;
    float (&csp)[4][RGT] = sm.csp;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = (int)p.cols;
    const int nchunks = K >> 3;
    const float Q = (float)((1 << p.nbits) - 1);
    // integer side channel (FQ::codes / csum): phase C leaves per-wave code sums in `csp`; wave w adds them up for row w of the
    // group in phase B of the workgroup's NEXT group (behind that group's first barrier: no extra barrier), once more after the loop
    const bool want_codes = p.codes != nullptr && p.nbits <= 8;
    const bool eight = p.nbits == 8;
    int64_t pend_r0 = -1;
    bool valid[CH];
    int cc[CH];
    float cm[CH][8], sh[CH][8];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = (j * 4 + wid) * 64 + lane;
        valid[j] = c < nchunks;
        cc[j] = (valid[j] ? c : nchunks - 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { cm[j][i] = 1.f; sh[j][i] = 0.f; }
        if (p.col_mul) Vec8<float>::load(p.col_mul + cc[j], cm[j]);
        if (p.shift && valid[j]) Vec8<float>::load(p.shift + cc[j], sh[j]);      // surplus lanes add nothing to w @ shift
    }
    const TIN* wbase = reinterpret_cast<const TIN*>(p.w);
    TOUT* ybase = reinterpret_cast<TOUT*>(p.y);
    const bool lwc = p.up != nullptr;
    const int64_t ngroups = (p.rows + RGT - 1) / RGT;
    for (int64_t g = bid; g < ngroups; g += nb) {
        const int64_t r0 = g * RGT;
        // ---- phase A: load + transform this wave's columns of the 4 rows, per-row partial min / max / NaN / w@shift ----
        Raw8<TIN> raw[RGT][CH];
        int64_t rows_[RGT];
#pragma unroll
        for (int rr = 0; rr < RGT; ++rr) {
            rows_[rr] = r0 + rr < p.rows ? r0 + rr : p.rows - 1;      // a short last group redoes the last row (same values)
#pragma unroll
            for (int j = 0; j < CH; ++j) raw[rr][j].load(wbase + rows_[rr] * K + cc[j]);
        }
        float x[RGT][CH][8];
#pragma unroll
        for (int rr = 0; rr < RGT; ++rr) {
            const float rd = has_rd ? p.row_div[rows_[rr]] : 1.f;
            const float rm = has_rm ? p.row_mul[rows_[rr]] : 1.f;
            const float inv_rd = 1.f / rd;
            float hi = -INFINITY, lo = INFINITY, dot = 0.f;
            uint64_t nanm = 0;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                raw[rr][j].unpack(x[rr][j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    dot = fmaf(x[rr][j][i], sh[j][i], dot);
                    float v = x[rr][j][i] * cm[j][i];
                    if (has_rd) v = div_nr(v, rd, inv_rd);
                    if (has_rm) v = v * rm;
                    x[rr][j][i] = v;
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    hi = vmax3(hi, x[rr][j][i], x[rr][j][i + 1]);
                    lo = vmin3(lo, x[rr][j][i], x[rr][j][i + 1]);
                    nanm |= __builtin_amdgcn_fcmpf(x[rr][j][i], x[rr][j][i + 1], 8);      // FCMP_UNO: either one is NaN
                }
            }
            hi = wave_max(hi);
            lo = wave_min(lo);
            dot = wave_sum(dot);
            if (lane == 0) *reinterpret_cast<f32x4*>(&part[wid][rr][0]) = f32x4{hi, lo, nanm != 0 ? 1.f : 0.f, dot};
        }
        __syncthreads();
        // ---- phase B: wave w finalises row w ------------------------------------------------------------------------
        if (wid < RGT) {
            const int rr = wid;
            if (want_codes && pend_r0 >= 0) {          // code sum of row w of the previous group
                const int64_t rp = pend_r0 + rr < p.rows ? pend_r0 + rr : p.rows - 1;
                p.csum[rp] = (csp[0][rr] + csp[1][rr]) + (csp[2][rr] + csp[3][rr]);
            }
            float hi = -INFINITY, lo = INFINITY, bad = 0.f, dot = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(&part[w2][rr][0]);
                hi = fmaxf(hi, q4[0]); lo = fminf(lo, q4[1]); bad = fmaxf(bad, q4[2]); dot += q4[3];
            }
            if (bad != 0.f) { hi = NAN; lo = NAN; }
            const int64_t r = r0 + rr < p.rows ? r0 + rr : p.rows - 1;
            float inv_s = 0.f;
            const QP q = make_qp(hi, lo, lwc, lwc ? p.up[r] : 0.f, lwc ? p.low[r] : 0.f, p.nbits, p.symmetric, p.inv_q, &inv_s);
            const bool regular = q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f && bad == 0.f;
            p.scale[r] = q.s;          // every lane stores the same value
            p.zp[r] = q.z;
            p.xmin[r] = lo;
            p.xmax[r] = hi;
            if (p.wshift) p.wshift[r] = dot;
            if (lane == 0) *reinterpret_cast<f32x4*>(&qps[rr][0]) = f32x4{q.s, q.z, inv_s, regular ? 1.f : 0.f};
        }
        __syncthreads();
        // ---- phase C: quantise and store ------------------------------------------------------------------------------
#pragma unroll
        for (int rr = 0; rr < RGT; ++rr) {
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(&qps[rr][0]);
            const float qs = q4[0], qz = q4[1], inv_s = q4[2];
            const bool regular = q4[3] != 0.f;
            TOUT* yrow = ybase + rows_[rr] * K;
            float csl = 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float yv[8];
                if (p.nbits >= 16) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) yv[i] = x[rr][j][i];
                } else if (regular) {
                    float rq[8];
                    uint64_t susp = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float tq = x[rr][j][i] * inv_s;
                        rq[i] = rintf(tq);
                        susp |= __builtin_amdgcn_fcmpf(fabsf(tq - rq[i]), fmaf(-4e-7f, fabsf(tq), 0.5f), 2);   // FCMP_OGT
                    }
                    if (susp != 0) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float tq = x[rr][j][i] * inv_s;
                            if (fabsf(tq - rq[i]) > fmaf(-4e-7f, fabsf(tq), 0.5f)) rq[i] = rintf(x[rr][j][i] / qs);
                        }
                    }
                    if (want_codes) {
                        float qv[8];
                        float cs8 = eight ? -1024.f : 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            qv[i] = __builtin_amdgcn_fmed3f(rq[i] + qz, 0.f, Q);
                            yv[i] = (qv[i] - qz) * qs;
                            cs8 += qv[i];
                        }
                        oq_store_codes8(p.codes + rows_[rr] * K + cc[j], qv, eight);
                        csl += valid[j] ? cs8 : 0.f;          // surplus lanes redo the row's last chunk
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) yv[i] = (__builtin_amdgcn_fmed3f(rq[i] + qz, 0.f, Q) - qz) * qs;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float v = rne_ste(x[rr][j][i] / qs) + qz;
                        v = (v != v) ? v : fminf(fmaxf(v, 0.f), Q);
                        yv[i] = (v - qz) * qs;
                    }
                    if (want_codes) {
                        const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        oq_store_codes8(p.codes + rows_[rr] * K + cc[j], zero8, false);
                    }
                }
                Vec8<TOUT>::store(yrow + cc[j], yv);
            }
            if (want_codes) {
                // a row whose scale is 0 / NaN is all-NaN in the reference (quirk Q1): a NaN code sum makes the integer GEMM say so
                const float cs = regular ? wave_sum(csl) : NAN;
                if (lane == 0) csp[wid][rr] = cs;
            }
        }
        pend_r0 = r0;
    }
    if (want_codes) {
        __syncthreads();
        if (wid < RGT && pend_r0 >= 0) {
            const int64_t rp = pend_r0 + wid < p.rows ? pend_r0 + wid : p.rows - 1;
            p.csum[rp] = (csp[0][wid] + csp[1][wid]) + (csp[2][wid] + csp[3][wid]);
        }
    }
}

template <typename TIN, typename TOUT, int CH, int MODE, int RGT>
__global__ void __launch_bounds__(256, CH <= 2 ? (RGT == 2 ? 4 : LETQ_FWD_WPE) : 2) letq_fwd_kernel(FQ p) {
    __shared__ __attribute__((aligned(16))) LetqFwdSmem<RGT> sm;
    letq_fwd_body<TIN, TOUT, CH, MODE, RGT>(p, (int)blockIdx.x, (int)gridDim.x, sm);
}

// Several weight matrices of one shape class (same row length, dtypes) in ONE launch: workgroup ranges [start[i],
// start[i+1]) belong to matrix i.  Four 67 MB problems launched one by one run at 3.9 TB/s each (ramp-up, tail and the
// launch gap are paid four times), the same kernel on a 180 MB problem at 5.2 TB/s.
constexpr int OQ_WQ_MAX = 4;
struct MultiFQ {
    FQ t[OQ_WQ_MAX];
    int start[OQ_WQ_MAX + 1];
    int mode[OQ_WQ_MAX];
    int n;
};

template <typename TIN, typename TOUT, int CH, int RGT>
__global__ void __launch_bounds__(256, CH <= 2 ? (RGT == 2 ? 4 : LETQ_FWD_WPE) : 2) letq_fwd_multi_kernel(MultiFQ m) {
    __shared__ __attribute__((aligned(16))) LetqFwdSmem<RGT> sm;
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.start[i + 1]) ++i;
    const int bid = (int)blockIdx.x - m.start[i], nb = m.start[i + 1] - m.start[i];
    switch (m.mode[i]) {
        case 0: letq_fwd_body<TIN, TOUT, CH, 0, RGT>(m.t[i], bid, nb, sm); break;
        case 1: letq_fwd_body<TIN, TOUT, CH, 1, RGT>(m.t[i], bid, nb, sm); break;
        case 2: letq_fwd_body<TIN, TOUT, CH, 2, RGT>(m.t[i], bid, nb, sm); break;
        default: letq_fwd_body<TIN, TOUT, CH, 3, RGT>(m.t[i], bid, nb, sm); break;
    }
}

// MODE: bit 0 = row_div present, bit 1 = row_mul present (compile-time: the element loop carries no selects for them)
template <typename TIN, int CH>
struct LetqBwdSmem {        // declared once per KERNEL (see LetqFwdSmem)
    float part[4][RG][4];
    float qps2[2][RG][12];    // by group parity: phase A of the next group overlaps phase D
    float ties[RG][4];
    // 16-bit weights: a chunk that holds an amax / amin element is parked in LDS by the wave that met it (phase B) and read
    // back by the same wave in phase D, instead of being fetched from global memory a second time
    Raw8<TIN> stash[sizeof(TIN) == 2 ? 4 * RG * CH * 64 : 1];
};

template <typename TIN, typename TG, int CH, int MODE>
__device__ __forceinline__ void letq_bwd_body(const FQ& p, const int bid, const int nb, LetqBwdSmem<TIN, CH>& sm) {
    float (&part)[4][RG][4] = sm.part;
    float (&qps2)[2][RG][12] = sm.qps2;
    float (&ties)[RG][4] = sm.ties;
    constexpr bool STASH = sizeof(TIN) == 2;
    Raw8<TIN>* stash = sm.stash;
    constexpr bool has_rd = (MODE & 1) != 0, has_rm = (MODE & 2) != 0;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = (int)p.cols;
    const int nchunks = K >> 3;
    const float Q = (float)((1 << p.nbits) - 1);
    const bool need_cm = p.g_col_mul != nullptr, need_sh = p.g_shift != nullptr;
    const bool need_row = p.g_row_div || p.g_row_mul;
    const bool ident = p.nbits >= 16;
    bool valid[CH];
    int cc[CH];
    float cm[CH][8], acc_cm[CH][8], acc_sh[CH][8];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = (j * 4 + wid) * 64 + lane;
        valid[j] = c < nchunks;
        cc[j] = (valid[j] ? c : nchunks - 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { cm[j][i] = 1.f; acc_cm[j][i] = 0.f; acc_sh[j][i] = 0.f; }
        if (p.col_mul) Vec8<float>::load(p.col_mul + cc[j], cm[j]);
    }
    const TIN* wbase = reinterpret_cast<const TIN*>(p.w);
    const TG* gbase = reinterpret_cast<const TG*>(p.g);
    const bool lwc = p.up != nullptr;
    const int64_t ngroups = (p.rows + RG - 1) / RG;
    int par = 0;
    // the rows a workgroup walks form one stream: while row t is processed row t+1 is in flight, across group boundaries
    // too (the next group's first row loads under phases C, D and A); two rows of raw vectors live at a time.  (Deeper
    // register rings were measured: copies between ring slots wait for the load they move, and four statically
    // assigned slots cost 30 VGPRs and ran 15 % slower.)
    auto stream_row = [&](int64_t g, int k) -> int64_t {
        const int64_t r = k < RG ? g * RG + k : (g + nb) * RG + (k - RG);
        return r < p.rows ? r : p.rows - 1;        // past the end: a re-read of the last row nobody uses
    };
    // the same for the per-row constants phase A needs: those of the workgroup's next group are fetched one group ahead
    struct RowIn { float hi, lo, rd, rm, gws, up, low; };
    auto fetch_row_in = [&](int64_t g) -> RowIn {
        int64_t r = g * RG + wid;
        r = r < p.rows ? r : p.rows - 1;
        RowIn o;
        o.hi = p.xmax[r]; o.lo = p.xmin[r];
        o.rd = has_rd ? p.row_div[r] : 1.f;
        o.rm = has_rm ? p.row_mul[r] : 1.f;
        o.gws = p.g_wshift ? p.g_wshift[r] : 0.f;
        o.up = lwc ? p.up[r] : 0.f;
        o.low = lwc ? p.low[r] : 0.f;
        return o;
    };
    Raw8<TIN> cw[CH];
    Raw8<TG> cg[CH];
    RowIn nxt{};
    if ((int64_t)bid < ngroups) {
        nxt = fetch_row_in(bid);
        const int64_t ra = stream_row(bid, 0);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            cw[j].load(wbase + ra * K + cc[j]);
            cg[j].load(gbase + ra * K + cc[j]);
        }
    }
    for (int64_t g = bid; g < ngroups; g += nb, par ^= 1) {
        const int64_t r0 = g * RG;
        float (*qps)[12] = qps2[par];
        auto row_of = [&](int rr) { return r0 + rr < p.rows ? r0 + rr : p.rows - 1; };
        // ---- phase A: wave w prepares the constants of row w ---------------------------------------------------------
        QP q;                       // kept by the finalising wave for phase C
        float my_rm = 1.f, my_rmrd = 1.f;
        {
            const RowIn in = nxt;
            nxt = fetch_row_in(g + nb);      // lands under phases B .. D
            const float hi = in.hi, lo = in.lo, rd = in.rd, rm = in.rm, gws = in.gws;
            const float inv_rd = 1.f / rd;
            float inv_s = 0.f;
            q = make_qp(hi, lo, lwc, in.up, in.low, p.nbits, p.symmetric, p.inv_q, &inv_s);
            // scale == 0 (quirk Q1): the reference's round_ste turns x / 0 = +-inf into NaN; a NaN zero-point inside
            // round(t) + z gives the same all-NaN row without a per-element select ((r - t) + t == r for every finite t)
            const float zr = q.s == 0.f ? NAN : q.z;
            my_rm = rm;
            my_rmrd = rm * inv_rd;
            if (lane == 0) {
                float* d = &qps[wid][0];
                *reinterpret_cast<f32x4*>(d) = f32x4{q.s, q.z, inv_s, zr};
                *reinterpret_cast<f32x4*>(d + 4) = f32x4{hi, lo, rd, rm};
                *reinterpret_cast<f32x4*>(d + 8) = f32x4{inv_rd, rm * inv_rd, gws, 0.f};
            }
        }
        __syncthreads();
        // ---- phase B: element pass over this wave's columns of the 4 rows --------------------------------------------
        uint32_t tieflag = 0;               // bit rr*CH + j
#pragma unroll 1                            // a real loop: unrolled, the four rows' constants and vectors cost 260+ VGPRs
        for (int rr = 0; rr < RG; ++rr) {
            Raw8<TIN> nw_[CH];
            Raw8<TG> ng_[CH];
            {
                const int64_t rn = stream_row(g, rr + 1);
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    nw_[j].load(wbase + rn * K + cc[j]);
                    ng_[j].load(gbase + rn * K + cc[j]);
                }
            }
            float gs = 0.f, arm = 0.f;
            int whi = 0, wlo = 0;
            if (r0 + rr < p.rows) {          // rows past the end of a short last group add nothing (uniform branch)
                const f32x4 qa = *reinterpret_cast<const f32x4*>(&qps[rr][0]);
                const f32x4 qb = *reinterpret_cast<const f32x4*>(&qps[rr][4]);
                const f32x4 qc = *reinterpret_cast<const f32x4*>(&qps[rr][8]);
                const float z = qa[1], inv_s = qa[2], zr = qa[3], hi = qb[0], lo = qb[1], rd = qb[2], rm = qb[3];
                const float inv_rd = qc[0], rmrd = qc[1], gws = qc[2];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    float w[8], G[8], x[8];
                    cw[j].unpack(w);
                    cg[j].unpack(G);
                    float gsc = 0.f, armc = 0.f;
                    float cmx = -INFINITY, cmn = INFINITY;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a2 = w[i] * cm[j][i];
                        float v = a2;
                        if (has_rd) v = div_nr(v, rd, inv_rd);
                        if (has_rm) v = v * rm;
                        x[i] = v;
                        cmx = __builtin_fmaxf(cmx, v);
                        cmn = __builtin_fminf(cmn, v);
                        const float tq = v * inv_s;
                        const float u = rintf(tq) + zr;
                        const float qv = __builtin_amdgcn_fmed3f(u, 0.f, Q);
                        const bool in = ident || qv == u;
                        const float gi = in ? G[i] : 0.f;
                        gsc = fmaf(G[i], qv - z, gsc);
                        gsc = fmaf(-gi, tq, gsc);
                        // accumulated unconditionally (a uniform `if` around an fma becomes a per-element select);
                        // what is not asked for is not stored
                        if (MODE != 0) armc = fmaf(gi, MODE == 1 ? v : (MODE == 2 ? a2 : a2 * inv_rd), armc);
                        acc_cm[j][i] = fmaf(MODE == 0 ? gi : gi * rmrd, w[i], acc_cm[j][i]);
                        acc_sh[j][i] = fmaf(gws, w[i], acc_sh[j][i]);
                    }
                    // surplus lanes re-read the row's last chunk: their column accumulators are never stored, their row
                    // sums and ties are dropped here
                    gs += valid[j] ? gsc : 0.f;
                    arm += valid[j] ? armc : 0.f;
                    const uint64_t hit = __builtin_amdgcn_ballot_w64(valid[j] && (cmx == hi || cmn == lo));
                    if (hit != 0) {          // this wave holds an amax / amin element of the row in this chunk (rare)
                        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid[j]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            whi += __builtin_popcountll(__builtin_amdgcn_fcmpf(x[i], hi, 1) & vmask);      // FCMP_OEQ
                            wlo += __builtin_popcountll(__builtin_amdgcn_fcmpf(x[i], lo, 1) & vmask);
                        }
                        tieflag |= 1u << (rr * CH + j);
                        if (STASH) stash[((wid * RG + rr) * CH + j) * 64 + lane] = cw[j];
                    }
                }
                if (ident) gs = 0.f;
            }
            gs = wave_sum(gs);
            arm = wave_sum(arm);
            if (lane == 0) *reinterpret_cast<f32x4*>(&part[wid][rr][0]) = f32x4{gs, (float)whi, (float)wlo, arm};
#pragma unroll
            for (int j = 0; j < CH; ++j) { cw[j] = nw_[j]; cg[j] = ng_[j]; }
        }
        __syncthreads();
        // ---- phase C: wave w finalises row w --------------------------------------------------------------------------
        {
            float gs = 0.f, nhi = 0.f, nlo = 0.f, arm = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(&part[w2][wid][0]);
                gs += q4[0]; nhi += q4[1]; nlo += q4[2]; arm += q4[3];
            }
            float ds_dhs, ds_dls;
            if (p.symmetric) {
                const float lvl = (float)((1 << (p.nbits - 1)) - 1);
                const float hs = q.su * q.hi, ls = q.sl * q.lo;
                const float a = fabsf(hs), b = fabsf(ls);
                const float raw = fmaxf(a, b) / lvl;
                const float pass = (raw >= 1e-5f && raw <= 1e4f) ? 1.f : 0.f;
                const float sh_ = hs > 0.f ? 1.f : (hs < 0.f ? -1.f : 0.f);
                const float sg = ls > 0.f ? 1.f : (ls < 0.f ? -1.f : 0.f);
                const float wa = a > b ? 1.f : (a == b ? 0.5f : 0.f);
                ds_dhs = pass * wa * sh_ / lvl;
                ds_dls = pass * (1.f - wa) * sg / lvl;
            } else {
                ds_dhs = p.inv_q;
                ds_dls = -p.inv_q;
            }
            const float g_hs = gs * ds_dhs, g_ls = gs * ds_dls;
            if (r0 + wid < p.rows) {
                const int64_t r = r0 + wid;
                if (p.g_up) p.g_up[r] = g_hs * q.hi * q.su * (1.f - q.su);
                if (p.g_low) p.g_low[r] = g_ls * q.lo * q.sl * (1.f - q.sl);
                if (need_row) {
                    const float tot = arm + (g_hs * q.su * q.hi + g_ls * q.sl * q.lo) / my_rm;
                    if (p.g_row_mul) p.g_row_mul[r] = tot;
                    if (p.g_row_div) p.g_row_div[r] = -my_rmrd * tot;
                }
            }
            if (lane == 0) *reinterpret_cast<f32x4*>(&ties[wid][0]) =
                f32x4{ident ? 0.f : g_hs * q.su / nhi, ident ? 0.f : g_ls * q.sl / nlo, 0.f, 0.f};
        }
        __syncthreads();
        // ---- phase D: straight-through terms of amax / amin for the chunks that hold such an element ------------------
        if (need_cm && tieflag != 0) {
#pragma unroll 1
            for (int rr = 0; rr < RG; ++rr) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    if ((tieflag >> (rr * CH + j)) & 1u) {
                        const f32x4 qb = *reinterpret_cast<const f32x4*>(&qps[rr][4]);
                        const f32x4 qc = *reinterpret_cast<const f32x4*>(&qps[rr][8]);
                        const f32x4 tt4 = *reinterpret_cast<const f32x4*>(&ties[rr][0]);
                        const float hi = qb[0], lo = qb[1], rd = qb[2], rm = qb[3], inv_rd = qc[0], rmrd = qc[1];
                        const float lv = valid[j] ? 1.f : 0.f;
                        float w[8];
                        if (STASH) stash[((wid * RG + rr) * CH + j) * 64 + lane].unpack(w);
                        else Vec8<TIN>::load(wbase + row_of(rr) * K + cc[j], w);      // 32-bit weights: reloaded (cache hit)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            float v = w[i] * cm[j][i];
                            if (has_rd) v = div_nr(v, rd, inv_rd);
                            if (has_rm) v = v * rm;
                            float tt = 0.f;
                            if (v == hi) tt += tt4[0];
                            if (v == lo) tt += tt4[1];
                            acc_cm[j][i] = fmaf((tt * lv) * rmrd, w[i], acc_cm[j][i]);
                        }
                    }
                }
            }
        }
        // (the next group writes the OTHER qps buffer; `part` and `ties` are rewritten only behind the next group's barriers)
    }
    if (need_cm || need_sh) {
        float* wcm = p.ws + (int64_t)bid * K;
        float* wsh = p.ws + ((int64_t)nb + bid) * K;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (valid[j]) {
                if (need_cm) Vec8<float>::store(wcm + cc[j], acc_cm[j]);
                if (need_sh) Vec8<float>::store(wsh + cc[j], acc_sh[j]);
            }
        }
    }
}

template <typename TIN, typename TG, int CH, int MODE>
__global__ void __launch_bounds__(256, CH <= 2 ? LETQ_BWD_WPE : 2) letq_bwd_kernel(FQ p) {
    __shared__ __attribute__((aligned(16))) LetqBwdSmem<TIN, CH> sm;
    letq_bwd_body<TIN, TG, CH, MODE>(p, (int)blockIdx.x, (int)gridDim.x, sm);
}

template <typename TIN, typename TG, int CH>
__global__ void __launch_bounds__(256, CH <= 2 ? LETQ_BWD_WPE : 2) letq_bwd_multi_kernel(MultiFQ m) {
    __shared__ __attribute__((aligned(16))) LetqBwdSmem<TIN, CH> sm;
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.start[i + 1]) ++i;
    const int bid = (int)blockIdx.x - m.start[i], nb = m.start[i + 1] - m.start[i];
    switch (m.mode[i]) {
        case 0: letq_bwd_body<TIN, TG, CH, 0>(m.t[i], bid, nb, sm); break;
        case 1: letq_bwd_body<TIN, TG, CH, 1>(m.t[i], bid, nb, sm); break;
        case 2: letq_bwd_body<TIN, TG, CH, 2>(m.t[i], bid, nb, sm); break;
        default: letq_bwd_body<TIN, TG, CH, 3>(m.t[i], bid, nb, sm); break;
    }
}

// dynamic LDS above 64 KiB has to be granted per kernel; remembered so that the attribute call happens once per size
int set_smem(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return OQ_OK;
    static const void* seen_k[64];
    static size_t seen_b[64];
    static int n_seen = 0;
    for (int i = 0; i < n_seen; ++i)
        if (seen_k[i] == kernel && seen_b[i] >= bytes) return OQ_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return OQ_E_LAUNCH;
    if (n_seen < 64) { seen_k[n_seen] = kernel; seen_b[n_seen] = bytes; ++n_seen; }
    return OQ_OK;
}

constexpr size_t LDS_BUDGET = 156 * 1024;

size_t bwd_smem(const FQ& p, const RowGeo& g, bool let) {
    const bool need_col = let && (p.g_col_mul || p.g_shift);
    return sizeof(float) * ((let ? p.cols : 0) + (need_col ? (size_t)g.wpb * 2 * g.chn * 512 : 0) + 64);
}

bool bwd_geo(const FQ& p, bool let, RowGeo* g) {
    // column accumulators live in LDS: spread the row over more waves until the slabs fit
    if (!row_geo(p.cols, (int)env_i("OQ_ROWQ_BWD_NW", 0), g)) return false;
    while (bwd_smem(p, *g, let) > LDS_BUDGET) {
        if (g->nw >= 8 || !row_geo(p.cols, g->nw * 2, g)) return false;
    }
    return true;
}

}  // namespace

#define RQ_FWD(TIN, TOUT)                                                                                                       \
    do {                                                                                                                        \
        if (let) {                                                                                                              \
            if (g.chn <= 4) { rc = set_smem((const void*)rowq_fwd_kernel<TIN, TOUT, true, 4>, smem); hipLaunchKernelGGL((rowq_fwd_kernel<TIN, TOUT, true, 4>), grid, blk, smem, st, p, g.nw, g.chn); }    \
            else { rc = set_smem((const void*)rowq_fwd_kernel<TIN, TOUT, true, 8>, smem); hipLaunchKernelGGL((rowq_fwd_kernel<TIN, TOUT, true, 8>), grid, blk, smem, st, p, g.nw, g.chn); }              \
        } else {                                                                                                                \
            if (g.chn <= 4) hipLaunchKernelGGL((rowq_fwd_kernel<TIN, TOUT, false, 4>), grid, blk, smem, st, p, g.nw, g.chn);    \
            else hipLaunchKernelGGL((rowq_fwd_kernel<TIN, TOUT, false, 8>), grid, blk, smem, st, p, g.nw, g.chn);               \
        }                                                                                                                       \
    } while (0)

// row-group LET kernels: rows of 2048 .. 6144 elements (CH = 2 or 3 chunks per lane with 4 waves per row)
static int letq_ch(int64_t cols) {
    if (cols % 8 != 0 || env_i("OQ_LETQ", 1) == 0) return 0;
    const int64_t chunks = cols / 8;
    if (chunks > 256 && chunks <= 512) return 2;
    if (chunks > 512 && chunks <= 768) return 3;
    return 0;
}

#define LQ_FWD_M(TIN, TOUT, M)                                                                               \
    do {                                                                                                     \
        if (ch == 2 && rgt == 2) hipLaunchKernelGGL((letq_fwd_kernel<TIN, TOUT, 2, M, 2>), grid, dim3(256), 0, st, p); \
        else if (ch == 2) hipLaunchKernelGGL((letq_fwd_kernel<TIN, TOUT, 2, M, 4>), grid, dim3(256), 0, st, p);        \
        else hipLaunchKernelGGL((letq_fwd_kernel<TIN, TOUT, 3, M, 4>), grid, dim3(256), 0, st, p);           \
    } while (0)
#define LQ_FWD(TIN, TOUT)                                                                                    \
    do {                                                                                                     \
        switch ((p.row_div ? 1 : 0) | (p.row_mul ? 2 : 0)) {                                                 \
            case 0: LQ_FWD_M(TIN, TOUT, 0); break;                                                           \
            case 1: LQ_FWD_M(TIN, TOUT, 1); break;                                                           \
            case 2: LQ_FWD_M(TIN, TOUT, 2); break;                                                           \
            default: LQ_FWD_M(TIN, TOUT, 3); break;                                                          \
        }                                                                                                    \
    } while (0)
#define LQ_BWD_M(TIN, TG, M)                                                                                 \
    do {                                                                                                     \
        if (ch == 2) hipLaunchKernelGGL((letq_bwd_kernel<TIN, TG, 2, M>), grid, dim3(256), 0, st, p);        \
        else hipLaunchKernelGGL((letq_bwd_kernel<TIN, TG, 3, M>), grid, dim3(256), 0, st, p);                \
    } while (0)
#define LQ_BWD(TIN, TG)                                                                                      \
    do {                                                                                                     \
        switch ((p.row_div ? 1 : 0) | (p.row_mul ? 2 : 0)) {                                                 \
            case 0: LQ_BWD_M(TIN, TG, 0); break;                                                             \
            case 1: LQ_BWD_M(TIN, TG, 1); break;                                                             \
            case 2: LQ_BWD_M(TIN, TG, 2); break;                                                             \
            default: LQ_BWD_M(TIN, TG, 3); break;                                                            \
        }                                                                                                    \
    } while (0)

static int letq_fwd(const FQ& p, int ch, int w_dtype, int y_dtype, void* stream) {
    // rows per group: 2 for rows <= 4096 elements (109 VGPRs, 4 workgroups per CU, twice the groups: [4096,4096] 19.3 -> 17.2 us,
    // [11008,4096] 37.7 -> 34.6 us = 5.2 TB/s; one row per group is no better: 17.0 / 38.1 us), 4 for longer rows
    const int rgt = ch == 2 && env_i("OQ_LETQ_FWD_RG", 2) == 2 ? 2 : 4;
    const int64_t ngroups = (p.rows + rgt - 1) / rgt;
    const int64_t cap = (int64_t)n_cus() * env_i("OQ_LETQ_FWD_WGS", 8);
    const dim3 grid((unsigned)(ngroups < cap ? ngroups : cap));
    hipStream_t st = (hipStream_t)stream;
    switch (w_dtype * 3 + y_dtype) {
        case OQ_F32 * 3 + OQ_F32: LQ_FWD(float, float); break;
        case OQ_F32 * 3 + OQ_BF16: LQ_FWD(float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: LQ_FWD(f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: LQ_FWD(f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: LQ_FWD(bf16_t, bf16_t); break;
        default: return 1;
    }
    OQ_CHECK_LAUNCH("oq_fakequant_fwd(letq)");
    return OQ_OK;
}

static bool letq_fwd_eligible(const FQ& p) {
    return p.seg == p.cols && env_i("OQ_ROWQ", 1) != 0 && (p.col_mul || p.row_div || p.row_mul || p.shift) && letq_ch(p.cols) != 0;
}

// n weight matrices in one launch (see MultiFQ).  Returns 0 when launched, 1 when the problems are not one shape class of
// the row-group LET kernels (the caller then launches them one by one), or a negative OQ_E_* code.
int oq_letq_fwd_multi(const FQ* ps, int n, int w_dtype, int y_dtype, void* stream) {
    if (n < 2 || n > OQ_WQ_MAX || env_i("OQ_WQ_MULTI", 1) == 0) return 1;
    const int ch = letq_fwd_eligible(ps[0]) ? letq_ch(ps[0].cols) : 0;
    if (ch != 2 && ch != 3) return 1;
    for (int i = 0; i < n; ++i)
        if (!letq_fwd_eligible(ps[i]) || ps[i].cols != ps[0].cols || ps[i].nbits != ps[0].nbits) return 1;
    const int rgt = (ch == 2 && env_i("OQ_LETQ_FWD_RG", 2) == 2) ? 2 : 4;      // rows of 4097..6144 elements: groups of 4 (as the single launch)
    MultiFQ m{};
    m.n = n;
    const int64_t cap = (int64_t)n_cus() * env_i("OQ_LETQ_FWD_WGS", 8);
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        m.t[i] = ps[i];
        m.mode[i] = (ps[i].row_div ? 1 : 0) | (ps[i].row_mul ? 2 : 0);
        const int64_t ng = (ps[i].rows + rgt - 1) / rgt;
        m.start[i] = (int)tot;
        tot += ng < cap ? ng : cap;
    }
    m.start[n] = (int)tot;
    const dim3 grid((unsigned)tot);
    hipStream_t st = (hipStream_t)stream;
#define LQ_FWD_MULTI(TIN, TOUT)                                                                              \
    do {                                                                                                     \
        if (ch == 3) hipLaunchKernelGGL((letq_fwd_multi_kernel<TIN, TOUT, 3, 4>), grid, dim3(256), 0, st, m);   \
        else if (rgt == 2) hipLaunchKernelGGL((letq_fwd_multi_kernel<TIN, TOUT, 2, 2>), grid, dim3(256), 0, st, m); \
        else hipLaunchKernelGGL((letq_fwd_multi_kernel<TIN, TOUT, 2, 4>), grid, dim3(256), 0, st, m);        \
    } while (0)
    switch (w_dtype * 3 + y_dtype) {
        case OQ_F32 * 3 + OQ_F32: LQ_FWD_MULTI(float, float); break;
        case OQ_F32 * 3 + OQ_BF16: LQ_FWD_MULTI(float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: LQ_FWD_MULTI(f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: LQ_FWD_MULTI(f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: LQ_FWD_MULTI(bf16_t, bf16_t); break;
        default: return 1;
    }
    OQ_CHECK_LAUNCH("oq_fakequant_fwd_multi(letq)");
    return OQ_OK;
}

static bool letq_bwd_eligible(const FQ& p) {
    const bool let = p.col_mul || p.row_div || p.row_mul || p.g_col_mul || p.g_shift || p.g_row_div || p.g_row_mul;
    return p.seg == p.cols && env_i("OQ_ROWQ", 1) != 0 && let && !p.gx && letq_ch(p.cols) != 0;
}

// Backward counterpart: ps[i].ws must already point at matrix i's workspace (2 * parts[i] * cols floats, checked by the
// caller); parts[i] receives the number of partial rows matrix i's workgroups wrote (0: no column gradients asked for).
int oq_letq_bwd_multi(FQ* ps, int n, int w_dtype, int g_dtype, const int64_t* workspace_floats, int64_t* parts, void* stream) {
    if (n < 2 || n > OQ_WQ_MAX || env_i("OQ_WQ_MULTI", 1) == 0) return 1;
    const int ch = letq_bwd_eligible(ps[0]) ? letq_ch(ps[0].cols) : 0;
    if (ch != 2 && ch != 3) return 1;
    for (int i = 0; i < n; ++i)
        if (!letq_bwd_eligible(ps[i]) || ps[i].cols != ps[0].cols || ps[i].nbits != ps[0].nbits) return 1;
    MultiFQ m{};
    m.n = n;
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t nblk = oq_letq_bwd_blocks(ps[i].rows);
        parts[i] = 0;
        if (ps[i].g_col_mul || ps[i].g_shift) {
            OQ_CHECK_ARG(ps[i].ws && workspace_floats[i] >= 2 * nblk * ps[i].cols,
                         "oq_fakequant_bwd_multi: workspace of %lld floats needed (oq_fakequant_bwd_workspace)",
                         (long long)(2 * nblk * ps[i].cols));
            parts[i] = nblk;
        } else {
            ps[i].ws = nullptr;
        }
        m.t[i] = ps[i];
        m.mode[i] = (ps[i].row_div ? 1 : 0) | (ps[i].row_mul ? 2 : 0);
        m.start[i] = (int)tot;
        tot += nblk;
    }
    m.start[n] = (int)tot;
    const dim3 grid((unsigned)tot);
    hipStream_t st = (hipStream_t)stream;
#define LQ_BWD_MULTI(TIN, TG)                                                                              \
    do {                                                                                                   \
        if (ch == 3) hipLaunchKernelGGL((letq_bwd_multi_kernel<TIN, TG, 3>), grid, dim3(256), 0, st, m);     \
        else hipLaunchKernelGGL((letq_bwd_multi_kernel<TIN, TG, 2>), grid, dim3(256), 0, st, m);            \
    } while (0)
    switch (w_dtype * 3 + g_dtype) {
        case OQ_F32 * 3 + OQ_F32: LQ_BWD_MULTI(float, float); break;
        case OQ_F32 * 3 + OQ_BF16: LQ_BWD_MULTI(float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: LQ_BWD_MULTI(f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: LQ_BWD_MULTI(f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: LQ_BWD_MULTI(bf16_t, bf16_t); break;
        default: return 1;
    }
    OQ_CHECK_LAUNCH("oq_fakequant_bwd_multi(letq)");
    return OQ_OK;
}

// The kernels that write the integer side channel (FQ::codes / csum): whole-row segments on grids of at most 8 bits; LET
// weights on the row-group kernels, everything else on the wave-per-row kernel without LET.
extern "C" int64_t oq_fakequant_codes_supported(int64_t cols, int64_t seg, int nbits, int let) {
    if (seg != cols || nbits < 2 || nbits > 8 || env_i("OQ_ROWQ", 1) == 0) return 0;
    if (let && letq_ch(cols) != 0) return 1;
    if (let && env_i("OQ_ROWQ_FWD_LET", 1) == 0) return 0;
    RowGeo g;
    if (!row_geo(cols, (int)env_i("OQ_ROWQ_FWD_NW", cols >= 8192 ? 8 : 0), &g)) return 0;
    return sizeof(float) * ((let ? 2 * cols : 0) + 64) <= LDS_BUDGET ? 1 : 0;
}

int oq_rowq_fwd(const FQ& p, int w_dtype, int y_dtype, void* stream) {
    if (p.seg != p.cols || env_i("OQ_ROWQ", 1) == 0) return 1;
    const bool let = p.col_mul || p.row_div || p.row_mul || p.shift;
    if (let) {
        const int ch = letq_ch(p.cols);
        if (ch) return letq_fwd(p, ch, w_dtype, y_dtype, stream);
    }
    RowGeo g;
    // rows of >= 8192 elements: 8 waves per row (3 .. 4 chunks per lane, the CH = 4 instantiation at 4 waves per SIMD) instead of
    // 4 waves with 6 .. 8 chunks (CH = 8: 171 VGPRs, 2 waves per SIMD): [4096, 11008] weights with the integer side channel
    // 52.0 -> 45.0 us, [2048, 11008] activations 32.4 -> 26.0 us (28.7 -> 22.2 us without codes); shorter rows lose (17 vs 12 us)
    if (!row_geo(p.cols, (int)env_i("OQ_ROWQ_FWD_NW", p.cols >= 8192 ? 8 : 0), &g)) return 1;
    if (let && env_i("OQ_ROWQ_FWD_LET", 1) == 0) return 1;
    const size_t smem = sizeof(float) * ((let ? 2 * p.cols : 0) + 64);
    if (smem > LDS_BUDGET) return 1;
    const int rpb = g.wpb / g.nw;
    const int64_t need = (p.rows + rpb - 1) / rpb;
    const int64_t cap = (int64_t)n_cus() * env_i("OQ_ROWQ_FWD_WGS", 8);
    const dim3 grid((unsigned)(need < cap ? need : cap)), blk((unsigned)(g.wpb * 64));
    hipStream_t st = (hipStream_t)stream;
    int rc = OQ_OK;
    switch (w_dtype * 3 + y_dtype) {
        case OQ_F32 * 3 + OQ_F32: RQ_FWD(float, float); break;
        case OQ_F32 * 3 + OQ_BF16: RQ_FWD(float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: RQ_FWD(f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: RQ_FWD(f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: RQ_FWD(bf16_t, bf16_t); break;
        default: return 1;
    }
    if (rc) {
        oq_set_error("oq_fakequant_fwd(rowq): cannot reserve %zu bytes of LDS", smem);
        return rc;
    }
    OQ_CHECK_LAUNCH("oq_fakequant_fwd(rowq)");
    return OQ_OK;
}

#define RQ_BWD(TIN, TG)                                                                                                         \
    do {                                                                                                                        \
        if (let) {                                                                                                              \
            if (g.chn <= 4) { rc = set_smem((const void*)rowq_bwd_kernel<TIN, TG, true, 4>, smem); hipLaunchKernelGGL((rowq_bwd_kernel<TIN, TG, true, 4>), grid, blk, smem, st, p, g.nw, g.chn); }        \
            else { rc = set_smem((const void*)rowq_bwd_kernel<TIN, TG, true, 8>, smem); hipLaunchKernelGGL((rowq_bwd_kernel<TIN, TG, true, 8>), grid, blk, smem, st, p, g.nw, g.chn); }                  \
        } else {                                                                                                                \
            if (g.chn <= 4) hipLaunchKernelGGL((rowq_bwd_kernel<TIN, TG, false, 4>), grid, blk, smem, st, p, g.nw, g.chn);      \
            else hipLaunchKernelGGL((rowq_bwd_kernel<TIN, TG, false, 8>), grid, blk, smem, st, p, g.nw, g.chn);                 \
        }                                                                                                                       \
    } while (0)

static int64_t bwd_grid(const FQ& p, const RowGeo& g, bool let) {
    const int rpb = g.wpb / g.nw;
    const int64_t need = (p.rows + rpb - 1) / rpb;
    const size_t smem = bwd_smem(p, g, let);
    int64_t per_cu = (int64_t)((160 * 1024) / (smem > 1024 ? smem : 1024));
    const int64_t mx = env_i("OQ_ROWQ_BWD_WGS", 8);
    if (per_cu > mx) per_cu = mx;
    if (per_cu < 1) per_cu = 1;
    const int64_t cap = (int64_t)n_cus() * per_cu;
    return need < cap ? need : cap;
}

int64_t oq_letq_bwd_blocks(int64_t rows) {
    // workgroups of the row-group LET backward = partial rows colreduce_kernel sums afterwards.  2 workgroups per CU, 3 when
    // there are enough row groups to keep them balanced (11008 x 4096: 59.8 -> 56.0 us; 4096 x 4096: 32.6 vs 33.0 us)
    const int64_t ngroups = (rows + RG - 1) / RG;
    const int64_t cap = env_i("OQ_LETQ_BWD_BLOCKS", ngroups >= 2048 ? 768 : 512);
    return ngroups < cap ? ngroups : cap;
}

int64_t oq_rowq_bwd_blocks(int64_t rows, int64_t cols) {
    // upper bound on the workgroups of a LET backward with column gradients (the workspace has 2 rows per workgroup)
    FQ p{};
    p.rows = rows; p.cols = cols; p.seg = cols;
    float dummy = 0.f;
    p.g_col_mul = &dummy; p.g_shift = &dummy;
    RowGeo g;
    if (env_i("OQ_ROWQ", 1) == 0 || !bwd_geo(p, true, &g)) return 0;
    return bwd_grid(p, g, true);
}

int oq_rowq_bwd(const FQ& pin, int w_dtype, int g_dtype, float* workspace, int64_t workspace_floats, int64_t* partial_rows,
                void* stream) {
    if (pin.seg != pin.cols || env_i("OQ_ROWQ", 1) == 0) return 1;
    FQ p = pin;
    const bool let = p.col_mul || p.row_div || p.row_mul || p.g_col_mul || p.g_shift || p.g_row_div || p.g_row_mul;
    *partial_rows = 0;
    if (let && !p.gx) {
        const int ch = letq_ch(p.cols);
        if (ch) {
            const int64_t nblk = oq_letq_bwd_blocks(p.rows);
            if (p.g_col_mul || p.g_shift) {
                OQ_CHECK_ARG(workspace && workspace_floats >= 2 * nblk * p.cols,
                             "oq_fakequant_bwd: workspace of %lld floats needed (oq_fakequant_bwd_workspace)", (long long)(2 * nblk * p.cols));
                p.ws = workspace;
                *partial_rows = nblk;
            }
            const dim3 grid((unsigned)nblk);
            hipStream_t st = (hipStream_t)stream;
            switch (w_dtype * 3 + g_dtype) {
                case OQ_F32 * 3 + OQ_F32: LQ_BWD(float, float); break;
                case OQ_F32 * 3 + OQ_BF16: LQ_BWD(float, bf16_t); break;
                case OQ_F16 * 3 + OQ_F32: LQ_BWD(f16_t, float); break;
                case OQ_F16 * 3 + OQ_BF16: LQ_BWD(f16_t, bf16_t); break;
                case OQ_BF16 * 3 + OQ_BF16: LQ_BWD(bf16_t, bf16_t); break;
                default: *partial_rows = 0; return 1;
            }
            OQ_CHECK_LAUNCH("oq_fakequant_bwd(letq)");
            return OQ_OK;
        }
    }
    // LET backward (column gradients): the segment kernel's register accumulators win (46 us vs 61 us with the LDS slabs
    // on a 4096 x 4096 weight, same box); the slab variant stays available for A/B
    if (let && env_i("OQ_ROWQ_BWD_LET", 0) == 0) return 1;
    RowGeo g;
    if (!bwd_geo(p, let, &g)) return 1;
    const size_t smem = bwd_smem(p, g, let);
    const int64_t nblk = bwd_grid(p, g, let);
    if (p.g_col_mul || p.g_shift) {
        OQ_CHECK_ARG(workspace && workspace_floats >= 2 * nblk * p.cols,
                     "oq_fakequant_bwd: workspace of %lld floats needed (oq_fakequant_bwd_workspace)", (long long)(2 * nblk * p.cols));
        p.ws = workspace;
        *partial_rows = nblk;
    }
    const dim3 grid((unsigned)nblk), blk((unsigned)(g.wpb * 64));
    hipStream_t st = (hipStream_t)stream;
    int rc = OQ_OK;
    switch (w_dtype * 3 + g_dtype) {
        case OQ_F32 * 3 + OQ_F32: RQ_BWD(float, float); break;
        case OQ_F32 * 3 + OQ_BF16: RQ_BWD(float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: RQ_BWD(f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: RQ_BWD(f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: RQ_BWD(bf16_t, bf16_t); break;
        default: return 1;
    }
    if (rc) {
        oq_set_error("oq_fakequant_bwd(rowq): cannot reserve %zu bytes of LDS", smem);
        return rc;
    }
    OQ_CHECK_LAUNCH("oq_fakequant_bwd(rowq)");
    return OQ_OK;
}

// ---- fused silu(gate) * up -> per-token fake quant (the down_proj input of QuantLlamaMLP) -----------------------------
// TP: dtype of gate / up (the projection's output); TY: dtype of y (forward) resp. of the gradients g, ggate, gup (backward)
template <typename TP, typename TY>
static int silu_q_launch(bool fwd, FQ& p, void* stream) {
    RowGeo g;
    // two inputs (and two gradient outputs) per element: 8 waves per row for the long rows of the MLP (3-4 chunks per lane,
    // the CH = 4 instantiation) instead of 4 waves with 6 chunks (CH = 8, 179 VGPRs in the backward): in-step
    // 36.4 -> 28.9 us forward, 55.4 -> 47.0 us backward at [2048, 11008]
    const int64_t dflt = p.cols >= 8192 ? 8 : env_i(fwd ? "OQ_ROWQ_FWD_NW" : "OQ_ROWQ_BWD_NW", 0);
    const int nw_env = (int)env_i(fwd ? "OQ_SILUQ_FWD_NW" : "OQ_SILUQ_BWD_NW", dflt);
    if (!row_geo(p.cols, nw_env, &g)) {
        oq_set_error("oq_silu_mul_quant: rows of %lld elements are not supported (512 .. 32768, multiple of 8)", (long long)p.cols);
        return OQ_E_UNSUPPORTED;
    }
    const int rpb = g.wpb / g.nw;
    const int64_t need = (p.rows + rpb - 1) / rpb;
    const int64_t cap = (int64_t)n_cus() * 8;
    const dim3 grid((unsigned)(need < cap ? need : cap)), blk((unsigned)(g.wpb * 64));
    const size_t smem = sizeof(float) * 64;
    hipStream_t st = (hipStream_t)stream;
    if (fwd) {
        if (g.chn <= 4) hipLaunchKernelGGL((rowq_fwd_kernel<TP, TY, false, 4, 1>), grid, blk, smem, st, p, g.nw, g.chn);
        else hipLaunchKernelGGL((rowq_fwd_kernel<TP, TY, false, 8, 1>), grid, blk, smem, st, p, g.nw, g.chn);
    } else {
        if (g.chn <= 4) hipLaunchKernelGGL((rowq_bwd_kernel<TP, TY, false, 4, 1>), grid, blk, smem, st, p, g.nw, g.chn);
        else hipLaunchKernelGGL((rowq_bwd_kernel<TP, TY, false, 8, 1>), grid, blk, smem, st, p, g.nw, g.chn);
    }
    OQ_CHECK_LAUNCH("oq_silu_mul_quant");
    return OQ_OK;
}

extern "C" int oq_silu_mul_quant_fwd(const void* gate, const void* up, int dtype, int64_t rows, int64_t cols, int64_t ld, int nbits,
                                     void* y, int y_dtype, float* scale, float* zp, float* xmin, float* xmax, void* codes,
                                     float* csum, void* stream) {
    OQ_CHECK_ARG(gate && up && y && scale && zp && xmin && xmax, "oq_silu_mul_quant_fwd: null pointer");
    OQ_CHECK_ARG((codes == nullptr) == (csum == nullptr) && (!codes || nbits <= 8),
                 "oq_silu_mul_quant_fwd: codes and csum go together and need a grid of at most 8 bits");
    OQ_CHECK_ARG(ld == 0 || (ld >= cols && ld % 8 == 0), "oq_silu_mul_quant_fwd: ld %lld (0 or a multiple of 8 >= cols)", (long long)ld);
    OQ_CHECK_ARG(rows > 0 && nbits >= 2 && nbits < 16, "oq_silu_mul_quant_fwd: rows %lld, bitwidth %d", (long long)rows, nbits);
    OQ_CHECK_ARG(oq_aligned16(gate) && oq_aligned16(up) && oq_aligned16(y), "oq_silu_mul_quant_fwd: 16-byte alignment");
    FQ p{};
    p.w = gate; p.w2 = up; p.rows = rows; p.cols = cols; p.seg = cols; p.nbits = nbits; p.ldw = ld;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.y = y; p.scale = scale; p.zp = zp; p.xmin = xmin; p.xmax = xmax;
    p.codes = (int8_t*)codes; p.csum = csum;
    if (dtype == OQ_BF16 && y_dtype == OQ_BF16) return silu_q_launch<bf16_t, bf16_t>(true, p, stream);
    if (dtype == OQ_F32 && y_dtype == OQ_F32) return silu_q_launch<float, float>(true, p, stream);
    if (dtype == OQ_F32 && y_dtype == OQ_BF16) return silu_q_launch<float, bf16_t>(true, p, stream);   // fp32 pre-activations
    oq_set_error("oq_silu_mul_quant_fwd: dtypes in %d / out %d unsupported", dtype, y_dtype);
    return OQ_E_UNSUPPORTED;
}

extern "C" int oq_silu_mul_quant_bwd(const void* gate, const void* up, const void* g, int dtype, int g_dtype, int64_t rows,
                                     int64_t cols, int64_t ld, int nbits, const float* xmin, const float* xmax, void* ggate,
                                     void* gup, void* stream) {
    OQ_CHECK_ARG(gate && up && g && ggate && gup && xmin && xmax, "oq_silu_mul_quant_bwd: null pointer");
    OQ_CHECK_ARG(ld == 0 || (ld >= cols && ld % 8 == 0), "oq_silu_mul_quant_bwd: ld %lld (0 or a multiple of 8 >= cols)", (long long)ld);
    OQ_CHECK_ARG(rows > 0 && nbits >= 2 && nbits < 16, "oq_silu_mul_quant_bwd: rows %lld, bitwidth %d", (long long)rows, nbits);
    OQ_CHECK_ARG(oq_aligned16(gate) && oq_aligned16(up) && oq_aligned16(g) && oq_aligned16(ggate) && oq_aligned16(gup),
                 "oq_silu_mul_quant_bwd: 16-byte alignment");
    FQ p{};
    p.w = gate; p.w2 = up; p.g = g; p.rows = rows; p.cols = cols; p.seg = cols; p.nbits = nbits; p.ldw = ld;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.xmin = const_cast<float*>(xmin); p.xmax = const_cast<float*>(xmax); p.gx = ggate; p.gx2 = gup;
    if (dtype == OQ_BF16 && g_dtype == OQ_BF16) return silu_q_launch<bf16_t, bf16_t>(false, p, stream);
    if (dtype == OQ_F32 && g_dtype == OQ_F32) return silu_q_launch<float, float>(false, p, stream);
    if (dtype == OQ_F32 && g_dtype == OQ_BF16) return silu_q_launch<float, bf16_t>(false, p, stream);
    oq_set_error("oq_silu_mul_quant_bwd: dtypes %d / gradients %d unsupported", dtype, g_dtype);
    return OQ_E_UNSUPPORTED;
}
