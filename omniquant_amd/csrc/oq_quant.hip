// UniformAffineQuantizer on gfx950: dynamic min/max + LWC + fake-quant, fused with the LET weight
// re-parameterisation.  Forward and closed-form backward.
//
// Replaces (reference, /root/reference): quantize/quantizer.py:15-19,84-105,122-147 and the weight side of
// models/transformation.py:24-69.  Every matrix element is read ONCE (one workgroup owns a whole row and keeps it in
// registers: 1-8 chunks of 8 elements per lane; the next row is prefetched as raw vectors), reduced with DPP wave
// reductions (+ one LDS exchange across waves for whole-row segments), quantised and written once.  Column-wise LET
// gradients are accumulated in registers across the rows a workgroup walks (lane -> column mapping is fixed), written
// as per-workgroup partial rows and finished by colreduce_kernel in a fixed order (no atomics anywhere).
// By bytes these kernels are HBM-bound; measured, the segment kernels of this file are bound by instruction issue and
// per-row synchronisation: whole-row segments are served by oq_rowq.hip, these remain for grouped
// weights and as the OQ_ROWQ=0 reference.
#include <stdlib.h>
#include <type_traits>
#include "oq_common.h"
#include "oq_quant_dev.h"

namespace {

constexpr int64_t OQ_FQ_BWD_MAX_BLOCKS = 512;   // workgroups when column partials are written
constexpr int64_t OQ_FQ_MAX_BLOCKS = 1024;      // otherwise: >= 2-4 rows per workgroup so the prefetch pipelines

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// A workgroup walks rows r = blockIdx.x, +gridDim.x, ...; the NEXT row's 16-byte chunks are already in flight (raw
// registers) while the current row is reduced and quantised, so HBM requests never drain between rows.  Everything
// that depends only on the column (col_mul, shift, segment index) is loaded once per workgroup.
// No divergent branch surrounds a global load or store inside the row loop (FULL: every lane owns a chunk; otherwise
// the surplus lanes redo the row's LAST chunk -- same inputs, same outputs, benign duplicate stores -- and are masked
// out of the sums): the compiler can then count outstanding memory operations and waits only for the prefetched row
// (vmcnt(N)), not for the previous row's stores to be acknowledged (vmcnt(0), a full memory round trip per row).
template <typename TIN, typename TOUT, bool LET, int CH, bool FULL>
__global__ void __launch_bounds__(512) fq_fwd_kernel(FQ p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 8 + 8];
    const int t = threadIdx.x, BT = blockDim.x;
    const bool small = p.seg <= 512;
    const int lps = small ? (int)(p.seg >> 3) : 64;
    const int64_t nseg = p.cols / p.seg;
    const float Q = (float)((1 << p.nbits) - 1);
    {
        const int op0[4] = {1, 2, 1, 0};
        const int op1[1] = {0};
        red_init<4>(op0, red);
        if (threadIdx.x >= 32 && threadIdx.x < 40) red[threadIdx.x] = 0.f;      // region of the short-segment w@shift sum
        (void)op1;
        __syncthreads();
    }
    const TIN* wbase = reinterpret_cast<const TIN*>(p.w);
    TOUT* ybase = reinterpret_cast<TOUT*>(p.y);

    bool valid[CH];
    int cc[CH], segi[CH];
    constexpr int NL = LET ? CH : 1;
    float cmv[NL][8], shv[NL][8];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c0 = (j * BT + t) * 8;
        valid[j] = FULL || c0 < p.cols;
        // surplus lanes redo the last chunk (whole-row segments) or, group-wise, the last segment (short segments)
        cc[j] = valid[j] ? c0 : (small ? (int)(p.cols - p.seg) + (t & (lps - 1)) * 8 : (int)p.cols - 8);
        segi[j] = (int)((uint32_t)cc[j] / (uint32_t)p.seg);
        if constexpr (LET) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { cmv[j][i] = 1.f; shv[j][i] = 0.f; }
            if (p.col_mul) Vec8<float>::load(p.col_mul + cc[j], cmv[j]);
            if (p.shift) {
                Vec8<float>::load(p.shift + cc[j], shv[j]);
                if (!valid[j]) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) shv[j][i] = 0.f;      // surplus lanes must not add to w @ shift
                }
            }
        }
    }
    Raw8<TIN> nxt[CH];
    float nup[CH], nlow[CH], nrd = 1.f, nrm = 1.f;   // next row's LWC logits / row factors, prefetched with its data
    const bool lwc = p.up != nullptr;
    const float* upp = lwc ? p.up : p.scale;          // dummy (ignored) source when LWC is off: keeps the loads unconditional
    const float* lowp = lwc ? p.low : p.scale;
#pragma unroll
    for (int j = 0; j < CH; ++j) { nup[j] = 0.f; nlow[j] = 0.f; }
    auto prefetch = [&](int64_t r) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            nxt[j].load(wbase + r * p.cols + cc[j]);
            nup[j] = upp[r * nseg + segi[j]];
            nlow[j] = lowp[r * nseg + segi[j]];
        }
        if (LET && p.row_div) nrd = p.row_div[r];
        if (LET && p.row_mul) nrm = p.row_mul[r];
    };
    int64_t r = blockIdx.x;
    if (r < p.rows) prefetch(r);
    for (; r < p.rows; r += gridDim.x) {
        float x[CH][8], cup[CH], clow[CH];
        const float rd = nrd, rm = nrm;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            cup[j] = nup[j];
            clow[j] = nlow[j];
            nxt[j].unpack(x[j]);
        }
        const int64_t rn = r + gridDim.x;
        if (rn < p.rows) prefetch(rn);
        float dot = 0.f;
        const float inv_rd = 1.f / rd;
        if constexpr (LET) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (p.shift) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) dot += x[j][i] * shv[j][i];
                }
                if (p.col_mul) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[j][i] = x[j][i] * cmv[j][i];
                }
                if (p.row_div) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[j][i] = div_nr(x[j][i], rd, inv_rd);
                }
                if (p.row_mul) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[j][i] = x[j][i] * rm;
                }
            }
        }
        // ---- min / max (NaN-propagating like torch.amax/amin) ---------------------------------
        // NaN test: one v_cmp_u per element into a wave mask that is OR-ed on the scalar unit
        float hi[CH], lo[CH], bad[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            hi[j] = -INFINITY;
            lo[j] = INFINITY;
            uint64_t nanm = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                hi[j] = vmax(hi[j], x[j][i]);
                lo[j] = vmin(lo[j], x[j][i]);
                nanm |= __builtin_amdgcn_fcmpf(x[j][i], x[j][i], 8);       // FCMP_UNO
            }
            bad[j] = small ? (((nanm >> (threadIdx.x & 63)) & 1) ? 1.f : 0.f) : (nanm != 0 ? 1.f : 0.f);
        }
        bool dot_done = false;
        if (small) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                hi[j] = wave_max(hi[j], lps);
                lo[j] = wave_min(lo[j], lps);
                bad[j] = wave_max(bad[j], lps);
            }
        } else {
            // ONE exchange per row: max, min, NaN flag (already per wave) and the w @ shift partial sum
            float v[4] = {-INFINITY, INFINITY, 0.f, dot};
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                v[0] = fmaxf(v[0], hi[j]);
                v[1] = fminf(v[1], lo[j]);
                v[2] = fmaxf(v[2], bad[j]);
            }
            const int op[4] = {1, 2, 1, 0};
            block_reduce<4>(v, op, red, 0x4u);
#pragma unroll
            for (int j = 0; j < CH; ++j) { hi[j] = v[0]; lo[j] = v[1]; bad[j] = v[2]; }
            dot = v[3];
            dot_done = true;
        }
        // ---- quantise ---------------------------------------------------------------------------
        QP q;
        float inv_s = 0.f;
        q.s = 1.f; q.z = 0.f; q.su = q.sl = 1.f; q.hi = q.lo = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int64_t sidx = r * nseg + segi[j];
            float h = hi[j], l = lo[j];
            if (bad[j] != 0.f) { h = NAN; l = NAN; }
            // whole-row segments: scale / zero-point / 1/scale are the same for every chunk of the thread -> once
            if (small || j == 0) {
                q = make_qp(h, l, lwc, cup[j], clow[j], p.nbits, p.symmetric, p.inv_q, &inv_s);
            }
            float yv[8];
            if (p.nbits >= 16) {
                // identity grid: the reference's quantizer returns its input unchanged for n_bits >= 16
                // (quantize/quantizer.py:109-110); only the fused LET transform / w @ shift of this kernel remain
#pragma unroll
                for (int i = 0; i < 8; ++i) yv[i] = x[j][i];
            } else if (q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f && bad[j] == 0.f) {
                // regular segment (finite non-zero scale, no NaN): no special values can appear below
                // rne(x / s) as rne(x * (1/s)); the (rare) lanes whose product lies within ~2 ulp of a half-integer redo
                // the exact division -- tested for the whole chunk with ONE wave-uniform branch
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
                // degenerate segment (constant row -> scale 0, inf/NaN inputs): replay the reference's op sequence
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = rne_ste(x[j][i] / q.s) + q.z;
                    v = (v != v) ? v : fminf(fmaxf(v, 0.f), Q);
                    yv[i] = (v - q.z) * q.s;
                }
            }
            Vec8<TOUT>::store(ybase + r * p.cols + cc[j], yv);
            // per-segment statistics: every lane of the segment stores the same value (no divergent branch)
            if (small || j == 0) {
                p.scale[sidx] = q.s;
                p.zp[sidx] = q.z;
                p.xmin[sidx] = l;
                p.xmax[sidx] = h;
            }
        }
        if (LET && p.wshift) {
            if (!dot_done) {
                float v[1] = {dot};
                const int op[1] = {0};
                block_reduce<1>(v, op, red + 32);
                dot = v[0];
            }
            p.wshift[r] = dot;        // same value from every lane
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
// ONE pass over the row and ONE block reduction per row: the segment min/max come from the forward (p.xmin/p.xmax),
// the pass accumulates gs, the tie counts and the "inside the clip range" part of every LET gradient, and the two
// straight-through terms that flow through max/min (they touch only the elements EQUAL to the row max/min) are patched
// in afterwards by the few waves that hold such an element.  Row-factor gradients use the closed forms
//   g_row_mul = sum_in G*b + (g_hi*su*hi + g_lo*sl*lo) / rm          (every tie element has b = x / rm)
//   g_row_div = -(rm / rd) * g_row_mul
template <typename TIN, typename TG, bool LET, int CH, bool FULL>
__global__ void __launch_bounds__(512) fq_bwd_kernel(FQ p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 8 + 8];
    const int t = threadIdx.x, BT = blockDim.x;
    const bool small = p.seg <= 512;
    const int lps = small ? (int)(p.seg >> 3) : 64;
    const int64_t nseg = p.cols / p.seg;
    const float Q = (float)((1 << p.nbits) - 1);
    if (threadIdx.x < 40) red[threadIdx.x] = 0.f;       // every reduction of this kernel is a sum: identity 0
    __syncthreads();
    const TIN* wbase = reinterpret_cast<const TIN*>(p.w);
    const TG* gbase = reinterpret_cast<const TG*>(p.g);
    TG* gxbase = reinterpret_cast<TG*>(p.gx);
    const bool need_cm = LET && p.g_col_mul;
    const bool need_row = LET && (p.g_row_div || p.g_row_mul);
    const bool need_sh = LET && p.g_shift;
    const bool need_tie = p.gx || need_cm;
    const bool ident = p.nbits >= 16;      // identity grid (see fq_fwd_kernel): y = x', so dL/dx' = G and nothing else

    constexpr int NACC = LET ? CH : 1;
    float acc_cm[NACC][8], acc_sh[NACC][8], cmv[NACC][8];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc_cm[j][i] = 0.f; acc_sh[j][i] = 0.f; cmv[j][i] = 1.f; }
    bool valid[CH];
    int cc[CH], segi[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c0 = (j * BT + t) * 8;
        valid[j] = FULL || c0 < p.cols;
        // surplus lanes redo the last chunk (whole-row segments) or, group-wise, the last segment (short segments);
        // they store the same values again and are masked out of every sum (see fq_fwd_kernel)
        cc[j] = valid[j] ? c0 : (small ? (int)(p.cols - p.seg) + (t & (lps - 1)) * 8 : (int)p.cols - 8);
        segi[j] = (int)((uint32_t)cc[j] / (uint32_t)p.seg);
        if constexpr (LET) {
            if (p.col_mul) Vec8<float>::load(p.col_mul + cc[j], cmv[j]);
        }
    }
    Raw8<TIN> nw[CH];
    Raw8<TG> ng[CH];
    float nup[CH], nlow[CH], nhi_[CH], nlo_[CH], nrd = 1.f, nrm = 1.f, ngws = 0.f;
    const bool lwc = p.up != nullptr;
    // per-segment statistics of the NEXT row ride along with its data; loaded unconditionally into their own
    // registers (dummy pointers when LWC is off) so that no select has to wait for them
    const float* upp = lwc ? p.up : p.xmax;
    const float* lowp = lwc ? p.low : p.xmin;
#pragma unroll
    for (int j = 0; j < CH; ++j) { nup[j] = 0.f; nlow[j] = 0.f; nhi_[j] = 0.f; nlo_[j] = 0.f; }
    auto prefetch = [&](int64_t r) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            nw[j].load(wbase + r * p.cols + cc[j]);
            ng[j].load(gbase + r * p.cols + cc[j]);
            const int64_t sidx = r * nseg + segi[j];
            nhi_[j] = p.xmax[sidx];
            nlo_[j] = p.xmin[sidx];
            nup[j] = upp[sidx];
            nlow[j] = lowp[sidx];
        }
        if (LET && p.row_div) nrd = p.row_div[r];
        if (LET && p.row_mul) nrm = p.row_mul[r];
        if (LET && p.g_wshift) ngws = p.g_wshift[r];
    };
    int64_t r = blockIdx.x;
    if (r < p.rows) prefetch(r);
#pragma unroll 2
    for (; r < p.rows; r += gridDim.x) {
        float w[CH][8], x[CH][8], G[CH][8], cup[CH], clow[CH], hi[CH], lo[CH];
        const float rd = nrd, rm = nrm, gws = ngws;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            cup[j] = nup[j];
            clow[j] = nlow[j];
            hi[j] = nhi_[j];
            lo[j] = nlo_[j];
            nw[j].unpack(w[j]);
            ng[j].unpack(G[j]);
        }
        const int64_t rn = r + gridDim.x;
        if (rn < p.rows) prefetch(rn);
        const float inv_rd = 1.f / rd;      // gradient path only: reciprocal multiplies instead of IEEE divides
        const float rmrd = rm * inv_rd;
        QP qp[CH];
        float gs[CH], inv_s[CH], acc_rm = 0.f;
        int chi[CH], clo[CH], whi = 0, wlo = 0;     // tie counts: per lane (short segments) / per wave (whole rows)
        float gin[CH][8];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            gs[j] = 0.f; chi[j] = 0; clo[j] = 0;
            if (small || j == 0) {
                qp[j] = make_qp(hi[j], lo[j], lwc, cup[j], clow[j], p.nbits, p.symmetric, p.inv_q, &inv_s[j]);
            } else {
                qp[j] = qp[0];
                inv_s[j] = inv_s[0];
            }
            const float z = qp[j].z;
            const float live = (FULL || valid[j]) ? 1.f : 0.f;   // surplus lanes contribute nothing to the sums
            // regular segment (finite non-zero scale): round(t) needs no inf -> NaN replay of the reference's
            // (round(t) - t) + t, which equals rint(t) for every finite t
            const bool regular = qp[j].s != 0.f && fabsf(qp[j].s) <= 3.4028234663852886e38f;
            const uint64_t vmask = FULL ? ~0ull : __builtin_amdgcn_ballot_w64(valid[j]);
            // Elements are processed in PAIRS on packed-f32 arithmetic (v_pk_mul/add/fma_f32: two lanes-worth per
            // instruction at full rate) -- these kernels are VALU-issue-bound, not HBM-bound.
            auto element_pass = [&](auto reg_tag) {
                constexpr bool REG = decltype(reg_tag)::value;
                const f32x2 z2 = {z, z}, is2 = {inv_s[j], inv_s[j]}, rd2 = {rd, rd}, ird2 = {inv_rd, inv_rd}, rm2 = {rm, rm};
                const f32x2 rmrd2 = {rmrd, rmrd}, gws2 = {gws, gws}, live2 = {live, live};
                f32x2 gs2 = {0.f, 0.f}, arm2 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const f32x2 wv = {w[j][i], w[j][i + 1]};
                    const f32x2 gv = {G[j][i], G[j][i + 1]};
                    f32x2 v = wv, a2 = wv;
                    if constexpr (LET) {
                        if (p.col_mul) { const f32x2 cm2 = {cmv[j][i], cmv[j][i + 1]}; v = v * cm2; a2 = v; }
                        if (p.row_div) {                               // div_nr pairwise: same values as the forward
                            const f32x2 q0 = v * ird2;
                            const f32x2 r0 = __builtin_elementwise_fma(-q0, rd2, v);
                            v = __builtin_elementwise_fma(r0, ird2, q0);
                        }
                        if (p.row_mul) v = v * rm2;                    // (ties with hi/lo and the clip mask depend on it)
                    }
                    x[j][i] = v[0];
                    x[j][i + 1] = v[1];
                    const f32x2 tq = v * is2;
                    f32x2 rr;
                    rr[0] = REG ? rintf(tq[0]) : rne_ste(tq[0]);
                    rr[1] = REG ? rintf(tq[1]) : rne_ste(tq[1]);
                    const f32x2 u = rr + z2;
                    f32x2 qv;
                    qv[0] = __builtin_amdgcn_fmed3f(u[0], 0.f, Q);
                    qv[1] = __builtin_amdgcn_fmed3f(u[1], 0.f, Q);
                    const bool in0 = ident || qv[0] == u[0], in1 = ident || qv[1] == u[1];   // inside [0, Q] (false for NaN)
                    // block-wide sums (whole-row segments, row factors) skip the surplus lanes; group-wide sums of the
                    // short-segment mode do not: a surplus GROUP recomputes the last segment completely
                    const f32x2 Gr = FULL ? gv : gv * live2;
                    const f32x2 Gg = (FULL || small) ? gv : Gr;
                    const f32x2 tsel = {in0 ? tq[0] : 0.f, in1 ? tq[1] : 0.f};
                    gs2 = __builtin_elementwise_fma(Gg, (qv - z2) - tsel, gs2);
                    if (small) {
                        chi[j] += ((v[0] == hi[j]) ? 1 : 0) + ((v[1] == hi[j]) ? 1 : 0);
                        clo[j] += ((v[0] == lo[j]) ? 1 : 0) + ((v[1] == lo[j]) ? 1 : 0);
                    } else {
                        // whole-row segments: one v_cmp per element, counted per wave on the scalar unit
                        whi += __builtin_popcountll(__builtin_amdgcn_fcmpf(v[0], hi[j], 1) & vmask);      // FCMP_OEQ
                        whi += __builtin_popcountll(__builtin_amdgcn_fcmpf(v[1], hi[j], 1) & vmask);
                        wlo += __builtin_popcountll(__builtin_amdgcn_fcmpf(v[0], lo[j], 1) & vmask);
                        wlo += __builtin_popcountll(__builtin_amdgcn_fcmpf(v[1], lo[j], 1) & vmask);
                    }
                    gin[j][i] = in0 ? gv[0] : 0.f;
                    gin[j][i + 1] = in1 ? gv[1] : 0.f;
                    if constexpr (LET) {
                        const f32x2 gi = {in0 ? Gr[0] : 0.f, in1 ? Gr[1] : 0.f};
                        if (need_row) arm2 = __builtin_elementwise_fma(gi, a2 * ird2, arm2);     // b = (w*cm)/rd, x = b*rm
                        if (need_cm) {
                            const f32x2 acc = __builtin_elementwise_fma(gi * rmrd2, wv, f32x2{acc_cm[j][i], acc_cm[j][i + 1]});
                            acc_cm[j][i] = acc[0];
                            acc_cm[j][i + 1] = acc[1];
                        }
                        if (need_sh) {
                            const f32x2 acc = __builtin_elementwise_fma(gws2, wv, f32x2{acc_sh[j][i], acc_sh[j][i + 1]});
                            acc_sh[j][i] = acc[0];
                            acc_sh[j][i + 1] = acc[1];
                        }
                    }
                }
                gs[j] += gs2[0] + gs2[1];
                acc_rm += arm2[0] + arm2[1];
            };
            // wave-uniform choice: every lane's segment is regular (the normal case) -> plain rint in the hot loop
            if (__builtin_amdgcn_ballot_w64(!regular) == 0) element_pass(std::true_type{});
            else element_pass(std::false_type{});
        }
        float nhi[CH], nlo[CH];
        bool wave_tie;
        if (small) {
            int mine = 0;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                mine |= chi[j] | clo[j];
                gs[j] = wave_sum(gs[j], lps);
                nhi[j] = wave_sum((float)chi[j], lps);
                nlo[j] = wave_sum((float)clo[j], lps);
            }
            wave_tie = __builtin_amdgcn_ballot_w64(mine != 0) != 0;
        } else {
            wave_tie = (whi | wlo) != 0;
            float v[4] = {0.f, (float)whi, (float)wlo, acc_rm};
#pragma unroll
            for (int j = 0; j < CH; ++j) v[0] += gs[j];
            const int op[4] = {0, 0, 0, 0};
            block_reduce<4>(v, op, red, 0x6u);
#pragma unroll
            for (int j = 0; j < CH; ++j) { gs[j] = v[0]; nhi[j] = v[1]; nlo[j] = v[2]; }
            acc_rm = v[3];
        }
        if (ident) {
#pragma unroll
            for (int j = 0; j < CH; ++j) gs[j] = 0.f;        // no scale in the identity grid: no LWC / tie terms
        }
        // ---- d s / d hi', d s / d lo'  (hi' = su*hi, lo' = sl*lo), LWC gradients, tie terms ------
        const bool any_tie = need_tie && wave_tie && !ident;                            // wave-uniform
        float row_tie = 0.f;        // grouped segments: sum over this thread's segments of the tie part of g_row_mul
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int64_t sidx = r * nseg + segi[j];
            const QP q = qp[j];
            float ds_dhs, ds_dls;   // d scale / d hi', d scale / d lo'
            if (p.symmetric) {
                const float lv = (float)((1 << (p.nbits - 1)) - 1);
                const float hs = q.su * q.hi, ls = q.sl * q.lo;
                const float a = fabsf(hs), b = fabsf(ls);
                const float raw = fmaxf(a, b) / lv;
                const float pass = (raw >= 1e-5f && raw <= 1e4f) ? 1.f : 0.f;
                const float sh = hs > 0.f ? 1.f : (hs < 0.f ? -1.f : 0.f);
                const float sg = ls > 0.f ? 1.f : (ls < 0.f ? -1.f : 0.f);
                const float wa = a > b ? 1.f : (a == b ? 0.5f : 0.f);
                ds_dhs = pass * wa * sh / lv;
                ds_dls = pass * (1.f - wa) * sg / lv;
            } else {
                ds_dhs = 1.f / Q;
                ds_dls = -1.f / Q;
            }
            const float g_hs = gs[j] * ds_dhs;   // dL/d hi'
            const float g_ls = gs[j] * ds_dls;   // dL/d lo'
            // every lane of the segment stores the same LWC gradients (no divergent branch around the store)
            if (small || j == 0) {
                if (p.g_up) p.g_up[sidx] = g_hs * q.hi * q.su * (1.f - q.su);
                if (p.g_low) p.g_low[sidx] = g_ls * q.lo * q.sl * (1.f - q.sl);
            }
            if constexpr (LET) {
                if (need_row) {
                    const float tie = g_hs * q.su * q.hi + g_ls * q.sl * q.lo;
                    if (small) {
                        if ((t & (lps - 1)) == 0 && (FULL || valid[j])) row_tie += tie;   // one lane per segment
                    } else if (j == 0) {
                        const float tot = acc_rm + tie / rm;
                        if (p.g_row_mul) p.g_row_mul[r] = tot;          // same value from every lane
                        if (p.g_row_div) p.g_row_div[r] = -rmrd * tot;
                    }
                }
            }
            if (any_tie) {
                const float tie_hi = g_hs * q.su / nhi[j], tie_lo = g_ls * q.sl / nlo[j];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float tt = 0.f;
                    if (x[j][i] == q.hi) tt += tie_hi;
                    if (x[j][i] == q.lo) tt += tie_lo;
                    if (x[j][i] == q.hi || x[j][i] == q.lo) {
                        gin[j][i] += tt;
                        if constexpr (LET) {
                            if (need_cm) acc_cm[j][i] = fmaf(tt * rmrd, w[j][i], acc_cm[j][i]);
                        }
                    }
                }
            }
            if (p.gx) Vec8<TG>::store(gxbase + r * p.cols + cc[j], gin[j]);
        }
        if constexpr (LET) {
            if (need_row && small) {
                float v[1] = {acc_rm + row_tie / rm};
                const int op[1] = {0};
                block_reduce<1>(v, op, red + 32);
                if (t == 0) {
                    if (p.g_row_mul) p.g_row_mul[r] = v[0];
                    if (p.g_row_div) p.g_row_div[r] = -rmrd * v[0];
                }
            }
        }
    }
    if constexpr (LET) {
      if (p.g_col_mul || p.g_shift) {
        // per-workgroup partial column sums -> workspace (plain coalesced stores); colreduce_kernel finishes.
        float* wcm = p.ws + (int64_t)blockIdx.x * p.cols;
        float* wsh = p.ws + ((int64_t)gridDim.x + blockIdx.x) * p.cols;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (valid[j]) {
                if (p.g_col_mul) Vec8<float>::store(wcm + cc[j], acc_cm[j]);
                if (p.g_shift) Vec8<float>::store(wsh + cc[j], acc_sh[j]);
            }
        }
      }
    }
}

// out[c] = sum_b part[b][c]   (deterministic: fixed order, no atomics).  blockIdx.y selects the vector
// (0: g_col_mul, 1: g_shift).  A workgroup owns 16 columns: 16 row-groups x 16 columns of lanes, 8 independent
// accumulators per lane (8 loads in flight), then one LDS tree over the row-groups.
__device__ __forceinline__ void colreduce_body(const float* part, int nblocks, int64_t cols, float* out0, float* out1) {
    const float* src = part + (int64_t)blockIdx.y * nblocks * cols;
    float* out = blockIdx.y == 0 ? out0 : out1;
    if (!out) return;
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t c = (int64_t)blockIdx.x * 16 + cl;
    __shared__ float red[16][17];
    // 32 independent loads in flight per lane (one memory round trip for up to 512 partial rows): the old loop of
    // 4 dependent batches of 8 was latency-bound at ~11 us for 8 MB
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    if (c < cols) {
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

__global__ void __launch_bounds__(256) colreduce_kernel(const float* part, int nblocks, int64_t cols, float* out0,
                                                        float* out1) {
    colreduce_body(part, nblocks, cols, out0, out1);
}

// the column reductions of several matrices (oq_fakequant_bwd_multi) in one launch: blockIdx.z selects the matrix
constexpr int OQ_WQ_MAX_ = 4;
struct ColRedMulti {
    const float* part[OQ_WQ_MAX_];
    float* out0[OQ_WQ_MAX_];
    float* out1[OQ_WQ_MAX_];
    int nblocks[OQ_WQ_MAX_];
};
__global__ void __launch_bounds__(256) colreduce_multi_kernel(ColRedMulti m, int64_t cols) {
    const int i = blockIdx.z;
    if (m.nblocks[i] > 0) colreduce_body(m.part[i], m.nblocks[i], cols, m.out0[i], m.out1[i]);
}

// chunks per thread (1, 4 or 8) and threads per workgroup for a row of `cols` elements: as many waves as the row
// allows (one 16-byte chunk per lane) before giving a lane more than one chunk.
int64_t dbg_env(const char* name, int64_t dflt) {
    const char* v = getenv(name);
    return v ? atoll(v) : dflt;
}

// Chunks per thread (1/2/4/8) and threads per workgroup for a row of `cols` elements.  Per-row work that every WAVE
// repeats (LWC sigmoids, scale/zero-point arithmetic, reductions, barriers) is amortised over chunks-per-thread, so
// the fewest waves whose registers still fit win: `pref` is the kernel kind's sweet spot (measured), widened only
// when a row needs more than 512 threads and narrowed for short rows.
void row_geometry(int64_t cols, int pref, int* ch, int* bt) {
    const int64_t lanes = (cols + 7) / 8;
    int c = (int)dbg_env("OQ_DBG_FQ_CH", pref);
    while (c > 1 && lanes < 64 * (int64_t)c) c >>= 1;        // short rows: keep one full wave busy
    while (c < 8 && lanes > 512 * (int64_t)c) c <<= 1;       // long rows: at most 512 threads
    int64_t t = (lanes + c - 1) / c;
    t = ((t + 63) / 64) * 64;
    *ch = c;
    *bt = (int)t;
}

int check_shape(const char* fn, int64_t rows, int64_t cols, int64_t seg, int nbits) {
    OQ_CHECK_ARG(rows > 0 && cols > 0 && seg > 0, "%s: empty shape rows=%lld cols=%lld seg=%lld", fn, (long long)rows,
                 (long long)cols, (long long)seg);
    OQ_CHECK_ARG(nbits >= 2 && nbits <= 16, "%s: bitwidth %d not supported (2..15, 16 = identity grid)", fn, nbits);
    OQ_CHECK_ARG(cols % seg == 0, "%s: cols %lld not a multiple of seg %lld (ragged groups unsupported)", fn,
                 (long long)cols, (long long)seg);
    OQ_CHECK_ARG(seg % 8 == 0, "%s: seg %lld must be a multiple of 8", fn, (long long)seg);
    if (seg <= 512) {
        const int64_t l = seg / 8;
        OQ_CHECK_ARG((l & (l - 1)) == 0, "%s: seg/8 = %lld must be a power of two when seg <= 512", fn, (long long)l);
    } else {
        OQ_CHECK_ARG(seg == cols, "%s: seg %lld > 512 must equal cols %lld", fn, (long long)seg, (long long)cols);
    }
    OQ_CHECK_ARG(cols <= 8 * 512 * 8, "%s: cols %lld exceeds %d", fn, (long long)cols, 8 * 512 * 8);
    return OQ_OK;
}


// ---------------------------------------------------------------------------------------------------
// generic-segment path: ANY segment length / row length, last segment of a row zero-padded to the full segment
// ---------------------------------------------------------------------------------------------------
// Reference: quantize/quantizer.py:64-69 (deficiency), :85-87 / :125-128 (zero-pad, reshape to [-1, group]) and
// :103-104 (the pad is cut off again).  One wave per segment, scalar element accesses: this is the shape-complete
// path for the segmentations the vector kernels above do not take (in_features % group_size != 0, groups that are not
// 8 * 2^k wide) -- not a hot path.  Same helpers (make_qp, rne_div) as the vector kernels, so values agree bit for bit
// where both apply.  The zero padding takes part in amax / amin (and in their tie counts in the backward); the padded
// outputs do not exist, so their upstream gradient is zero.  No LET arguments here.
template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) fq_generic_fwd_kernel(FQ p, int64_t nsr /* segments per row */) {
    const int lane = threadIdx.x & 63;
    const int64_t sidx = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (sidx >= p.rows * nsr) return;                        // wave-uniform
    const int64_t r = sidx / nsr, c0 = (sidx % nsr) * p.seg;
    const int64_t valid = p.cols - c0 < p.seg ? p.cols - c0 : p.seg;
    const TIN* w = reinterpret_cast<const TIN*>(p.w) + r * p.cols + c0;
    TOUT* y = reinterpret_cast<TOUT*>(p.y) + r * p.cols + c0;
    const float Q = (float)((1 << p.nbits) - 1);
    float hi = -INFINITY, lo = INFINITY, bad = 0.f;
    for (int64_t i = lane; i < valid; i += 64) {
        const float x = ld1(w + i);
        hi = vmax(hi, x);
        lo = vmin(lo, x);
        if (x != x) bad = 1.f;
    }
    hi = wave_max(hi);
    lo = wave_min(lo);
    bad = wave_max(bad);
    if (valid < p.seg) { hi = fmaxf(hi, 0.f); lo = fminf(lo, 0.f); }
    if (bad != 0.f) { hi = NAN; lo = NAN; }
    const bool lwc = p.up != nullptr;
    float inv_s = 0.f;
    const QP q = make_qp(hi, lo, lwc, lwc ? p.up[sidx] : 0.f, lwc ? p.low[sidx] : 0.f, p.nbits, p.symmetric, p.inv_q, &inv_s);
    const bool regular = q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f && bad == 0.f;
    for (int64_t i = lane; i < valid; i += 64) {
        const float x = ld1(w + i);
        float yv;
        if (p.nbits >= 16) {
            yv = x;
        } else if (regular) {
            float tq;
            const float rq = rne_div(x, q.s, inv_s, &tq);
            yv = (__builtin_amdgcn_fmed3f(rq + q.z, 0.f, Q) - q.z) * q.s;
        } else {
            float v = rne_ste(x / q.s) + q.z;
            v = (v != v) ? v : fminf(fmaxf(v, 0.f), Q);
            yv = (v - q.z) * q.s;
        }
        st1(y + i, yv);
    }
    if (lane == 0) {
        p.scale[sidx] = q.s;
        p.zp[sidx] = q.z;
        p.xmin[sidx] = lo;
        p.xmax[sidx] = hi;
    }
}

template <typename TIN, typename TG>
__global__ void __launch_bounds__(256) fq_generic_bwd_kernel(FQ p, int64_t nsr) {
    const int lane = threadIdx.x & 63;
    const int64_t sidx = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (sidx >= p.rows * nsr) return;
    const int64_t r = sidx / nsr, c0 = (sidx % nsr) * p.seg;
    const int64_t valid = p.cols - c0 < p.seg ? p.cols - c0 : p.seg;
    const TIN* w = reinterpret_cast<const TIN*>(p.w) + r * p.cols + c0;
    const TG* g = reinterpret_cast<const TG*>(p.g) + r * p.cols + c0;
    const float Q = (float)((1 << p.nbits) - 1);
    const bool lwc = p.up != nullptr, ident = p.nbits >= 16;
    const float hi = p.xmax[sidx], lo = p.xmin[sidx];
    float inv_s = 0.f;
    const QP q = make_qp(hi, lo, lwc, lwc ? p.up[sidx] : 0.f, lwc ? p.low[sidx] : 0.f, p.nbits, p.symmetric, p.inv_q, &inv_s);
    const bool regular = q.s != 0.f && fabsf(q.s) <= 3.4028234663852886e38f;
    float gs = 0.f, nhi = 0.f, nlo = 0.f;
    for (int64_t i = lane; i < valid; i += 64) {
        const float x = ld1(w + i), G = ld1(g + i);
        const float tq = x * inv_s;
        const float u = (regular ? rintf(tq) : rne_ste(tq)) + q.z;
        const float qv = __builtin_amdgcn_fmed3f(u, 0.f, Q);
        const bool in = qv == u;
        gs = fmaf(G, (qv - q.z) - (in ? tq : 0.f), gs);
        nhi += x == hi ? 1.f : 0.f;
        nlo += x == lo ? 1.f : 0.f;
    }
    gs = ident ? 0.f : wave_sum(gs);
    nhi = wave_sum(nhi);
    nlo = wave_sum(nlo);
    if (valid < p.seg) {                                     // the padding zeros share the amax / amin gradient
        if (hi == 0.f) nhi += (float)(p.seg - valid);
        if (lo == 0.f) nlo += (float)(p.seg - valid);
    }
    float ds_dhs, ds_dls;
    if (p.symmetric) {
        const float lv = (float)((1 << (p.nbits - 1)) - 1);
        const float hs = q.su * q.hi, ls = q.sl * q.lo;
        const float a = fabsf(hs), b = fabsf(ls);
        const float raw = fmaxf(a, b) / lv;
        const float pass = (raw >= 1e-5f && raw <= 1e4f) ? 1.f : 0.f;
        const float sh = hs > 0.f ? 1.f : (hs < 0.f ? -1.f : 0.f);
        const float sg = ls > 0.f ? 1.f : (ls < 0.f ? -1.f : 0.f);
        const float wa = a > b ? 1.f : (a == b ? 0.5f : 0.f);
        ds_dhs = pass * wa * sh / lv;
        ds_dls = pass * (1.f - wa) * sg / lv;
    } else {
        ds_dhs = 1.f / Q;
        ds_dls = -1.f / Q;
    }
    const float g_hs = gs * ds_dhs, g_ls = gs * ds_dls;
    if (lane == 0) {
        if (p.g_up) p.g_up[sidx] = g_hs * q.hi * q.su * (1.f - q.su);
        if (p.g_low) p.g_low[sidx] = g_ls * q.lo * q.sl * (1.f - q.sl);
    }
    if (p.gx) {
        TG* gx = reinterpret_cast<TG*>(p.gx) + r * p.cols + c0;
        const float tie_hi = g_hs * q.su / nhi, tie_lo = g_ls * q.sl / nlo;
        for (int64_t i = lane; i < valid; i += 64) {
            const float x = ld1(w + i), G = ld1(g + i);
            const float tq = x * inv_s;
            const float u = (regular ? rintf(tq) : rne_ste(tq)) + q.z;
            const bool in = ident || __builtin_amdgcn_fmed3f(u, 0.f, Q) == u;
            float o = in ? G : 0.f;
            if (!ident) {
                if (x == hi) o += tie_hi;
                if (x == lo) o += tie_lo;
            }
            st1(gx + i, o);
        }
    }
}

// true when the vector kernels cannot take this segmentation (see check_shape)
bool needs_generic(int64_t cols, int64_t seg) {
    if (seg <= 0 || cols <= 0) return false;
    if (cols % seg != 0 || seg % 8 != 0 || cols > 8 * 512 * 8) return true;
    if (seg <= 512) { const int64_t l = seg / 8; return (l & (l - 1)) != 0; }
    return seg != cols;
}

}  // namespace

#define FQ_LAUNCH_FWD(TIN, TOUT, L, C_)                                                                              \
    do {                                                                                                             \
        if (full) hipLaunchKernelGGL((fq_fwd_kernel<TIN, TOUT, L, C_, true>), dim3(grid), dim3(bt), 0, (hipStream_t)stream, p);  \
        else hipLaunchKernelGGL((fq_fwd_kernel<TIN, TOUT, L, C_, false>), dim3(grid), dim3(bt), 0, (hipStream_t)stream, p);      \
    } while (0)
#define FQ_DISPATCH_FWD(TIN, TOUT)                                     \
    do {                                                               \
        if (let) {                                                     \
            if (ch == 1) FQ_LAUNCH_FWD(TIN, TOUT, true, 1);            \
            else if (ch == 2) FQ_LAUNCH_FWD(TIN, TOUT, true, 2);       \
            else if (ch == 4) FQ_LAUNCH_FWD(TIN, TOUT, true, 4);       \
            else FQ_LAUNCH_FWD(TIN, TOUT, true, 8);                    \
        } else {                                                       \
            if (ch == 1) FQ_LAUNCH_FWD(TIN, TOUT, false, 1);           \
            else if (ch == 2) FQ_LAUNCH_FWD(TIN, TOUT, false, 2);      \
            else if (ch == 4) FQ_LAUNCH_FWD(TIN, TOUT, false, 4);      \
            else FQ_LAUNCH_FWD(TIN, TOUT, false, 8);                   \
        }                                                              \
    } while (0)

extern "C" int oq_fakequant_fwd(const void* w, int w_dtype, int64_t rows, int64_t cols, int64_t seg, int nbits,
                                int symmetric, const float* col_mul, const float* row_div, const float* row_mul,
                                const float* shift, const float* up, const float* low, void* y, int y_dtype,
                                float* scale, float* zp, float* xmin, float* xmax, float* wshift, void* codes, float* csum,
                                void* stream) {
    OQ_CHECK_ARG((codes == nullptr) == (csum == nullptr), "oq_fakequant_fwd: codes and csum must both be given or both NULL");
    if (codes && !oq_fakequant_codes_supported(cols, seg, nbits, (col_mul || row_div || row_mul || shift) ? 1 : 0)) {
        oq_set_error("oq_fakequant_fwd: integer codes are not available for cols %lld seg %lld nbits %d "
                     "(oq_fakequant_codes_supported)", (long long)cols, (long long)seg, nbits);
        return OQ_E_UNSUPPORTED;
    }
    if (needs_generic(cols, seg)) {
        OQ_CHECK_ARG(rows > 0 && nbits >= 2 && nbits <= 16, "oq_fakequant_fwd: bad shape / bitwidth %d", nbits);
        OQ_CHECK_ARG(w && y && scale && zp && xmin && xmax, "oq_fakequant_fwd: null pointer");
        OQ_CHECK_ARG((up == nullptr) == (low == nullptr), "oq_fakequant_fwd: up/low must both be given or both NULL");
        if (col_mul || row_div || row_mul || shift || wshift) {
            oq_set_error("oq_fakequant_fwd: the LET transform is not available for ragged / non-power-of-two groups "
                         "(cols %lld, seg %lld)", (long long)cols, (long long)seg);
            return OQ_E_UNSUPPORTED;
        }
        FQ g{};
        g.w = w; g.rows = rows; g.cols = cols; g.seg = seg; g.nbits = nbits; g.symmetric = symmetric;
        g.inv_q = 1.0f / (float)((1 << nbits) - 1);
        g.up = up; g.low = low; g.y = y; g.scale = scale; g.zp = zp; g.xmin = xmin; g.xmax = xmax;
        const int64_t nsr = (cols + seg - 1) / seg, nseg = rows * nsr;
        const dim3 grid((unsigned)((nseg + 3) / 4));
        const int gkey = w_dtype * 3 + y_dtype;
#define FQ_GEN_FWD(TIN, TOUT) hipLaunchKernelGGL((fq_generic_fwd_kernel<TIN, TOUT>), grid, dim3(256), 0, (hipStream_t)stream, g, nsr)
        switch (gkey) {
            case OQ_F32 * 3 + OQ_F32: FQ_GEN_FWD(float, float); break;
            case OQ_F32 * 3 + OQ_BF16: FQ_GEN_FWD(float, bf16_t); break;
            case OQ_F16 * 3 + OQ_F32: FQ_GEN_FWD(f16_t, float); break;
            case OQ_F16 * 3 + OQ_BF16: FQ_GEN_FWD(f16_t, bf16_t); break;
            case OQ_BF16 * 3 + OQ_BF16: FQ_GEN_FWD(bf16_t, bf16_t); break;
            default:
                oq_set_error("oq_fakequant_fwd: unsupported dtype pair in=%d out=%d", w_dtype, y_dtype);
                return OQ_E_UNSUPPORTED;
        }
        OQ_CHECK_LAUNCH("oq_fakequant_fwd(generic)");
        return OQ_OK;
    }
    int rc = check_shape("oq_fakequant_fwd", rows, cols, seg, nbits);
    if (rc) return rc;
    OQ_CHECK_ARG(w && y, "oq_fakequant_fwd: null w/y");
    OQ_CHECK_ARG(scale && zp && xmin && xmax, "oq_fakequant_fwd: scale/zp/xmin/xmax outputs are required");
    OQ_CHECK_ARG(oq_aligned16(w) && oq_aligned16(y) && oq_aligned16(col_mul) && oq_aligned16(shift),
                 "oq_fakequant_fwd: w/y/col_mul/shift must be 16-byte aligned");
    OQ_CHECK_ARG((up == nullptr) == (low == nullptr), "oq_fakequant_fwd: up/low must both be given or both NULL");
    OQ_CHECK_ARG(!wshift || shift, "oq_fakequant_fwd: wshift requested without shift");
    FQ p{};
    p.w = w; p.rows = rows; p.cols = cols; p.seg = seg; p.nbits = nbits; p.symmetric = symmetric;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.col_mul = col_mul; p.row_div = row_div; p.row_mul = row_mul; p.shift = wshift ? shift : nullptr;
    p.up = up; p.low = low; p.y = y; p.scale = scale; p.zp = zp; p.xmin = xmin; p.xmax = xmax; p.wshift = wshift;
    p.codes = (int8_t*)codes; p.csum = csum;
    {   // whole-row segments: the wave-per-row kernels (oq_rowq.hip) take them when the shape is theirs
        const int rq = oq_rowq_fwd(p, w_dtype, y_dtype, stream);
        if (rq <= 0) return rq;
    }
    if (codes) {
        oq_set_error("oq_fakequant_fwd: integer codes requested but the wave-per-row kernels did not take the problem");
        return OQ_E_UNSUPPORTED;
    }
    if (oq_groupq_eligible(p, false)) {     // grouped weights without LET: the lanes-per-group kernels (oq_groupq.hip)
        const int gq = oq_groupq_fwd(p, w_dtype, y_dtype, stream);
        if (gq <= 0) return gq;
    }
    const bool let = col_mul || row_div || row_mul || shift;
    int ch, bt;
    row_geometry(cols, 2, &ch, &bt);
    const bool full = (int64_t)ch * bt * 8 == cols;
    const int64_t gcap = dbg_env("OQ_DBG_FQ_BLOCKS", OQ_FQ_MAX_BLOCKS);
    const int64_t grid = rows < gcap ? rows : gcap;
    const int key = w_dtype * 3 + y_dtype;
    switch (key) {
        case OQ_F32 * 3 + OQ_F32: FQ_DISPATCH_FWD(float, float); break;
        case OQ_F32 * 3 + OQ_BF16: FQ_DISPATCH_FWD(float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: FQ_DISPATCH_FWD(f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: FQ_DISPATCH_FWD(f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: FQ_DISPATCH_FWD(bf16_t, bf16_t); break;
        default:
            oq_set_error("oq_fakequant_fwd: unsupported dtype pair in=%d out=%d", w_dtype, y_dtype);
            return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_fakequant_fwd");
    return OQ_OK;
}

#define FQ_LAUNCH_BWD(TIN, TG, L, C_)                                                                              \
    do {                                                                                                           \
        if (full) hipLaunchKernelGGL((fq_bwd_kernel<TIN, TG, L, C_, true>), dim3(grid), dim3(bt), 0, (hipStream_t)stream, p);  \
        else hipLaunchKernelGGL((fq_bwd_kernel<TIN, TG, L, C_, false>), dim3(grid), dim3(bt), 0, (hipStream_t)stream, p);      \
    } while (0)
#define FQ_DISPATCH_BWD(TIN, TG)                                     \
    do {                                                             \
        if (let) {                                                   \
            if (ch == 1) FQ_LAUNCH_BWD(TIN, TG, true, 1);            \
            else if (ch == 2) FQ_LAUNCH_BWD(TIN, TG, true, 2);       \
            else if (ch == 4) FQ_LAUNCH_BWD(TIN, TG, true, 4);       \
            else FQ_LAUNCH_BWD(TIN, TG, true, 8);                    \
        } else {                                                     \
            if (ch == 1) FQ_LAUNCH_BWD(TIN, TG, false, 1);           \
            else if (ch == 2) FQ_LAUNCH_BWD(TIN, TG, false, 2);      \
            else if (ch == 4) FQ_LAUNCH_BWD(TIN, TG, false, 4);      \
            else FQ_LAUNCH_BWD(TIN, TG, false, 8);                   \
        }                                                            \
    } while (0)

extern "C" int oq_fakequant_bwd(const void* w, int w_dtype, int64_t rows, int64_t cols, int64_t seg, int nbits,
                                int symmetric, const float* col_mul, const float* row_div, const float* row_mul,
                                const float* shift, const float* up, const float* low, const float* xmin,
                                const float* xmax, const void* g, int g_dtype,
                                const float* g_wshift, float* g_up, float* g_low, void* gx, int gx_dtype,
                                float* g_col_mul, float* g_shift, float* g_row_div, float* g_row_mul,
                                float* workspace, int64_t workspace_floats, void* stream) {
    if (needs_generic(cols, seg)) {
        OQ_CHECK_ARG(rows > 0 && nbits >= 2 && nbits <= 16, "oq_fakequant_bwd: bad shape / bitwidth %d", nbits);
        OQ_CHECK_ARG(w && g && xmin && xmax, "oq_fakequant_bwd: null pointer");
        OQ_CHECK_ARG(!gx || gx_dtype == g_dtype, "oq_fakequant_bwd: gx dtype must equal g dtype");
        OQ_CHECK_ARG((up == nullptr) == (low == nullptr), "oq_fakequant_bwd: up/low must both be given or both NULL");
        if (col_mul || row_div || row_mul || shift || g_col_mul || g_shift || g_row_div || g_row_mul) {
            oq_set_error("oq_fakequant_bwd: the LET transform is not available for ragged / non-power-of-two groups "
                         "(cols %lld, seg %lld)", (long long)cols, (long long)seg);
            return OQ_E_UNSUPPORTED;
        }
        FQ q{};
        q.w = w; q.rows = rows; q.cols = cols; q.seg = seg; q.nbits = nbits; q.symmetric = symmetric;
        q.inv_q = 1.0f / (float)((1 << nbits) - 1);
        q.up = up; q.low = low; q.g = g; q.g_up = g_up; q.g_low = g_low; q.gx = gx;
        q.xmin = const_cast<float*>(xmin); q.xmax = const_cast<float*>(xmax);
        const int64_t nsr = (cols + seg - 1) / seg, nseg = rows * nsr;
        const dim3 grid((unsigned)((nseg + 3) / 4));
        const int gkey = w_dtype * 3 + g_dtype;
#define FQ_GEN_BWD(TIN, TG) hipLaunchKernelGGL((fq_generic_bwd_kernel<TIN, TG>), grid, dim3(256), 0, (hipStream_t)stream, q, nsr)
        switch (gkey) {
            case OQ_F32 * 3 + OQ_F32: FQ_GEN_BWD(float, float); break;
            case OQ_F32 * 3 + OQ_BF16: FQ_GEN_BWD(float, bf16_t); break;
            case OQ_F16 * 3 + OQ_F32: FQ_GEN_BWD(f16_t, float); break;
            case OQ_F16 * 3 + OQ_BF16: FQ_GEN_BWD(f16_t, bf16_t); break;
            case OQ_BF16 * 3 + OQ_BF16: FQ_GEN_BWD(bf16_t, bf16_t); break;
            default:
                oq_set_error("oq_fakequant_bwd: unsupported dtype pair w=%d g=%d", w_dtype, g_dtype);
                return OQ_E_UNSUPPORTED;
        }
        OQ_CHECK_LAUNCH("oq_fakequant_bwd(generic)");
        return OQ_OK;
    }
    int rc = check_shape("oq_fakequant_bwd", rows, cols, seg, nbits);
    if (rc) return rc;
    OQ_CHECK_ARG(w && g, "oq_fakequant_bwd: null w/g");
    OQ_CHECK_ARG(xmin && xmax, "oq_fakequant_bwd: xmin/xmax (the forward's per-segment min/max) are required");
    OQ_CHECK_ARG(oq_aligned16(w) && oq_aligned16(g) && oq_aligned16(gx) && oq_aligned16(col_mul),
                 "oq_fakequant_bwd: 16-byte alignment");
    OQ_CHECK_ARG(!gx || gx_dtype == g_dtype, "oq_fakequant_bwd: gx dtype must equal g dtype");
    OQ_CHECK_ARG((up == nullptr) == (low == nullptr), "oq_fakequant_bwd: up/low must both be given or both NULL");
    OQ_CHECK_ARG(!g_shift || g_wshift, "oq_fakequant_bwd: g_shift needs g_wshift");
    OQ_CHECK_ARG(!g_col_mul || col_mul, "oq_fakequant_bwd: g_col_mul needs col_mul");
    OQ_CHECK_ARG(!g_row_div || row_div, "oq_fakequant_bwd: g_row_div needs row_div");
    OQ_CHECK_ARG(!g_row_mul || row_mul, "oq_fakequant_bwd: g_row_mul needs row_mul");
    FQ p{};
    p.w = w; p.rows = rows; p.cols = cols; p.seg = seg; p.nbits = nbits; p.symmetric = symmetric;
    p.inv_q = 1.0f / (float)((1 << nbits) - 1);
    p.col_mul = col_mul; p.row_div = row_div; p.row_mul = row_mul; p.shift = shift; p.up = up; p.low = low;
    p.g = g; p.g_wshift = g_wshift; p.g_up = g_up; p.g_low = g_low; p.gx = gx;
    p.xmin = const_cast<float*>(xmin); p.xmax = const_cast<float*>(xmax);
    p.g_col_mul = g_col_mul; p.g_shift = g_shift; p.g_row_div = g_row_div; p.g_row_mul = g_row_mul;
    {   // whole-row segments: the wave-per-row kernels (oq_rowq.hip) take them when the shape is theirs
        int64_t parts = 0;
        const int rq = oq_rowq_bwd(p, w_dtype, g_dtype, workspace, workspace_floats, &parts, stream);
        if (rq < 0) return rq;
        if (rq == 0) {
            if (parts > 0) {
                const dim3 rg((unsigned)((cols + 15) / 16), 2);
                hipLaunchKernelGGL(colreduce_kernel, rg, dim3(256), 0, (hipStream_t)stream, workspace, (int)parts, cols, g_col_mul, g_shift);
                OQ_CHECK_LAUNCH("oq_fakequant_bwd(colreduce)");
            }
            return OQ_OK;
        }
    }
    if (oq_groupq_eligible(p, true)) {      // grouped frozen weights without LET: the lanes-per-group kernels (oq_groupq.hip)
        const int gq = oq_groupq_bwd(p, w_dtype, g_dtype, stream);
        if (gq <= 0) return gq;
    }
    // the LET instantiation is needed whenever the transform is present (x must be recomputed), not only for its grads
    const bool let = col_mul || row_div || row_mul || g_col_mul || g_shift || g_row_div || g_row_mul;
    int ch, bt;
    row_geometry(cols, 2, &ch, &bt);
    const bool full = (int64_t)ch * bt * 8 == cols;
    // column accumulators are flushed once per workgroup: keep the grid small when they are live
    const int64_t cap = (g_col_mul || g_shift) ? dbg_env("OQ_DBG_FQ_BWD_BLOCKS", OQ_FQ_BWD_MAX_BLOCKS) : dbg_env("OQ_DBG_FQ_BLOCKS", OQ_FQ_MAX_BLOCKS);
    const int64_t grid = rows < cap ? rows : cap;
    if (g_col_mul || g_shift) {
        OQ_CHECK_ARG(workspace && workspace_floats >= 2 * grid * cols,
                     "oq_fakequant_bwd: workspace of %lld floats needed (oq_fakequant_bwd_workspace)", (long long)(2 * grid * cols));
        p.ws = workspace;
    }
    const int key = w_dtype * 3 + g_dtype;
    switch (key) {
        case OQ_F32 * 3 + OQ_F32: FQ_DISPATCH_BWD(float, float); break;
        case OQ_F32 * 3 + OQ_BF16: FQ_DISPATCH_BWD(float, bf16_t); break;
        case OQ_F16 * 3 + OQ_F32: FQ_DISPATCH_BWD(f16_t, float); break;
        case OQ_F16 * 3 + OQ_BF16: FQ_DISPATCH_BWD(f16_t, bf16_t); break;
        case OQ_BF16 * 3 + OQ_BF16: FQ_DISPATCH_BWD(bf16_t, bf16_t); break;
        default:
            oq_set_error("oq_fakequant_bwd: unsupported dtype pair w=%d g=%d", w_dtype, g_dtype);
            return OQ_E_UNSUPPORTED;
    }
    if (g_col_mul || g_shift) {
        const dim3 rg((unsigned)((cols + 15) / 16), 2);
        hipLaunchKernelGGL(colreduce_kernel, rg, dim3(256), 0, (hipStream_t)stream, workspace, (int)grid, cols, g_col_mul, g_shift);
    }
    OQ_CHECK_LAUNCH("oq_fakequant_bwd");
    return OQ_OK;
}

// ---- several weight matrices per call (include/oq_hip.h: oq_fakequant_fwd_multi / oq_fakequant_bwd_multi) -------------------
// Same results as calling the single-matrix entry points in order; ONE launch per direction (plus one for the column
// reductions) when all problems are LET weights of one row length the row-group kernels take.
extern "C" int oq_fakequant_fwd_multi(const oq_fakequant_fwd_args* a, int n, void* stream) {
    OQ_CHECK_ARG(a && n > 0, "oq_fakequant_fwd_multi: bad args");
    if (n > 1 && n <= OQ_WQ_MAX_) {
        FQ ps[OQ_WQ_MAX_];
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            const oq_fakequant_fwd_args& t = a[i];
            ok = !needs_generic(t.cols, t.seg) && check_shape("oq_fakequant_fwd_multi", t.rows, t.cols, t.seg, t.nbits) == OQ_OK &&
                 t.w && t.y && t.scale && t.zp && t.xmin && t.xmax && oq_aligned16(t.w) && oq_aligned16(t.y) &&
                 oq_aligned16(t.col_mul) && oq_aligned16(t.shift) && ((t.up == nullptr) == (t.low == nullptr)) &&
                 (!t.wshift || t.shift) && t.w_dtype == a[0].w_dtype && t.y_dtype == a[0].y_dtype &&
                 ((t.codes == nullptr) == (t.csum == nullptr)) &&
                 (!t.codes || oq_fakequant_codes_supported(t.cols, t.seg, t.nbits, (t.col_mul || t.row_div || t.row_mul || t.shift) ? 1 : 0));
            if (!ok) break;
            FQ p{};
            p.w = t.w; p.rows = t.rows; p.cols = t.cols; p.seg = t.seg; p.nbits = t.nbits; p.symmetric = t.symmetric;
            p.inv_q = 1.0f / (float)((1 << t.nbits) - 1);
            p.col_mul = t.col_mul; p.row_div = t.row_div; p.row_mul = t.row_mul; p.shift = t.wshift ? t.shift : nullptr;
            p.up = t.up; p.low = t.low; p.y = t.y; p.scale = t.scale; p.zp = t.zp; p.xmin = t.xmin; p.xmax = t.xmax;
            p.wshift = t.wshift;
            p.codes = (int8_t*)t.codes; p.csum = t.csum;
            ps[i] = p;
        }
        if (ok) {
            const int rc = oq_letq_fwd_multi(ps, n, a[0].w_dtype, a[0].y_dtype, stream);
            if (rc <= 0) return rc;
        }
    }
    for (int i = 0; i < n; ++i) {
        const oq_fakequant_fwd_args& t = a[i];
        const int rc = oq_fakequant_fwd(t.w, t.w_dtype, t.rows, t.cols, t.seg, t.nbits, t.symmetric, t.col_mul, t.row_div, t.row_mul,
                                        t.shift, t.up, t.low, t.y, t.y_dtype, t.scale, t.zp, t.xmin, t.xmax, t.wshift, t.codes, t.csum,
                                        stream);
        if (rc) return rc;
    }
    return OQ_OK;
}

extern "C" int oq_fakequant_bwd_multi(const oq_fakequant_bwd_args* a, int n, void* stream) {
    OQ_CHECK_ARG(a && n > 0, "oq_fakequant_bwd_multi: bad args");
    if (n > 1 && n <= OQ_WQ_MAX_) {
        FQ ps[OQ_WQ_MAX_];
        int64_t wsf[OQ_WQ_MAX_], parts[OQ_WQ_MAX_];
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            const oq_fakequant_bwd_args& t = a[i];
            ok = !needs_generic(t.cols, t.seg) && check_shape("oq_fakequant_bwd_multi", t.rows, t.cols, t.seg, t.nbits) == OQ_OK &&
                 t.w && t.g && t.xmin && t.xmax && oq_aligned16(t.w) && oq_aligned16(t.g) && oq_aligned16(t.col_mul) && !t.gx &&
                 ((t.up == nullptr) == (t.low == nullptr)) && (!t.g_shift || t.g_wshift) && (!t.g_col_mul || t.col_mul) &&
                 (!t.g_row_div || t.row_div) && (!t.g_row_mul || t.row_mul) && t.w_dtype == a[0].w_dtype && t.g_dtype == a[0].g_dtype;
            if (!ok) break;
            FQ p{};
            p.w = t.w; p.rows = t.rows; p.cols = t.cols; p.seg = t.seg; p.nbits = t.nbits; p.symmetric = t.symmetric;
            p.inv_q = 1.0f / (float)((1 << t.nbits) - 1);
            p.col_mul = t.col_mul; p.row_div = t.row_div; p.row_mul = t.row_mul; p.shift = t.shift; p.up = t.up; p.low = t.low;
            p.g = t.g; p.g_wshift = t.g_wshift; p.g_up = t.g_up; p.g_low = t.g_low; p.gx = nullptr;
            p.xmin = const_cast<float*>(t.xmin); p.xmax = const_cast<float*>(t.xmax);
            p.g_col_mul = t.g_col_mul; p.g_shift = t.g_shift; p.g_row_div = t.g_row_div; p.g_row_mul = t.g_row_mul;
            p.ws = t.workspace;
            ps[i] = p;
            wsf[i] = t.workspace_floats;
        }
        if (ok) {
            const int rc = oq_letq_bwd_multi(ps, n, a[0].w_dtype, a[0].g_dtype, wsf, parts, stream);
            if (rc < 0) return rc;
            if (rc == 0) {
                ColRedMulti m{};
                bool any = false;
                for (int i = 0; i < n; ++i) {
                    m.part[i] = a[i].workspace; m.out0[i] = a[i].g_col_mul; m.out1[i] = a[i].g_shift; m.nblocks[i] = (int)parts[i];
                    any = any || parts[i] > 0;
                }
                if (any) {
                    const dim3 rg((unsigned)((a[0].cols + 15) / 16), 2, (unsigned)n);
                    hipLaunchKernelGGL(colreduce_multi_kernel, rg, dim3(256), 0, (hipStream_t)stream, m, a[0].cols);
                    OQ_CHECK_LAUNCH("oq_fakequant_bwd_multi(colreduce)");
                }
                return OQ_OK;
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        const oq_fakequant_bwd_args& t = a[i];
        const int rc = oq_fakequant_bwd(t.w, t.w_dtype, t.rows, t.cols, t.seg, t.nbits, t.symmetric, t.col_mul, t.row_div, t.row_mul,
                                        t.shift, t.up, t.low, t.xmin, t.xmax, t.g, t.g_dtype, t.g_wshift, t.g_up, t.g_low, t.gx,
                                        t.gx_dtype, t.g_col_mul, t.g_shift, t.g_row_div, t.g_row_mul, t.workspace, t.workspace_floats,
                                        stream);
        if (rc) return rc;
    }
    return OQ_OK;
}

extern "C" int64_t oq_fakequant_bwd_workspace(int64_t rows, int64_t cols) {
    const int64_t cap = dbg_env("OQ_DBG_FQ_BWD_BLOCKS", OQ_FQ_BWD_MAX_BLOCKS);
    int64_t grid = rows < cap ? rows : cap;
    const int64_t rq = oq_rowq_bwd_blocks(rows, cols);      // the wave-per-row kernels write one partial row per workgroup
    if (rq > grid) grid = rq;
    const int64_t lq = oq_letq_bwd_blocks(rows);            // row-group LET kernels: one partial row per workgroup
    if (lq > grid) grid = lq;
    return 2 * grid * cols;
}
