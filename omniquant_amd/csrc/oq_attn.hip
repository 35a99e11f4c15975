// Fused causal attention for the calibration step (gfx950), bf16, head_dim 128.
//
// Replaces, for the exact-causal-mask case, the chain at models/int_llama_layer.py:143-163
//     qkt_matmul(q, k^T) / sqrt(hd)  ->  + mask, max(finfo.min)  ->  softmax(fp32)  ->  pv_matmul(p, v)
// and its autograd, without ever writing the [heads, T, T] score / probability tensors:
//   attn_fwd_kernel      O = softmax(scale * Q K^T) V, online softmax in f32, also stores the row log-sum-exp
//   attn_bwd_prep_kernel D[q] = sum_d O[q,d] dO[q,d]
//   attn_bwd_kernel      one workgroup per 64-key tile: recomputes P from (Q, K, lse), accumulates dK and dV in
//                        registers over the query tiles at/below the diagonal and writes dS^T (bf16, [key][query]);
//                        dQ = dS K is then one causal GEMM (oq_gemm, tri_mode 2) -- no atomics, deterministic.
// q/k/v fake-quant (the 4-bit head-wise activations of W4A4) happens BEFORE these kernels exactly as in the
// reference; the p-quantiser is the identity at the reference's default 16 bits (the caller falls back to the
// unfused kernels otherwise).
//
// Layout: q, o, dO, dQ [bs, T, nh, 128]; k, v [bs, T, nkv, 128]; dK, dV [bs, T, nh, 128] (per q-head; the caller
// reduces GQA groups).  LDS tiles are [64 rows][128 d] bf16 (256-B rows) with the 16-B chunk index XOR-swizzled by
// s(row) (see swz()): conflict-free for both ds_read_b128 row fragments and ds_read_b64_tr_b16 transposed fragments.
//
// MFMA operand trick (no cross-lane shuffles between the two GEMMs of a tile): the first product is computed so that
// the softmax'd block comes out of the accumulator with the contraction index of the SECOND product along the
// register index i (C row = 4*(lane>>4)+i); two such blocks (rows 0-15 and 16-31 of a 32-chunk) packed in-lane give a
// 16x16x32 operand whose k-slot 8g+i <-> row 4g+i and 8g+4+i <-> row 16+4g+i; the other operand is read from LDS with
// the same permutation (two transposed reads, rows 4g+q and 16+4g+q).
#include "oq_common.h"

namespace {

constexpr int HD = 128;
constexpr int TILE = 64 * HD * 2;   // 16 KiB

struct AttnP {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    bf16_t* o;
    float* lse;            // [bs, nh, T] log2-domain log-sum-exp of scale*log2e*S
    const bf16_t* go;
    const float* dsum;     // [bs, nh, T]
    bf16_t* gk;
    bf16_t* gv;
    bf16_t* dst;           // dS^T [bs, nh, T(key), T(query)]
    int64_t T;
    int nh, nkv;
    float scale, scale_log2;
    // grid mode (oq_attn_*_grid): q / k / v hold the quantisers' integer grid coordinates (code - zero_point, exact in bf16);
    // value = coordinate * s?[(b*T + t) * lds + head]
    const float *sq, *sk, *sv;
    int64_t lds;
    float* o32;            // optional second copy of O in f32 (grid forward)
};

// s(row): built for the DOCUMENTED lane groups of the two read instructions (MI355X_MICROARCH.md, LDS table):
//   ds_read_b128 is served in 4 groups of 16 NON-contiguous lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...): a group
//   mixes rows {0-3,12-15} at chunk c with rows {4-11} at chunk c+1, so s must map {0-3,12-15} and {4-11} onto chunk
//   sets that are each closed under XOR 1;  ds_read_b64_tr_b16 is served in two 32-lane halves = 8 consecutive rows x
//   32 B, so s>>1 must be distinct over every aligned octet of rows.  (The first version used (row&7)<<1 | row>>3&1:
//   fine for the transposed reads, 2-way conflicts on every ds_read_b128 -- SQ_LDS_BANK_CONFLICT = 24-27 % of
//   SQ_LDS_IDX_ACTIVE.)
__device__ __forceinline__ int swz(int row) {
    const int hi = (row >> 3) & 1;
    return ((((row & 7) ^ (hi << 2)) << 1) | hi);
}

__device__ __forceinline__ void tile_gload(const bf16_t* base, int64_t ld, int tid, u32x4 (&r)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
        r[p] = *reinterpret_cast<const u32x4*>(base + (int64_t)((tid >> 4) + 16 * p) * ld + (tid & 15) * 8);
}

__device__ __forceinline__ void tile_sstore(char* lds, int tid, const u32x4 (&r)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (tid >> 4) + 16 * p, c = tid & 15;
        *reinterpret_cast<u32x4*>(lds + row * 256 + ((c ^ swz(row)) << 4)) = r[p];
    }
}

// k-contiguous fragment: lane's 8 bf16 = X[row0 + (lane&15)][32*ks + 8*(lane>>4) .. +8]
__device__ __forceinline__ bf16x8 frag_row(const char* lds, int row0, int ks, int lane) {
    const int row = row0 + (lane & 15), c = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + row * 256 + ((c ^ swz(row)) << 4));
}

// transposed fragment with the permuted k-slots: element j<4 = X[r0 + 4g + j][col0 + (lane&15)],
// element 4+j = X[r0 + 16 + 4g + j][col0 + (lane&15)]   (g = lane>>4, col0 a multiple of 16)
__device__ __forceinline__ bf16x8 frag_tr(const char* lds, int r0, int col0, int lane) {
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int ra = r0 + 4 * g + qq, rb = ra + 16;
    const int c = (col0 >> 3) + (pp >> 1), sub = (pp & 1) * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + ra * 256 + ((c ^ swz(ra)) << 4) + sub));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + rb * 256 + ((c ^ swz(rb)) << 4) + sub));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = (bf16_t)a[i]; r[4 + i] = (bf16_t)b[i]; }
    return r;
}

__device__ __forceinline__ float xor16(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// ---------------------------------------------------------------------------------------------------
// forward: workgroup = 128 queries of one head (4 waves x 32 queries), key/value tiles of 64 through LDS
// ---------------------------------------------------------------------------------------------------
// HAZARD NOTE: the accumulators of an MFMA must only be read by compiler-generated instructions.  The first version
// took the row maximum with the inline-asm `vmax` helper: hipcc's hazard recogniser does not insert the wait states an
// MFMA result needs before an INLINE-ASM VALU reads it, so the maximum was occasionally taken from a not-yet-written
// register -- still a valid softmax shift (results differed by 1 bf16 ulp in whole rows), but not run-to-run
// reproducible.  `tools/debug_attn_det.py` checks bitwise reproducibility; the tests do as well.
// One key/value tile for one wave.  DIAG: the tile crosses the diagonal of this wave's queries (mask needed).
// dq = (query index of lane's column) - (first key of the tile) - 4*(lane>>4): key 16kb+i is masked iff 16kb+i > dq.
// GRID: the operands are integer grid coordinates; the S^T block is scaled by sk[key] (ksc[0..63]) here and by
// c2q[qb] = scale * log2e * sq[query] in the exponent, the probabilities by sv[key] (ksc[64..127]) and split into a bf16 high
// and low part (two MFMAs per block: 16 mantissa bits instead of 8 in front of the exact integer V operand).
// MID: called between the S^T MFMAs and the softmax (the kernel stores the next K tile and starts loading the next V tile there).
template <bool DIAG, bool GRID, typename MID>
__device__ __forceinline__ void fwd_tile(const char* Ks, const char* Vs, const bf16x8 (&qf)[2][4], f32x4 (&o)[2][8],
                                         float (&m)[2], float (&l)[2], const int (&dq)[2], const float (&c2q)[2],
                                         const float* ksc, int lane, MID&& mid) {
    f32x4 st[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        st[kb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[kb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // d-step outermost: the 8 accumulators (4 key blocks x 2 query blocks) are independent, so consecutive MFMAs never
    // wait on each other (a key-block-outer order chains 4 dependent MFMAs per accumulator back to back)
    bf16x8 kf[2];
    kf[0] = frag_row(Ks, 0, 0, lane);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int ks = it >> 2, kb = it & 3, nx = it + 1;
        if (nx < 16) kf[nx & 1] = frag_row(Ks, 16 * (nx & 3), nx >> 2, lane);
        st[kb][0] = MFMA(kf[it & 1], qf[0][ks], st[kb][0]);     // S^T block: row = key 16kb+4g+i, col = query
        st[kb][1] = MFMA(kf[it & 1], qf[1][ks], st[kb][1]);
    }
    mid();
    bf16x8 vf[2];
    vf[0] = frag_tr(Vs, 0, 0, lane);             // in flight under the softmax arithmetic
    if (GRID) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x4 skk = *reinterpret_cast<const f32x4*>(ksc + 16 * kb + 4 * (lane >> 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st[kb][0][i] *= skk[i];
                st[kb][1][i] *= skk[i];
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float c2 = c2q[qb];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (DIAG && 16 * kb + i > dq[qb]) st[kb][qb][i] = -INFINITY;
                mx = fmaxf(mx, st[kb][qb][i]);     // NOT the inline-asm vmax: see the hazard note above fwd_tile
            }
        mx = fmaxf(mx, xor16(mx));
        mx = fmaxf(mx, xor32(mx));
        mx = fmaxf(m[qb], mx * c2);               // running maximum in the scaled log2 domain
        const float alpha = __builtin_amdgcn_exp2f(m[qb] - mx);
        m[qb] = mx;
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[kb][qb][i], c2, -mx));
                st[kb][qb][i] = pv;
                ls += pv;
            }
        l[qb] = fmaf(l[qb], alpha, ls);
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {       // wave-uniform: most late tiles leave the maxima alone
#pragma unroll
            for (int db = 0; db < 8; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) o[qb][db][i] *= alpha;
        }
    }
    if (GRID) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x4 svk = *reinterpret_cast<const f32x4*>(ksc + 64 + 16 * kb + 4 * (lane >> 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st[kb][0][i] *= svk[i];
                st[kb][1][i] *= svk[i];
            }
        }
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        const bf16x8 p0 = pack8(st[2 * kc][0], st[2 * kc + 1][0]);
        const bf16x8 p1 = pack8(st[2 * kc][1], st[2 * kc + 1][1]);
        bf16x8 r0, r1;
        if (GRID) {                 // low parts: what the bf16 rounding of the high parts dropped
            f32x4 e[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e[h][0][i] = st[2 * kc + h][0][i] - (float)p0[4 * h + i];
                    e[h][1][i] = st[2 * kc + h][1][i] - (float)p1[4 * h + i];
                }
            r0 = pack8(e[0][0], e[1][0]);
            r1 = pack8(e[0][1], e[1][1]);
        }
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            const int nx = kc * 8 + db + 1;
            if (nx < 16) vf[nx & 1] = frag_tr(Vs, 32 * (nx >> 3), 16 * (nx & 7), lane);   // V^T: row = d 16db+(lane&15)
            o[0][db] = MFMA(vf[(nx - 1) & 1], p0, o[0][db]);        // O^T block: row = d 16db+4g+i, col = query
            o[1][db] = MFMA(vf[(nx - 1) & 1], p1, o[1][db]);
            if (GRID) {
                o[0][db] = MFMA(vf[(nx - 1) & 1], r0, o[0][db]);
                o[1][db] = MFMA(vf[(nx - 1) & 1], r1, o[1][db]);
            }
        }
    }
}

template <bool GRID>
__global__ void __launch_bounds__(256, 2) attn_fwd_kernel(AttnP p, int nqt, int nhb) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE + 2 * 128 * 4];   // [stage][K | V], [stage][sk | sv] (grid mode)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    // Work placement (speed only): items sorted heaviest-first (rank r: qt = nqt-1 - r/nhb) are dealt round-robin to
    // the 8 XCDs (id % 8), and inside an XCD the second half of the local sequence is reversed, so the two
    // workgroups that share a CU (local n and n + per/2 under round-robin dispatch) are a heavy and a light one.
    const int total = nqt * nhb;
    int r = blockIdx.x;
    if (total % 16 == 0) {
        const int per = total / 8, xcd = r & 7, n = r >> 3;
        const int j = n < per / 2 ? n : per - 1 - (n - per / 2);
        r = j * 8 + xcd;
    }
    const int qt = nqt - 1 - r / nhb, hb = r % nhb;
    const int h = hb % p.nh, b = hb / p.nh, hk = h / (p.nh / p.nkv);
    const int64_t ldq = (int64_t)p.nh * HD, ldk = (int64_t)p.nkv * HD;
    const int q0 = qt * 128, qw0 = q0 + 32 * w;
    const bf16_t* qp = p.q + ((int64_t)b * p.T) * ldq + (int64_t)h * HD;
    const bf16_t* kp = p.k + ((int64_t)b * p.T) * ldk + (int64_t)hk * HD;
    const bf16_t* vp = p.v + ((int64_t)b * p.T) * ldk + (int64_t)hk * HD;

    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + (int64_t)(qw0 + 16 * qb + c) * ldq + 32 * ks + 8 * g);

    f32x4 o[2][8];
    float m[2], l[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        m[qb] = -INFINITY;
        l[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < 8; ++db) o[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int ntiles = 2 * qt + 2;
    float* kscs = reinterpret_cast<float*>(smem + 4 * TILE);
    // grid mode: per-key scales of the tile (threads 0-63: sk, 64-127: sv); per-query exponent factors
    const float* kscp = nullptr;
    float c2q[2] = {p.scale_log2, p.scale_log2};
    if (GRID) {
        kscp = (tid < 64 ? p.sk : p.sv) + ((int64_t)b * p.T) * p.lds + hk;
        const float* sqp = p.sq + ((int64_t)b * p.T) * p.lds + h;
        c2q[0] *= sqp[(int64_t)(qw0 + c) * p.lds];
        c2q[1] *= sqp[(int64_t)(qw0 + 16 + c) * p.lds];
    }
    float rsc = 0.f;
    // staging registers are shared by K and V (16 instead of 32): K(j+1) is loaded before the S^T MFMAs of tile j and stored
    // behind them, V(j+1) is loaded then and stored behind the P V MFMAs
    u32x4 rr[4];
    tile_gload(kp, ldk, tid, rr);
    if (GRID && tid < 128) rsc = kscp[(int64_t)(tid & 63) * p.lds];
    tile_sstore(smem, tid, rr);
    tile_gload(vp, ldk, tid, rr);
    if (GRID && tid < 128) kscs[tid] = rsc;
    tile_sstore(smem + TILE, tid, rr);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): nothing (q fragments included) is pending at loop entry
    __syncthreads();
    for (int j = 0; j < ntiles; ++j) {
        const int cur = j & 1;
        const bool more = j + 1 < ntiles;
        if (more) {
            tile_gload(kp + (int64_t)(64 * (j + 1)) * ldk, ldk, tid, rr);
            if (GRID && tid < 128) rsc = kscp[(int64_t)(64 * (j + 1) + (tid & 63)) * p.lds];
        }
        const char* Ks = smem + cur * 2 * TILE;
        const char* Vs = Ks + TILE;
        char* nxt = smem + (cur ^ 1) * 2 * TILE;
        auto mid = [&]() {
            if (more) {
                tile_sstore(nxt, tid, rr);
                if (GRID && tid < 128) kscs[(cur ^ 1) * 128 + tid] = rsc;
                tile_gload(vp + (int64_t)(64 * (j + 1)) * ldk, ldk, tid, rr);
            }
        };
        const int kpos0 = 64 * j;
        if (kpos0 <= qw0 + 31) {     // wave-uniform: this wave has at least one unmasked (query, key) pair in the tile
            const int dq[2] = {qw0 + c - kpos0 - 4 * g, qw0 + 16 + c - kpos0 - 4 * g};
            if (kpos0 + 63 > qw0) fwd_tile<true, GRID>(Ks, Vs, qf, o, m, l, dq, c2q, kscs + cur * 128, lane, mid);
            else fwd_tile<false, GRID>(Ks, Vs, qf, o, m, l, dq, c2q, kscs + cur * 128, lane, mid);
        } else {
            mid();
        }
        if (more) tile_sstore(nxt + TILE, tid, rr);
        __syncthreads();
    }
    bf16_t* op = p.o + ((int64_t)b * p.T) * ldq + (int64_t)h * HD;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float ls = l[qb];
        ls += xor16(ls);
        ls += xor32(ls);
        const float inv = 1.0f / ls;
        const int qi = qw0 + 16 * qb + c;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
            bf4 ov;
#pragma unroll
            for (int i = 0; i < 4; ++i) ov[i] = (bf16_t)(o[qb][db][i] * inv);
            *reinterpret_cast<bf4*>(op + (int64_t)qi * ldq + 16 * db + 4 * g) = ov;
            if (GRID && p.o32) {
                const f32x4 of = {o[qb][db][0] * inv, o[qb][db][1] * inv, o[qb][db][2] * inv, o[qb][db][3] * inv};
                *reinterpret_cast<f32x4*>(p.o32 + ((int64_t)b * p.T + qi) * ldq + (int64_t)h * HD + 16 * db + 4 * g) = of;
            }
        }
        if (g == 0) p.lse[((int64_t)b * p.nh + h) * p.T + qi] = m[qb] + __builtin_amdgcn_logf(ls);   // v_log_f32 = log2
    }
}

// ---------------------------------------------------------------------------------------------------
// D[b,h,q] = sum_d O[q,h,d] * dO[q,h,d]   (one 16-lane group per (q,h) row of 128)
// ---------------------------------------------------------------------------------------------------
template <typename TO>        // TO: float = the un-rounded output of the grid forward (p.o32), else its bf16 copy
__global__ void __launch_bounds__(256) attn_bwd_prep_kernel(AttnP p, int64_t nrows) {
    const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);    // row index over [bs*T*nh]
    if (r >= nrows) return;
    const int c = threadIdx.x & 15;
    float a[8], bq[8];
    if (sizeof(TO) == 4) Vec8<float>::load(p.o32 + r * HD + c * 8, a);
    else Vec8<bf16_t>::load(p.o + r * HD + c * 8, a);
    Vec8<bf16_t>::load(p.go + r * HD + c * 8, bq);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s = fmaf(a[i], bq[i], s);
    s = wave_sum(s, 16);
    if (c == 0) {
        const int64_t bt = r / p.nh;          // b*T + q
        const int h = (int)(r - bt * p.nh);
        const int64_t b = bt / p.T, q = bt - b * p.T;
        const_cast<float*>(p.dsum)[(b * p.nh + h) * p.T + q] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// backward: workgroup = 64 keys of one head (4 waves x 16 keys); loops over the 64-query tiles at/below the diagonal
// ---------------------------------------------------------------------------------------------------
// One 64-query tile for one wave (16 keys).  DIAG: the diagonal tile; query 16qb+4g+i is masked iff 16qb+i < mk
// (mk = key - first query of the tile - 4g, per lane).
// GRID: q / k / v are grid coordinates.  c2 then carries scale*log2e*sk[key] and `scale` carries scale (both per lane: the
// lane's column is its key), skk = sk[key], svk = sv[key], sq_s = the tile's 64 query scales: the score is s*sq[query]*c2,
// dP = dp*svk, the stored dS^T (operand of the dQ GEMM against the K coordinates) is scaled by sk[key] and the dK operand
// by sq[query].
template <bool DIAG, bool GRID>
__device__ __forceinline__ void bwd_phase1(const char* Qs, const char* Gs, const float* lse_s, const float* d_s,
                                           const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], bf16x8 (&pf)[2], bf16x8 (&sf)[2],
                                           bf16_t* dst_row, int mk, float c2, float scale, float skk, float svk,
                                           const float* sq_s, int lane) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
    const int g = lane >> 4;
    f32x4 pprev = {0.f, 0.f, 0.f, 0.f}, sprev = {0.f, 0.f, 0.f, 0.f};
    bf16x8 qa[4], ga[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qa[ks] = frag_row(Qs, 16 * qb, ks, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ga[ks] = frag_row(Gs, 16 * qb, ks, lane);
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * qb + 4 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(d_s + 16 * qb + 4 * g);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {        // the two accumulation chains alternate: no back-to-back dependent MFMAs
            s = MFMA(qa[ks], kf[ks], s);        // row = query 16qb+4g+i, col = key
            dp = MFMA(ga[ks], vf[ks], dp);
        }
        f32x4 pv, sv, q4 = {1.f, 1.f, 1.f, 1.f};
        if (GRID) q4 = *reinterpret_cast<const f32x4*>(sq_s + 16 * qb + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float e = __builtin_amdgcn_exp2f(fmaf(GRID ? s[i] * q4[i] : s[i], c2, -l4[i]));
            if (DIAG && 16 * qb + i < mk) e = 0.f;
            pv[i] = e;
            sv[i] = e * ((GRID ? dp[i] * svk : dp[i]) - d4[i]) * scale;
        }
        bf4 dsv;
#pragma unroll
        for (int i = 0; i < 4; ++i) dsv[i] = (bf16_t)(GRID ? sv[i] * skk : sv[i]);
        *reinterpret_cast<bf4*>(dst_row + 16 * qb) = dsv;           // dS^T[key][query 16qb+4g .. +4]
        if (GRID) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] *= q4[i];
        }
        if (qb & 1) {
            pf[qb >> 1] = pack8(pprev, pv);                        // B operand: k-slot = query, col = key
            sf[qb >> 1] = pack8(sprev, sv);
        } else {
            pprev = pv;
            sprev = sv;
        }
    }
}

// dV^T += dO^T P, dK^T += Q^T dS for one 64-query tile (the contraction runs over the queries)
__device__ __forceinline__ void bwd_phase2(const char* Qs, const char* Gs, const bf16x8 (&pf)[2], const bf16x8 (&sf)[2],
                                           f32x4 (&dk)[8], f32x4 (&dv)[8], int lane) {
    bf16x8 tg[2], tq[2];
    tg[0] = frag_tr(Gs, 0, 0, lane);
    tq[0] = frag_tr(Qs, 0, 0, lane);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int kc = it >> 3, db = it & 7, nx = it + 1;
        if (nx < 16) {
            tg[nx & 1] = frag_tr(Gs, 32 * (nx >> 3), 16 * (nx & 7), lane);
            tq[nx & 1] = frag_tr(Qs, 32 * (nx >> 3), 16 * (nx & 7), lane);
        }
        dv[db] = MFMA(tg[it & 1], pf[kc], dv[db]);     // dV^T: row = d 16db+4g+i, col = key
        dk[db] = MFMA(tq[it & 1], sf[kc], dk[db]);
    }
}

template <bool GRID>
__global__ void __launch_bounds__(256, 2) attn_bwd_kernel(AttnP p, int nhb) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE + 2 * 3 * 64 * 4];   // [stage][Q | dO], [stage][lse | D | sq]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    // heaviest key tiles (kvt = 0: all query tiles) first; consecutive ids = different heads of the same key tile, so
    // the round-robin id -> XCD placement gives every XCD the same mix
    const int kvt = (int)blockIdx.x / nhb, hb = (int)blockIdx.x % nhb;
    const int h = hb % p.nh, b = hb / p.nh, hk = h / (p.nh / p.nkv);
    const int64_t ldq = (int64_t)p.nh * HD, ldk = (int64_t)p.nkv * HD;
    const int kv0 = kvt * 64, kw0 = kv0 + 16 * w;
    const int nqt = (int)(p.T / 64);
    const bf16_t* qp = p.q + ((int64_t)b * p.T) * ldq + (int64_t)h * HD;
    const bf16_t* gop = p.go + ((int64_t)b * p.T) * ldq + (int64_t)h * HD;
    const bf16_t* kp = p.k + ((int64_t)b * p.T) * ldk + (int64_t)hk * HD;
    const bf16_t* vp = p.v + ((int64_t)b * p.T) * ldk + (int64_t)hk * HD;
    const float* lsep = p.lse + ((int64_t)b * p.nh + h) * p.T;
    const float* dsp = p.dsum + ((int64_t)b * p.nh + h) * p.T;
    bf16_t* dstp = p.dst + (((int64_t)b * p.nh + h) * p.T) * p.T;       // [key][query]
    float* stat = reinterpret_cast<float*>(smem + 4 * TILE);

    // zero the part of dS^T the dQ GEMM reads inside the diagonal 256-block but no (key tile, query tile) pair writes
    {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
        const bf4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        for (int qt = kvt & ~3; qt < kvt; ++qt)
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                const int row = (tid >> 4) + 16 * rep, col = (tid & 15) * 4;
                *reinterpret_cast<bf4*>(dstp + (int64_t)(kv0 + row) * p.T + qt * 64 + col) = z;
            }
    }
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(kp + (int64_t)(kw0 + c) * ldk + 32 * ks + 8 * g);
        vf[ks] = *reinterpret_cast<const bf16x8*>(vp + (int64_t)(kw0 + c) * ldk + 32 * ks + 8 * g);
    }
    f32x4 dk[8], dv[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) {
        dk[db] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 rq[4];
    float rs = 0.f;
    // grid mode: per-lane key scales, per-tile query scales (threads 128-191 stage them next to lse / D)
    float c2 = p.scale_log2, skk = 1.f, svk = 1.f;
    const float* sqp = nullptr;
    if (GRID) {
        skk = p.sk[((int64_t)b * p.T + kw0 + c) * p.lds + hk];
        svk = p.sv[((int64_t)b * p.T + kw0 + c) * p.lds + hk];
        c2 *= skk;
        sqp = p.sq + ((int64_t)b * p.T) * p.lds + h;
    }
    const int nstat = GRID ? 192 : 128;
    auto stat_load = [&](int64_t q0) {
        return tid < 64 ? lsep[q0 + tid] : tid < 128 ? dsp[q0 + tid - 64] : sqp[(q0 + tid - 128) * p.lds];
    };
    {
        u32x4 rg[4];
        const int64_t q0 = (int64_t)kvt * 64;
        tile_gload(qp + q0 * ldq, ldq, tid, rq);
        tile_gload(gop + q0 * ldq, ldq, tid, rg);
        if (tid < nstat) rs = stat_load(q0);
        tile_sstore(smem, tid, rq);
        tile_sstore(smem + TILE, tid, rg);
        if (tid < nstat) stat[tid] = rs;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): k/v fragments and the zero-fill stores are done
    __syncthreads();
    bf16_t* dst_row = dstp + (int64_t)(kw0 + c) * p.T + 4 * g;
    for (int qt = kvt; qt < nqt; ++qt) {
        const int cur = (qt - kvt) & 1;
        const bool more = qt + 1 < nqt;
        const int64_t q1 = (int64_t)(qt + 1) * 64;
        if (more) {            // staging in two halves (Q under phase 1, dO under phase 2): 16 registers instead of 32
            tile_gload(qp + q1 * ldq, ldq, tid, rq);
            if (tid < nstat) rs = stat_load(q1);
        }
        const char* Qs = smem + cur * 2 * TILE;
        const char* Gs = Qs + TILE;
        const float* lse_s = stat + cur * 192;
        bf16x8 pf[2], sf[2];
        if (qt == kvt)
            bwd_phase1<true, GRID>(Qs, Gs, lse_s, lse_s + 64, kf, vf, pf, sf, dst_row + qt * 64, 16 * w + c - 4 * g,
                                   c2, p.scale, skk, svk, lse_s + 128, lane);
        else
            bwd_phase1<false, GRID>(Qs, Gs, lse_s, lse_s + 64, kf, vf, pf, sf, dst_row + qt * 64, 0, c2, p.scale, skk, svk,
                                    lse_s + 128, lane);
        if (more) {
            tile_sstore(smem + (cur ^ 1) * 2 * TILE, tid, rq);
            if (tid < nstat) stat[(cur ^ 1) * 192 + tid] = rs;
            tile_gload(gop + q1 * ldq, ldq, tid, rq);
        }
        bwd_phase2(Qs, Gs, pf, sf, dk, dv, lane);
        if (more) tile_sstore(smem + (cur ^ 1) * 2 * TILE + TILE, tid, rq);
        __syncthreads();
    }
    bf16_t* gkp = p.gk + ((int64_t)b * p.T) * ldq + (int64_t)h * HD;
    bf16_t* gvp = p.gv + ((int64_t)b * p.T) * ldq + (int64_t)h * HD;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
        bf4 a, bb;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = (bf16_t)dk[db][i]; bb[i] = (bf16_t)dv[db][i]; }
        *reinterpret_cast<bf4*>(gkp + (int64_t)(kw0 + c) * ldq + 16 * db + 4 * g) = a;
        *reinterpret_cast<bf4*>(gvp + (int64_t)(kw0 + c) * ldq + 16 * db + 4 * g) = bb;
    }
}

int check_common(const char* who, int dtype, int64_t bs, int64_t T, int nh, int nkv, int hd, int causal) {
    OQ_CHECK_ARG(dtype == OQ_BF16, "%s: bf16 only (the fp32 parity mode uses the unfused exact kernels)", who);
    OQ_CHECK_ARG(hd == HD, "%s: head_dim %d unsupported (128 only)", who, hd);
    OQ_CHECK_ARG(causal == 1, "%s: only the exact causal mask is fused", who);
    OQ_CHECK_ARG(bs > 0 && bs <= 65535 && (T == 128 || (T > 0 && T % 256 == 0)),
                 "%s: T=%lld must be 128 or a multiple of 256 (the causal dQ GEMM contracts in 256-blocks)", who, (long long)T);
    OQ_CHECK_ARG(nh > 0 && nkv > 0 && nh % nkv == 0 && nh <= 65535, "%s: heads %d / kv heads %d", who, nh, nkv);
    return OQ_OK;
}

}  // namespace

extern "C" int64_t oq_attn_supported(int dtype, int64_t T, int hd, int causal) {
    return dtype == OQ_BF16 && hd == HD && causal == 1 && (T == 128 || (T > 0 && T % 256 == 0));
}

extern "C" int oq_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int64_t bs,
                           int64_t T, int nh, int nkv, int hd, float scale, int causal, void* stream) {
    OQ_CHECK_ARG(q && k && v && o && lse, "oq_attn_fwd: null pointer");
    if (int rc = check_common("oq_attn_fwd", dtype, bs, T, nh, nkv, hd, causal)) return rc;
    OQ_CHECK_ARG(oq_aligned16(q) && oq_aligned16(k) && oq_aligned16(v) && oq_aligned16(o), "oq_attn_fwd: 16-B alignment");
    AttnP p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
    p.T = T; p.nh = nh; p.nkv = nkv; p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
    const int nqt = (int)(T / 128);
    OQ_CHECK_ARG((int64_t)nqt * nh * bs < (1ll << 30), "oq_attn_fwd: too many workgroups");
    hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3((unsigned)(nqt * nh * bs)), dim3(256), 0, (hipStream_t)stream, p, nqt,
                       (int)(nh * bs));
    OQ_CHECK_LAUNCH("oq_attn_fwd");
    return OQ_OK;
}

extern "C" int oq_attn_fwd_grid(const void* nq, const void* nk, const void* nv, const float* sq, const float* sk,
                                const float* sv, int64_t ld_s, void* o, float* o32, float* lse, int64_t bs, int64_t T, int nh,
                                int nkv, int hd, float scale, int causal, void* stream) {
    OQ_CHECK_ARG(nq && nk && nv && sq && sk && sv && o && lse, "oq_attn_fwd_grid: null pointer");
    if (int rc = check_common("oq_attn_fwd_grid", OQ_BF16, bs, T, nh, nkv, hd, causal)) return rc;
    OQ_CHECK_ARG(ld_s >= nh, "oq_attn_fwd_grid: scale row stride %lld < %d heads", (long long)ld_s, nh);
    OQ_CHECK_ARG(oq_aligned16(nq) && oq_aligned16(nk) && oq_aligned16(nv) && oq_aligned16(o) && oq_aligned16(o32),
                 "oq_attn_fwd_grid: 16-B alignment");
    AttnP p{};
    p.q = (const bf16_t*)nq; p.k = (const bf16_t*)nk; p.v = (const bf16_t*)nv; p.o = (bf16_t*)o; p.o32 = o32; p.lse = lse;
    p.sq = sq; p.sk = sk; p.sv = sv; p.lds = ld_s;
    p.T = T; p.nh = nh; p.nkv = nkv; p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
    const int nqt = (int)(T / 128);
    OQ_CHECK_ARG((int64_t)nqt * nh * bs < (1ll << 30), "oq_attn_fwd_grid: too many workgroups");
    hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3((unsigned)(nqt * nh * bs)), dim3(256), 0, (hipStream_t)stream, p, nqt,
                       (int)(nh * bs));
    OQ_CHECK_LAUNCH("oq_attn_fwd_grid");
    return OQ_OK;
}

extern "C" int oq_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* go, const float* lse,
                           float* dsum, void* ds_t, void* gk, void* gv, int dtype, int64_t bs, int64_t T, int nh, int nkv,
                           int hd, float scale, int causal, void* stream) {
    OQ_CHECK_ARG(q && k && v && o && go && lse && dsum && ds_t && gk && gv, "oq_attn_bwd: null pointer");
    if (int rc = check_common("oq_attn_bwd", dtype, bs, T, nh, nkv, hd, causal)) return rc;
    OQ_CHECK_ARG(oq_aligned16(q) && oq_aligned16(k) && oq_aligned16(v) && oq_aligned16(o) && oq_aligned16(go) &&
                     oq_aligned16(ds_t) && oq_aligned16(gk) && oq_aligned16(gv) && oq_aligned16(lse) && oq_aligned16(dsum),
                 "oq_attn_bwd: 16-B alignment");
    AttnP p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)const_cast<void*>(o);
    p.go = (const bf16_t*)go; p.lse = const_cast<float*>(lse); p.dsum = dsum; p.dst = (bf16_t*)ds_t;
    p.gk = (bf16_t*)gk; p.gv = (bf16_t*)gv;
    p.T = T; p.nh = nh; p.nkv = nkv; p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
    const int64_t nrows = bs * T * nh;
    hipLaunchKernelGGL(attn_bwd_prep_kernel<bf16_t>, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, p,
                       nrows);
    OQ_CHECK_ARG((T / 64) * nh * bs < (1ll << 30), "oq_attn_bwd: too many workgroups");
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3((unsigned)((T / 64) * nh * bs)), dim3(256), 0, (hipStream_t)stream, p,
                       (int)(nh * bs));
    OQ_CHECK_LAUNCH("oq_attn_bwd");
    return OQ_OK;
}

extern "C" int oq_attn_bwd_grid(const void* nq, const void* nk, const void* nv, const float* sq, const float* sk,
                                const float* sv, int64_t ld_s, const void* o, const float* o32, const void* go, const float* lse,
                                float* dsum, void* ds_t, void* gk, void* gv, int64_t bs, int64_t T, int nh, int nkv, int hd,
                                float scale, int causal, void* stream) {
    OQ_CHECK_ARG(nq && nk && nv && sq && sk && sv && o && go && lse && dsum && ds_t && gk && gv, "oq_attn_bwd_grid: null pointer");
    if (int rc = check_common("oq_attn_bwd_grid", OQ_BF16, bs, T, nh, nkv, hd, causal)) return rc;
    OQ_CHECK_ARG(ld_s >= nh, "oq_attn_bwd_grid: scale row stride %lld < %d heads", (long long)ld_s, nh);
    OQ_CHECK_ARG(oq_aligned16(nq) && oq_aligned16(nk) && oq_aligned16(nv) && oq_aligned16(o) && oq_aligned16(go) &&
                     oq_aligned16(ds_t) && oq_aligned16(gk) && oq_aligned16(gv) && oq_aligned16(lse) && oq_aligned16(dsum) &&
                     oq_aligned16(o32),
                 "oq_attn_bwd_grid: 16-B alignment");
    AttnP p{};
    p.q = (const bf16_t*)nq; p.k = (const bf16_t*)nk; p.v = (const bf16_t*)nv; p.o = (bf16_t*)const_cast<void*>(o);
    p.go = (const bf16_t*)go; p.lse = const_cast<float*>(lse); p.dsum = dsum; p.dst = (bf16_t*)ds_t;
    p.gk = (bf16_t*)gk; p.gv = (bf16_t*)gv;
    p.sq = sq; p.sk = sk; p.sv = sv; p.lds = ld_s; p.o32 = const_cast<float*>(o32);
    p.T = T; p.nh = nh; p.nkv = nkv; p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
    const int64_t nrows = bs * T * nh;
    if (o32)
        hipLaunchKernelGGL(attn_bwd_prep_kernel<float>, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, p,
                           nrows);
    else
        hipLaunchKernelGGL(attn_bwd_prep_kernel<bf16_t>, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                           p, nrows);
    OQ_CHECK_ARG((T / 64) * nh * bs < (1ll << 30), "oq_attn_bwd_grid: too many workgroups");
    hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3((unsigned)((T / 64) * nh * bs)), dim3(256), 0, (hipStream_t)stream, p,
                       (int)(nh * bs));
    OQ_CHECK_LAUNCH("oq_attn_bwd_grid");
    return OQ_OK;
}
