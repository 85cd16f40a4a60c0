// Fused ("flash") attention for the Perceiver blocks on the bf16 matrix cores ('bf16' and 'bf16x3' precisions).
// Replaces, without ever materialising the [B*h, i, j] score tensor (perceiver_lang_io.py:116-130):
//     sim = q k^T * scale ; attn = softmax(sim) ; attn = dropout(attn) ; out = attn v
// for the three attention shapes of the model: cross (2048 x 8077, 1 head), self (2048 x 2048, 8 heads), decoder
// (8077 x 2048, 1 head); head dim 64.  q / k / v are read in place from the fp32 projection outputs
// (q [B, Nq, H*64], kv [B, Nk, 2*H*64]) and rounded to bf16 while being staged; softmax statistics, accumulators and
// outputs are fp32.
//
// CDNA4 mapping (wave64, v_mfma_f32_32x32x16_bf16), everything in the "swapped" form so that the softmax is lane-local:
//   S^T[key][q] = K Q^T   : A = K rows from LDS (ds_read_b128), B = Q^T fragments held in registers for the whole loop
//                           -> lane (q = l & 31) owns one query row: 16 keys per 32-key block, its partner lane l ^ 32 the
//                              other 16; row max / sum need ONE cross-lane exchange per tile
//   O^T[d][q]  += V^T P^T : A = V^T via ds_read_b64_tr_b16 (V stays [key][d] in LDS, transposed by the read),
//                           B = P^T = the S^T accumulator registers themselves, packed to bf16 -- no shuffles, no LDS
//                           round trip for P (any k-permutation inside one MFMA cancels as long as A and B agree)
// One workgroup = 4 waves x 32 queries; K/V tiles of 64 keys, register-prefetched one tile ahead.
// Dropout uses a counter-based hash of (seed, row, key pair): the backward kernels regenerate the same mask.
//
// X3 = 1 ('bf16x3'): every matrix-core operand -- q, k, v, dO and also the probabilities P and score gradients dS -- is
// carried as hi = bf16(a), lo = bf16(a - hi) and each product is evaluated as hi*hi + hi*lo + lo*hi (fp32 accumulate),
// which keeps the fused kernels inside the 1e-4 Q-value bound of the exact-fp32 attention path.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int HD = 64;          // head dim
constexpr int BQ = 128;         // queries per workgroup (4 waves x 32)
constexpr int BKV = 64;         // keys per tile
constexpr int LDK = 72;         // Ks row stride in bf16 (36 dwords: conflict-free ds_read_b128)
constexpr int LDV = 96;         // Vs row stride in bf16 (48 dwords: conflict-free ds_read_b64_tr_b16)
constexpr int KPL = BKV * LDK;  // u16 per b128-layout plane
constexpr int VPL = BKV * LDV;  // u16 per transposed-read-layout plane
constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
    const float* q;      // [B, Nq, H*64]
    const float* kv;     // [B, Nk, 2*H*64]
    float* o;            // [B, Nq, H*64]
    float* lse;          // [B*H, Nq]   natural-log sum-exp of the scaled scores
    int B, H, Nq, Nk;
    float scale, p_drop;
    unsigned seed;
};

__device__ __forceinline__ unsigned fa_hash(unsigned x) {
    // one multiply round: the inputs are already products with odd constants; keep-rate, row / column sums and lag
    // correlations of the 16-bit halves are indistinguishable from the two-round finaliser (checked offline on 4096 x 2048
    // masks), and every hash costs a quarter-rate multiply less in the softmax loops
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
    return x;
}
// keep decisions for keys (2c, 2c+1) of row `row`: low / high 16 bits of one hash against thr = p * 65536
__device__ __forceinline__ unsigned fa_keep_pair(unsigned seed, unsigned row, unsigned colpair) {
    return fa_hash((row * 0x9E3779B1U + seed) ^ (colpair * 0x85EBCA77U + 0xC2B2AE3DU));
}
__device__ __forceinline__ unsigned fa_pack2(float lo, float hi) { return vxb_pack_bf16(lo, hi); }
// hi/lo split of a value pair: ph = bf16 pair, pl = bf16 pair of the (exact) residuals
__device__ __forceinline__ void fa_split2(float a, float b, unsigned& ph, unsigned& pl) {
    ph = fa_pack2(a, b);
    pl = fa_pack2(a - __uint_as_float(ph << 16), b - __uint_as_float(ph & 0xffff0000u));
}
__device__ __forceinline__ unsigned long long fa_tr16(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// acc += A B with A = ah + al, B = bh + bl (X3) or A = ah, B = bh
template <int X3>
__device__ __forceinline__ f32x16 fa_mma(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 c) {
    if (X3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}
// 8 consecutive fp32 (two float4) -> one bf16x8 fragment (+ residual fragment), optionally pre-scaled
template <int X3>
__device__ __forceinline__ void fa_frag8(const float* p, float s, bf16x8& fh, bf16x8& fl) {
    union { unsigned u[4]; bf16x8 v; } h, l;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 c = *reinterpret_cast<const float4*>(p + 4);
    if (X3) {
        fa_split2(a.x * s, a.y * s, h.u[0], l.u[0]); fa_split2(a.z * s, a.w * s, h.u[1], l.u[1]);
        fa_split2(c.x * s, c.y * s, h.u[2], l.u[2]); fa_split2(c.z * s, c.w * s, h.u[3], l.u[3]);
        fl = l.v;
    } else {
        h.u[0] = fa_pack2(a.x * s, a.y * s); h.u[1] = fa_pack2(a.z * s, a.w * s);
        h.u[2] = fa_pack2(c.x * s, c.y * s); h.u[3] = fa_pack2(c.z * s, c.w * s);
        fl = h.v;
    }
    fh = h.v;
}
// one float4 -> 8-byte bf16 store into a plane (and its residual into the lo plane PL u16 further)
template <int X3>
__device__ __forceinline__ void fa_store4(u16* dst, int PL, const float4 v, float s) {
    uint2 ph, pl;
    if (X3) {
        fa_split2(v.x * s, v.y * s, ph.x, pl.x); fa_split2(v.z * s, v.w * s, ph.y, pl.y);
        *reinterpret_cast<uint2*>(dst + PL) = pl;
    } else {
        ph.x = fa_pack2(v.x * s, v.y * s); ph.y = fa_pack2(v.z * s, v.w * s);
    }
    *reinterpret_cast<uint2*>(dst) = ph;
}
__device__ __forceinline__ bf16x8 fa_join(unsigned long long a, unsigned long long b) {
    union { unsigned long long u[2]; bf16x8 v; } t;
    t.u[0] = a; t.u[1] = b;
    return t.v;
}
__device__ __forceinline__ bf16x8 fa_from4(const unsigned* p) {
    union { unsigned u[4]; bf16x8 v; } t;
    t.u[0] = p[0]; t.u[1] = p[1]; t.u[2] = p[2]; t.u[3] = p[3];
    return t.v;
}

template <int X3>
__global__ void __launch_bounds__(256) flash_fwd_kernel(AttnArgs g) {
    __shared__ __attribute__((aligned(16))) u16 Ks[(1 + X3) * KPL];
    __shared__ __attribute__((aligned(16))) u16 Vs[(1 + X3) * VPL];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hi = lane >> 5, lq = lane & 31;
    const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int q0 = blockIdx.x * BQ + wid * 32;
    const int qrow = q0 + lq;                                  // this lane's query
    const bool q_ok = qrow < g.Nq;
    const float* qp = g.q + ((long long)b * g.Nq + (q_ok ? qrow : 0)) * inner + h * HD;
    const float* kbase = g.kv + (long long)b * g.Nk * 2 * inner + h * HD;
    const float* vbase = kbase + inner;
    const float qs = g.scale * LOG2E;                          // scores live in the log2 domain

    // Q^T fragments: lane (q, hi) holds q[16 ks + 8 hi .. +8] for ks = 0..3
    bf16x8 qfh[4], qfl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa_frag8<X3>(qp + 16 * ks + 8 * hi, qs, qfh[ks], qfl[ks]);
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging: tile = 64 keys x 64 d (K and V): 1024 float4 each -> 4 + 4 per thread; row = e / 16, col4 = (e % 16) * 4
    float4 rk[4], rv[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int key = kt * BKV + (e >> 4);
            const int c4 = (e & 15) * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
            if (key < g.Nk) {
                a = *reinterpret_cast<const float4*>(kbase + (long long)key * 2 * inner + c4);
                c = *reinterpret_cast<const float4*>(vbase + (long long)key * 2 * inner + c4);
            }
            rk[i] = a; rv[i] = c;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int r = e >> 4, c4 = (e & 15) * 4;
            fa_store4<X3>(&Ks[r * LDK + c4], KPL, rk[i], 1.f);
            fa_store4<X3>(&Vs[r * LDV + c4], VPL, rv[i], 1.f);
        }
    };

    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const float keep_scale = 1.0f / (1.0f - g.p_drop);
    const unsigned row_id = (unsigned)bh * (unsigned)g.Nq + (unsigned)qrow;
    // V^T transposed-read addressing: group gq = lane >> 4 (gq & 1: which 16 d-columns, gq >> 1 = hi), t = lane & 15
    const int t16 = lane & 15, gq = lane >> 4;
    const unsigned v_base = (unsigned)(size_t)(&Vs[0]) + 2u * (unsigned)((4 * (gq >> 1) + (t16 >> 2)) * LDV + 16 * (gq & 1) + 4 * (t16 & 3));

    const int nkt = (g.Nk + BKV - 1) / BKV;
    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);

        // ---- S^T = K Q^T for the two 32-key blocks
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u16* kp = &Ks[(kb * 32 + lq) * LDK + 16 * ks + 8 * hi];
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kp);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(kp + X3 * KPL);
                sacc[kb] = fa_mma<X3>(ah, al, qfh[ks], qfl[ks], sacc[kb]);
            }
        }
        // sacc[kb][r] = score(key = kt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*hi, q = this lane's query), log2 domain
        const int kbase_t = kt * BKV;
        float mt = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase_t + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= g.Nk) sacc[kb][r] = -INFINITY;
                mt = fmaxf(mt, sacc[kb][r]);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp2f(m_run - m_new);            // 0 on the first tile (m_run = -inf)
        m_run = m_new;
        float psum = 0.f;
        unsigned pbh[2][8], pbl[2][8];                         // P^T fragments (bf16 pairs): [key block][4 dwords x 2 k-steps]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float p0 = exp2f(sacc[kb][r] - m_new), p1 = exp2f(sacc[kb][r + 1] - m_new);
                psum += p0 + p1;
                if (thr > 0u) {
                    const unsigned key = (unsigned)(kbase_t + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                    const unsigned hsh = fa_keep_pair(g.seed, row_id, key >> 1);
                    p0 = (hsh & 0xffffu) >= thr ? p0 * keep_scale : 0.f;
                    p1 = (hsh >> 16) >= thr ? p1 * keep_scale : 0.f;
                }
                if (X3) fa_split2(p0, p1, pbh[kb][r >> 1], pbl[kb][r >> 1]);
                else pbh[kb][r >> 1] = fa_pack2(p0, p1);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        // ---- O^T += V^T P^T : 2 d-blocks x (2 key blocks x 2 k-steps of 16 keys)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pfh = fa_from4(&pbh[kb][4 * ks]);
                const bf16x8 pfl = X3 ? fa_from4(&pbl[kb][4 * ks]) : pfh;
                unsigned long long va[2][2], vl[2][2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned ad = v_base + 2u * (unsigned)((kb * 32 + 16 * ks) * LDV + db * 32);
                    va[db][0] = fa_tr16(ad);
                    va[db][1] = fa_tr16(ad + 2u * (unsigned)(8 * LDV));
                    if (X3) {
                        vl[db][0] = fa_tr16(ad + 2u * (unsigned)VPL);
                        vl[db][1] = fa_tr16(ad + 2u * (unsigned)(VPL + 8 * LDV));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vfh = fa_join(va[db][0], va[db][1]);
                    const bf16x8 vfl = X3 ? fa_join(vl[db][0], vl[db][1]) : vfh;
                    oacc[db] = fa_mma<X3>(vfh, vfl, pfh, pfl, oacc[db]);
                }
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (q_ok) {
        const float inv = 1.0f / l_tot;
        float* op = g.o + ((long long)b * g.Nq + qrow) * inner + h * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float4 v;
                v.x = oacc[db][4 * r4 + 0] * inv; v.y = oacc[db][4 * r4 + 1] * inv;
                v.z = oacc[db][4 * r4 + 2] * inv; v.w = oacc[db][4 * r4 + 3] * inv;
                *reinterpret_cast<float4*>(op + db * 32 + 8 * r4 + 4 * hi) = v;
            }
        if (hi == 0) g.lse[(long long)bh * g.Nq + qrow] = (m_run + log2f(l_tot)) * (1.0f / LOG2E);
    }
}


// ------------------------------------------------------------------------------------------------ backward
struct AttnBwdArgs {
    const float* q;      // [B, Nq, H*64]
    const float* kv;     // [B, Nk, 2*H*64]
    const float* d_o;    // [B, Nq, H*64]
    const float* lse;    // [B*H, Nq]
    const float* dsum;   // [B*H, Nq]   rowsum(dO * O)
    float* dq;           // [B, Nq, H*64]
    float* dkv;          // [B, Nk, 2*H*64]
    int B, H, Nq, Nk;
    float scale, p_drop;
    unsigned seed;
};

// D[bh][q] = sum_d dO[q, h*64 + d] * O[q, h*64 + d]     (one 16-lane group per (row, head))
__global__ void __launch_bounds__(256) flash_rowdot_kernel(const float* __restrict__ d_o, const float* __restrict__ o,
                                                           float* __restrict__ dsum, int B, int H, int Nq) {
    const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;      // (b, q, h)
    const int sub = threadIdx.x & 15;
    const long long total = (long long)B * Nq * H;
    float s = 0.f;
    if (idx < total) {
        const float4 a = *reinterpret_cast<const float4*>(d_o + idx * HD + sub * 4);
        const float4 c = *reinterpret_cast<const float4*>(o + idx * HD + sub * 4);
        s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (idx < total && sub == 0) {
        const int h = (int)(idx % H);
        const long long bq = idx / H;
        const int qq = (int)(bq % Nq);
        const int b = (int)(bq / Nq);
        dsum[((long long)b * H + h) * Nq + qq] = s;
    }
}

// dQ: same swapped structure as the forward pass (lane = query).  Per K/V tile:
//   S^T = K Q^T (K via ds_read_b128), P = exp2(S^T - lse);  dP^T = V dO^T (V via ds_read_b128);
//   dS^T = scale * P * (dP * keep/(1-p) - D);  dQ^T += K^T dS^T (K^T via ds_read_b64_tr_b16, dS^T packed from registers)
template <int X3>
__global__ void __launch_bounds__(256) flash_bwd_dq_kernel(AttnBwdArgs g) {
    __shared__ __attribute__((aligned(16))) u16 Kb[(1 + X3) * KPL];     // K, b128-friendly stride
    __shared__ __attribute__((aligned(16))) u16 Kt[(1 + X3) * VPL];     // K, transposed-read-friendly stride
    __shared__ __attribute__((aligned(16))) u16 Vb[(1 + X3) * KPL];     // V, b128-friendly stride
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hi = lane >> 5, lq = lane & 31;
    const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int qrow = blockIdx.x * BQ + wid * 32 + lq;
    const bool q_ok = qrow < g.Nq;
    const long long qoff = ((long long)b * g.Nq + (q_ok ? qrow : 0)) * inner + h * HD;
    const float* kbase = g.kv + (long long)b * g.Nk * 2 * inner + h * HD;
    const float* vbase = kbase + inner;
    const float qs = g.scale * LOG2E;
    bf16x8 qfh[4], qfl[4], dofh[4], dofl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa_frag8<X3>(g.q + qoff + 16 * ks + 8 * hi, qs, qfh[ks], qfl[ks]);
        fa_frag8<X3>(g.d_o + qoff + 16 * ks + 8 * hi, 1.f, dofh[ks], dofl[ks]);
    }
    const float lse2 = q_ok ? g.lse[(long long)bh * g.Nq + qrow] * LOG2E : 0.f;
    const float dsum = q_ok ? g.dsum[(long long)bh * g.Nq + qrow] : 0.f;
    f32x16 dqacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

    float4 rk[4], rv[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int key = kt * BKV + (e >> 4);
            const int c4 = (e & 15) * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
            if (key < g.Nk) {
                a = *reinterpret_cast<const float4*>(kbase + (long long)key * 2 * inner + c4);
                c = *reinterpret_cast<const float4*>(vbase + (long long)key * 2 * inner + c4);
            }
            rk[i] = a; rv[i] = c;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int r = e >> 4, c4 = (e & 15) * 4;
            fa_store4<X3>(&Kb[r * LDK + c4], KPL, rk[i], 1.f);
            fa_store4<X3>(&Kt[r * LDV + c4], VPL, rk[i], 1.f);
            fa_store4<X3>(&Vb[r * LDK + c4], KPL, rv[i], 1.f);
        }
    };
    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const float keep_scale = 1.0f / (1.0f - g.p_drop);
    const unsigned row_id = (unsigned)bh * (unsigned)g.Nq + (unsigned)qrow;
    const int t16 = lane & 15, gq = lane >> 4;
    const unsigned k_base = (unsigned)(size_t)(&Kt[0]) + 2u * (unsigned)((4 * (gq >> 1) + (t16 >> 2)) * LDV + 16 * (gq & 1) + 4 * (t16 & 3));

    const int nkt = (g.Nk + BKV - 1) / BKV;
    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);
        f32x16 sacc[2], pacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[kb][r] = 0.f; pacc[kb][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u16* kp = &Kb[(kb * 32 + lq) * LDK + 16 * ks + 8 * hi];
                const u16* vp = &Vb[(kb * 32 + lq) * LDK + 16 * ks + 8 * hi];
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kp), al = *reinterpret_cast<const bf16x8*>(kp + X3 * KPL);
                const bf16x8 ch = *reinterpret_cast<const bf16x8*>(vp), cl = *reinterpret_cast<const bf16x8*>(vp + X3 * KPL);
                sacc[kb] = fa_mma<X3>(ah, al, qfh[ks], qfl[ks], sacc[kb]);
                pacc[kb] = fa_mma<X3>(ch, cl, dofh[ks], dofl[ks], pacc[kb]);
            }
        }
        const int kbase_t = kt * BKV;
        unsigned sbh[2][8], sbl[2][8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int key = kbase_t + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float p0 = key < g.Nk ? exp2f(sacc[kb][r] - lse2) : 0.f;
                float p1 = key + 1 < g.Nk ? exp2f(sacc[kb][r + 1] - lse2) : 0.f;
                float d0 = pacc[kb][r], d1 = pacc[kb][r + 1];
                if (thr > 0u) {
                    const unsigned hsh = fa_keep_pair(g.seed, row_id, (unsigned)key >> 1);
                    d0 = (hsh & 0xffffu) >= thr ? d0 * keep_scale : 0.f;
                    d1 = (hsh >> 16) >= thr ? d1 * keep_scale : 0.f;
                }
                const float s0 = g.scale * p0 * (d0 - dsum), s1 = g.scale * p1 * (d1 - dsum);
                if (X3) fa_split2(s0, s1, sbh[kb][r >> 1], sbl[kb][r >> 1]);
                else sbh[kb][r >> 1] = fa_pack2(s0, s1);
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 sfh = fa_from4(&sbh[kb][4 * ks]);
                const bf16x8 sfl = X3 ? fa_from4(&sbl[kb][4 * ks]) : sfh;
                unsigned long long ka[2][2], kl[2][2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned ad = k_base + 2u * (unsigned)((kb * 32 + 16 * ks) * LDV + db * 32);
                    ka[db][0] = fa_tr16(ad);
                    ka[db][1] = fa_tr16(ad + 2u * (unsigned)(8 * LDV));
                    if (X3) {
                        kl[db][0] = fa_tr16(ad + 2u * (unsigned)VPL);
                        kl[db][1] = fa_tr16(ad + 2u * (unsigned)(VPL + 8 * LDV));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 kfh = fa_join(ka[db][0], ka[db][1]);
                    const bf16x8 kfl = X3 ? fa_join(kl[db][0], kl[db][1]) : kfh;
                    dqacc[db] = fa_mma<X3>(kfh, kfl, sfh, sfl, dqacc[db]);
                }
            }
    }
    if (q_ok) {
        float* op = g.dq + qoff;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4*>(op + db * 32 + 8 * r4 + 4 * hi) =
                    make_float4(dqacc[db][4 * r4], dqacc[db][4 * r4 + 1], dqacc[db][4 * r4 + 2], dqacc[db][4 * r4 + 3]);
    }
}

// dK, dV: one wave owns 32 keys (lane = key), loops over 64-query tiles.  Non-swapped form, so that the contraction
// index of dV = Pd^T dO and dK = dS^T Q (the queries) sits in the accumulator REGISTERS:
//   S = Q K^T, dP = dO V^T   : A = Q / dO rows from LDS (ds_read_b128), B = K^T / V^T fragments in registers
//   P = exp2(S - lse[q]), dS = scale * P * (dP * keep/(1-p) - D[q])      (lse / D per register row, from LDS)
//   dV += Pd^T dO, dK += dS^T Q : A = the P / dS registers packed to bf16, B = dO / Q via ds_read_b64_tr_b16
template <int X3>
__global__ void __launch_bounds__(256) flash_bwd_dkv_kernel(AttnBwdArgs g) {
    __shared__ __attribute__((aligned(16))) u16 Qb[(1 + X3) * KPL];
    __shared__ __attribute__((aligned(16))) u16 Qt[(1 + X3) * VPL];
    __shared__ __attribute__((aligned(16))) u16 Ob[(1 + X3) * KPL];
    __shared__ __attribute__((aligned(16))) u16 Ot[(1 + X3) * VPL];
    __shared__ float s_lse[BKV], s_dsum[BKV];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hi = lane >> 5, lk = lane & 31;
    const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int key = blockIdx.x * BQ + wid * 32 + lk;
    const bool k_ok = key < g.Nk;
    const long long koff = ((long long)b * g.Nk + (k_ok ? key : 0)) * 2 * inner + h * HD;
    const float* qbase = g.q + (long long)b * g.Nq * inner + h * HD;
    const float* dobase = g.d_o + (long long)b * g.Nq * inner + h * HD;
    const float qs = g.scale * LOG2E;
    bf16x8 kfh[4], kfl[4], vfh[4], vfl[4];     // K^T / V^T fragments: lane (key, hi) holds k[16 ks + 8 hi .. +8]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa_frag8<X3>(g.kv + koff + 16 * ks + 8 * hi, 1.f, kfh[ks], kfl[ks]);
        fa_frag8<X3>(g.kv + koff + inner + 16 * ks + 8 * hi, 1.f, vfh[ks], vfl[ks]);
    }
    f32x16 dkacc[2], dvacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[i][r] = 0.f; dvacc[i][r] = 0.f; }

    float4 rq[4], ro[4];
    float rl = 0.f, rd = 0.f;
    auto load_tile = [&](int qt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int qq = qt * BKV + (e >> 4);
            const int c4 = (e & 15) * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
            if (qq < g.Nq) {
                a = *reinterpret_cast<const float4*>(qbase + (long long)qq * inner + c4);
                c = *reinterpret_cast<const float4*>(dobase + (long long)qq * inner + c4);
            }
            rq[i] = a; ro[i] = c;
        }
        if (tid < BKV) {
            const int qq = qt * BKV + tid;
            rl = qq < g.Nq ? g.lse[(long long)bh * g.Nq + qq] * LOG2E : INFINITY;     // +inf -> P = 0 for padded rows
            rd = qq < g.Nq ? g.dsum[(long long)bh * g.Nq + qq] : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int r = e >> 4, c4 = (e & 15) * 4;
            fa_store4<X3>(&Qb[r * LDK + c4], KPL, rq[i], qs);
            fa_store4<X3>(&Qt[r * LDV + c4], VPL, rq[i], qs);
            fa_store4<X3>(&Ob[r * LDK + c4], KPL, ro[i], 1.f);
            fa_store4<X3>(&Ot[r * LDV + c4], VPL, ro[i], 1.f);
        }
        if (tid < BKV) { s_lse[tid] = rl; s_dsum[tid] = rd; }
    };
    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const float keep_scale = 1.0f / (1.0f - g.p_drop);
    const int t16 = lane & 15, gq = lane >> 4;
    const unsigned tr_off = 2u * (unsigned)((4 * (gq >> 1) + (t16 >> 2)) * LDV + 16 * (gq & 1) + 4 * (t16 & 3));
    const unsigned q_base = (unsigned)(size_t)(&Qt[0]) + tr_off;
    const unsigned o_base = (unsigned)(size_t)(&Ot[0]) + tr_off;
    const float inv_qs = 1.0f / LOG2E;      // dK = dS^T (scale-free Q): Qt holds q * scale * log2e, undone at the end

    const int nqt = (g.Nq + BKV - 1) / BKV;
    load_tile(0);
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (qt + 1 < nqt) load_tile(qt + 1);
        f32x16 sacc[2], pacc[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[qb][r] = 0.f; pacc[qb][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u16* qp = &Qb[(qb * 32 + lk) * LDK + 16 * ks + 8 * hi];
                const u16* op = &Ob[(qb * 32 + lk) * LDK + 16 * ks + 8 * hi];
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(qp), al = *reinterpret_cast<const bf16x8*>(qp + X3 * KPL);
                const bf16x8 ch = *reinterpret_cast<const bf16x8*>(op), cl = *reinterpret_cast<const bf16x8*>(op + X3 * KPL);
                sacc[qb] = fa_mma<X3>(ah, al, kfh[ks], kfl[ks], sacc[qb]);
                pacc[qb] = fa_mma<X3>(ch, cl, vfh[ks], vfl[ks], pacc[qb]);
            }
        }
        // sacc[qb][r] = S[q = qt*64 + qb*32 + (r&3) + 8*(r>>2) + 4*hi][key = this lane's key]
        unsigned pbh[2][8], pbl[2][8], sbh[2][8], sbl[2][8];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int ql = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;          // local query row of reg r (r+1 -> ql+1)
                float p0 = k_ok ? exp2f(sacc[qb][r] - s_lse[ql]) : 0.f;
                float p1 = k_ok ? exp2f(sacc[qb][r + 1] - s_lse[ql + 1]) : 0.f;
                float d0 = pacc[qb][r], d1 = pacc[qb][r + 1];
                float pd0 = p0, pd1 = p1;
                if (thr > 0u) {
                    const unsigned row0 = (unsigned)bh * (unsigned)g.Nq + (unsigned)(qt * BKV + ql);
                    const unsigned h0 = fa_keep_pair(g.seed, row0, (unsigned)key >> 1);
                    const unsigned h1 = fa_keep_pair(g.seed, row0 + 1u, (unsigned)key >> 1);
                    const bool k0 = ((key & 1) ? (h0 >> 16) : (h0 & 0xffffu)) >= thr;
                    const bool k1 = ((key & 1) ? (h1 >> 16) : (h1 & 0xffffu)) >= thr;
                    d0 = k0 ? d0 * keep_scale : 0.f; d1 = k1 ? d1 * keep_scale : 0.f;
                    pd0 = k0 ? p0 * keep_scale : 0.f; pd1 = k1 ? p1 * keep_scale : 0.f;
                }
                const float s0 = g.scale * p0 * (d0 - s_dsum[ql]), s1 = g.scale * p1 * (d1 - s_dsum[ql + 1]);
                if (X3) {
                    fa_split2(pd0, pd1, pbh[qb][r >> 1], pbl[qb][r >> 1]);
                    fa_split2(s0, s1, sbh[qb][r >> 1], sbl[qb][r >> 1]);
                } else {
                    pbh[qb][r >> 1] = fa_pack2(pd0, pd1);
                    sbh[qb][r >> 1] = fa_pack2(s0, s1);
                }
            }
        // dV += Pd^T dO ; dK += dS^T Q : contraction over the 64 queries = 2 q-blocks x 2 k-steps of 16
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pfh = fa_from4(&pbh[qb][4 * ks]), sfh = fa_from4(&sbh[qb][4 * ks]);
                const bf16x8 pfl = X3 ? fa_from4(&pbl[qb][4 * ks]) : pfh, sfl = X3 ? fa_from4(&sbl[qb][4 * ks]) : sfh;
                unsigned long long oa[2][2], qa[2][2], ol[2][2], ql2[2][2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned ro2 = 2u * (unsigned)((qb * 32 + 16 * ks) * LDV + db * 32);
                    oa[db][0] = fa_tr16(o_base + ro2);
                    oa[db][1] = fa_tr16(o_base + ro2 + 2u * (unsigned)(8 * LDV));
                    qa[db][0] = fa_tr16(q_base + ro2);
                    qa[db][1] = fa_tr16(q_base + ro2 + 2u * (unsigned)(8 * LDV));
                    if (X3) {
                        ol[db][0] = fa_tr16(o_base + ro2 + 2u * (unsigned)VPL);
                        ol[db][1] = fa_tr16(o_base + ro2 + 2u * (unsigned)(VPL + 8 * LDV));
                        ql2[db][0] = fa_tr16(q_base + ro2 + 2u * (unsigned)VPL);
                        ql2[db][1] = fa_tr16(q_base + ro2 + 2u * (unsigned)(VPL + 8 * LDV));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 ofh = fa_join(oa[db][0], oa[db][1]), q2h = fa_join(qa[db][0], qa[db][1]);
                    const bf16x8 ofl = X3 ? fa_join(ol[db][0], ol[db][1]) : ofh, q2l = X3 ? fa_join(ql2[db][0], ql2[db][1]) : q2h;
                    dvacc[db] = fa_mma<X3>(pfh, pfl, ofh, ofl, dvacc[db]);
                    dkacc[db] = fa_mma<X3>(sfh, sfl, q2h, q2l, dkacc[db]);
                }
            }
    }
    // accumulators: C[i = key (row, regs)][j = d (lane)]: row = (r&3) + 8*(r>>2) + 4*hi of the wave's 32 keys, col = db*32 + (lane & 31)
    const int kw0 = blockIdx.x * BQ + wid * 32;
    float* dkp = g.dkv + (long long)b * g.Nk * 2 * inner + h * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = kw0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kk < g.Nk) {
                dkp[(long long)kk * 2 * inner + db * 32 + lk] = dkacc[db][r] * (inv_qs / g.scale);
                dkp[(long long)kk * 2 * inner + inner + db * 32 + lk] = dvacc[db][r];
            }
        }
}

int fa_fwd_impl(int x3, const float* q, const float* kv, float* o, float* lse, int B, int H, int Nq, int Nk, int head_dim,
                float scale, float dropout_p, uint32_t seed, vxb_stream_t stream) {
    if (!q || !kv || !o || !lse || B < 1 || H < 1 || Nq < 1 || Nk < 1) return VXB_EARG;
    if (head_dim != HD || dropout_p < 0.f || dropout_p >= 1.f) return VXB_ESIZE;
    AttnArgs g;
    g.q = q; g.kv = kv; g.o = o; g.lse = lse; g.B = B; g.H = H; g.Nq = Nq; g.Nk = Nk;
    g.scale = scale; g.p_drop = dropout_p; g.seed = seed;
    const dim3 grid(vxb_cdiv(Nq, BQ), B * H);
    if (x3) hipLaunchKernelGGL(flash_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(flash_fwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, g);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

int fa_bwd_impl(int x3, const float* q, const float* kv, const float* o, const float* d_o, const float* lse, float* dq,
                float* dkv, float* dsum_ws, int B, int H, int Nq, int Nk, int head_dim, float scale, float dropout_p,
                uint32_t seed, vxb_stream_t stream) {
    if (!q || !kv || !o || !d_o || !lse || !dq || !dkv || !dsum_ws || B < 1 || H < 1 || Nq < 1 || Nk < 1) return VXB_EARG;
    if (head_dim != HD || dropout_p < 0.f || dropout_p >= 1.f) return VXB_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    const long long groups = (long long)B * Nq * H;
    hipLaunchKernelGGL(flash_rowdot_kernel, dim3(vxb_cdiv(groups * 16, 256)), dim3(256), 0, st, d_o, o, dsum_ws, B, H, Nq);
    AttnBwdArgs g;
    g.q = q; g.kv = kv; g.d_o = d_o; g.lse = lse; g.dsum = dsum_ws; g.dq = dq; g.dkv = dkv;
    g.B = B; g.H = H; g.Nq = Nq; g.Nk = Nk; g.scale = scale; g.p_drop = dropout_p; g.seed = seed;
    if (x3) {
        hipLaunchKernelGGL(flash_bwd_dq_kernel<1>, dim3(vxb_cdiv(Nq, BQ), B * H), dim3(256), 0, st, g);
        hipLaunchKernelGGL(flash_bwd_dkv_kernel<1>, dim3(vxb_cdiv(Nk, BQ), B * H), dim3(256), 0, st, g);
    } else {
        hipLaunchKernelGGL(flash_bwd_dq_kernel<0>, dim3(vxb_cdiv(Nq, BQ), B * H), dim3(256), 0, st, g);
        hipLaunchKernelGGL(flash_bwd_dkv_kernel<0>, dim3(vxb_cdiv(Nk, BQ), B * H), dim3(256), 0, st, g);
    }
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

}  // namespace

// O = dropout(softmax(scale * Q K^T)) V per (b, h), head dim 64; lse[b*H + h][q] = log sum exp of the scaled scores.
extern "C" int vxb_flash_attn_fwd_bf16(const float* q, const float* kv, float* o, float* lse, int B, int H, int Nq, int Nk,
                                       int head_dim, float scale, float dropout_p, uint32_t seed, vxb_stream_t stream) {
    return fa_fwd_impl(0, q, kv, o, lse, B, H, Nq, Nk, head_dim, scale, dropout_p, seed, stream);
}
extern "C" int vxb_flash_attn_fwd_bf16x3(const float* q, const float* kv, float* o, float* lse, int B, int H, int Nq, int Nk,
                                         int head_dim, float scale, float dropout_p, uint32_t seed, vxb_stream_t stream) {
    return fa_fwd_impl(1, q, kv, o, lse, B, H, Nq, Nk, head_dim, scale, dropout_p, seed, stream);
}

// Backward of the fused attention: dq [B,Nq,H*64] and dkv [B,Nk,2*H*64] are WRITTEN.  o / lse come from the forward call;
// dsum_ws: B*H*Nq floats of scratch.  Same (seed, dropout_p) as the forward call.
extern "C" int vxb_flash_attn_bwd_bf16(const float* q, const float* kv, const float* o, const float* d_o, const float* lse,
                                       float* dq, float* dkv, float* dsum_ws, int B, int H, int Nq, int Nk, int head_dim,
                                       float scale, float dropout_p, uint32_t seed, vxb_stream_t stream) {
    return fa_bwd_impl(0, q, kv, o, d_o, lse, dq, dkv, dsum_ws, B, H, Nq, Nk, head_dim, scale, dropout_p, seed, stream);
}
extern "C" int vxb_flash_attn_bwd_bf16x3(const float* q, const float* kv, const float* o, const float* d_o, const float* lse,
                                         float* dq, float* dkv, float* dsum_ws, int B, int H, int Nq, int Nk, int head_dim,
                                         float scale, float dropout_p, uint32_t seed, vxb_stream_t stream) {
    return fa_bwd_impl(1, q, kv, o, d_o, lse, dq, dkv, dsum_ws, B, H, Nq, Nk, head_dim, scale, dropout_p, seed, stream);
}
