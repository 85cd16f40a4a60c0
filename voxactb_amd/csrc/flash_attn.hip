// Fused ("flash") attention for the Perceiver blocks on the bf16 matrix cores (throughput mode).
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
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// keep decisions for keys (2c, 2c+1) of row `row`: low / high 16 bits of one hash against thr = p * 65536
__device__ __forceinline__ unsigned fa_keep_pair(unsigned seed, unsigned row, unsigned colpair) {
    return fa_hash((row * 0x9E3779B1U + seed) ^ (colpair * 0x85EBCA77U + 0xC2B2AE3DU));
}
__device__ __forceinline__ unsigned fa_pack2(float lo, float hi) {
    unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7fffu + ((a >> 16) & 1u);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xffff0000u);
}
__device__ __forceinline__ unsigned long long fa_tr16(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

__global__ void __launch_bounds__(256) flash_fwd_kernel(AttnArgs g) {
    __shared__ __attribute__((aligned(16))) u16 Ks[BKV * LDK];
    __shared__ __attribute__((aligned(16))) u16 Vs[BKV * LDV];
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
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        union { unsigned u[4]; bf16x8 v; } t;
        const float4 a = *reinterpret_cast<const float4*>(qp + 16 * ks + 8 * hi);
        const float4 c = *reinterpret_cast<const float4*>(qp + 16 * ks + 8 * hi + 4);
        t.u[0] = fa_pack2(a.x * qs, a.y * qs); t.u[1] = fa_pack2(a.z * qs, a.w * qs);
        t.u[2] = fa_pack2(c.x * qs, c.y * qs); t.u[3] = fa_pack2(c.z * qs, c.w * qs);
        qf[ks] = t.v;
    }
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
            uint2 pk, pv;
            pk.x = fa_pack2(rk[i].x, rk[i].y); pk.y = fa_pack2(rk[i].z, rk[i].w);
            pv.x = fa_pack2(rv[i].x, rv[i].y); pv.y = fa_pack2(rv[i].z, rv[i].w);
            *reinterpret_cast<uint2*>(&Ks[r * LDK + c4]) = pk;
            *reinterpret_cast<uint2*>(&Vs[r * LDV + c4]) = pv;
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
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Ks[(kb * 32 + lq) * LDK + 16 * ks + 8 * hi]);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], sacc[kb], 0, 0, 0);
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
        unsigned pb[2][8];                                     // P^T fragments (bf16 pairs): [key block][4 dwords x 2 k-steps]
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
                pb[kb][r >> 1] = fa_pack2(p0, p1);
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
                union { unsigned u[4]; bf16x8 v; } pf;
                pf.u[0] = pb[kb][4 * ks]; pf.u[1] = pb[kb][4 * ks + 1]; pf.u[2] = pb[kb][4 * ks + 2]; pf.u[3] = pb[kb][4 * ks + 3];
                unsigned long long va[2][2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned ad = v_base + 2u * (unsigned)((kb * 32 + 16 * ks) * LDV + db * 32);
                    va[db][0] = fa_tr16(ad);
                    va[db][1] = fa_tr16(ad + 2u * (unsigned)(8 * LDV));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    union { unsigned long long u[2]; bf16x8 v; } vf;
                    vf.u[0] = va[db][0]; vf.u[1] = va[db][1];
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, oacc[db], 0, 0, 0);
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
__global__ void __launch_bounds__(256) flash_bwd_dq_kernel(AttnBwdArgs g) {
    __shared__ __attribute__((aligned(16))) u16 Kb[BKV * LDK];     // K, b128-friendly stride
    __shared__ __attribute__((aligned(16))) u16 Kt[BKV * LDV];     // K, transposed-read-friendly stride
    __shared__ __attribute__((aligned(16))) u16 Vb[BKV * LDK];     // V, b128-friendly stride
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
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        union { unsigned u[4]; bf16x8 v; } t, u;
        const float4 a = *reinterpret_cast<const float4*>(g.q + qoff + 16 * ks + 8 * hi);
        const float4 c = *reinterpret_cast<const float4*>(g.q + qoff + 16 * ks + 8 * hi + 4);
        t.u[0] = fa_pack2(a.x * qs, a.y * qs); t.u[1] = fa_pack2(a.z * qs, a.w * qs);
        t.u[2] = fa_pack2(c.x * qs, c.y * qs); t.u[3] = fa_pack2(c.z * qs, c.w * qs);
        qf[ks] = t.v;
        const float4 e = *reinterpret_cast<const float4*>(g.d_o + qoff + 16 * ks + 8 * hi);
        const float4 f = *reinterpret_cast<const float4*>(g.d_o + qoff + 16 * ks + 8 * hi + 4);
        u.u[0] = fa_pack2(e.x, e.y); u.u[1] = fa_pack2(e.z, e.w); u.u[2] = fa_pack2(f.x, f.y); u.u[3] = fa_pack2(f.z, f.w);
        dof[ks] = u.v;
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
            uint2 pk, pv;
            pk.x = fa_pack2(rk[i].x, rk[i].y); pk.y = fa_pack2(rk[i].z, rk[i].w);
            pv.x = fa_pack2(rv[i].x, rv[i].y); pv.y = fa_pack2(rv[i].z, rv[i].w);
            *reinterpret_cast<uint2*>(&Kb[r * LDK + c4]) = pk;
            *reinterpret_cast<uint2*>(&Kt[r * LDV + c4]) = pk;
            *reinterpret_cast<uint2*>(&Vb[r * LDK + c4]) = pv;
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
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Kb[(kb * 32 + lq) * LDK + 16 * ks + 8 * hi]);
                const bf16x8 c = *reinterpret_cast<const bf16x8*>(&Vb[(kb * 32 + lq) * LDK + 16 * ks + 8 * hi]);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], sacc[kb], 0, 0, 0);
                pacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, dof[ks], pacc[kb], 0, 0, 0);
            }
        }
        const int kbase_t = kt * BKV;
        unsigned sb[2][8];
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
                sb[kb][r >> 1] = fa_pack2(g.scale * p0 * (d0 - dsum), g.scale * p1 * (d1 - dsum));
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                union { unsigned u[4]; bf16x8 v; } sf;
                sf.u[0] = sb[kb][4 * ks]; sf.u[1] = sb[kb][4 * ks + 1]; sf.u[2] = sb[kb][4 * ks + 2]; sf.u[3] = sb[kb][4 * ks + 3];
                unsigned long long ka[2][2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned ad = k_base + 2u * (unsigned)((kb * 32 + 16 * ks) * LDV + db * 32);
                    ka[db][0] = fa_tr16(ad);
                    ka[db][1] = fa_tr16(ad + 2u * (unsigned)(8 * LDV));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    union { unsigned long long u[2]; bf16x8 v; } kf;
                    kf.u[0] = ka[db][0]; kf.u[1] = ka[db][1];
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v, sf.v, dqacc[db], 0, 0, 0);
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
__global__ void __launch_bounds__(256) flash_bwd_dkv_kernel(AttnBwdArgs g) {
    __shared__ __attribute__((aligned(16))) u16 Qb[BKV * LDK];
    __shared__ __attribute__((aligned(16))) u16 Qt[BKV * LDV];
    __shared__ __attribute__((aligned(16))) u16 Ob[BKV * LDK];
    __shared__ __attribute__((aligned(16))) u16 Ot[BKV * LDV];
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
    bf16x8 kf[4], vf[4];                 // K^T / V^T fragments: lane (key, hi) holds k[16 ks + 8 hi .. +8]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        union { unsigned u[4]; bf16x8 v; } t, u;
        const float4 a = *reinterpret_cast<const float4*>(g.kv + koff + 16 * ks + 8 * hi);
        const float4 c = *reinterpret_cast<const float4*>(g.kv + koff + 16 * ks + 8 * hi + 4);
        t.u[0] = fa_pack2(a.x, a.y); t.u[1] = fa_pack2(a.z, a.w); t.u[2] = fa_pack2(c.x, c.y); t.u[3] = fa_pack2(c.z, c.w);
        kf[ks] = t.v;
        const float4 e = *reinterpret_cast<const float4*>(g.kv + koff + inner + 16 * ks + 8 * hi);
        const float4 f = *reinterpret_cast<const float4*>(g.kv + koff + inner + 16 * ks + 8 * hi + 4);
        u.u[0] = fa_pack2(e.x, e.y); u.u[1] = fa_pack2(e.z, e.w); u.u[2] = fa_pack2(f.x, f.y); u.u[3] = fa_pack2(f.z, f.w);
        vf[ks] = u.v;
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
            uint2 pq, po;
            pq.x = fa_pack2(rq[i].x * qs, rq[i].y * qs); pq.y = fa_pack2(rq[i].z * qs, rq[i].w * qs);
            po.x = fa_pack2(ro[i].x, ro[i].y); po.y = fa_pack2(ro[i].z, ro[i].w);
            *reinterpret_cast<uint2*>(&Qb[r * LDK + c4]) = pq;
            *reinterpret_cast<uint2*>(&Qt[r * LDV + c4]) = pq;
            *reinterpret_cast<uint2*>(&Ob[r * LDK + c4]) = po;
            *reinterpret_cast<uint2*>(&Ot[r * LDV + c4]) = po;
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
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Qb[(qb * 32 + lk) * LDK + 16 * ks + 8 * hi]);
                const bf16x8 c = *reinterpret_cast<const bf16x8*>(&Ob[(qb * 32 + lk) * LDK + 16 * ks + 8 * hi]);
                sacc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[ks], sacc[qb], 0, 0, 0);
                pacc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, vf[ks], pacc[qb], 0, 0, 0);
            }
        }
        // sacc[qb][r] = S[q = qt*64 + qb*32 + (r&3) + 8*(r>>2) + 4*hi][key = this lane's key]
        unsigned pb[2][8], sb[2][8];
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
                pb[qb][r >> 1] = fa_pack2(pd0, pd1);
                sb[qb][r >> 1] = fa_pack2(g.scale * p0 * (d0 - s_dsum[ql]), g.scale * p1 * (d1 - s_dsum[ql + 1]));
            }
        // dV += Pd^T dO ; dK += dS^T Q : contraction over the 64 queries = 2 q-blocks x 2 k-steps of 16
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                union { unsigned u[4]; bf16x8 v; } pf, sf;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) { pf.u[w4] = pb[qb][4 * ks + w4]; sf.u[w4] = sb[qb][4 * ks + w4]; }
                unsigned long long oa[2][2], qa[2][2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned ro2 = 2u * (unsigned)((qb * 32 + 16 * ks) * LDV + db * 32);
                    oa[db][0] = fa_tr16(o_base + ro2);
                    oa[db][1] = fa_tr16(o_base + ro2 + 2u * (unsigned)(8 * LDV));
                    qa[db][0] = fa_tr16(q_base + ro2);
                    qa[db][1] = fa_tr16(q_base + ro2 + 2u * (unsigned)(8 * LDV));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    union { unsigned long long u[2]; bf16x8 v; } of, qf2;
                    of.u[0] = oa[db][0]; of.u[1] = oa[db][1];
                    qf2.u[0] = qa[db][0]; qf2.u[1] = qa[db][1];
                    dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf.v, of.v, dvacc[db], 0, 0, 0);
                    dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sf.v, qf2.v, dkacc[db], 0, 0, 0);
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

}  // namespace

// O = dropout(softmax(scale * Q K^T)) V per (b, h), head dim 64; lse[b*H + h][q] = log sum exp of the scaled scores.
extern "C" int vxb_flash_attn_fwd_bf16(const float* q, const float* kv, float* o, float* lse, int B, int H, int Nq, int Nk,
                                       int head_dim, float scale, float dropout_p, uint32_t seed, vxb_stream_t stream) {
    if (!q || !kv || !o || !lse || B < 1 || H < 1 || Nq < 1 || Nk < 1) return VXB_EARG;
    if (head_dim != HD || dropout_p < 0.f || dropout_p >= 1.f) return VXB_ESIZE;
    AttnArgs g;
    g.q = q; g.kv = kv; g.o = o; g.lse = lse; g.B = B; g.H = H; g.Nq = Nq; g.Nk = Nk;
    g.scale = scale; g.p_drop = dropout_p; g.seed = seed;
    hipLaunchKernelGGL(flash_fwd_kernel, dim3(vxb_cdiv(Nq, BQ), B * H), dim3(256), 0, (hipStream_t)stream, g);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// Backward of the fused attention: dq [B,Nq,H*64] and dkv [B,Nk,2*H*64] are WRITTEN.  o / lse come from the forward call;
// dsum_ws: B*H*Nq floats of scratch.  Same (seed, dropout_p) as the forward call.
extern "C" int vxb_flash_attn_bwd_bf16(const float* q, const float* kv, const float* o, const float* d_o, const float* lse,
                                       float* dq, float* dkv, float* dsum_ws, int B, int H, int Nq, int Nk, int head_dim,
                                       float scale, float dropout_p, uint32_t seed, vxb_stream_t stream) {
    if (!q || !kv || !o || !d_o || !lse || !dq || !dkv || !dsum_ws || B < 1 || H < 1 || Nq < 1 || Nk < 1) return VXB_EARG;
    if (head_dim != HD || dropout_p < 0.f || dropout_p >= 1.f) return VXB_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    const long long groups = (long long)B * Nq * H;
    hipLaunchKernelGGL(flash_rowdot_kernel, dim3(vxb_cdiv(groups * 16, 256)), dim3(256), 0, st, d_o, o, dsum_ws, B, H, Nq);
    AttnBwdArgs g;
    g.q = q; g.kv = kv; g.d_o = d_o; g.lse = lse; g.dsum = dsum_ws; g.dq = dq; g.dkv = dkv;
    g.B = B; g.H = H; g.Nq = Nq; g.Nk = Nk; g.scale = scale; g.p_drop = dropout_p; g.seed = seed;
    hipLaunchKernelGGL(flash_bwd_dq_kernel, dim3(vxb_cdiv(Nq, BQ), B * H), dim3(256), 0, st, g);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel, dim3(vxb_cdiv(Nk, BQ), B * H), dim3(256), 0, st, g);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
