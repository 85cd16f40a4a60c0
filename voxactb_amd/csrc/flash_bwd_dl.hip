// Fused attention backward with direct-to-LDS operand tiles ('bf16' and 'bf16x3'): the dQ and dK/dV kernels of
// flash_attn.hip with the operand path of flash_fwd_dl.hip.  k | v, q and dO come as bf16 planes (vxb_split_bf16_f32);
// a 64-row x 64-wide tile travels global -> LDS by global_load_lds_dwordx4 into one of two stages (next tile in flight
// during the current tile's work, one barrier per tile).  ONE LDS copy of a tile serves both access patterns -- row
// fragments (ds_read_b128) and transposed fragments (ds_read_b64_tr_b16): unpadded 128-byte rows whose 16-byte chunks are
// XOR-swizzled with f(row) = rotate-right of ((row >> 1) & 7), which is conflict-free for both (the 16 rows of a b128
// service group get 8 distinct keys per row parity; the even rows r, r+2 of a transposed read differ in the key's 4-bit).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int HD = 64, BQ = 128, BT = 64;
constexpr int TILE = BT * HD;               // u16 per plane tile (8 KB)
constexpr float LOG2E = 1.4426950408889634f;

struct BdArgs {
    const float* q;        // [B, Nq, H*64] fp32
    const float* kv;       // [B, Nk, 2*H*64] fp32
    const float* d_o;      // [B, Nq, H*64] fp32
    const u16* kvp;        // planes [npl][B*Nk][2*H*64]
    const u16* qp;         // planes [npl][B*Nq][H*64]
    const u16* dop;        // planes [npl][B*Nq][H*64]
    long long kv_plane, q_plane;
    const float* lse;
    const float* dsum;
    float* dq;
    float* dkv;
    int B, H, Nq, Nk;
    float scale, p_drop;
    unsigned seed;
};

__device__ __forceinline__ unsigned bd_hash(unsigned x) {
    // one multiply round: the inputs are already products with odd constants; keep-rate, row / column sums and lag
    // correlations of the 16-bit halves are indistinguishable from the two-round finaliser (checked offline on 4096 x 2048
    // masks), and every hash costs a quarter-rate multiply less in the softmax loops
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ unsigned bd_keep_pair(unsigned seed, unsigned row, unsigned colpair) {     // == fa_keep_pair
    return bd_hash((row * 0x9E3779B1U + seed) ^ (colpair * 0x85EBCA77U + 0xC2B2AE3DU));
}
__device__ __forceinline__ void bd_split2(float a, float b, unsigned& ph, unsigned& pl) {
    ph = vxb_pack_bf16(a, b);
    pl = vxb_pack_bf16(a - __uint_as_float(ph << 16), b - __uint_as_float(ph & 0xffff0000u));
}
__device__ __forceinline__ bf16x8 bd_from4(const unsigned* p) {
    union { unsigned u[4]; bf16x8 v; } t;
    t.u[0] = p[0]; t.u[1] = p[1]; t.u[2] = p[2]; t.u[3] = p[3];
    return t.v;
}
__device__ __forceinline__ bf16x8 bd_join(unsigned long long a, unsigned long long b) {
    union { unsigned long long u[2]; bf16x8 v; } t;
    t.u[0] = a; t.u[1] = b;
    return t.v;
}
typedef short bd_v4s __attribute__((ext_vector_type(4)));
// transposed 8-byte LDS read through the compiler builtin (lane base + immediate offset, compiler-placed waits)
__device__ __forceinline__ unsigned long long bd_tr16(const u16* p) {
    union { bd_v4s v; unsigned long long u; } t;
    t.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bd_v4s*)p);
    return t.u;
}
template <int X3>
__device__ __forceinline__ f32x16 bd_mma(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 c) {
    if (X3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}
__device__ __forceinline__ void bd_load16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int bd_swz(int row) {           // chunk XOR key of a tile row
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x >> 1);
}
// 8 consecutive fp32 -> bf16x8 fragment (+ residual), pre-scaled
template <int X3>
__device__ __forceinline__ void bd_frag8(const float* p, float s, bf16x8& fh, bf16x8& fl) {
    union { unsigned u[4]; bf16x8 v; } h, l;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 c = *reinterpret_cast<const float4*>(p + 4);
    if (X3) {
        bd_split2(a.x * s, a.y * s, h.u[0], l.u[0]); bd_split2(a.z * s, a.w * s, h.u[1], l.u[1]);
        bd_split2(c.x * s, c.y * s, h.u[2], l.u[2]); bd_split2(c.z * s, c.w * s, h.u[3], l.u[3]);
        fl = l.v;
    } else {
        h.u[0] = vxb_pack_bf16(a.x * s, a.y * s); h.u[1] = vxb_pack_bf16(a.z * s, a.w * s);
        h.u[2] = vxb_pack_bf16(c.x * s, c.y * s); h.u[3] = vxb_pack_bf16(c.z * s, c.w * s);
        fl = h.v;
    }
    fh = h.v;
}

// ------------------------------------------------------------------------------------------------ dQ
// lane = query (swapped form).  Per K/V tile: S^T = K Q^T, P = exp2(S^T - lse); dP^T = V dO^T;
// dS^T = scale * P * (dP * keep/(1-p) - D); dQ^T += K^T dS^T (K^T by transposed reads of the same K tile).
template <int X3, int DROP>
__global__ void __launch_bounds__(256, 2) flash_bwd_dq_dl_kernel(BdArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int NPL = 1 + X3;
    constexpr int STAGE = 2 * NPL * TILE;           // [K planes][V planes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int qrow = blockIdx.x * BQ + wid * 32 + lq;
    const bool q_ok = qrow < g.Nq;
    const long long qoff = ((long long)b * g.Nq + (q_ok ? qrow : 0)) * inner + h * HD;
    const float qs = g.scale * LOG2E;
    bf16x8 qfh[4], qfl[4], dofh[4], dofl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bd_frag8<X3>(g.q + qoff + 16 * ks + 8 * hi, qs, qfh[ks], qfl[ks]);
        bd_frag8<X3>(g.d_o + qoff + 16 * ks + 8 * hi, 1.f, dofh[ks], dofl[ks]);
    }
    const float lse2 = q_ok ? g.lse[(long long)bh * g.Nq + qrow] * LOG2E : 0.f;
    const float dsum = q_ok ? g.dsum[(long long)bh * g.Nq + qrow] : 0.f;
    f32x16 dqacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

    const long long kv_row0 = (long long)b * g.Nk;
    int lkey[2], lch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        lkey[i] = (2 * wid + i) * 8 + (lane >> 3);
        lch[i] = (lane & 7) ^ bd_swz(lkey[i]);
    }
    auto issue = [&](int stage, int kt) {
        u16* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = min(kt * BT + lkey[i], g.Nk - 1);
            const u16* rowp = g.kvp + (kv_row0 + key) * (2 * inner) + h * HD + lch[i] * 8;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                bd_load16(rowp + p * g.kv_plane, sb + p * TILE + (2 * wid + i) * 512);
                bd_load16(rowp + p * g.kv_plane + inner, sb + (NPL + p) * TILE + (2 * wid + i) * 512);
            }
        }
    };
    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const float keep_scale = 1.0f / (1.0f - g.p_drop);
    const unsigned row_id = (unsigned)bh * (unsigned)g.Nq + (unsigned)qrow;
    int rbase[2], rkey[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int row = kb * 32 + lq; rbase[kb] = row * 64; rkey[kb] = bd_swz(row); }
    const int t16 = lane & 15, gq = lane >> 4;
    const int trow0 = 4 * (gq >> 1) + (t16 >> 2);
    const int tchunk0 = 2 * (gq & 1) + ((t16 & 3) >> 1), thalf = (t16 & 1) * 4;
    // a read's tile offset (multiples of 8 rows) only reaches the swizzle key through its 8-row bit: one lane offset per
    // (d-block, second-read) pair, the rest of the address is an immediate
    int tlane[2][2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
            tlane[db][rr] = trow0 * 64 + ((tchunk0 + 4 * db) ^ bd_swz(8 * rr + trow0)) * 8 + thalf;

    const int nkt = (g.Nk + BT - 1) / BT;
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) issue((kt + 1) & 1, kt + 1);
        const u16* sb = smem + (kt & 1) * STAGE;
        f32x16 sacc[2], pacc[2];
        __builtin_amdgcn_s_setprio(1);              // favour the wave that has matrix work to issue
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[kb][r] = 0.f; pacc[kb][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int co = rbase[kb] + ((2 * ks + hi) ^ rkey[kb]) * 8;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sb + co), al = *reinterpret_cast<const bf16x8*>(sb + X3 * TILE + co);
                const bf16x8 ch = *reinterpret_cast<const bf16x8*>(sb + NPL * TILE + co);
                const bf16x8 cl = *reinterpret_cast<const bf16x8*>(sb + (NPL + X3) * TILE + co);
                sacc[kb] = bd_mma<X3>(ah, al, qfh[ks], qfl[ks], sacc[kb]);
                pacc[kb] = bd_mma<X3>(ch, cl, dofh[ks], dofl[ks], pacc[kb]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        const int kbase_t = kt * BT;
        unsigned sbh[2][8], sbl[2][8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int key = kbase_t + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float p0 = key < g.Nk ? __builtin_amdgcn_exp2f(sacc[kb][r] - lse2) : 0.f;
                float p1 = key + 1 < g.Nk ? __builtin_amdgcn_exp2f(sacc[kb][r + 1] - lse2) : 0.f;
                float d0 = pacc[kb][r], d1 = pacc[kb][r + 1];
                if (DROP) {
                    const unsigned hsh = bd_keep_pair(g.seed, row_id, (unsigned)key >> 1);
                    d0 = (hsh & 0xffffu) >= thr ? d0 * keep_scale : 0.f;
                    d1 = (hsh >> 16) >= thr ? d1 * keep_scale : 0.f;
                }
                const float s0 = g.scale * p0 * (d0 - dsum), s1 = g.scale * p1 * (d1 - dsum);
                if (X3) bd_split2(s0, s1, sbh[kb][r >> 1], sbl[kb][r >> 1]);
                else sbh[kb][r >> 1] = vxb_pack_bf16(s0, s1);
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 sfh = bd_from4(&sbh[kb][4 * ks]);
                const bf16x8 sfl = X3 ? bd_from4(&sbl[kb][4 * ks]) : sfh;
                unsigned long long ka[2][2], kl[2][2];
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const u16* ad = sb + tlane[db][rr] + (kb * 32 + 16 * ks + 8 * rr) * 64;
                        ka[db][rr] = bd_tr16(ad);
                        if (X3) kl[db][rr] = bd_tr16(ad + TILE);
                    }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 kfh = bd_join(ka[db][0], ka[db][1]);
                    const bf16x8 kfl = X3 ? bd_join(kl[db][0], kl[db][1]) : kfh;
                    dqacc[db] = bd_mma<X3>(kfh, kfl, sfh, sfl, dqacc[db]);
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

// ------------------------------------------------------------------------------------------------ dK, dV
// lane = key (non-swapped form), loops over 64-query tiles of Q and dO:
//   S = Q K^T, dP = dO V^T   : A = Q / dO rows from LDS (ds_read_b128), B = K^T * (scale log2e) / V^T fragments in registers
//   P = exp2(S - lse[q]), dS = scale * P * (dP * keep/(1-p) - D[q])
//   dV += Pd^T dO, dK += dS^T Q : A = the P / dS registers packed to bf16, B = dO / Q by transposed reads of the same tiles
template <int X3, int DROP>
__global__ void __launch_bounds__(256, 2) flash_bwd_dkv_dl_kernel(BdArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int NPL = 1 + X3;
    constexpr int STAGE = 2 * NPL * TILE;           // [Q planes][dO planes]
    __shared__ float s_lse[2][BT], s_dsum[2][BT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lk = lane & 31;
    const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int key = blockIdx.x * BQ + wid * 32 + lk;
    const bool k_ok = key < g.Nk;
    const long long koff = ((long long)b * g.Nk + (k_ok ? key : 0)) * 2 * inner + h * HD;
    const float qs = g.scale * LOG2E;
    bf16x8 kfh[4], kfl[4], vfh[4], vfl[4];     // K^T (x scale log2e) / V^T fragments: lane (key, hi) holds k[16 ks + 8 hi .. +8]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bd_frag8<X3>(g.kv + koff + 16 * ks + 8 * hi, qs, kfh[ks], kfl[ks]);
        bd_frag8<X3>(g.kv + koff + inner + 16 * ks + 8 * hi, 1.f, vfh[ks], vfl[ks]);
    }
    f32x16 dkacc[2], dvacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[i][r] = 0.f; dvacc[i][r] = 0.f; }

    const long long q_row0 = (long long)b * g.Nq;
    int lrow[2], lch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        lrow[i] = (2 * wid + i) * 8 + (lane >> 3);
        lch[i] = (lane & 7) ^ bd_swz(lrow[i]);
    }
    auto issue = [&](int stage, int qt) {
        u16* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qq = min(qt * BT + lrow[i], g.Nq - 1);
            const long long ro = (q_row0 + qq) * inner + h * HD + lch[i] * 8;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                bd_load16(g.qp + p * g.q_plane + ro, sb + p * TILE + (2 * wid + i) * 512);
                bd_load16(g.dop + p * g.q_plane + ro, sb + (NPL + p) * TILE + (2 * wid + i) * 512);
            }
        }
        if (tid < BT) {
            const int qq = qt * BT + tid;
            s_lse[stage][tid] = qq < g.Nq ? g.lse[(long long)bh * g.Nq + qq] * LOG2E : INFINITY;     // +inf -> P = 0 for padded rows
            s_dsum[stage][tid] = qq < g.Nq ? g.dsum[(long long)bh * g.Nq + qq] : 0.f;
        }
    };
    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const float keep_scale = 1.0f / (1.0f - g.p_drop);
    const unsigned kodd = (unsigned)key & 1u, hshift = kodd ? 0u : 16u, thr16 = thr << 16;      // key parity == lane parity
    const int t16 = lane & 15, gq = lane >> 4;
    const int trow0 = 4 * (gq >> 1) + (t16 >> 2);
    const int tchunk0 = 2 * (gq & 1) + ((t16 & 3) >> 1), thalf = (t16 & 1) * 4;
    // a read's tile offset (multiples of 8 rows) only reaches the swizzle key through its 8-row bit: one lane offset per
    // (d-block, second-read) pair, the rest of the address is an immediate
    int tlane[2][2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
            tlane[db][rr] = trow0 * 64 + ((tchunk0 + 4 * db) ^ bd_swz(8 * rr + trow0)) * 8 + thalf;

    const int nqt = (g.Nq + BT - 1) / BT;
    issue(0, 0);
    for (int qt = 0; qt < nqt; ++qt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 1 < nqt) issue((qt + 1) & 1, qt + 1);
        const int st = qt & 1;
        const u16* sb = smem + st * STAGE;
        const u16* q_t = sb;
        const u16* o_t = sb + NPL * TILE;
        // the two 32-query blocks of the tile one after the other (a real loop): only one block's S / dP / P / dS
        // registers are live at a time, which is what lets two waves share a SIMD
#pragma unroll 1
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 sacc, pacc;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
            const int arow = qb * 32 + lk;
            const int abase = arow * 64, akey = bd_swz(arow);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int co = abase + ((2 * ks + hi) ^ akey) * 8;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sb + co), al = *reinterpret_cast<const bf16x8*>(sb + X3 * TILE + co);
                const bf16x8 ch = *reinterpret_cast<const bf16x8*>(sb + NPL * TILE + co);
                const bf16x8 cl = *reinterpret_cast<const bf16x8*>(sb + (NPL + X3) * TILE + co);
                sacc = bd_mma<X3>(ah, al, kfh[ks], kfl[ks], sacc);
                pacc = bd_mma<X3>(ch, cl, vfh[ks], vfl[ks], pacc);
            }
            __builtin_amdgcn_s_setprio(0);
            // sacc[r] = S[q = qt*64 + qb*32 + (r&3) + 8*(r>>2) + 4*hi][key = this lane's key]
            unsigned pbh[8], pbl[8], sbh[8], sbl[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int ql = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;          // local query row of reg r (r+1 -> ql+1)
                float p0 = k_ok ? __builtin_amdgcn_exp2f(sacc[r] - s_lse[st][ql]) : 0.f;
                float p1 = k_ok ? __builtin_amdgcn_exp2f(sacc[r + 1] - s_lse[st][ql + 1]) : 0.f;
                float d0 = pacc[r], d1 = pacc[r + 1];
                float pd0 = p0, pd1 = p1;
                if (DROP) {
                    const unsigned row0 = (unsigned)bh * (unsigned)g.Nq + (unsigned)(qt * BT + ql);
                    // the mask word of (row, key pair) covers this lane's key and its neighbour's (lane ^ 1): the even lane
                    // hashes row0, the odd lane row0 + 1 and the two swap words (one DPP move) instead of hashing both
                    const unsigned hm = bd_keep_pair(g.seed, row0 + kodd, (unsigned)key >> 1);
                    const unsigned ho = (unsigned)__builtin_amdgcn_mov_dpp((int)hm, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
                    const unsigned h0 = kodd ? ho : hm, h1 = kodd ? hm : ho;
                    const bool k0 = (h0 << hshift) >= thr16, k1 = (h1 << hshift) >= thr16;      // this key's 16-bit half >= thr
                    d0 = k0 ? d0 * keep_scale : 0.f; d1 = k1 ? d1 * keep_scale : 0.f;
                    pd0 = k0 ? p0 * keep_scale : 0.f; pd1 = k1 ? p1 * keep_scale : 0.f;
                }
                const float s0 = g.scale * p0 * (d0 - s_dsum[st][ql]), s1 = g.scale * p1 * (d1 - s_dsum[st][ql + 1]);
                if (X3) {
                    bd_split2(pd0, pd1, pbh[r >> 1], pbl[r >> 1]);
                    bd_split2(s0, s1, sbh[r >> 1], sbl[r >> 1]);
                } else {
                    pbh[r >> 1] = vxb_pack_bf16(pd0, pd1);
                    sbh[r >> 1] = vxb_pack_bf16(s0, s1);
                }
            }
            // dV += Pd^T dO ; dK += dS^T Q : contraction over this block's 32 queries = 2 k-steps of 16
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pfh = bd_from4(&pbh[4 * ks]), sfh = bd_from4(&sbh[4 * ks]);
                const bf16x8 pfl = X3 ? bd_from4(&pbl[4 * ks]) : pfh, sfl = X3 ? bd_from4(&sbl[4 * ks]) : sfh;
                unsigned long long oa[2][2], qa[2][2], ol[2][2], ql2[2][2];
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int off = qb * 32 * 64 + tlane[db][rr] + (16 * ks + 8 * rr) * 64;
                        oa[db][rr] = bd_tr16(o_t + off);
                        qa[db][rr] = bd_tr16(q_t + off);
                        if (X3) {
                            ol[db][rr] = bd_tr16(o_t + off + TILE);
                            ql2[db][rr] = bd_tr16(q_t + off + TILE);
                        }
                    }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 ofh = bd_join(oa[db][0], oa[db][1]), q2h = bd_join(qa[db][0], qa[db][1]);
                    const bf16x8 ofl = X3 ? bd_join(ol[db][0], ol[db][1]) : ofh, q2l = X3 ? bd_join(ql2[db][0], ql2[db][1]) : q2h;
                    dvacc[db] = bd_mma<X3>(pfh, pfl, ofh, ofl, dvacc[db]);
                    dkacc[db] = bd_mma<X3>(sfh, sfl, q2h, q2l, dkacc[db]);
                }
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
                dkp[(long long)kk * 2 * inner + db * 32 + lk] = dkacc[db][r];          // Q tiles are unscaled: no factor to undo
                dkp[(long long)kk * 2 * inner + inner + db * 32 + lk] = dvacc[db][r];
            }
        }
}

// D[bh][q] = sum_d dO[q, h*64 + d] * O[q, h*64 + d]     (one 16-lane group per (row, head))
__global__ void __launch_bounds__(256) bd_rowdot_kernel(const float* __restrict__ d_o, const float* __restrict__ o,
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

}  // namespace

// Backward of the fused attention with the matrix-core operands as bf16 planes (vxb_split_bf16_f32 of kv, q and dO):
// kv_planes [npl][B*Nk][2*H*64], q_planes / do_planes [npl][B*Nq][H*64].  q, kv, d_o (fp32) are still read for the
// per-lane register fragments.  dq [B,Nq,H*64] and dkv [B,Nk,2*H*64] are WRITTEN; dsum_ws: B*H*Nq floats.
extern "C" int vxb_flash_attn_bwd_dl(const float* q, const float* kv, const float* o, const float* d_o, const float* lse,
                                     const void* kv_planes, const void* q_planes, const void* do_planes, int nplanes,
                                     float* dq, float* dkv, float* dsum_ws, int B, int H, int Nq, int Nk, int head_dim,
                                     float scale, float dropout_p, uint32_t seed, vxb_stream_t stream) {
    if (!q || !kv || !o || !d_o || !lse || !kv_planes || !q_planes || !do_planes || !dq || !dkv || !dsum_ws) return VXB_EARG;
    if (B < 1 || H < 1 || Nq < 1 || Nk < 1 || (nplanes != 1 && nplanes != 2)) return VXB_EARG;
    if (head_dim != HD || dropout_p < 0.f || dropout_p >= 1.f) return VXB_ESIZE;
    if ((((uintptr_t)kv_planes) | ((uintptr_t)q_planes) | ((uintptr_t)do_planes)) & 15) return VXB_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    const long long groups = (long long)B * Nq * H;
    hipLaunchKernelGGL(bd_rowdot_kernel, dim3(vxb_cdiv(groups * 16, 256)), dim3(256), 0, st, d_o, o, dsum_ws, B, H, Nq);
    BdArgs g;
    g.q = q; g.kv = kv; g.d_o = d_o; g.kvp = (const u16*)kv_planes; g.qp = (const u16*)q_planes; g.dop = (const u16*)do_planes;
    g.kv_plane = (long long)B * Nk * 2 * H * HD; g.q_plane = (long long)B * Nq * H * HD;
    g.lse = lse; g.dsum = dsum_ws; g.dq = dq; g.dkv = dkv;
    g.B = B; g.H = H; g.Nq = Nq; g.Nk = Nk; g.scale = scale; g.p_drop = dropout_p; g.seed = seed;
    const size_t lds = (size_t)2 * 2 * nplanes * TILE * sizeof(u16);
    const bool drop = (unsigned)(dropout_p * 65536.0f) > 0u;
    const dim3 gq(vxb_cdiv(Nq, BQ), B * H), gk(vxb_cdiv(Nk, BQ), B * H);
    if (nplanes == 2) {
        if (hipFuncSetAttribute((const void*)flash_bwd_dq_dl_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)flash_bwd_dq_dl_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)flash_bwd_dkv_dl_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)flash_bwd_dkv_dl_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VXB_ELAUNCH;
        if (drop) {
            hipLaunchKernelGGL((flash_bwd_dq_dl_kernel<1, 1>), gq, dim3(256), lds, st, g);
            hipLaunchKernelGGL((flash_bwd_dkv_dl_kernel<1, 1>), gk, dim3(256), lds, st, g);
        } else {
            hipLaunchKernelGGL((flash_bwd_dq_dl_kernel<1, 0>), gq, dim3(256), lds, st, g);
            hipLaunchKernelGGL((flash_bwd_dkv_dl_kernel<1, 0>), gk, dim3(256), lds, st, g);
        }
    } else if (drop) {
        hipLaunchKernelGGL((flash_bwd_dq_dl_kernel<0, 1>), gq, dim3(256), lds, st, g);
        hipLaunchKernelGGL((flash_bwd_dkv_dl_kernel<0, 1>), gk, dim3(256), lds, st, g);
    } else {
        hipLaunchKernelGGL((flash_bwd_dq_dl_kernel<0, 0>), gq, dim3(256), lds, st, g);
        hipLaunchKernelGGL((flash_bwd_dkv_dl_kernel<0, 0>), gk, dim3(256), lds, st, g);
    }
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
