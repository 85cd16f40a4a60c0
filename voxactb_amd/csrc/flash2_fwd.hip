// Fused attention forward, second structure (round 4): every wave runs a three-stage software pipeline over the key tiles,
//
//      region j:   S'(j+1) = K(j+1) Q^T - m      (matrix cores)
//                  P(j)    = exp2(S'(j))          (vector ALU: exp2, dropout, pack -- nothing else)
//                  O      += V(j-1)^T P(j-1)      (matrix cores)
//
// so the matrix-core stream and the vector stream of ONE wave are independent inside a region and the compiler interleaves
// them (round 3's kernel ran S-MFMAs -> softmax -> PV-MFMAs serially per tile and relied on the second wave of the SIMD to
// fill the gaps: 42 % matrix-pipe duty).  What makes the vector stream short enough:
//
//  * the running maximum is NOT recomputed per tile.  The row offset m (an integer, exactly representable as a 16-bit hi | lo
//    pair) is subtracted INSIDE the matrix product: a fifth k-step whose K-side fragment is the constant (1, 1, tail, 0 ..) and
//    whose Q-side fragment is (-m_hi, -m_lo, -BIG, 0 ..) -- one extra MFMA per 32 x 32 block of scores instead of 32 v_max +
//    32 v_sub + the rescale of the output accumulators per tile and lane, and exp2 can start on the first finished register
//    instead of waiting for a row maximum.  m is the (rounded-up) row maximum of the FIRST tile; a later tile whose scores
//    exceed it by more than 2^12 (detected on the tile's row sum, which is needed anyway) takes a rare fix-up path that raises m
//    by an integer -- the rescale factors are exact powers of two.  Keys beyond Nk in the last tile are masked by the same
//    extra k-step (tail = 1 -> score - 60000 / -3e38 -> P = 0).
//  * dropout's keep_scale and the softmax normaliser are applied once, to the output.
//
// K | V tiles travel global -> LDS by global_load_lds_dwordx4 (as in flash_fwd_dl.hip) into rings of four stages each,
// tracked with a COUNTED s_waitcnt vmcnt(n): one bare s_barrier per tile, the loads never drain inside the loop.  A region's
// top waits for the tiles of the NEXT region, so the first fragments of the next region are read before its barrier and the
// fragment pipeline (two groups ahead) never restarts.  Workgroups of 8 waves (256 queries) share the tiles; the block index is remapped so that the
// workgroups of one (batch, head) run on one XCD (their K | V stay in that L2).
//
// Products (MODE): 0 = bf16 operands, 1 = fp16 operands; one MFMA per product.  (An 'f16x2' variant -- q and v as hi + lo, k and
// the probabilities single -- was built and measured first: 1.9x the MFMAs for a 1.4x smaller error, since each product still has
// one singly-rounded side; dropped.)
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#ifndef F2_ABLATE
#define F2_ABLATE 0          // timing experiments only (tools/ab_flash2.sh): 1 no tile loads, 2 no barrier, 4 no exponentials, 8 no LDS fragment reads, 16 no output stores
#endif
constexpr int HD = 64, BKV = 64;
constexpr int TILE = BKV * HD;              // u16 per K or V plane tile (8 KB)
constexpr int NST = 4;                      // ring stages of K and of V
constexpr int VOFF = NST * TILE;            // V ring behind the K ring
constexpr float LOG2E = 1.4426950408889634f;
#ifndef F2_PLIMIT
#define F2_PLIMIT 4096.0f
#endif
constexpr float P_LIMIT = F2_PLIMIT;          // a tile's row sum above this -> raise m (every P <= 4096 fits fp16)

enum { M_BF16 = 0, M_F16 = 1 };

struct F2Args {
    const float* q;       // [B, Nq, H*64] fp32
    const u16* kv;        // [B*Nk][2*H*64] bf16 or fp16
    float* o;
    float* lse;
    unsigned* mask;       // DROP == 2: the keep words of the dropout mask (layout: vxb_flash2_drop_mask_bytes), else unused
    int B, H, Nq, Nk, nqb;
    float scale, p_drop;
    unsigned seed;
};

__device__ __forceinline__ unsigned f2_hash(unsigned x) {      // == fd_hash (flash_fwd_dl.hip): the backward kernels rebuild the same mask
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
    return x;
}
template <int MODE>
__device__ __forceinline__ unsigned f2_pack(float a, float b) {
    return MODE == M_BF16 ? vxb_pack_bf16(a, b) : vxb_pack_f16(a, b);
}
// the 16-bit value a pack kept, back as fp32
template <int MODE>
__device__ __forceinline__ void f2_unpack(unsigned p, float& a, float& b) {
    if (MODE == M_BF16) { a = __uint_as_float(p << 16); b = __uint_as_float(p & 0xffff0000u); }
    else {
        union { unsigned u; vxb_f16x2 h; } t; t.u = p;
        a = (float)t.h[0]; b = (float)t.h[1];
    }
}
template <int MODE>
__device__ __forceinline__ f32x16 f2_mma(const bf16x8 a, const bf16x8 b, f32x16 c) {
    if (MODE == M_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 f2_from4(unsigned a, unsigned b, unsigned c, unsigned d) {
    union { unsigned u[4]; bf16x8 v; } t;
    t.u[0] = a; t.u[1] = b; t.u[2] = c; t.u[3] = d;
    return t.v;
}
__device__ __forceinline__ bf16x8 f2_join(unsigned long long a, unsigned long long b) {
    union { unsigned long long u[2]; bf16x8 v; } t;
    t.u[0] = a; t.u[1] = b;
    return t.v;
}
typedef short f2_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long f2_tr16(const u16* p) {
    union { f2_v4s v; unsigned long long u; } t;
    t.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) f2_v4s*)p);
    return t.u;
}
// 16 bytes per lane, global (base + 32-bit lane offset) -> LDS[wave base + 16 lane], as INLINE ASM: the compiler's waitcnt pass orders
// every later LDS read behind a direct-to-LDS load it knows about (s_waitcnt vmcnt(0) in front of the first ds_read_b64_tr_b16 of a
// region -- the loads this kernel keeps in flight on purpose); issued this way it sees nothing and the hand-placed counted waits are
// the only ones.  m0 = wave base.
__device__ __forceinline__ void f2_load16(const u16* base, unsigned byte_off, unsigned lds_wave_base) {
    if (F2_ABLATE & 1) return;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(byte_off), "s"(base), "s"(lds_wave_base) : "memory");
}
template <int N>
__device__ __forceinline__ void f2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// acc += x that the optimiser cannot re-associate: written as plain C the 32 additions of a region's row sum are collected into ONE
// dependent chain of v_pk_add_f32 behind the last MFMA (16 x (add + s_nop) with nothing to overlap).  The add itself stays a compiler
// instruction -- as inline asm it would read v_exp_f32's result without the wait state the hazard recogniser inserts (lanes of the
// later transcendental passes then see the register's old contents: measured, rows with (q & 4) == 0 wrong); the EMPTY asm only pins
// the value at this point of the program.
__device__ __forceinline__ void f2_acc(float& acc, float x) {
    acc += x;
    asm volatile("" : "+v"(acc));
}

// Dropout keep words (round 6).  The compare of a score's 16-bit hash field against the threshold leaves a 64-bit lane mask in an SGPR pair
// anyway (lanes 0 - 31: the wave's 32 query rows for key k, lanes 32 - 63: the same rows for key k + 4); with DROP == 2 the forward stores
// those pairs -- scalar stores: no vector ALU work -- and the backward kernels read the mask instead of hashing it again (csrc/flash2_bwd.hip:
// 11 and 24 vector instructions per score pair in the dQ and the dK | dV kernel become 4 and 6).  Word (bh, 32-row block, 64-key tile, kb, r, half) =
// keep bits of rows 32 qb32 .. + 32 for key 64 tile + 32 kb + (r & 3) + 8 (r >> 2) + 4 half; 64 words per (row block, tile).
typedef unsigned f2_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void f2_store_keep(unsigned long long m0, unsigned long long m1, unsigned* base, unsigned byte_off) {
    const f2_v4u v = {(unsigned)m0, (unsigned)(m0 >> 32), (unsigned)m1, (unsigned)(m1 >> 32)};
    asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(v), "s"(base), "s"(byte_off));
}

template <int MODE, int DROP, int NW>
__global__ void __launch_bounds__(NW * 64, 2) flash2_fwd_kernel(F2Args g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int LPW = 8 / NW;                     // load instructions per wave and plane tile
    constexpr int NLOAD = 2 * LPW;                  // per wave and load group {K(t + 4), V(t + 2)}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    // block -> (bh, query block): consecutive block ids go round-robin over the 8 XCDs, so ids congruent mod 8 get one
    // contiguous range of (bh, qb) pairs -- the query blocks of a (batch, head) share an L2
    int vb = blockIdx.x;
    {
        const int total = gridDim.x;
        if ((total & 7) == 0) vb = (vb & 7) * (total >> 3) + (vb >> 3);
    }
    const int bh = vb / g.nqb, qblk = vb - bh * g.nqb;
    const int b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int qrow = qblk * (NW * 32) + wid * 32 + lq;         // this lane's query
    const bool q_ok = qrow < g.Nq;
    const float* qp = g.q + ((long long)b * g.Nq + (q_ok ? qrow : 0)) * inner + h * HD;
    const float qs = g.scale * LOG2E;                          // scores live in the log2 domain

    // Q^T fragments: lane (q, hi) holds q[16 ks + 8 hi .. +8] for ks = 0..3
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(qp + 16 * ks + 8 * hi);
        const float4 c = *reinterpret_cast<const float4*>(qp + 16 * ks + 8 * hi + 4);
        float v[8] = {a.x * qs, a.y * qs, a.z * qs, a.w * qs, c.x * qs, c.y * qs, c.z * qs, c.w * qs};
        unsigned ph[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x0 = v[2 * i], x1 = v[2 * i + 1];
            if (MODE != M_BF16) { x0 = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f); x1 = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f); }
            ph[i] = f2_pack<MODE>(x0, x1);
        }
        qf[ks] = f2_from4(ph[0], ph[1], ph[2], ph[3]);
    }
    // the fifth k-step: K side (1, 1, tail, 0 ..) in the hi = 0 lanes, query side (-m_hi, -m_lo, -BIG, 0 ..)
    const unsigned one2 = hi ? 0u : f2_pack<MODE>(1.f, 1.f);
    const unsigned big2 = hi ? 0u : f2_pack<MODE>(MODE == M_BF16 ? -3.0e38f : -60000.f, 0.f);
    unsigned mslot = 0u;                                       // pack(-m_hi, -m_lo)
    float m_run = 0.f, l_run = 0.f;
    auto set_m = [&](float m) {
        const unsigned ph = f2_pack<MODE>(-m, 0.f);
        float h0, h1;
        f2_unpack<MODE>(ph, h0, h1);
        const unsigned p2 = f2_pack<MODE>(-m, -m - h0);
        mslot = hi ? 0u : p2;
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;

    // ---- tile loads: a plane tile = 64 keys x 8 chunks = 8 wave instructions of 1 KB; wave w issues instructions w + NW t.
    //      Address = (b, h) base (SGPR pair) + 32-bit lane offset; keys beyond Nk re-read the last row (masked by the tail slot)
    const u16* kbase_g = g.kv + (long long)b * g.Nk * (2 * inner) + h * HD;
    const u16* vbase_g = kbase_g + inner;
    const unsigned rowb = 2u * 2u * (unsigned)inner;           // bytes per k | v row
    const unsigned smem0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    int lkey[LPW];
    unsigned kchb[LPW], vchb[LPW];
#pragma unroll
    for (int t = 0; t < LPW; ++t) {
        lkey[t] = (wid + NW * t) * 8 + (lane >> 3);
        kchb[t] = (unsigned)(((lane & 7) ^ ((lkey[t] >> 1) & 7)) * 16);
        vchb[t] = (unsigned)(((lane & 7) ^ (4 * ((lkey[t] >> 1) & 1))) * 16);
    }
    auto issue_kv = [&](int ktk, int ktv) {                   // K(ktk) and V(ktv) into their ring slots (ktv < 0: K only)
#pragma unroll
        for (int t = 0; t < LPW; ++t) {
            const unsigned key = (unsigned)min(ktk * BKV + lkey[t], g.Nk - 1);
            f2_load16(kbase_g, key * rowb + kchb[t], smem0 + (unsigned)(((ktk & 3) * TILE + (wid + NW * t) * 512) * 2));
        }
        if (ktv < 0) return;
#pragma unroll
        for (int t = 0; t < LPW; ++t) {
            const unsigned key = (unsigned)min(ktv * BKV + lkey[t], g.Nk - 1);
            f2_load16(vbase_g, key * rowb + vchb[t], smem0 + (unsigned)((VOFF + (ktv & 3) * TILE + (wid + NW * t) * 512) * 2));
        }
    };

    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const unsigned row_id = (unsigned)bh * (unsigned)g.Nq + (unsigned)qrow;
    const unsigned rowh = row_id * 0x9E3779B1U + g.seed;
    unsigned lcg = f2_hash(rowh ^ (hi ? 0x68E31DA4u : 0u));      // DROP == 2: this lane's sequence (the two lanes of a row draw different keys)
    // keep words of this wave's 32 rows: 64 words per key tile (f2_store_keep)
    unsigned* const mwave = DROP == 2 ? g.mask + ((size_t)bh * (size_t)(((g.Nq + 255) >> 8) << 3) + (size_t)(qblk * NW + wid)) * (size_t)((g.Nk + BKV - 1) / BKV) * 64 : nullptr;
    // K fragment (A operand of S^T): key row kb*32 + lq, chunk 2 ks + hi, swizzle key (row >> 1) & 7
    int kbase[2], kkey[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int row = kb * 32 + lq; kbase[kb] = row * 64; kkey[kb] = (row >> 1) & 7; }
    // V^T transposed-read addressing (as flash_fwd_dl.hip)
    const int t16 = lane & 15, gq = lane >> 4;
    const int vrow0 = 4 * (gq >> 1) + (t16 >> 2);
    const int vchunk0 = 2 * (gq & 1) + ((t16 & 3) >> 1), vhalf = (t16 & 1) * 4;
    int vlane[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) vlane[db] = vrow0 * 64 + ((vchunk0 + 4 * db) ^ (4 * ((vrow0 >> 1) & 1))) * 8 + vhalf;

    const int nkt = (g.Nk + BKV - 1) / BKV;

    // fragment group gi = 2 i + kb (i = 0..3: k-step of the scores / (key step, d block) of the P V product; kb: 32-key block)
    auto k_frag = [&](int kt, int gi) -> bf16x8 {
        const int kb = gi & 1, i = gi >> 1;
        if (F2_ABLATE & 8) return qf[i];
        return *reinterpret_cast<const bf16x8*>(smem + (kt & 3) * TILE + kbase[kb] + (((2 * i + hi) ^ kkey[kb]) * 8));
    };
    auto v_frag = [&](int kt, int gi) -> bf16x8 {             // V^T (d block (i & 1) ^ kb, keys kb*32 + 16 (i >> 1) .. +16)
        const int kb = gi & 1, i = gi >> 1;
        if (F2_ABLATE & 8) return qf[i ^ 1];
        const u16* ad = smem + VOFF + (kt & 3) * TILE + vlane[(i & 1) ^ kb] + (kb * 32 + 16 * (i >> 1)) * 64;
        return f2_join(f2_tr16(ad), f2_tr16(ad + 8 * 64));
    };
    auto ones_frag = [&](int kt, int kb) -> bf16x8 {
        const unsigned tail = (kt * BKV + kb * 32 + lq >= g.Nk) ? one2 & 0xffffu : 0u;        // (1, 0): hi = 0 lanes of keys >= Nk
        return f2_from4(one2, tail, 0u, 0u);
    };
    // element pair t = 0..15 of tile kts: P = exp2(S') (dropout applied) packed into pc, row-sum contribution into ps[t & 3]
    auto pair = [&](int kts_colb, unsigned* mtile, int t, const f32x16 (&sa)[2], unsigned (&pc)[2][8], float (&ps)[4]) {
        const int kb = t >> 3, r = 2 * (t & 7);
        if (F2_ABLATE & 4) { pc[kb][r >> 1] = __float_as_uint(sa[kb][r]); return; }
        float p0 = __builtin_amdgcn_exp2f(sa[kb][r]), p1 = __builtin_amdgcn_exp2f(sa[kb][r + 1]);
        f2_acc(ps[t & 3], p0);
        f2_acc(ps[(t + 2) & 3], p1);
        if (DROP == 2) {
            // the mask is DATA here (the backward reads the stored words), so it need not be a function of (seed, row, key) that other
            // kernels with other lane mappings can re-evaluate: one step of a per-lane 24-bit linear congruential sequence per score pair
            // (v_mad_u32_u24, full rate) instead of the xorshift-multiply hash (7 instructions, one of them a quarter-rate 32-bit
            // multiply).  Both 16-bit halves are used: the high one mixes all 24 state bits, the low one is itself a full-period sequence
            // modulo 2^16 whose top bits -- the ones a 10 % threshold looks at -- have periods of 2^13 .. 2^16 steps against the 512 a lane
            // draws.  Keep rate, row / column / block counts and the correlations at 26 lags match the hash's (worst 0.0015 against
            // 0.0017; tools/experiments/README.md, round 6).
            lcg = (lcg & 0xffffffu) * 0x8DA6B3u + 0x9E3779B9u;
            const bool k0 = (lcg & 0xffffu) >= thr, k1 = (lcg >> 16) >= thr;
            f2_store_keep(__builtin_amdgcn_ballot_w64(k0), __builtin_amdgcn_ballot_w64(k1), mtile, (unsigned)((kb * 32 + 2 * r) * 4));
            p0 = k0 ? p0 : 0.f;
            p1 = k1 ? p1 : 0.f;
        } else if (DROP) {
            const unsigned cp = (unsigned)((kb * 32 + (r & 3) + 8 * (r >> 2)) >> 1) * 0x85EBCA77U;
            const unsigned hsh = f2_hash(rowh ^ ((unsigned)kts_colb + cp));
            p0 = (hsh & 0xffffu) >= thr ? p0 : 0.f;
            p1 = (hsh >> 16) >= thr ? p1 : 0.f;
        }
        pc[kb][r >> 1] = f2_pack<MODE>(p0, p1);
    };
    // fragments of the NEXT region's first two groups, read at the end of a region (their tiles were waited for at its top)
    bf16x8 kfc[2], vfc[2];

    // One region, pinned issue order (sched_barrier between the steps; inside a step the compiler orders freely).  For the eight
    // groups gi = 2 i + kb:   [i == 0: S'(kb) = ones x m-slot]
    //                         O(d block (i & 1) ^ kb) += V^T(ktv; kb, i) P_prev(kb, i >> 1)     | element pair 2 gi of P_cur = exp2(S_a)
    //                         S'(kb) += K(ktq; kb, i) Q^T(i)                                    | element pair 2 gi + 1
    // so every accumulator is touched once in four MFMAs.  The fragments of group gi + 2 are requested in group gi -- the last two
    // requests are for the next region (tiles ktq + 1, ktv + 1, landed and synchronised at this region's top).
    auto region = [&](auto has_pv, int ktq, int kts, int ktv, const f32x16 (&sa)[2], f32x16 (&sb)[2], const unsigned (&pp)[2][8],
                      unsigned (&pc)[2][8]) -> float {
        constexpr bool HAS_PV = decltype(has_pv)::value;
        const int colb = (int)((unsigned)((kts * BKV + 4 * hi) >> 1) * 0x85EBCA77U + 0xC2B2AE3DU);
        unsigned* const mtile = DROP == 2 ? mwave + (size_t)kts * 64 : nullptr;
        const bf16x8 msl = f2_from4(mslot, big2, 0u, 0u);
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        bf16x8 kf[10], vf[10];
        kf[0] = kfc[0]; kf[1] = kfc[1]; vf[0] = vfc[0]; vf[1] = vfc[1];
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            const int kb = gi & 1, i = gi >> 1;
            {
                const int gn = gi + 2;
                kf[gn] = k_frag(gn < 8 ? ktq : ktq + 1, gn & 7);
                if (HAS_PV || gn >= 8) vf[gn] = v_frag(gn < 8 ? ktv : ktv + 1, gn & 7);
            }
            if (i == 0) {
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
                sb[kb] = f2_mma<MODE>(ones_frag(ktq, kb), msl, c);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (HAS_PV) {
                const int ks = i >> 1, db = (i & 1) ^ kb;
                oacc[db] = f2_mma<MODE>(vf[gi], f2_from4(pp[kb][4 * ks], pp[kb][4 * ks + 1], pp[kb][4 * ks + 2], pp[kb][4 * ks + 3]), oacc[db]);
            }
            pair(colb, mtile, 2 * gi, sa, pc, ps);
            __builtin_amdgcn_sched_barrier(0);
            sb[kb] = f2_mma<MODE>(kf[gi], qf[i], sb[kb]);
            pair(colb, mtile, 2 * gi + 1, sa, pc, ps);
            __builtin_amdgcn_sched_barrier(0);
        }
        kfc[0] = kf[8]; kfc[1] = kf[9]; vfc[0] = vf[8]; vfc[1] = vf[9];
        return (ps[0] + ps[1]) + (ps[2] + ps[3]);
    };
    // P of a tile without the pipeline (the rare fix-up path)
    auto softmax = [&](int kt, const f32x16 (&s)[2], unsigned (&pk)[2][8]) -> float {
        const int colb = (int)((unsigned)((kt * BKV + 4 * hi) >> 1) * 0x85EBCA77U + 0xC2B2AE3DU);
        unsigned* const mtile = DROP == 2 ? mwave + (size_t)kt * 64 : nullptr;
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; ++t) pair(colb, mtile, t, s, pk, ps);
        return (ps[0] + ps[1]) + (ps[2] + ps[3]);
    };
    // rare: some P of tile kt (scores sa, already offset by the current m) may exceed P_LIMIT -> raise m by an integer, rescale
    // what has been accumulated (exact: powers of two), redo the tile's P and shift the next tile's scores
    auto raise_m = [&](int kt, f32x16 (&sa)[2], f32x16 (&sb2)[2], unsigned (&pk)[2][8], float& ps) {
        float mt = sa[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sa[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float delta = fmaxf(ceilf(mt), 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { oacc[i][r] *= alpha; sa[i][r] -= delta; sb2[i][r] -= delta; }
        l_run *= alpha;
        m_run += delta;
        set_m(m_run);
        ps = softmax(kt, sa, pk);
    };

    // ---- prologue: K0 K1 {K2 V0} {K3 V1} in flight; first tile's scores with m = 0, then m = ceil(row maximum).
    //      Region j consumes K(j+1), V(j-1) and reads the first fragments of K(j+2), V(j): its top waits for {K(j+2), V(j)} and
    //      issues {K(j+4), V(j+2)} -- K slot (j+4) & 3 held K(j) (read in region j-1), V slot (j+2) & 3 held V(j-2) (likewise).
    issue_kv(0, -1); issue_kv(1, -1); issue_kv(2, 0); issue_kv(3, 1);
    f2_wait_vm<2 * NLOAD>();                               // K0 K1 landed (this wave's part)
    vxb_raw_barrier();
    f32x16 sa[2], sb2[2];
    unsigned pa[2][8], pb[2][8];
    {
        const bf16x8 msl0 = f2_from4(0u, big2, 0u, 0u);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
            c = f2_mma<MODE>(ones_frag(0, kb), msl0, c);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = f2_mma<MODE>(k_frag(0, 2 * ks + kb), qf[ks], c);
            sa[kb] = c;
        }
        float mt = sa[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sa[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        m_run = ceilf(mt);
        set_m(m_run);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[kb][r] -= m_run;
        kfc[0] = k_frag(1, 0); kfc[1] = k_frag(1, 1);
        vfc[0] = kfc[0]; vfc[1] = kfc[1];                  // (region 0 has no P V product)
    }
    // ---- region 0: S'(1) | P(0)
    f2_wait_vm<NLOAD>();                                   // {K2 V0} landed
    vxb_raw_barrier();
    issue_kv(4, 2);
    {
        float ps = region(std::false_type(), 1, 0, -1, sa, sb2, pa, pa);
        if (__builtin_amdgcn_ballot_w64(!(ps <= P_LIMIT))) raise_m(0, sa, sb2, pa, ps);
        l_run += ps;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sa[kb] = sb2[kb];
    }
    // ---- regions 1 .. nkt-1: S'(j+1) | P(j) | O += V(j-1) P(j-1); P alternates between pa and pb
    int j = 1;
    for (; j + 1 < nkt; j += 2) {
        f2_wait_vm<NLOAD>();                               // {K(j+2), V(j)} landed (needed by the NEXT region); {K(j+3), V(j+1)} may be in flight
        vxb_raw_barrier();
        issue_kv(j + 4, j + 2);
        {
            float ps = region(std::true_type(), j + 1, j, j - 1, sa, sb2, pa, pb);
            if (__builtin_amdgcn_ballot_w64(!(ps <= P_LIMIT))) raise_m(j, sa, sb2, pb, ps);
            l_run += ps;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sa[kb] = sb2[kb];
        }
        f2_wait_vm<NLOAD>();
        vxb_raw_barrier();
        issue_kv(j + 5, j + 3);
        {
            float ps = region(std::true_type(), j + 2, j + 1, j, sa, sb2, pb, pa);
            if (__builtin_amdgcn_ballot_w64(!(ps <= P_LIMIT))) raise_m(j + 1, sa, sb2, pa, ps);
            l_run += ps;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sa[kb] = sb2[kb];
        }
    }
    // O += V(nkt-1) P(nkt-1): the tile was synchronised at the last region's top, its first two fragments are in vfc.  (Two calls, not
    // a runtime choice between pa and pb: a select between the two ARRAYS sends both to scratch memory.)
    auto last_pv = [&](const unsigned (&pl)[2][8]) {
        bf16x8 vf[8];
        vf[0] = vfc[0]; vf[1] = vfc[1];
#pragma unroll
        for (int gi = 2; gi < 8; ++gi) vf[gi] = v_frag(nkt - 1, gi);
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            const int kb = gi & 1, i = gi >> 1, ks = i >> 1, db = (i & 1) ^ kb;
            oacc[db] = f2_mma<MODE>(vf[gi], f2_from4(pl[kb][4 * ks], pl[kb][4 * ks + 1], pl[kb][4 * ks + 2], pl[kb][4 * ks + 3]), oacc[db]);
        }
    };
    if (j < nkt) {                                          // one region left: P(j) into pb
        f2_wait_vm<NLOAD>();
        vxb_raw_barrier();
        issue_kv(j + 4, j + 2);
        float ps = region(std::true_type(), j + 1, j, j - 1, sa, sb2, pa, pb);
        if (__builtin_amdgcn_ballot_w64(!(ps <= P_LIMIT))) raise_m(j, sa, sb2, pb, ps);
        l_run += ps;
        last_pv(pb);
    } else {
        last_pv(pa);
    }
    f2_wait_vm<0>();                                        // the clamped loads of tiles >= nkt must not outlive the workgroup's LDS
    if (DROP == 2) asm volatile("s_dcache_wb");             // the scalar stores of the keep words leave the scalar data cache

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (q_ok) {
        const float inv = (1.0f / (1.0f - g.p_drop)) / l_tot;
        float* op = g.o + ((long long)b * g.Nq + qrow) * inner + h * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float4 v;
                v.x = oacc[db][4 * r4 + 0] * inv; v.y = oacc[db][4 * r4 + 1] * inv;
                v.z = oacc[db][4 * r4 + 2] * inv; v.w = oacc[db][4 * r4 + 3] * inv;
                if (!(F2_ABLATE & 16) || v.x == 12345.f) *reinterpret_cast<float4*>(op + db * 32 + 8 * r4 + 4 * hi) = v;      // (16: no output stores)
            }
        if (hi == 0) g.lse[(long long)bh * g.Nq + qrow] = (m_run + log2f(l_tot)) * (1.0f / LOG2E);
    }
}

// ------------------------------------------------------------------------------------------------ bf16x3 forward
// The arithmetic the default precision needs (every product as hi*hi + hi*lo + lo*hi on bf16 operands: a single 16-bit product in the
// attention FORWARD moves the parameter gradients by 5e-3, tests/test_grad_noise_gpu.py) in the pipelined structure, at the granularity
// of the backward kernels: units u = 2 kt + kb of 32 keys,
//      step u:   S'(u+1) = K(u+1) Q^T - m  (13 MFMAs)  |  P(u) = exp2(S'(u)), hi + lo split  |  O += V(u-1)^T P(u-1)  (12 MFMAs)
// so only ONE unit's scores and probabilities are double-buffered (the tile-granular form above would need 270 registers).  Region j =
// steps (j, 1), (j + 1, 0): K tile j + 1 and V tile j (hi | lo planes each), rings of three stages, 8 waves share them.
struct F3Args {
    const float* q;
    const u16* kv;        // planes [2][B*Nk][2*H*64] bf16 (vxb_split_bf16_f32)
    long long kv_plane;   // u16 per plane
    float* o;
    float* lse;
    int B, H, Nq, Nk, nqb;
    float scale, p_drop;
    unsigned seed;
};

template <int DROP, int NW>
__global__ void __launch_bounds__(NW * 64, 2) flash2_fwd_x3_kernel(F3Args g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int MODE = M_BF16;
    constexpr int NS = 3;
    constexpr int VOF = NS * 2 * TILE;              // V ring behind the K ring; a stage = [hi plane][lo plane]
    constexpr int LPW = 8 / NW;
    constexpr int NLOAD = 4 * LPW;                  // K hi | lo and V hi | lo pieces per wave and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int vb = blockIdx.x;
    {
        const int total = gridDim.x;
        if ((total & 7) == 0) vb = (vb & 7) * (total >> 3) + (vb >> 3);
    }
    const int bh = vb / g.nqb, qblk = vb - bh * g.nqb;
    const int b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int qrow = qblk * (NW * 32) + wid * 32 + lq;
    const bool q_ok = qrow < g.Nq;
    const float* qp = g.q + ((long long)b * g.Nq + (q_ok ? qrow : 0)) * inner + h * HD;
    const float qs = g.scale * LOG2E;
    bf16x8 qfh[4], qfl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(qp + 16 * ks + 8 * hi);
        const float4 c = *reinterpret_cast<const float4*>(qp + 16 * ks + 8 * hi + 4);
        const float v[8] = {a.x * qs, a.y * qs, a.z * qs, a.w * qs, c.x * qs, c.y * qs, c.z * qs, c.w * qs};
        unsigned ph[4], pl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ph[i] = vxb_pack_bf16(v[2 * i], v[2 * i + 1]);
            pl[i] = vxb_pack_bf16(v[2 * i] - __uint_as_float(ph[i] << 16), v[2 * i + 1] - __uint_as_float(ph[i] & 0xffff0000u));
        }
        qfh[ks] = f2_from4(ph[0], ph[1], ph[2], ph[3]);
        qfl[ks] = f2_from4(pl[0], pl[1], pl[2], pl[3]);
    }
    const unsigned one2 = hi ? 0u : vxb_pack_bf16(1.f, 1.f);
    const unsigned big2 = hi ? 0u : vxb_pack_bf16(-3.0e38f, 0.f);
    unsigned mslot = 0u;
    float m_run = 0.f, l_run = 0.f;
    auto set_m = [&](float m) {
        const unsigned ph = vxb_pack_bf16(-m, 0.f);
        const unsigned p2 = vxb_pack_bf16(-m, -m - __uint_as_float(ph << 16));
        mslot = hi ? 0u : p2;
    };
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;

    const u16* kbase_g = g.kv + (long long)b * g.Nk * (2 * inner) + h * HD;
    const u16* vbase_g = kbase_g + inner;
    const unsigned rowb = 2u * 2u * (unsigned)inner;
    const unsigned planeb = (unsigned)(g.kv_plane * 2);
    const unsigned smem0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    int lkey[LPW];
    unsigned kchb[LPW], vchb[LPW];
#pragma unroll
    for (int t = 0; t < LPW; ++t) {
        lkey[t] = (wid + NW * t) * 8 + (lane >> 3);
        kchb[t] = (unsigned)(((lane & 7) ^ ((lkey[t] >> 1) & 7)) * 16);
        vchb[t] = (unsigned)(((lane & 7) ^ (4 * ((lkey[t] >> 1) & 1))) * 16);
    }
    auto issue_kv = [&](int ktk, int ktv) {                   // K(ktk) and V(ktv), both planes (ktv < 0: K only; ktk < 0: V only)
#pragma unroll
        for (int t = 0; t < LPW; ++t) {
            if (ktk >= 0) {
                const unsigned key = (unsigned)min(ktk * BKV + lkey[t], g.Nk - 1);
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    f2_load16(kbase_g, key * rowb + kchb[t] + p * planeb, smem0 + (unsigned)((((ktk % NS) * 2 + p) * TILE + (wid + NW * t) * 512) * 2));
            }
            if (ktv >= 0) {
                const unsigned key = (unsigned)min(ktv * BKV + lkey[t], g.Nk - 1);
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    f2_load16(vbase_g, key * rowb + vchb[t] + p * planeb, smem0 + (unsigned)((VOF + ((ktv % NS) * 2 + p) * TILE + (wid + NW * t) * 512) * 2));
            }
        }
    };
    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const unsigned row_id = (unsigned)bh * (unsigned)g.Nq + (unsigned)qrow;
    const unsigned rowh = row_id * 0x9E3779B1U + g.seed;
    int kbase[2], kkey[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int row = kb * 32 + lq; kbase[kb] = row * 64; kkey[kb] = (row >> 1) & 7; }
    const int t16 = lane & 15, gq = lane >> 4;
    const int vrow0 = 4 * (gq >> 1) + (t16 >> 2);
    const int vchunk0 = 2 * (gq & 1) + ((t16 & 3) >> 1), vhalf = (t16 & 1) * 4;
    int vlane[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) vlane[db] = vrow0 * 64 + ((vchunk0 + 4 * db) ^ (4 * ((vrow0 >> 1) & 1))) * 8 + vhalf;
    const int nkt = (g.Nk + BKV - 1) / BKV;

    auto k_frag = [&](int kt, int kb, int i, int p) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(smem + ((kt % NS) * 2 + p) * TILE + kbase[kb] + (((2 * i + hi) ^ kkey[kb]) * 8));
    };
    auto v_frag = [&](int kt, int kb, int i, int p) -> bf16x8 {             // V^T: d block i & 1, keys kb*32 + 16 (i >> 1) .. +16
        const u16* ad = smem + VOF + ((kt % NS) * 2 + p) * TILE + vlane[i & 1] + (kb * 32 + 16 * (i >> 1)) * 64;
        return f2_join(f2_tr16(ad), f2_tr16(ad + 8 * 64));
    };
    auto ones_frag = [&](int kt, int kb) -> bf16x8 {
        const unsigned tail = (kt * BKV + kb * 32 + lq >= g.Nk) ? one2 & 0xffffu : 0u;
        return f2_from4(one2, tail, 0u, 0u);
    };
    // element pair t = 0..7 of unit (kt, kb): P = exp2(S') (dropout applied) as hi | lo pairs, row-sum contributions
    auto pair = [&](int colb, int kb, int t, const f32x16& s, unsigned (&pc)[2][8], float (&ps)[4]) {
        const int r = 2 * t;
        float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
        f2_acc(ps[t & 3], p0);
        f2_acc(ps[(t + 2) & 3], p1);
        if (DROP) {
            const unsigned cp = (unsigned)((kb * 32 + (r & 3) + 8 * (r >> 2)) >> 1) * 0x85EBCA77U;
            const unsigned hsh = f2_hash(rowh ^ ((unsigned)colb + cp));
            p0 = (hsh & 0xffffu) >= thr ? p0 : 0.f;
            p1 = (hsh >> 16) >= thr ? p1 : 0.f;
        }
        pc[0][t] = vxb_pack_bf16(p0, p1);
        pc[1][t] = vxb_pack_bf16(p0 - __uint_as_float(pc[0][t] << 16), p1 - __uint_as_float(pc[0][t] & 0xffff0000u));
    };
    // One step.  Unit n = (ktn, kbn): its scores are formed (sn); unit c = (ktc, kbc): P from sc_ into pc, returns its row-sum part;
    // unit d = (ktd, kbd): O += V^T pp.  Fragments are requested one group ahead (two: 256 registers + spills).
    auto step = [&](auto has_n, auto has_c, auto has_d, int ktn, int kbn, int ktc, int kbc, int ktd, int kbd, f32x16& sn, const f32x16& sc_,
                    unsigned (&pc)[2][8], const unsigned (&pp)[2][8]) -> float {
        constexpr bool HN = decltype(has_n)::value, HC = decltype(has_c)::value, HD_ = decltype(has_d)::value;
        const int colb = (int)((unsigned)((ktc * BKV + 4 * hi) >> 1) * 0x85EBCA77U + 0xC2B2AE3DU);
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        bf16x8 kh[4], kl[4], vh[4], vl[4];
        auto load_group = [&](int i) {
            if (HN) { kh[i] = k_frag(ktn, kbn, i, 0); kl[i] = k_frag(ktn, kbn, i, 1); }
            if (HD_) { vh[i] = v_frag(ktd, kbd, i, 0); vl[i] = v_frag(ktd, kbd, i, 1); }
        };
        load_group(0);
        if (HN) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            sn = f2_mma<MODE>(ones_frag(ktn, kbn), f2_from4(mslot, big2, 0u, 0u), z);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i + 1 < 4) load_group(i + 1);
            if (HD_) {
                const int ks = i >> 1, db = i & 1;
                const bf16x8 ph = f2_from4(pp[0][4 * ks], pp[0][4 * ks + 1], pp[0][4 * ks + 2], pp[0][4 * ks + 3]);
                const bf16x8 pl = f2_from4(pp[1][4 * ks], pp[1][4 * ks + 1], pp[1][4 * ks + 2], pp[1][4 * ks + 3]);
                oacc[db] = f2_mma<MODE>(vl[i], ph, oacc[db]);
                oacc[db] = f2_mma<MODE>(vh[i], pl, oacc[db]);
                oacc[db] = f2_mma<MODE>(vh[i], ph, oacc[db]);
            }
            if (HC) pair(colb, kbc, 2 * i, sc_, pc, ps);
            __builtin_amdgcn_sched_barrier(0);
            if (HN) {
                sn = f2_mma<MODE>(kl[i], qfh[i], sn);
                sn = f2_mma<MODE>(kh[i], qfl[i], sn);
                sn = f2_mma<MODE>(kh[i], qfh[i], sn);
            }
            if (HC) pair(colb, kbc, 2 * i + 1, sc_, pc, ps);
            __builtin_amdgcn_sched_barrier(0);
        }
        return (ps[0] + ps[1]) + (ps[2] + ps[3]);
    };
    // rare: raise m by an integer (see the single-product kernel above); sa: the unit just exponentiated, sb: the next unit's scores
    auto raise_m = [&](int kt, int kb, f32x16& sa, f32x16& sb, unsigned (&pk)[2][8], float& psum) {
        float mt = sa[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sa[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float delta = fmaxf(ceilf(mt), 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; sa[r] -= delta; sb[r] -= delta; }
        l_run *= alpha;
        m_run += delta;
        set_m(m_run);
        const int colb = (int)((unsigned)((kt * BKV + 4 * hi) >> 1) * 0x85EBCA77U + 0xC2B2AE3DU);
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) pair(colb, kb, t, sa, pk, ps);
        psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    };
    const std::true_type T_;
    const std::false_type F_;

    // ---- prologue: unit 0's scores with m = 0 -> m = ceil(row maximum over the whole first TILE is not needed: unit 0 suffices)
    issue_kv(0, -1); issue_kv(1, 0); issue_kv(2, 1);
    f2_wait_vm<2 * NLOAD>();                                // K0 landed ({K1 V0} {K2 V1} may be in flight)
    vxb_raw_barrier();
    f32x16 s_[2];
    unsigned pb[2][2][8];
    float dummy;
    dummy = step(T_, F_, F_, 0, 0, 0, 0, 0, 0, s_[0], s_[0], pb[0], pb[0]);
    {
        float mt = s_[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s_[0][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        m_run = ceilf(mt);
        set_m(m_run);
#pragma unroll
        for (int r = 0; r < 16; ++r) s_[0][r] -= m_run;
    }
    // step 0: S'(0, 1), P(0, 0)
    {
        float ps = step(T_, T_, F_, 0, 1, 0, 0, 0, 0, s_[1], s_[0], pb[0], pb[0]);
        if (__builtin_amdgcn_ballot_w64(!(ps <= P_LIMIT))) raise_m(0, 0, s_[0], s_[1], pb[0], ps);
        l_run += ps;
    }
    (void)dummy;
    // ---- region j = steps 2j + 1, 2j + 2: needs K(j + 1), V(j).  Top: {K(j+1), V(j)} landed; issue {K(j+3), V(j+2)}:
    //      K stage (j+3) % 3 held K(j) (read in region j-1), V stage (j+2) % 3 held V(j-1) (read in region j-1)
    for (int j = 0; j < nkt; ++j) {
        f2_wait_vm<NLOAD>();
        vxb_raw_barrier();
        issue_kv(j + 3, j + 2);
        {
            float ps = step(T_, T_, T_, j + 1, 0, j, 1, j, 0, s_[0], s_[1], pb[1], pb[0]);
            if (__builtin_amdgcn_ballot_w64(!(ps <= P_LIMIT))) raise_m(j, 1, s_[1], s_[0], pb[1], ps);
            l_run += ps;
        }
        if (j + 1 < nkt) {
            float ps = step(T_, T_, T_, j + 1, 1, j + 1, 0, j, 1, s_[1], s_[0], pb[0], pb[1]);
            if (__builtin_amdgcn_ballot_w64(!(ps <= P_LIMIT))) raise_m(j + 1, 0, s_[0], s_[1], pb[0], ps);
            l_run += ps;
        } else {
            step(F_, F_, T_, 0, 0, 0, 0, j, 1, s_[1], s_[0], pb[0], pb[1]);        // the last unit's P V product
        }
    }
    f2_wait_vm<0>();

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (q_ok) {
        const float inv = (1.0f / (1.0f - g.p_drop)) / l_tot;
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

template <int NW>
int f3_launch(const F3Args& g, bool drop, hipStream_t st) {
    const size_t lds = (size_t)2 * 3 * 2 * TILE * sizeof(u16);
    const dim3 grid(g.nqb * g.B * g.H);
    if (drop) {
        if (hipFuncSetAttribute((const void*)flash2_fwd_x3_kernel<1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((flash2_fwd_x3_kernel<1, NW>), grid, dim3(NW * 64), lds, st, g);
    } else {
        if (hipFuncSetAttribute((const void*)flash2_fwd_x3_kernel<0, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((flash2_fwd_x3_kernel<0, NW>), grid, dim3(NW * 64), lds, st, g);
    }
    return VXB_OK;
}

template <int MODE, int NW>
int f2_launch(const F2Args& g, bool drop, hipStream_t st) {
    const size_t lds = (size_t)2 * NST * TILE * sizeof(u16);
    const dim3 grid(g.nqb * g.B * g.H);
    if (drop && g.mask) {
        if (hipFuncSetAttribute((const void*)flash2_fwd_kernel<MODE, 2, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((flash2_fwd_kernel<MODE, 2, NW>), grid, dim3(NW * 64), lds, st, g);
    } else if (drop) {
        if (hipFuncSetAttribute((const void*)flash2_fwd_kernel<MODE, 1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((flash2_fwd_kernel<MODE, 1, NW>), grid, dim3(NW * 64), lds, st, g);
    } else {
        if (hipFuncSetAttribute((const void*)flash2_fwd_kernel<MODE, 0, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((flash2_fwd_kernel<MODE, 0, NW>), grid, dim3(NW * 64), lds, st, g);
    }
    return VXB_OK;
}

// fp32 [rows][cols] -> fp16 planes [nplanes][rows][cols]: plane 0 = RNE(x) (saturated at +-65504), plane 1 = RNE(x - plane 0)
__global__ void __launch_bounds__(256) split_f16_kernel(const float* __restrict__ src, long long ld, long long rows, int cols,
                                                        u16* __restrict__ hi, u16* __restrict__ lo) {
    const int q = cols >> 2;
    const long long total = rows * q;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / q;
        const int c4 = (int)(i - r * q) * 4;
        float4 v = *reinterpret_cast<const float4*>(src + r * ld + c4);
        v.x = __builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f); v.y = __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f);
        v.z = __builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f); v.w = __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f);
        uint2 ph;
        ph.x = vxb_pack_f16(v.x, v.y); ph.y = vxb_pack_f16(v.z, v.w);
        *reinterpret_cast<uint2*>(hi + r * cols + c4) = ph;
        if (lo) {
            float a0, a1, a2, a3;
            f2_unpack<M_F16>(ph.x, a0, a1); f2_unpack<M_F16>(ph.y, a2, a3);
            uint2 pl;
            pl.x = vxb_pack_f16(v.x - a0, v.y - a1); pl.y = vxb_pack_f16(v.z - a2, v.w - a3);
            *reinterpret_cast<uint2*>(lo + r * cols + c4) = pl;
        }
    }
}

}  // namespace

// fp16 twin of vxb_split_bf16_f32
extern "C" int vxb_split_f16_f32(const float* src, int64_t ld, int64_t rows, int cols, void* dst_planes, int nplanes,
                                 vxb_stream_t stream) {
    if (!src || !dst_planes || rows < 1 || cols < 4 || (nplanes != 1 && nplanes != 2)) return VXB_EARG;
    if ((cols & 3) || (ld & 3) || (((uintptr_t)src) & 15) || (((uintptr_t)dst_planes) & 15)) return VXB_ESIZE;
    u16* hi = (u16*)dst_planes;
    u16* lo = nplanes == 2 ? hi + rows * cols : nullptr;
    const long long total = rows * (cols >> 2);
    const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(split_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (long long)ld, (long long)rows, cols, hi, lo);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// bytes of the dropout keep words one forward call with dropout writes (drop_mask of vxb_flash2_attn_fwd_mask; 256-byte aligned buffer):
// [B*H][8 ceil(Nq / 256)][ceil(Nk / 64)][64] 32-bit words (32-row blocks, rounded up to whole 256-row workgroups)
extern "C" size_t vxb_flash2_drop_mask_bytes(int B, int H, int Nq, int Nk) {
    if (B < 1 || H < 1 || Nq < 1 || Nk < 1) return 0;
    return (size_t)B * H * ((((size_t)Nq + 255) / 256) * 8) * (((size_t)Nk + BKV - 1) / BKV) * 64 * sizeof(unsigned) + 1024;      // (+ 1 KB: the dK | dV kernel fetches two key tiles at a time)
}

static int f2_fwd(const float* q, const void* kv_planes, int mode, float* o, float* lse, void* drop_mask, int B, int H, int Nq, int Nk,
                  int head_dim, float scale, float dropout_p, uint32_t seed, int waves, vxb_stream_t stream);

// Forward of the fused attention, pipelined structure.  mode 0: kv_planes = ONE bf16 plane [B*Nk][2*H*64]; mode 1: one fp16 plane.
// waves: 4 or 8 per workgroup (0 = choose).  Same outputs (o, lse) and the same dropout mask as vxb_flash_attn_fwd_dl.
extern "C" int vxb_flash2_attn_fwd(const float* q, const void* kv_planes, int mode, float* o, float* lse, int B, int H, int Nq, int Nk,
                                   int head_dim, float scale, float dropout_p, uint32_t seed, int waves, vxb_stream_t stream) {
    return f2_fwd(q, kv_planes, mode, o, lse, nullptr, B, H, Nq, Nk, head_dim, scale, dropout_p, seed, waves, stream);
}

// The same call that also WRITES the dropout keep words (modes 0 / 1, dropout_p > 0) for vxb_flash2_attn_bwd_mask.  NOT the mask of
// vxb_flash2_attn_fwd: with the mask handed on as data the forward draws it from a cheaper per-row sequence (see pair() above), so the
// backward of THIS forward must be vxb_flash2_attn_bwd_mask with these words (tests/test_flash2_gpu.py checks both against a float64
// attention that applies the stored mask).  Same (seed, shape) -> same mask, run to run.
extern "C" int vxb_flash2_attn_fwd_mask(const float* q, const void* kv_planes, int mode, float* o, float* lse, void* drop_mask, int B, int H,
                                        int Nq, int Nk, int head_dim, float scale, float dropout_p, uint32_t seed, int waves, vxb_stream_t stream) {
    if (!drop_mask || (((uintptr_t)drop_mask) & 15) || mode == 2) return VXB_EARG;
    return f2_fwd(q, kv_planes, mode, o, lse, drop_mask, B, H, Nq, Nk, head_dim, scale, dropout_p, seed, waves, stream);
}

static int f2_fwd(const float* q, const void* kv_planes, int mode, float* o, float* lse, void* drop_mask, int B, int H, int Nq, int Nk,
                  int head_dim, float scale, float dropout_p, uint32_t seed, int waves, vxb_stream_t stream) {
    if (!q || !kv_planes || !o || !lse || B < 1 || H < 1 || Nq < 1 || Nk < 1 || mode < 0 || mode > 2) return VXB_EARG;
    if (head_dim != HD || dropout_p < 0.f || dropout_p >= 1.f || (((uintptr_t)kv_planes) & 15)) return VXB_ESIZE;
    if (mode == 2) {                    // 'bf16x3': kv_planes = the hi | lo bf16 planes of vxb_split_bf16_f32, 8 waves share the rings
        if ((long long)Nk * 2 * H * HD * 2 * 2 > 0xffffffffLL) return VXB_ESIZE;
        F3Args g3;
        g3.q = q; g3.kv = (const u16*)kv_planes; g3.kv_plane = (long long)B * Nk * 2 * H * HD; g3.o = o; g3.lse = lse;
        g3.B = B; g3.H = H; g3.Nq = Nq; g3.Nk = Nk; g3.nqb = vxb_cdiv(Nq, 256); g3.scale = scale; g3.p_drop = dropout_p; g3.seed = seed;
        const int rc3 = f3_launch<8>(g3, (unsigned)(dropout_p * 65536.0f) > 0u, (hipStream_t)stream);
        if (rc3 != VXB_OK) return rc3;
        VXB_CHECK_LAUNCH();
        return VXB_OK;
    }
    if (waves == 0) waves = 4;          // two 4-wave workgroups per CU overlap each other's prologue / epilogue; 8 waves measured 3-30 % slower
    if (waves != 4 && waves != 8) return VXB_EARG;
    F2Args g;
    g.q = q; g.kv = (const u16*)kv_planes; g.o = o; g.lse = lse; g.mask = (unsigned*)drop_mask;
    g.B = B; g.H = H; g.Nq = Nq; g.Nk = Nk; g.nqb = vxb_cdiv(Nq, waves * 32); g.scale = scale; g.p_drop = dropout_p; g.seed = seed;
    const bool drop = (unsigned)(dropout_p * 65536.0f) > 0u;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((long long)Nk * 2 * H * HD * 2 * 2 > 0xffffffffLL) return VXB_ESIZE;          // 32-bit lane offsets inside one sample's k | v rows
    if (waves == 8) rc = mode == M_BF16 ? f2_launch<M_BF16, 8>(g, drop, st) : f2_launch<M_F16, 8>(g, drop, st);
    else rc = mode == M_BF16 ? f2_launch<M_BF16, 4>(g, drop, st) : f2_launch<M_F16, 4>(g, drop, st);
    if (rc != VXB_OK) return rc;
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
