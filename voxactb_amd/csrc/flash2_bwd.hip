// Fused attention backward, pipelined structure (round 4; forward: flash2_fwd.hip).  Two kernels, both built from "units" of
// 32 x 32 scores and steps in which three independent instruction streams of one wave overlap:
//
//      step s:   S'(s+1) = scores - lse,  T'(s+1) = c (V dO^T) - D      (matrix cores; both offsets folded into a fifth k-step)
//                dS(s)   = exp2(S'(s)) * (keep ? T'(s) : -D)             (vector ALU: exp2, select, multiply, pack)
//                dQ += K^T dS(s-1)          |  dV += P^T dO, dK += dS^T Q (matrix cores)
//
//  * dQ kernel: lane = query, units walk the key tiles (K | V tiles by direct-to-LDS loads, rings of four stages, counted vmcnt,
//    one barrier per 64-key tile); regions are cut so that only two tiles are live: region j = steps (j, 1) and (j + 1, 0).
//  * dK / dV kernel: lane = key, units walk the query tiles (Q and dO' tiles + per-row offset fragments by direct-to-LDS loads).
//
// Arithmetic (MODE 0 = bf16, 1 = fp16 operands): scores and probabilities exactly as the forward kernel forms them; the gradient
// operands dO and dS are single 16-bit values (GX = 0) or hi + lo pairs (GX = 1: two MFMAs per product, the rounding of the
// propagating gradient drops from 2^-12 to 2^-23).  dO is pre-multiplied by 2^k / (1 - p) with 2^k max|dO| in [16, 32) (fp16 range;
// bf16: k = 0): dS, dQ, dK, dV come out multiplied by 2^k and are divided again in the epilogues (exact).
// A preparation pass (f2b_prep_*) makes D = rowsum(dO * O), max|dO|, the 16-bit planes of q and dO' and, per query row, the offset
// fragment {-lse2 as three 16-bit parts, row-beyond-Nq flag, -2^k D as three parts}.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int HD = 64, BT = 64;
constexpr int TILE = BT * HD;               // u16 per plane tile (8 KB)
constexpr int NST = 4;
constexpr float LOG2E = 1.4426950408889634f;

enum { M_BF16 = 0, M_F16 = 1 };

struct F2bArgs {
    const float* q;        // [B, Nq, H*64] fp32
    const float* kv;       // [B, Nk, 2*H*64] fp32
    const float* d_o;      // [B, Nq, H*64] fp32
    const u16* kvp;        // [B*Nk][2*H*64]   16-bit plane of k | v (the forward's)
    const u16* qp;         // [B*Nq][H*64]     16-bit plane of q
    const u16* dop;        // [NG][B*Nq][H*64] 16-bit plane(s) of dO' = dO * 2^k / (1 - p)
    long long do_plane;    // u16 between the hi and the lo plane of dO'
    const unsigned* rowslots;   // [B*H][Nq64][4]
    const float* negd;     // [B*H][Nq64]  -2^k D
    const float* scale_ws; // [0] = 2^k, [1] = 2^-k
    const unsigned* mask;  // DROP == 2: the forward's dropout keep words (flash2_fwd.hip: f2_store_keep), else unused
    int nrb, ntile;        // 32-row blocks per (batch, head) and 64-key tiles per row block of that layout
    float* dq;
    float* dkv;
    int B, H, Nq, Nk, Nq64, nblk;
    float scale, p_drop;
    unsigned seed;
};

__device__ __forceinline__ unsigned fb_hash(unsigned x) {      // == f2_hash: the forward's dropout mask
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
    return x;
}
template <int MODE>
__device__ __forceinline__ unsigned fb_pack(float a, float b) {
    return MODE == M_BF16 ? vxb_pack_bf16(a, b) : vxb_pack_f16(a, b);
}
template <int MODE>
__device__ __forceinline__ void fb_unpack(unsigned p, float& a, float& b) {
    if (MODE == M_BF16) { a = __uint_as_float(p << 16); b = __uint_as_float(p & 0xffff0000u); }
    else {
        union { unsigned u; vxb_f16x2 h; } t; t.u = p;
        a = (float)t.h[0]; b = (float)t.h[1];
    }
}
template <int MODE>
__device__ __forceinline__ float fb_clamp(float x) { return MODE == M_BF16 ? x : __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }
template <int MODE>
__device__ __forceinline__ f32x16 fb_mma(const bf16x8 a, const bf16x8 b, f32x16 c) {
    if (MODE == M_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 fb_from4(unsigned a, unsigned b, unsigned c, unsigned d) {
    union { unsigned u[4]; bf16x8 v; } t;
    t.u[0] = a; t.u[1] = b; t.u[2] = c; t.u[3] = d;
    return t.v;
}
__device__ __forceinline__ bf16x8 fb_join(unsigned long long a, unsigned long long b) {
    union { unsigned long long u[2]; bf16x8 v; } t;
    t.u[0] = a; t.u[1] = b;
    return t.v;
}
typedef short fb_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long fb_tr16(const u16* p) {
    union { fb_v4s v; unsigned long long u; } t;
    t.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fb_v4s*)p);
    return t.u;
}
// direct-to-LDS loads as inline asm (see flash2_fwd.hip: the compiler must not know about them); 16 or 4 bytes per lane
__device__ __forceinline__ void fb_load16(const void* base, unsigned byte_off, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(byte_off), "s"(base), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ void fb_load4(const void* base, unsigned byte_off, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(byte_off), "s"(base), "s"(lds_wave_base) : "memory");
}
template <int N>
__device__ __forceinline__ void fb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int fb_swz(int row) {           // chunk XOR key of a tile row: conflict-free for ds_read_b128 AND transposed reads
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x >> 1);
}
// x as three 16-bit parts (hi + mid + lo == x to 2^-33 |x| in fp16, 2^-24 in bf16)
template <int MODE>
__device__ __forceinline__ void fb_split3(float x, unsigned& w01, unsigned& w2) {
    x = fb_clamp<MODE>(x);
    const unsigned ph = fb_pack<MODE>(x, 0.f);
    float h, z;
    fb_unpack<MODE>(ph, h, z);
    const float r1 = x - h;
    const unsigned pm = fb_pack<MODE>(r1, 0.f);
    float m;
    fb_unpack<MODE>(pm, m, z);
    w01 = (ph & 0xffffu) | (pm << 16);
    w2 = fb_pack<MODE>(r1 - m, 0.f) & 0xffffu;
}

// DROP == 2: the forward's keep words (flash2_fwd.hip: f2_store_keep) instead of the hash.  Both kernels take them through LDS with their
// tile loads and test ONE bit per score: dQ (lane = query) word (key) bit (own row), dK | dV (lane = key) word (own key) bit (row).  (The
// first version of the dQ kernel loaded the lane masks by scalar loads straight into v_cndmask -- one instruction per score instead of
// two -- and was no faster than hashing: a scalar load can only be waited for with lgkmcnt(0), which drains the LDS fragment reads in
// flight; 48.8 % of its time in s_waitcnt, profiles/r06_v1_sq_summary.txt.)

// ------------------------------------------------------------------------------------------------ preparation
// D[bh][q] = sum_d dO O, and the largest |dO| (magnitude bits, atomicMax) -- one 16-lane group per (row, head)
__global__ void __launch_bounds__(256) f2b_prep1_kernel(const float* __restrict__ d_o, const float* __restrict__ o, float* __restrict__ dsum,
                                                        unsigned* __restrict__ amax, int B, int H, int Nq, int Nq64) {
    __shared__ float s_mx[4];
    const int sub = threadIdx.x & 15;
    const long long total = (long long)B * Nq * H;
    float mx = 0.f, nanw = 0.f;             // nanw: NaN once a NaN / inf of dO was seen (fmaxf drops NaN)
    for (long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4; idx < total; idx += (long long)gridDim.x * 16) {
        const float4 a = *reinterpret_cast<const float4*>(d_o + idx * HD + sub * 4);
        const float4 c = *reinterpret_cast<const float4*>(o + idx * HD + sub * 4);
        float s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        nanw = fmaf((a.x + a.y) + (a.z + a.w), 0.0f, nanw);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (sub == 0) {
            const int h = (int)(idx % H);
            const long long bq = idx / H;
            dsum[((long long)(bq / Nq) * H + h) * Nq64 + (int)(bq % Nq)] = s;
        }
    }
    mx = wave_max(mx);                      // (one atomic per block: per-wave atomics on one word cost 0.7 ms at 65 k waves)
    nanw = wave_sum(nanw);
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = nanw != nanw ? INFINITY : mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        // a non-finite dO: the word becomes NaN bits and f2b_prep2 turns the operand scale into NaN -- dQ, dK and dV come out NaN as
        // the reference's autograd would hand them on (fb_clamp alone would saturate the NaN to a finite value)
        if (mx > 0.f) atomicMax(amax, mx < INFINITY ? __float_as_uint(mx) : 0x7fc00000u);
    }
}
// scale words, the row fragments and the 16-bit planes of q and dO'
template <int MODE, int GX>
__global__ void __launch_bounds__(256) f2b_prep2_kernel(const float* __restrict__ q, const float* __restrict__ d_o, const float* __restrict__ lse,
                                                        const float* __restrict__ dsum, const unsigned* __restrict__ amax,
                                                        float* __restrict__ scale_ws, unsigned* __restrict__ rowslots, float* __restrict__ negd,
                                                        u16* __restrict__ qp, u16* __restrict__ dop, long long do_plane,
                                                        int B, int H, int Nq, int Nq64, float p_drop) {
    float sc = 1.f;
    if (MODE == M_F16) {
        const unsigned a = *amax;                       // 0 (all-zero dO) -> 1
        if (a) {
            const int e = (int)((a >> 23) & 0xff) - 127;            // floor(log2 max)
            sc = __uint_as_float((unsigned)(127 + 4 - e) << 23);     // max * sc in [16, 32)
            if (a >= 0x7f800000u) sc = __uint_as_float(0x7fc00000u);  // non-finite dO (f2b_prep1): NaN through every product and the 1 / sc epilogue
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale_ws[0] = sc; scale_ws[1] = 1.0f / sc; }
    const float c1 = sc / (1.0f - p_drop);
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    // planes: 4 values per thread
    const long long nvec = (long long)B * Nq * H * (HD / 4);
    if (gid < nvec) {
        const float4 a = *reinterpret_cast<const float4*>(q + gid * 4);
        uint2 w;
        w.x = fb_pack<MODE>(fb_clamp<MODE>(a.x), fb_clamp<MODE>(a.y)); w.y = fb_pack<MODE>(fb_clamp<MODE>(a.z), fb_clamp<MODE>(a.w));
        *reinterpret_cast<uint2*>(qp + gid * 4) = w;
        const float4 d = *reinterpret_cast<const float4*>(d_o + gid * 4);
        const float x0 = d.x * c1, x1 = d.y * c1, x2 = d.z * c1, x3 = d.w * c1;
        uint2 hw;
        hw.x = fb_pack<MODE>(x0, x1); hw.y = fb_pack<MODE>(x2, x3);
        *reinterpret_cast<uint2*>(dop + gid * 4) = hw;
        if (GX) {
            float h0, h1, h2, h3;
            fb_unpack<MODE>(hw.x, h0, h1); fb_unpack<MODE>(hw.y, h2, h3);
            uint2 lw;
            lw.x = fb_pack<MODE>(x0 - h0, x1 - h1); lw.y = fb_pack<MODE>(x2 - h2, x3 - h3);
            *reinterpret_cast<uint2*>(dop + do_plane + gid * 4) = lw;
        }
    }
    // row fragments (padded rows: the beyond-Nq flag)
    const long long nrow = (long long)B * H * Nq64;
    if (gid < nrow) {
        const int qq = (int)(gid % Nq64);
        const long long bh = gid / Nq64;
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        float nd = 0.f;
        if (qq < Nq) {
            unsigned a01, a2, d01, d2;
            fb_split3<MODE>(-lse[bh * Nq + qq] * LOG2E, a01, a2);
            nd = -sc * dsum[gid];
            fb_split3<MODE>(nd, d01, d2);
            w = make_uint4(a01, a2, d01, d2);
        } else {
            w.y = fb_pack<MODE>(0.f, MODE == M_BF16 ? -3.0e38f : -60000.f);
        }
        *reinterpret_cast<uint4*>(rowslots + gid * 4) = w;
        negd[gid] = nd;
    }
}

// ------------------------------------------------------------------------------------------------ dQ
template <int MODE, int GX, int DROP, int NW>
__global__ void __launch_bounds__(NW * 64, 2) f2b_dq_kernel(F2bArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int NG = 1 + GX;
    constexpr int LPW = 8 / NW;
    constexpr int NLOAD = 2 * LPW + (DROP == 2 ? 1 : 0);      // K and V tile pieces (+ this wave's 64 keep words of the tile) per wave
    constexpr int VOFF = NST * TILE;
    constexpr int MOFF = 2 * NST * TILE;            // DROP == 2: keep words, [stage][wave][64 words] (u16 offset)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int vb = blockIdx.x;
    {
        const int total = gridDim.x;
        if ((total & 7) == 0) vb = (vb & 7) * (total >> 3) + (vb >> 3);
    }
    const int bh = vb / g.nblk, qblk = vb - bh * g.nblk;
    const int b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int qrow = qblk * (NW * 32) + wid * 32 + lq;
    const bool q_ok = qrow < g.Nq;
    const long long qoff = ((long long)b * g.Nq + (q_ok ? qrow : 0)) * inner + h * HD;
    const float qs = g.scale * LOG2E;
    const float sc = g.scale_ws[0];
    const float c1 = sc / (1.0f - g.p_drop);

    bf16x8 qf[4], dof[NG][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(g.q + qoff + 16 * ks + 8 * hi);
        const float4 c = *reinterpret_cast<const float4*>(g.q + qoff + 16 * ks + 8 * hi + 4);
        qf[ks] = fb_from4(fb_pack<MODE>(fb_clamp<MODE>(a.x * qs), fb_clamp<MODE>(a.y * qs)), fb_pack<MODE>(fb_clamp<MODE>(a.z * qs), fb_clamp<MODE>(a.w * qs)),
                          fb_pack<MODE>(fb_clamp<MODE>(c.x * qs), fb_clamp<MODE>(c.y * qs)), fb_pack<MODE>(fb_clamp<MODE>(c.z * qs), fb_clamp<MODE>(c.w * qs)));
        const float4 d0 = *reinterpret_cast<const float4*>(g.d_o + qoff + 16 * ks + 8 * hi);
        const float4 d1 = *reinterpret_cast<const float4*>(g.d_o + qoff + 16 * ks + 8 * hi + 4);
        const float v[8] = {d0.x * c1, d0.y * c1, d0.z * c1, d0.w * c1, d1.x * c1, d1.y * c1, d1.z * c1, d1.w * c1};
        unsigned ph[4], pl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ph[i] = fb_pack<MODE>(v[2 * i], v[2 * i + 1]);
            if (GX) {
                float h0, h1;
                fb_unpack<MODE>(ph[i], h0, h1);
                pl[i] = fb_pack<MODE>(v[2 * i] - h0, v[2 * i + 1] - h1);
            }
        }
        dof[0][ks] = fb_from4(ph[0], ph[1], ph[2], ph[3]);
        if (GX) dof[NG - 1][ks] = fb_from4(pl[0], pl[1], pl[2], pl[3]);
    }
    // offset fragments of this lane's query (hi = 0 lanes), K-side ones
    const long long rsi = (long long)bh * g.Nq64 + min(qrow, g.Nq64 - 1);
    const uint4 rs = *reinterpret_cast<const uint4*>(g.rowslots + rsi * 4);
    const float negd = g.negd[rsi];
    const unsigned bigw = fb_pack<MODE>(0.f, MODE == M_BF16 ? -3.0e38f : -60000.f) & 0xffff0000u;
    const bf16x8 slot_s = hi ? fb_from4(0u, 0u, 0u, 0u) : fb_from4(rs.x, (rs.y & 0xffffu) | bigw, 0u, 0u);
    const bf16x8 slot_t = hi ? fb_from4(0u, 0u, 0u, 0u) : fb_from4(rs.z, rs.w, 0u, 0u);
    const unsigned one2 = hi ? 0u : fb_pack<MODE>(1.f, 1.f);

    f32x16 dqacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

    // ---- tile loads
    const u16* kbase_g = g.kvp + (long long)b * g.Nk * (2 * inner) + h * HD;
    const u16* vbase_g = kbase_g + inner;
    const unsigned rowb = 2u * 2u * (unsigned)inner;
    const unsigned smem0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    int lkey[LPW];
    unsigned chb[LPW];
#pragma unroll
    for (int t = 0; t < LPW; ++t) {
        lkey[t] = (wid + NW * t) * 8 + (lane >> 3);
        chb[t] = (unsigned)(((lane & 7) ^ fb_swz(lkey[t])) * 16);
    }
    // keep words of this wave's 32 rows (DROP == 2): 64 words per key tile, [kb][r][half]
    const unsigned* mwave_g = DROP == 2 ? g.mask + ((size_t)bh * g.nrb + (size_t)(qblk * NW + wid)) * (size_t)g.ntile * 64 : nullptr;
    const int nkt_ = (g.Nk + BT - 1) / BT;
    auto issue = [&](int kt) {
#pragma unroll
        for (int t = 0; t < LPW; ++t) {
            const unsigned key = (unsigned)min(kt * BT + lkey[t], g.Nk - 1);
            fb_load16(kbase_g, key * rowb + chb[t], smem0 + (unsigned)(((kt & 3) * TILE + (wid + NW * t) * 512) * 2));
            fb_load16(vbase_g, key * rowb + chb[t], smem0 + (unsigned)((VOFF + (kt & 3) * TILE + (wid + NW * t) * 512) * 2));
        }
        // this wave's keep words of the tile: 64 words, lane l -> word l (the words travel through LDS, not by scalar loads: a scalar
        // load's result can only be waited for with lgkmcnt(0), which also drains the LDS fragment reads in flight -- measured: the dQ
        // kernel with scalar-loaded lane masks spent 48.8 % of its time in s_waitcnt and was no faster than hashing the mask)
        if (DROP == 2) fb_load4(mwave_g, (unsigned)((min(kt, nkt_ - 1) * 64 + lane) * 4), smem0 + (unsigned)((MOFF + ((kt & 3) * NW + wid) * 128) * 2));
    };
    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const unsigned row_id = (unsigned)bh * (unsigned)g.Nq + (unsigned)qrow;
    const unsigned rowh = row_id * 0x9E3779B1U + g.seed;
    int rbase[2], rkey[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int row = kb * 32 + lq; rbase[kb] = row * 64; rkey[kb] = fb_swz(row); }
    const int t16 = lane & 15, gq = lane >> 4;
    const int trow0 = 4 * (gq >> 1) + (t16 >> 2);
    const int tchunk0 = 2 * (gq & 1) + ((t16 & 3) >> 1), thalf = (t16 & 1) * 4;
    int tlane[2][2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
            tlane[db][rr] = trow0 * 64 + ((tchunk0 + 4 * db) ^ fb_swz(8 * rr + trow0)) * 8 + thalf;

    const int nkt = (g.Nk + BT - 1) / BT;

    auto row_frag = [&](int ring_off, int kt, int kb, int ks) -> bf16x8 {        // K or V rows kb*32 + lq, d = 16 ks + 8 hi .. +8
        return *reinterpret_cast<const bf16x8*>(smem + ring_off + (kt & 3) * TILE + rbase[kb] + (((2 * ks + hi) ^ rkey[kb]) * 8));
    };
    auto kt_frag = [&](int kt, int kb, int i) -> bf16x8 {                        // K^T: d block i & 1, keys kb*32 + 16 (i >> 1) .. +16
        const u16* ad = smem + (kt & 3) * TILE + (kb * 32 + 16 * (i >> 1)) * 64;
        return fb_join(fb_tr16(ad + tlane[i & 1][0]), fb_tr16(ad + tlane[i & 1][1] + 8 * 64));
    };
    auto ones_frag = [&](int kt, int kb) -> bf16x8 {
        const unsigned w1 = (kt * BT + kb * 32 + lq >= g.Nk) ? one2 : one2 & 0xffffu;        // (1, tail)
        return fb_from4(one2, w1, 0u, 0u);
    };
    // element pair t = 0..7 of a unit: dS' = exp2(S') * (keep ? T' : -D'), packed hi (| lo)
    auto pair = [&](int colb, const unsigned (&munit)[16], int kb, int t, const f32x16& s, const f32x16& tt, unsigned (&ds)[NG][8]) {
        const int r = 2 * t;
        const float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
        float t0 = tt[r], t1 = tt[r + 1];
        if (DROP == 2) {
            // munit[k] (per lane) = word 2 k + hi of the unit: this lane half's key of register k; bit lq = this row
            int m0 = __builtin_amdgcn_sbfe((int)munit[r], lq, 1), m1 = __builtin_amdgcn_sbfe((int)munit[r + 1], lq, 1);
            asm("" : "+v"(m0), "+v"(m1));                   // (opaque 0 / -1: keeps the bit selects below as v_bfi)
            t0 = __int_as_float((m0 & __float_as_int(t0)) | (~m0 & __float_as_int(negd)));
            t1 = __int_as_float((m1 & __float_as_int(t1)) | (~m1 & __float_as_int(negd)));
        } else if (DROP) {
            const unsigned cp = (unsigned)((kb * 32 + (r & 3) + 8 * (r >> 2)) >> 1) * 0x85EBCA77U;
            const unsigned hsh = fb_hash(rowh ^ ((unsigned)colb + cp));
            t0 = (hsh & 0xffffu) >= thr ? t0 : negd;
            t1 = (hsh >> 16) >= thr ? t1 : negd;
        }
        const float s0 = fb_clamp<MODE>(p0 * t0), s1 = fb_clamp<MODE>(p1 * t1);
        ds[0][t] = fb_pack<MODE>(s0, s1);
        if (GX) {
            float h0, h1;
            fb_unpack<MODE>(ds[0][t], h0, h1);
            ds[NG - 1][t] = fb_pack<MODE>(s0 - h0, s1 - h1);
        }
    };
    // One step (pinned issue order).  Unit n = (ktn, kbn): its S' and T' are formed; unit c = (ktc, kbc): its dS from (sc_, tc_);
    // unit d = (ktd, kbd): dQ += K^T dS_prev.  Fragments are requested two groups ahead.
    auto step = [&](auto has_n, auto has_c, auto has_d, int ktn, int kbn, int ktc, int kbc, int ktd, int kbd, f32x16& sn, f32x16& tn,
                    const f32x16& sc_, const f32x16& tc_, unsigned (&dsc)[NG][8], const unsigned (&dsp)[NG][8]) {
        constexpr bool HN = decltype(has_n)::value, HC = decltype(has_c)::value, HD_ = decltype(has_d)::value;
        const int colb = (int)((unsigned)((ktc * BT + 4 * hi) >> 1) * 0x85EBCA77U + 0xC2B2AE3DU);
        unsigned munit[16];                                 // the unit's keep words of this lane half, read with the step's first fragments
        if (DROP == 2 && HC) {
            const unsigned* const mp = reinterpret_cast<const unsigned*>(smem + MOFF + ((ktc & 3) * NW + wid) * 128) + kbc * 32 + hi;
#pragma unroll
            for (int k = 0; k < 16; ++k) munit[k] = mp[2 * k];
        }
        bf16x8 kf[4], vf[4], tf[4];
        auto load_group = [&](int i) {
            if (HN) { kf[i] = row_frag(0, ktn, kbn, i); vf[i] = row_frag(VOFF, ktn, kbn, i); }
            if (HD_) tf[i] = kt_frag(ktd, kbd, i);
        };
        load_group(0); load_group(1);
        if (HN) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            const bf16x8 on = ones_frag(ktn, kbn);
            sn = fb_mma<MODE>(on, slot_s, z);
            tn = fb_mma<MODE>(on, slot_t, z);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i + 2 < 4) load_group(i + 2);
            if (HD_) {
                const int ks = i >> 1, db = i & 1;
#pragma unroll
                for (int p = 0; p < NG; ++p)
                    dqacc[db] = fb_mma<MODE>(tf[i], fb_from4(dsp[p][4 * ks], dsp[p][4 * ks + 1], dsp[p][4 * ks + 2], dsp[p][4 * ks + 3]), dqacc[db]);
            }
            if (HC) pair(colb, munit, kbc, 2 * i, sc_, tc_, dsc);
            __builtin_amdgcn_sched_barrier(0);
            if (HN) {
                sn = fb_mma<MODE>(kf[i], qf[i], sn);
#pragma unroll
                for (int p = 0; p < NG; ++p) tn = fb_mma<MODE>(vf[i], dof[p][i], tn);
            }
            if (HC) pair(colb, munit, kbc, 2 * i + 1, sc_, tc_, dsc);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const std::true_type T_;
    const std::false_type F_;

    // ---- units u = 2 kt + kb; step u forms S'T'(u + 1), dS(u), dQ(u - 1).  Prologue = "step -1" and step 0 (tile 0 only);
    //      region j = steps 2j + 1, 2j + 2 (tiles j and j + 1).  Buffers: S'T'(u) in s_[u & 1], dS(u) in dsb[u & 1].
    issue(0); issue(1); issue(2);
    fb_wait_vm<2 * NLOAD>();
    vxb_raw_barrier();
    f32x16 s_[2], t_[2];
    unsigned dsb[2][NG][8];
    step(T_, F_, F_, 0, 0, 0, 0, 0, 0, s_[0], t_[0], s_[0], t_[0], dsb[0], dsb[0]);
    step(T_, T_, F_, 0, 1, 0, 0, 0, 0, s_[1], t_[1], s_[0], t_[0], dsb[0], dsb[0]);
    for (int j = 0; j < nkt; ++j) {
        fb_wait_vm<NLOAD>();                                // tiles <= j + 1 landed; tile j + 2 may be in flight
        vxb_raw_barrier();
        issue(j + 3);
        step(T_, T_, T_, j + 1, 0, j, 1, j, 0, s_[0], t_[0], s_[1], t_[1], dsb[1], dsb[0]);
        step(T_, T_, T_, j + 1, 1, j + 1, 0, j, 1, s_[1], t_[1], s_[0], t_[0], dsb[0], dsb[1]);
    }
    fb_wait_vm<0>();
    if (q_ok) {
        const float f = g.scale * g.scale_ws[1];
        float* op = g.dq + qoff;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4*>(op + db * 32 + 8 * r4 + 4 * hi) =
                    make_float4(dqacc[db][4 * r4] * f, dqacc[db][4 * r4 + 1] * f, dqacc[db][4 * r4 + 2] * f, dqacc[db][4 * r4 + 3] * f);
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// lane = key.  Units u = 2 qt + qb walk the 32-query blocks of the 64-query tiles; step u forms S'T'(u + 1) (A = Q / dO' rows and the
// per-row offset fragment from LDS, B = K^T (x scale log2e) / V^T register fragments), the P and dS of unit u, and
// dV += P^T dO', dK += dS^T Q of unit u - 1 (B = dO' / Q by transposed reads of the same tiles).  Region j = steps 2j + 1, 2j + 2
// (tiles j, j + 1); rings of three stages, tile j + 2 is requested at the top of region j.
template <int MODE, int GX, int DROP, int NW>
__global__ void __launch_bounds__(NW * 64, GX ? 1 : 2) f2b_dkv_kernel(F2bArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int NG = 1 + GX;
    constexpr int NS = 3;
    constexpr int LPW = 8 / NW;
    constexpr int QOFF = 0, DOFF = NS * TILE, ROFF = DOFF + NS * NG * TILE;       // u16 offsets: Q ring | dO' ring | row fragments | -D' | zeros
    constexpr int NOFF = ROFF + NS * 512, ZOFF = NOFF + NS * 128;
    constexpr int MOFF = ZOFF + 16;                 // DROP == 2: keep words of the tile's two 32-row blocks x this workgroup's 128 keys, 1 KB per stage
    constexpr int PF = GX ? 1 : 2;                  // fragment groups requested ahead (registers)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lk = lane & 31;
    int vb = blockIdx.x;
    {
        const int total = gridDim.x;
        if ((total & 7) == 0) vb = (vb & 7) * (total >> 3) + (vb >> 3);
    }
    const int bh = vb / g.nblk, kblk = vb - bh * g.nblk;
    const int b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int key = kblk * (NW * 32) + wid * 32 + lk;
    const bool k_ok = key < g.Nk;
    const long long koff = ((long long)b * g.Nk + (k_ok ? key : 0)) * 2 * inner + h * HD;
    const float qs = g.scale * LOG2E;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(g.kv + koff + 16 * ks + 8 * hi);
        const float4 c = *reinterpret_cast<const float4*>(g.kv + koff + 16 * ks + 8 * hi + 4);
        kf[ks] = fb_from4(fb_pack<MODE>(fb_clamp<MODE>(a.x * qs), fb_clamp<MODE>(a.y * qs)), fb_pack<MODE>(fb_clamp<MODE>(a.z * qs), fb_clamp<MODE>(a.w * qs)),
                          fb_pack<MODE>(fb_clamp<MODE>(c.x * qs), fb_clamp<MODE>(c.y * qs)), fb_pack<MODE>(fb_clamp<MODE>(c.z * qs), fb_clamp<MODE>(c.w * qs)));
        const float4 d = *reinterpret_cast<const float4*>(g.kv + koff + inner + 16 * ks + 8 * hi);
        const float4 e = *reinterpret_cast<const float4*>(g.kv + koff + inner + 16 * ks + 8 * hi + 4);
        vf[ks] = fb_from4(fb_pack<MODE>(fb_clamp<MODE>(d.x), fb_clamp<MODE>(d.y)), fb_pack<MODE>(fb_clamp<MODE>(d.z), fb_clamp<MODE>(d.w)),
                          fb_pack<MODE>(fb_clamp<MODE>(e.x), fb_clamp<MODE>(e.y)), fb_pack<MODE>(fb_clamp<MODE>(e.z), fb_clamp<MODE>(e.w)));
    }
    const unsigned one2 = hi ? 0u : fb_pack<MODE>(1.f, 1.f);
    const bf16x8 ones = fb_from4(one2, one2, 0u, 0u);
    f32x16 dkacc[2], dvacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[i][r] = 0.f; dvacc[i][r] = 0.f; }

    // ---- tile loads: Q tile and dO' plane tile(s) (8 pieces of 1 KB each), the row fragments (1 KB) and -D' (256 B) of the 64 rows
    const u16* qbase_g = g.qp + (long long)b * g.Nq * inner + h * HD;
    const u16* dbase_g = g.dop + (long long)b * g.Nq * inner + h * HD;
    const unsigned* rs_g = g.rowslots + (long long)bh * g.Nq64 * 4;
    const float* nd_g = g.negd + (long long)bh * g.Nq64;
    const unsigned* mk_g = DROP == 2 ? g.mask + ((size_t)bh * g.nrb * (size_t)g.ntile + (size_t)kblk * (NW * 32 / BT)) * 64 : nullptr;
    // this lane's word inside a row block's 128: key kk = wid * 32 + lk of the workgroup -> [tile kk / 64][kb][r][half] (flash2_fwd.hip)
    const int mword = (wid >> 1) * 64 + (wid & 1) * 32 + (((lk & 3) | ((lk >> 3) << 2)) << 1) + ((lk >> 2) & 1);
    const unsigned rowb = 2u * (unsigned)inner;
    const unsigned planeb = (unsigned)(g.do_plane * 2);
    const unsigned smem0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    int lrow[LPW];
    unsigned chb[LPW];
#pragma unroll
    for (int t = 0; t < LPW; ++t) {
        lrow[t] = (wid + NW * t) * 8 + (lane >> 3);
        chb[t] = (unsigned)(((lane & 7) ^ fb_swz(lrow[t])) * 16);
    }
    const int nqt = (g.Nq + BT - 1) / BT;
    auto issue = [&](int qt) {
        const int st = qt % NS;
#pragma unroll
        for (int t = 0; t < LPW; ++t) {
            const unsigned row = (unsigned)min(qt * BT + lrow[t], g.Nq - 1);
            fb_load16(qbase_g, row * rowb + chb[t], smem0 + (unsigned)((QOFF + st * TILE + (wid + NW * t) * 512) * 2));
#pragma unroll
            for (int p = 0; p < NG; ++p)
                fb_load16(dbase_g, row * rowb + chb[t] + p * planeb, smem0 + (unsigned)((DOFF + (st * NG + p) * TILE + (wid + NW * t) * 512) * 2));
        }
        const int qc = min(qt, nqt - 1) * BT;
        if (wid == 0) fb_load16(rs_g, (unsigned)((qc + lane) * 16), smem0 + (unsigned)((ROFF + st * 512) * 2));
        if (wid == 1 % NW) fb_load4(nd_g, (unsigned)((qc + lane) * 4), smem0 + (unsigned)((NOFF + st * 128) * 2));
        // keep words: row blocks 2 qt, 2 qt + 1 (lanes 0 - 31 | 32 - 63) x the 128 words of this workgroup's two key tiles
        if (DROP == 2 && wid == 2 % NW)
            fb_load16(mk_g, (unsigned)(((2 * min(qt, nqt - 1) + (lane >> 5)) * g.ntile) * 256 + (lane & 31) * 16), smem0 + (unsigned)((MOFF + st * 512) * 2));
    };
    if (tid < 8) reinterpret_cast<unsigned*>(smem + ZOFF)[tid] = 0u;          // 32 bytes of zeros: the offset fragment of the hi = 1 lanes

    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const unsigned kodd = (unsigned)key & 1u, hshift = kodd ? 0u : 16u, thr16 = thr << 16;
    const unsigned colc = ((unsigned)key >> 1) * 0x85EBCA77U + 0xC2B2AE3DU;
    int rbase[2], rkey[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) { const int row = qb * 32 + lk; rbase[qb] = row * 64; rkey[qb] = fb_swz(row); }
    const int t16 = lane & 15, gq = lane >> 4;
    const int trow0 = 4 * (gq >> 1) + (t16 >> 2);
    const int tchunk0 = 2 * (gq & 1) + ((t16 & 3) >> 1), thalf = (t16 & 1) * 4;
    int tlane[2][2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
            tlane[db][rr] = trow0 * 64 + ((tchunk0 + 4 * db) ^ fb_swz(8 * rr + trow0)) * 8 + thalf;

    auto row_frag = [&](int off, int qb, int ks) -> bf16x8 {                       // rows qb*32 + lk of the tile at u16 offset `off`
        return *reinterpret_cast<const bf16x8*>(smem + off + rbase[qb] + (((2 * ks + hi) ^ rkey[qb]) * 8));
    };
    auto tr_frag = [&](int off, int qb, int i) -> bf16x8 {                         // [16 queries qb*32 + 16 (i >> 1) ..][d block i & 1]
        const u16* ad = smem + off + (qb * 32 + 16 * (i >> 1)) * 64;
        return fb_join(fb_tr16(ad + tlane[i & 1][0]), fb_tr16(ad + tlane[i & 1][1] + 8 * 64));
    };
    // element pair t of unit (qt, qb): rows ql, ql + 1 of this lane's key
    auto pair = [&](int qt, int qb, int mw, int t, const f32x16& s, const f32x16& tt, const float* nd_tile, unsigned (&pp)[8], unsigned (&ds)[NG][8]) {
        const int r = 2 * t;
        const float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
        float t0 = tt[r], t1 = tt[r + 1], pd0 = p0, pd1 = p1;
        if (DROP == 2) {
            // mw: the keep word of (row block, this lane's key) shifted right by 4 hi -> bit (r & 3) + 8 (r >> 2) is row r of this lane
            int m0 = __builtin_amdgcn_sbfe(mw, (r & 3) + 8 * (r >> 2), 1), m1 = __builtin_amdgcn_sbfe(mw, ((r + 1) & 3) + 8 * ((r + 1) >> 2), 1);
            asm("" : "+v"(m0), "+v"(m1));                   // (opaque 0 / -1: otherwise the bit selects below become compare + v_cndmask pairs, six instructions per score)
            const float2 nd = *reinterpret_cast<const float2*>(nd_tile + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
            t0 = __int_as_float((m0 & __float_as_int(t0)) | (~m0 & __float_as_int(nd.x)));
            t1 = __int_as_float((m1 & __float_as_int(t1)) | (~m1 & __float_as_int(nd.y)));
            pd0 = __int_as_float(m0 & __float_as_int(p0));
            pd1 = __int_as_float(m1 & __float_as_int(p1));
        } else if (DROP) {
            const unsigned row0 = (unsigned)bh * (unsigned)g.Nq + (unsigned)(qt * BT + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
            // the mask word of (row, key pair) covers this lane's key and its neighbour's: the even lane hashes row0, the odd lane
            // row0 + 1, and the two swap words (one DPP move) instead of hashing both
            const unsigned hm = fb_hash(((row0 + kodd) * 0x9E3779B1U + g.seed) ^ colc);
            const unsigned ho = (unsigned)__builtin_amdgcn_mov_dpp((int)hm, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
            const unsigned h0 = kodd ? ho : hm, h1 = kodd ? hm : ho;
            const bool k0 = (h0 << hshift) >= thr16, k1 = (h1 << hshift) >= thr16;
            const float2 nd = *reinterpret_cast<const float2*>(nd_tile + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);      // -D' of rows ql, ql + 1
            t0 = k0 ? t0 : nd.x; t1 = k1 ? t1 : nd.y;
            pd0 = k0 ? p0 : 0.f; pd1 = k1 ? p1 : 0.f;
        }
        pp[t] = fb_pack<MODE>(pd0, pd1);
        const float s0 = fb_clamp<MODE>(p0 * t0), s1 = fb_clamp<MODE>(p1 * t1);
        ds[0][t] = fb_pack<MODE>(s0, s1);
        if (GX) {
            float h0, h1;
            fb_unpack<MODE>(ds[0][t], h0, h1);
            ds[NG - 1][t] = fb_pack<MODE>(s0 - h0, s1 - h1);
        }
    };
    auto step = [&](auto has_n, auto has_c, auto has_d, int qtn, int qbn, int qtc, int qbc, int qtd, int qbd, f32x16& sn, f32x16& tn,
                    const f32x16& sc_, const f32x16& tc_, unsigned (&ppc)[8], unsigned (&dsc)[NG][8], const unsigned (&ppp)[8],
                    const unsigned (&dsp)[NG][8]) {
        constexpr bool HN = decltype(has_n)::value, HC = decltype(has_c)::value, HD_ = decltype(has_d)::value;
        const int stn = qtn % NS, stc = qtc % NS, std_ = qtd % NS;
        bf16x8 qa[4], da[NG][4], tq[4], td[NG][4];
        auto load_group = [&](int i) {
            if (HN) {
                qa[i] = row_frag(QOFF + stn * TILE, qbn, i);
#pragma unroll
                for (int p = 0; p < NG; ++p) da[p][i] = row_frag(DOFF + (stn * NG + p) * TILE, qbn, i);
            }
            if (HD_) {
                tq[i] = tr_frag(QOFF + std_ * TILE, qbd, i);
#pragma unroll
                for (int p = 0; p < NG; ++p) td[p][i] = tr_frag(DOFF + (std_ * NG + p) * TILE, qbd, i);
            }
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) load_group(i);
        const float* nd = reinterpret_cast<const float*>(smem + NOFF + stc * 128);
        const int mw = (DROP == 2 && HC) ? (int)(*reinterpret_cast<const unsigned*>(smem + MOFF + stc * 512 + qbc * 256 + mword * 2) >> (4 * hi)) : 0;
        if (HN) {
            const uint4 w = *reinterpret_cast<const uint4*>(hi ? smem + ZOFF : smem + ROFF + stn * 512 + (qbn * 32 + lk) * 8);
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            sn = fb_mma<MODE>(fb_from4(w.x, w.y, 0u, 0u), ones, z);
            tn = fb_mma<MODE>(fb_from4(w.z, w.w, 0u, 0u), ones, z);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i + PF < 4) load_group(i + PF);
            if (HD_) {
                const int ks = i >> 1, db = i & 1;
                const bf16x8 pf = fb_from4(ppp[4 * ks], ppp[4 * ks + 1], ppp[4 * ks + 2], ppp[4 * ks + 3]);
#pragma unroll
                for (int p = 0; p < NG; ++p) dvacc[db] = fb_mma<MODE>(pf, td[p][i], dvacc[db]);
            }
            if (HC) pair(qtc, qbc, mw, 2 * i, sc_, tc_, nd, ppc, dsc);
            __builtin_amdgcn_sched_barrier(0);
            if (HN) {
                sn = fb_mma<MODE>(qa[i], kf[i], sn);
#pragma unroll
                for (int p = 0; p < NG; ++p) tn = fb_mma<MODE>(da[p][i], vf[i], tn);
            }
            if (HD_) {
                const int ks = i >> 1, db = i & 1;
#pragma unroll
                for (int p = 0; p < NG; ++p)
                    dkacc[db] = fb_mma<MODE>(fb_from4(dsp[p][4 * ks], dsp[p][4 * ks + 1], dsp[p][4 * ks + 2], dsp[p][4 * ks + 3]), tq[i], dkacc[db]);
            }
            if (HC) pair(qtc, qbc, mw, 2 * i + 1, sc_, tc_, nd, ppc, dsc);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // LEAN step (GX = 0, two waves per SIMD): S'T'(next unit) | P, dS of the current unit | dV, dK of the CURRENT unit as soon as its
    // words exist (key-step 0 after the first four pairs, key-step 1 behind the loop): no second P / dS buffer, fragments one group ahead
    // (stn_, stc_: the ring stages of the tiles of units n and c as COMPILE-TIME constants -- the tile loop below is unrolled over the
    // three stages.  With `qt % NS` at run time every one of a step's 33 LDS fragment reads cost two vector adds for its stage offset:
    // 48 of the step's 154 vector instructions)
    auto step_lean = [&](auto has_n, auto has_c, auto stn_, auto stc_, int qtn, int qbn, int qtc, int qbc, f32x16& sn, f32x16& tn, const f32x16& sc_,
                         const f32x16& tc_, unsigned (&pp)[8], unsigned (&ds)[NG][8]) {
        constexpr bool HN = decltype(has_n)::value, HC = decltype(has_c)::value;
        constexpr int stn = decltype(stn_)::value, stc = decltype(stc_)::value;
        const float* nd = reinterpret_cast<const float*>(smem + NOFF + stc * 128);
        const int mw = (DROP == 2 && HC) ? (int)(*reinterpret_cast<const unsigned*>(smem + MOFF + stc * 512 + qbc * 256 + mword * 2) >> (4 * hi)) : 0;
        bf16x8 qa[4], da[NG][4], tq[4], td[NG][4];
        auto load_n = [&](int i) {
            if (HN) {
                qa[i] = row_frag(QOFF + stn * TILE, qbn, i);
#pragma unroll
                for (int p = 0; p < NG; ++p) da[p][i] = row_frag(DOFF + (stn * NG + p) * TILE, qbn, i);
            }
        };
        auto load_d = [&](int i) {
            if (HC) {
                tq[i] = tr_frag(QOFF + stc * TILE, qbc, i);
#pragma unroll
                for (int p = 0; p < NG; ++p) td[p][i] = tr_frag(DOFF + (stc * NG + p) * TILE, qbc, i);
            }
        };
        auto dvdk = [&](int i) {
            if (!HC) return;
            const int ks = i >> 1, db = i & 1;
            const bf16x8 pf = fb_from4(pp[4 * ks], pp[4 * ks + 1], pp[4 * ks + 2], pp[4 * ks + 3]);
#pragma unroll
            for (int p = 0; p < NG; ++p) {
                dvacc[db] = fb_mma<MODE>(pf, td[p][i], dvacc[db]);
                dkacc[db] = fb_mma<MODE>(fb_from4(ds[p][4 * ks], ds[p][4 * ks + 1], ds[p][4 * ks + 2], ds[p][4 * ks + 3]), tq[i], dkacc[db]);
            }
        };
        load_n(0);
        if (HN) {
            const uint4 w = *reinterpret_cast<const uint4*>(hi ? smem + ZOFF : smem + ROFF + stn * 512 + (qbn * 32 + lk) * 8);
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            sn = fb_mma<MODE>(fb_from4(w.x, w.y, 0u, 0u), ones, z);
            tn = fb_mma<MODE>(fb_from4(w.z, w.w, 0u, 0u), ones, z);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i + 1 < 4) load_n(i + 1);
            if (i == 1) { load_d(0); load_d(1); }
            if (i == 3) { load_d(2); load_d(3); }
            if (HC) pair(qtc, qbc, mw, 2 * i, sc_, tc_, nd, pp, ds);
            __builtin_amdgcn_sched_barrier(0);
            if (HN) {
                sn = fb_mma<MODE>(qa[i], kf[i], sn);
#pragma unroll
                for (int p = 0; p < NG; ++p) tn = fb_mma<MODE>(da[p][i], vf[i], tn);
            }
            if (i == 2) { dvdk(0); }
            if (i == 3) { dvdk(1); }
            if (HC) pair(qtc, qbc, mw, 2 * i + 1, sc_, tc_, nd, pp, ds);
            __builtin_amdgcn_sched_barrier(0);
        }
        dvdk(2); dvdk(3);
        __builtin_amdgcn_sched_barrier(0);
    };
    const std::true_type T_;
    const std::false_type F_;

    if constexpr (!GX) {
        issue(0); issue(1);
        fb_wait_vm<0>();
        __syncthreads();
        f32x16 s_[2], t_[2];
        unsigned pp1[8], ds1[NG][8];
        typedef std::integral_constant<int, 0> S0;
        typedef std::integral_constant<int, 1> S1;
        typedef std::integral_constant<int, 2> S2;
        step_lean(T_, F_, S0(), S0(), 0, 0, 0, 0, s_[0], t_[0], s_[0], t_[0], pp1, ds1);
        // region j = units (j, 0), (j, 1): tiles j (current, stage j % 3) and j + 1 (the next unit's scores)
        auto region = [&](auto sa, auto sb, int j) {
            step_lean(T_, T_, sa, sa, j, 1, j, 0, s_[1], t_[1], s_[0], t_[0], pp1, ds1);
            fb_wait_vm<0>();                                // tile j + 1 (requested one region ago) landed
            vxb_raw_barrier();
            if (j + 2 < nqt + 1) issue(j + 2);              // stage (j + 2) % 3 held tile j - 1
            step_lean(T_, T_, sb, sa, j + 1, 0, j, 1, s_[0], t_[0], s_[1], t_[1], pp1, ds1);
        };
        int j = 0;
        for (; j + 3 <= nqt; j += 3) { region(S0(), S1(), j); region(S1(), S2(), j + 1); region(S2(), S0(), j + 2); }
        if (j < nqt) region(S0(), S1(), j);
        if (j + 1 < nqt) region(S1(), S2(), j + 1);
        fb_wait_vm<0>();
    } else {
    issue(0); issue(1);
    fb_wait_vm<0>();
    __syncthreads();
    f32x16 s_[2], t_[2];
    unsigned ppb[2][8], dsb[2][NG][8];
    step(T_, F_, F_, 0, 0, 0, 0, 0, 0, s_[0], t_[0], s_[0], t_[0], ppb[0], dsb[0], ppb[0], dsb[0]);
    step(T_, T_, F_, 0, 1, 0, 0, 0, 0, s_[1], t_[1], s_[0], t_[0], ppb[0], dsb[0], ppb[0], dsb[0]);
    for (int j = 0; j < nqt; ++j) {
        fb_wait_vm<0>();                                    // tile j + 1 (requested one region ago) landed
        vxb_raw_barrier();
        issue(j + 2);                                       // stage (j + 2) % 3 held tile j - 1
        step(T_, T_, T_, j + 1, 0, j, 1, j, 0, s_[0], t_[0], s_[1], t_[1], ppb[1], dsb[1], ppb[0], dsb[0]);
        step(T_, T_, T_, j + 1, 1, j + 1, 0, j, 1, s_[1], t_[1], s_[0], t_[0], ppb[0], dsb[0], ppb[1], dsb[1]);
    }
    fb_wait_vm<0>();
    }
    // accumulators: C[i = key (rows, regs)][j = d (lane)]
    const int kw0 = kblk * (NW * 32) + wid * 32;
    const float fk = g.scale * g.scale_ws[1], fv = g.scale_ws[1];
    float* dkp = g.dkv + (long long)b * g.Nk * 2 * inner + h * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = kw0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kk < g.Nk) {
                dkp[(long long)kk * 2 * inner + db * 32 + lk] = dkacc[db][r] * fk;
                dkp[(long long)kk * 2 * inner + inner + db * 32 + lk] = dvacc[db][r] * fv;
            }
        }
}

template <int MODE, int GX, int NW>
int f2b_dkv_launch(const F2bArgs& g, bool drop, hipStream_t st) {
    constexpr int NG = 1 + GX;
    const size_t lds = (size_t)(3 * TILE + 3 * NG * TILE + 3 * 512 + 3 * 128 + 16 + 3 * 512) * sizeof(u16);
    const dim3 grid(g.nblk * g.B * g.H);
    if (drop && g.mask) {
        if (hipFuncSetAttribute((const void*)f2b_dkv_kernel<MODE, GX, 2, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((f2b_dkv_kernel<MODE, GX, 2, NW>), grid, dim3(NW * 64), lds, st, g);
    } else if (drop) {
        if (hipFuncSetAttribute((const void*)f2b_dkv_kernel<MODE, GX, 1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((f2b_dkv_kernel<MODE, GX, 1, NW>), grid, dim3(NW * 64), lds, st, g);
    } else {
        if (hipFuncSetAttribute((const void*)f2b_dkv_kernel<MODE, GX, 0, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((f2b_dkv_kernel<MODE, GX, 0, NW>), grid, dim3(NW * 64), lds, st, g);
    }
    return VXB_OK;
}

template <int MODE, int GX, int NW>
int f2b_dq_launch(const F2bArgs& g, bool drop, hipStream_t st) {
    const size_t lds = (size_t)2 * NST * TILE * sizeof(u16);
    const dim3 grid(g.nblk * g.B * g.H);
    if (drop && g.mask) {
        const size_t ldsm = lds + (size_t)NST * NW * 256;       // + the keep-word ring
        if (hipFuncSetAttribute((const void*)f2b_dq_kernel<MODE, GX, 2, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((f2b_dq_kernel<MODE, GX, 2, NW>), grid, dim3(NW * 64), ldsm, st, g);
    } else if (drop) {
        if (hipFuncSetAttribute((const void*)f2b_dq_kernel<MODE, GX, 1, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((f2b_dq_kernel<MODE, GX, 1, NW>), grid, dim3(NW * 64), lds, st, g);
    } else {
        if (hipFuncSetAttribute((const void*)f2b_dq_kernel<MODE, GX, 0, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL((f2b_dq_kernel<MODE, GX, 0, NW>), grid, dim3(NW * 64), lds, st, g);
    }
    return VXB_OK;
}

}  // namespace

// workspace of vxb_flash2_attn_bwd in bytes: D | max word + scale words | row fragments | -2^k D | q plane | dO' plane(s)
extern "C" size_t vxb_flash2_attn_bwd_ws_bytes(int B, int H, int Nq, int gx) {
    const size_t Nq64 = ((size_t)Nq + 63) / 64 * 64, rows = (size_t)B * H * Nq64;
    return rows * 4 + 256 + rows * 16 + rows * 4 + (size_t)B * Nq * H * HD * 2 * (size_t)(2 + (gx ? 1 : 0)) + 1024;
}

// Backward of the fused attention, pipelined structure.  kv_plane: the forward's 16-bit plane of k | v (mode 0 bf16, 1 fp16);
// gx = 1: dO and dS as hi + lo pairs.  which: 1 = dQ only, 2 = dK | dV only, 3 = both.  ws: vxb_flash2_attn_bwd_ws_bytes, 256-byte
// aligned.  Same dropout mask as the forward entries.
static int f2b_run(const float* q, const float* kv, const float* o, const float* d_o, const float* lse, const void* kv_plane, const void* drop_mask,
                   int mode, int gx, float* dq, float* dkv, void* ws, int B, int H, int Nq, int Nk, int head_dim, float scale,
                   float dropout_p, uint32_t seed, int which, vxb_stream_t stream);

extern "C" int vxb_flash2_attn_bwd(const float* q, const float* kv, const float* o, const float* d_o, const float* lse, const void* kv_plane,
                                   int mode, int gx, float* dq, float* dkv, void* ws, int B, int H, int Nq, int Nk, int head_dim, float scale,
                                   float dropout_p, uint32_t seed, int which, vxb_stream_t stream) {
    return f2b_run(q, kv, o, d_o, lse, kv_plane, nullptr, mode, gx, dq, dkv, ws, B, H, Nq, Nk, head_dim, scale, dropout_p, seed, which, stream);
}

// The same backward reading the dropout keep words the forward wrote (vxb_flash2_attn_fwd_mask, same B, H, Nq, Nk and dropout_p > 0)
// instead of regenerating the mask: identical results, 9 - 15 fewer vector instructions per score pair.
extern "C" int vxb_flash2_attn_bwd_mask(const float* q, const float* kv, const float* o, const float* d_o, const float* lse, const void* kv_plane,
                                        const void* drop_mask, int mode, int gx, float* dq, float* dkv, void* ws, int B, int H, int Nq, int Nk,
                                        int head_dim, float scale, float dropout_p, uint32_t seed, int which, vxb_stream_t stream) {
    if (!drop_mask || (((uintptr_t)drop_mask) & 15)) return VXB_EARG;
    return f2b_run(q, kv, o, d_o, lse, kv_plane, drop_mask, mode, gx, dq, dkv, ws, B, H, Nq, Nk, head_dim, scale, dropout_p, seed, which, stream);
}

static int f2b_run(const float* q, const float* kv, const float* o, const float* d_o, const float* lse, const void* kv_plane, const void* drop_mask,
                   int mode, int gx, float* dq, float* dkv, void* ws, int B, int H, int Nq, int Nk, int head_dim, float scale,
                   float dropout_p, uint32_t seed, int which, vxb_stream_t stream) {
    if (!q || !kv || !o || !d_o || !lse || !kv_plane || !ws || B < 1 || H < 1 || Nq < 1 || Nk < 1 || mode < 0 || mode > 1) return VXB_EARG;
    if (((which & 1) && !dq) || ((which & 2) && !dkv) || !(which & 3)) return VXB_EARG;
    if (head_dim != HD || dropout_p < 0.f || dropout_p >= 1.f || (((uintptr_t)kv_plane | (uintptr_t)ws) & 15)) return VXB_ESIZE;
    if ((long long)Nk * 2 * H * HD * 2 * 2 > 0xffffffffLL || (long long)Nq * H * HD * 2 * 2 > 0xffffffffLL) return VXB_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    const int Nq64 = (Nq + 63) / 64 * 64;
    const size_t rows = (size_t)B * H * Nq64;
    char* w = (char*)ws;
    float* dsum = (float*)w; w += rows * 4;
    unsigned* amax = (unsigned*)w; float* scale_ws = (float*)(w + 16); w += 256;
    unsigned* rowslots = (unsigned*)w; w += rows * 16;
    float* negd = (float*)w; w += rows * 4;
    w = (char*)(((uintptr_t)w + 255) & ~(uintptr_t)255);
    u16* qp = (u16*)w; w += (size_t)B * Nq * H * HD * 2;
    u16* dop = (u16*)w;
    const long long do_plane = (long long)B * Nq * H * HD;
    if (hipMemsetAsync(amax, 0, 16, st) != hipSuccess) return VXB_ELAUNCH;
    const long long groups = (long long)B * Nq * H;
    hipLaunchKernelGGL(f2b_prep1_kernel, dim3(min(vxb_cdiv(groups * 16, 256), 2048)), dim3(256), 0, st, d_o, o, dsum, amax, B, H, Nq, Nq64);
    const long long n2 = groups * (HD / 4) > (long long)rows ? groups * (HD / 4) : (long long)rows;
    const dim3 g2(vxb_cdiv(n2, 256));
#define F2B_PREP2(M, G) hipLaunchKernelGGL((f2b_prep2_kernel<M, G>), g2, dim3(256), 0, st, q, d_o, lse, dsum, amax, scale_ws, rowslots, negd, qp, dop, do_plane, B, H, Nq, Nq64, dropout_p)
    if (mode == M_BF16) { if (gx) F2B_PREP2(M_BF16, 1); else F2B_PREP2(M_BF16, 0); }
    else { if (gx) F2B_PREP2(M_F16, 1); else F2B_PREP2(M_F16, 0); }
#undef F2B_PREP2
    F2bArgs g;
    g.q = q; g.kv = kv; g.d_o = d_o; g.kvp = (const u16*)kv_plane; g.qp = qp; g.dop = dop; g.do_plane = do_plane;
    g.rowslots = rowslots; g.negd = negd; g.scale_ws = scale_ws; g.dq = dq; g.dkv = dkv;
    g.mask = (const unsigned*)drop_mask; g.nrb = ((Nq + 255) / 256) * 8; g.ntile = (Nk + BT - 1) / BT;
    g.B = B; g.H = H; g.Nq = Nq; g.Nk = Nk; g.Nq64 = Nq64; g.scale = scale; g.p_drop = dropout_p; g.seed = seed;
    const bool drop = (unsigned)(dropout_p * 65536.0f) > 0u;
    int rc = VXB_OK;
    if (which & 1) {
        g.nblk = vxb_cdiv(Nq, 128);
        if (mode == M_BF16) rc = gx ? f2b_dq_launch<M_BF16, 1, 4>(g, drop, st) : f2b_dq_launch<M_BF16, 0, 4>(g, drop, st);
        else rc = gx ? f2b_dq_launch<M_F16, 1, 4>(g, drop, st) : f2b_dq_launch<M_F16, 0, 4>(g, drop, st);
        if (rc != VXB_OK) return rc;
    }
    if (which & 2) {
        g.nblk = vxb_cdiv(Nk, 128);
        if (mode == M_BF16) rc = gx ? f2b_dkv_launch<M_BF16, 1, 4>(g, drop, st) : f2b_dkv_launch<M_BF16, 0, 4>(g, drop, st);
        else rc = gx ? f2b_dkv_launch<M_F16, 1, 4>(g, drop, st) : f2b_dkv_launch<M_F16, 0, 4>(g, drop, st);
        if (rc != VXB_OK) return rc;
    }
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
