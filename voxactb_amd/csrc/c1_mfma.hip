// Matrix-core versions of the Cout = 1 3x3x3 conv of the translation head (trans_decoder, perceiver_lang_io.py:316-321 /
// :466) for the 'bf16x3' / 'bf16' precisions.  The VALU kernels in c1_conv.hip walk a sliding 3x3x3 window (27 float4 LDS
// weight reads and 108 FMAs per voxel and thread) and sit at 3.5 ms per pass over the 4.1 GB tensor; with ONE output
// channel a direct implicit GEMM would waste 31 of 32 MFMA columns.  Instead the 27 TAPS are the MFMA column dimension:
//
//   forward   P[v'][t] = sum_c u[v'][c] * w[c][t]        one 64-deep GEMM per voxel (no halo on the GEMM itself)
//             q[v]     = bias + sum_t P[clamp(v + t - 1)][t]     27 LDS reads per output voxel
//             A workgroup owns 4x8x8 outputs: P is computed for the 6x10x10 halo (19 M tiles of 32 voxels, A fragments
//             straight from global memory -- a lane's 8 channels are 32 contiguous bytes --, B = the weights in registers),
//             kept in LDS as fp32 [600][29], and gathered.
//   weight    dW[c][t] = sum_v' u[v'][c] * Bq[v'][t],  Bq[v'][t] = sum of dq[v] over the outputs v with clamp(v + t - 1) = v'
//   gradient  (dq[v' - t + 1] plus, on a face of the cube, the output whose tap was clamped onto v').  A workgroup streams
//             2x8x8 voxel tiles: u tile and Bq tile are staged voxel-major as bf16 hi|lo planes and both go through
//             ds_read_b64_tr_b16 into v_mfma_f32_16x16x32_bf16 (k = 32 voxels), exactly like wgrad_halo.hip.
// Products are bf16x3 (hi*hi + hi*lo + lo*hi, fp32 accumulate) in both precisions: the pass is HBM-bound either way.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ void cm_split2(float a, float b, unsigned& ph, unsigned& pl) {
    ph = vxb_pack_bf16(a, b);
    pl = vxb_pack_bf16(a - __uint_as_float(ph << 16), b - __uint_as_float(ph & 0xffff0000u));
}

// ------------------------------------------------------------------------------------------------ forward
constexpr int FT_D = 4, FT_H = 8, FT_W = 8;            // output tile
constexpr int FH_D = FT_D + 2, FH_H = FT_H + 2, FH_W = FT_W + 2;
constexpr int FNPOS = FH_D * FH_H * FH_W;              // 600 halo voxels
constexpr int FMT = (FNPOS + 31) / 32;                 // 19 M tiles
constexpr int PST = 29;                                // floats per P row: odd -> conflict-free writes and gathers

// w [64][27] fp32 -> B fragments of v_mfma_f32_32x32x16_bf16: [k-step 4][plane 2][lane 64][8] bf16, lane (n = lane & 31,
// half = lane >> 5) holds w[16 ks + 8 half + e][n] (0 for n >= 27)
__global__ void __launch_bounds__(256) c1_wfrag_kernel(const float* __restrict__ w, u16* __restrict__ wf) {
    const int ks = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, half = lane >> 5;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const int c = 16 * ks + 8 * half + e;
        const float a = n < 27 ? w[c * 27 + n] : 0.f, b = n < 27 ? w[(c + 1) * 27 + n] : 0.f;
        cm_split2(a, b, hi[e >> 1], lo[e >> 1]);
    }
    uint4* o = reinterpret_cast<uint4*>(wf);
    o[(ks * 2 + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    o[(ks * 2 + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__global__ void __launch_bounds__(256, 2) c1_fwd_mfma_kernel(const float* __restrict__ u, const u16* __restrict__ wf,
                                                            const float* __restrict__ bias, float* __restrict__ q, int B, int S,
                                                            int ntd, int nth, int ntw) {
    extern __shared__ __attribute__((aligned(16))) float P[];      // [FMT * 32][PST]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, lq = lane & 31;
    // XCD-aware tile order (workgroup id b runs on XCD b % 8): neighbouring tiles share their halo through one L2
    int t;
    {
        const int nwg = gridDim.x, lid = blockIdx.x;
        const int xcd = lid & 7, slot = lid >> 3;
        const int qq = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
    }
    const int tw = t % ntw; t /= ntw;
    const int th = t % nth; t /= nth;
    const int td = t % ntd; t /= ntd;
    const int b = t;      // (row order: the 4 x 4 x 4 block order of the LDS-halo convs was measured here and is 3 % slower, 1.35 vs 1.31 ms)
    const int d0 = td * FT_D, h0 = th * FT_H, w0 = tw * FT_W;

    bf16x8 bh[4], bl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bh[ks] = *reinterpret_cast<const bf16x8*>(wf + ((ks * 2 + 0) * 64 + lane) * 8);
        bl[ks] = *reinterpret_cast<const bf16x8*>(wf + ((ks * 2 + 1) * 64 + lane) * 8);
    }
    const float* ub = u + (long long)b * S * S * S * 64 + 8 * half;
#pragma unroll
    for (int i = 0; i < (FMT + 3) / 4; ++i) {
        const int mt = wid + 4 * i;
        if (mt < FMT) {
            int p = min(mt * 32 + lq, FNPOS - 1);
            const int pw = p % FH_W; p /= FH_W;
            const int ph = p % FH_H; p /= FH_H;
            const int id = min(max(d0 + p - 1, 0), S - 1), ih = min(max(h0 + ph - 1, 0), S - 1), iw = min(max(w0 + pw - 1, 0), S - 1);
            const float* up = ub + (((long long)id * S + ih) * S + iw) * 64;
            float4 va[4], vb[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                va[ks] = *reinterpret_cast<const float4*>(up + 16 * ks);
                vb[ks] = *reinterpret_cast<const float4*>(up + 16 * ks + 4);
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { unsigned x[4]; bf16x8 v; } ah, al;
                cm_split2(va[ks].x, va[ks].y, ah.x[0], al.x[0]); cm_split2(va[ks].z, va[ks].w, ah.x[1], al.x[1]);
                cm_split2(vb[ks].x, vb[ks].y, ah.x[2], al.x[2]); cm_split2(vb[ks].z, vb[ks].w, ah.x[3], al.x[3]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh[ks], acc, 0, 0, 0);
            }
            // C layout: column n = lq (tap), rows (r & 3) + 8 (r >> 2) + 4 half
            if (lq < 27) {
#pragma unroll
                for (int r = 0; r < 16; ++r) P[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * PST + lq] = acc[r];
            }
        }
    }
    __syncthreads();
    const int ow = tid & 7, oh = (tid >> 3) & 7, od = tid >> 6;
    const int gd = d0 + od, gh = h0 + oh, gw = w0 + ow;
    if (gd < S && gh < S && gw < S) {
        const float* pp = P + ((od * FH_H + oh) * FH_W + ow) * PST;
        float s = bias[0];
#pragma unroll
        for (int tp = 0; tp < 27; ++tp)
            s += pp[(((tp / 9) * FH_H + (tp / 3) % 3) * FH_W + tp % 3) * PST + tp];
        q[(((long long)b * S + gd) * S + gh) * S + gw] = s;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int GT_D = 2, GT_H = 8, GT_W = 8;            // voxel tile = 128 voxels = 4 k-steps of 32
constexpr int ULD = 80;                                // u16 per u row (64 + 16 pad): conflict-free transposed reads
constexpr int BLD = 48;                                // u16 per Bq row (32 taps + 16 pad)
constexpr int UPL = 128 * ULD, BPL = 128 * BLD;        // u16 per plane
constexpr int QH_D = GT_D + 2, QH_H = GT_H + 2, QH_W = GT_W + 2;
constexpr int QNPOS = QH_D * QH_H * QH_W;              // 400 dq halo voxels

__device__ __forceinline__ bf16x8 cm_frag(const u16* p0, const u16* p1) {
    union { s16x4 s[2]; bf16x8 v; } t;
    t.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    t.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p1));
    return t.v;
}

// part[blk][c * 27 + t], partB[blk]: partial sums of workgroup blk over its run of voxel tiles (summed by c1_reduce_kernel)
// F16 = 1 (round 4): ONE fp16 product per term -- this weight gradient is a leaf of the backward pass like the other conv weight
// gradients -- with dq * scale[0] (a device-side power of two that brings max |dq| into [2^14, 2^15): dq = softmax - onehot
// spans 1 .. 1e-9) and the sums multiplied by scale[1].  The bf16x3 form spends 24 VALU instructions per MFMA on the hi / lo splits of u
// and of the 128 x 32 Bq tile (2.6 TB/s for a pass that reads u once); fp16 needs a third of the MFMAs and a quarter of the conversions.
typedef _Float16 cm_f16x8 __attribute__((ext_vector_type(8)));
template <int F16>
__global__ void __launch_bounds__(256, 2) c1_wgrad_mfma_kernel(const float* __restrict__ u, const float* __restrict__ dq,
                                                              float* __restrict__ part, float* __restrict__ partB, int B, int S,
                                                              int ntd, int nth, int ntw, long long ntiles, int tiles_per_block,
                                                              const float* __restrict__ scale) {
    const float qsc = F16 ? scale[0] : 1.f, qinv = F16 ? scale[1] : 1.f;
    extern __shared__ __attribute__((aligned(16))) u16 gsm[];
    u16* us = gsm;                                                 // [plane][128 voxels][ULD]
    u16* bs = gsm + 2 * UPL;                                       // [plane][128 voxels][BLD]
    float* dqh = reinterpret_cast<float*>(gsm + 2 * UPL + 2 * BPL);    // [QNPOS]
    float* redb = dqh + QNPOS;                                     // [4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long t_begin = (long long)blockIdx.x * tiles_per_block;
    const long long t_end = min(ntiles, t_begin + tiles_per_block);

    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;

    // two register sets of loads: the tiles t + 1 and t + 2 are in flight while tile t is converted and multiplied (one set left the
    // loads in flight only between a tile's staging and the next one's: 2.6 TB/s for a pass that reads u once)
    float4 pu2[2][8];
    float pq2[2][2];
    auto issue = [&](long long tile, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        float4 (&pu)[8] = pu2[SET];
        float (&pq)[2] = pq2[SET];
        long long t = tile;
        const int tw = (int)(t % ntw); t /= ntw;
        const int th = (int)(t % nth); t /= nth;
        const int td = (int)(t % ntd); t /= ntd;
        const long long bb = t * S * S * S;
        const int d0 = td * GT_D, h0 = th * GT_H, w0 = tw * GT_W;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, pos = e >> 4, c4 = (e & 15) * 4;
            const int od = d0 + (pos >> 6), oh = h0 + ((pos >> 3) & 7), ow = w0 + (pos & 7);
            pu[i] = make_float4(0.f, 0.f, 0.f, 0.f);       // voxels outside the cube contribute nothing
            if (od < S && oh < S && ow < S) pu[i] = *reinterpret_cast<const float4*>(u + (bb + ((long long)od * S + oh) * S + ow) * 64 + c4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int p = tid + 256 * i;
            pq[i] = 0.f;
            if (p < QNPOS) {
                const int pw = p % QH_W; p /= QH_W;
                const int ph = p % QH_H; p /= QH_H;
                const int id = d0 + p - 1, ih = h0 + ph - 1, iw = w0 + pw - 1;
                if (id >= 0 && id < S && ih >= 0 && ih < S && iw >= 0 && iw < S) pq[i] = dq[bb + ((long long)id * S + ih) * S + iw];
            }
        }
    };

    // fragment addressing (wgrad_halo.hip): lane group g = lane >> 4, tl = lane & 15 -> voxel (h = hb + 2 (g >> 1) + r,
    // w = 4 (g & 1) + (tl >> 2)), column quad tl & 3
    const int g4 = lane >> 4, tl = lane & 15;
    const int fh = 2 * (g4 >> 1), fw = 4 * (g4 & 1) + (tl >> 2), fc = 4 * (tl & 3);

    auto one_tile = [&](long long tile, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        float4 (&pu)[8] = pu2[SET];
        float (&pq)[2] = pq2[SET];
        long long t = tile;
        const int tw = (int)(t % ntw); t /= ntw;
        const int th = (int)(t % nth); t /= nth;
        const int td = (int)(t % ntd);
        const int d0 = td * GT_D, h0 = th * GT_H, w0 = tw * GT_W;
        __syncthreads();                 // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            uint2 hi2, lo2;
            if (F16) {
                hi2.x = vxb_pack_f16(__builtin_amdgcn_fmed3f(pu[i].x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(pu[i].y, -65504.f, 65504.f));
                hi2.y = vxb_pack_f16(__builtin_amdgcn_fmed3f(pu[i].z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(pu[i].w, -65504.f, 65504.f));
                *reinterpret_cast<uint2*>(&us[(e >> 4) * ULD + (e & 15) * 4]) = hi2;
            } else {
                cm_split2(pu[i].x, pu[i].y, hi2.x, lo2.x); cm_split2(pu[i].z, pu[i].w, hi2.y, lo2.y);
                *reinterpret_cast<uint2*>(&us[(e >> 4) * ULD + (e & 15) * 4]) = hi2;
                *reinterpret_cast<uint2*>(&us[UPL + (e >> 4) * ULD + (e & 15) * 4]) = lo2;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = tid + 256 * i;
            if (p < QNPOS) {
                dqh[p] = pq[i];
                // bias gradient: every output voxel of the cube is the centre of exactly one tile's halo interior
                const int pw = p % QH_W, ph = (p / QH_W) % QH_H, pd = p / (QH_W * QH_H);
                if (pw >= 1 && pw <= GT_W && ph >= 1 && ph <= GT_H && pd >= 1 && pd <= GT_D) accb += pq[i];
            }
        }
        __syncthreads();
        if (tile + 2 < t_end) issue(tile + 2, set_c);
        // ---- Bq[v'][t]: 128 x 27 entries (columns 27..31 are written as zeros).  A thread keeps its tap (tid & 31) and its
        // w position ((tid >> 5) & 7) for all 16 entries; tiles that touch no face of the cube read one dq per entry at a
        // compile-time offset from a per-thread base
        const bool interior = d0 >= 1 && d0 + GT_D <= S - 1 && h0 >= 1 && h0 + GT_H <= S - 1 && w0 >= 1 && w0 + GT_W <= S - 1;
        if (interior) {
            const int tp = tid & 31, lw = tid >> 5;
            const int tpc = tp < 27 ? tp : 0;
            const float* dbase = dqh + ((2 - tpc / 9) * QH_H + 2 - (tpc / 3) % 3) * QH_W + 2 - tpc % 3 + lw;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int pos = lw + 8 * i;                         // lh = i & 7, ld = i >> 3
                const float v = tp < 27 ? dbase[((i >> 3) * QH_H + (i & 7)) * QH_W] : 0.f;
                if (F16) { bs[pos * BLD + tp] = (u16)(vxb_pack_f16(v * qsc, 0.f) & 0xffffu); continue; }
                const unsigned hb = __float_as_uint(v);
                const unsigned h16 = (hb + 0x7fffu + ((hb >> 16) & 1u)) >> 16;
                const unsigned rb = __float_as_uint(v - __uint_as_float(h16 << 16));
                bs[pos * BLD + tp] = (u16)h16;
                bs[BPL + pos * BLD + tp] = (u16)((rb + 0x7fffu + ((rb >> 16) & 1u)) >> 16);
            }
        } else
        for (int e = tid; e < 128 * 32; e += 256) {
            const int pos = e >> 5, tp = e & 31;
            float v = 0.f;
            if (tp < 27) {
                const int lw = pos & 7, lh = (pos >> 3) & 7, ld = pos >> 6;
                const int tdd = tp / 9, thh = (tp / 3) % 3, tww = tp % 3;
                const int gd = d0 + ld, gh = h0 + lh, gw = w0 + lw;
                // per axis: the regular source x' - (t - 1) (halo index l + 2 - t; zero outside the cube) and, on a face,
                // the output x' itself whose tap was clamped onto x'
                const int nd = 1 + ((gd == 0 && tdd == 0) || (gd == S - 1 && tdd == 2));
                const int nh = 1 + ((gh == 0 && thh == 0) || (gh == S - 1 && thh == 2));
                const int nw = 1 + ((gw == 0 && tww == 0) || (gw == S - 1 && tww == 2));
                for (int a = 0; a < nd; ++a)
                    for (int bq = 0; bq < nh; ++bq)
                        for (int c = 0; c < nw; ++c) {
                            const int id = a ? ld + 1 : ld + 2 - tdd, ih = bq ? lh + 1 : lh + 2 - thh, iw = c ? lw + 1 : lw + 2 - tww;
                            v += dqh[(id * QH_H + ih) * QH_W + iw];
                        }
                if (gd >= S || gh >= S || gw >= S) v = 0.f;
            }
            if (F16) { bs[pos * BLD + tp] = (u16)(vxb_pack_f16(v * qsc, 0.f) & 0xffffu); continue; }
            const unsigned hb = __float_as_uint(v);
            // scalar bf16 split (one value): hi = RNE(v), lo = RNE(v - hi)
            unsigned h16 = (hb + 0x7fffu + ((hb >> 16) & 1u)) >> 16;
            const float rem = v - __uint_as_float(h16 << 16);
            const unsigned rb = __float_as_uint(rem);
            const unsigned l16 = (rb + 0x7fffu + ((rb >> 16) & 1u)) >> 16;
            bs[pos * BLD + tp] = (u16)h16;
            bs[BPL + pos * BLD + tp] = (u16)l16;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int dd = ks >> 1, hb = (ks & 1) * 4;
            const int vox = (dd * GT_H + hb + fh) * GT_W + fw;
            const u16* ua0 = us + vox * ULD + fc + 16 * wid;       // wave w owns channels 16 w .. 16 w + 15
            const u16* ua1 = ua0 + GT_W * ULD;
            const u16* bb0 = bs + vox * BLD + fc;
            const u16* bb1 = bb0 + GT_W * BLD;
            const bf16x8 ah = cm_frag(ua0, ua1);
            if (F16) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 bh = cm_frag(bb0 + 16 * j, bb1 + 16 * j);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cm_f16x8, ah), __builtin_bit_cast(cm_f16x8, bh), acc[j], 0, 0, 0);
                }
                continue;
            }
            const bf16x8 al = cm_frag(ua0 + UPL, ua1 + UPL);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 bh = cm_frag(bb0 + 16 * j, bb1 + 16 * j), bl = cm_frag(bb0 + BPL + 16 * j, bb1 + BPL + 16 * j);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        }
    };
    if (t_begin < t_end) issue(t_begin, std::integral_constant<int, 0>{});
    if (t_begin + 1 < t_end) issue(t_begin + 1, std::integral_constant<int, 1>{});
#pragma unroll 1
    for (long long tile = t_begin; tile < t_end; tile += 2) {
        one_tile(tile, std::integral_constant<int, 0>{});
        if (tile + 1 < t_end) one_tile(tile + 1, std::integral_constant<int, 1>{});
    }
    // D tile (16 c x 16 taps): lane l holds column (tap) l & 15, rows (channels) 4 (l >> 4) + r
    float* __restrict__ pw_ = part + (long long)blockIdx.x * 64 * 27;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tp = 16 * j + tl;
        if (tp < 27) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pw_[(16 * wid + 4 * g4 + r) * 27 + tp] = F16 ? acc[j][r] * qinv : acc[j][r];
        }
    }
    accb = wave_sum(accb);
    if (lane == 0) redb[wid] = accb;
    __syncthreads();
    if (tid == 0) partB[blockIdx.x] = (redb[0] + redb[1]) + (redb[2] + redb[3]);
}

// dst[i] += sum_k part[k][i] over nb partial rows, i < n (same fixed order as c1_reduce_kernel in c1_conv.hip)
__global__ void __launch_bounds__(256) c1m_reduce_kernel(const float* __restrict__ part, int nb, int n, float* __restrict__ dst) {
    __shared__ float red[256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < n) {
        int r = rl;
        for (; r + 4 < nb; r += 8) { s0 += part[(long long)r * n + c]; s1 += part[(long long)(r + 4) * n + c]; }
        for (; r < nb; r += 4) s0 += part[(long long)r * n + c];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < n) dst[c] += (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
}

}  // namespace

// q[B, S^3] = bias + conv3x3x3(u [B, S^3, 64], w [64][27]), replicate padding -- matrix-core version of
// vxb_conv3_c1_fwd_f32 (bf16x3 products).  wfrag_ws: 8 KB of device scratch (the weights in fragment order).
extern "C" int vxb_conv3_c1_fwd_mfma(const float* u, const float* w, const float* bias, float* q, int B, int S, void* wfrag_ws,
                                     vxb_stream_t stream) {
    if (!u || !w || !bias || !q || !wfrag_ws || B < 1 || S < 2) return VXB_EARG;
    if ((((uintptr_t)u) & 15) || (((uintptr_t)wfrag_ws) & 15)) return VXB_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(c1_wfrag_kernel, dim3(1), dim3(256), 0, st, w, (u16*)wfrag_ws);
    const int ntd = vxb_cdiv(S, FT_D), nth = vxb_cdiv(S, FT_H), ntw = vxb_cdiv(S, FT_W);
    const long long nblk = (long long)B * ntd * nth * ntw;
    if (nblk >= INT32_MAX) return VXB_ESIZE;
    const size_t lds = (size_t)FMT * 32 * PST * sizeof(float);
    if (hipFuncSetAttribute((const void*)c1_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
    hipLaunchKernelGGL(c1_fwd_mfma_kernel, dim3((unsigned)nblk), dim3(256), lds, st, u, (const u16*)wfrag_ws, bias, q, B, S, ntd, nth, ntw);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// dw [64][27] += sum_v dq[v] * u[clamp(v + t - 1)][c], db[0] += sum dq -- matrix-core version of vxb_conv3_c1_wgrad_f32.
// part_ws: nblocks * (64 * 27 + 1) floats with nblocks = vxb_conv3_c1_wgrad_mfma_blocks(B, S).
static int c1_wgrad_mfma_launch(const float* u, const float* dq, const float* dq_scale, float* dw, float* db, float* part_ws, int B, int S,
                                vxb_stream_t stream) {
    if (!u || !dq || !dw || !db || !part_ws || B < 1 || S < 2) return VXB_EARG;
    if (((uintptr_t)u) & 15) return VXB_ESIZE;
    const int ntd = vxb_cdiv(S, GT_D), nth = vxb_cdiv(S, GT_H), ntw = vxb_cdiv(S, GT_W);
    const long long ntiles = (long long)B * ntd * nth * ntw;
    const int nb = (int)(ntiles < 2048 ? ntiles : 2048);
    const int tpb = (int)((ntiles + nb - 1) / nb);
    const int nblk = (int)((ntiles + tpb - 1) / tpb);
    float* pW = part_ws;
    float* pB = part_ws + (size_t)nblk * 64 * 27;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(2 * UPL + 2 * BPL) * sizeof(u16) + (size_t)(QNPOS + 4) * sizeof(float);
    if (dq_scale) {
        if (hipFuncSetAttribute((const void*)c1_wgrad_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL(c1_wgrad_mfma_kernel<1>, dim3(nblk), dim3(256), lds, st, u, dq, pW, pB, B, S, ntd, nth, ntw, ntiles, tpb, dq_scale);
    } else {
        if (hipFuncSetAttribute((const void*)c1_wgrad_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
        hipLaunchKernelGGL(c1_wgrad_mfma_kernel<0>, dim3(nblk), dim3(256), lds, st, u, dq, pW, pB, B, S, ntd, nth, ntw, ntiles, tpb, dq_scale);
    }
    hipLaunchKernelGGL(c1m_reduce_kernel, dim3(27), dim3(256), 0, st, pW, nblk, 64 * 27, dw);
    hipLaunchKernelGGL(c1m_reduce_kernel, dim3(1), dim3(256), 0, st, pB, nblk, 1, db);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_conv3_c1_wgrad_mfma(const float* u, const float* dq, float* dw, float* db, float* part_ws, int B, int S,
                                       vxb_stream_t stream) {
    return c1_wgrad_mfma_launch(u, dq, nullptr, dw, db, part_ws, B, S, stream);
}

// The same weight gradient on ONE fp16 product per term (a leaf of the backward pass): dq * dq_scale[0] as the fp16 operand, the sums
// times dq_scale[1]; dq_scale: device {2^k, 2^-k} with max |dq| * 2^k in [2^14, 2^15) (vxb_absmax_scale_f32).  db is summed in fp32 as before.
extern "C" int vxb_conv3_c1_wgrad_f16(const float* u, const float* dq, const float* dq_scale, float* dw, float* db, float* part_ws, int B,
                                      int S, vxb_stream_t stream) {
    if (!dq_scale) return VXB_EARG;
    return c1_wgrad_mfma_launch(u, dq, dq_scale, dw, db, part_ws, B, S, stream);
}

extern "C" size_t vxb_conv3_c1_wgrad_mfma_blocks(int B, int S) {
    if (B < 1 || S < 2) return 0;
    const long long ntiles = (long long)B * vxb_cdiv(S, GT_D) * vxb_cdiv(S, GT_H) * vxb_cdiv(S, GT_W);
    const int nb = (int)(ntiles < 2048 ? ntiles : 2048);
    const int tpb = (int)((ntiles + nb - 1) / nb);
    return (size_t)((ntiles + tpb - 1) / tpb);
}
