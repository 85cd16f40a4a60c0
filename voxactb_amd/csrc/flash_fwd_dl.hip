// Fused attention forward with direct-to-LDS K/V tiles ('bf16' and 'bf16x3').
//
// Same math and the same "swapped" matrix-core mapping as flash_fwd_kernel (flash_attn.hip: S^T = K Q^T with the query
// of a lane fixed, O^T += V^T P^T with P^T taken straight from the S^T accumulators), but the K/V operand path is the
// one of gemm_dl.hip: k | v come as bf16 planes [nplanes][B*Nk][2*H*64] (vxb_split_bf16_f32 of the to_kv output) and a
// 64-key tile travels global -> LDS by `global_load_lds_dwordx4` -- no fp32 -> bf16 conversion, no ds_write, no staging
// registers -- into one of two LDS stages, so the next tile is in flight during the current tile's MFMAs and softmax
// and there is ONE barrier per tile.  LDS tiles are unpadded 128-byte rows; 16-byte chunks are XOR-swizzled on the source
// side: K (read by ds_read_b128, 32 keys x one chunk per half wave) with key (row >> 1) & 7, V (read transposed by
// ds_read_b64_tr_b16, 8 keys x 64 bytes per half wave) with key 4 * ((row >> 1) & 1) -- both conflict-free.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int HD = 64, BQ = 128, BKV = 64;
constexpr int TILE = BKV * HD;              // u16 per K or V plane tile (8 KB)
constexpr float LOG2E = 1.4426950408889634f;

struct FdArgs {
    const float* q;       // [B, Nq, H*64] fp32
    const u16* kv;        // planes [npl][B*Nk][2*H*64] bf16
    long long kv_plane;   // u16 per plane
    float* o;
    float* lse;
    int B, H, Nq, Nk;
    float scale, p_drop;
    unsigned seed;
};

__device__ __forceinline__ unsigned fd_hash(unsigned x) {
    // one multiply round: the inputs are already products with odd constants; keep-rate, row / column sums and lag
    // correlations of the 16-bit halves are indistinguishable from the two-round finaliser (checked offline on 4096 x 2048
    // masks), and every hash costs a quarter-rate multiply less in the softmax loops
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ unsigned fd_keep_pair(unsigned seed, unsigned row, unsigned colpair) {     // == fa_keep_pair
    return fd_hash((row * 0x9E3779B1U + seed) ^ (colpair * 0x85EBCA77U + 0xC2B2AE3DU));
}
__device__ __forceinline__ void fd_split2(float a, float b, unsigned& ph, unsigned& pl) {
    ph = vxb_pack_bf16(a, b);
    pl = vxb_pack_bf16(a - __uint_as_float(ph << 16), b - __uint_as_float(ph & 0xffff0000u));
}
__device__ __forceinline__ bf16x8 fd_from4(const unsigned* p) {
    union { unsigned u[4]; bf16x8 v; } t;
    t.u[0] = p[0]; t.u[1] = p[1]; t.u[2] = p[2]; t.u[3] = p[3];
    return t.v;
}
__device__ __forceinline__ bf16x8 fd_join(unsigned long long a, unsigned long long b) {
    union { unsigned long long u[2]; bf16x8 v; } t;
    t.u[0] = a; t.u[1] = b;
    return t.v;
}
typedef short fd_v4s __attribute__((ext_vector_type(4)));
// transposed 8-byte LDS read through the compiler builtin: a lane base + compile-time offset folds into the instruction's
// immediate (no address arithmetic per read) and the compiler places the waits
__device__ __forceinline__ unsigned long long fd_tr16(const u16* p) {
    union { fd_v4s v; unsigned long long u; } t;
    t.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fd_v4s*)p);
    return t.u;
}
template <int X3>
__device__ __forceinline__ f32x16 fd_mma(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 c) {
    if (X3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}
__device__ __forceinline__ void fd_load16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int X3, int DROP>
__global__ void __launch_bounds__(256, 2) flash_fwd_dl_kernel(FdArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int NPL = 1 + X3;
    constexpr int STAGE = 2 * NPL * TILE;           // [K planes][V planes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
    const int inner = g.H * HD;
    const int qrow = blockIdx.x * BQ + wid * 32 + lq;          // this lane's query
    const bool q_ok = qrow < g.Nq;
    const float* qp = g.q + ((long long)b * g.Nq + (q_ok ? qrow : 0)) * inner + h * HD;
    const float qs = g.scale * LOG2E;                          // scores live in the log2 domain

    // Q^T fragments: lane (q, hi) holds q[16 ks + 8 hi .. +8] for ks = 0..3
    bf16x8 qfh[4], qfl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        union { unsigned u[4]; bf16x8 v; } th, tl;
        const float4 a = *reinterpret_cast<const float4*>(qp + 16 * ks + 8 * hi);
        const float4 c = *reinterpret_cast<const float4*>(qp + 16 * ks + 8 * hi + 4);
        if (X3) {
            fd_split2(a.x * qs, a.y * qs, th.u[0], tl.u[0]); fd_split2(a.z * qs, a.w * qs, th.u[1], tl.u[1]);
            fd_split2(c.x * qs, c.y * qs, th.u[2], tl.u[2]); fd_split2(c.z * qs, c.w * qs, th.u[3], tl.u[3]);
        } else {
            th.u[0] = vxb_pack_bf16(a.x * qs, a.y * qs); th.u[1] = vxb_pack_bf16(a.z * qs, a.w * qs);
            th.u[2] = vxb_pack_bf16(c.x * qs, c.y * qs); th.u[3] = vxb_pack_bf16(c.z * qs, c.w * qs);
            tl = th;
        }
        qfh[ks] = th.v; qfl[ks] = tl.v;
    }
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- tile loads: a K (or V) plane tile = 64 keys x 8 chunks = 512 slots = 8 wave instructions; wave w issues
    //      instructions 2w, 2w+1 (keys 16w + 8i + (lane >> 3)), lane -> destination chunk lane & 7
    const long long kv_row0 = (long long)b * g.Nk;
    int lkey[2], kch[2], vch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        lkey[i] = (2 * wid + i) * 8 + (lane >> 3);
        kch[i] = (lane & 7) ^ ((lkey[i] >> 1) & 7);
        vch[i] = (lane & 7) ^ (4 * ((lkey[i] >> 1) & 1));
    }
    auto issue = [&](int stage, int kt) {
        u16* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = min(kt * BKV + lkey[i], g.Nk - 1);            // tail keys re-read the last row; masked below
            const u16* rowp = g.kv + (kv_row0 + key) * (2 * inner) + h * HD;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                fd_load16(rowp + p * g.kv_plane + kch[i] * 8, sb + p * TILE + (2 * wid + i) * 512);
                fd_load16(rowp + p * g.kv_plane + inner + vch[i] * 8, sb + (NPL + p) * TILE + (2 * wid + i) * 512);
            }
        }
    };

    const unsigned thr = (unsigned)(g.p_drop * 65536.0f);
    const float keep_scale = 1.0f / (1.0f - g.p_drop);
    const unsigned row_id = (unsigned)bh * (unsigned)g.Nq + (unsigned)qrow;
    // K fragment (A operand of S^T): key row kb*32 + lq, chunk 2 ks + hi, swizzle key (row >> 1) & 7
    int kbase[2], kkey[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int row = kb * 32 + lq; kbase[kb] = row * 64; kkey[kb] = (row >> 1) & 7; }
    // V^T transposed-read addressing: group gq = lane >> 4, t = lane & 15: key row 4 (gq >> 1) + (t >> 2) (+ tile offsets),
    // d columns 16 (gq & 1) + 4 (t & 3) (+ 32 db): 16-byte chunk index (2 (gq & 1) + ((t & 3) >> 1)) + 4 db, 8-byte half (t & 1)
    const int t16 = lane & 15, gq = lane >> 4;
    const int vrow0 = 4 * (gq >> 1) + (t16 >> 2);
    const int vchunk0 = 2 * (gq & 1) + ((t16 & 3) >> 1), vhalf = (t16 & 1) * 4;     // u16 offset inside the chunk
    // the tile offsets of a read (32 kb + 16 ks + 8 rr keys) are multiples of 8 rows, so the swizzle bit (row >> 1) & 1 is the
    // lane's own: one lane offset per d-block, everything else is an immediate
    int vlane[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) vlane[db] = vrow0 * 64 + ((vchunk0 + 4 * db) ^ (4 * ((vrow0 >> 1) & 1))) * 8 + vhalf;

    const int nkt = (g.Nk + BKV - 1) / BKV;
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) issue((kt + 1) & 1, kt + 1);
        const u16* sb = smem + (kt & 1) * STAGE;

        // ---- S^T = K Q^T for the two 32-key blocks
        f32x16 sacc[2];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
            // all K fragments of the key block in flight before its chain of MFMAs (the registers are there: two waves per SIMD by
            // the LDS budget): read - wait - multiply per k-step exposed an LDS round trip eight times per tile
            bf16x8 ahf[4], alf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int co = kbase[kb] + ((2 * ks + hi) ^ kkey[kb]) * 8;
                ahf[ks] = *reinterpret_cast<const bf16x8*>(sb + co);
                alf[ks] = *reinterpret_cast<const bf16x8*>(sb + X3 * TILE + co);
            }
            __builtin_amdgcn_sched_barrier(0);          // (without it the scheduler sinks every read next to its MFMA again)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sacc[kb] = fd_mma<X3>(ahf[ks], alf[ks], qfh[ks], qfl[ks], sacc[kb]);
        }
        __builtin_amdgcn_s_setprio(0);
        // the V^T fragments of the whole tile are requested here, before the softmax arithmetic that does not depend on them: their
        // LDS round trips run under it instead of in front of every group of PV MFMAs (64 VGPRs; two waves per SIMD either way)
        const u16* vt = sb + NPL * TILE;
        unsigned long long vaf[2][2][2][2], vlf[2][2][2][2];        // [kb][ks][db][rr]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {            // second read: keys + 8
                        const u16* ad = vt + vlane[db] + (kb * 32 + 16 * ks + 8 * rr) * 64;
                        vaf[kb][ks][db][rr] = fd_tr16(ad);
                        if (X3) vlf[kb][ks][db][rr] = fd_tr16(ad + TILE);
                    }
        __builtin_amdgcn_sched_barrier(0);
        const int kbase_t = kt * BKV;
        float mt = -INFINITY;
        if (kbase_t + BKV > g.Nk) {                      // only the last tile can hold keys >= Nk
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_t + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= g.Nk) sacc[kb][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // 0 on the first tile (m_run = -inf)
        m_run = m_new;
        float psum = 0.f;
        unsigned pbh[2][8], pbl[2][8];                         // P^T fragments (bf16 pairs)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float p0 = __builtin_amdgcn_exp2f(sacc[kb][r] - m_new), p1 = __builtin_amdgcn_exp2f(sacc[kb][r + 1] - m_new);
                psum += p0 + p1;
                if (DROP) {
                    const unsigned key = (unsigned)(kbase_t + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                    const unsigned hsh = fd_keep_pair(g.seed, row_id, key >> 1);
                    p0 = (hsh & 0xffffu) >= thr ? p0 * keep_scale : 0.f;
                    p1 = (hsh >> 16) >= thr ? p1 * keep_scale : 0.f;
                }
                if (X3) fd_split2(p0, p1, pbh[kb][r >> 1], pbl[kb][r >> 1]);
                else pbh[kb][r >> 1] = vxb_pack_bf16(p0, p1);
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
                const bf16x8 pfh = fd_from4(&pbh[kb][4 * ks]);
                const bf16x8 pfl = X3 ? fd_from4(&pbl[kb][4 * ks]) : pfh;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vfh = fd_join(vaf[kb][ks][db][0], vaf[kb][ks][db][1]);
                    const bf16x8 vfl = X3 ? fd_join(vlf[kb][ks][db][0], vlf[kb][ks][db][1]) : vfh;
                    oacc[db] = fd_mma<X3>(vfh, vfl, pfh, pfl, oacc[db]);
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

}  // namespace

// Forward of the fused attention with k | v given as bf16 planes [nplanes][B*Nk][2*H*64] (vxb_split_bf16_f32 of the
// to_kv projection output; nplanes = 1: 'bf16', 2: 'bf16x3').  Same outputs and dropout mask as vxb_flash_attn_fwd_bf16*.
extern "C" int vxb_flash_attn_fwd_dl(const float* q, const void* kv_planes, int nplanes, float* o, float* lse, int B, int H,
                                     int Nq, int Nk, int head_dim, float scale, float dropout_p, uint32_t seed,
                                     vxb_stream_t stream) {
    if (!q || !kv_planes || !o || !lse || B < 1 || H < 1 || Nq < 1 || Nk < 1 || (nplanes != 1 && nplanes != 2)) return VXB_EARG;
    if (head_dim != HD || dropout_p < 0.f || dropout_p >= 1.f || (((uintptr_t)kv_planes) & 15)) return VXB_ESIZE;
    FdArgs g;
    g.q = q; g.kv = (const u16*)kv_planes; g.kv_plane = (long long)B * Nk * 2 * H * HD; g.o = o; g.lse = lse;
    g.B = B; g.H = H; g.Nq = Nq; g.Nk = Nk; g.scale = scale; g.p_drop = dropout_p; g.seed = seed;
    const dim3 grid(vxb_cdiv(Nq, BQ), B * H);
    const size_t lds = (size_t)2 * 2 * nplanes * TILE * sizeof(u16);
    const bool drop = (unsigned)(dropout_p * 65536.0f) > 0u;
    if (nplanes == 2) {
        if (hipFuncSetAttribute((const void*)flash_fwd_dl_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)flash_fwd_dl_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VXB_ELAUNCH;
        if (drop) hipLaunchKernelGGL((flash_fwd_dl_kernel<1, 1>), grid, dim3(256), lds, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((flash_fwd_dl_kernel<1, 0>), grid, dim3(256), lds, (hipStream_t)stream, g);
    } else {
        if (drop) hipLaunchKernelGGL((flash_fwd_dl_kernel<0, 1>), grid, dim3(256), lds, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((flash_fwd_dl_kernel<0, 0>), grid, dim3(256), lds, (hipStream_t)stream, g);
    }
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
