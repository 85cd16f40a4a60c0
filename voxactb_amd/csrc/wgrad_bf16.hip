// bf16 matrix-core weight gradient of the implicit-GEMM conv3d ("throughput mode" twin of vxb_conv3d_wgrad_f32):
//     part[z][(tap, ci)][n] = sum over the z-th slice of positions p of  gather(src)[p][(tap, ci)] * dY[p][n]
// Both operands reduce over POSITIONS, but live in HBM position-major ([p][channel], fp32).  They are staged into LDS
// in that natural layout (rounded to bf16, 8-byte stores) and fed to v_mfma_f32_32x32x16_bf16 through the gfx950
// transpose read ds_read_b64_tr_b16, whose semantics were measured on hardware (tools/ubench/trprobe.hip):
//     within a 16-lane group, lane i receives  in[4j + (i >> 2)][i & 3]  for j = 0..3
// where in[t][s] is the s-th b16 at lane t's address.  Pointing lane t at LDS[(p0 + (t >> 2))][c0 + 4 (t & 3)] therefore
// hands lane i channel c0 + i at positions p0 .. p0+3: a [4 pos][16 ch] block, transposed for free.
// LDS row strides are = 16 (mod 64) dwords, which makes the 32 lanes served per LDS cycle hit 64 distinct banks.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

struct WgArgs {
    const float* src0;
    const float* src1;
    const float* dy;
    float* part;
    int C0, C1, S_in, S_out, stride, kext, off, replicate;
    int Krows;           // kext^3 * (C0 + C1)
    int N;               // output channels (columns)
    long long P;         // positions = B * S_out^3
    long long ldy;       // dY row stride (row-major mode)
    int d2s_s, d2s_C;    // > 0: dY is a fine grid [B, (S_out*s)^3, d2s_C], column n = (phase, co)
    int tiles_per_split;
    // 'fp16' products (PM = 2): the GRADIENT operand (src0 when grad_is_src0, else dy) is multiplied by scale[0] before the conversion
    // to half; `part` comes out as scale[0] * dW.  amax_part (optional): word [z * nblk + blk] = largest |gradient operand| (bits,
    // unscaled) seen by a workgroup that walks all of that operand's columns once -- the next step's scale (delayed scaling)
    const float* scale;
    int grad_is_src0;
    unsigned* amax_part;
    float* possum;       // optional, plain mode only: possum[z][Krows] = sum over the z-th slice of positions of src0's rows (fp32, fixed
                         // order) -- the bias gradient of a linear layer falls out of its weight-gradient launch
};

template <int PM>
__device__ __forceinline__ unsigned pack_bf16_2(float lo, float hi) {
    if (PM == 2) return vxb_pack_f16(vxb_sat_f16(lo), vxb_sat_f16(hi));      // (NaN / inf stay non-finite: common.h)
    return vxb_pack_bf16(lo, hi);
}
template <int PM>
__device__ __forceinline__ f32x16 wg_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    if (PM == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned long long ds_read_tr16(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

constexpr int BP = 32;      // positions per K-tile

// X3 = 1: "bf16x3" split products (see gemm_conv.hip): both operands are staged as hi/lo bf16 planes, 3 MFMAs per product.
template <int BN, int PM, int BM = 128>          // BM = (tap, ci) rows per block: 128, or 256 (fp16 products: a quarter fewer LDS fragment
                                                 // reads per MFMA -- 12 per 8 instead of 8 per 4 -- for the same staging per row)
__global__ void __launch_bounds__(256) wgrad_bf16_kernel(WgArgs g) {
    constexpr int X3 = PM == 1;
    constexpr int TPR = BM / 4;                   // threads per position row of the A tile (32 / 64), RPP = position rows per pass
    constexpr int RPP = 256 / TPR;
    constexpr int LDA = BM + 32;                  // 160 bf16 = 80 dwords  (= 16 mod 64)
    constexpr int LDB = BN == 128 ? 160 : 96;     // 80 / 48 dwords        (= 16 / 48 mod 64)
    constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_F4 = BP * BM / 4 / 256;       // 4
    constexpr int B_F4 = BP * BN / 4 / 256;       // 4 or 2
    __shared__ __attribute__((aligned(16))) u16 As[(1 + X3) * BP * LDA];
    __shared__ __attribute__((aligned(16))) u16 Bs[(1 + X3) * BP * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int S = g.S_out;
    const int Ct = g.C0 + g.C1;

    // ---- A: thread-constant (tap, channel) part; e = tid + 256 i -> pos = e / 32, row4 = (e % 32) * 4 = (tid % 32) * 4
    const int kr = m0 + (tid % TPR) * 4;
    const bool a_ok = kr < g.Krows;
    int a_td = 0, a_th = 0, a_tw = 0, a_ch = 0, a_Cs = g.C0;
    const float* a_src = g.src0;
    if (a_ok) {
        const int tap = kr / Ct;
        const int cc = kr - tap * Ct;
        a_tw = tap % g.kext; a_th = (tap / g.kext) % g.kext; a_td = tap / (g.kext * g.kext);
        if (cc >= g.C0) { a_src = g.src1; a_Cs = g.C1; a_ch = cc - g.C0; } else { a_ch = cc; }
    }
    // ---- B: thread-constant column part
    const int bn = n0 + (tid % (BN / 4)) * 4;
    const bool b_ok = bn < g.N;
    int b_rd = 0, b_rh = 0, b_rw = 0, b_co = 0;
    if (g.d2s_s > 0 && b_ok) {
        const int ph = bn / g.d2s_C;
        b_co = bn - ph * g.d2s_C;
        b_rw = ph % g.d2s_s; b_rh = (ph / g.d2s_s) % g.d2s_s; b_rd = ph / (g.d2s_s * g.d2s_s);
    }

    const bool plain = g.S_in == 1 && g.S_out == 1 && g.kext == 1 && g.off == 0 && g.C1 == 0 && g.d2s_s == 0;     // (kernel arguments: uniform)
    long long nkt = (g.P + BP - 1) / BP;
    long long kt_begin = (long long)blockIdx.z * g.tiles_per_split;
    if (nkt > kt_begin + g.tiles_per_split) nkt = kt_begin + g.tiles_per_split;

    // position coordinates of the 8-row sub-slot this thread loads for A (pos = k0 + tid/32 + 8 i) and for B
    // (pos = k0 + tid/(BN/4) + (256/(BN/4)) i): decoded once, advanced incrementally by 32 per tile
    int aw[A_F4], ah[A_F4], ad[A_F4], ab[A_F4];
    int bw[B_F4], bh[B_F4], bd[B_F4], bb[B_F4];
    auto decode = [&](long long pos, int& w, int& h, int& d, int& b) {
        long long r = pos;
        w = (int)(r % S); r /= S;
        h = (int)(r % S); r /= S;
        d = (int)(r % S); r /= S;
        b = (int)r;
    };
    auto advance = [&](int& w, int& h, int& d, int& b) {
        if (S == 1) { b += BP; return; }       // plain GEMM (linear-layer weight gradient): rows are the positions
        w += BP;
        if (w >= S) {
            const int c1 = w / S;
            w -= c1 * S;
            h += c1;
            if (h >= S) {
                const int c2 = h / S;
                h -= c2 * S;
                d += c2;
                if (d >= S) { const int c3 = d / S; d -= c3 * S; b += c3; }
            }
        }
    };
#pragma unroll
    for (int i = 0; i < A_F4; ++i) decode(kt_begin * BP + tid / TPR + RPP * i, aw[i], ah[i], ad[i], ab[i]);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) decode(kt_begin * BP + tid / (BN / 4) + (256 / (BN / 4)) * i, bw[i], bh[i], bd[i], bb[i]);

    float4 ra[A_F4], rb[B_F4];
    float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool want_psum = g.possum != nullptr && blockIdx.x == 0;      // (uniform) the first column block adds up src0's rows
    auto load_tile = [&](long long kt) {
        const long long k0 = kt * BP;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const long long pos = k0 + tid / TPR + RPP * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (plain) {
                // linear-layer weight gradient (a 1 x 1 x 1 "conv" over M positions): row pos of a row-major matrix, no gather
                if (a_ok && pos < g.P) v = *reinterpret_cast<const float4*>(a_src + pos * a_Cs + a_ch);
            } else if (a_ok && pos < g.P) {
                int id = ad[i] * g.stride + a_td + g.off;
                int ih = ah[i] * g.stride + a_th + g.off;
                int iw = aw[i] * g.stride + a_tw + g.off;
                bool ok = true;
                if (g.replicate) {
                    id = min(max(id, 0), g.S_in - 1); ih = min(max(ih, 0), g.S_in - 1); iw = min(max(iw, 0), g.S_in - 1);
                } else {
                    ok = id >= 0 && id < g.S_in && ih >= 0 && ih < g.S_in && iw >= 0 && iw < g.S_in;
                }
                if (ok) {
                    const long long vox = (((long long)ab[i] * g.S_in + id) * g.S_in + ih) * g.S_in + iw;
                    v = *reinterpret_cast<const float4*>(a_src + vox * a_Cs + a_ch);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const long long pos = k0 + tid / (BN / 4) + (256 / (BN / 4)) * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_ok && pos < g.P) {
                if (g.d2s_s > 0) {
                    const int s = g.d2s_s;
                    const long long Vv = (long long)S * s;
                    const long long o = ((((long long)bb[i] * Vv + bd[i] * s + b_rd) * Vv + bh[i] * s + b_rh) * Vv + bw[i] * s + b_rw) * g.d2s_C + b_co;
                    v = *reinterpret_cast<const float4*>(g.dy + o);
                } else {
                    v = *reinterpret_cast<const float4*>(g.dy + pos * g.ldy + bn);
                }
            }
            rb[i] = v;
        }
    };
    const float sc_a = (PM == 2 && g.scale && g.grad_is_src0) ? g.scale[0] : 1.0f;
    const float sc_b = (PM == 2 && g.scale && !g.grad_is_src0) ? g.scale[0] : 1.0f;
    // (uniform) this workgroup sees every value of the gradient operand's slice exactly once per row / column block
    const bool want_amax = PM == 2 && g.amax_part != nullptr && (g.grad_is_src0 ? blockIdx.x == 0 : blockIdx.y == 0);
    float amxf = 0.f, nanw = 0.f;          // largest |gradient operand| seen; nanw: NaN once a NaN / inf was seen (fmaxf drops NaN)
    auto store_tile = [&]() {
        if (want_psum) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) { psum.x += ra[i].x; psum.y += ra[i].y; psum.z += ra[i].z; psum.w += ra[i].w; }
        }
        if (PM == 2) {
            if (want_amax) {
                if (g.grad_is_src0) {
#pragma unroll
                    for (int i = 0; i < A_F4; ++i) {
                        amxf = fmaxf(fmaxf(amxf, fabsf(ra[i].x)), fmaxf(fmaxf(fabsf(ra[i].y), fabsf(ra[i].z)), fabsf(ra[i].w)));
                        nanw = fmaf((ra[i].x + ra[i].y) + (ra[i].z + ra[i].w), 0.0f, nanw);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < B_F4; ++i) {
                        amxf = fmaxf(fmaxf(amxf, fabsf(rb[i].x)), fmaxf(fmaxf(fabsf(rb[i].y), fabsf(rb[i].z)), fabsf(rb[i].w)));
                        nanw = fmaf((rb[i].x + rb[i].y) + (rb[i].z + rb[i].w), 0.0f, nanw);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < A_F4; ++i) { ra[i].x *= sc_a; ra[i].y *= sc_a; ra[i].z *= sc_a; ra[i].w *= sc_a; }
#pragma unroll
            for (int i = 0; i < B_F4; ++i) { rb[i].x *= sc_b; rb[i].y *= sc_b; rb[i].z *= sc_b; rb[i].w *= sc_b; }
        }
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            uint2 p;
            p.x = pack_bf16_2<PM>(ra[i].x, ra[i].y); p.y = pack_bf16_2<PM>(ra[i].z, ra[i].w);
            *reinterpret_cast<uint2*>(&As[(tid / TPR + RPP * i) * LDA + (tid % TPR) * 4]) = p;
            if (X3) {
                uint2 q;
                q.x = pack_bf16_2<PM>(ra[i].x - __uint_as_float(p.x << 16), ra[i].y - __uint_as_float(p.x & 0xffff0000u));
                q.y = pack_bf16_2<PM>(ra[i].z - __uint_as_float(p.y << 16), ra[i].w - __uint_as_float(p.y & 0xffff0000u));
                *reinterpret_cast<uint2*>(&As[BP * LDA + (tid / TPR + RPP * i) * LDA + (tid % TPR) * 4]) = q;
            }
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            uint2 p;
            p.x = pack_bf16_2<PM>(rb[i].x, rb[i].y); p.y = pack_bf16_2<PM>(rb[i].z, rb[i].w);
            *reinterpret_cast<uint2*>(&Bs[(tid / (BN / 4) + (256 / (BN / 4)) * i) * LDB + (tid % (BN / 4)) * 4]) = p;
            if (X3) {
                uint2 q;
                q.x = pack_bf16_2<PM>(rb[i].x - __uint_as_float(p.x << 16), rb[i].y - __uint_as_float(p.x & 0xffff0000u));
                q.y = pack_bf16_2<PM>(rb[i].z - __uint_as_float(p.y << 16), rb[i].w - __uint_as_float(p.y & 0xffff0000u));
                *reinterpret_cast<uint2*>(&Bs[BP * LDB + (tid / (BN / 4) + (256 / (BN / 4)) * i) * LDB + (tid % (BN / 4)) * 4]) = q;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing (see header): group gq = lane >> 4, t = lane & 15
    const int gq = lane >> 4, t = lane & 15;
    const int f_pos = 8 * (gq >> 1) + (t >> 2);          // + kk + 4 r
    const int f_ch = 16 * (gq & 1) + 4 * (t & 3);        // + 32 * tile + wave offset
    const unsigned a_base = (unsigned)(size_t)(&As[0]) + 2u * (unsigned)(f_pos * LDA + wm * (BM / WM) + f_ch);
    const unsigned b_base = (unsigned)(size_t)(&Bs[0]) + 2u * (unsigned)(f_pos * LDB + wn * (BN / WN) + f_ch);

    if (kt_begin < nkt) load_tile(kt_begin);
    for (long long kt = kt_begin; kt < nkt; ++kt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) {
            if (!plain) {
#pragma unroll
                for (int i = 0; i < A_F4; ++i) advance(aw[i], ah[i], ad[i], ab[i]);
#pragma unroll
                for (int i = 0; i < B_F4; ++i) advance(bw[i], bh[i], bd[i], bb[i]);
            }
            load_tile(kt + 1);
        }
#pragma unroll
        for (int kk = 0; kk < BP; kk += 16) {
            unsigned long long a0[TM], a1[TM], b0[TN], b1[TN];
            unsigned long long al0[TM], al1[TM], bl0[TN], bl1[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a0[i] = ds_read_tr16(a_base + 2u * (unsigned)((kk + 0) * LDA + i * 32));
                a1[i] = ds_read_tr16(a_base + 2u * (unsigned)((kk + 4) * LDA + i * 32));
                if (X3) {
                    al0[i] = ds_read_tr16(a_base + 2u * (unsigned)(BP * LDA + (kk + 0) * LDA + i * 32));
                    al1[i] = ds_read_tr16(a_base + 2u * (unsigned)(BP * LDA + (kk + 4) * LDA + i * 32));
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                b0[j] = ds_read_tr16(b_base + 2u * (unsigned)((kk + 0) * LDB + j * 32));
                b1[j] = ds_read_tr16(b_base + 2u * (unsigned)((kk + 4) * LDB + j * 32));
                if (X3) {
                    bl0[j] = ds_read_tr16(b_base + 2u * (unsigned)(BP * LDB + (kk + 0) * LDB + j * 32));
                    bl1[j] = ds_read_tr16(b_base + 2u * (unsigned)(BP * LDB + (kk + 4) * LDB + j * 32));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (X3) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    union { unsigned long long u[2]; bf16x8 v; } fa, fl;
                    fa.u[0] = a0[i]; fa.u[1] = a1[i];
                    fl.u[0] = al0[i]; fl.u[1] = al1[i];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        union { unsigned long long u[2]; bf16x8 v; } fb, fm;
                        fb.u[0] = b0[j]; fb.u[1] = b1[j];
                        fm.u[0] = bl0[j]; fm.u[1] = bl1[j];
                        acc[i][j] = wg_mfma<PM>(fl.v, fb.v, acc[i][j]);
                        acc[i][j] = wg_mfma<PM>(fa.v, fm.v, acc[i][j]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                union { unsigned long long u[2]; bf16x8 v; } fa;
                fa.u[0] = a0[i]; fa.u[1] = a1[i];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    union { unsigned long long u[2]; bf16x8 v; } fb;
                    fb.u[0] = b0[j]; fb.u[1] = b1[j];
                    acc[i][j] = wg_mfma<PM>(fa.v, fb.v, acc[i][j]);
                }
            }
        }
    }

    if (want_amax) {
        __shared__ unsigned wamx[4];
        unsigned amx = vxb_amax_word(amxf, nanw);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
        if (lane == 0) wamx[wid] = amx;
        __syncthreads();
        if (tid == 0) {
            const int nblk = g.grad_is_src0 ? gridDim.y : gridDim.x, blk = g.grad_is_src0 ? blockIdx.y : blockIdx.x;
            g.amax_part[(long long)blockIdx.z * nblk + blk] = max(max(wamx[0], wamx[1]), max(wamx[2], wamx[3]));
        }
    }
    if (want_psum) {
        // fold the RPP position lanes (tid / TPR) of every row quad in a fixed order
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(&As[0]);          // 256 float4 = 4 KB <= the A tile
        red[tid] = psum;
        __syncthreads();
        if (tid < TPR && a_ok) {
            float4 a = red[tid];
#pragma unroll
            for (int j = 1; j < RPP; ++j) { const float4 b = red[tid + TPR * j]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            *reinterpret_cast<float4*>(g.possum + (long long)blockIdx.z * g.Krows + kr) = a;
        }
    }
    float* __restrict__ C = g.part + (long long)blockIdx.z * g.Krows * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
            if (n >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < g.Krows) C[(long long)m * g.N + n] = acc[i][j][r];
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The plain-GEMM form (weight gradient of a linear layer: part[z][r][n] = sum_p src0[p][r] * dy[p][n]) on single fp16 products,
// pipelined: the generic kernel above spends two barriers and an LDS drain on every 32-position tile of 8 MFMAs per wave (12 % of the
// matrix pipe busy, half of the wave time in s_waitcnt: profiles/r03_v3_sq_summary.txt).  Here the LDS tiles are double-buffered:
// tile t+1 is converted and stored into the other stage while tile t is multiplied, the global loads of tile t+2 are issued before
// that, and there is ONE barrier per tile; the transposed fragment reads go through the compiler builtin (immediate offsets, waits
// placed by the compiler).  Same tile shape, same operand rounding, same accumulation order as the generic kernel: bit-identical.
typedef short wl_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long wl_tr16(const u16* p) {
    union { wl_v4s v; unsigned long long u; } t;
    t.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wl_v4s*)p);
    return t.u;
}

template <int BN>
__global__ void __launch_bounds__(256) wgrad_lin_f16_kernel(WgArgs g) {
    constexpr int BM = 128, PM = 2;
    constexpr int LDA = BM + 32;
    constexpr int LDB = BN == 128 ? 160 : 96;
    constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_F4 = BP * BM / 4 / 256;       // 4
    constexpr int B_F4 = BP * BN / 4 / 256;       // 4 or 2
    constexpr int BTR = BN / 4;                   // threads per position row of the B tile
    __shared__ __attribute__((aligned(16))) u16 As[2][BP * LDA];
    __shared__ __attribute__((aligned(16))) u16 Bs[2][BP * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kr = m0 + (tid & 31) * 4;
    const bool a_ok = kr < g.Krows;
    const int bn = n0 + (tid % BTR) * 4;
    const bool b_ok = bn < g.N;
    const float* __restrict__ ap = g.src0 + kr;
    const float* __restrict__ bp = g.dy + bn;
    const long long lda = g.C0;

    long long nkt = (g.P + BP - 1) / BP;
    const long long kt_begin = (long long)blockIdx.z * g.tiles_per_split;
    if (nkt > kt_begin + g.tiles_per_split) nkt = kt_begin + g.tiles_per_split;

    float4 ra[A_F4], rb[B_F4];
    auto load_tile = [&](long long kt) {
        const long long k0 = kt * BP;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const long long pos = k0 + (tid >> 5) + 8 * i;
            ra[i] = (a_ok && pos < g.P) ? *reinterpret_cast<const float4*>(ap + pos * lda) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const long long pos = k0 + tid / BTR + (256 / BTR) * i;
            rb[i] = (b_ok && pos < g.P) ? *reinterpret_cast<const float4*>(bp + pos * g.ldy) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const float sc_a = (g.scale && g.grad_is_src0) ? g.scale[0] : 1.0f;
    const float sc_b = (g.scale && !g.grad_is_src0) ? g.scale[0] : 1.0f;
    const bool want_psum = g.possum != nullptr && blockIdx.x == 0;
    const bool want_amax = g.amax_part != nullptr && (g.grad_is_src0 ? blockIdx.x == 0 : blockIdx.y == 0);
    float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
    float amxf = 0.f, nanw = 0.f;          // largest |gradient operand| seen; nanw: NaN once a NaN / inf was seen (fmaxf drops NaN)
    auto store_tile = [&](int stage) {
        if (want_psum) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) { psum.x += ra[i].x; psum.y += ra[i].y; psum.z += ra[i].z; psum.w += ra[i].w; }
        }
        if (want_amax) {
            if (g.grad_is_src0) {
#pragma unroll
                for (int i = 0; i < A_F4; ++i) {
                    amxf = fmaxf(fmaxf(amxf, fabsf(ra[i].x)), fmaxf(fmaxf(fabsf(ra[i].y), fabsf(ra[i].z)), fabsf(ra[i].w)));
                    nanw = fmaf((ra[i].x + ra[i].y) + (ra[i].z + ra[i].w), 0.0f, nanw);
                }
            } else {
#pragma unroll
                for (int i = 0; i < B_F4; ++i) {
                    amxf = fmaxf(fmaxf(amxf, fabsf(rb[i].x)), fmaxf(fmaxf(fabsf(rb[i].y), fabsf(rb[i].z)), fabsf(rb[i].w)));
                    nanw = fmaf((rb[i].x + rb[i].y) + (rb[i].z + rb[i].w), 0.0f, nanw);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            uint2 p;
            p.x = pack_bf16_2<PM>(ra[i].x * sc_a, ra[i].y * sc_a); p.y = pack_bf16_2<PM>(ra[i].z * sc_a, ra[i].w * sc_a);
            *reinterpret_cast<uint2*>(&As[stage][((tid >> 5) + 8 * i) * LDA + (tid & 31) * 4]) = p;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            uint2 p;
            p.x = pack_bf16_2<PM>(rb[i].x * sc_b, rb[i].y * sc_b); p.y = pack_bf16_2<PM>(rb[i].z * sc_b, rb[i].w * sc_b);
            *reinterpret_cast<uint2*>(&Bs[stage][(tid / BTR + (256 / BTR) * i) * LDB + (tid % BTR) * 4]) = p;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing as in the generic kernel: group gq = lane >> 4, t = lane & 15
    const int gq = lane >> 4, t = lane & 15;
    const int f_pos = 8 * (gq >> 1) + (t >> 2);
    const int f_ch = 16 * (gq & 1) + 4 * (t & 3);
    const int a_lane = f_pos * LDA + wm * (BM / WM) + f_ch;
    const int b_lane = f_pos * LDB + wn * (BN / WN) + f_ch;

    if (kt_begin < nkt) {
        load_tile(kt_begin);
        store_tile(0);
        if (kt_begin + 1 < nkt) load_tile(kt_begin + 1);
    }
    __syncthreads();
    for (long long kt = kt_begin; kt < nkt; ++kt) {
        const int cur = (int)(kt - kt_begin) & 1;
        if (kt + 1 < nkt) store_tile(cur ^ 1);                 // tile kt + 1 (in registers since the previous iteration)
        if (kt + 2 < nkt) load_tile(kt + 2);
        const u16* at = &As[cur][a_lane];
        const u16* bt = &Bs[cur][b_lane];
#pragma unroll
        for (int kk = 0; kk < BP; kk += 16) {
            union { unsigned long long u[2]; bf16x8 v; } fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fa[i].u[0] = wl_tr16(at + (kk + 0) * LDA + i * 32);
                fa[i].u[1] = wl_tr16(at + (kk + 4) * LDA + i * 32);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fb[j].u[0] = wl_tr16(bt + (kk + 0) * LDB + j * 32);
                fb[j].u[1] = wl_tr16(bt + (kk + 4) * LDB + j * 32);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = wg_mfma<PM>(fa[i].v, fb[j].v, acc[i][j]);
        }
        __syncthreads();
    }

    if (want_amax) {
        __shared__ unsigned wamx[4];
        unsigned amx = vxb_amax_word(amxf, nanw);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
        if (lane == 0) wamx[wid] = amx;
        __syncthreads();
        if (tid == 0) {
            const int nblk = g.grad_is_src0 ? gridDim.y : gridDim.x, blk = g.grad_is_src0 ? blockIdx.y : blockIdx.x;
            g.amax_part[(long long)blockIdx.z * nblk + blk] = max(max(wamx[0], wamx[1]), max(wamx[2], wamx[3]));
        }
    }
    if (want_psum) {
        // fold the 8 position lanes (tid >> 5) of every row quad in a fixed order (the last barrier of the loop freed the tiles)
        float4* red = reinterpret_cast<float4*>(&As[0][0]);       // 256 float4 = 4 KB <= one A stage
        red[tid] = psum;
        __syncthreads();
        if (tid < 32 && a_ok) {
            float4 a = red[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) { const float4 b = red[tid + 32 * j]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            *reinterpret_cast<float4*>(g.possum + (long long)blockIdx.z * g.Krows + kr) = a;
        }
    }
    float* __restrict__ C = g.part + (long long)blockIdx.z * g.Krows * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
            if (n >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < g.Krows) C[(long long)m * g.N + n] = acc[i][j][r];
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Wide form of the same plain GEMM for the linear layers whose narrower operand has exactly 512 channels (every big linear layer of
// the Perceiver trunk: 4096x512, 512x2048, 512x512, 1024x512).  The 128 x 128 kernels move 128 B of fp32 operands per CU and clock
// at full matrix rate and re-read the wide operand once per 128-column block from beyond the L2 (537 MB x 4 for the GEGLU up-projection):
// that traffic, not the matrix pipe (12-15 % busy), is what they wait for.  Here a workgroup of 8 waves owns 128 channels of the WIDE
// operand U and ALL 512 channels of the narrow operand V: U is read from HBM exactly once, V (67 MB) is re-read out of the L2 /
// Infinity Cache, and a staged byte feeds 2.6x the MFMAs.  T[i][j] = sum_p U[p][i] * V[p][j]; the host says which of (src0, dy) is U
// and the tile is written through (so_i, so_j) strides, i.e. transposed when U is the `dy` argument.
//   waves 2 (i) x 4 (j): wave tile 64 x 128 = 2 x 4 MFMA tiles (128 accumulator VGPRs); 16-position tiles (one k-step), two LDS
//   stages (2 x 22 KB); tile t+1 is converted + stored while tile t is multiplied, and the loads of tiles t+2 .. t+4 are in flight in
//   three rotating register sets (120 KB per CU) -- a loaded tile takes ~2.5 us to arrive and nothing else hides that with one
//   workgroup per CU; one barrier per tile.
struct WideArgs {
    const float* U; const float* V;       // [P][ldu], [P][ldv] fp32
    long long ldu, ldv;
    int Ru;                                // channels of U (a multiple of 128); V has 512
    long long P;
    int tiles_per_split;
    float* part;                           // [nsplit][...]: element (i, j) at i * so_i + j * so_j, slices part_stride apart
    long long so_i, so_j, part_stride;
    const float* scale;                    // scale[0] multiplies the gradient operand (U when grad_is_u, else V)
    int grad_is_u;
    unsigned* amax_part;                   // optional: [z][i-tiles] (grad_is_u) or [z] (else: written by the i-tile 0 workgroups)
    float* possum;                         // optional: position sums of src0's channels: [z][channels of src0]
    int possum_is_u;                       // src0 is U (its 128 channels of this tile) or V (all 512, i-tile 0 only)
    int possum_stride;                     // channels of src0
    int dbg_order;                         // experiment: 1 = workgroups in the plain (i tile fastest) launch order
    int dbg_slow;                          // experiment: 1 = every step through the guarded form (rounds 3 - 5)
};

// WP = positions per tile: 16 (one MFMA k-step, three register sets of loads in flight: rounds 3 - 5) or 32 (round 6: two k-steps per tile and
// barrier, two register sets -- the 16-position tile gives a wave 8 MFMAs between barriers, 16.4 % matrix-pipe duty and 53 % of its time in
// s_waitcnt, profiles/r06_v2_sq_summary.txt: the LDS store -> barrier -> transposed read -> MFMA chain of a tile is not covered)
template <int WP, int AFL, int PFL>
__global__ void __launch_bounds__(512) wgrad_wide_f16_kernel(WideArgs g) {
    constexpr int PM = 2, BI = 128, BJ = 512;
    constexpr int NSET = WP == 16 ? 3 : 2, NA = WP / 16, NB = WP / 4;         // register sets; A / B loads per thread and tile
    constexpr int LDA = BI + 32, LDB = BJ + 32;          // 80 / 272 dwords per position row: both = 16 (mod 64)
    extern __shared__ __attribute__((aligned(16))) u16 wsm[];
    u16* As0 = wsm;                                      // [2][WP * LDA]
    u16* Bs0 = wsm + 2 * WP * LDA;                       // [2][WP * LDB]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    // Workgroup -> (i tile bi, position slice bz).  Every i tile of a slice streams the SAME rows of V: hardware places linear workgroup id
    // L on XCD L % 8 (each with a private L2), so the ids are permuted to give an XCD a contiguous run of (slice, i tile) pairs -- the i
    // tiles of a slice run side by side behind ONE L2 and V comes out of it for all but the first (round 6; before, the 32 i tiles of the
    // GEGLU up-projection's gradient were spread over all eight XCDs and V was re-read 32 x from beyond the L2: 2.1 GB per launch).
    // Bijective for any grid; only speed depends on it.  g.dbg_order != 0: the plain order (A/B).
    int bi = blockIdx.x, bz = blockIdx.y;
    if (!g.dbg_order) {
        const int nwg = gridDim.x * gridDim.y, lid = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = lid & 7, slot = lid >> 3;
        const int q_ = nwg >> 3, r_ = nwg & 7;
        const int t = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot;
        bi = t % gridDim.x;
        bz = t / gridDim.x;
    }
    const int i0 = bi * BI;
    // tile rows: A (U): pos = tid / 32 (16 rows x 32 quads); B (V): pos = tid / 128 + 4 i (4 rows per pass x 128 quads)
    long long nkt = g.P / WP;                            // (host: P % 16 == 0)
    const long long kt_begin = (long long)bz * g.tiles_per_split;
    if (nkt > kt_begin + g.tiles_per_split) nkt = kt_begin + g.tiles_per_split;
    const int nt = nkt > kt_begin ? (int)(nkt - kt_begin) : 0;
    // running load pointers: tiles are loaded strictly in order (0, 1, 2, ...), each load advances them by one tile
    const float* __restrict__ up = g.U + i0 + (tid & 31) * 4 + (kt_begin * WP + (tid >> 5)) * g.ldu;
    const float* __restrict__ vp = g.V + (tid & 127) * 4 + (kt_begin * WP + (tid >> 7)) * g.ldv;
    const long long u16s = 16 * g.ldu, v4 = 4 * g.ldv;

    // three register sets: the loads of tiles t+2, t+3, t+4 are in flight while tile t is multiplied (a workgroup per CU: nothing
    // else hides the ~2.5 us a loaded tile takes to arrive)
    float4 ra[NSET][NA], rb[NSET][NB];
#define WW_LOAD(S, t_)                                                                                               \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) { ra[S][i] = *reinterpret_cast<const float4*>(up); up += u16s; } \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) { rb[S][i] = *reinterpret_cast<const float4*>(vp); vp += v4; } \
    }
    const float sc_a = (g.scale && g.grad_is_u) ? g.scale[0] : 1.0f;
    const float sc_b = (g.scale && !g.grad_is_u) ? g.scale[0] : 1.0f;
    // AFL / PFL (host: which operand the |x| maximum / the position sums are taken of: 0 none, 1 U, 2 V).  The V-side work is the same
    // in every i tile of a slice; all of them do it and the i tile 0 workgroup writes it -- one instruction stream per kernel (two loop
    // bodies in one kernel, with and without the V-side work, spilled 50 - 60 registers), ~24 more VALU instructions per tile next to
    // the 60 of the conversion, issued between the MFMAs
    constexpr bool psum_u = PFL == 1, psum_v = PFL == 2, amax_u = AFL == 1, amax_v = AFL == 2;
    constexpr int AF = AFL, PF = PFL;
    float4 ps = make_float4(0.f, 0.f, 0.f, 0.f);        // position sums: of this thread's U quad (psum_u) or V quad (psum_v), never both
    float amxf = 0.f;                      // largest |gradient operand| seen
    vxb_f32x2 nanw = {0.f, 0.f};               // NaN once a NaN / inf was seen (fmaxf drops NaN): x * 0 accumulated, two lanes at a time
    const int a_st = (tid >> 5) * LDA + (tid & 31) * 4;
    const int b_st = (tid >> 7) * LDB + (tid & 127) * 4;
#define WW_AMAX4(v)                                                                                                  \
    {                                                                                                                \
        amxf = fmaxf(fmaxf(amxf, fabsf((v).x)), fmaxf(fmaxf(fabsf((v).y), fabsf((v).z)), fabsf((v).w)));             \
        nanw = __builtin_elementwise_fma(vxb_f32x2{(v).x, (v).y}, vxb_f32x2{0.f, 0.f}, nanw);                                \
        nanw = __builtin_elementwise_fma(vxb_f32x2{(v).z, (v).w}, vxb_f32x2{0.f, 0.f}, nanw);                                \
    }
#define WW_FLAGS_A(S, i)                                                                                             \
    {                                                                                                                \
        if (PF == 1) { ps.x += ra[S][i].x; ps.y += ra[S][i].y; ps.z += ra[S][i].z; ps.w += ra[S][i].w; }             \
        if (AF == 1) WW_AMAX4(ra[S][i])                                                                              \
    }
#define WW_FLAGS_B(S, i)                                                                                             \
    {                                                                                                                \
        if (PF == 2) { ps.x += rb[S][i].x; ps.y += rb[S][i].y; ps.z += rb[S][i].z; ps.w += rb[S][i].w; }             \
        if (AF == 2) WW_AMAX4(rb[S][i])                                                                              \
    }
#define WW_STORE(S, stage_)                                                                                          \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) WW_FLAGS_A(S, i)                                               \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) WW_FLAGS_B(S, i)                                               \
        u16* as_ = As0 + (stage_) * WP * LDA;                                                                        \
        u16* bs_ = Bs0 + (stage_) * WP * LDB;                                                                        \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                              \
            uint2 p_;                                                                                                \
            p_.x = pack_bf16_2<PM>(ra[S][i].x * sc_a, ra[S][i].y * sc_a); p_.y = pack_bf16_2<PM>(ra[S][i].z * sc_a, ra[S][i].w * sc_a); \
            *reinterpret_cast<uint2*>(&as_[a_st + 16 * i * LDA]) = p_;                                               \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                              \
            uint2 q_;                                                                                                \
            q_.x = pack_bf16_2<PM>(rb[S][i].x * sc_b, rb[S][i].y * sc_b); q_.y = pack_bf16_2<PM>(rb[S][i].z * sc_b, rb[S][i].w * sc_b); \
            *reinterpret_cast<uint2*>(&bs_[b_st + 4 * i * LDB]) = q_;                                                \
        }                                                                                                            \
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int gq = lane >> 4, t16 = lane & 15;
    const int f_pos = 8 * (gq >> 1) + (t16 >> 2);
    const int f_ch = 16 * (gq & 1) + 4 * (t16 & 3);
    const int a_lane = f_pos * LDA + wm * 64 + f_ch;
    const int b_lane = f_pos * LDB + wn * 128 + f_ch;

    // one step: tile t is multiplied out of LDS stage t & 1; tile t+1 (register set S1 = (t+1) % 3) is converted into the other
    // stage and that set is refilled with tile t+4
#define WW_STEP(t_, S1)                                                                                              \
    {                                                                                                                \
        const int tt_ = (t_);                                                                                        \
        if (tt_ < nt) {                                                                                              \
            const int cur_ = tt_ & 1;                                                                                \
            if (tt_ + 1 < nt) WW_STORE(S1, cur_ ^ 1)                                                                 \
            if (tt_ + 1 + NSET < nt) WW_LOAD(S1, tt_ + 1 + NSET)                                                     \
            _Pragma("unroll") for (int ks = 0; ks < NA; ++ks) {                                                       \
                const u16* at_ = As0 + cur_ * WP * LDA + ks * 16 * LDA + a_lane;                                     \
                const u16* bt_ = Bs0 + cur_ * WP * LDB + ks * 16 * LDB + b_lane;                                     \
                union { unsigned long long u[2]; bf16x8 v; } fa_[2], fb_[4];                                         \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
                    fa_[i].u[0] = wl_tr16(at_ + i * 32);                                                             \
                    fa_[i].u[1] = wl_tr16(at_ + 4 * LDA + i * 32);                                                   \
                }                                                                                                    \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                       \
                    fb_[j].u[0] = wl_tr16(bt_ + j * 32);                                                             \
                    fb_[j].u[1] = wl_tr16(bt_ + 4 * LDB + j * 32);                                                   \
                }                                                                                                    \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = wg_mfma<PM>(fa_[i].v, fb_[j].v, acc[i][j]); \
            }                                                                                                        \
            __syncthreads();                                                                                         \
        }                                                                                                            \
    }
    if (nt > 0) {
        WW_LOAD(0, 0)
        WW_STORE(0, 0)
        if (nt > 1) WW_LOAD(1, 1)
        if (NSET == 3) {
            if (nt > 2) WW_LOAD(2 % NSET, 2)
            if (nt > 3) WW_LOAD(0, 3)
        } else if (nt > 2) WW_LOAD(0, 2)
    }
    __syncthreads();

    // ---- the main loop: steps whose loads and stores are all unconditional, the flag work chosen at compile time (round 6) ----------------
    // In the guarded step above the compiler cannot count the loads in flight across the `if`s and waits for all but the last four
    // (s_waitcnt vmcnt(4) at the top of every step): the set loaded ONE step ago has to arrive, not the one loaded three steps ago, and
    // every tile pays a memory round trip.  Here a step is one basic block: exact vmcnt, and the conversion of tile t+1 is issued between
    // the MFMAs of tile t (pinned with sched_barrier) instead of in front of them with every wave of the workgroup in the same phase.
    int t0 = 0;
    // (a register quad is reloaded -- with the tile NSET steps on -- right after its conversion: the loads of a step are spread between its
    //  MFMAs, so that a load waiting to be accepted by the memory pipeline stalls the wave while matrix work is already in the pipe)
#if VXB_ABL & 1
#define WW_ABL_LOAD_A(S, i) { up += u16s; }
#define WW_ABL_LOAD_B(S, i) { vp += v4; }
#else
#define WW_ABL_LOAD_A(S, i) { ra[S][i] = *reinterpret_cast<const float4*>(up); up += u16s; }
#define WW_ABL_LOAD_B(S, i) { rb[S][i] = *reinterpret_cast<const float4*>(vp); vp += v4; }
#endif
#if VXB_ABL & 2
#define WW_ABL_MFMA(a, b, c) ({ asm volatile("" :: "v"(a), "v"(b)); c; })
#else
#define WW_ABL_MFMA(a, b, c) wg_mfma<PM>(a, b, c)
#endif
#define WW_CVT_A(S, i, as_)                                                                                          \
    {                                                                                                                \
        WW_FLAGS_A(S, i)                                                                                             \
        uint2 p_;                                                                                                    \
        p_.x = pack_bf16_2<PM>(ra[S][i].x * sc_a, ra[S][i].y * sc_a); p_.y = pack_bf16_2<PM>(ra[S][i].z * sc_a, ra[S][i].w * sc_a); \
        *reinterpret_cast<uint2*>(&(as_)[a_st + 16 * (i) * LDA]) = p_;                                               \
    }
#define WW_CVT_B(S, i, bs_)                                                                                          \
    {                                                                                                                \
        WW_FLAGS_B(S, i)                                                                                             \
        uint2 q_;                                                                                                    \
        q_.x = pack_bf16_2<PM>(rb[S][i].x * sc_b, rb[S][i].y * sc_b); q_.y = pack_bf16_2<PM>(rb[S][i].z * sc_b, rb[S][i].w * sc_b); \
        *reinterpret_cast<uint2*>(&(bs_)[b_st + 4 * (i) * LDB]) = q_;                                                \
    }
#if VXB_ABL & 4
#define WW_ABL_CVT_A(S, i, as_) { asm volatile("" :: "v"(ra[S][i].x), "v"(ra[S][i].y), "v"(ra[S][i].z), "v"(ra[S][i].w)); }
#define WW_ABL_CVT_B(S, i, bs_) { asm volatile("" :: "v"(rb[S][i].x), "v"(rb[S][i].y), "v"(rb[S][i].z), "v"(rb[S][i].w)); }
#else
#define WW_ABL_CVT_A(S, i, as_) WW_CVT_A(S, i, as_)
#define WW_ABL_CVT_B(S, i, bs_) WW_CVT_B(S, i, bs_)
#endif
#define WW_FSTEP(t_, S1)                                                                                             \
    {                                                                                                                \
        const int cur_ = (t_) & 1;                                                                                   \
        u16* as_ = As0 + (cur_ ^ 1) * WP * LDA;                                                                      \
        u16* bs_ = Bs0 + (cur_ ^ 1) * WP * LDB;                                                                      \
        _Pragma("unroll") for (int ks = 0; ks < NA; ++ks) {                                                           \
            const u16* at_ = As0 + cur_ * WP * LDA + ks * 16 * LDA + a_lane;                                         \
            const u16* bt_ = Bs0 + cur_ * WP * LDB + ks * 16 * LDB + b_lane;                                         \
            union { unsigned long long u[2]; bf16x8 v; } fa_[2], fb_[4];                                             \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
                fa_[i].u[0] = wl_tr16(at_ + i * 32);                                                                 \
                fa_[i].u[1] = wl_tr16(at_ + 4 * LDA + i * 32);                                                       \
            }                                                                                                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
                fb_[j].u[0] = wl_tr16(bt_ + j * 32);                                                                 \
                fb_[j].u[1] = wl_tr16(bt_ + 4 * LDB + j * 32);                                                       \
            }                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            WW_ABL_CVT_A(S1, ks, as_)                                                                                 \
            WW_ABL_LOAD_A(S1, ks)                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                           \
                acc[c >> 1][2 * (c & 1)] = WW_ABL_MFMA(fa_[c >> 1].v, fb_[2 * (c & 1)].v, acc[c >> 1][2 * (c & 1)]); \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
                WW_ABL_CVT_B(S1, 4 * ks + c, bs_)                                                                     \
                WW_ABL_LOAD_B(S1, 4 * ks + c)                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
                acc[c >> 1][2 * (c & 1) + 1] = WW_ABL_MFMA(fa_[c >> 1].v, fb_[2 * (c & 1) + 1].v, acc[c >> 1][2 * (c & 1) + 1]); \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
            }                                                                                                        \
        }                                                                                                            \
        __syncthreads();                                                                                             \
    }
    if (!g.dbg_slow) {
        if (NSET == 3) {
#pragma unroll 1
            for (; t0 + 3 + NSET < nt; t0 += 3) {          // (the last step of a round loads tile t0 + 3 + NSET)
                WW_FSTEP(t0, 1)
                WW_FSTEP(t0 + 1, 2 % NSET)
                WW_FSTEP(t0 + 2, 0)
            }
        } else {
#pragma unroll 1
            for (; t0 + 2 + NSET < nt; t0 += 2) {
                WW_FSTEP(t0, 1)
                WW_FSTEP(t0 + 1, 0)
            }
        }
    }
#undef WW_FSTEP
#undef WW_CVT_A
#undef WW_CVT_B
    // the remaining steps (and all of a short slice), guarded
    if (NSET == 3) {
#pragma unroll 1
        for (int t = t0; t < nt; t += 3) {
            WW_STEP(t, 1)
            WW_STEP(t + 1, 2 % NSET)
            WW_STEP(t + 2, 0)
        }
    } else {
#pragma unroll 1
        for (int t = t0; t < nt; t += 2) {
            WW_STEP(t, 1)
            WW_STEP(t + 1, 0)
        }
    }
#undef WW_LOAD
#undef WW_STORE
#undef WW_STEP
#undef WW_FLAGS_A
#undef WW_FLAGS_B
#undef WW_AMAX4

    if (amax_u || amax_v) {
        __shared__ unsigned wamx[8];
        unsigned amx = vxb_amax_word(amxf, nanw[0] + nanw[1]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
        if (lane == 0) wamx[wid] = amx;
        __syncthreads();
        if (tid == 0) {
            unsigned m = 0;
            for (int w = 0; w < 8; ++w) m = max(m, wamx[w]);
            if (amax_u) g.amax_part[(long long)bz * gridDim.x + bi] = m;
            else if (bi == 0) g.amax_part[bz] = m;
        }
    }
    if (psum_u || psum_v) {
        // fixed-order fold over the position lanes of a channel quad (the loop's last barrier freed the stages)
        float4* red = reinterpret_cast<float4*>(wsm);       // 512 float4 = 8 KB
        __syncthreads();
        red[tid] = ps;
        __syncthreads();
        if (psum_u && tid < 32) {
            float4 a = red[tid];
#pragma unroll
            for (int j = 1; j < 16; ++j) { const float4 b = red[tid + 32 * j]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            *reinterpret_cast<float4*>(g.possum + (long long)bz * g.possum_stride + i0 + tid * 4) = a;
        }
        if (psum_v && bi == 0 && tid < 128) {
            float4 a = red[tid];
#pragma unroll
            for (int j = 1; j < 4; ++j) { const float4 b = red[tid + 128 * j]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            *reinterpret_cast<float4*>(g.possum + (long long)bz * g.possum_stride + tid * 4) = a;
        }
    }
    float* __restrict__ C = g.part + (long long)bz * g.part_stride;
    if (g.so_i == 1) {
        // transposed tile (U is the `dy` argument): the 4 consecutive i of an accumulator register quad are contiguous in memory
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long jn = wn * 128 + j * 32 + (lane & 31);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const long long im = i0 + wm * 64 + i * 32 + 8 * r4 + 4 * (lane >> 5);
                    *reinterpret_cast<float4*>(C + jn * g.so_j + im) =
                        make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long jn = wn * 128 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long im = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[im * g.so_i + jn] = acc[i][j][r];
            }
        }
}

}  // namespace

static int g_wg_bm256 = 0;        // experiment knob (vxb_debug_set_wgrad_bm256): 256-row tiles for the fp16 products.  Measured (round 3, linear_bwd
                                  // at M = 32768, N x K = 4096x512 / 512x2048 / 1024x512): 0.548 / 0.271 / 0.149 ms against 0.491 / 0.239 / 0.128 ms
                                  // with 128-row tiles -- a quarter fewer LDS fragment reads per MFMA, but half the workgroups and 176 VGPRs: OFF

static long long g_wide_min_rows = 16384;      // positions from which the wide kernel's tiles x slices fill the chip (ops.WIDE_MIN_M)
static int g_wg_wide_wp = 16;     // positions per tile of the wide kernel (vxb_debug_set_wgrad_lin(mode + 32): 32 -- measured no faster, and it spills)
static int g_wg_wide_slow = 0;    // vxb_debug_set_wgrad_lin(mode + 64): no unconditional main loop in the wide kernel (A/B)
static int g_wg_wide_order = 0;   // vxb_debug_set_wgrad_lin(mode + 16): the wide kernel's workgroups in the plain launch order (A/B of round 6's XCD-aware order)
static int g_wg_lin = 2;          // plain-GEMM form of the fp16 products: 2 = wide kernel where it applies, else the pipelined 128^2 one;
                                  // 1 = pipelined 128^2 only; 0 = the generic kernel (vxb_debug_set_wgrad_lin: A/B switch)

static int wgrad_bf16_impl(int x3, const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                          int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                          int d2s_s, int d2s_C, float* part, int nsplit, float* possum, vxb_stream_t stream,
                          const float* scale = nullptr, int grad_is_src0 = 0, float* next_scale = nullptr, float* amax_ws = nullptr,
                          float* sum_dst = nullptr, int sum_accumulate = 0) {
    if (!src0 || !dy || !part || B < 1 || S_in < 1 || S_out < 1 || kext < 1 || stride < 1 || N < 1 || nsplit < 1) return VXB_EARG;
    if (possum && !(S_in == 1 && S_out == 1 && kext == 1 && off == 0 && C1 == 0 && d2s_s <= 0)) return VXB_EARG;     // plain GEMM form only
    if ((C0 & 3) || (C1 & 3) || C0 < 4 || (C1 > 0 && !src1) || (N & 3)) return VXB_ESIZE;
    if (d2s_s > 0 && (d2s_C < 4 || (d2s_C & 3) || N % d2s_C)) return VXB_EARG;
    if (d2s_s <= 0 && (ldy & 3)) return VXB_ESIZE;
    WgArgs g;
    g.src0 = src0; g.src1 = src1; g.dy = dy; g.part = part; g.possum = possum;
    g.scale = scale; g.grad_is_src0 = grad_is_src0; g.amax_part = (next_scale && amax_ws) ? reinterpret_cast<unsigned*>(amax_ws) : nullptr;
    g.C0 = C0; g.C1 = C1; g.S_in = S_in; g.S_out = S_out; g.stride = stride; g.kext = kext; g.off = off; g.replicate = replicate;
    const long long K = (long long)kext * kext * kext * (C0 + C1);
    if (K >= INT32_MAX) return VXB_ESIZE;
    g.Krows = (int)K; g.N = N; g.P = (long long)B * S_out * S_out * S_out; g.ldy = ldy; g.d2s_s = d2s_s; g.d2s_C = d2s_C;
    const long long nkt = (g.P + BP - 1) / BP;
    g.tiles_per_split = (int)((nkt + nsplit - 1) / nsplit);
    hipStream_t st = (hipStream_t)stream;
    int nblk;
    const bool plain = S_in == 1 && S_out == 1 && kext == 1 && off == 0 && C1 == 0 && d2s_s <= 0 && stride == 1;
    // (the wide kernel's 128 x 512 tiles need >= 16384 positions to fill the chip with tiles x slices; below that -- replay batches of
    //  1 .. 4 samples -- the 128^2 kernel with its finer split runs: ops.WIDE_MIN_M mirrors this)
    if (x3 == 2 && plain && g_wg_lin >= 2 && g.P >= g_wide_min_rows && !(((uintptr_t)src0 | (uintptr_t)dy) & 15)) {
        // wide form: the operand with exactly 512 channels is V (see wgrad_wide_f16_kernel); ops.wide_wgrad_tiles mirrors this test
        const bool c1 = N == 512 && K >= 128 && (K & 127) == 0;
        const bool c2 = !c1 && K == 512 && N >= 128 && (N & 127) == 0;
        if ((c1 || c2) && (g.P & 15) == 0) {
            WideArgs w;
            w.U = c1 ? src0 : dy; w.V = c1 ? dy : src0;
            w.ldu = c1 ? C0 : ldy; w.ldv = c1 ? ldy : C0;
            w.Ru = c1 ? (int)K : N;
            const int wp = (g_wg_wide_wp == 32 && (g.P & 31) == 0) ? 32 : 16;
            w.P = g.P; w.tiles_per_split = (int)((g.P / wp + nsplit - 1) / nsplit);
            w.part = part; w.part_stride = K * N;
            w.so_i = c1 ? N : 1; w.so_j = c1 ? 1 : N;
            w.scale = scale; w.grad_is_u = c1 ? grad_is_src0 : !grad_is_src0;
            w.amax_part = g.amax_part; w.possum = possum; w.possum_is_u = c1 ? 1 : 0; w.possum_stride = (int)K;
            w.dbg_order = g_wg_wide_order; w.dbg_slow = g_wg_wide_slow;
            const size_t lds = (size_t)2 * wp * (160 + 544) * sizeof(u16);
            const int tiles_i = w.Ru / 128;
            const int afl = w.amax_part ? (w.grad_is_u ? 1 : 2) : 0, pfl = w.possum ? (w.possum_is_u ? 1 : 2) : 0;
            void (*kern)(WideArgs) = nullptr;
#define VXB_WIDE_PICK(A_, P_)                                                                                        \
    if (afl == A_ && pfl == P_) kern = wp == 32 ? wgrad_wide_f16_kernel<32, A_, P_> : wgrad_wide_f16_kernel<16, A_, P_>;
            VXB_WIDE_PICK(0, 0) VXB_WIDE_PICK(1, 0) VXB_WIDE_PICK(2, 0)
            VXB_WIDE_PICK(0, 1) VXB_WIDE_PICK(1, 1) VXB_WIDE_PICK(2, 1)
            VXB_WIDE_PICK(0, 2) VXB_WIDE_PICK(1, 2) VXB_WIDE_PICK(2, 2)
#undef VXB_WIDE_PICK
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
            hipLaunchKernelGGL(kern, dim3(tiles_i, nsplit), dim3(512), lds, st, w);
            VXB_CHECK_LAUNCH();
            if (g.amax_part && sum_dst)
                return vxb_wgrad_finish_launch(part, nsplit, K * N, sum_dst, sum_accumulate, scale + 1, g.amax_part,
                                               (w.grad_is_u ? tiles_i : 1) * nsplit, next_scale, 5, st);
            if (g.amax_part) return vxb_absmax_finish_launch(g.amax_part, (w.grad_is_u ? tiles_i : 1) * nsplit, next_scale, st, 5);
            return VXB_OK;
        }
    }
    if (N > 64) {
        dim3 grid(vxb_cdiv(N, 128), vxb_cdiv(K, 128), nsplit);
        nblk = grad_is_src0 ? grid.y : grid.x;
        if (x3 == 2 && plain && g_wg_lin) {
            hipLaunchKernelGGL((wgrad_lin_f16_kernel<128>), grid, dim3(256), 0, st, g);
        } else if (x3 == 2 && K >= 512 && g_wg_bm256) {
            grid.y = vxb_cdiv(K, 256);
            nblk = grad_is_src0 ? grid.y : grid.x;
            hipLaunchKernelGGL((wgrad_bf16_kernel<128, 2, 256>), grid, dim3(256), 0, st, g);
        } else if (x3 == 2) hipLaunchKernelGGL((wgrad_bf16_kernel<128, 2>), grid, dim3(256), 0, st, g);
        else if (x3) hipLaunchKernelGGL((wgrad_bf16_kernel<128, 1>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((wgrad_bf16_kernel<128, 0>), grid, dim3(256), 0, st, g);
    } else {
        dim3 grid(vxb_cdiv(N, 64), vxb_cdiv(K, 128), nsplit);
        nblk = grad_is_src0 ? grid.y : grid.x;
        if (x3 == 2 && plain && g_wg_lin) hipLaunchKernelGGL((wgrad_lin_f16_kernel<64>), grid, dim3(256), 0, st, g);
        else if (x3 == 2) hipLaunchKernelGGL((wgrad_bf16_kernel<64, 2>), grid, dim3(256), 0, st, g);
        else if (x3) hipLaunchKernelGGL((wgrad_bf16_kernel<64, 1>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((wgrad_bf16_kernel<64, 0>), grid, dim3(256), 0, st, g);
    }
    VXB_CHECK_LAUNCH();
    // five bits of headroom: the scale is used by the NEXT step's launch, whose gradient may be up to 32x larger before anything
    // saturates (half still keeps 11 bits down to 2^-19 of the maximum: nothing is lost at the small end)
    if (g.amax_part && sum_dst)
        return vxb_wgrad_finish_launch(part, nsplit, K * N, sum_dst, sum_accumulate, scale + 1, g.amax_part, nblk * nsplit, next_scale, 5, st);
    if (g.amax_part) return vxb_absmax_finish_launch(g.amax_part, nblk * nsplit, next_scale, st, 5);
    return VXB_OK;
}

// workspace words of `amax_ws` for the entry below (an upper bound over the tile shapes the launch may pick)
extern "C" size_t vxb_conv3d_wgrad_f16_amax_words(int C0, int C1, int kext, int N, int nsplit, int grad_is_src0) {
    const long long K = (long long)kext * kext * kext * (C0 + C1);
    const int bn = N > 64 ? 128 : 64;
    return (size_t)nsplit * (size_t)(grad_is_src0 ? vxb_cdiv(K, 128) : vxb_cdiv(N, bn));
}
extern "C" void vxb_debug_set_wgrad_bm256(int on) { g_wg_bm256 = on ? 1 : 0; }
extern "C" void vxb_debug_set_wide_min_rows(int rows) { g_wide_min_rows = rows < 16 ? 16 : rows; }
extern "C" void vxb_debug_set_wgrad_lin(int mode) {
    g_wg_wide_slow = (mode >= 64); if (mode >= 64) mode -= 64;
    g_wg_wide_wp = (mode >= 32) ? 32 : 16; if (mode >= 32) mode -= 32;
    g_wg_wide_order = (mode >= 16); if (mode >= 16) mode -= 16;
    g_wg_lin = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
}

// ONE fp16 product per term (fp32 accumulate), same contract and `part` layout as the entries below.  The GRADIENT operand (src0 when
// grad_is_src0 != 0 -- the plain-GEMM form of a linear layer's weight gradient, src0 = its dY -- else dy) is multiplied by scale[0]
// (device, a power of two) before the conversion to half and `part` holds scale[0] * dW (undo it with vxb_sum_splits_dev_f32 and
// scale[1]); the other operand saturates at +-65504.  next_scale (optional, [2], needs amax_ws of vxb_conv3d_wgrad_f16_amax_words
// words): the scale vxb_absmax_scale_f32 would compute from the gradient operand as seen by THIS launch -- for the next step's call
// (delayed scaling: the operand's magnitude moves slowly from step to step; the reported scale leaves 5 bits of headroom, a 32-fold
// jump is still inside half's range).
extern "C" int vxb_conv3d_wgrad_f16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                        int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                                        int d2s_s, int d2s_C, float* part, int nsplit, float* possum, const float* scale,
                                        int grad_is_src0, float* next_scale, float* amax_ws, float* sum_dst, int sum_accumulate,
                                        vxb_stream_t stream) {
    // sum_dst (optional, with next_scale): [(tap, ci)][N] (+)= scale[1] * the sum of the nsplit partial results, from the same
    // finishing launch that computes next_scale -- the caller then skips vxb_sum_splits_dev_f32
    if (!scale || (sum_dst && !(next_scale && amax_ws))) return VXB_EARG;
    return wgrad_bf16_impl(2, src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, replicate, dy, N, ldy, d2s_s, d2s_C, part,
                           nsplit, possum, stream, scale, grad_is_src0, next_scale, amax_ws, sum_dst, sum_accumulate);
}

// Same contract as vxb_conv3d_wgrad_f32 (include/voxactb_hip.h); operands are rounded to bf16 while staged, fp32 accumulate.
extern "C" int vxb_conv3d_wgrad_bf16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                         int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                                         int d2s_s, int d2s_C, float* part, int nsplit, float* possum, vxb_stream_t stream) {
    return wgrad_bf16_impl(0, src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, replicate, dy, N, ldy, d2s_s, d2s_C, part,
                           nsplit, possum, stream);
}

// "bf16x3" twin: both operands split into hi/lo bf16 planes while staged; hi*hi + hi*lo + lo*hi, fp32 accumulate.
extern "C" int vxb_conv3d_wgrad_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                           int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                                           int d2s_s, int d2s_C, float* part, int nsplit, float* possum, vxb_stream_t stream) {
    return wgrad_bf16_impl(1, src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, replicate, dy, N, ldy, d2s_s, d2s_C, part,
                           nsplit, possum, stream);
}
