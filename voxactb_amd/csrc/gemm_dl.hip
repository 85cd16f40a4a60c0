// Direct-to-LDS GEMM / implicit-GEMM conv on the bf16 matrix cores ('bf16' and 'bf16x3' precisions).
//
// The generic kernels in gemm_conv.hip stage fp32 operands global -> VGPR -> (convert / split) -> ds_write -> LDS; per
// 32-deep k-tile that costs ~200 VALU and 12 wide ds_write instructions per thread next to 24 MFMAs, and it is what
// bounds them at 150-300 TF/s.  Here BOTH operands are bf16 planes in HBM already (weights are pre-split once per step,
// activations by one streaming pass, vxb_split_bf16_f32) and travel global -> LDS with `global_load_lds_dwordx4`:
// no VGPRs, no conversion, no ds_write; the loads of k-tile t+1 are in flight while the MFMAs of k-tile t run
// (two LDS stages, ONE barrier per k-tile).
//
// A wave-level direct load writes lane l's 16 bytes to LDS[base + 16 l] (measured: tools/ubench/ldsload_probe.hip), but
// every lane may fetch from any global address.  A 128-row x 32-bf16 operand tile is 512 slots of 16 bytes, slot
// (row, chunk) at index row*4 + (chunk ^ ((row >> 2) & 3)): one instruction fills 16 consecutive rows, and the XOR
// swizzle makes the fragment reads (ds_read_b128, 32 rows x one chunk per half-wave) conflict-free without padding --
// the swizzle is applied on the SOURCE side (which global chunk a lane fetches), since the destination is fixed.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int DL_A_KCONTIG = 0, DL_A_CONV = 1;
constexpr int DTILE = 128 * 32;            // u16 per operand plane tile (8 KB)

struct DlArgs {
    const u16* A;            // KCONTIG: bf16 [M][lda] plane(s); CONV: bf16 [B, S_in^3, C] plane(s)
    long long a_plane;       // u16 between the hi and the lo plane of A
    long long lda;
    const u16* Bw;           // bf16 [N][K] plane(s)
    long long b_plane;
    const u16* Bfrag;        // BD kernels: the same weights in MFMA fragment order [N/32][K/16][plane][lane 64][8] (ops.gemm_wfrag)
    const u16* zeros;        // >= 16 bytes of zeros (source of zero-padded conv taps)
    float* C;
    const float* bias;
    const float* residual;
    int M, N, K;
    long long ldc;
    int act;
    float slope;
    int accumulate;
    int d2s_s, d2s_G, d2s_C;
    int Cin, S_in, S_out, stride, kext, off, replicate;
    const unsigned* tapmask; // CONV, kext^3 <= 32: bit t of tapmask[column tile] clear -> the tile's weights of tap t are all zero
    unsigned d2s_magic;      // floor(2^32 / d2s_G) + 1 when M * d2s_G < 2^32 (exact multiply-high division), else 0
    const int* d2s_perm;     // d2s: column block p holds fine-grid phase d2s_perm[p] (nullptr = identity)
};

__device__ __forceinline__ void dl_load16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// BD: the B (weight) fragments come straight from global memory in fragment order, one k-tile ahead in two alternating
// register sets -- no weight traffic through LDS.  With both operands in LDS the 2x2-tile waves of the x3 kernel need
// 16 ds_read_b128 + 8 KB of direct loads per 24 MFMAs, which saturates the CU's 128 B/clk; the B half of that moves to
// the L1/L2 path (weights are shared by every workgroup) and the A stages shrink to 16 KB.
// TN = 32-column MFMA tiles per wave: 2 (128-column workgroup tiles) or, BD only, 1 (64-column tiles for N <= 64: the 5^3 conv of the
// decoder's up-block has 64 output channels, and half of a 128-column tile's MFMAs would multiply zero padding)
template <int AMODE, int X3, int BD, int TN = 2>
__global__ void __launch_bounds__(256, 2) gemm_dl_kernel(DlArgs g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int NPL = 1 + X3;
    constexpr int STAGE = (BD ? 1 : 2) * NPL * DTILE;     // [A planes][B planes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    // XCD-aware tile order (see gemm_conv.hip)
    int tile_x, tile_y;
    {
        const int gx = gridDim.x, nwg = gridDim.x * gridDim.y;
        const int lid = blockIdx.y * gx + blockIdx.x;
        const int xcd = lid & 7, slot = lid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        const int lid2 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        tile_x = lid2 % gx;
        tile_y = lid2 / gx;
    }
    constexpr int BN = 64 * TN;
    const int m0 = tile_y * 128, n0 = tile_x * BN;

    // ---- load slots of this lane: instruction (wid, i) fills rows (2 wid + i) * 16 .. +15; lane -> row + (lane >> 2),
    //      destination chunk position lane & 3, i.e. source chunk (lane & 3) ^ ((row >> 2) & 3)
    int lrow[2], lchunk[2];
    long long a_off[2], b_off[2];
    int a_b[2], a_d[2], a_h[2], a_w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        lrow[i] = (2 * wid + i) * 16 + (lane >> 2);
        lchunk[i] = (lane & 3) ^ ((lrow[i] >> 2) & 3);
        const int m = min(m0 + lrow[i], g.M - 1);
        const int n = min(n0 + lrow[i], g.N - 1);
        b_off[i] = (long long)n * g.K + lchunk[i] * 8;
        if (AMODE == DL_A_KCONTIG) {
            a_off[i] = (long long)m * g.lda + lchunk[i] * 8;
        } else {
            int r = m;
            const int S = g.S_out;
            a_w[i] = r % S; r /= S;
            a_h[i] = r % S; r /= S;
            a_d[i] = r % S; r /= S;
            a_b[i] = r;
            a_off[i] = 0;
        }
    }

    int it_cc = 0, it_tw = 0, it_th = 0, it_td = 0;       // conv gather iterator of the NEXT k-tile to be issued
    // block-sparse weights (polyphase up-conv: a phase only sees the low-res taps its interpolation footprint reaches):
    // the k loop visits only the taps whose bit is set in this column tile's mask
    unsigned it_rem = 0xffffffffu;
    int it_kt = 0;
    if (AMODE == DL_A_CONV && g.tapmask) it_rem = g.tapmask[tile_x];
    const bool masked = AMODE == DL_A_CONV && g.tapmask != nullptr;
    int k0_issued = 0;                      // weight k offset of the k-tile issued last (the BD fragment loads follow it)
    auto issue = [&](int stage, int kt) {
        int k0 = kt * 32;
        u16* sb = smem + stage * STAGE;
        const u16* asrc[2];
        if (AMODE == DL_A_KCONTIG) {
#pragma unroll
            for (int i = 0; i < 2; ++i) asrc[i] = g.A + a_off[i] + k0;
        } else {
            // k-tiles are issued in order: (tap, channel offset) advance incrementally, and the gathered voxel of a row
            // is recomputed only when the tap changes (every Cin / 32 k-tiles) -- no divisions in the loop
            if (masked && it_cc == 0) {
                const int tap = __builtin_ctz(it_rem);
                it_rem &= it_rem - 1;
                it_tw = tap % g.kext; it_th = (tap / g.kext) % g.kext; it_td = tap / (g.kext * g.kext);
                it_kt = tap * (g.Cin >> 5);
            }
            if (masked) k0 = (it_kt + (it_cc >> 5)) * 32;
            if (it_cc == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int id = a_d[i] * g.stride + it_td + g.off;
                    int ih = a_h[i] * g.stride + it_th + g.off;
                    int iw = a_w[i] * g.stride + it_tw + g.off;
                    bool ok = true;
                    if (g.replicate) {
                        id = min(max(id, 0), g.S_in - 1); ih = min(max(ih, 0), g.S_in - 1); iw = min(max(iw, 0), g.S_in - 1);
                    } else {
                        ok = id >= 0 && id < g.S_in && ih >= 0 && ih < g.S_in && iw >= 0 && iw < g.S_in;
                    }
                    const long long vox = (((long long)a_b[i] * g.S_in + id) * g.S_in + ih) * g.S_in + iw;
                    a_off[i] = ok ? vox * g.Cin + lchunk[i] * 8 : -1;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) asrc[i] = a_off[i] >= 0 ? g.A + a_off[i] + it_cc : nullptr;
            it_cc += 32;
            if (it_cc == g.Cin) {
                it_cc = 0;
                if (!masked) { if (++it_tw == g.kext) { it_tw = 0; if (++it_th == g.kext) { it_th = 0; ++it_td; } } }
            }
        }
        k0_issued = k0;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u16* s = asrc[i] ? asrc[i] + p * g.a_plane : g.zeros;
                dl_load16(s, sb + p * DTILE + (2 * wid + i) * 512);
                if (!BD) dl_load16(g.Bw + p * g.b_plane + b_off[i] + k0, sb + (NPL + p) * DTILE + (2 * wid + i) * 512);
            }
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment slots: lane (row = lane & 31, hi = lane >> 5) of tile t reads chunk 2 ks + hi of its row
    const int lm = lane & 31, hi = lane >> 5;
    int fa[2], fb[2], fx[2];                    // row * 32 u16 base and the row's swizzle key, per 32-row tile
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + lm, rb = wn * (32 * TN) + t * 32 + lm;
        fa[t] = ra * 32; fb[t] = rb * 32;
        fx[t] = (lm >> 2) & 3;                  // (row >> 2) & 3: the tile bases are multiples of 32, so only lm matters
    }

    const int nkt = masked ? __builtin_popcount(it_rem) * (g.Cin >> 5) : g.K / 32;
    if (BD) {
        // B fragments of tile t (32 columns), k-step ks, plane p: 1 KB at ((ntile * K/16 + k0/16 + ks) * NPL + p) * 512 + lane * 8
        bf16x8 b0[TN][2][NPL], b1[TN][2][NPL];
        const long long nks = g.K >> 4;
        const u16* bfb = g.Bfrag + ((long long)((n0 + wn * (32 * TN)) >> 5) * nks * NPL) * 512 + lane * 8;
        // The fragment loads are inline asm on purpose: the compiler's waitcnt pass cannot count register loads and
        // direct-to-LDS loads on one in-order counter and drains vmcnt to 0 before the first use of a loaded register --
        // i.e. it would wait for the A tiles deliberately left in flight.  Hidden from it, the loads are covered by the
        // hand-placed s_waitcnt at the top of the next k-tile (they are issued BEFORE that tile's direct loads).
#define DL_LOADB(SET, k0_)                                                                                            \
        _Pragma("unroll") for (int t = 0; t < TN; ++t) {                                                               \
            const u16* bp_ = bfb + (((long long)t * nks + ((k0_) >> 4)) * NPL) * 512;                                  \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
            _Pragma("unroll") for (int p = 0; p < NPL; ++p)                                                            \
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(SET[t][ks][p]) : "v"(bp_), "n"((ks * NPL + p) * 1024) : "memory"); \
        }
        // one k-tile.  Loads retire in order, so they are issued as  B(kt+1) -> registers,  then  A(kt+2) -> LDS stage
        // (kt+2) % 3:  at the top of tile kt only the 2*NPL direct loads of A(kt+1) may still be in flight -- A gets two
        // tiles of MFMAs to arrive (HBM / fabric latency), B (L2-resident weights) one.
#define DL_KTILE(kt_, CUR, NXT)                                                                                       \
        {                                                                                                             \
            const int kt = (kt_);                                                                                     \
            if (kt + 1 < nkt) { if (X3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
            /* tie the fragments to the wait: the compiler must not move an MFMA that reads them above it */          \
            _Pragma("unroll") for (int t = 0; t < TN; ++t)                                                             \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
            _Pragma("unroll") for (int p = 0; p < NPL; ++p) asm volatile("" : "+v"(CUR[t][ks][p]));                    \
            vxb_raw_barrier();                                                                                        \
            if (kt + 1 < nkt) { DL_LOADB(NXT, k0_next) }                                                              \
            if (kt + 2 < nkt) { issue((kt + 2) % 3, kt + 2); }                                                        \
            const u16* sb = smem + (kt % 3) * STAGE;                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                         \
                bf16x8 ah[2], al[2];                                                                                  \
                _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                        \
                    const int co = ((2 * ks + hi) ^ fx[t]) * 8;                                                       \
                    ah[t] = *reinterpret_cast<const bf16x8*>(sb + fa[t] + co);                                        \
                    if (X3) al[t] = *reinterpret_cast<const bf16x8*>(sb + DTILE + fa[t] + co);                        \
                }                                                                                                     \
                if (X3) {                                                                                             \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                      \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], CUR[j][ks][0], acc[i][j], 0, 0, 0); \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                      \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], CUR[j][ks][NPL - 1], acc[i][j], 0, 0, 0); \
                }                                                                                                     \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                          \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], CUR[j][ks][0], acc[i][j], 0, 0, 0);     \
            }                                                                                                         \
            k0_next = k0_issued;                                                                                      \
        }
        // k0_next: weight k offset of tile kt + 1 (the A issue runs one tile further ahead than the B loads)
        issue(0, 0);
        DL_LOADB(b0, k0_issued)
        int k0_next = 0;
        if (nkt > 1) { issue(1, 1); k0_next = k0_issued; }
        int kt2 = 0;
#pragma unroll 1
        for (; kt2 + 2 <= nkt; kt2 += 2) {
            DL_KTILE(kt2, b0, b1)
            DL_KTILE(kt2 + 1, b1, b0)
        }
        if (kt2 < nkt) DL_KTILE(kt2, b0, b1)
    } else {
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of k-tile kt have landed in LDS
        __syncthreads();                                      // ... everyone's have; and everyone finished k-tile kt-1
        if (kt + 1 < nkt) issue((kt + 1) & 1, kt + 1);
        const u16* sb = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int co = ((2 * ks + hi) ^ fx[t]) * 8;
                ah[t] = *reinterpret_cast<const bf16x8*>(sb + fa[t] + co);
                bh[t] = *reinterpret_cast<const bf16x8*>(sb + NPL * DTILE + fb[t] + co);
                if (X3) {
                    al[t] = *reinterpret_cast<const bf16x8*>(sb + DTILE + fa[t] + co);
                    bl[t] = *reinterpret_cast<const bf16x8*>(sb + (NPL + 1) * DTILE + fb[t] + co);
                }
            }
            if (X3) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }

    }

    float* __restrict__ C = g.C;
    const float* __restrict__ R = g.residual;
    if (g.d2s_s > 0) {
        // depth-to-space store: offset = row part (low-res voxel of row m) + column part (phase, channel of column n);
        // both are decoded once per row / per column tile, the row with multiply-high divisions (host-checked range)
        const int s = g.d2s_s, G = g.d2s_G, Cc = g.d2s_C;
        const long long Vv = (long long)G * s;
        long long coloff[TN];
        float bsv[TN];
        bool cok[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (32 * TN) + j * 32 + (lane & 31);
            cok[j] = n < g.N;
            const int nn = cok[j] ? n : 0;
            int ph = nn / Cc;
            const int co = nn - ph * Cc;
            if (g.d2s_perm) ph = g.d2s_perm[ph];
            const int rw = ph % s, rh = (ph / s) % s, rd = ph / (s * s);
            coloff[j] = (((long long)rd * Vv + rh) * Vv + rw) * Cc + co;
            bsv[j] = g.bias ? g.bias[nn] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                unsigned q1, q2, q3;
                if (g.d2s_magic) {
                    q1 = __umulhi((unsigned)m, g.d2s_magic); q2 = __umulhi(q1, g.d2s_magic); q3 = __umulhi(q2, g.d2s_magic);
                } else {
                    q1 = (unsigned)m / G; q2 = q1 / G; q3 = q2 / G;
                }
                const int qw = m - q1 * G, qh = q1 - q2 * G, qd = q2 - q3 * G;
                const long long rowoff = ((((long long)q3 * Vv + qd * s) * Vv + qh * s) * Vv + qw * s) * Cc;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!cok[j]) continue;
                    float v = acc[i][j][r] + bsv[j];
                    if (g.act == 1) v = v > 0.f ? v : v * g.slope;
                    const long long off = rowoff + coloff[j];
                    if (R) v += R[off];
                    if (g.accumulate) v += C[off];
                    C[off] = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (32 * TN) + j * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bsv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = acc[i][j][r] + bsv;
                if (g.act == 1) v = v > 0.f ? v : v * g.slope;
                const long long off = (long long)m * g.ldc + n;
                if (R) v += R[off];
                if (g.accumulate) v += C[off];
                C[off] = v;
            }
        }
    }
}

// fp32 [rows][cols] (row stride ld) -> bf16 planes: hi [rows][cols] and, when lo != nullptr, the residual plane
__global__ void __launch_bounds__(256) split_bf16_kernel(const float* __restrict__ src, long long ld, long long rows, int cols,
                                                         u16* __restrict__ hi, u16* __restrict__ lo) {
    const int q = cols >> 2;
    const long long total = rows * q;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / q;
        const int c4 = (int)(i - r * q) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + r * ld + c4);
        uint2 ph;
        ph.x = vxb_pack_bf16(v.x, v.y); ph.y = vxb_pack_bf16(v.z, v.w);
        *reinterpret_cast<uint2*>(hi + r * cols + c4) = ph;
        if (lo) {
            uint2 pl;
            pl.x = vxb_pack_bf16(v.x - __uint_as_float(ph.x << 16), v.y - __uint_as_float(ph.x & 0xffff0000u));
            pl.y = vxb_pack_bf16(v.z - __uint_as_float(ph.y << 16), v.w - __uint_as_float(ph.y & 0xffff0000u));
            *reinterpret_cast<uint2*>(lo + r * cols + c4) = pl;
        }
    }
}

// dst[i] = cvt(src[idx[i]]): a weight tensor re-laid out by a cached index table in ONE pass (the polyphase up-conv's 13.8 M-element
// effective weight went through five ATen permute / gather / transpose copies per step, 0.9 ms).  MODE 0: fp16 (RNE); MODE 1: bf16
// planes -- entry j < plane_off is the hi half of src[j], j >= plane_off the lo half bf16(v - hi) of src[j - plane_off] (the values
// vxb_split_bf16_f32 produces).  idx < 0: zero (padding rows of a fragment layout).  Eight outputs per thread, one 16-byte store.
template <int MODE>
__global__ void __launch_bounds__(256) gather_cvt_kernel(const float* __restrict__ src, const int* __restrict__ idx, long long n8,
                                                         u16* __restrict__ dst, int plane_off) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n8; t += (long long)gridDim.x * 256) {
        const int4 i0 = *reinterpret_cast<const int4*>(idx + t * 8), i1 = *reinterpret_cast<const int4*>(idx + t * 8 + 4);
        const int j[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int jj = j[e];
            float x = 0.f;
            if (jj >= 0) {
                if (MODE == 1 && jj >= plane_off) {
                    const float f = src[jj - plane_off];
                    x = f - __uint_as_float((vxb_pack_bf16(f, 0.f) & 0xffffu) << 16);
                } else {
                    x = src[jj];
                }
            }
            v[e] = x;
        }
        uint4 o;
        if (MODE == 0) { o.x = vxb_pack_f16(v[0], v[1]); o.y = vxb_pack_f16(v[2], v[3]); o.z = vxb_pack_f16(v[4], v[5]); o.w = vxb_pack_f16(v[6], v[7]); }
        else { o.x = vxb_pack_bf16(v[0], v[1]); o.y = vxb_pack_bf16(v[2], v[3]); o.z = vxb_pack_bf16(v[4], v[5]); o.w = vxb_pack_bf16(v[6], v[7]); }
        *reinterpret_cast<uint4*>(dst + t * 8) = o;
    }
}

// All linear-layer weights of a step in ONE launch: entry e of the device table desc[e] = {src, dst, rows, cols, flags,
// first tile} describes an fp32 [rows][cols] matrix whose planes go to dst as [nplanes][rows][cols] or, transposed (flags bit 0),
// as [nplanes][cols][rows] (the B operand of the data-gradient GEMM) -- the same hi / lo values vxb_split_bf16_f32 produces; with
// flags bit 1 the (possibly transposed) [n][k] matrix is written in MFMA fragment order [n / 32][k / 16][nplanes][64 lanes][8]
// instead (n % 32 == 0, k % 16 == 0): the weight operand of vxb_gemm_wide_bf16x3_f32 / vxb_gemm_dl_f32 without a shuffling copy;
// with bit 2 as well (n % 64 == 0) the rows are interleaved for vxb_gemm_wide_geglu_fwd_f32: row q < n / 2 goes to position
// (q / 32) * 64 + q % 32, row n / 2 + q to (q / 32) * 64 + 32 + q % 32; with bit 3 (and bit 1) the entry is ONE plane of fp16 values
// (RNE; [n / 32][k / 16][64 lanes][8]): the weight operand of vxb_gemm_wide_f16x2_f32.
// The per-weight launches (one split + one ATen transpose copy each, ~120 per step at ~11 us of latency apiece) cost more
// than moving the 33 M parameters.  A workgroup converts one 64 x 64 tile through LDS.
struct SplitDesc { const float* src; u16* dst; long long rows, cols, transposed, tile0; };

__global__ void __launch_bounds__(256) split_batch_kernel(const SplitDesc* __restrict__ desc, int n, int nplanes) {
    __shared__ float tile[64][65];
    int lo_i = 0, hi_i = n - 1;                                   // last entry whose first tile is <= blockIdx.x
    while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i + 1) >> 1;
        if (desc[mid].tile0 <= (long long)blockIdx.x) lo_i = mid; else hi_i = mid - 1;
    }
    const SplitDesc d = desc[lo_i];
    const bool tr = (d.transposed & 1) != 0, frag = (d.transposed & 2) != 0, glu = (d.transposed & 4) != 0;
    const bool h16 = (d.transposed & 8) != 0;     // ONE fp16 plane (fragment order): the weight operand of vxb_gemm_wide_f16x2_f32
    if (h16) nplanes = 1;
    const int tcols = (int)((d.cols + 63) >> 6);
    const int t = (int)(blockIdx.x - d.tile0);
    const int r0 = (t / tcols) * 64, c0 = (t % tcols) * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < d.rows && c0 + c < d.cols) ? d.src[(long long)(r0 + r) * d.cols + c0 + c] : 0.f;
    }
    __syncthreads();
    // output pairs: (orow, ocol..ocol+1); transposed: out[c][r]
    const long long orows = tr ? d.cols : d.rows, ocols = tr ? d.rows : d.cols;
    const int or0 = tr ? c0 : r0, oc0 = tr ? r0 : c0;
    u16* hi = d.dst;
    // plain: the lo plane follows the hi plane; fragment order: the planes of a (column tile, k-step) block are adjacent (512 each)
    u16* lo = nplanes == 2 ? d.dst + (frag ? 512 : d.rows * d.cols) : nullptr;
    const long long nks = ocols >> 4;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int orr = i >> 5, oc = (i & 31) * 2;
        if (or0 + orr >= orows || oc0 + oc >= ocols) continue;
        const float a = tr ? tile[oc][orr] : tile[orr][oc];
        const float b = tr ? tile[oc + 1][orr] : tile[orr][oc + 1];
        const unsigned ph = h16 ? vxb_pack_f16(a, b) : vxb_pack_bf16(a, b);
        long long n_ = or0 + orr;
        const long long k_ = oc0 + oc;
        if (glu) {           // GEGLU up-projection: every 64-row block of the fragment order = [32 value rows | their 32 gate rows]
            const long long F = orows >> 1, q = n_ < F ? n_ : n_ - F;
            n_ = (q >> 5) * 64 + (n_ < F ? 0 : 32) + (q & 31);
        }
        // MFMA fragment order [n / 32][k / 16][plane][half = (k % 16) / 8][n % 32][k % 8] (ops.gemm_wfrag) or row-major [n][k]
        const long long o = frag ? (((n_ >> 5) * nks + (k_ >> 4)) * nplanes) * 512 + ((k_ >> 3) & 1) * 256 + (n_ & 31) * 8 + (k_ & 7)
                                 : n_ * ocols + k_;
        *reinterpret_cast<unsigned*>(hi + o) = ph;
        if (lo) *reinterpret_cast<unsigned*>(lo + o) = vxb_pack_bf16(a - __uint_as_float(ph << 16), b - __uint_as_float(ph & 0xffff0000u));
    }
}

template <int AMODE, int X3, int BD, int TN = 2>
int dl_launch2(const DlArgs& g, hipStream_t st) {
    const dim3 grid(vxb_cdiv(g.N, 64 * TN), vxb_cdiv(g.M, 128));
    const size_t lds = (size_t)(BD ? 3 : 2 * 2) * (1 + X3) * DTILE * sizeof(u16);     // BD: three A stages; else 2 x (A + B)
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)gemm_dl_kernel<AMODE, X3, BD, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
    hipLaunchKernelGGL((gemm_dl_kernel<AMODE, X3, BD, TN>), grid, dim3(256), lds, st, g);
    return hipGetLastError() == hipSuccess ? VXB_OK : VXB_ELAUNCH;
}

template <int AMODE>
int dl_launch(const DlArgs& g, int x3, hipStream_t st) {
    // N <= 64 with weight fragments, no depth-to-space store, no tap mask: 64-column tiles (no MFMAs on zero-padded columns)
    if (g.Bfrag && x3 && g.N <= 64 && g.d2s_s <= 0 && !g.tapmask) return dl_launch2<AMODE, 1, 1, 1>(g, st);
    if (g.Bfrag) return x3 ? dl_launch2<AMODE, 1, 1>(g, st) : dl_launch2<AMODE, 0, 1>(g, st);
    return x3 ? dl_launch2<AMODE, 1, 0>(g, st) : dl_launch2<AMODE, 0, 0>(g, st);
}

inline bool dl_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// dst planes [nplanes (1 or 2)][rows][cols] bf16 <- src fp32 [rows][cols] (row stride ld): plane 0 = bf16(src) (RNE),
// plane 1 = bf16(src - plane 0): the operand format of the direct-to-LDS kernels.  cols % 4 == 0.
// dst [n] (fp16: mode 0, bf16: mode 1) = converted src[idx[i]] (see gather_cvt_kernel); n % 8 == 0, idx and dst 16-byte aligned.
extern "C" int vxb_gather_cvt_f32(const float* src, const int32_t* idx, int64_t n, void* dst, int mode, int32_t plane_off,
                                  vxb_stream_t stream) {
    if (!src || !idx || !dst || n < 8 || (mode != 0 && mode != 1)) return VXB_EARG;
    if ((n & 7) || (((uintptr_t)idx | (uintptr_t)dst) & 15)) return VXB_ESIZE;
    const long long n8 = n >> 3;
    const int grid = (int)(n8 + 255) / 256 > 8192 ? 8192 : (int)((n8 + 255) / 256);
    if (mode == 0) hipLaunchKernelGGL(gather_cvt_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, idx, n8, (u16*)dst, plane_off);
    else hipLaunchKernelGGL(gather_cvt_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, idx, n8, (u16*)dst, plane_off);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_split_bf16_f32(const float* src, int64_t ld, int64_t rows, int cols, void* dst_planes, int nplanes,
                                  vxb_stream_t stream) {
    if (!src || !dst_planes || rows < 1 || cols < 4 || (nplanes != 1 && nplanes != 2)) return VXB_EARG;
    if ((cols & 3) || (ld & 3) || !dl_al16(src) || !dl_al16(dst_planes)) return VXB_ESIZE;
    u16* hi = (u16*)dst_planes;
    u16* lo = nplanes == 2 ? hi + rows * cols : nullptr;
    const long long total = rows * (cols >> 2);
    const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (long long)ld, (long long)rows, cols, hi, lo);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// desc: device table of n entries x 6 int64 {src pointer, dst pointer, rows, cols, flags (bit 0: transposed, bit 1: fragment order),
// first tile}, entries in ascending first-tile order with tiles = ceil(rows/64) * ceil(cols/64); total_tiles = their sum.
// rows, cols even; fragment order: output rows % 32 == 0, output columns % 16 == 0.
extern "C" int vxb_split_bf16_batch_f32(const int64_t* desc, int n, int64_t total_tiles, int nplanes, vxb_stream_t stream) {
    if (!desc || n < 1 || total_tiles < 1 || total_tiles >= INT32_MAX || (nplanes != 1 && nplanes != 2)) return VXB_EARG;
    static_assert(sizeof(SplitDesc) == 6 * sizeof(int64_t), "descriptor layout");
    hipLaunchKernelGGL(split_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const SplitDesc*>(desc), n, nplanes);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// C[M,N] (+)= act(A @ Bw^T + bias) (+ residual) with BOTH operands as bf16 planes: A_planes [nplanes][M][K] (from
// vxb_split_bf16_f32), Bw_planes [nplanes][N][K]; nplanes = 1 ('bf16') or 2 ('bf16x3': hi*hi + hi*lo + lo*hi).  K % 32 == 0.
// Bw_frag (optional): the same weights, rows zero-padded to a multiple of 128, in fragment order
// [ceil(N/128)*4][K/16][nplanes][64 lanes][8 bf16] -- lane
// (col = lane & 31, half = lane >> 5) of column tile t, k-step ks holds Bw[32 t + col][16 ks + 8 half .. + 7]; the kernel
// then reads its B fragments straight from global memory (no weight traffic through LDS).
extern "C" int vxb_gemm_dl_f32(const void* A_planes, const void* Bw_planes, const void* Bw_frag, int nplanes, float* C, int64_t ldc,
                               const float* bias, const float* residual, int M, int N, int K, int act, float slope,
                               int accumulate, vxb_stream_t stream) {
    if (!A_planes || !Bw_planes || !C || M < 1 || N < 1 || K < 32 || (nplanes != 1 && nplanes != 2)) return VXB_EARG;
    if ((K & 31) || !dl_al16(A_planes) || !dl_al16(Bw_planes)) return VXB_ESIZE;
    if (Bw_frag && !dl_al16(Bw_frag)) return VXB_ESIZE;
    DlArgs g = {};
    g.Bfrag = (const u16*)Bw_frag;
    g.A = (const u16*)A_planes; g.a_plane = (long long)M * K; g.lda = K;
    g.Bw = (const u16*)Bw_planes; g.b_plane = (long long)N * K; g.zeros = (const u16*)Bw_planes;
    g.C = C; g.bias = bias; g.residual = residual; g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.act = act; g.slope = slope;
    g.accumulate = accumulate;
    return dl_launch<DL_A_KCONTIG>(g, nplanes == 2, (hipStream_t)stream);
}

// Implicit-GEMM conv3d twin of vxb_conv3d_bf16w/bf16x3_f32 for ONE source whose activations were pre-split:
// src_planes [nplanes][B, S_in^3, Cin] bf16; weights [nplanes][N][K = kext^3 * Cin]; Cin % 32 == 0.
// zeros: >= 16 bytes of zeros in device memory (fetched for zero-padded taps).
// wt_frag (optional): the weights in fragment order, see vxb_gemm_dl_f32.
// tapmask (optional, kext^3 <= 32): one word per 128-column tile, bit t set <=> tap t has non-zero weights in that tile
// (never all-zero); the other taps are skipped.  d2s_perm (optional, d2s_C == 64): column block p of the weights / output
// is phase d2s_perm[p] of the fine grid -- lets the host pair phases with the same tap footprint in one column tile.
extern "C" int vxb_conv3d_dl_f32(const void* src_planes, int Cin, int B, int S_in, int S_out, int stride, int kext, int off,
                                 int replicate, const void* wt_planes, const void* wt_frag, int nplanes, int N, const float* bias, float* out,
                                 int64_t ldc, int act, float slope, int accumulate, int d2s_s, int d2s_C, const void* zeros,
                                 const uint32_t* tapmask, const int32_t* d2s_perm, vxb_stream_t stream) {
    if (!src_planes || !wt_planes || !out || !zeros || B < 1 || S_in < 1 || S_out < 1 || kext < 1 || stride < 1 || N < 1) return VXB_EARG;
    if ((nplanes != 1 && nplanes != 2) || (Cin & 31) || Cin < 32) return VXB_ESIZE;
    if (!dl_al16(src_planes) || !dl_al16(wt_planes) || !dl_al16(zeros)) return VXB_ESIZE;
    const long long M = (long long)B * S_out * S_out * S_out;
    const long long K = (long long)kext * kext * kext * Cin;
    if (M >= INT32_MAX || K >= INT32_MAX) return VXB_ESIZE;
    if (d2s_s > 0 && (d2s_C < 1 || N % d2s_C)) return VXB_EARG;
    if (tapmask && kext * kext * kext > 32) return VXB_ESIZE;
    if (d2s_perm && (d2s_s <= 0 || d2s_C != 64)) return VXB_EARG;      // a permuted phase must not straddle column tiles
    if (wt_frag && !dl_al16(wt_frag)) return VXB_ESIZE;
    DlArgs g = {};
    g.Bfrag = (const u16*)wt_frag;
    g.tapmask = tapmask; g.d2s_perm = d2s_perm;
    g.A = (const u16*)src_planes; g.a_plane = (long long)B * S_in * S_in * S_in * Cin;
    g.Bw = (const u16*)wt_planes; g.b_plane = (long long)N * K; g.zeros = (const u16*)zeros;
    g.C = out; g.bias = bias; g.M = (int)M; g.N = N; g.K = (int)K; g.ldc = ldc; g.act = act; g.slope = slope;
    g.accumulate = accumulate; g.d2s_s = d2s_s; g.d2s_G = S_out; g.d2s_C = d2s_C;
    g.d2s_magic = (d2s_s > 0 && S_out > 1 && M * S_out < (1ll << 32)) ? (unsigned)((1ull << 32) / (unsigned)S_out) + 1u : 0u;
    g.Cin = Cin; g.S_in = S_in; g.S_out = S_out; g.stride = stride; g.kext = kext; g.off = off; g.replicate = replicate;
    return dl_launch<DL_A_CONV>(g, nplanes == 2, (hipStream_t)stream);
}
