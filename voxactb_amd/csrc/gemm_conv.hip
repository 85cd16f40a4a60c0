// fp32 GEMM and implicit-GEMM conv3d on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bitwise an fmaf chain, 157 TF peak on MI355X).  This is the parity ("fp32 mode") compute path of the
// PerceiverIO Q-function: Linear layers, attention GEMMs, and every Conv3DBlock of
// peract/helpers/network_utils.py:128-170 / :237-254 as used by perceiver_lang_io.py:345-485.
//
//   C[M,N] (+)= act(alpha * A[M,K] @ B[K,N] + bias[N]) (+ residual)
//
// A operand loaders:  K-contiguous rows, M-contiguous columns (transposed operand), or the conv gather
//                     (channels-last source(s), replicate / zero padding, stride, two concatenated sources).
// B operand loaders:  N-contiguous ([K][N] weights) or K-contiguous (torch Linear weight [N][K]).
// Output mappings:    row-major with ldc, or depth-to-space (polyphase up-conv, see conv_plan.py).
//
// Tile: BM x BN x 16, 256 threads = 4 waves, each wave owns a (BM/WM) x (BN/WN) sub-tile of 32x32 MFMA
// accumulators.  LDS tiles are k-major ([k][m], [k][n]) so that an MFMA operand read is one conflict-free
// ds_read_b32 per lane (lane l: A[m = l&31][k = l>>5]).  Global loads of tile t+1 are issued into
// registers before the MFMAs of tile t (register double buffering), one barrier pair per tile.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

enum { A_KCONTIG = 0, A_MCONTIG = 1, A_CONV = 2, A_CONVT = 3 };
enum { B_NCONTIG = 0, B_KCONTIG = 1, B_D2S = 2 };
enum { ACT_NONE = 0, ACT_LRELU = 1 };

struct ConvGeom {
    const float* src0;
    const float* src1;
    int C0, C1;          // channels of source 0 / 1 (C1 == 0: single source)
    int S_in;            // source cube side
    int S_out;           // row-grid cube side (rows m = ((b*S_out + d)*S_out + h)*S_out + w)
    int stride;          // src = o*stride + tap + off
    int kext;            // taps per axis
    int off;             // usually -pad
    int replicate;       // 1: clamp to [0, S_in-1]; 0: zero outside
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;       // [N] or null
    const float* residual;   // same mapping as C, or null
    int M, N, K;
    long long sAm, sAk, sBk, sBn;      // element strides
    long long ldc;
    long long bA1, bA2, bB1, bB2, bC1, bC2;   // batch strides: z -> (z / H, z % H)
    int H;
    float alpha;
    int act;                 // ACT_*
    float slope;
    int accumulate;          // C += result
    int d2s_s, d2s_G, d2s_C; // depth-to-space output (d2s_s > 0): n -> (phase, co), m -> (b, q)
    int tiles_per_split;     // > 0: blockIdx.z is a split of the reduction dimension (wgrad), C += z * bC1
    ConvGeom cg;
};

template <int AMODE, int BMODE, int BM, int BN, int WM, int WN, int BKT>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int KQ = BKT / 4;                  // float4 per tile row
    constexpr int RPP = 256 / KQ;                // rows covered per pass of the 256 threads
    constexpr int A_F4 = BM * BKT / 4 / 256;      // float4 loads per thread for the A tile
    constexpr int B_F4 = BN * BKT / 4 / 256;
    __shared__ __attribute__((aligned(16))) float As[BKT * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BKT * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int z = g.tiles_per_split > 0 ? 0 : blockIdx.z;
    const long long zo1 = z / g.H, zo2 = z % g.H;
    const float* __restrict__ A = g.A + zo1 * g.bA1 + zo2 * g.bA2;
    const float* __restrict__ Bp = g.B + zo1 * g.bB1 + zo2 * g.bB2;
    // XCD-aware tile order: the dispatcher places linear block id b on XCD b % 8 (each XCD has a private 4 MB L2), so
    // give every XCD a CONTIGUOUS range of tiles -- neighbouring conv tiles share their tap windows through that L2.
    // Bijective for any grid size; a different hardware placement would only change speed, never results.
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
    const int m0 = tile_y * BM, n0 = tile_x * BN;

    // ---------------------------------------------------------------- per-thread load descriptors
    // A, K-contiguous / conv: thread -> (row = tid/4 + 64*i, kq = tid%4); M-contiguous: (k = tid/(BM/4)+..., mq)
    int a_b[A_F4], a_d[A_F4], a_h[A_F4], a_w[A_F4];   // conv row coordinates
    bool a_rowok[A_F4];
    if (AMODE == A_CONV) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int m = m0 + (tid / KQ) + RPP * i;
            a_rowok[i] = m < g.M;
            const int S = g.cg.S_out;
            int r = a_rowok[i] ? m : 0;
            a_w[i] = r % S; r /= S;
            a_h[i] = r % S; r /= S;
            a_d[i] = r % S; r /= S;
            a_b[i] = r;
        }
    }
    float4 ra[A_F4], rb[B_F4];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BKT;
        // ---- A
        if (AMODE == A_KCONTIG) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int m = m0 + (tid / KQ) + RPP * i;
                const int k = k0 + (tid % KQ) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < g.M && k < g.K) v = *reinterpret_cast<const float4*>(A + (long long)m * g.sAm + k);
                ra[i] = v;
            }
        } else if (AMODE == A_MCONTIG) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int e = tid + 256 * i;
                const int k = k0 + e / (BM / 4);
                const int m = m0 + (e % (BM / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < g.K) {
                    const float* p = A + (long long)k * g.sAk + m;
                    if (m + 3 < g.M) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (m < g.M) v.x = p[0];
                        if (m + 1 < g.M) v.y = p[1];
                        if (m + 2 < g.M) v.z = p[2];
                    }
                }
                ra[i] = v;
            }
        } else if (AMODE == A_CONVT) {
            // transposed gather for the weight gradient: rows = (tap, channel), reduction = positions
            const ConvGeom& c = g.cg;
            const int Ct = c.C0 + c.C1;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int e = tid + 256 * i;
                const int pos = k0 + e / (BM / 4);
                const int kr = m0 + (e % (BM / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kr < g.M && pos < g.K) {
                    const int tap = kr / Ct;
                    const int cc = kr - tap * Ct;
                    const int tw = tap % c.kext, th = (tap / c.kext) % c.kext, td = tap / (c.kext * c.kext);
                    const bool second = cc >= c.C0;
                    const float* src = second ? c.src1 : c.src0;
                    const int Cs = second ? c.C1 : c.C0;
                    const int ch = second ? cc - c.C0 : cc;
                    const int S = c.S_out;
                    int r = pos;
                    const int ow = r % S; r /= S;
                    const int oh = r % S; r /= S;
                    const int od = r % S; r /= S;
                    int id = od * c.stride + td + c.off;
                    int ih = oh * c.stride + th + c.off;
                    int iw = ow * c.stride + tw + c.off;
                    bool ok = true;
                    if (c.replicate) {
                        id = min(max(id, 0), c.S_in - 1);
                        ih = min(max(ih, 0), c.S_in - 1);
                        iw = min(max(iw, 0), c.S_in - 1);
                    } else {
                        ok = id >= 0 && id < c.S_in && ih >= 0 && ih < c.S_in && iw >= 0 && iw < c.S_in;
                    }
                    if (ok) {
                        const long long vox = (((long long)r * c.S_in + id) * c.S_in + ih) * c.S_in + iw;
                        v = *reinterpret_cast<const float4*>(src + vox * Cs + ch);
                    }
                }
                ra[i] = v;
            }
        } else {
            const ConvGeom& c = g.cg;
            const int Ct = c.C0 + c.C1;
            const int tap = k0 / Ct;
            const int cc = k0 - tap * Ct + (tid % KQ) * 4;     // channel within the concatenated sources
            const int tw = tap % c.kext, th = (tap / c.kext) % c.kext, td = tap / (c.kext * c.kext);
            const bool second = cc >= c.C0;
            const float* src = second ? c.src1 : c.src0;
            const int Cs = second ? c.C1 : c.C0;
            const int ch = second ? cc - c.C0 : cc;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                int id = a_d[i] * c.stride + td + c.off;
                int ih = a_h[i] * c.stride + th + c.off;
                int iw = a_w[i] * c.stride + tw + c.off;
                bool ok = a_rowok[i] && (k0 < g.K);
                if (c.replicate) {
                    id = min(max(id, 0), c.S_in - 1);
                    ih = min(max(ih, 0), c.S_in - 1);
                    iw = min(max(iw, 0), c.S_in - 1);
                } else {
                    ok = ok && id >= 0 && id < c.S_in && ih >= 0 && ih < c.S_in && iw >= 0 && iw < c.S_in;
                }
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const long long vox = (((long long)a_b[i] * c.S_in + id) * c.S_in + ih) * c.S_in + iw;
                    v = *reinterpret_cast<const float4*>(src + vox * Cs + ch);
                }
                ra[i] = v;
            }
        }
        // ---- B
        if (BMODE == B_NCONTIG) {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + 256 * i;
                const int k = k0 + e / (BN / 4);
                const int n = n0 + (e % (BN / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < g.K) {
                    const float* p = Bp + (long long)k * g.sBk + n;
                    if (n + 3 < g.N) v = *reinterpret_cast<const float4*>(p);
                    else {
                        if (n < g.N) v.x = p[0];
                        if (n + 1 < g.N) v.y = p[1];
                        if (n + 2 < g.N) v.z = p[2];
                    }
                }
                rb[i] = v;
            }
        } else if (BMODE == B_D2S) {
            // B(pos, n) read from a fine grid: n -> (phase, co), pos -> (b, q)   (gradient of a depth-to-space output)
            const int s = g.d2s_s < 0 ? -g.d2s_s : g.d2s_s, G = g.d2s_G, Cc = g.d2s_C;
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + 256 * i;
                const int k = k0 + e / (BN / 4);
                const int n = n0 + (e % (BN / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < g.K && n < g.N) {
                    const int ph = n / Cc, co = n - ph * Cc;
                    const int rw = ph % s, rh = (ph / s) % s, rd = ph / (s * s);
                    int q = k;
                    const int qw = q % G; q /= G;
                    const int qh = q % G; q /= G;
                    const int qd = q % G; q /= G;
                    const long long Vv = (long long)G * s;
                    const long long off = ((((long long)q * Vv + qd * s + rd) * Vv + qh * s + rh) * Vv + qw * s + rw) * Cc + co;
                    v = *reinterpret_cast<const float4*>(Bp + off);
                }
                rb[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int n = n0 + (tid / KQ) + RPP * i;
                const int k = k0 + (tid % KQ) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < g.N && k < g.K) v = *reinterpret_cast<const float4*>(Bp + (long long)n * g.sBn + k);
                rb[i] = v;
            }
        }
    };

    auto store_tile = [&]() {
        if (AMODE == A_MCONTIG || AMODE == A_CONVT) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int e = tid + 256 * i;
                *reinterpret_cast<float4*>(&As[(e / (BM / 4)) * LDA + (e % (BM / 4)) * 4]) = ra[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int r = (tid / KQ) + RPP * i, kq = (tid % KQ) * 4;
                As[(kq + 0) * LDA + r] = ra[i].x;
                As[(kq + 1) * LDA + r] = ra[i].y;
                As[(kq + 2) * LDA + r] = ra[i].z;
                As[(kq + 3) * LDA + r] = ra[i].w;
            }
        }
        if (BMODE == B_NCONTIG || BMODE == B_D2S) {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + 256 * i;
                *reinterpret_cast<float4*>(&Bs[(e / (BN / 4)) * LDB + (e % (BN / 4)) * 4]) = rb[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int r = (tid / KQ) + RPP * i, kq = (tid % KQ) * 4;
                Bs[(kq + 0) * LDB + r] = rb[i].x;
                Bs[(kq + 1) * LDB + r] = rb[i].y;
                Bs[(kq + 2) * LDB + r] = rb[i].z;
                Bs[(kq + 3) * LDB + r] = rb[i].w;
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

    int nkt = (g.K + BKT - 1) / BKT;
    int kt_begin = 0;
    if (g.tiles_per_split > 0) {
        kt_begin = blockIdx.z * g.tiles_per_split;
        nkt = min(nkt, kt_begin + g.tiles_per_split);
    }
    if (kt_begin < nkt) load_tile(kt_begin);
    for (int kt = kt_begin; kt < nkt; ++kt) {
        __syncthreads();               // previous tile's LDS reads are done
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);
        const int lk = lane >> 5, lm = lane & 31;
        // issue every LDS fragment read of the tile first: the matrix pipe then never waits for an LDS round trip
        // (a read-then-multiply loop left ~100 cycles of LDS latency exposed per 256-cycle MFMA group)
        float av[BKT / 2][TM], bv[BKT / 2][TN];
#pragma unroll
        for (int kk = 0; kk < BKT; kk += 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[kk / 2][i] = As[(kk + lk) * LDA + wm * (BM / WM) + i * 32 + lm];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[kk / 2][j] = Bs[(kk + lk) * LDB + wn * (BN / WN) + j * 32 + lm];
        }
#pragma unroll
        for (int kk = 0; kk < BKT; kk += 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk / 2][i], bv[kk / 2][j], acc[i][j], 0, 0, 0);
        }
    }

    // ---------------------------------------------------------------- epilogue
    float* __restrict__ C = g.C + zo1 * g.bC1 + zo2 * g.bC2 + (g.tiles_per_split > 0 ? (long long)blockIdx.z * g.bC1 : 0);
    const float* __restrict__ R = g.residual ? g.residual + zo1 * g.bC1 + zo2 * g.bC2 : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bsv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = g.alpha * acc[i][j][r] + bsv;
                if (g.act == ACT_LRELU) v = v > 0.f ? v : v * g.slope;
                long long off;
                if (g.d2s_s > 0) {
                    const int s = g.d2s_s, G = g.d2s_G, Cc = g.d2s_C;
                    const int ph = n / Cc, co = n - ph * Cc;
                    const int rw = ph % s, rh = (ph / s) % s, rd = ph / (s * s);
                    int q = m;
                    const int qw = q % G; q /= G;
                    const int qh = q % G; q /= G;
                    const int qd = q % G; q /= G;
                    const long long Vv = (long long)G * s;
                    off = ((((long long)q * Vv + qd * s + rd) * Vv + qh * s + rh) * Vv + qw * s + rw) * Cc + co;
                } else {
                    off = (long long)m * g.ldc + n;
                }
                if (R) v += R[off];
                if (g.accumulate) v += C[off];
                C[off] = v;
            }
        }
    }
}

template <int AMODE, int BMODE, int BKT>
int launch_gemm_bk(const GemmArgs& g, int batch, hipStream_t st) {
    if (g.N > 64) {
        dim3 grid(vxb_cdiv(g.N, 128), vxb_cdiv(g.M, 128), batch);
        hipLaunchKernelGGL((gemm_kernel<AMODE, BMODE, 128, 128, 2, 2, BKT>), grid, dim3(256), 0, st, g);
    } else {
        dim3 grid(vxb_cdiv(g.N, 64), vxb_cdiv(g.M, 128), batch);
        hipLaunchKernelGGL((gemm_kernel<AMODE, BMODE, 128, 64, 2, 2, BKT>), grid, dim3(256), 0, st, g);
    }
    if (hipGetLastError() != hipSuccess) return VXB_ELAUNCH;
    return VXB_OK;
}

// K tile of 32 (fewer barriers per FLOP) whenever a tile cannot straddle a tap / source boundary, else 16
template <int AMODE, int BMODE>
int launch_gemm(const GemmArgs& g, int batch, hipStream_t st) {
    const bool conv = AMODE == A_CONV || AMODE == A_CONVT;
    const bool wide = conv ? ((g.cg.C0 & 31) == 0 && (g.cg.C1 & 31) == 0) : true;
    if (AMODE == A_CONVT) return launch_gemm_bk<AMODE, BMODE, 16>(g, batch, st);   // reduction runs over positions: keep 16
    return wide ? launch_gemm_bk<AMODE, BMODE, 32>(g, batch, st) : launch_gemm_bk<AMODE, BMODE, 16>(g, batch, st);
}

template <int AMODE>
int dispatch_b(const GemmArgs& g, int batch, hipStream_t st) {
    if (g.sBn == 1) return launch_gemm<AMODE, B_NCONTIG>(g, batch, st);
    if (g.sBk == 1) return launch_gemm<AMODE, B_KCONTIG>(g, batch, st);
    return VXB_EARG;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// C[z] (+)= act(alpha * A[z] @ B[z] + bias) (+ residual[z]);  z in [0, batch), offsets z/H, z%H (see header).
extern "C" int vxb_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                            int M, int N, int K, int64_t sAm, int64_t sAk, int64_t sBk, int64_t sBn, int64_t ldc,
                            int batch, int H, int64_t bA1, int64_t bA2, int64_t bB1, int64_t bB2, int64_t bC1,
                            int64_t bC2, float alpha, int act, float slope, int accumulate, vxb_stream_t stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || H < 1) return VXB_EARG;
    if (!aligned16(A) || !aligned16(B)) return VXB_EARG;
    GemmArgs g = {};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.residual = residual;
    g.M = M; g.N = N; g.K = K; g.sAm = sAm; g.sAk = sAk; g.sBk = sBk; g.sBn = sBn; g.ldc = ldc;
    g.bA1 = bA1; g.bA2 = bA2; g.bB1 = bB1; g.bB2 = bB2; g.bC1 = bC1; g.bC2 = bC2; g.H = H;
    g.alpha = alpha; g.act = act; g.slope = slope; g.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    // float4 loads along the contiguous dimension need 4-element alignment of every row start
    // K need not be a multiple of 4, but then rows must be padded to one (stride >= roundup4(K)) with ZEROS:
    // the last 16-byte load of a row reads the padding.
    if (sAk == 1) {
        if ((sAm & 3) || (bA1 & 3) || (bA2 & 3) || sAm < ((K + 3) & ~3)) return VXB_ESIZE;
    } else if (sAm == 1) {
        if ((sAk & 3) || (bA1 & 3) || (bA2 & 3)) return VXB_ESIZE;
    } else return VXB_EARG;
    if (sBn == 1) {
        if ((sBk & 3) || (bB1 & 3) || (bB2 & 3)) return VXB_ESIZE;
    } else if (sBk == 1) {
        if ((sBn & 3) || (bB1 & 3) || (bB2 & 3) || sBn < ((K + 3) & ~3)) return VXB_ESIZE;
    } else return VXB_EARG;
    if (sAk == 1) return dispatch_b<A_KCONTIG>(g, batch, st);
    return dispatch_b<A_MCONTIG>(g, batch, st);
}

// Implicit-GEMM conv3d over channels-last cubes (forward, and -- with flipped weights and zero padding --
// the data gradient).  rows m = (b, d, h, w) over S_out^3, K = kext^3 * (C0 + C1), weights wt[K][N].
extern "C" int vxb_conv3d_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                              int stride, int kext, int off, int replicate, const float* wt, int N,
                              const float* bias, float* out, int64_t ldc, int act, float slope, int accumulate,
                              int d2s_s, int d2s_C, vxb_stream_t stream) {
    if (!src0 || !wt || !out || B < 1 || S_in < 1 || S_out < 1 || kext < 1 || stride < 1 || N < 1) return VXB_EARG;
    if ((C0 & 15) || (C1 & 15) || C0 < 16 || (C1 > 0 && !src1)) return VXB_ESIZE;
    if ((N & 3) || !aligned16(src0) || !aligned16(wt) || (src1 && !aligned16(src1))) return VXB_ESIZE;
    const long long M = (long long)B * S_out * S_out * S_out;
    const long long K = (long long)kext * kext * kext * (C0 + C1);
    if (M >= INT32_MAX || K >= INT32_MAX) return VXB_ESIZE;
    GemmArgs g = {};
    g.A = src0; g.B = wt; g.C = out; g.bias = bias; g.residual = nullptr;
    g.M = (int)M; g.N = N; g.K = (int)K; g.sBk = N; g.sBn = 1; g.ldc = ldc; g.H = 1;
    g.alpha = 1.f; g.act = act; g.slope = slope; g.accumulate = accumulate;
    g.d2s_s = d2s_s; g.d2s_G = S_out; g.d2s_C = d2s_C;
    g.cg.src0 = src0; g.cg.src1 = src1; g.cg.C0 = C0; g.cg.C1 = C1; g.cg.S_in = S_in; g.cg.S_out = S_out;
    g.cg.stride = stride; g.cg.kext = kext; g.cg.off = off; g.cg.replicate = replicate;
    if (d2s_s > 0 && (d2s_C < 1 || N % d2s_C)) return VXB_EARG;
    return launch_gemm<A_CONV, B_NCONTIG>(g, 1, (hipStream_t)stream);
}

// Weight gradient of the same conv: part[z][K][N] = sum over the z-th slice of positions of gather(src)^T @ dY.
// dY is row-major [M][N] (ldy) or, with d2s_s > 0, a fine grid [B,(S_out*s)^3,d2s_C] (depth-to-space adjoint).
// `part` must hold nsplit*K*N floats; vxb_sum_splits_f32 reduces it deterministically.
extern "C" int vxb_conv3d_wgrad_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                    int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                                    int d2s_s, int d2s_C, float* part, int nsplit, vxb_stream_t stream) {
    if (!src0 || !dy || !part || B < 1 || S_in < 1 || S_out < 1 || kext < 1 || stride < 1 || N < 1 || nsplit < 1) return VXB_EARG;
    if ((C0 & 15) || (C1 & 15) || C0 < 16 || (C1 > 0 && !src1)) return VXB_ESIZE;
    if ((N & 3) || !aligned16(src0) || !aligned16(dy) || (src1 && !aligned16(src1))) return VXB_ESIZE;
    const long long P = (long long)B * S_out * S_out * S_out;          // reduction length
    const long long K = (long long)kext * kext * kext * (C0 + C1);    // output rows
    if (P >= INT32_MAX || K >= INT32_MAX) return VXB_ESIZE;
    GemmArgs g = {};
    g.A = src0; g.B = dy; g.C = part; g.M = (int)K; g.N = N; g.K = (int)P;
    g.sBk = ldy; g.sBn = 1; g.ldc = N; g.H = 1; g.alpha = 1.f; g.act = ACT_NONE;
    g.bC1 = K * N;
    const int nkt = (int)((P + BK - 1) / BK);
    g.tiles_per_split = (nkt + nsplit - 1) / nsplit;
    g.d2s_s = 0; g.d2s_G = S_out; g.d2s_C = d2s_C;
    g.cg.src0 = src0; g.cg.src1 = src1; g.cg.C0 = C0; g.cg.C1 = C1; g.cg.S_in = S_in; g.cg.S_out = S_out;
    g.cg.stride = stride; g.cg.kext = kext; g.cg.off = off; g.cg.replicate = replicate;
    hipStream_t st = (hipStream_t)stream;
    if (d2s_s > 0) {
        if (d2s_C < 4 || (d2s_C & 3) || N % d2s_C) return VXB_EARG;
        GemmArgs gg = g;
        // B loader uses the d2s fields; the epilogue must stay row-major, so pass the geometry via a copy
        gg.d2s_s = -d2s_s;   // negative: "d2s for the B operand only"
        if (N > 64) {
            dim3 grid(vxb_cdiv(N, 128), vxb_cdiv(K, 128), nsplit);
            hipLaunchKernelGGL((gemm_kernel<A_CONVT, B_D2S, 128, 128, 2, 2, BK>), grid, dim3(256), 0, st, gg);
        } else {
            dim3 grid(vxb_cdiv(N, 64), vxb_cdiv(K, 128), nsplit);
            hipLaunchKernelGGL((gemm_kernel<A_CONVT, B_D2S, 128, 64, 2, 2, BK>), grid, dim3(256), 0, st, gg);
        }
    } else {
        if (ldy & 3) return VXB_ESIZE;
        if (N > 64) {
            dim3 grid(vxb_cdiv(N, 128), vxb_cdiv(K, 128), nsplit);
            hipLaunchKernelGGL((gemm_kernel<A_CONVT, B_NCONTIG, 128, 128, 2, 2, BK>), grid, dim3(256), 0, st, g);
        } else {
            dim3 grid(vxb_cdiv(N, 64), vxb_cdiv(K, 128), nsplit);
            hipLaunchKernelGGL((gemm_kernel<A_CONVT, B_NCONTIG, 128, 64, 2, 2, BK>), grid, dim3(256), 0, st, g);
        }
    }
    if (hipGetLastError() != hipSuccess) return VXB_ELAUNCH;
    return VXB_OK;
}

// =====================================================================================================================
// bf16 matrix-core variant ("throughput mode"): C = act(A @ Bw^T + bias).
//   A  : fp32 in HBM (activations / gradients), K-contiguous rows or the conv gather; rounded to bf16 (RNE) while being
//        staged into LDS, so the surrounding fp32 kernels and buffers are unchanged.
//   Bw : bf16 weights [N][K] (K contiguous), prepared once per step.
//   acc: fp32 (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate).
// LDS tiles are [row][32 k] bf16 with an 80-byte row stride: a fragment read is one ds_read_b128 per lane and the 16
// lanes of a read group hit 16 distinct 16-byte bank slots.  Both operands use the same (lane>>5, j) -> k placement, so
// the k-permutation inside one instruction cancels in the dot product.
// =====================================================================================================================
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    // round-to-nearest-even fp32 -> bf16, two values per dword
    return vxb_pack_bf16(lo, hi);
}

constexpr int BK16 = 32;
constexpr int LDS16 = 40;      // bf16 elements per LDS row (32 + 8 pad) = 80 bytes

// X3 = 1: "bf16x3" split products -- every fp32 A value is staged as hi = bf16(a) and lo = bf16(a - hi), the weights come
// pre-split the same way (planes [2][N][K]), and each product is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation:
// the dropped terms are <= 2^-16 relative per product (vs 2^-24 for fp32), at 3 bf16 MFMAs = 5.3x the fp32-MFMA rate.
template <int AMODE, int BM, int BN, int WM, int WN, int X3>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(GemmArgs g, const u16* __restrict__ Bw) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_F4 = BM * BK16 / 4 / 256;        // fp32 float4 loads per thread (4 for BM = 128)
    constexpr int B_V8 = BN * BK16 / 8 / 256;        // 16-byte bf16 loads per thread (2 for BN = 128, 1 for 64)
    __shared__ __attribute__((aligned(16))) u16 As[(1 + X3) * BM * LDS16];
    __shared__ __attribute__((aligned(16))) u16 Bs[(1 + X3) * BN * LDS16];
    const u16* __restrict__ Bw_lo = Bw + (long long)g.N * g.K;     // X3: second plane

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const float* __restrict__ A = g.A;
    // XCD-aware tile order: the dispatcher places linear block id b on XCD b % 8 (each XCD has a private 4 MB L2), so
    // give every XCD a CONTIGUOUS range of tiles -- neighbouring conv tiles share their tap windows through that L2.
    // Bijective for any grid size; a different hardware placement would only change speed, never results.
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
    const int m0 = tile_y * BM, n0 = tile_x * BN;

    int a_b[A_F4], a_d[A_F4], a_h[A_F4], a_w[A_F4];
    bool a_rowok[A_F4];
    if (AMODE == A_CONV) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int m = m0 + (tid >> 3) + 32 * i;
            a_rowok[i] = m < g.M;
            const int S = g.cg.S_out;
            int r = a_rowok[i] ? m : 0;
            a_w[i] = r % S; r /= S;
            a_h[i] = r % S; r /= S;
            a_d[i] = r % S; r /= S;
            a_b[i] = r;
        }
    }
    // NPF register sets of loads in flight.  The 64 x 64 tiles serve the few-row linear layers (act(), the released recipe's replay batch of
    // 1: M = 2048): 256 workgroups of 16-128 k-tiles that each waited a memory round trip (~2 us) for 6 MFMAs per wave -- 35 us per launch,
    // 48 launches per step.  With four tiles in flight (64 VGPRs; the accumulator is 16) the loop runs at the LDS / barrier rate.  Same k
    // order, same sums.
    // (also the 128 x 64 conv tiles: the patchify conv of a single sample is 63 workgroups of 250 k-tiles -- 0.37 ms of a 5.9 ms act())
    constexpr int NPF = ((BM == 64 && BN == 64 && AMODE == A_KCONTIG) || (BM == 128 && BN == 64 && AMODE == A_CONV)) ? 4 : 1;
    float4 ra[NPF][A_F4];
    uint4 rb[NPF][B_V8], rb_lo[NPF][X3 ? B_V8 : 1];

    auto load_tile = [&](int kt, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int k0 = kt * BK16;
        if (AMODE == A_KCONTIG) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int m = m0 + (tid >> 3) + 32 * i;
                const int k = k0 + (tid & 7) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < g.M && k < g.K) v = *reinterpret_cast<const float4*>(A + (long long)m * g.sAm + k);
                ra[SET][i] = v;
            }
        } else {
            const ConvGeom& c = g.cg;
            const int Ct = c.C0 + c.C1;
            const int tap = k0 / Ct;
            const int cc = k0 - tap * Ct + (tid & 7) * 4;
            const int tw = tap % c.kext, th = (tap / c.kext) % c.kext, td = tap / (c.kext * c.kext);
            const bool second = cc >= c.C0;
            const float* src = second ? c.src1 : c.src0;
            const int Cs = second ? c.C1 : c.C0;
            const int ch = second ? cc - c.C0 : cc;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                int id = a_d[i] * c.stride + td + c.off;
                int ih = a_h[i] * c.stride + th + c.off;
                int iw = a_w[i] * c.stride + tw + c.off;
                bool ok = a_rowok[i] && (k0 < g.K);
                if (c.replicate) {
                    id = min(max(id, 0), c.S_in - 1);
                    ih = min(max(ih, 0), c.S_in - 1);
                    iw = min(max(iw, 0), c.S_in - 1);
                } else {
                    ok = ok && id >= 0 && id < c.S_in && ih >= 0 && ih < c.S_in && iw >= 0 && iw < c.S_in;
                }
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const long long vox = (((long long)a_b[i] * c.S_in + id) * c.S_in + ih) * c.S_in + iw;
                    v = *reinterpret_cast<const float4*>(src + vox * Cs + ch);
                }
                ra[SET][i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_V8; ++i) {
            const int n = n0 + (tid >> 2) + 64 * i;
            const int k = k0 + (tid & 3) * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (n < g.N && k < g.K) v = *reinterpret_cast<const uint4*>(Bw + (long long)n * g.K + k);
            rb[SET][i] = v;
            if (X3) {
                uint4 w = make_uint4(0u, 0u, 0u, 0u);
                if (n < g.N && k < g.K) w = *reinterpret_cast<const uint4*>(Bw_lo + (long long)n * g.K + k);
                rb_lo[SET][i] = w;
            }
        }
    };
    auto store_tile = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int r = (tid >> 3) + 32 * i, kq = (tid & 7) * 4;
            uint2 p;
            p.x = pack_bf16(ra[SET][i].x, ra[SET][i].y);
            p.y = pack_bf16(ra[SET][i].z, ra[SET][i].w);
            *reinterpret_cast<uint2*>(&As[r * LDS16 + kq]) = p;
            if (X3) {
                // residuals a - float(hi): exact in fp32, then rounded to bf16
                uint2 q;
                q.x = pack_bf16(ra[SET][i].x - __uint_as_float(p.x << 16), ra[SET][i].y - __uint_as_float(p.x & 0xffff0000u));
                q.y = pack_bf16(ra[SET][i].z - __uint_as_float(p.y << 16), ra[SET][i].w - __uint_as_float(p.y & 0xffff0000u));
                *reinterpret_cast<uint2*>(&As[BM * LDS16 + r * LDS16 + kq]) = q;
            }
        }
#pragma unroll
        for (int i = 0; i < B_V8; ++i) {
            const int r = (tid >> 2) + 64 * i, kq = (tid & 3) * 8;
            *reinterpret_cast<uint4*>(&Bs[r * LDS16 + kq]) = rb[SET][i];
            if (X3) *reinterpret_cast<uint4*>(&Bs[BN * LDS16 + r * LDS16 + kq]) = rb_lo[SET][i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (g.K + BK16 - 1) / BK16;
    auto mma_tile = [&]() {
        const int lk = (lane >> 5) * 8, lm = lane & 31;
#pragma unroll
        for (int kk = 0; kk < BK16; kk += 16) {
            bf16x8 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                av[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * (BM / WM) + i * 32 + lm) * LDS16 + kk + lk]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bv[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * (BN / WN) + j * 32 + lm) * LDS16 + kk + lk]);
            if (X3) {
                bf16x8 al[TM], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    al[i] = *reinterpret_cast<const bf16x8*>(&As[BM * LDS16 + (wm * (BM / WM) + i * 32 + lm) * LDS16 + kk + lk]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bl[j] = *reinterpret_cast<const bf16x8*>(&Bs[BN * LDS16 + (wn * (BN / WN) + j * 32 + lm) * LDS16 + kk + lk]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bv[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bl[j], acc[i][j], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    auto one_tile = [&](int kt, auto set_c) {
        __syncthreads();
        store_tile(set_c);
        __syncthreads();
        if (kt + NPF < nkt) load_tile(kt + NPF, set_c);
        mma_tile();
    };
    if (NPF == 1) {
        load_tile(0, std::integral_constant<int, 0>{});
        for (int kt = 0; kt < nkt; ++kt) one_tile(kt, std::integral_constant<int, 0>{});
    } else {
        load_tile(0, std::integral_constant<int, 0>{});
        if (1 < nkt) load_tile(1, std::integral_constant<int, 1 % NPF>{});
        if (2 < nkt) load_tile(2, std::integral_constant<int, 2 % NPF>{});
        if (3 < nkt) load_tile(3, std::integral_constant<int, 3 % NPF>{});
#pragma unroll 1
        for (int kt = 0; kt < nkt; kt += 4) {
            one_tile(kt, std::integral_constant<int, 0>{});
            if (kt + 1 < nkt) one_tile(kt + 1, std::integral_constant<int, 1 % NPF>{});
            if (kt + 2 < nkt) one_tile(kt + 2, std::integral_constant<int, 2 % NPF>{});
            if (kt + 3 < nkt) one_tile(kt + 3, std::integral_constant<int, 3 % NPF>{});
        }
    }

    float* __restrict__ C = g.C;
    const float* __restrict__ R = g.residual;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bsv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = g.alpha * acc[i][j][r] + bsv;
                if (g.act == ACT_LRELU) v = v > 0.f ? v : v * g.slope;
                long long off;
                if (g.d2s_s > 0) {
                    const int s = g.d2s_s, G = g.d2s_G, Cc = g.d2s_C;
                    const int ph = n / Cc, co = n - ph * Cc;
                    const int rw = ph % s, rh = (ph / s) % s, rd = ph / (s * s);
                    int q = m;
                    const int qw = q % G; q /= G;
                    const int qh = q % G; q /= G;
                    const int qd = q % G; q /= G;
                    const long long Vv = (long long)G * s;
                    off = ((((long long)q * Vv + qd * s + rd) * Vv + qh * s + rh) * Vv + qw * s + rw) * Cc + co;
                } else {
                    off = (long long)m * g.ldc + n;
                }
                if (R) v += R[off];
                if (g.accumulate) v += C[off];
                C[off] = v;
            }
        }
    }
}

template <int AMODE, int X3>
int launch_gemm_bf16(const GemmArgs& g, const u16* Bw, hipStream_t st) {
    if (AMODE == 0 && g.N > 64 && (long long)vxb_cdiv(g.N, 128) * vxb_cdiv(g.M, 128) < 192) {
        // few rows (act(): M = 2048 latents): 64 x 64 tiles fill the chip four times better; same k order, same sums
        dim3 grid(vxb_cdiv(g.N, 64), vxb_cdiv(g.M, 64), 1);
        hipLaunchKernelGGL((gemm_bf16_kernel<AMODE == 0 ? 0 : AMODE, 64, 64, 2, 2, X3>), grid, dim3(256), 0, st, g, Bw);
    } else if (g.N > 64) {
        dim3 grid(vxb_cdiv(g.N, 128), vxb_cdiv(g.M, 128), 1);
        hipLaunchKernelGGL((gemm_bf16_kernel<AMODE, 128, 128, 2, 2, X3>), grid, dim3(256), 0, st, g, Bw);
    } else {
        dim3 grid(vxb_cdiv(g.N, 64), vxb_cdiv(g.M, 128), 1);
        hipLaunchKernelGGL((gemm_bf16_kernel<AMODE, 128, 64, 2, 2, X3>), grid, dim3(256), 0, st, g, Bw);
    }
    if (hipGetLastError() != hipSuccess) return VXB_ELAUNCH;
    return VXB_OK;
}

}  // namespace

// C[M,N] (+)= act(A[M,K] (fp32, row stride lda, rounded to bf16) @ Bw[N,K]^T (bf16) + bias) (+ residual); K % 8 == 0.
extern "C" int vxb_gemm_bf16w_f32(const float* A, int64_t lda, const void* Bw, float* C, int64_t ldc, const float* bias,
                                  const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                                  vxb_stream_t stream) {
    if (!A || !Bw || !C || M < 1 || N < 1 || K < 8) return VXB_EARG;
    if ((K & 7) || (lda & 3) || !aligned16(A) || !aligned16(Bw)) return VXB_ESIZE;
    GemmArgs g = {};
    g.A = A; g.C = C; g.bias = bias; g.residual = residual; g.M = M; g.N = N; g.K = K; g.sAm = lda; g.sAk = 1; g.ldc = ldc;
    g.H = 1; g.alpha = 1.f; g.act = act; g.slope = slope; g.accumulate = accumulate;
    return launch_gemm_bf16<A_KCONTIG, 0>(g, (const u16*)Bw, (hipStream_t)stream);
}

// "bf16x3" twin: Bw = bf16 planes [2][N][K] (hi, lo); products hi*hi + hi*lo + lo*hi, fp32-faithful to ~2^-16.
extern "C" int vxb_gemm_bf16x3_f32(const float* A, int64_t lda, const void* Bw, float* C, int64_t ldc, const float* bias,
                                   const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                                   vxb_stream_t stream) {
    if (!A || !Bw || !C || M < 1 || N < 1 || K < 8) return VXB_EARG;
    if ((K & 7) || (lda & 3) || !aligned16(A) || !aligned16(Bw)) return VXB_ESIZE;
    GemmArgs g = {};
    g.A = A; g.C = C; g.bias = bias; g.residual = residual; g.M = M; g.N = N; g.K = K; g.sAm = lda; g.sAk = 1; g.ldc = ldc;
    g.H = 1; g.alpha = 1.f; g.act = act; g.slope = slope; g.accumulate = accumulate;
    return launch_gemm_bf16<A_KCONTIG, 1>(g, (const u16*)Bw, (hipStream_t)stream);
}

// bf16-matrix-core twin of vxb_conv3d_f32: same geometry, weights as bf16 [N][K = kext^3*(C0+C1)]; C0, C1 multiples of 32.
static int conv3d_bf16_impl(int x3, const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                           int stride, int kext, int off, int replicate, const void* wt_bf16, int N,
                           const float* bias, float* out, int64_t ldc, int act, float slope, int accumulate,
                           int d2s_s, int d2s_C, vxb_stream_t stream) {
    if (!src0 || !wt_bf16 || !out || B < 1 || S_in < 1 || S_out < 1 || kext < 1 || stride < 1 || N < 1) return VXB_EARG;
    if ((C0 & 31) || (C1 & 31) || C0 < 32 || (C1 > 0 && !src1)) return VXB_ESIZE;
    if (!aligned16(src0) || !aligned16(wt_bf16) || (src1 && !aligned16(src1))) return VXB_ESIZE;
    const long long M = (long long)B * S_out * S_out * S_out;
    const long long K = (long long)kext * kext * kext * (C0 + C1);
    if (M >= INT32_MAX || K >= INT32_MAX) return VXB_ESIZE;
    GemmArgs g = {};
    g.A = src0; g.C = out; g.bias = bias; g.M = (int)M; g.N = N; g.K = (int)K; g.ldc = ldc; g.H = 1;
    g.alpha = 1.f; g.act = act; g.slope = slope; g.accumulate = accumulate;
    g.d2s_s = d2s_s; g.d2s_G = S_out; g.d2s_C = d2s_C;
    g.cg.src0 = src0; g.cg.src1 = src1; g.cg.C0 = C0; g.cg.C1 = C1; g.cg.S_in = S_in; g.cg.S_out = S_out;
    g.cg.stride = stride; g.cg.kext = kext; g.cg.off = off; g.cg.replicate = replicate;
    if (d2s_s > 0 && (d2s_C < 1 || N % d2s_C)) return VXB_EARG;
    if (x3) return launch_gemm_bf16<A_CONV, 1>(g, (const u16*)wt_bf16, (hipStream_t)stream);
    return launch_gemm_bf16<A_CONV, 0>(g, (const u16*)wt_bf16, (hipStream_t)stream);
}

extern "C" int vxb_conv3d_bf16w_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                    int stride, int kext, int off, int replicate, const void* wt_bf16, int N,
                                    const float* bias, float* out, int64_t ldc, int act, float slope, int accumulate,
                                    int d2s_s, int d2s_C, vxb_stream_t stream) {
    return conv3d_bf16_impl(0, src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, replicate, wt_bf16, N, bias, out, ldc,
                            act, slope, accumulate, d2s_s, d2s_C, stream);
}

// "bf16x3" twin: weights = bf16 planes [2][N][K] (hi, lo).
extern "C" int vxb_conv3d_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                     int stride, int kext, int off, int replicate, const void* wt_bf16, int N,
                                     const float* bias, float* out, int64_t ldc, int act, float slope, int accumulate,
                                     int d2s_s, int d2s_C, vxb_stream_t stream) {
    return conv3d_bf16_impl(1, src0, src1, C0, C1, B, S_in, S_out, stride, kext, off, replicate, wt_bf16, N, bias, out, ldc,
                            act, slope, accumulate, d2s_s, d2s_C, stream);
}
