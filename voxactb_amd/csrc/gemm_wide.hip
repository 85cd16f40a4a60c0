// Wide bf16x3 GEMM for the linear layers whose output width is a multiple of 512 (perceiver_lang_io.py:74-132: to_q, to_kv, to_out,
// both FeedForward projections and their data gradients):
//     C[M, N] (+)= act(A[M, K] @ W[N, K]^T + bias) (+ residual),   every product as hi*hi + hi*lo + lo*hi (common.h / DESIGN 4a).
//
// The 128 x 128 register-staged kernel (gemm_conv.hip: gemm_bf16_kernel) gives these shapes 1024 workgroups of 16-128 k-tiles that
// each keep ONE tile of loads in flight: a workgroup waits a memory round trip (~2.5 us under load) per 768 matrix-pipe cycles, and
// the A matrix is re-read by the four column blocks from beyond the L2 (28 % pipe-busy, 39 % of the wave time in s_waitcnt:
// profiles/r03_v3_sq_summary.txt).  Here ONE workgroup of 8 waves owns 128 rows and 512 columns (all of them for N = 512; grid.y walks
// the 512-column groups of a wider layer, whose small A matrix is then re-read out of the L2):
//   * A (fp32, streamed once from HBM) travels global -> registers -> hi / lo split -> LDS, three k-tiles ahead in three rotating
//     register sets (8 floats per thread and tile), two LDS stages, ONE barrier per 32-deep k-tile;
//   * W never touches LDS: wave w owns columns 64 w .. 64 w + 63 and reads its B fragments straight from global memory in MFMA
//     fragment order (ops.gemm_wfrag: 1 KB lane-contiguous per fragment, L2-resident, shared by all workgroups), two k-steps ahead
//     in three rotating register sets -- issued BEFORE the A load of the same iteration, so that waiting for them (vmcnt counts in
//     order) never waits for the newest A tile;
//   * wave tile 128 x 64 = 4 x 2 MFMA tiles (128 accumulator VGPRs): 48 MFMAs per wave and k-tile against 16 ds_read_b128.
// Per output element the products are accumulated in the same order as in gemm_bf16_kernel (k ascending; lo*hi, hi*lo, hi*hi per
// 16-deep step), so the results are bit-identical to the kernel this replaces.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int WBM = 128, WBK = 32, WLD = 40;       // rows per workgroup, k per tile, u16 per LDS row (32 + 8 pad: conflict-free b128 reads)

struct GwArgs {
    const float* A;
    long long lda;
    const u16* Bfrag;        // [N / 32][K / 16][2 planes][64 lanes][8] (16 column tiles per 512-column group)
    float* C;
    long long ldc;
    const float* bias;
    const float* residual;
    int M, K;
    int act;
    float slope;
    int accumulate;
};

__global__ void __launch_bounds__(512) gemm_wide_x3_kernel(GwArgs g) {
    __shared__ __attribute__((aligned(16))) u16 As[2][2][WBM * WLD];          // [stage][plane]: 40 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);                  // this wave's 64 columns
    const int m0 = blockIdx.x * WBM;
    const int nkt = g.K / WBK, nks = g.K >> 4;

    // ---- A: thread -> row tid / 4, 8 consecutive k
    const int ar = tid >> 2, akq = (tid & 3) * 8;
    const float* __restrict__ ap = g.A + (long long)min(m0 + ar, g.M - 1) * g.lda + akq;
    float4 ra[3][2];
#define GW_LOADA(S, kt_)                                                                                              \
    {                                                                                                                \
        const float* p_ = ap + (long long)min((kt_), nkt - 1) * WBK;                                                 \
        ra[S][0] = *reinterpret_cast<const float4*>(p_);                                                             \
        ra[S][1] = *reinterpret_cast<const float4*>(p_ + 4);                                                         \
    }
#define GW_STOREA(S, stage_)                                                                                          \
    {                                                                                                                \
        const float v_[8] = {ra[S][0].x, ra[S][0].y, ra[S][0].z, ra[S][0].w, ra[S][1].x, ra[S][1].y, ra[S][1].z, ra[S][1].w}; \
        uint4 h_, l_;                                                                                                \
        unsigned* hp_ = reinterpret_cast<unsigned*>(&h_);                                                            \
        unsigned* lp_ = reinterpret_cast<unsigned*>(&l_);                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                               \
            hp_[e] = vxb_pack_bf16(v_[2 * e], v_[2 * e + 1]);                                                        \
            lp_[e] = vxb_pack_bf16(v_[2 * e] - __uint_as_float(hp_[e] << 16), v_[2 * e + 1] - __uint_as_float(hp_[e] & 0xffff0000u)); \
        }                                                                                                            \
        *reinterpret_cast<uint4*>(&As[(stage_)][0][ar * WLD + akq]) = h_;                                            \
        *reinterpret_cast<uint4*>(&As[(stage_)][1][ar * WLD + akq]) = l_;                                            \
    }
    // ---- B fragments of this wave's two 32-column tiles: frag(j, ks, plane) at ((j * nks + ks) * 2 + plane) * 512 + lane * 8
    const int cg = blockIdx.y;                                                // 512-column group (N = 512: one)
    const u16* __restrict__ bfb = g.Bfrag + ((long long)(cg * 16 + wn * 2) * nks * 2) * 512 + lane * 8;
    bf16x8 bq[3][2][2];
#define GW_LOADB(S, ks_)                                                                                              \
    {                                                                                                                \
        const long long k_ = min((ks_), nks - 1);                                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
        _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                 \
            bq[S][j][p] = *reinterpret_cast<const bf16x8*>(bfb + (((long long)j * nks + k_) * 2 + p) * 512);         \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lm = lane & 31, lk = (lane >> 5) * 8;
    // one 16-deep step on LDS stage st_, k offset kk_, B set SB: row tiles in pairs, terms in the order lo*hi, hi*lo, hi*hi
#define GW_STEP(st_, kk_, SB)                                                                                         \
    {                                                                                                                \
        _Pragma("unroll") for (int ip = 0; ip < 2; ++ip) {                                                            \
            bf16x8 ah_[2], al_[2];                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
                ah_[i] = *reinterpret_cast<const bf16x8*>(&As[(st_)][0][((2 * ip + i) * 32 + lm) * WLD + (kk_) + lk]); \
                al_[i] = *reinterpret_cast<const bf16x8*>(&As[(st_)][1][((2 * ip + i) * 32 + lm) * WLD + (kk_) + lk]); \
            }                                                                                                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[2 * ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_[i], bq[SB][j][0], acc[2 * ip + i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[2 * ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_[i], bq[SB][j][1], acc[2 * ip + i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[2 * ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_[i], bq[SB][j][0], acc[2 * ip + i][j], 0, 0, 0); \
        }                                                                                                            \
    }
    // one k-tile kt_ (kt_ % 3 == R_): LDS stage kt_ & 1 holds it; register set (R_+1) % 3 holds tile kt_+1 (stored into the other
    // stage now), sets (R_+2) % 3 and R_ hold tiles kt_+2, kt_+3; the freed set is refilled with tile kt_+4 AFTER this tile's B loads
#define GW_TILE(kt_, R_)                                                                                              \
    {                                                                                                                \
        const int tk_ = (kt_);                                                                                       \
        const int st = tk_ & 1;                                                                                      \
        GW_STOREA((R_ + 1) % 3, st ^ 1)                                                                              \
        GW_LOADB((2 * R_ + 2) % 3, 2 * tk_ + 2)                                                                      \
        GW_STEP(st, 0, (2 * R_) % 3)                                                                                 \
        GW_LOADB((2 * R_ + 3) % 3, 2 * tk_ + 3)                                                                      \
        GW_LOADA((R_ + 1) % 3, tk_ + 4)                                                                              \
        GW_STEP(st, 16, (2 * R_ + 1) % 3)                                                                            \
        __syncthreads();                                                                                             \
    }

    GW_LOADA(0, 0)
    GW_STOREA(0, 0)
    GW_LOADB(0, 0)
    GW_LOADB(1, 1)
    GW_LOADA(1, 1)
    GW_LOADA(2, 2)
    GW_LOADA(0, 3)
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nkt; kt += 3) {
        GW_TILE(kt, 0)
        if (kt + 1 < nkt) GW_TILE(kt + 1, 1)
        if (kt + 2 < nkt) GW_TILE(kt + 2, 2)
    }
#undef GW_LOADA
#undef GW_STOREA
#undef GW_LOADB
#undef GW_STEP
#undef GW_TILE

    float* __restrict__ C = g.C;
    const float* __restrict__ R = g.residual;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = cg * 512 + wn * 64 + j * 32 + (lane & 31);
            const float bsv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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

}  // namespace

// C[M, N] (+)= act(A[M, K] (fp32, row stride lda) @ W^T + bias) (+ residual) in 'bf16x3', N % 512 == 0, W given ONLY in MFMA fragment
// order (Bw_frag: [N / 32][K / 16][2][64][8] bf16 = ops.gemm_wfrag of the hi / lo planes [2][N][K]); K % 32 == 0, K >= 64.  Same contract and
// the same bits as vxb_gemm_bf16x3_f32 for these shapes (act: 0 none, 1 LeakyReLU(slope)).
extern "C" int vxb_gemm_wide_bf16x3_f32(const float* A, int64_t lda, const void* Bw_frag, float* C, int64_t ldc, const float* bias,
                                        const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                                        vxb_stream_t stream) {
    if (!A || !Bw_frag || !C || M < 1 || K < 64) return VXB_EARG;
    if (N < 512 || (N & 511) || (K & 31) || (lda & 3) || (((uintptr_t)A | (uintptr_t)Bw_frag) & 15)) return VXB_ESIZE;
    GwArgs g;
    g.A = A; g.lda = lda; g.Bfrag = (const u16*)Bw_frag; g.C = C; g.ldc = ldc; g.bias = bias; g.residual = residual;
    g.M = M; g.K = K; g.act = act; g.slope = slope; g.accumulate = accumulate;
    hipLaunchKernelGGL(gemm_wide_x3_kernel, dim3(vxb_cdiv(M, WBM), N / 512), dim3(512), 0, (hipStream_t)stream, g);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
