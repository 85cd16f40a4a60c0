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
    // GEGLU fused into the epilogue (perceiver_lang_io.py:74-78; F = half the width of the up-projection):
    //   geglu 1 (forward of the up-projection, N = 2 F): the weight fragments come with the rows of every 64-column block interleaved
    //     as [32 value columns 32 b .. | 32 gate columns F + 32 b ..] (vxb_split_bf16_batch_f32 flag bit 2), so a lane holds h[m][c] and
    //     h[m][F + c] side by side: C = h [M][2 F] in its natural column order (the backward pass reads it) and C2 = h[:, c] *
    //     gelu(h[:, F + c]) [M][F] -- the same bits as vxb_geglu_fwd_f32 on the stored h
    //   geglu 2 (data gradient of the down-projection, N = F): the tile is d(gg); C = dh [M][2 F] = [d * gelu(h_gate) | d * h_value *
    //     gelu'(h_gate)] with h = H [M][2 F] -- vxb_geglu_bwd_f32 without the round trip of d(gg) through HBM
    int geglu, F;
    float* C2;
    const float* H;
    // X2 kernel ("fp16x2", the data gradients of the linear layers: two MFMAs per product instead of three): A * scale[0] * 2^-4 is
    // carried as an fp16 hi | lo pair, Bfrag holds ONE fp16 plane ([N / 32][K / 16][64 lanes][8]), the sums are multiplied by
    // scale[1] * 2^4.  scale = {2^k, 2^-k} on the device: the operand scale of this dY from the weight-gradient launch of the same
    // linear_bwd (delayed by one step, 5 bits of headroom; 4 more here: a propagating gradient may jump 512-fold between steps)
    const float* scale;
    int dbg;                 // timing experiments (WRONG results): 1 no B loads in the loop, 2 no A loads, 4 no A store / barrier, 8 no epilogue; 32: 2-D grid (row blocks fastest) for N > 512
    int ncg;                 // NW = 4 launches: column groups per row block (1-D grid)
    int N_total;             // columns of the whole product (row length of aux16)
    u16* aux16 = nullptr;    // plain epilogue, optional: the result also as ONE fp16 plane [M][N] (saturating RNE: the bits of vxb_split_f16_f32 on C)
};

typedef _Float16 gw_f16x8 __attribute__((ext_vector_type(8)));
template <int X2>
__device__ __forceinline__ f32x16 gw_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    if (X2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gw_f16x8, a), __builtin_bit_cast(gw_f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// X2 = 0: bf16x3 (hi*hi + hi*lo + lo*hi), X2 = 1: fp16x2 (A hi | lo, one weight plane)
// NW = 8: one workgroup of 8 waves per CU owns 128 x 512; NW = 4: 128 x 256 per workgroup of 4 waves, TWO workgroups per CU (one wave of
// each per SIMD), so that one workgroup's barrier, prologue and epilogue stores run under the other's MFMAs.  The column groups of a row
// block get consecutive virtual block ids on the same XCD (A comes out of that XCD's L2 for all but the first).
template <int X2, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) gemm_wide_kernel(GwArgs g) {
    __shared__ __attribute__((aligned(16))) u16 As[2][2][WBM * WLD];          // [stage][plane]: 40 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);                  // this wave's 64 columns
    int bx = blockIdx.x, cg = blockIdx.y;                                     // row block, column group of 64 NW columns
    if (g.ncg > 0) {                                                          // 1-D grid: the column groups of a row block side by side on one XCD
        const int total = gridDim.x, ncg = g.ncg;
        int vb = bx;
        if ((total & 7) == 0) vb = (bx & 7) * (total >> 3) + (bx >> 3);
        cg = vb % ncg;
        bx = vb / ncg;
    }
    const int m0 = bx * WBM;
    const int nkt = g.K / WBK, nks = g.K >> 4;
    constexpr int NPB = X2 ? 1 : 2;                                           // weight planes
    const float in_sc = X2 ? g.scale[0] * 0.0625f : 1.f, out_sc = X2 ? g.scale[1] * 16.f : 1.f;

    // ---- A: thread -> row tid / 4, 8 consecutive k
    constexpr int HR = 8 / NW;                                                // rows per thread (NW = 4: rows ar and ar + 64)
    const int ar = tid >> 2, akq = (tid & 3) * 8;
    const float* __restrict__ ap[HR];
#pragma unroll
    for (int h = 0; h < HR; ++h) ap[h] = g.A + (long long)min(m0 + ar + 64 * h, g.M - 1) * g.lda + akq;
    float4 ra[3][HR][2];
#define GW_LOADA(S, kt_)                                                                                              \
    _Pragma("unroll") for (int h = 0; h < HR; ++h) {                                                                  \
        const float* p_ = ap[h] + (long long)min((kt_), nkt - 1) * WBK;                                              \
        ra[S][h][0] = *reinterpret_cast<const float4*>(p_);                                                          \
        ra[S][h][1] = *reinterpret_cast<const float4*>(p_ + 4);                                                      \
    }
#define GW_STOREA(S, stage_)                                                                                          \
    _Pragma("unroll") for (int h = 0; h < HR; ++h) {                                                                  \
        const float v_[8] = {ra[S][h][0].x, ra[S][h][0].y, ra[S][h][0].z, ra[S][h][0].w, ra[S][h][1].x, ra[S][h][1].y, ra[S][h][1].z, ra[S][h][1].w}; \
        uint4 h_, l_;                                                                                                \
        unsigned* hp_ = reinterpret_cast<unsigned*>(&h_);                                                            \
        unsigned* lp_ = reinterpret_cast<unsigned*>(&l_);                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                               \
            if (X2) {                                                                                                \
                const float a_ = vxb_sat_f16(v_[2 * e] * in_sc);          /* (NaN / inf stay non-finite: common.h) */  \
                const float b_ = vxb_sat_f16(v_[2 * e + 1] * in_sc);                                                 \
                hp_[e] = vxb_pack_f16(a_, b_);                                                                       \
                lp_[e] = vxb_pack_f16(a_ - (float)__builtin_bit_cast(_Float16, (unsigned short)(hp_[e] & 0xffffu)),  \
                                      b_ - (float)__builtin_bit_cast(_Float16, (unsigned short)(hp_[e] >> 16)));     \
            } else {                                                                                                 \
                hp_[e] = vxb_pack_bf16(v_[2 * e], v_[2 * e + 1]);                                                    \
                lp_[e] = vxb_pack_bf16(v_[2 * e] - __uint_as_float(hp_[e] << 16), v_[2 * e + 1] - __uint_as_float(hp_[e] & 0xffff0000u)); \
            }                                                                                                        \
        }                                                                                                            \
        *reinterpret_cast<uint4*>(&As[(stage_)][0][(ar + 64 * h) * WLD + akq]) = h_;                                 \
        *reinterpret_cast<uint4*>(&As[(stage_)][1][(ar + 64 * h) * WLD + akq]) = l_;                                 \
    }
    // ---- B fragments of this wave's two 32-column tiles: frag(j, ks, plane) at ((j * nks + ks) * 2 + plane) * 512 + lane * 8
    const u16* __restrict__ bfb = g.Bfrag + ((long long)(cg * 2 * NW + wn * 2) * nks * NPB) * 512 + lane * 8;
    bf16x8 bq[3][2][2];
#define GW_LOADB(S, ks_)                                                                                              \
    {                                                                                                                \
        const long long k_ = min((ks_), nks - 1);                                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
        _Pragma("unroll") for (int p = 0; p < NPB; ++p)                                                               \
            bq[S][j][p] = *reinterpret_cast<const bf16x8*>(bfb + (((long long)j * nks + k_) * NPB + p) * 512);       \
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
                acc[2 * ip + i][j] = gw_mfma<X2>(al_[i], bq[SB][j][0], acc[2 * ip + i][j]);                          \
            if (!X2) {                                                                                               \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                    acc[2 * ip + i][j] = gw_mfma<X2>(ah_[i], bq[SB][j][1], acc[2 * ip + i][j]);                      \
            }                                                                                                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[2 * ip + i][j] = gw_mfma<X2>(ah_[i], bq[SB][j][0], acc[2 * ip + i][j]);                          \
        }                                                                                                            \
    }
    // one k-tile kt_ (kt_ % 3 == R_): LDS stage kt_ & 1 holds it; register set (R_+1) % 3 holds tile kt_+1 (stored into the other
    // stage now), sets (R_+2) % 3 and R_ hold tiles kt_+2, kt_+3; the freed set is refilled with tile kt_+4 AFTER this tile's B loads
#define GW_TILE(kt_, R_)                                                                                              \
    {                                                                                                                \
        const int tk_ = (kt_);                                                                                       \
        const int st = tk_ & 1;                                                                                      \
        if (!(dbg & 4)) GW_STOREA((R_ + 1) % 3, st ^ 1)                                                              \
        if (!(dbg & 1)) GW_LOADB((2 * R_ + 2) % 3, 2 * tk_ + 2)                                                      \
        GW_STEP(st, 0, (2 * R_) % 3)                                                                                 \
        if (!(dbg & 1)) GW_LOADB((2 * R_ + 3) % 3, 2 * tk_ + 3)                                                      \
        if (!(dbg & 2)) GW_LOADA((R_ + 1) % 3, tk_ + 4)                                                              \
        GW_STEP(st, 16, (2 * R_ + 1) % 3)                                                                            \
        if (!(dbg & 4)) __syncthreads();                                                                             \
    }

#ifdef VXB_GW_ABLATE       // timing-experiment build (tools/bench_gemm_ablate.py): branches around the loads split the loop's basic block
    const int dbg = g.dbg;
#else
    constexpr int dbg = 0;
#endif
    GW_LOADA(0, 0)
    GW_STOREA(0, 0)
    GW_LOADB(0, 0)
    GW_LOADB(1, 1)
    GW_LOADA(1, 1)
    GW_LOADA(2, 2)
    GW_LOADA(0, 3)
    __syncthreads();
    // Whole rounds of three k-tiles in a loop body WITHOUT branches, the 0 - 2 tiles left over after it: with `if (kt + 1 < nkt)` inside
    // the body the compiler cannot count the loads in flight at the loop header and drains them all there (s_waitcnt vmcnt(0) once per
    // round: the newest A tile, issued half a k-tile earlier, had to land before the round could start -- round 6).
    // The first round is peeled: at the loop header the state after the prologue (4 loads younger than the set converted first) would
    // otherwise be merged with the steady state's (20 younger loads) into s_waitcnt vmcnt(4).
    int kt = 0;
    if (nkt >= 3) {
        GW_TILE(0, 0)
        GW_TILE(1, 1)
        GW_TILE(2, 2)
        kt = 3;
#pragma unroll 1
        for (; kt + 3 <= nkt; kt += 3) {
            GW_TILE(kt, 0)
            GW_TILE(kt + 1, 1)
            GW_TILE(kt + 2, 2)
        }
    }
    if (kt < nkt) {
        GW_TILE(kt, 0)
        if (kt + 1 < nkt) GW_TILE(kt + 1, 1)
    }
#undef GW_LOADA
#undef GW_STOREA
#undef GW_LOADB
#undef GW_STEP
#undef GW_TILE

    float* __restrict__ C = g.C;
    const float* __restrict__ R = g.residual;
    if ((dbg & 8) && acc[0][0][0] != 12345.f) return;
    // ---- row-contiguous epilogue (NW = 8; round 6).  Straight out of the accumulators a store instruction writes two 128-byte pieces (32
    // columns of two rows) and a 128 x 512 tile takes 1024 of them per workgroup: measured, the epilogue is 0.27 of the 0.59 ms of a
    // 4096 x 512 launch (tools/bench_gemm_ablate.py) -- 7.7 GB/s per CU.  Here the tile goes through LDS (the operand stages are free),
    // sixteen rows at a time, and leaves as 1 KB per wave instruction: whole 2 KB rows (1 KB row pieces of h's two halves and of gg with
    // GEGLU).  Per element the same operations in the same order as below: identical bits.
    if (NW == 8 && !(g.dbg & 64) && !(g.ldc & 3) && !((uintptr_t)C & 15) && (!R || !((uintptr_t)R & 15)) && !((uintptr_t)g.bias & 15) &&
        (g.geglu != 1 || (!(g.F & 3) && !((uintptr_t)g.C2 & 15))) && (g.geglu != 2 || (!(g.F & 3) && !((uintptr_t)g.H & 15)))) {
        constexpr int CLD = 512 + 8;
        float* Cs = reinterpret_cast<float*>(&As[0][0][0]);                 // [16][CLD] fp32 = 33 KB of the 40 KB
        const int lrow = 4 * (lane >> 5), lcol = lane & 31;
        if (g.geglu == 1) {
            const int F = g.F;
            const int er = tid >> 6, ec = (tid & 63) * 4;                   // 8 rows x 64 column quads per pass of the read-out
            const int c0 = cg * 256 + ec;                                   // value column; its gate is column F + c0
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
            if (g.bias) { bv = *reinterpret_cast<const float4*>(g.bias + c0); bg = *reinterpret_cast<const float4*>(g.bias + F + c0); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            const int r = 8 * hf + rr;
                            Cs[((r & 3) + 8 * ((r >> 2) & 1) + lrow) * CLD + j * 256 + wn * 32 + lcol] = acc[i][j][r];
                        }
                    __syncthreads();
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const int row = er + 8 * st;
                        const int m = m0 + i * 32 + 16 * hf + row;
                        if (m >= g.M) continue;
                        const float4 a = *reinterpret_cast<const float4*>(&Cs[row * CLD + ec]);
                        const float4 b = *reinterpret_cast<const float4*>(&Cs[row * CLD + 256 + ec]);
                        const float4 hv = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
                        const float4 hg = make_float4(b.x + bg.x, b.y + bg.y, b.z + bg.z, b.w + bg.w);
                        *reinterpret_cast<float4*>(C + (long long)m * g.ldc + c0) = hv;
                        *reinterpret_cast<float4*>(C + (long long)m * g.ldc + F + c0) = hg;
                        *reinterpret_cast<float4*>(g.C2 + (long long)m * F + c0) =
                            make_float4(hv.x * gelu_erf(hg.x), hv.y * gelu_erf(hg.y), hv.z * gelu_erf(hg.z), hv.w * gelu_erf(hg.w));
                    }
                }
            return;
        }
        const int er = tid >> 7, ec = (tid & 127) * 4;                      // 4 rows x 128 column quads per pass of the read-out
        const int n4 = cg * 512 + ec;
        if (g.geglu == 2) {
            // the tile is d(gg): dh = [d * gelu(h_gate) | d * h_value * gelu'(h_gate)], h read and dh written in whole row pieces
            const int F = g.F;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            const int r = 8 * hf + rr;
                            Cs[((r & 3) + 8 * ((r >> 2) & 1) + lrow) * CLD + wn * 64 + j * 32 + lcol] = X2 ? acc[i][j][r] * out_sc : acc[i][j][r];
                        }
                    __syncthreads();
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const int row = er + 4 * st;
                        const int m = m0 + i * 32 + 16 * hf + row;
                        if (m >= g.M) continue;
                        const float4 d = *reinterpret_cast<const float4*>(&Cs[row * CLD + ec]);
                        const long long o = (long long)m * 2 * F + n4;
                        const float4 a = *reinterpret_cast<const float4*>(g.H + o), gt = *reinterpret_cast<const float4*>(g.H + o + F);
                        *reinterpret_cast<float4*>(C + o) = make_float4(d.x * gelu_erf(gt.x), d.y * gelu_erf(gt.y), d.z * gelu_erf(gt.z), d.w * gelu_erf(gt.w));
                        *reinterpret_cast<float4*>(C + o + F) = make_float4(d.x * a.x * gelu_erf_grad(gt.x), d.y * a.y * gelu_erf_grad(gt.y),
                                                                            d.z * a.z * gelu_erf_grad(gt.z), d.w * a.w * gelu_erf_grad(gt.w));
                    }
                }
            return;
        }
        const float4 bs = g.bias ? *reinterpret_cast<const float4*>(g.bias + n4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = 8 * hf + rr;
                        Cs[((r & 3) + 8 * ((r >> 2) & 1) + lrow) * CLD + wn * 64 + j * 32 + lcol] = X2 ? acc[i][j][r] * out_sc : acc[i][j][r];
                    }
                __syncthreads();
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int row = er + 4 * st;
                    const int m = m0 + i * 32 + 16 * hf + row;
                    if (m >= g.M) continue;
                    float4 v = *reinterpret_cast<const float4*>(&Cs[row * CLD + ec]);
                    v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
                    if (g.act == 1) {
                        v.x = v.x > 0.f ? v.x : v.x * g.slope; v.y = v.y > 0.f ? v.y : v.y * g.slope;
                        v.z = v.z > 0.f ? v.z : v.z * g.slope; v.w = v.w > 0.f ? v.w : v.w * g.slope;
                    }
                    const long long off = (long long)m * g.ldc + n4;
                    if (R) { const float4 q = *reinterpret_cast<const float4*>(R + off); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
                    if (g.accumulate) { const float4 q = *reinterpret_cast<const float4*>(C + off); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
                    *reinterpret_cast<float4*>(C + off) = v;
                    if (g.aux16) {
                        uint2 ph;
                        ph.x = vxb_pack_f16(__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f));
                        ph.y = vxb_pack_f16(__builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f));
                        *reinterpret_cast<uint2*>(g.aux16 + (long long)m * g.N_total + n4) = ph;
                    }
                }
            }
        return;
    }
    if (g.geglu == 1) {
        const int F = g.F, c = (cg * NW + wn) * 32 + (lane & 31);          // value column; its gate is column F + c
        const float bv = g.bias ? g.bias[c] : 0.f, bg = g.bias ? g.bias[F + c] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                const float hv = acc[i][0][r] + bv, hg = acc[i][1][r] + bg;
                C[(long long)m * g.ldc + c] = hv;
                C[(long long)m * g.ldc + F + c] = hg;
                g.C2[(long long)m * F + c] = hv * gelu_erf(hg);
            }
        return;
    }
    if (g.geglu == 2) {
        const int F = g.F;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = cg * 64 * NW + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m >= g.M) continue;
                    const long long o = (long long)m * 2 * F + c;
                    const float a = g.H[o], gt = g.H[o + F], d = X2 ? acc[i][j][r] * out_sc : acc[i][j][r];
                    C[o] = d * gelu_erf(gt);
                    C[o + F] = d * a * gelu_erf_grad(gt);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = cg * 64 * NW + wn * 64 + j * 32 + (lane & 31);
            const float bsv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = (X2 ? acc[i][j][r] * out_sc : acc[i][j][r]) + bsv;
                if (g.act == 1) v = v > 0.f ? v : v * g.slope;
                const long long off = (long long)m * g.ldc + n;
                if (R) v += R[off];
                if (g.accumulate) v += C[off];
                C[off] = v;
                if (g.aux16) g.aux16[(long long)m * g.N_total + n] = (u16)(vxb_pack_f16(__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f), 0.f) & 0xffffu);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same workgroup shape for the polyphase up-conv forward (network_utils.py:245-250 as the low-res 3^3 conv with s^3 * 64 phase
// columns and a depth-to-space store, ops.conv3_polyphase_fwd): implicit GEMM  out[m = low-res voxel][n = (phase, co)] =
// sum over (tap, ci) z[clamp(m + tap + off)][ci] * W[n][(tap, ci)],  N = 8000, K = 27 * 64 with 17.6 of the 27 (tap, phase) weight
// blocks non-zero.  The 128 x 128 direct-to-LDS kernel (gemm_dl.hip) asks the vector L1 for 12 KB of operands per 24 MFMAs and wave
// (64 B/clk per CU at full matrix rate: the L1's peak) and sits at 38 % pipe-busy; with 128 x 64 wave tiles it is 8 KB per 48.
//   * A: the gathered low-res rows, fp32 -> registers -> hi / lo -> LDS as in gemm_wide_x3_kernel; a k-tile is 32 channels of one tap,
//     the source voxel of a thread's row is recomputed when the tap changes; the workgroup walks the UNION of its eight phases' taps;
//   * every wave is one phase (64 columns): it skips the MFMAs of the taps its phase does not reach (their weights are exact zeros)
//     but keeps loading B fragments so that the rotating register sets stay in step;
//   * epilogue: bias, LeakyReLU, depth-to-space store through the phase permutation (ops.polyphase_structure).
// Same products in the same order as the kernel it replaces (and as the dense evaluation): bit-identical.
struct PwArgs {
    const float* z;          // [B, S^3, Cin] fp32
    const u16* Bfrag;        // [ceil(N / 128) * 4][K / 16][2][64][8]
    const float* bias;       // [N] or nullptr
    float* out;              // fine grid [B, (S s)^3, 64]
    const int* perm;         // column block p holds phase perm[p]
    const int* blockmask;    // tap mask of column block p (bit t set <=> its phase has weights at tap t)
    int B, S, Cin, kext, off, replicate;
    int N, s;
    int act;
    float slope;
    int dbg;                 // vxb_debug_set_gemm_wide_experiment bit 64: the epilogue of rounds 4 - 5 (stores straight out of the accumulators)
};

__global__ void __launch_bounds__(512) conv_poly_wide_x3_kernel(PwArgs g) {
    __shared__ __attribute__((aligned(16))) u16 As[2][2][WBM * WLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * WBM;
    const int cg = blockIdx.y;
    const int M = g.B * g.S * g.S * g.S;
    const int K = g.kext * g.kext * g.kext * g.Cin;
    const int nks = K >> 4, cpt = g.Cin >> 5;                                   // k-steps of the whole K; 32-channel tiles per tap
    const int nblk = g.N >> 6;                                                  // 64-column blocks (phases)
    const int blk = cg * 8 + wn;                                                // this wave's block
    const bool wave_on = blk < nblk;
    const unsigned wmask = (unsigned)__builtin_amdgcn_readfirstlane(wave_on ? g.blockmask[blk] : 0);
    unsigned umask = 0;                                                         // union over the workgroup's blocks
    for (int w = 0; w < 8; ++w) if (cg * 8 + w < nblk) umask |= (unsigned)g.blockmask[cg * 8 + w];
    umask = (unsigned)__builtin_amdgcn_readfirstlane((int)umask);
    const int nkt = __builtin_popcount(umask) * cpt;

    // ---- A: thread -> row tid / 4 (its low-res voxel), 8 consecutive channels of the current 32-channel tile
    const int ar = tid >> 2, akq = (tid & 3) * 8;
    int aw, ah_, ad, ab;
    {
        int r = min(m0 + ar, M - 1);
        aw = r % g.S; r /= g.S;
        ah_ = r % g.S; r /= g.S;
        ad = r % g.S; r /= g.S;
        ab = r;
    }
    // two walkers over the union's taps in ascending order: the A loads run 4 tiles ahead of the MFMAs, the B loads one tile
    unsigned a_rem = umask, b_rem = umask;
    int a_cc = 0, a_tap = 0, b_cc = 0, b_tap = 0;
    long long a_src = -1;                                                       // element offset of the row's source voxel (-1: zero padding)
    float4 ra[3][2];
#define PW_LOADA(S_)                                                                                                  \
    {                                                                                                                \
        if (a_cc == 0) {                                                                                             \
            a_tap = a_rem ? __builtin_ctz(a_rem) : a_tap;                                                            \
            a_rem &= a_rem - 1;                                                                                      \
            const int tw_ = a_tap % g.kext, th_ = (a_tap / g.kext) % g.kext, td_ = a_tap / (g.kext * g.kext);        \
            int id_ = ad + td_ + g.off, ih_ = ah_ + th_ + g.off, iw_ = aw + tw_ + g.off;                             \
            bool ok_ = true;                                                                                         \
            if (g.replicate) { id_ = min(max(id_, 0), g.S - 1); ih_ = min(max(ih_, 0), g.S - 1); iw_ = min(max(iw_, 0), g.S - 1); } \
            else ok_ = id_ >= 0 && id_ < g.S && ih_ >= 0 && ih_ < g.S && iw_ >= 0 && iw_ < g.S;                      \
            a_src = ok_ ? ((((long long)ab * g.S + id_) * g.S + ih_) * g.S + iw_) * g.Cin + akq : -1;                \
        }                                                                                                            \
        const float* p_ = g.z + max(a_src, 0LL) + a_cc * 32;                                                         \
        ra[S_][0] = *reinterpret_cast<const float4*>(p_);                                                            \
        ra[S_][1] = *reinterpret_cast<const float4*>(p_ + 4);                                                        \
        if (a_src < 0) { ra[S_][0] = make_float4(0.f, 0.f, 0.f, 0.f); ra[S_][1] = ra[S_][0]; }                       \
        if (++a_cc == cpt) a_cc = 0;                                                                                 \
    }
#define PW_STOREA(S_, stage_)                                                                                         \
    {                                                                                                                \
        const float v_[8] = {ra[S_][0].x, ra[S_][0].y, ra[S_][0].z, ra[S_][0].w, ra[S_][1].x, ra[S_][1].y, ra[S_][1].z, ra[S_][1].w}; \
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
    // ---- B fragments of this wave's two 32-column tiles (block blk = column tiles 2 blk, 2 blk + 1); idle waves read block 0
    const u16* __restrict__ bfb = g.Bfrag + ((long long)((wave_on ? blk : 0) * 2) * nks * 2) * 512 + lane * 8;
    bf16x8 bq[4][2][2];          // the two k-steps of a tile: sets {0, 1} and {2, 3} alternate per tile
    int b_ks0 = 0;               // first k-step (in W's k order) of the tile whose fragments were loaded last
    const long long jstride = (long long)nks * 1024;         // u16 between the fragments of the wave's two 32-column tiles
    // loads the fragments of the NEXT tile of the walk into sets (2 P, 2 P + 1)
#define PW_LOADB(P_)                                                                                                  \
    {                                                                                                                \
        if (b_cc == 0) { b_tap = b_rem ? __builtin_ctz(b_rem) : b_tap; b_rem &= b_rem - 1; }                         \
        b_ks0 = b_tap * (g.Cin >> 4) + b_cc * 2;                                                                     \
        const u16* q0_ = bfb + (long long)b_ks0 * 1024;            /* one 64-bit add per tile; (h, p) are immediates */ \
        const u16* q1_ = q0_ + jstride;                                                                              \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                 \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                               \
            bq[2 * (P_) + h][0][p] = *reinterpret_cast<const bf16x8*>(q0_ + (h * 2 + p) * 512);                      \
            bq[2 * (P_) + h][1][p] = *reinterpret_cast<const bf16x8*>(q1_ + (h * 2 + p) * 512);                      \
        }                                                                                                            \
        if (++b_cc == cpt) b_cc = 0;                                                                                 \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lm = lane & 31, lk = (lane >> 5) * 8;
#define PW_STEP(st_, kk_, SB)                                                                                         \
    {                                                                                                                \
        _Pragma("unroll") for (int ip = 0; ip < 2; ++ip) {                                                            \
            bf16x8 ah2_[2], al2_[2];                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
                ah2_[i] = *reinterpret_cast<const bf16x8*>(&As[(st_)][0][((2 * ip + i) * 32 + lm) * WLD + (kk_) + lk]); \
                al2_[i] = *reinterpret_cast<const bf16x8*>(&As[(st_)][1][((2 * ip + i) * 32 + lm) * WLD + (kk_) + lk]); \
            }                                                                                                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[2 * ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al2_[i], bq[SB][j][0], acc[2 * ip + i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[2 * ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah2_[i], bq[SB][j][1], acc[2 * ip + i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[2 * ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah2_[i], bq[SB][j][0], acc[2 * ip + i][j], 0, 0, 0); \
        }                                                                                                            \
    }
    // the tap of the tile being multiplied: a third walker, advanced once per tile (wave-uniform)
    unsigned c_rem = umask;
    int c_cc = 0, c_tap = 0;
    // one k-tile (kt_ % 3 == R_, kt_ % 2 == P_): as GW_TILE, with the B fragments of tile kt_ + 1 loaded in one go (both k-steps)
#define PW_TILE(kt_, R_, P_)                                                                                          \
    {                                                                                                                \
        const int st = (P_);                                                                                         \
        if (c_cc == 0) { c_tap = c_rem ? __builtin_ctz(c_rem) : c_tap; c_rem &= c_rem - 1; }                         \
        if (++c_cc == cpt) c_cc = 0;                                                                                 \
        const bool mine_ = (wmask >> c_tap) & 1u;                                                                    \
        PW_STOREA((R_ + 1) % 3, st ^ 1)                                                                              \
        PW_LOADB(1 - (P_))                                                                                           \
        PW_LOADA((R_ + 1) % 3)                                                                                       \
        if (mine_) {                                                                                                 \
            PW_STEP(st, 0, 2 * (P_))                                                                                 \
            PW_STEP(st, 16, 2 * (P_) + 1)                                                                            \
        }                                                                                                            \
        __syncthreads();                                                                                             \
    }

    // prologue: tile 0 -> LDS stage 0, tiles 1..3 in flight, B fragments of tile 0
    PW_LOADA(0)
    PW_STOREA(0, 0)
    PW_LOADB(0)
    PW_LOADA(1)
    PW_LOADA(2)
    PW_LOADA(0)
    __syncthreads();
    // (whole rounds without branches around the tiles, then the remainder: see gemm_wide_kernel)
    int kt = 0;
#pragma unroll 1
    for (; kt + 6 <= nkt; kt += 6) {
        PW_TILE(kt, 0, 0)
        PW_TILE(kt + 1, 1, 1)
        PW_TILE(kt + 2, 2, 0)
        PW_TILE(kt + 3, 0, 1)
        PW_TILE(kt + 4, 1, 0)
        PW_TILE(kt + 5, 2, 1)
    }
    if (kt < nkt) {
        PW_TILE(kt, 0, 0)
        if (kt + 1 < nkt) PW_TILE(kt + 1, 1, 1)
        if (kt + 2 < nkt) PW_TILE(kt + 2, 2, 0)
        if (kt + 3 < nkt) PW_TILE(kt + 3, 0, 1)
        if (kt + 4 < nkt) PW_TILE(kt + 4, 1, 0)
    }
#undef PW_LOADA
#undef PW_STOREA
#undef PW_LOADB
#undef PW_STEP
#undef PW_TILE

    if (!wave_on) return;
    // ---- epilogue: column block blk is phase perm[blk] of the fine grid; row m is a low-res voxel
    const int sfac = g.s, ph = g.perm ? g.perm[blk] : blk;
    const int rw = ph % sfac, rh = (ph / sfac) % sfac, rd = ph / (sfac * sfac);
    const long long Vf = (long long)g.S * sfac;
    if (!(g.dbg & 64) && !((uintptr_t)g.out & 15) && !((uintptr_t)g.bias & 15)) {
        // voxel-contiguous stores (round 6; see gemm_wide_kernel's epilogue): a wave's 128 x 64 tile goes through its OWN 4 KB of the
        // (now free) operand stages, sixteen low-res voxels at a time, and leaves as whole 256-byte voxels, four per store instruction,
        // instead of two 128-byte halves of two voxels.  Wave-local: no workgroup barrier (waves that were switched off have left).
        float* Ws = reinterpret_cast<float*>(&As[0][0][0]) + wn * (16 * 64);
        const int er = lane >> 4, ec = (lane & 15) * 4;
        const float4 bs = g.bias ? *reinterpret_cast<const float4*>(g.bias + blk * 64 + ec) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = 8 * hf + rr;
                        Ws[((r & 3) + 8 * ((r >> 2) & 1) + 4 * (lane >> 5)) * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int row = er + 4 * st;
                    const int m = m0 + i * 32 + 16 * hf + row;
                    if (m >= M) continue;
                    int q = m;
                    const int qw = q % g.S; q /= g.S;
                    const int qh = q % g.S; q /= g.S;
                    const int qd = q % g.S; q /= g.S;
                    float4 v = *reinterpret_cast<const float4*>(&Ws[row * 64 + ec]);
                    v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
                    if (g.act == 1) {
                        v.x = v.x > 0.f ? v.x : v.x * g.slope; v.y = v.y > 0.f ? v.y : v.y * g.slope;
                        v.z = v.z > 0.f ? v.z : v.z * g.slope; v.w = v.w > 0.f ? v.w : v.w * g.slope;
                    }
                    *reinterpret_cast<float4*>(g.out + ((((long long)q * Vf + qd * sfac + rd) * Vf + qh * sfac + rh) * Vf + qw * sfac + rw) * 64 + ec) = v;
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= M) continue;
            int q = m;
            const int qw = q % g.S; q /= g.S;
            const int qh = q % g.S; q /= g.S;
            const int qd = q % g.S; q /= g.S;
            float* op = g.out + ((((long long)q * Vf + qd * sfac + rd) * Vf + qh * sfac + rh) * Vf + qw * sfac + rw) * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = j * 32 + (lane & 31);
                float v = acc[i][j][r] + (g.bias ? g.bias[blk * 64 + co] : 0.f);
                if (g.act == 1) v = v > 0.f ? v : v * g.slope;
                op[co] = v;
            }
        }
    }
}

int g_wide_waves = 8, g_wide_dbg = 0;
template <int X2>
void gw_launch(GwArgs& g, int M, int N, hipStream_t stream) {
    g.dbg = g_wide_dbg; g.N_total = N;
    if (g_wide_waves == 4) {
        g.ncg = N / 256;
        hipLaunchKernelGGL((gemm_wide_kernel<X2, 4>), dim3(vxb_cdiv(M, WBM) * g.ncg), dim3(256), 0, stream, g);
    } else if (N > 512 && !(g_wide_dbg & 32)) {
        // the column groups of a row block on adjacent workgroups of one XCD: A out of that XCD's L2, C rows written side by side
        // (4096 x 512: 0.55-0.60 -> 0.52-0.54 ms)
        g.ncg = N / 512;
        hipLaunchKernelGGL((gemm_wide_kernel<X2, 8>), dim3(vxb_cdiv(M, WBM) * g.ncg), dim3(512), 0, stream, g);
    } else {
        g.ncg = 0;
        hipLaunchKernelGGL((gemm_wide_kernel<X2, 8>), dim3(vxb_cdiv(M, WBM), N / 512), dim3(512), 0, stream, g);
    }
}

}  // namespace

extern "C" void vxb_debug_set_gemm_wide_waves(int waves) { g_wide_waves = waves == 4 ? 4 : 8; }
extern "C" void vxb_debug_set_gemm_wide_experiment(int bits) { g_wide_dbg = bits; }

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
    g.geglu = 0; g.F = 0; g.C2 = nullptr; g.H = nullptr; g.scale = nullptr;
    gw_launch<0>(g, M, N, (hipStream_t)stream);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// The same product with the result ALSO written as one fp16 plane [M][N] (saturating round-to-nearest: the bits vxb_split_f16_f32 makes of
// C) -- the k | v operand plane of the pipelined attention kernels out of to_kv's epilogue instead of by a pass over kv (round 6).
extern "C" int vxb_gemm_wide_bf16x3_f16out_f32(const float* A, int64_t lda, const void* Bw_frag, float* C, int64_t ldc, const float* bias,
                                               const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                                               void* f16_out, vxb_stream_t stream) {
    if (!A || !Bw_frag || !C || !f16_out || M < 1 || K < 64) return VXB_EARG;
    if (N < 512 || (N & 511) || (K & 31) || (lda & 3) || (((uintptr_t)A | (uintptr_t)Bw_frag | (uintptr_t)f16_out) & 15)) return VXB_ESIZE;
    GwArgs g;
    g.A = A; g.lda = lda; g.Bfrag = (const u16*)Bw_frag; g.C = C; g.ldc = ldc; g.bias = bias; g.residual = residual;
    g.M = M; g.K = K; g.act = act; g.slope = slope; g.accumulate = accumulate;
    g.geglu = 0; g.F = 0; g.C2 = nullptr; g.H = nullptr; g.scale = nullptr; g.aux16 = (u16*)f16_out;
    gw_launch<0>(g, M, N, (hipStream_t)stream);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// The same GEMM on TWO fp16 products per term ("fp16x2") for the DATA GRADIENTS of the linear layers (dX = dY @ W: A = dY, the weight
// operand = W^T): A * scale[0] / 16 is carried as an fp16 hi + lo pair, the weights as ONE fp16 value -- Bw_frag16: fragment order of the
// fp16 matrix [N][K], single plane ([N / 32][K / 16][64][8]; vxb_split_bf16_batch_f32 flag bit 3) -- and the sums are multiplied by
// 16 scale[1].  scale: device {2^k, 2^-k}, the operand scale of dY (vxb_absmax_scale_f32, or the delayed scale of the weight-gradient
// launch that read the same dY).  Rounding the weights of a data gradient to 11 bits moves none of the reference's gradient gates
// (tools/experiments/emu_precision.py --round5); two thirds of the MFMAs of vxb_gemm_wide_bf16x3_f32.
extern "C" int vxb_gemm_wide_f16x2_f32(const float* A, int64_t lda, const void* Bw_frag16, float* C, int64_t ldc, const float* residual,
                                       int M, int N, int K, int accumulate, const float* scale, vxb_stream_t stream) {
    if (!A || !Bw_frag16 || !C || !scale || M < 1 || K < 64) return VXB_EARG;
    if (N < 512 || (N & 511) || (K & 31) || (lda & 3) || (((uintptr_t)A | (uintptr_t)Bw_frag16) & 15)) return VXB_ESIZE;
    GwArgs g;
    g.A = A; g.lda = lda; g.Bfrag = (const u16*)Bw_frag16; g.C = C; g.ldc = ldc; g.bias = nullptr; g.residual = residual;
    g.M = M; g.K = K; g.act = 0; g.slope = 0.f; g.accumulate = accumulate;
    g.geglu = 0; g.F = 0; g.C2 = nullptr; g.H = nullptr; g.scale = scale;
    gw_launch<1>(g, M, N, (hipStream_t)stream);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// GEGLU up-projection (perceiver_lang_io.py:74-78 with :100-106): h [M][2 F] = A @ W^T + bias and gg [M][F] = h[:, :F] * gelu(h[:, F:])
// from one launch.  Bw_frag: the fragment order of W's planes with the rows of every 64-column block interleaved [32 value | 32 gate]
// (vxb_split_bf16_batch_f32 flag bit 2 / ops.gemm_wfrag_geglu).  2 F % 512 == 0.  Same bits as vxb_gemm_bf16x3_f32 + vxb_geglu_fwd_f32.
extern "C" int vxb_gemm_wide_geglu_fwd_f32(const float* A, int64_t lda, const void* Bw_frag, const float* bias, float* h, float* gg,
                                           int M, int F, int K, vxb_stream_t stream) {
    if (!A || !Bw_frag || !h || !gg || M < 1 || K < 64 || F < 256) return VXB_EARG;
    if ((F & 255) || (K & 31) || (lda & 3) || (((uintptr_t)A | (uintptr_t)Bw_frag) & 15)) return VXB_ESIZE;
    GwArgs g;
    g.A = A; g.lda = lda; g.Bfrag = (const u16*)Bw_frag; g.C = h; g.ldc = 2 * (long long)F; g.bias = bias; g.residual = nullptr;
    g.M = M; g.K = K; g.act = 0; g.slope = 0.f; g.accumulate = 0;
    g.geglu = 1; g.F = F; g.C2 = gg; g.H = nullptr; g.scale = nullptr;
    gw_launch<0>(g, M, 2 * F, (hipStream_t)stream);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// Data gradient of the down-projection fused with GEGLU's backward: dh [M][2 F] = geglu'(h) applied to d(gg) = dY [M][K] @ W2 [K][F]
// (Bw_frag: fragment order of the TRANSPOSED weight's planes [2][F][K], i.e. of the B operand of vxb_gemm_bf16x3_f32 for that data
// gradient); d(gg) never goes to HBM.  F % 512 == 0.  Same bits as vxb_gemm_bf16x3_f32 + vxb_geglu_bwd_f32.
extern "C" int vxb_gemm_wide_geglu_bwd_f32(const float* dY, int64_t lda, const void* Bw_frag, const float* h, float* dh, int M, int F,
                                           int K, vxb_stream_t stream) {
    if (!dY || !Bw_frag || !h || !dh || M < 1 || K < 64 || F < 512) return VXB_EARG;
    if ((F & 511) || (K & 31) || (lda & 3) || (((uintptr_t)dY | (uintptr_t)Bw_frag) & 15)) return VXB_ESIZE;
    GwArgs g;
    g.A = dY; g.lda = lda; g.Bfrag = (const u16*)Bw_frag; g.C = dh; g.ldc = 2 * (long long)F; g.bias = nullptr; g.residual = nullptr;
    g.M = M; g.K = K; g.act = 0; g.slope = 0.f; g.accumulate = 0;
    g.geglu = 2; g.F = F; g.C2 = nullptr; g.H = h; g.scale = nullptr;
    gw_launch<0>(g, M, F, (hipStream_t)stream);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// The same on two fp16 products (round 6): dY * scale[0] as an fp16 hi + lo pair, W2 as ONE fp16 plane in fragment order (the operand of
// vxb_gemm_wide_f16x2_f32: same products, same order -> d(gg) has its bits), GEGLU's backward in the row-contiguous epilogue.
extern "C" int vxb_gemm_wide_geglu_bwd_f16x2_f32(const float* dY, int64_t lda, const void* Bw_frag16, const float* h, float* dh, int M, int F,
                                                 int K, const float* scale, vxb_stream_t stream) {
    if (!dY || !Bw_frag16 || !h || !dh || !scale || M < 1 || K < 64 || F < 512) return VXB_EARG;
    if ((F & 511) || (K & 31) || (lda & 3) || (((uintptr_t)dY | (uintptr_t)Bw_frag16 | (uintptr_t)h | (uintptr_t)dh) & 15)) return VXB_ESIZE;
    GwArgs g;
    g.A = dY; g.lda = lda; g.Bfrag = (const u16*)Bw_frag16; g.C = dh; g.ldc = 2 * (long long)F; g.bias = nullptr; g.residual = nullptr;
    g.M = M; g.K = K; g.act = 0; g.slope = 0.f; g.accumulate = 0;
    g.geglu = 2; g.F = F; g.C2 = nullptr; g.H = h; g.scale = scale;
    gw_launch<1>(g, M, F, (hipStream_t)stream);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// Forward of the polyphase up-conv (see conv_poly_wide_x3_kernel): z [B, S^3, Cin] fp32 -> out fine grid [B, (S s)^3, 64] fp32 with
// out[.., phase perm[p], co] = act(sum_(tap, ci) z[clamp / zero-pad(m + tap + off)][ci] * W[p * 64 + co][(tap, ci)] + bias[p * 64 + co]).
// wt_frag: the [N][kext^3 * Cin] weights (column blocks already in `perm` order) as hi / lo bf16 planes in MFMA fragment order
// (ops.gemm_wfrag).  blockmask [N / 64]: bit t set <=> block p has non-zero weights at tap t (kext^3 <= 32).  N % 64 == 0, Cin % 32 == 0.
extern "C" int vxb_conv3_poly_wide_bf16x3_f32(const float* z, int Cin, int B, int S, int kext, int off, int replicate,
                                              const void* wt_frag, int N, const float* bias, float* out, int act, float slope,
                                              int d2s_s, const int32_t* blockmask, const int32_t* perm, vxb_stream_t stream) {
    if (!z || !wt_frag || !out || !blockmask || B < 1 || S < 1 || kext < 1 || N < 64 || d2s_s < 1) return VXB_EARG;
    if ((Cin & 31) || Cin < 32 || (N & 63) || kext * kext * kext > 32 || (((uintptr_t)z | (uintptr_t)wt_frag) & 15)) return VXB_ESIZE;
    const long long M = (long long)B * S * S * S;
    if (M >= INT32_MAX || M * Cin >= (1ll << 40)) return VXB_ESIZE;
    PwArgs g;
    g.z = z; g.Bfrag = (const u16*)wt_frag; g.bias = bias; g.out = out; g.perm = perm; g.blockmask = blockmask;
    g.B = B; g.S = S; g.Cin = Cin; g.kext = kext; g.off = off; g.replicate = replicate; g.N = N; g.s = d2s_s; g.act = act; g.slope = slope; g.dbg = g_wide_dbg;
    hipLaunchKernelGGL(conv_poly_wide_x3_kernel, dim3((unsigned)vxb_cdiv(M, WBM), vxb_cdiv(N, 512)), dim3(512), 0, (hipStream_t)stream, g);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
