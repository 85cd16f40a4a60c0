// 256 x 256-tile GEMM on the bf16 matrix cores for the big linear layers of the Perceiver latents ('bf16' / 'bf16x3'):
//     C[M][N] (+)= act(A[M][K] @ W[N][K]^T + bias) (+ residual)
// (nn.Linear forward and data gradient of to_q / to_kv / to_out / the GEGLU feed-forward, perceiver_lang_io.py:80-132, at
// M = B * 2048 latent rows.)
//
// gemm_dl.hip moves 128 x 128 tiles: every k-tile of 32 needs 16 KB + 16 KB of operands for 24 x 4 MFMAs, two workgroups per
// CU.  Here one workgroup of 8 waves owns a 256 x 256 tile -- a wave computes 128 x 64 of it (4 x 2 accumulator tiles of
// 32 x 32, 128 VGPRs) -- so every operand byte staged into LDS feeds twice as many MFMAs, and each fragment read from LDS is
// reused by 2 (A) or 4 (B) MFMA tiles of the wave instead of 2 / 2.
//   * both operands are bf16 planes in HBM (hi, and lo = bf16(x - hi) for 'bf16x3': products hi*hi + lo*hi + hi*lo), moved
//     global -> LDS by `global_load_lds_dwordx4` (no VGPRs, no conversion, no ds_write);
//   * LDS: two stages of [A hi | A lo | B hi | B lo] 256 x 32 tiles (64 KB per stage, 128 KB in all); a 16-byte slot
//     (row, chunk) sits at row * 4 + (chunk ^ ((row >> 2) & 3)) -- the XOR is applied on the SOURCE side of the direct load
//     (the destination of lane l is fixed at base + 16 l) and makes the 32-row fragment reads conflict-free;
//   * the direct loads of k-tile t + 1 are issued right after the barrier that opens k-tile t and have all of its
//     48 MFMAs per wave to land (one barrier and one s_waitcnt vmcnt(0) per k-tile).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int PTILE = BM * BK;                 // u16 per operand plane tile (16 KB)

struct G256Args {
    const u16* A;            // bf16 [planes][M][lda]
    long long a_plane, lda;
    const u16* W;            // bf16 [planes][N][K]
    long long w_plane;
    float* C;
    long long ldc;
    const float* bias;
    const float* residual;
    int M, N, K;
    int act;
    float slope;
    int accumulate;
};

__device__ __forceinline__ void g256_load16(const u16* src, u16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int X3>
__global__ void __launch_bounds__(512, 1) gemm256_kernel(G256Args g) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    constexpr int NPL = 1 + X3;
    constexpr int STAGE = 2 * NPL * PTILE;     // [A planes][B planes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;     // 2 x 4 waves: rows wm * 128, columns wn * 64
    // XCD-aware tile order: consecutive tiles of one XCD share their A rows (same tile_y) through that XCD's L2
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

    // direct-load slots: instruction (wid, i) fills rows (2 wid + i) * 16 .. + 15 of a plane tile; lane -> row + (lane >> 2),
    // destination chunk position lane & 3, i.e. source chunk (lane & 3) ^ ((row >> 2) & 3)
    long long a_off[2], w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * wid + i) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        a_off[i] = (long long)min(m0 + row, g.M - 1) * g.lda + chunk * 8;
        w_off[i] = (long long)min(n0 + row, g.N - 1) * g.K + chunk * 8;
    }
    auto issue = [&](int stage, int kt) {
        u16* sb = smem + stage * STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                g256_load16(g.A + p * g.a_plane + a_off[i] + k0, sb + p * PTILE + (2 * wid + i) * 512);
                g256_load16(g.W + p * g.w_plane + w_off[i] + k0, sb + (NPL + p) * PTILE + (2 * wid + i) * 512);
            }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment slots: lane (row = lane & 31, hi = lane >> 5) of a 32-row tile reads chunk 2 ks + hi of its row
    const int lm = lane & 31, hi = lane >> 5;
    const int fx = (lm >> 2) & 3;               // (row >> 2) & 3: tile bases are multiples of 32
    const int fa = (wm * 128 + lm) * BK, fb = (wn * 64 + lm) * BK;

    const int nkt = g.K / BK;
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of k-tile kt have landed in LDS
        __syncthreads();                                      // ... everyone's have; and everyone finished k-tile kt - 1
        if (kt + 1 < nkt) issue((kt + 1) & 1, kt + 1);
        const u16* sb = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int co = ((2 * ks + hi) ^ fx) * 8;
            bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(sb + NPL * PTILE + fb + j * 32 * BK + co);
                if (X3) bl[j] = *reinterpret_cast<const bf16x8*>(sb + (NPL + 1) * PTILE + fb + j * 32 * BK + co);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(sb + fa + i * 32 * BK + co);
                if (X3) al[i] = *reinterpret_cast<const bf16x8*>(sb + PTILE + fa + i * 32 * BK + co);
            }
            // term-major: consecutive MFMAs write different accumulators
            if (X3) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C layout of v_mfma_f32_32x32x16: register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    float* __restrict__ C = g.C;
    const float* __restrict__ R = g.residual;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + lm;
        if (n >= g.N) continue;
        const float bsv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
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

extern "C" int vxb_gemm256_f32(const void* A_planes, int64_t lda, const void* W_planes, int nplanes, float* C, int64_t ldc,
                               const float* bias, const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                               vxb_stream_t stream) {
    if (!A_planes || !W_planes || !C || M < 1 || N < 1 || K < BK || (K % BK) || (nplanes != 1 && nplanes != 2)) return VXB_EARG;
    if ((lda & 7) || (((uintptr_t)A_planes) & 15) || (((uintptr_t)W_planes) & 15)) return VXB_EARG;
    G256Args g;
    g.A = (const u16*)A_planes; g.a_plane = (long long)M * lda; g.lda = lda;
    g.W = (const u16*)W_planes; g.w_plane = (long long)N * K;
    g.C = C; g.ldc = ldc; g.bias = bias; g.residual = residual; g.M = M; g.N = N; g.K = K; g.act = act; g.slope = slope;
    g.accumulate = accumulate;
    const dim3 grid(vxb_cdiv(N, BN), vxb_cdiv(M, BM));
    const size_t lds = (size_t)2 * 2 * nplanes * PTILE * sizeof(u16);
    if (nplanes == 2) {
        static bool set = false;
        if (!set) {
            if (hipFuncSetAttribute((const void*)gemm256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return VXB_ELAUNCH;
            set = true;
        }
        hipLaunchKernelGGL(gemm256_kernel<1>, grid, dim3(512), lds, (hipStream_t)stream, g);
    } else {
        static bool set = false;
        if (!set) {
            if (hipFuncSetAttribute((const void*)gemm256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return VXB_ELAUNCH;
            set = true;
        }
        hipLaunchKernelGGL(gemm256_kernel<0>, grid, dim3(512), lds, (hipStream_t)stream, g);
    }
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
