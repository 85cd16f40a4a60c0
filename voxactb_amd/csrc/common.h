// Shared device/host helpers for the gfx950 kernels (internal; the public C ABI is include/voxactb_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/voxactb_hip.h"

#define VXB_OK 0
#define VXB_EARG (-1)
#define VXB_ESIZE (-2)
#define VXB_EWS (-3)
#define VXB_ELAUNCH (-4)

#define VXB_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return VXB_ELAUNCH; \
    } while (0)

static inline int vxb_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// float4 variants of the Cout = 1 conv kernels (c1_conv.hip), used by the C entry points in vox_ops.hip when S % 4 == 0
int vxb_c1_fwd4_launch(const float* u, const float* w, const float* bias, float* q, int B, int S, hipStream_t st);
int vxb_c1_dgrad_ss_blocks_per_sample(int S);
int vxb_c1_dgrad4_ss_launch(const float* dq, const float* w, const float* u, float* du, int B, int S, int accumulate, float slope,
                            const float* lin, const float* stats, const float* out_ss, const int* argmax, const float* g_ss,
                            const float* g_max, float* dbias, float* part_ws, unsigned* part_amax, hipStream_t st);
// scale[0] = 2^k mapping the largest of n per-block |x| maxima (magnitude bits) into [2^14, 2^15), scale[1] = 1 / scale[0] (nn_ops.hip)
int vxb_absmax_finish_launch(const unsigned* part, int n, float* scale, hipStream_t st, int headroom_bits = 0);
// ... fused with the split-sum of the same launch's partial results: dst (+)= alpha[0] * sum_s part[s][0..n) (nn_ops.hip)
int vxb_wgrad_finish_launch(const float* part, int nsplit, long long n, float* dst, int accumulate, const float* alpha,
                            const unsigned* amax, int nb, float* scale, int headroom_bits, hipStream_t st);
// out[64] += column sums of part[nrows][64] in a fixed order; part must have room for 64 more rows behind the nrows (c1_conv.hip)
// dW [64][10] += scale[1] * sum_n part[n][c][0..9], db [64] += scale[1] * sum_n part[n][c][10] (part: n x [64][11]); patch_wgrad.hip
constexpr int VXB_WGIN_FINISH_ROWS = 512;
int vxb_wgin_finish_launch(float* part, int n, const float* scale, float* dW, float* db, hipStream_t st);
int vxb_rows64_sum_launch(float* part, int nrows, float* out, hipStream_t st);
// SpatialSoftmax3D + max: final merge of the per-tile partials [B][C][ntiles] written by the `final` conv's epilogue (vox_ops.hip)
int vxb_ss3d_final_tiles_launch(const float* part, int ntiles, int B, int C, float* out_ss, float* out_max, float* stats, int32_t* argmax,
                                hipStream_t st);
int vxb_c1_dgrad4_launch(const float* dq, const float* w, const float* u, float* du, int B, int S, int accumulate, int mask,
                         float slope, hipStream_t st);
int vxb_c1_wgrad4_launch(const float* u, const float* dq, float* dw, float* db, float* part_ws, int B, int S, hipStream_t st);

constexpr int WAVE = 64;

// two fp32 -> packed bf16 pair (low half = first value), round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 vxb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float vxb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned vxb_pack_bf16(float lo, float hi) {
    const vxb_f32x2 v = {lo, hi};
    union { vxb_bf16x2 b; unsigned u; } t;
    t.b = __builtin_convertvector(v, vxb_bf16x2);
    return t.u;
}

// two fp32 -> packed fp16 pair, round-to-nearest-even: one v_cvt_pk_f16_f32 on gfx950 (overflow -> inf: clamp first)
typedef _Float16 vxb_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned vxb_pack_f16(float lo, float hi) {
    const vxb_f32x2 v = {lo, hi};
    union { vxb_f16x2 h; unsigned u; } t;
    t.h = __builtin_convertvector(v, vxb_f16x2);
    return t.u;
}

// fp32 -> the fp16 range for a GRADIENT operand: finite values saturate at +-65504 (v_med3_f32), NaN and +-inf come out as NaN
// (x * 0 is NaN exactly for those; +-0 otherwise) -- v_med3_f32 alone maps NaN to -65504, i.e. a NaN born in the backward pass would
// become finite garbage the optimizer applies, where the reference's autograd propagates it (qattention_peract_bc_agent.py:578-590)
__device__ __forceinline__ float vxb_sat_f16(float x) { return fmaf(x, 0.0f, __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f)); }
// magnitude word of a per-workgroup |x| maximum for the operand-scale finish kernels (nn_ops.hip): NaN bits (larger than any finite
// magnitude as an unsigned integer) when `witness` -- any sum or product the kernel formed from the same values -- is NaN
__device__ __forceinline__ unsigned vxb_amax_word(float amxf, float witness) { return witness != witness ? 0x7fc00000u : __float_as_uint(amxf); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory (s_waitcnt vmcnt(0)), which
// drains every global load a wave has in flight -- fatal for kernels that keep operand prefetches (register loads or
// direct-to-LDS loads, tracked by hand with s_waitcnt vmcnt(n)) running across the barrier.
__device__ __forceinline__ void vxb_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Bare s_barrier for kernels whose LDS is written ONLY by direct-to-LDS loads that the code tracks itself with
// s_waitcnt vmcnt(n): even the LDS-only fence above waits for every outstanding direct load (they are LDS writes), i.e.
// for the tiles deliberately left in flight.  The asm clobbers keep the compiler from moving LDS reads across it.
// exact (erf) GELU and its derivative: GEGLU (perceiver_lang_io.py:74-78), as a stand-alone pass (nn_ops.hip) and in the GEMM
// epilogues that fuse it (gemm_wide.hip) -- one definition, so both give the same bits
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__device__ __forceinline__ void vxb_raw_barrier() {
    asm volatile("" ::: "memory");
#if defined(F2_ABLATE) && (F2_ABLATE & 2)
    return;
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// s_barrier once this wave's LDS reads and writes have completed; its vector-memory loads stay in flight (no vmcnt wait).  May sit
// in wave-uniform control flow as long as every wave of the workgroup executes the same NUMBER of barriers.
__device__ __forceinline__ void vxb_raw_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- SpatialSoftmax3D arithmetic shared by vox_ops.hip and c1_conv.hip
// x / T for the SpatialSoftmax3D temperature.  The reference divides (network_utils.py:801: feature / self.temperature); the
// IEEE division the compiler emits costs ~12 VALU instructions plus a scaling branch, which made the statistics pass
// VALU-bound (2.6 TB/s).  With rT = RN(1/T): q = RN(x rT), r = x - q T (exact, fma), RN(q + r rT) IS the correctly rounded
// quotient (Markstein) -- checked exhaustively on the CPU for T = 0.01f and every float with 2^-100 <= |x| < 2^119; outside
// that range the result is within 1 ulp of x / T (|x| < 2^-100: a difference of 2^-124 or less in the softmax exponent).
// The temperature is the network's constant 0.01 (network_utils.py:776).
struct DivT {
    float T, rT;
    __device__ __forceinline__ explicit DivT(float) : T(0.01f), rT(__fdiv_rn(1.0f, 0.01f)) {}      // (every caller passes 0.01f)
    __device__ __forceinline__ float operator()(float x) const {
        const float q = x * rT;
        const float r = fmaf(-q, T, x);
        return fmaf(r, rT, q);
    }
};

// e^d on v_exp_f32 with the rounding error of d * log2(e) carried along (<= ~1.5 ulp for |d| < 10^4; results below the normal
// range flush to zero -- those terms are < 2^-126 of the running sum)
__device__ __forceinline__ float exp_v(float d) {
    const float L2E = 1.44269502162933349609375f;           // float(log2 e)
    const float t = d * L2E;
    float r = fmaf(d, L2E, -t);
    r = fmaf(d, 1.92596299e-8f, r);                          // log2 e - float(log2 e)
    return __builtin_amdgcn_exp2f(t) * fmaf(r, 0.693147180559945f, 1.0f);
}

// Index of a tile inside an ntd x nth x ntw grid of tiles -> its coordinates, walking 4 x 4 x 4 BLOCKS of tiles (ragged at the far faces)
// instead of rows: the ~64 workgroups resident on one XCD then cover a compact block of the volume and their halo overlaps meet in that
// XCD's L2 (conv_halo_bf16.hip: halo_tile_coords; profiles/r05_final_conv_tile_order.log: FETCH_SIZE -25 %).  A bijection for any extents.
__device__ __forceinline__ void vxb_tile_block_coords(int t, int ntd, int nth, int ntw, int& td, int& th, int& tw) {
    constexpr int BL = 4;
    const int slab = BL * nth * ntw;                           // tiles in a full block layer along d
    const int sbd = min(t / slab, (ntd - 1) / BL);
    t -= sbd * slab;
    const int sd = min(BL, ntd - sbd * BL);
    const int rowsz = sd * BL * ntw;                           // ... in a full row of blocks along h inside that layer
    const int sbh = min(t / rowsz, (nth - 1) / BL);
    t -= sbh * rowsz;
    const int sh = min(BL, nth - sbh * BL);
    const int blksz = sd * sh * BL;
    const int sbw = min(t / blksz, (ntw - 1) / BL);
    t -= sbw * blksz;
    const int sw = min(BL, ntw - sbw * BL);
    const int lw = t % sw; t /= sw;
    const int lh = t % sh;
    const int ld = t / sh;
    td = sbd * BL + ld; th = sbh * BL + lh; tw = sbw * BL + lw;
}
