// Probe: do v_mfma_f32_16x16x32_f16 and plain VALU instructions (v_fma_f32) overlap on a gfx950 SIMD --
//   (a) inside one wave (K independent VALU instructions issued after every MFMA),
//   (b) across the two waves of a SIMD (wave 0 MFMAs only, wave 1 VALU only)?
// Reports cycles per loop iteration per wave (s_memtime-free: wall time / iterations at the measured clock).
// build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-result tools/ubench/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE 0: MFMA only; 1: VALU only (K per iteration slot); 2: both interleaved in every wave; 3: even waves MFMA, odd waves VALU
// BIG: 0 = 16x16x32 (16 cycles), 1 = 32x32x16 (32 cycles)
template <int MODE, int K, int BIG>
__global__ void __launch_bounds__(512) k(float* out, int iters, unsigned long long* clk) {
    f32x4 acc[8];
    f32x16 accb[4];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x ^ e)); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.5f + 0.01f * e + 1e-3f * threadIdx.x;
    const float m = 0.999f;
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 4) == 0);      // waves 0-3 / 4-7 pair up on the four SIMDs
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 4) != 0);
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    // (the role of a wave is chosen OUTSIDE the loop: a branch per slot costs more than the slot)
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (BIG) accb[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accb[u & 3], 0, 0, 0);
                else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(u * K + j) & 7]) : "v"(m));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (BIG) accb[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accb[u & 3], 0, 0, 0);
                else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(u * K + j) & 7]) : "v"(m));
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += accb[i][r];
    for (int e = 0; e < 8; ++e) s += v[e];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int MODE, int K, int BIG>
void run(const char* what, float* o, unsigned long long* c) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;                     // one 8-wave workgroup per CU: two waves per SIMD
    hipLaunchKernelGGL((k<MODE, K, BIG>), dim3(blocks), dim3(512), 0, 0, o, 200, c);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, K, BIG>), dim3(blocks), dim3(512), 0, 0, o, iters, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    const double cyc = ms * 1e-3 * mhz * 1e6 / ((double)iters * 8);
    printf("%-58s K=%d  %7.3f ms  %6.1f cycles per slot (clock %.0f MHz)\n", what, K, ms, cyc, mhz);
}
int main() {
    float* o; unsigned long long* c; hipMalloc(&o, 4); hipMalloc(&c, 16);
    printf("two waves per SIMD; a slot = one MFMA and / or K v_fma_f32 per wave\n");
    run<0, 4, 0>("16x16x32 MFMA only (both waves)", o, c);
    run<1, 4, 0>("VALU only (both waves)", o, c);
    run<2, 4, 0>("16x16x32 MFMA + VALU interleaved in every wave", o, c);
    run<3, 4, 0>("16x16x32: one wave MFMA only, its SIMD partner VALU only", o, c);
    run<1, 8, 0>("VALU only (both waves)", o, c);
    run<2, 8, 0>("16x16x32 MFMA + VALU interleaved in every wave", o, c);
    run<3, 8, 0>("16x16x32: one wave MFMA only, its SIMD partner VALU only", o, c);
    run<0, 8, 1>("32x32x16 MFMA only (both waves)", o, c);
    run<2, 8, 1>("32x32x16 MFMA + VALU interleaved in every wave", o, c);
    run<3, 8, 1>("32x32x16: one wave MFMA only, its SIMD partner VALU only", o, c);
    run<2, 4, 1>("32x32x16 MFMA + VALU interleaved in every wave", o, c);
    run<2, 2, 1>("32x32x16 MFMA + VALU interleaved in every wave", o, c);
    return 0;
}
