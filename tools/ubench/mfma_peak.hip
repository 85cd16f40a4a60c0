// Probe: sustained v_mfma_f32_32x32x16_bf16 rate of the whole chip (what a perfect MFMA-bound kernel could reach under the
// power-managed clock) and the shader clock during the run.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x ^ e)); }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
    float* o; unsigned long long* c; hipMalloc(&o, 4); hipMalloc(&c, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = rep == 0 ? 2000 : 60000;
            const int blocks = 256 * waves_per_simd;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, iters, c);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            const double fl = (double)blocks * 4 * iters * 16 * 32768.0;
            printf("waves/SIMD %d  %8.3f ms  %7.1f TF/s  shader clock %.0f MHz (clock64 / wall_clock64 at 100 MHz)\n", waves_per_simd, ms,
                   fl / ms * 1e-9, (double)h[0] / (double)h[1] * 100.0);
        }
    }
    return 0;
}
