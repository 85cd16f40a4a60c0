// Store patterns of a GEMM epilogue: a 512-thread workgroup writes its 128 x 512 fp32 tile of C[32768][N].
//   A: the MFMA accumulator layout as it is (dword stores, one instruction = 2 rows x 128 B), x = row block fastest
//   B: row-contiguous float4 stores (one instruction = 1 KB of one row; what an LDS round trip would give)
//   C: A with the column groups of a row block on adjacent workgroups of one XCD
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tilestore.hip -o /tmp/tilestore && /tmp/tilestore
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ void __launch_bounds__(512) k_tile(float* C, int N, int ncg) {
    int bx = blockIdx.x, cg = blockIdx.y;
    if (PAT == 2) {
        const int total = gridDim.x;
        int vb = (bx & 7) * (total >> 3) + (bx >> 3);
        cg = vb % ncg; bx = vb / ncg;
    }
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const long long m0 = (long long)bx * 128;
    if (PAT == 1) {
        // wave wn: rows wn * 16 .. + 15; per row two instructions of 64 lanes x 16 B
        for (int r = 0; r < 16; ++r)
            for (int h = 0; h < 2; ++h) {
                f32x4 v = {1.f, 2.f, (float)r, (float)lane};
                *reinterpret_cast<f32x4*>(C + (m0 + wn * 16 + r) * N + cg * 512 + h * 256 + lane * 4) = v;
            }
    } else {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j)
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    C[m * N + cg * 512 + wn * 64 + j * 32 + (lane & 31)] = (float)(r + lane);
                }
    }
}
template <typename F> float timeit(F f, int it = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < it; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / it;
}
int main() {
    const int M = 32768;
    float* c; hipMalloc(&c, (size_t)M * 4096 * 4); hipMemset(c, 0, (size_t)M * 4096 * 4);
    for (int N : {512, 1024, 4096}) {
        const int ncg = N / 512; const double bytes = (double)M * N * 4;
        float ta = timeit([&] { hipLaunchKernelGGL(k_tile<0>, dim3(M / 128, ncg), dim3(512), 0, 0, c, N, ncg); });
        float tb = timeit([&] { hipLaunchKernelGGL(k_tile<1>, dim3(M / 128, ncg), dim3(512), 0, 0, c, N, ncg); });
        float tc = timeit([&] { hipLaunchKernelGGL(k_tile<2>, dim3(M / 128 * ncg, 1), dim3(512), 0, 0, c, N, ncg); });
        printf("N %4d (%.0f MB): A dword/acc layout %.1f us (%.2f TB/s) | B float4 rows %.1f (%.2f) | C = A + xcd remap %.1f (%.2f)\n", N, bytes / 1e6,
               ta * 1e3, bytes / ta / 1e9, tb * 1e3, bytes / tb / 1e9, tc * 1e3, bytes / tc / 1e9);
    }
    return 0;
}
