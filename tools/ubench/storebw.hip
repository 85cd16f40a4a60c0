// Store-bandwidth microbenchmark: what write rate can a pure 640 MB store stream reach on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_flat(f32x4* out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (; i < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            size_t j = i + (size_t)u * 256;
            if (j < n4) { if (NT) __builtin_nontemporal_store(v, out + j); else out[j] = v; }
        }
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) k_chunk(f32x4* out, size_t n4, size_t per_block) {
    // each block owns one contiguous chunk
    size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block; if (b1 > n4) b1 = n4;
    f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
}
template <int RB>
__global__ void __launch_bounds__(256) k_rows(float* out, int rows, int q4) {
    // row-aligned pattern of the voxel fill: thread j < q4 stores float4 j of RB consecutive rows
    const int j = threadIdx.x; if (j >= q4) return;
    f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int grp = blockIdx.x; grp * RB < rows; grp += gridDim.x) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) { int row = grp * RB + rr; if (row < rows) reinterpret_cast<f32x4*>(out + (size_t)row * q4 * 4)[j] = v; }
    }
}
__global__ void __launch_bounds__(256) k_copy(const f32x4* in, f32x4* out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_read(const f32x4* in, float* sink, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256;
    f32x4 a = {0,0,0,0};
    for (; i < n4; i += stride) a += in[i];
    if (a.x == 123.456f) sink[0] = a.y + a.z + a.w;
}
template <typename F> float timeit(F f, int it = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < it; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / it;
}
int main() {
    const size_t bytes = 640000000; const size_t n4 = bytes / 16;
    f32x4 *a, *b; float* sink; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        float t1 = timeit([&] { hipLaunchKernelGGL((k_flat<false, 1>), dim3(grid), dim3(256), 0, 0, a, n4); });
        float t2 = timeit([&] { hipLaunchKernelGGL((k_flat<true, 1>), dim3(grid), dim3(256), 0, 0, a, n4); });
        float t3 = timeit([&] { hipLaunchKernelGGL((k_flat<false, 4>), dim3(grid), dim3(256), 0, 0, a, n4); });
        float t4 = timeit([&] { hipLaunchKernelGGL((k_flat<true, 4>), dim3(grid), dim3(256), 0, 0, a, n4); });
        size_t pb = (n4 + grid - 1) / grid;
        float t5 = timeit([&] { hipLaunchKernelGGL((k_chunk<false>), dim3(grid), dim3(256), 0, 0, a, n4, pb); });
        float t6 = timeit([&] { hipLaunchKernelGGL((k_chunk<true>), dim3(grid), dim3(256), 0, 0, a, n4, pb); });
        printf("grid %6d: flat %.1f us (%.2f TB/s) | flat-nt %.1f (%.2f) | flat-u4 %.1f (%.2f) | flat-u4-nt %.1f (%.2f) | chunk %.1f (%.2f) | chunk-nt %.1f (%.2f)\n", grid,
               t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, bytes / t2 / 1e9, t3 * 1e3, bytes / t3 / 1e9, t4 * 1e3, bytes / t4 / 1e9, t5 * 1e3, bytes / t5 / 1e9, t6 * 1e3, bytes / t6 / 1e9);
    }
    for (int grid : {2048, 4096, 20000}) {
        float t8 = timeit([&] { hipLaunchKernelGGL((k_rows<8>), dim3(grid), dim3(256), 0, 0, (float*)a, 160000, 250); });
        float t1 = timeit([&] { hipLaunchKernelGGL((k_rows<1>), dim3(grid * 8), dim3(256), 0, 0, (float*)a, 160000, 250); });
        float t9 = timeit([&] { hipLaunchKernelGGL((k_rows<8>), dim3(grid), dim3(256), 0, 0, (float*)a, 156250, 256); });
        printf("rows grid %d: RB8 q4=250 %.1f us (%.2f TB/s) | RB1 %.1f (%.2f) | RB8 q4=256 (aligned) %.1f (%.2f)\n", grid, t8 * 1e3, bytes / t8 / 1e9, t1 * 1e3, bytes / t1 / 1e9, t9 * 1e3, bytes / t9 / 1e9);
    }
    float tm = timeit([&] { hipMemsetAsync(a, 0, bytes, 0); });
    printf("hipMemsetAsync: %.1f us (%.2f TB/s)\n", tm * 1e3, bytes / tm / 1e9);
    float tc = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, a, b, n4); });
    printf("copy (r+w 1.28 GB): %.1f us (%.2f TB/s total)\n", tc * 1e3, 2.0 * bytes / tc / 1e9);
    float tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, a, sink, n4); });
    printf("read 640 MB: %.1f us (%.2f TB/s)\n", tr * 1e3, bytes / tr / 1e9);
    return 0;
}
