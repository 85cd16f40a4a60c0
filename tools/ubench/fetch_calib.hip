// FETCH_SIZE calibration (round 6; round-5 review, item 5: "settle the FETCH_SIZE factor with a micro-benchmark: the same 64-byte-segment
// global_load pattern over a buffer of known size").  MI355X_MICROARCH.md: FETCH_SIZE reports HALF the bytes of a wide coalesced streaming
// read on gfx950 and is uncalibrated for other access widths.  Three kernels read a known number of bytes from a 2 GiB buffer (far beyond
// the 256 MiB Infinity Cache), every byte once per launch:
//   wide        : 16 B per lane, a wave covers 1 KB contiguous                                  (the guide's case: expect 0.5)
//   seg64       : the LDS-halo conv kernels' pattern -- 4 lanes x 16 B = one 64-byte segment (16 fp32 channels) of a 256-byte voxel row,
//                 a wave covers the same segment of 16 consecutive rows; one launch per segment index reads 1/4 of the buffer
//   seg64_rows4 : the same with the four segments of a row read by four CONSECUTIVE waves of one workgroup (what a workgroup of the conv
//                 does over its chunk loop, compressed in time): do neighbouring 64-byte halves pair up into one 128-byte request?
// Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv  and divide the counter (KiB) by the bytes each launch reads.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) wide(const float4* __restrict__ p, float* __restrict__ out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) seg64(const float4* __restrict__ p, float* __restrict__ out, size_t rows, int seg) {
    float s = 0.f;                          // thread t of the grid: row = t / 4, 16-byte piece t & 3 of segment `seg` of that 256-byte row
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < rows * 4; t += (size_t)gridDim.x * 256) {
        const float4 v = p[(t >> 2) * 16 + seg * 4 + (t & 3)];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) seg64_rows4(const float4* __restrict__ p, float* __restrict__ out, size_t rows) {
    float s = 0.f;                          // wave w of the workgroup reads segment w of 16 rows per iteration
    const int seg = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (size_t r0 = (size_t)blockIdx.x * 16; r0 < rows; r0 += (size_t)gridDim.x * 16) {
        const float4 v = p[(r0 + (l >> 2)) * 16 + seg * 4 + (l & 3)];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}
int main() {
    const size_t bytes = 2ull << 30, n4 = bytes / 16, rows = bytes / 256;
    float4* p; float* out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(p, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(wide, dim3(8192), dim3(256), 0, 0, p, out, n4);
        for (int seg = 0; seg < 4; ++seg) hipLaunchKernelGGL(seg64, dim3(8192), dim3(256), 0, 0, p, out, rows, seg);
        hipLaunchKernelGGL(seg64_rows4, dim3(8192), dim3(256), 0, 0, p, out, rows);
    }
    hipDeviceSynchronize();
    printf("{\"buffer_bytes\": %zu, \"wide_bytes_per_launch\": %zu, \"seg64_bytes_per_launch\": %zu, \"seg64_rows4_bytes_per_launch\": %zu}\n", bytes, bytes, bytes / 4, bytes);
    return 0;
}
