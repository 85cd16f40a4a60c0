// Probe: LDS read throughput per CU of ds_read_b64_tr_b16 vs ds_read_b64 vs ds_read_b128 (conflict-free, lane-contiguous
// addresses), 8 waves per CU.  build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_tr_bw.hip -o tools/ubench/lds_tr_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned acc = 0;
    // every wave reads 8 distinct 1 KB (b128) / 512 B (b64) regions per iteration
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int base = ((wid * 8 + j) * 512 + ((it & 3) * 4096)) & 32767;      // u16 index
            if (MODE == 0) {
                s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + base + lane * 4));
                acc ^= (unsigned)v.x ^ (unsigned)v.w;
            } else if (MODE == 1) {
                uint2 v = *reinterpret_cast<const uint2*>(lds + base + lane * 4);
                acc ^= v.x ^ v.y;
            } else {
                uint4 v = *reinterpret_cast<const uint4*>(lds + ((base + lane * 8) & 32767));
                acc ^= v.x ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
void run(const char* name, int bytes_per_lane) {
    unsigned* o; hipMalloc(&o, 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, o, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = (double)iters * 8 * 8 /*waves*/ * 64 * bytes_per_lane;
    printf("%-22s %.3f ms  %.1f B/ns per CU (= B/clk at 1 GHz; divide by the clock in GHz)\n", name, ms, bytes_cu / (ms * 1e6));
}
int main() {
    run<0>("ds_read_b64_tr_b16", 8);
    run<1>("ds_read_b64", 8);
    run<2>("ds_read_b128", 16);
    return 0;
}
