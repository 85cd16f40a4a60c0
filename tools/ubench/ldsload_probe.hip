// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes?  (expect LDS[base + lane*16])
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/ldsload_probe.hip -o /tmp/ldsload_probe && /tmp/ldsload_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* __restrict__ in, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = 0xdeadbeef;
    __syncthreads();
    // lane l of wave w fetches element (w*64 + (63 - l)): a reversed pattern, to see that placement follows the LANE id
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + wid * 64 + (63 - lane)),
                                     (__attribute__((address_space(3))) void*)(lds + wid * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
    std::vector<uint4> h(256);
    for (unsigned i = 0; i < 256; ++i) h[i] = make_uint4(i, 1000 + i, 2000 + i, 3000 + i);
    uint4* d; unsigned* o;
    hipMalloc(&d, 256 * 16); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 256 * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        printf("wave %d:", w);
        for (int i = 0; i < 12; ++i) printf(" [%d]=%u,%u", i, r[w * 256 + i * 4], r[w * 256 + i * 4 + 1]);
        printf(" ... [63]=%u\n", r[w * 256 + 63 * 4]);
    }
    return 0;
}
