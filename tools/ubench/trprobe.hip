// Empirical semantics of ds_read_b64_tr_b16 on gfx950: every lane passes the byte address of 4 consecutive b16
// (lane L -> elements 4L..4L+3 of an LDS array holding its own index); print which element each lane gets back.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16;
__global__ void probe(u16* out, int mode) {
    __shared__ __attribute__((aligned(16))) u16 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (u16)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = (unsigned)(size_t)(&lds[0]) + lane * 8;                       // lane L -> elems 4L..4L+3
    else addr = (unsigned)(size_t)(&lds[0]) + ((lane & 15) * 64 + (lane >> 4) * 8);      // row (l&15) of a [16][32] b16 tile, col block (l>>4)*4
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (u16)(v >> (16 * j));
}
int main() {
    u16* d; hipMalloc(&d, 64 * 4 * 2);
    u16 h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
