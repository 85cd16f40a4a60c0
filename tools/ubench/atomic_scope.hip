// Probe: throughput of 1M returning integer atomics on random cells of a count grid, by memory scope and table size,
// and whether a workgroup's XCC_ID follows blockIdx % 8.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/atomic_scope.hip -o /tmp/atomic_scope && /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int SCOPE, int RET, int LOCAL>
__global__ void k_atomic(int* __restrict__ tab, unsigned cells, int n, int* __restrict__ sink) {
    // LOCAL: the table is split in 8 equal parts and a workgroup only touches the part of its XCD
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned c = rnd(i * 2654435761u + 12345u);
    if (LOCAL) { const unsigned part = cells >> 3; c = xcc * part + c % part; }
    else c %= cells;
    int r = 0;
    if (RET) r = __hip_atomic_fetch_add(tab + c, 1, __ATOMIC_RELAXED, SCOPE);
    else __hip_atomic_fetch_add(tab + c, 1, __ATOMIC_RELAXED, SCOPE);
    if (RET && r == 0x7fffffff) sink[0] = r;
}

__global__ void k_xcc(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;
}

__global__ void k_sum(const int* __restrict__ tab, unsigned cells, unsigned long long* out) {
    unsigned long long s = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) s += tab[i];
    atomicAdd(out, s);
}

template <int SCOPE, int RET, int LOCAL>
void run(const char* name, int* tab, unsigned cells, int n, int* sink, unsigned long long* dsum) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    unsigned long long total = 0;
    for (int it = 0; it < 5; ++it) {
        hipMemset(tab, 0, (size_t)cells * 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_atomic<SCOPE, RET, LOCAL>), dim3((n + 255) / 256), dim3(256), 0, 0, tab, cells, n, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemset(dsum, 0, 8);
        hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, 0, tab, cells, dsum);
        hipMemcpy(&total, dsum, 8, hipMemcpyDeviceToHost);
    }
    printf("%-44s cells %9u  %8.1f us  %6.2f atomics/ns  sum %llu (%s)\n", name, cells, best * 1e3f, n / (best * 1e6f), total,
           total == (unsigned long long)n ? "ok" : "LOST UPDATES");
}

int main() {
    const int n = 1 << 20;
    int* tab; int* sink; unsigned long long* dsum;
    hipMalloc(&tab, (size_t)(16 << 20) * 4); hipMalloc(&sink, 4); hipMalloc(&dsum, 8);
    int* x; hipMalloc(&x, 64 * 4);
    hipLaunchKernelGGL(k_xcc, dim3(64), dim3(64), 0, 0, x);
    std::vector<int> hx(64); hipMemcpy(hx.data(), x, 256, hipMemcpyDeviceToHost);
    printf("xcc of blocks 0..31:");
    for (int i = 0; i < 32; ++i) printf(" %d", hx[i]);
    printf("\n");
    for (unsigned cells : {16u << 20, 1u << 20, 1u << 16}) {
        run<__HIP_MEMORY_SCOPE_AGENT, 1, 0>("agent, returning", tab, cells, n, sink, dsum);
        run<__HIP_MEMORY_SCOPE_AGENT, 0, 0>("agent, no return", tab, cells, n, sink, dsum);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, 1, 0>("workgroup scope, returning (any XCD: unsafe)", tab, cells, n, sink, dsum);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, 1, 1>("workgroup scope, returning, XCD-local part", tab, cells, n, sink, dsum);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, 0, 1>("workgroup scope, no return, XCD-local part", tab, cells, n, sink, dsum);
        run<__HIP_MEMORY_SCOPE_AGENT, 1, 1>("agent, returning, XCD-local part", tab, cells, n, sink, dsum);
    }
    return 0;
}
