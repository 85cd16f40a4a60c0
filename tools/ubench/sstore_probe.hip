// Probe (round 6): do scalar stores (s_store_dwordx2 / x4 through the scalar data cache) work on gfx950, and are the bytes visible to a later
// kernel's vector AND scalar loads?  Each wave writes its ballot words with s_store, a second kernel reads them back both ways.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void wr(unsigned long long* p, const float* x, int n_words) {
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    unsigned long long* q = p + (size_t)wave * n_words;
    for (int i = 0; i < n_words; i += 2) {
        const float thr = 0.1f * (float)(i + 1);
        unsigned long long m0 = __builtin_amdgcn_ballot_w64(x[(blockIdx.x * blockDim.x + threadIdx.x) ^ i] > thr);
        unsigned long long m1 = __builtin_amdgcn_ballot_w64(x[(blockIdx.x * blockDim.x + threadIdx.x) ^ (i + 1)] > thr);
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u v = {(unsigned)m0, (unsigned)(m0 >> 32), (unsigned)m1, (unsigned)(m1 >> 32)};
        asm volatile("s_store_dwordx4 %0, %1, 0x0" ::"s"(v), "s"(q + i) : "memory");
    }
    asm volatile("s_dcache_wb" ::: "memory");
}
__global__ void rd(const unsigned long long* p, unsigned long long* outv, unsigned long long* outs, int n_words, int n_waves) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_waves) return;
    const unsigned long long* q = p + (size_t)__builtin_amdgcn_readfirstlane(wave) * n_words;
    unsigned long long sv = 0, ss = 0;
    for (int i = 0; i < n_words; ++i) {
        { const int w_ = (i + lane) % n_words; sv += p[(size_t)wave * n_words + w_] * (unsigned long long)(2 * w_ + 1); }      // vector loads
        ss += q[i] * (unsigned long long)(2 * i + 1);                                                    // uniform address: scalar loads
    }
    if (lane == 0) outs[wave] = ss;
    outv[wave * 64 + lane] = sv;
}
int main() {
    const int blocks = 2048, threads = 256, n_words = 32, n_waves = blocks * threads / 64;
    std::vector<float> hx(blocks * threads);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 250.0f;
    float* x; unsigned long long *p, *ov, *os;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&p, (size_t)n_waves * n_words * 8); hipMalloc(&ov, (size_t)n_waves * 64 * 8); hipMalloc(&os, (size_t)n_waves * 8);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemset(p, 0xff, (size_t)n_waves * n_words * 8);
    hipLaunchKernelGGL(wr, dim3(blocks), dim3(threads), 0, 0, p, x, n_words);
    hipLaunchKernelGGL(rd, dim3(blocks), dim3(threads), 0, 0, p, ov, os, n_words, n_waves);
    std::vector<unsigned long long> hp((size_t)n_waves * n_words), hos(n_waves), hov((size_t)n_waves * 64);
    hipMemcpy(hp.data(), p, hp.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hos.data(), os, hos.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hov.data(), ov, hov.size() * 8, hipMemcpyDeviceToHost);
    long bad = 0, bads = 0, badv = 0;
    for (int w = 0; w < n_waves; ++w) {
        unsigned long long ss = 0;
        for (int i = 0; i < n_words; ++i) {
            const float thr = 0.1f * (float)((i & ~1) + 1);
            unsigned long long m = 0;
            for (int l = 0; l < 64; ++l) if (hx[(size_t)(w * 64 + l) ^ i] > thr) m |= 1ull << l;
            if (hp[(size_t)w * n_words + i] != m) ++bad;
            ss += m * (unsigned long long)(2 * i + 1);
        }
        if (hos[w] != ss) ++bads;
        for (int l = 0; l < 64; ++l) if (hov[(size_t)w * 64 + l] != ss) ++badv;
    }
    printf("{\"scalar_store_words_wrong\": %ld, \"of\": %ld, \"scalar_load_sums_wrong\": %ld, \"vector_load_sums_wrong\": %ld}\n", bad, (long)hp.size(), bads, badv);
    return bad || bads || badv;
}
