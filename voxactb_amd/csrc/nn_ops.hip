// Normalisation / activation / softmax / small-GEMM kernels of the PerceiverIO Q-function (fp32).
// Reference ops: PreNorm LayerNorm (perceiver_lang_io.py:56-71), GEGLU (:74-77), attention softmax +
// dropout (:124-128), LeakyReLU(0.02) of Conv3DBlock / DenseBlock (network_utils.py:12-27,166-170,285-289).
// All kernels are HBM-bound streaming passes: 16-byte accesses where the layout allows, one wave per
// row for the row-wise ops (64-lane shuffles, no LDS), deterministic two-stage column reductions.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// counter-based keep mask: same (seed, row, col) -> same decision in forward and backward
__device__ __forceinline__ bool keep_elem(unsigned seed, unsigned row, unsigned col, float p) {
    const unsigned h = hash32(hash32(row * 0x9E3779B1U + seed) ^ (col * 0x85EBCA77U + 0xC2B2AE3DU));
    return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}

// ------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; D <= 64*16
template <int VPL>   // values per lane
__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     long long rows, int D, float eps) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * D;
    float v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? xr[c] : 0.f;
        s += v[i];
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        const float d = c < D ? v[i] - mu : 0.f;
        q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    float* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < D) yr[c] = (v[i] - mu) * rs * gamma[c] + beta[c];
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx (+)= rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat)); per-block partial dgamma/dbeta -> part[blk][2][D]
template <int VPL>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                                     float* __restrict__ part, long long rows, int D,
                                                     int rows_per_block, int accumulate) {
    __shared__ float sg[4][VPL * 64], sb[4][VPL * 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float ag[VPL], ab[VPL], gm[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        ag[i] = 0.f; ab[i] = 0.f;
        const int c = lane + 64 * i;
        gm[i] = c < D ? gamma[c] : 0.f;
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    for (int rr = wid; rr < rows_per_block; rr += 4) {
        const long long row = r0 + rr;
        if (row >= rows) break;
        const float mu = mean[row], rs = rstd[row];
        float xh[VPL], dg[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            const float xv = c < D ? x[row * D + c] : 0.f;
            const float dv = c < D ? dy[row * D + c] : 0.f;
            xh[i] = c < D ? (xv - mu) * rs : 0.f;
            dg[i] = dv * gm[i];
            s1 += dg[i];
            s2 += dg[i] * xh[i];
            ag[i] += dv * xh[i];
            ab[i] += dv;
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float v = rs * (dg[i] - s1 - xh[i] * s2);
                if (accumulate) dx[row * D + c] += v; else dx[row * D + c] = v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) { sg[wid][lane + 64 * i] = ag[i]; sb[wid][lane + 64 * i] = ab[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        part[((long long)blockIdx.x * 2 + 0) * D + c] = sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c];
        part[((long long)blockIdx.x * 2 + 1) * D + c] = sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c];
    }
}

// float4 variant for D = 256 * NV (D = 512: the latent width): a lane owns 4 adjacent columns per 256-column group, and TWO rows
// of a wave are in flight (the scalar kernel keeps one row per wave between its load, its wave reduction and its store: 3 TB/s).
// Every column still adds its rows in the same order, so dgamma / dbeta are bit-identical to ln_bwd_kernel.
template <int NV>
__global__ void __launch_bounds__(256) ln_bwd4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, float* __restrict__ dx,
                                                      float* __restrict__ part, long long rows, int rows_per_block, int accumulate) {
    constexpr int D = 256 * NV;
    __shared__ float sg[4][D], sb[4][D];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float4 ag[NV], ab[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i];
        gm[i] = *reinterpret_cast<const float4*>(gamma + 256 * i + 4 * lane);
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    for (int rr = wid; rr < rows_per_block; rr += 8) {
        float4 xv[2][NV], dv[2][NV];
        float mu[2], rs[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long row = r0 + rr + 4 * u;
            ok[u] = rr + 4 * u < rows_per_block && row < rows;
            mu[u] = ok[u] ? mean[row] : 0.f; rs[u] = ok[u] ? rstd[row] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                xv[u][i] = ok[u] ? *reinterpret_cast<const float4*>(x + row * D + 256 * i + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
                dv[u][i] = ok[u] ? *reinterpret_cast<const float4*>(dy + row * D + 256 * i + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;                     // (wave-uniform)
            const long long row = r0 + rr + 4 * u;
            float xh[NV][4], dg[NV][4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float xs[4] = {xv[u][i].x, xv[u][i].y, xv[u][i].z, xv[u][i].w};
                const float ds[4] = {dv[u][i].x, dv[u][i].y, dv[u][i].z, dv[u][i].w};
                const float gs[4] = {gm[i].x, gm[i].y, gm[i].z, gm[i].w};
                float* agp = reinterpret_cast<float*>(&ag[i]);
                float* abp = reinterpret_cast<float*>(&ab[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xs[e] - mu[u]) * rs[u];
                    dg[i][e] = ds[e] * gs[e];
                    s1 += dg[i][e];
                    s2 += dg[i][e] * xh[i][e];
                    agp[e] += ds[e] * xh[i][e];
                    abp[e] += ds[e];
                }
            }
            s1 = wave_sum(s1) / (float)D;
            s2 = wave_sum(s2) / (float)D;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float4 v;
                v.x = rs[u] * (dg[i][0] - s1 - xh[i][0] * s2); v.y = rs[u] * (dg[i][1] - s1 - xh[i][1] * s2);
                v.z = rs[u] * (dg[i][2] - s1 - xh[i][2] * s2); v.w = rs[u] * (dg[i][3] - s1 - xh[i][3] * s2);
                float4* o = reinterpret_cast<float4*>(dx + row * D + 256 * i + 4 * lane);
                if (accumulate) { const float4 p = *o; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
                *o = v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        *reinterpret_cast<float4*>(&sg[wid][256 * i + 4 * lane]) = ag[i];
        *reinterpret_cast<float4*>(&sb[wid][256 * i + 4 * lane]) = ab[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        part[((long long)blockIdx.x * 2 + 0) * D + c] = sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c];
        part[((long long)blockIdx.x * 2 + 1) * D + c] = sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c];
    }
}

// dst[i] (+)= alpha * sum_s part[s][i]
__global__ void __launch_bounds__(256) sum_splits_kernel(const float* __restrict__ part, int nsplit, long long n,
                                                         float* __restrict__ dst, int accumulate, float alpha) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[(long long)k * n + i];
        s *= alpha;
        if (accumulate) dst[i] += s; else dst[i] = s;
    }
}

__global__ void __launch_bounds__(256) sum_splits_dev_kernel(const float* __restrict__ part, int nsplit, long long n,
                                                             float* __restrict__ dst, int accumulate, const float* __restrict__ alpha_p) {
    const float alpha = *alpha_p;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[(long long)k * n + i];
        s *= alpha;
        if (accumulate) dst[i] += s; else dst[i] = s;
    }
}

// max |x| as an integer maximum of the magnitude bits: NaN > inf > every finite value, so a non-finite element is not lost
__global__ void __launch_bounds__(256) absmax_part_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ part) {
    __shared__ unsigned red[4];
    unsigned m = 0;
    const long long n4 = n >> 2;
    const uint4* __restrict__ p = reinterpret_cast<const uint4*>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const uint4 v = p[i];
        m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(x[(n4 << 2) + threadIdx.x]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}

__global__ void __launch_bounds__(256) absmax_final_kernel(const unsigned* __restrict__ part, int nb, float* __restrict__ scale, int headroom) {
    __shared__ unsigned red[4];
    unsigned m = 0;
    for (int i = threadIdx.x; i < nb; i += 256) m = max(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        int k = 0;                                           // scale = 2^k
        if (m != 0 && m < 0x7f800000u) {
            const int e = (int)(m >> 23) - 127;              // m in [2^e, 2^(e+1)) (a denormal maximum: e = -127, clamped below)
            k = min(max(14 - headroom - e, -100), 100);   // headroom bits below [2^14, 2^15): for scales that are used one step late
        }
        scale[0] = __uint_as_float((unsigned)(k + 127) << 23);
        scale[1] = __uint_as_float((unsigned)(127 - k) << 23);
        // an inf / NaN in the tensor: both factors NaN, so every kernel that multiplies by them hands on NaN instead of the
        // saturated (finite) values its fp16 conversion would produce -- the reference's autograd propagates non-finite gradients
        if (m >= 0x7f800000u) { scale[0] = __uint_as_float(0x7fc00000u); scale[1] = __uint_as_float(0x7fc00000u); }
    }
}

// column sums of a DENSE [rows, N] matrix whose row length divides 1024 floats (N = 64 ... 1024, the bias gradients of
// the voxel-sized convs): the slab is streamed as one flat array with float4 loads -- thread t always lands on the same
// 4 columns, so it keeps 4 running sums and the block folds the 1024 / N threads of a column group at the end.
// Fixed summation order -> deterministic.
__global__ void __launch_bounds__(256) colsum_flat_kernel(const float* __restrict__ x, long long rows, int N,
                                                          int rows_per_block, float* __restrict__ part) {
    __shared__ float4 red[256];
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    const float4* __restrict__ p = reinterpret_cast<const float4*>(x + r0 * N);
    const long long n4 = (r1 - r0) * N / 4;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    long long i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) {
        const float4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
        s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
    }
    for (; i < n4; i += 256) { const float4 a = p[i]; s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
    s0.x += s1.x + (s2.x + s3.x); s0.y += s1.y + (s2.y + s3.y); s0.z += s1.z + (s2.z + s3.z); s0.w += s1.w + (s2.w + s3.w);
    red[threadIdx.x] = s0;
    __syncthreads();
    const int q = N / 4;                       // threads t, t + q, t + 2q, ... share columns 4 (t % q) .. +3
    if ((int)threadIdx.x < q) {
        float4 a = red[threadIdx.x];
        for (int j = threadIdx.x + q; j < 256; j += q) { const float4 b = red[j]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        *reinterpret_cast<float4*>(part + (long long)blockIdx.x * N + 4 * threadIdx.x) = a;
    }
}

// stage 2 of the column sums: out[c] (+)= sum over nb partial rows; 64 columns x 4 row lanes per block
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ part, int nb, int N, long long ld,
                                                           float* __restrict__ out, int accumulate,
                                                           float* __restrict__ out1, long long part1) {
    __shared__ float red[256];
    // grid.y = 2: a second set of partial rows (part + part1) summed into out1 by the same launch (LayerNorm: dgamma | dbeta)
    if (blockIdx.y == 1) { part += part1; out = out1; }
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        int r = rl;
        // sixteen rows of a thread in flight (with four, the 256 rows per thread of a 1024-row partial table -- LayerNorm's dgamma /
        // dbeta, 33 launches per step of 16 workgroups each -- were 64 dependent round trips: 24 us per launch)
        for (; r + 60 < nb; r += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(long long)(r + 4 * u) * ld + c];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
        }
        for (; r + 12 < nb; r += 16) {
            s0 += part[(long long)r * ld + c]; s1 += part[(long long)(r + 4) * ld + c];
            s2 += part[(long long)(r + 8) * ld + c]; s3 += part[(long long)(r + 12) * ld + c];
        }
        for (; r < nb; r += 4) s0 += part[(long long)r * ld + c];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < N) {
        const float s = (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
        out[c] = accumulate ? out[c] + s : s;
    }
}

// column sums, stage 1 (general N / row stride): part[blk][N] over a slab of rows
__global__ void __launch_bounds__(256) colsum_part_kernel(const float* __restrict__ x, long long rows, int N, long long ld,
                                                          int rows_per_block, float* __restrict__ part) {
    // 64 column lanes x 4 row lanes: a wave reads 256 contiguous bytes of one row
    __shared__ float red[256];
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    for (int c0 = 0; c0 < N; c0 += 64) {
        const int c = c0 + cl;
        float s0 = 0.f, s1 = 0.f;
        if (c < N) {
            long long r = r0 + rl;
            for (; r + 4 < r1; r += 8) { s0 += x[r * ld + c]; s1 += x[(r + 4) * ld + c]; }
            for (; r < r1; r += 4) s0 += x[r * ld + c];
        }
        red[threadIdx.x] = s0 + s1;
        __syncthreads();
        if (rl == 0 && c < N) part[(long long)blockIdx.x * N + c] = red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl];
        __syncthreads();
    }
}

// float4 variant for wide matrices (N % 4 == 0, 16-byte aligned rows): 64 column lanes x 4 columns x 4 row lanes -- a wave reads
// 1 KB of one row per instruction (the scalar kernel above moved 2.3 TB/s on the 32768 x 4096 GEGLU gradient)
__global__ void __launch_bounds__(256) colsum_part4_kernel(const float* __restrict__ x, long long rows, int N, long long ld,
                                                           int rows_per_block, float* __restrict__ part) {
    __shared__ float4 red[256];
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    for (int c0 = 0; c0 < N; c0 += 256) {
        const int c = c0 + 4 * cl;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        if (c < N) {
            long long r = r0 + rl;
            for (; r + 4 < r1; r += 8) {
                const float4 a = *reinterpret_cast<const float4*>(x + r * ld + c);
                const float4 b = *reinterpret_cast<const float4*>(x + (r + 4) * ld + c);
                s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
                s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            }
            for (; r < r1; r += 4) {
                const float4 a = *reinterpret_cast<const float4*>(x + r * ld + c);
                s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            }
        }
        red[threadIdx.x] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
        __syncthreads();
        if (rl == 0 && c < N) {
            const float4 a = red[cl], b = red[64 + cl], d = red[128 + cl], e = red[192 + cl];
            *reinterpret_cast<float4*>(part + (long long)blockIdx.x * N + c) =
                make_float4(a.x + b.x + d.x + e.x, a.y + b.y + d.y + e.y, a.z + b.z + d.z + e.z, a.w + b.w + d.w + e.w);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ softmax rows
// in place: S[row, :cols] -> P; optional second output Pd = P * keep / (1-p)
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ S, float* __restrict__ Pd, int cols,
                                                           long long ld, float p, unsigned seed) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    float* s = S + row * ld;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, s[c]);
    m = wave_max(m);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float e = expf(s[c] - m);
        s[c] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    const float keep_scale = 1.0f / (1.0f - p);
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float pv = s[c] * inv;
        s[c] = pv;
        if (Pd) Pd[row * ld + c] = keep_elem(seed, (unsigned)row, (unsigned)c, p) ? pv * keep_scale : 0.f;
    }
    // zero the row padding (ld is rounded up to 4 so that the GEMMs may read it with 16-byte loads)
    for (int c = cols + threadIdx.x; c < ld; c += 256) { s[c] = 0.f; if (Pd) Pd[row * ld + c] = 0.f; }
}

// ---- long rows (the V^3-wide translation softmax of act(), qattention_peract_bc_agent.py:705): one workgroup per row
// takes 1.9 ms for 10^6 columns at B = 1; here a row is cut into 8192-column chunks -- pass 1 leaves (max, sum exp) per
// chunk, pass 2 lets every chunk's workgroup combine the row's partials (fixed order) and normalise its own columns.
constexpr int SM_CHUNK = 8192;
__global__ void __launch_bounds__(256) softmax_long_part_kernel(const float* __restrict__ S, float2* __restrict__ part, int cols,
                                                                long long ld, int nchunk) {
    __shared__ float red[4];
    const int row = blockIdx.y, ch = blockIdx.x;
    const float* s = S + (long long)row * ld;
    const int c0 = ch * SM_CHUNK, c1 = min(cols, c0 + SM_CHUNK);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int c = c0 + threadIdx.x; c < c1; c += 256) m = fmaxf(m, s[c]);
    m = wave_max(m);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = c0 + threadIdx.x; c < c1; c += 256) sum += expf(s[c] - m);
    sum = wave_sum(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    if (threadIdx.x == 0) part[(long long)row * nchunk + ch] = make_float2(m, (red[0] + red[1]) + (red[2] + red[3]));
}
__global__ void __launch_bounds__(256) softmax_long_norm_kernel(float* __restrict__ S, const float2* __restrict__ part, int cols,
                                                                long long ld, int nchunk) {
    __shared__ float s_m, s_inv;
    const int row = blockIdx.y, ch = blockIdx.x;
    const float2* pr = part + (long long)row * nchunk;
    if (threadIdx.x < 64) {
        float m = -INFINITY;
        for (int i = threadIdx.x; i < nchunk; i += 64) m = fmaxf(m, pr[i].x);
        m = wave_max(m);
        float sum = 0.f;
        for (int i = threadIdx.x; i < nchunk; i += 64) sum += pr[i].y * expf(pr[i].x - m);
        sum = wave_sum(sum);
        if (threadIdx.x == 0) { s_m = m; s_inv = 1.0f / sum; }
    }
    __syncthreads();
    const float m = s_m, inv = s_inv;
    float* s = S + (long long)row * ld;
    const int c0 = ch * SM_CHUNK, c1 = min(cols, c0 + SM_CHUNK);
    for (int c = c0 + threadIdx.x; c < c1; c += 256) s[c] = expf(s[c] - m) * inv;
    if (ch == nchunk - 1)
        for (long long c = cols + threadIdx.x; c < ld; c += 256) s[c] = 0.f;
}

// dS = scale * P * (dP - sum(dP*P)),  dP = dPd * keep/(1-p); written over dPd
__global__ void __launch_bounds__(256) softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dPd, int cols,
                                                               long long ld, float scale, float p, unsigned seed) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const float* pr = P + row * ld;
    float* d = dPd + row * ld;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float ks = 1.0f / (1.0f - p);
    float acc = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        float g = d[c];
        if (p > 0.f) g = keep_elem(seed, (unsigned)row, (unsigned)c, p) ? g * ks : 0.f;
        acc += g * pr[c];
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wid] = acc;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    for (int c = threadIdx.x; c < cols; c += 256) {
        float g = d[c];
        if (p > 0.f) g = keep_elem(seed, (unsigned)row, (unsigned)c, p) ? g * ks : 0.f;
        d[c] = scale * pr[c] * (g - tot);
    }
    for (int c = cols + threadIdx.x; c < ld; c += 256) d[c] = 0.f;
}

// ------------------------------------------------------------------------------------------ GEGLU / lrelu
// (gelu_erf / gelu_erf_grad: common.h -- shared with the GEMM epilogues that fuse GEGLU, gemm_wide.hip)
__global__ void __launch_bounds__(256) geglu_fwd_kernel(const float* __restrict__ h, float* __restrict__ out, long long rows, int F) {
    const long long n = rows * F;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / F;
        const int c = (int)(i - r * F);
        out[i] = h[r * 2 * F + c] * gelu_erf(h[r * 2 * F + F + c]);
    }
}
__global__ void __launch_bounds__(256) geglu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dout,
                                                        float* __restrict__ dh, long long rows, int F) {
    const long long n = rows * F;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / F;
        const int c = (int)(i - r * F);
        const float a = h[r * 2 * F + c], g = h[r * 2 * F + F + c], d = dout[i];
        dh[r * 2 * F + c] = d * gelu_erf(g);
        dh[r * 2 * F + F + c] = d * a * gelu_erf_grad(g);
    }
}
__global__ void __launch_bounds__(256) lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        float* __restrict__ dx, long long n, float slope) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dx[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}
__global__ void __launch_bounds__(256) add_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n, float alpha) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] += alpha * src[i];
}

// ------------------------------------------------------------------------------------------ naive GEMM (tiny layers)
// C[m,n] (+)= act(sum_k A[m*sAm + k*sAk] * B[k*sBk + n*sBn] + bias[n])
__global__ void __launch_bounds__(256) naive_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                                         int K, long long sAm, long long sAk, long long sBk, long long sBn,
                                                         long long ldc, int act, float slope, int accumulate) {
    const long long total = (long long)M * N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(A[m * sAm + k * sAk], B[k * sBk + n * sBn], s);
        if (bias) s += bias[n];
        if (act == 1) s = s > 0.f ? s : s * slope;
        if (accumulate) C[m * ldc + n] += s; else C[m * ldc + n] = s;
    }
}

// the same for FEW outputs and a long reduction (the dense heads at a replay batch of 1-16: M N <= 16 K, K up to 4 K): one thread per
// output walks K alone and waits a memory round trip per step (28-33 us per launch); here a WAVE owns an output, its lanes stride
// over K and a fixed-order butterfly adds the 64 partial sums
__global__ void __launch_bounds__(256) naive_gemm_wave_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                                              int K, long long sAm, long long sAk, long long sBk, long long sBn,
                                                              long long ldc, int act, float slope, int accumulate) {
    const long long total = (long long)M * N;
    const int lane = threadIdx.x & 63;
    for (long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); i < total; i += (long long)gridDim.x * 4) {
        const int m = (int)(i / N), n = (int)(i % N);
        const float* __restrict__ a = A + m * sAm;
        const float* __restrict__ b = B + n * sBn;
        float s0 = 0.f, s1 = 0.f;
        int k = lane;
        for (; k + 64 < K; k += 128) {
            s0 = fmaf(a[k * sAk], b[k * sBk], s0);
            s1 = fmaf(a[(k + 64) * sAk], b[(k + 64) * sBk], s1);
        }
        if (k < K) s0 = fmaf(a[k * sAk], b[k * sBk], s0);
        float s = s0 + s1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            if (bias) s += bias[n];
            if (act == 1) s = s > 0.f ? s : s * slope;
            if (accumulate) C[m * ldc + n] += s; else C[m * ldc + n] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------ context assembly
// ctx[b, t, :] = (t < T0 ? lang[b,t,:] : cat(patch[b,t-T0,:C], pp[b,:Cp])) + pos[t,:]     (perceiver_lang_io.py:370-422;
// Cp = C for one proprio vector, 2 C for the right | left pair of the 2Robots encoder, :721-727)
__global__ void __launch_bounds__(256) ctx_build_kernel(const float* __restrict__ lang, const float* __restrict__ patch,
                                                        const float* __restrict__ pp, const float* __restrict__ pos,
                                                        float* __restrict__ ctx, int B, int T0, int T1, int C, int Cp) {
    const int Cx = C + Cp;
    const long long n = (long long)B * (T0 + T1) * Cx;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % Cx);
        const long long bt = i / Cx;
        const int t = (int)(bt % (T0 + T1));
        const int b = (int)(bt / (T0 + T1));
        float v;
        if (t < T0) v = lang[((long long)b * T0 + t) * Cx + c];
        else if (c < C) v = patch[((long long)b * T1 + (t - T0)) * C + c];
        else v = pp[b * Cp + (c - C)];
        ctx[i] = v + pos[(long long)t * Cx + c];
    }
}
// adjoint: dlang, dpatch written; dpos[t,c] = sum_b dctx; dpp[b,c] = sum_t dctx[b, T0+t, C+c]
__global__ void __launch_bounds__(256) ctx_bwd_kernel(const float* __restrict__ dctx, float* __restrict__ dlang,
                                                      float* __restrict__ dpatch, float* __restrict__ dpos, int B, int T0,
                                                      int T1, int C, int Cp) {
    const int Cx = C + Cp;
    const long long n = (long long)(T0 + T1) * Cx;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % Cx);
        const int t = (int)(i / Cx);
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            const float v = dctx[((long long)b * (T0 + T1) + t) * Cx + c];
            s += v;
            if (t < T0) dlang[((long long)b * T0 + t) * Cx + c] = v;
            else if (c < C) dpatch[((long long)b * T1 + (t - T0)) * C + c] = v;
        }
        dpos[i] += s;
    }
}
// dpp[b,c] = sum_t dctx[b, T0+t, C+c], c < Cp: stage 1 per (b, chunk) -> part[b][chunk][Cp]; stage 2 sums the chunks
__global__ void __launch_bounds__(256) ctx_bwd_pp_kernel(const float* __restrict__ dctx, float* __restrict__ part, int T0,
                                                         int T1, int C, int Cp, int nchunk) {
    __shared__ float red[256];
    const int b = blockIdx.x, ch = blockIdx.y;
    const int Cx = C + Cp;
    const int per = (T1 + nchunk - 1) / nchunk;
    const int t0 = ch * per, t1 = min(T1, t0 + per);
    const int nstripe = 256 / Cp;                // Cp in {64, 128}
    const int c = threadIdx.x % Cp, sidx = threadIdx.x / Cp;
    float s = 0.f;
    for (int t = t0 + sidx; t < t1; t += nstripe) s += dctx[((long long)b * (T0 + T1) + T0 + t) * Cx + C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < Cp) {
        float v = 0.f;
        for (int k = 0; k < nstripe; ++k) v += red[k * Cp + threadIdx.x];
        part[((long long)b * nchunk + ch) * Cp + threadIdx.x] = v;
    }
}
__global__ void __launch_bounds__(256) ctx_bwd_pp2_kernel(const float* __restrict__ part, float* __restrict__ dpp, int B, int C, int nchunk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[((long long)b * nchunk + k) * C + c];
    dpp[i] = s;
}


// ------------------------------------------------------------------------------------------ CLIP text transformer (act() path)
// Pieces of the language encoder the reference runs once per act() (helpers/clip/core/clip.py:426-440 encode_text_with_embeddings,
// :224-245 ResidualAttentionBlock): tiny shapes (n x 77 tokens x 512), latency not throughput -- the linear layers and
// LayerNorms reuse the kernels above, these three fill the gaps.
// out[r][:] = table[idx[r]][:] (+ pos[r % L][:]): token embedding + positional embedding, or a plain row gather (pos = nullptr)
__global__ void __launch_bounds__(256) embed_rows_kernel(const int* __restrict__ idx, const float* __restrict__ table,
                                                         const float* __restrict__ pos, float* __restrict__ out, long long rows,
                                                         int L, int D, long long table_rows) {
    const long long n = rows * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / D;
        const int c = (int)(i - r * D);
        long long t = idx[r];
        t = t < 0 ? 0 : (t >= table_rows ? table_rows - 1 : t);
        float v = table[t * D + c];
        if (pos) v += pos[(r % L) * D + c];
        out[i] = v;
    }
}
// QuickGELU (clip.py:219-221): x * sigmoid(1.702 x), in place
__global__ void __launch_bounds__(256) quick_gelu_kernel(float* __restrict__ x, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        x[i] = v / (1.0f + expf(-1.702f * v));
    }
}
// causal multi-head self-attention of a short sequence (nn.MultiheadAttention with the upper-triangular -inf mask,
// clip.py:396-402): qkv [n * L][3 * H * 64] = in_proj output (q | k | v, head h at columns 64 h), out [n * L][H * 64].
// One workgroup per (head, sequence); thread i owns query i: K and V of the head sit in LDS, scores of keys j <= i go
// through a running-max softmax in fp32.
__global__ void __launch_bounds__(128) attn_causal_small_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int H) {
    extern __shared__ float smem[];
    float* Ks = smem;                 // [L][65]
    float* Vs = smem + L * 65;        // [L][64]
    const int h = blockIdx.x, b = blockIdx.y, i = threadIdx.x;
    const int D = H * 64;
    const float* base = qkv + (long long)b * L * 3 * D;
    for (int e = threadIdx.x; e < L * 64; e += 128) {
        const int j = e >> 6, c = e & 63;
        Ks[j * 65 + c] = base[(long long)j * 3 * D + D + h * 64 + c];
        Vs[j * 64 + c] = base[(long long)j * 3 * D + 2 * D + h * 64 + c];
    }
    __syncthreads();
    if (i >= L) return;
    float q[64], acc[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) { q[c] = base[(long long)i * 3 * D + h * 64 + c] * 0.125f; acc[c] = 0.f; }     // q / sqrt(64) first (torch)
    float m = -INFINITY, ssum = 0.f;
    for (int j = 0; j <= i; ++j) {
        float sc = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c) sc = fmaf(q[c], Ks[j * 65 + c], sc);
        const float mn = fmaxf(m, sc);
        const float f = expf(m - mn), p = expf(sc - mn);
        ssum = ssum * f + p;
#pragma unroll
        for (int c = 0; c < 64; ++c) acc[c] = fmaf(p, Vs[j * 64 + c], acc[c] * f);
        m = mn;
    }
    const float inv = 1.0f / ssum;
    float* o = out + ((long long)b * L + i) * D + h * 64;
#pragma unroll
    for (int c = 0; c < 64; ++c) o[c] = acc[c] * inv;
}

inline int grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int vxb_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                     float* rstd, int64_t rows, int D, float eps, vxb_stream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 1 || D < 1) return VXB_EARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = vxb_cdiv(rows, 4);
    if (D <= 128) hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, D, eps);
    else if (D <= 512) hipLaunchKernelGGL(ln_fwd_kernel<8>, dim3(grid), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(ln_fwd_kernel<16>, dim3(grid), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, D, eps);
    else return VXB_ESIZE;
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// part_ws: at least 2*D*ceil(rows/64) floats.  dgamma/dbeta are ACCUMULATED (+=).
extern "C" int vxb_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                     const float* rstd, float* dx, float* dgamma, float* dbeta, float* part_ws,
                                     int64_t rows, int D, int accumulate_dx, vxb_stream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !part_ws || rows < 1 || D < 1) return VXB_EARG;
    hipStream_t st = (hipStream_t)stream;
    const int rpb = 32;        // 32 rows per workgroup: 1024 workgroups at 32768 rows (16 waves per CU keep ~130 KB in flight; with 64: 3.7 TB/s)
    const int grid = vxb_cdiv(rows, rpb);
    const bool al16 = ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dx) | ((uintptr_t)gamma)) & 15) == 0;
    if (D == 512 && al16) hipLaunchKernelGGL(ln_bwd4_kernel<2>, dim3(grid), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx, part_ws, rows, rpb, accumulate_dx);
    else if (D == 256 && al16) hipLaunchKernelGGL(ln_bwd4_kernel<1>, dim3(grid), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx, part_ws, rows, rpb, accumulate_dx);
    else if (D <= 128) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(grid), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx, part_ws, rows, D, rpb, accumulate_dx);
    else if (D <= 512) hipLaunchKernelGGL(ln_bwd_kernel<8>, dim3(grid), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx, part_ws, rows, D, rpb, accumulate_dx);
    else return VXB_ESIZE;
    // part layout [grid][2][D]: dgamma += column sums of the first half, dbeta += of the second (parallel over columns)
    hipLaunchKernelGGL(colsum_final_kernel, dim3(vxb_cdiv(D, 64), 2), dim3(256), 0, st, part_ws, grid, D, (long long)2 * D, dgamma, 1,
                       dbeta, (long long)D);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_sum_splits_f32(const float* part, int nsplit, int64_t n, float* dst, int accumulate, float alpha,
                                  vxb_stream_t stream) {
    if (!part || !dst || nsplit < 1 || n < 1) return VXB_EARG;
    if (nsplit >= 32 && n < (1 << 20) && alpha == 1.0f)   // many partials of a short vector: 4 row lanes x 4-way ILP per column
        hipLaunchKernelGGL(colsum_final_kernel, dim3(vxb_cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream, part, nsplit, (int)n,
                           (long long)n, dst, accumulate, (float*)nullptr, 0LL);
    else
        hipLaunchKernelGGL(sum_splits_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, part, nsplit, (long long)n, dst, accumulate, alpha);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_sum_splits_dev_f32(const float* part, int nsplit, int64_t n, float* dst, int accumulate, const float* alpha,
                                      vxb_stream_t stream) {
    if (!part || !dst || !alpha || nsplit < 1 || n < 1) return VXB_EARG;
    hipLaunchKernelGGL(sum_splits_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, part, nsplit, (long long)n, dst, accumulate, alpha);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// dst (+)= alpha[0] * sum_s part[s] (blocks 0 .. gridDim.x - 2) AND the next operand scale from the launch's per-workgroup maxima (the
// last block): the two finishing launches of a delayed-scaling fp16 weight gradient in one
__global__ void __launch_bounds__(256) wgrad_finish_kernel(const float* __restrict__ part, int nsplit, long long n, float* __restrict__ dst,
                                                           int accumulate, const float* __restrict__ alpha_p,
                                                           const unsigned* __restrict__ amax, int nb, float* __restrict__ scale, int headroom) {
    if (blockIdx.x == gridDim.x - 1) {
        __shared__ unsigned red[4];
        unsigned m = 0;
        for (int i = threadIdx.x; i < nb; i += 256) m = max(m, amax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = max(max(red[0], red[1]), max(red[2], red[3]));
            int k = 0;
            if (m != 0 && m < 0x7f800000u) {
                const int e = (int)(m >> 23) - 127;
                k = min(max(14 - headroom - e, -100), 100);
            }
            scale[0] = __uint_as_float((unsigned)(k + 127) << 23);
            scale[1] = __uint_as_float((unsigned)(127 - k) << 23);
            if (m >= 0x7f800000u) { scale[0] = __uint_as_float(0x7fc00000u); scale[1] = __uint_as_float(0x7fc00000u); }     // (see absmax_final_kernel)
        }
        return;
    }
    const float alpha = *alpha_p;
    const long long nblk = gridDim.x - 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += nblk * 256) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[(long long)k * n + i];
        s *= alpha;
        if (accumulate) dst[i] += s; else dst[i] = s;
    }
}

int vxb_wgrad_finish_launch(const float* part, int nsplit, long long n, float* dst, int accumulate, const float* alpha,
                            const unsigned* amax, int nb, float* scale, int headroom_bits, hipStream_t st) {
    if (!part || !dst || !alpha || !amax || !scale || nsplit < 1 || n < 1 || nb < 1) return VXB_EARG;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(grid_for(n) + 1), dim3(256), 0, st, part, nsplit, n, dst, accumulate, alpha, amax, nb, scale,
                       headroom_bits);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

int vxb_absmax_finish_launch(const unsigned* part, int n, float* scale, hipStream_t st, int headroom_bits) {
    if (!part || !scale || n < 1) return VXB_EARG;
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(256), 0, st, part, n, scale, headroom_bits);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_absmax_scale_f32(const float* x, int64_t n, float* ws, float* scale, vxb_stream_t stream) {
    if (!x || !ws || !scale || n < 1) return VXB_EARG;
    if ((uintptr_t)x & 15) return VXB_ESIZE;
    const int nb = (int)min((long long)1024, (long long)vxb_cdiv(n, 1024));
    hipLaunchKernelGGL(absmax_part_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, (long long)n, reinterpret_cast<unsigned*>(ws));
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned*>(ws), nb, scale, 0);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// out[N] (+)= column sums of x[rows, N] (row stride ld).  part_ws: ceil(rows/rpb)*N floats with rpb = max(64, rows/1024).
extern "C" int vxb_colsum_f32(const float* x, int64_t rows, int N, int64_t ld, float* part_ws, float* out, int accumulate,
                              vxb_stream_t stream) {
    if (!x || !part_ws || !out || rows < 1 || N < 1) return VXB_EARG;
    hipStream_t st = (hipStream_t)stream;
    long long rpb = rows / 1024;
    if (rpb < 64) rpb = 64;
    const int nb = vxb_cdiv(rows, rpb);
    if (ld == N && N >= 4 && N <= 1024 && 1024 % N == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL(colsum_flat_kernel, dim3(nb), dim3(256), 0, st, x, (long long)rows, N, (int)rpb, part_ws);
    else if ((N & 3) == 0 && (ld & 3) == 0 && N >= 256 && ((((uintptr_t)x) | ((uintptr_t)part_ws)) & 15) == 0)
        hipLaunchKernelGGL(colsum_part4_kernel, dim3(nb), dim3(256), 0, st, x, (long long)rows, N, (long long)ld, (int)rpb, part_ws);
    else
        hipLaunchKernelGGL(colsum_part_kernel, dim3(nb), dim3(256), 0, st, x, (long long)rows, N, (long long)ld, (int)rpb, part_ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(vxb_cdiv(N, 64)), dim3(256), 0, st, part_ws, nb, N, (long long)N, out, accumulate, (float*)nullptr, 0LL);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_softmax_rows_f32(float* S, float* P_drop, int64_t rows, int cols, int64_t ld, float dropout_p,
                                    uint32_t seed, vxb_stream_t stream) {
    if (!S || rows < 1 || cols < 1 || dropout_p < 0.f || dropout_p >= 1.f) return VXB_EARG;
    if (rows >= INT32_MAX) return VXB_ESIZE;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S,
                       dropout_p > 0.f ? P_drop : nullptr, cols, (long long)ld, dropout_p, seed);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// in-place softmax of a few very long rows (no dropout): ws = rows * ceil(cols / 8192) * 2 floats of scratch
extern "C" int vxb_softmax_long_rows_f32(float* S, float* ws, int64_t rows, int cols, int64_t ld, vxb_stream_t stream) {
    if (!S || !ws || rows < 1 || cols < 1 || ld < cols) return VXB_EARG;
    if (rows > 65535) return VXB_ESIZE;
    const int nchunk = (cols + SM_CHUNK - 1) / SM_CHUNK;
    hipLaunchKernelGGL(softmax_long_part_kernel, dim3(nchunk, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, (float2*)ws,
                       cols, (long long)ld, nchunk);
    hipLaunchKernelGGL(softmax_long_norm_kernel, dim3(nchunk, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, S,
                       (const float2*)ws, cols, (long long)ld, nchunk);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_softmax_bwd_rows_f32(const float* P, float* dP_inout, int64_t rows, int cols, int64_t ld, float scale,
                                        float dropout_p, uint32_t seed, vxb_stream_t stream) {
    if (!P || !dP_inout || rows < 1 || cols < 1) return VXB_EARG;
    if (rows >= INT32_MAX) return VXB_ESIZE;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, P, dP_inout, cols,
                       (long long)ld, scale, dropout_p, seed);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_geglu_fwd_f32(const float* h, float* out, int64_t rows, int F, vxb_stream_t stream) {
    if (!h || !out || rows < 1 || F < 1) return VXB_EARG;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(rows * F)), dim3(256), 0, (hipStream_t)stream, h, out, (long long)rows, F);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
extern "C" int vxb_geglu_bwd_f32(const float* h, const float* dout, float* dh, int64_t rows, int F, vxb_stream_t stream) {
    if (!h || !dout || !dh || rows < 1 || F < 1) return VXB_EARG;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * F)), dim3(256), 0, (hipStream_t)stream, h, dout, dh, (long long)rows, F);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
extern "C" int vxb_lrelu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, float slope, vxb_stream_t stream) {
    if (!dy || !y || !dx || n < 1) return VXB_EARG;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (long long)n, slope);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
extern "C" int vxb_axpy_f32(float* dst, const float* src, int64_t n, float alpha, vxb_stream_t stream) {
    if (!dst || !src || n < 1) return VXB_EARG;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, src, (long long)n, alpha);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
extern "C" int vxb_naive_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                                  int64_t sAm, int64_t sAk, int64_t sBk, int64_t sBn, int64_t ldc, int act, float slope,
                                  int accumulate, vxb_stream_t stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1) return VXB_EARG;
    if ((long long)M * N <= 16384 && K >= 128)
        hipLaunchKernelGGL(naive_gemm_wave_kernel, dim3((unsigned)(((long long)M * N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, A, B, C, bias,
                           M, N, K, (long long)sAm, (long long)sAk, (long long)sBk, (long long)sBn, (long long)ldc, act, slope, accumulate);
    else
        hipLaunchKernelGGL(naive_gemm_kernel, dim3(grid_for((long long)M * N)), dim3(256), 0, (hipStream_t)stream, A, B, C, bias, M, N, K,
                           (long long)sAm, (long long)sAk, (long long)sBk, (long long)sBn, (long long)ldc, act, slope, accumulate);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
extern "C" int vxb_ctx_build_f32(const float* lang, const float* patch, const float* pp, const float* pos, float* ctx, int B,
                                 int T0, int T1, int C, int Cp, vxb_stream_t stream) {
    if (!lang || !patch || !pp || !pos || !ctx || B < 1 || T0 < 0 || T1 < 1 || C < 1 || Cp < 1) return VXB_EARG;
    hipLaunchKernelGGL(ctx_build_kernel, dim3(grid_for((long long)B * (T0 + T1) * (C + Cp))), dim3(256), 0, (hipStream_t)stream, lang, patch, pp,
                       pos, ctx, B, T0, T1, C, Cp);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
// dlang [B,T0,C+Cp], dpatch [B,T1,C], dpp [B,Cp] are WRITTEN; dpos [T0+T1,C+Cp] is ACCUMULATED.
extern "C" int vxb_ctx_bwd_f32(const float* dctx, float* dlang, float* dpatch, float* dpp, float* dpos, float* part_ws, int B,
                               int T0, int T1, int C, int Cp, vxb_stream_t stream) {
    if (!dctx || !dlang || !dpatch || !dpp || !dpos || !part_ws || B < 1 || T1 < 1 || C < 1 || Cp < 1) return VXB_EARG;
    if (Cp > 256 || (256 % Cp)) return VXB_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = 32;     // part_ws: B*32*Cp floats
    hipLaunchKernelGGL(ctx_bwd_kernel, dim3(grid_for((long long)(T0 + T1) * (C + Cp))), dim3(256), 0, st, dctx, dlang, dpatch, dpos, B, T0, T1, C, Cp);
    hipLaunchKernelGGL(ctx_bwd_pp_kernel, dim3(B, nchunk), dim3(256), 0, st, dctx, part_ws, T0, T1, C, Cp, nchunk);
    hipLaunchKernelGGL(ctx_bwd_pp2_kernel, dim3(vxb_cdiv((long long)B * Cp, 256)), dim3(256), 0, st, part_ws, dpp, B, Cp, nchunk);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// out[r][:] = table[idx[r]][:] (+ pos[r % L][:] when pos != nullptr); idx is clamped to [0, table_rows).
extern "C" int vxb_embed_rows_f32(const int32_t* idx, const float* table, const float* pos, float* out, int64_t rows, int L, int D,
                                  int64_t table_rows, vxb_stream_t stream) {
    if (!idx || !table || !out || rows < 1 || L < 1 || D < 1 || table_rows < 1) return VXB_EARG;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream, idx, table, pos, out,
                       (long long)rows, L, D, (long long)table_rows);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
extern "C" int vxb_quick_gelu_f32(float* x, int64_t n, vxb_stream_t stream) {
    if (!x || n < 1) return VXB_EARG;
    hipLaunchKernelGGL(quick_gelu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
// causal self-attention of n sequences of L <= 128 tokens, H heads of 64: qkv [n * L][3 * H * 64] -> out [n * L][H * 64]
extern "C" int vxb_attn_causal_small_f32(const float* qkv, float* out, int n, int L, int H, vxb_stream_t stream) {
    if (!qkv || !out || n < 1 || L < 1 || H < 1) return VXB_EARG;
    if (L > 128) return VXB_ESIZE;
    const size_t lds = (size_t)L * (65 + 64) * sizeof(float);
    if (lds > 64 * 1024) return VXB_ESIZE;
    hipLaunchKernelGGL(attn_causal_small_kernel, dim3(H, n), dim3(128), lds, (hipStream_t)stream, qkv, out, L, H);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
