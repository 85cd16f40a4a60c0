// Voxel-grid-sized streaming kernels of the Q-function (fp32, channels-last [B, S^3, C]):
//   * 1x1x1 input conv (10 -> 64) + LeakyReLU and its weight gradient    (perceiver_lang_io.py:357)
//   * SpatialSoftmax3D + global max pool, forward and backward           (network_utils.py:773-809; perceiver :360,:451,:470)
//   * 3x3x3 conv with ONE output channel (trans_decoder), fwd / dgrad / wgrad   (perceiver :465)
//   * replicate-padding adjoint ("fold") used after every implicit-GEMM data gradient
//   * cross-entropy over 10^6 voxel logits + small heads, with argmax    (agent :57-80, :517-578)
//   * polyphase weight transform of upsample(x s, trilinear) o conv(k)   (network_utils.py:245-250) and its adjoint
//   * fused multi-tensor LAMB                                            (helpers/optim/lamb.py:60-124)
// All are HBM/L2-bound; none is reshaped into a GEMM.
#include "common.h"
#include "ss3d.h"

namespace {

inline int grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// =====================================================================================================
// pointwise conv  y[v, co] = lrelu(sum_ci x[v, ci] * W[co, ci] + b[co]),  Cin <= 16, Cout % 4 == 0, Cout <= 128
// =====================================================================================================
__global__ void __launch_bounds__(256) pw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     long long nvox, int Cin, int Cout, float slope) {
    __shared__ float sw[16 * 128];     // [ci][co]
    __shared__ float sb[128];
    for (int i = threadIdx.x; i < Cin * Cout; i += 256) {
        const int co = i / Cin, ci = i % Cin;
        sw[ci * Cout + co] = W[i];
    }
    for (int i = threadIdx.x; i < Cout; i += 256) sb[i] = bias[i];
    __syncthreads();
    const int q = Cout >> 2;
    const long long total = nvox * q;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / q;
        const int c4 = (int)(i - v * q) * 4;
        float a0 = sb[c4], a1 = sb[c4 + 1], a2 = sb[c4 + 2], a3 = sb[c4 + 3];
        const float* xr = x + v * Cin;
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = xr[ci];
            const float* wr = sw + ci * Cout + c4;
            a0 = fmaf(xv, wr[0], a0); a1 = fmaf(xv, wr[1], a1); a2 = fmaf(xv, wr[2], a2); a3 = fmaf(xv, wr[3], a3);
        }
        float4 o;
        o.x = a0 > 0.f ? a0 : a0 * slope; o.y = a1 > 0.f ? a1 : a1 * slope;
        o.z = a2 > 0.f ? a2 : a2 * slope; o.w = a3 > 0.f ? a3 : a3 * slope;
        *reinterpret_cast<float4*>(y + v * Cout + c4) = o;
    }
}

// partW[blk][co*Cin + ci], partB[blk][co] from dpre = dy * lrelu'(y); Cout == 64
__global__ void __launch_bounds__(256) pw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ dy, float* __restrict__ partW,
                                                       float* __restrict__ partB, long long nvox, int Cin, int vox_per_block,
                                                       float slope) {
    __shared__ float red[256 * 17];
    const int co = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float acc[16], accb = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const long long v0 = (long long)blockIdx.x * vox_per_block;
    const long long v1 = min(nvox, v0 + vox_per_block);
    for (long long v = v0 + grp; v < v1; v += 4) {
        const float yv = y[v * 64 + co];
        float d = dy[v * 64 + co];
        d = yv > 0.f ? d : d * slope;
        accb += d;
        const float* xr = x + v * Cin;
#pragma unroll
        for (int ci = 0; ci < 16; ++ci)
            if (ci < Cin) acc[ci] = fmaf(d, xr[ci], acc[ci]);
    }
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) red[threadIdx.x * 17 + ci] = acc[ci];
    red[threadIdx.x * 17 + 16] = accb;
    __syncthreads();
    if (threadIdx.x < 64) {
        for (int ci = 0; ci < Cin; ++ci) {
            float s = 0.f;
            for (int g = 0; g < 4; ++g) s += red[(g * 64 + threadIdx.x) * 17 + ci];
            partW[(long long)blockIdx.x * 64 * Cin + threadIdx.x * Cin + ci] = s;
        }
        float s = 0.f;
        for (int g = 0; g < 4; ++g) s += red[(g * 64 + threadIdx.x) * 17 + 16];
        partB[(long long)blockIdx.x * 64 + threadIdx.x] = s;
    }
}

// float4 variant: a thread owns 4 output channels (16 threads per voxel, 16 voxels per pass, two passes in flight)
__global__ void __launch_bounds__(256) pw_wgrad4_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ dy, float* __restrict__ partW,
                                                        float* __restrict__ partB, long long nvox, int Cin, int vox_per_block,
                                                        float slope) {
    __shared__ float red[4][64 * 17];
    const int c4 = (threadIdx.x & 15) * 4, gl = threadIdx.x >> 4, wid = threadIdx.x >> 6;
    float acc[4][16], accb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        accb[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[e][i] = 0.f;
    }
    const long long v0 = (long long)blockIdx.x * vox_per_block;
    const long long v1 = min(nvox, v0 + vox_per_block);
    for (long long vb = v0 + gl; vb < v1; vb += 32) {
        float4 yy[2], dd[2];
        float xr[2][16];
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const long long v = vb + 16 * uu;
            const bool ok = v < v1;
            yy[uu] = ok ? *reinterpret_cast<const float4*>(y + v * 64 + c4) : make_float4(1.f, 1.f, 1.f, 1.f);
            dd[uu] = ok ? *reinterpret_cast<const float4*>(dy + v * 64 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) xr[uu][ci] = (ok && ci < Cin) ? x[v * Cin + ci] : 0.f;
        }
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const float d[4] = {yy[uu].x > 0.f ? dd[uu].x : dd[uu].x * slope, yy[uu].y > 0.f ? dd[uu].y : dd[uu].y * slope,
                                yy[uu].z > 0.f ? dd[uu].z : dd[uu].z * slope, yy[uu].w > 0.f ? dd[uu].w : dd[uu].w * slope};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                accb[e] += d[e];
#pragma unroll
                for (int ci = 0; ci < 16; ++ci) acc[e][ci] = fmaf(d[e], xr[uu][ci], acc[e][ci]);
            }
        }
    }
    // fold the 4 voxel groups of a wave (lanes l, l^16, l^32, l^48), then the 4 waves through LDS
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int ci = 0; ci < 17; ++ci) {
            float v = ci < 16 ? acc[e][ci] : accb[e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if ((threadIdx.x & 63) < 16) red[wid][(c4 + e) * 17 + ci] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int co = threadIdx.x;
        for (int ci = 0; ci < Cin; ++ci)
            partW[(long long)blockIdx.x * 64 * Cin + co * Cin + ci] =
                (red[0][co * 17 + ci] + red[1][co * 17 + ci]) + (red[2][co * 17 + ci] + red[3][co * 17 + ci]);
        partB[(long long)blockIdx.x * 64 + co] = (red[0][co * 17 + 16] + red[1][co * 17 + 16]) + (red[2][co * 17 + 16] + red[3][co * 17 + 16]);
    }
}

// =====================================================================================================
// SpatialSoftmax3D (T = 0.01) + global max.   x: [B, S^3, C] (batch stride bs), C in {64, 128}
// stage 1: per (b, row-chunk) online-softmax partials part[b][chunk][c][7] = {m, s, sx, sy, sz, xmax, argmax}
// =====================================================================================================
// (SsPart, ss_merge: ss3d.h -- shared with the `final` conv's epilogue, conv_halo_bf16.hip)

__global__ void __launch_bounds__(256) ss_part_kernel(const float* __restrict__ x, long long bs, int S, int C, int ld,
                                                      const float* __restrict__ lin, int rows_per_chunk,
                                                      SsPart* __restrict__ part, int nchunk, float T) {
    __shared__ SsPart red[256];
    const DivT divT(T);
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int c = threadIdx.x % C, pl = threadIdx.x / C, npl = 256 / C;
    const float* xb = x + (long long)b * bs;
    SsPart a;
    a.m = -INFINITY; a.s = 0.f; a.sx = 0.f; a.sy = 0.f; a.sz = 0.f; a.xmax = -INFINITY; a.arg = 0x7fffffff;
    const int row0 = chunk * rows_per_chunk, row1 = min(S * S, row0 + rows_per_chunk);
    for (int row = row0; row < row1; ++row) {
        const int i = row / S, j = row - i * S;
        const float wy = lin[i], wx = lin[j];        // meshgrid 'xy' quirk: pos_x follows axis 1, pos_y axis 0
        for (int k = pl; k < S; k += npl) {
            const int p = row * S + k;
            const float xv = xb[(long long)p * ld + c];
            const float l = divT(xv);
            if (xv > a.xmax) { a.xmax = xv; a.arg = p; }
            if (l > a.m) {
                const float f = a.m > -INFINITY ? exp_v(a.m - l) : 0.f;
                a.s *= f; a.sx *= f; a.sy *= f; a.sz *= f;
                a.m = l;
            }
            const float e = exp_v(l - a.m);
            a.s += e; a.sx = fmaf(e, wx, a.sx); a.sy = fmaf(e, wy, a.sy); a.sz = fmaf(e, lin[k], a.sz);
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < C) {
        SsPart r = red[threadIdx.x];
        for (int g = 1; g < npl; ++g) ss_merge(r, red[g * C + threadIdx.x]);
        part[((long long)b * nchunk + chunk) * C + threadIdx.x] = r;
    }
}

// float4 variant of stage 1 (16-byte aligned rows): a thread owns 4 channels, C/4 threads cover a voxel, and four voxels
// per thread are loaded before any of them is consumed -- the scalar kernel keeps one 256-byte row per wave in flight,
// far too little to cover HBM latency on 256 CUs.
// One channel, four voxels (p0, p0 + dp, ...; bit u of `vm` = voxel u exists; voxel 0 always does): ONE rescale of the running
// sums to the new maximum, then four terms -- no data-dependent branch (the per-element "new maximum?" branch of the scalar
// kernel cost more scalar instructions than the arithmetic).
__device__ __forceinline__ void ss_update4(SsPart& a, float x0, float x1, float x2, float x3, int p0, int dp, unsigned vm, float wx,
                                           float wy, float wz0, float wz1, float wz2, float wz3, const DivT& divT) {
    const float xs[4] = {x0, x1, x2, x3};
    const float wz[4] = {wz0, wz1, wz2, wz3};
    float l[4];
    float mn = a.m;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool ok = (vm >> u) & 1u;
        l[u] = divT(xs[u]);
        mn = ok ? fmaxf(mn, l[u]) : mn;
        const bool gt = ok && xs[u] > a.xmax;
        a.xmax = gt ? xs[u] : a.xmax;
        a.arg = gt ? p0 + u * dp : a.arg;
    }
    const float f = a.m > -INFINITY ? exp_v(a.m - mn) : 0.f;
    a.s *= f; a.sx *= f; a.sy *= f; a.sz *= f;
    a.m = mn;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool ok = (vm >> u) & 1u;
        const float e = ok ? exp_v(l[u] - mn) : 0.f;
        a.s += e; a.sx = fmaf(e, wx, a.sx); a.sy = fmaf(e, wy, a.sy); a.sz = fmaf(e, wz[u], a.sz);
    }
}

__global__ void __launch_bounds__(256) ss_part4_kernel(const float* __restrict__ x, long long bs, int S, int C, int ld,
                                                       const float* __restrict__ lin, int rows_per_chunk,
                                                       SsPart* __restrict__ part, int nchunk, float T) {
    __shared__ SsPart red[256 * 4];
    const DivT divT(T);
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int q = C >> 2;
    const int cq = threadIdx.x % q, pl = threadIdx.x / q, npl = 256 / q;
    const float* xb = x + (long long)b * bs + 4 * cq;
    SsPart a[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e].m = -INFINITY; a[e].s = 0.f; a[e].sx = 0.f; a[e].sy = 0.f; a[e].sz = 0.f; a[e].xmax = -INFINITY; a[e].arg = 0x7fffffff; }
    const int row0 = chunk * rows_per_chunk, row1 = min(S * S, row0 + rows_per_chunk);
    for (int row = row0; row < row1; ++row) {
        const int i = row / S, j = row - i * S;
        const float wy = lin[i], wx = lin[j];        // meshgrid 'xy' quirk: pos_x follows axis 1, pos_y axis 0
        for (int k0 = pl; k0 < S; k0 += 4 * npl) {
            float4 v[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int k = k0 + uu * npl;
                v[uu] = k < S ? *reinterpret_cast<const float4*>(xb + (long long)(row * S + k) * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            unsigned vm = 0;
            float wz[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int k = k0 + uu * npl;
                vm |= (k < S ? 1u : 0u) << uu;
                wz[uu] = lin[k < S ? k : 0];
            }
            const int p0 = row * S + k0;
            ss_update4(a[0], v[0].x, v[1].x, v[2].x, v[3].x, p0, npl, vm, wx, wy, wz[0], wz[1], wz[2], wz[3], divT);
            ss_update4(a[1], v[0].y, v[1].y, v[2].y, v[3].y, p0, npl, vm, wx, wy, wz[0], wz[1], wz[2], wz[3], divT);
            ss_update4(a[2], v[0].z, v[1].z, v[2].z, v[3].z, p0, npl, vm, wx, wy, wz[0], wz[1], wz[2], wz[3], divT);
            ss_update4(a[3], v[0].w, v[1].w, v[2].w, v[3].w, p0, npl, vm, wx, wy, wz[0], wz[1], wz[2], wz[3], divT);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = a[e];
    __syncthreads();
    if ((int)threadIdx.x < C) {
        const int tq = threadIdx.x >> 2, e = threadIdx.x & 3;      // channel c = threadIdx.x = 4 tq + e
        SsPart r = red[tq * 4 + e];
        for (int g = 1; g < npl; ++g) ss_merge(r, red[(g * q + tq) * 4 + e]);
        part[((long long)b * nchunk + chunk) * C + threadIdx.x] = r;
    }
}

// The input conv and the statistics pass over its output in one kernel (perceiver :357 + :360): thread -> (voxel, 4 channels)
// exactly as in ss_part4_kernel (same chunks, same visiting order, so the partials are bit-identical to the two-kernel
// path), y = lrelu(W x + b) is stored and folded into the running softmax / max statistics while it is still in registers:
// the 256 B per voxel are written once and never read back (4.1 GB per step at B = 16, V = 100).  Cout == 64.
// A (d, h) row of x (S voxels x CIN floats, contiguous) is staged in LDS by the whole workgroup, one row ahead: the 16 threads
// that share a voxel read its CIN inputs as LDS broadcasts instead of 16 x CIN global loads.
constexpr int PWSS_MAX_S = 256;
template <int CIN>
__global__ void __launch_bounds__(256) pw_fwd_ss_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ y, int S,
                                                        float slope, const float* __restrict__ lin, int rows_per_chunk,
                                                        SsPart* __restrict__ part, int nchunk, float T) {
    __shared__ SsPart red[256 * 4];
    __shared__ float sx[2][PWSS_MAX_S * CIN];
    // the coordinate table in LDS: as global loads its reads queue BEHIND the next row's prefetch on the in-order vector-memory counter,
    // and every row waited for the prefetch it had just issued (round 5)
    __shared__ float slin[PWSS_MAX_S];
    for (int i = threadIdx.x; i < S; i += 256) slin[i] = lin[i];
    const DivT divT(T);
    constexpr int C = 64, q = 16, npl = 16;
    constexpr int NLD = (PWSS_MAX_S * CIN + 255) / 256;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cq = threadIdx.x % q, pl = threadIdx.x / q;
    const long long vox0 = (long long)b * S * S * S;
    const float* xb = x + vox0 * CIN;
    float* yb = y + vox0 * C + 4 * cq;
    float wreg[CIN][4];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
        for (int e = 0; e < 4; ++e) wreg[ci][e] = W[(4 * cq + e) * CIN + ci];
    }
    const float b0 = bias[4 * cq], b1 = bias[4 * cq + 1], b2 = bias[4 * cq + 2], b3 = bias[4 * cq + 3];
    SsPart a[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e].m = -INFINITY; a[e].s = 0.f; a[e].sx = 0.f; a[e].sy = 0.f; a[e].sz = 0.f; a[e].xmax = -INFINITY; a[e].arg = 0x7fffffff; }
    const int row0 = chunk * rows_per_chunk, row1 = min(S * S, row0 + rows_per_chunk);
    const int rowlen = S * CIN;
    float nxt[NLD];
    if (row0 < row1) {
        const float* xr = xb + (long long)row0 * rowlen;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < rowlen) sx[0][i] = xr[i];
        }
    }
    __syncthreads();
    int cur = 0;
    for (int row = row0; row < row1; ++row) {
        if (row + 1 < row1) {
            const float* xr = xb + (long long)(row + 1) * rowlen;
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int i = threadIdx.x + 256 * u;
                nxt[u] = i < rowlen ? xr[i] : 0.f;
            }
        }
        const int i = row / S, j = row - i * S;
        const float wy = slin[i], wx = slin[j];      // meshgrid 'xy' quirk: pos_x follows axis 1, pos_y axis 0
        const float* sxr = sx[cur];
        for (int k0 = pl; k0 < S; k0 += 4 * npl) {
            float4 o[4];
            float wz[4];
            unsigned vm = 0;
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int k = k0 + uu * npl;
                const bool ok = k < S;
                const int kc = ok ? k : 0;
                vm |= (ok ? 1u : 0u) << uu;
                wz[uu] = slin[kc];
                float a0 = b0, a1 = b1, a2 = b2, a3 = b3;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float xv = sxr[kc * CIN + ci];
                    a0 = fmaf(xv, wreg[ci][0], a0); a1 = fmaf(xv, wreg[ci][1], a1);
                    a2 = fmaf(xv, wreg[ci][2], a2); a3 = fmaf(xv, wreg[ci][3], a3);
                }
                o[uu].x = a0 > 0.f ? a0 : a0 * slope; o[uu].y = a1 > 0.f ? a1 : a1 * slope;
                o[uu].z = a2 > 0.f ? a2 : a2 * slope; o[uu].w = a3 > 0.f ? a3 : a3 * slope;
                if (ok) *reinterpret_cast<float4*>(yb + (long long)(row * S + k) * C) = o[uu];
            }
            const int p0 = row * S + k0;
            ss_update4(a[0], o[0].x, o[1].x, o[2].x, o[3].x, p0, npl, vm, wx, wy, wz[0], wz[1], wz[2], wz[3], divT);
            ss_update4(a[1], o[0].y, o[1].y, o[2].y, o[3].y, p0, npl, vm, wx, wy, wz[0], wz[1], wz[2], wz[3], divT);
            ss_update4(a[2], o[0].z, o[1].z, o[2].z, o[3].z, p0, npl, vm, wx, wy, wz[0], wz[1], wz[2], wz[3], divT);
            ss_update4(a[3], o[0].w, o[1].w, o[2].w, o[3].w, p0, npl, vm, wx, wy, wz[0], wz[1], wz[2], wz[3], divT);
        }
        if (row + 1 < row1) {
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int i2 = threadIdx.x + 256 * u;
                if (i2 < rowlen) sx[cur ^ 1][i2] = nxt[u];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = a[e];
    __syncthreads();
    if ((int)threadIdx.x < C) {
        const int tq = threadIdx.x >> 2, e = threadIdx.x & 3;      // channel c = threadIdx.x = 4 tq + e
        SsPart r = red[tq * 4 + e];
        for (int g = 1; g < npl; ++g) ss_merge(r, red[(g * q + tq) * 4 + e]);
        part[((long long)b * nchunk + chunk) * C + threadIdx.x] = r;
    }
}

// stage 2: combine chunks -> out_ss[b][3c + {x,y,z}], out_max[b][c], stats[b][c] = {m, s}, argmax[b][c]
__global__ void __launch_bounds__(256) ss_final_kernel(const SsPart* __restrict__ part, int nchunk, int B, int C, int Ct, int c0,
                                                       float* __restrict__ out_ss, float* __restrict__ out_max,
                                                       float* __restrict__ stats, int* __restrict__ argmax) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    SsPart r = part[((long long)b * nchunk) * C + c];
    for (int k = 1; k < nchunk; ++k) ss_merge(r, part[((long long)b * nchunk + k) * C + c]);
    const int o = b * Ct + c0 + c;                    // channel c of this slab is channel c0 + c of the Ct-wide tensor
    out_ss[3LL * o + 0] = r.sx / r.s;
    out_ss[3LL * o + 1] = r.sy / r.s;
    out_ss[3LL * o + 2] = r.sz / r.s;
    out_max[o] = r.xmax;
    stats[2 * o] = r.m;
    stats[2 * o + 1] = r.s;
    argmax[o] = r.arg;
}

// stage 2 for many chunks (small batches cut a sample into up to 1024 chunks): one workgroup per (b, c), threads merge a
// strided subset of the chunks, then a fixed-order tree over the 256 partial results
__global__ void __launch_bounds__(256) ss_final_wide_kernel(const SsPart* __restrict__ part, int nchunk, int B, int C, int Ct, int c0,
                                                            float* __restrict__ out_ss, float* __restrict__ out_max,
                                                            float* __restrict__ stats, int* __restrict__ argmax,
                                                            long long sb, long long sk, long long sc) {
    // partial (b, k, c) at part[b * sb + k * sk + c * sc]: [b][chunk][c] from the statistics kernels, [b][c][tile] from the conv epilogue
    __shared__ SsPart red[256];
    const int i = blockIdx.x, b = i / C, c = i % C;
    SsPart r;
    r.m = -INFINITY; r.s = 0.f; r.sx = 0.f; r.sy = 0.f; r.sz = 0.f; r.xmax = -INFINITY; r.arg = 0x7fffffff;
    for (int k = threadIdx.x; k < nchunk; k += 256) ss_merge(r, part[b * sb + k * sk + c * sc]);
    red[threadIdx.x] = r;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { SsPart a = red[threadIdx.x]; ss_merge(a, red[threadIdx.x + o]); red[threadIdx.x] = a; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        r = red[0];
        const int o = b * Ct + c0 + c;
        out_ss[3LL * o + 0] = r.sx / r.s;
        out_ss[3LL * o + 1] = r.sy / r.s;
        out_ss[3LL * o + 2] = r.sz / r.s;
        out_max[o] = r.xmax;
        stats[2 * o] = r.m;
        stats[2 * o + 1] = r.s;
        argmax[o] = r.arg;
    }
}

// float4 variant of the backward pass below (same formula per element; 4 channels per thread, 4 voxels in flight)
__global__ void __launch_bounds__(256) ss_bwd4_kernel(const float* __restrict__ x, long long bs, int S, int C, int ld, int Ct, int c0,
                                                      const float* __restrict__ lin, const float* __restrict__ stats,
                                                      const float* __restrict__ out_ss, const int* __restrict__ argmax,
                                                      const float* __restrict__ g_ss, const float* __restrict__ g_max,
                                                      float* __restrict__ dx, long long dbs, int rows_per_chunk, float T,
                                                      int accumulate) {
    const DivT divT(T);
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int q = C >> 2;
    const int cq = threadIdx.x % q, pl = threadIdx.x / q, npl = 256 / q;
    const float* xb = x + (long long)b * bs + 4 * cq;
    float* db = dx + (long long)b * dbs + 4 * cq;
    float m[4], inv_s[4], ex[4], ey[4], ez[4], gx[4], gy[4], gz[4], gm[4];
    int am[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * cq + e, bc = b * Ct + c0 + c;
        m[e] = stats[2 * bc]; inv_s[e] = 1.0f / stats[2 * bc + 1];
        ex[e] = out_ss[3LL * bc]; ey[e] = out_ss[3LL * bc + 1]; ez[e] = out_ss[3LL * bc + 2];
        gx[e] = g_ss[3LL * bc]; gy[e] = g_ss[3LL * bc + 1]; gz[e] = g_ss[3LL * bc + 2];
        gm[e] = g_max[bc]; am[e] = argmax[bc];
    }
    const int row0 = chunk * rows_per_chunk, row1 = min(S * S, row0 + rows_per_chunk);
    for (int row = row0; row < row1; ++row) {
        const int i = row / S, j = row - i * S;
        float base[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) base[e] = gx[e] * (lin[j] - ex[e]) + gy[e] * (lin[i] - ey[e]);
        for (int k0 = pl; k0 < S; k0 += 4 * npl) {
            float4 v[4], old[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int k = k0 + uu * npl;
                const long long o = (long long)(row * S + k) * ld;
                v[uu] = k < S ? *reinterpret_cast<const float4*>(xb + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                old[uu] = (k < S && accumulate) ? *reinterpret_cast<const float4*>(db + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int k = k0 + uu * npl;
                if (k < S) {
                    const int p = row * S + k;
                    const float lk = lin[k];
                    const float xs[4] = {v[uu].x, v[uu].y, v[uu].z, v[uu].w};
                    float r[4] = {old[uu].x, old[uu].y, old[uu].z, old[uu].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float l = divT(xs[e]);
                        const float a = exp_v(l - m[e]) * inv_s[e];
                        float g = divT(a * (base[e] + gz[e] * (lk - ez[e])));
                        if (p == am[e]) g += gm[e];
                        r[e] = accumulate ? r[e] + g : g;
                    }
                    *reinterpret_cast<float4*>(db + (long long)p * ld) = make_float4(r[0], r[1], r[2], r[3]);
                }
            }
        }
    }
}

// backward: dx[b,p,c] (+)= a_p/T * (gx*(lin[j]-ex) + gy*(lin[i]-ey) + gz*(lin[k]-ez)) + (p == argmax) * gmax
__global__ void __launch_bounds__(256) ss_bwd_kernel(const float* __restrict__ x, long long bs, int S, int C, int ld, int Ct, int c0,
                                                     const float* __restrict__ lin, const float* __restrict__ stats,
                                                     const float* __restrict__ out_ss, const int* __restrict__ argmax,
                                                     const float* __restrict__ g_ss, const float* __restrict__ g_max,
                                                     float* __restrict__ dx, long long dbs, int rows_per_chunk, float T,
                                                     int accumulate) {
    const DivT divT(T);
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int c = threadIdx.x % C, pl = threadIdx.x / C, npl = 256 / C;
    const float* xb = x + (long long)b * bs;
    float* db = dx + (long long)b * dbs;
    const int bc = b * Ct + c0 + c;
    const float m = stats[2 * bc], inv_s = 1.0f / stats[2 * bc + 1];
    const float ex = out_ss[3LL * bc], ey = out_ss[3LL * bc + 1], ez = out_ss[3LL * bc + 2];
    const float gx = g_ss[3LL * bc], gy = g_ss[3LL * bc + 1], gz = g_ss[3LL * bc + 2];
    const float gm = g_max[bc];
    const int am = argmax[bc];
    const int row0 = chunk * rows_per_chunk, row1 = min(S * S, row0 + rows_per_chunk);
    for (int row = row0; row < row1; ++row) {
        const int i = row / S, j = row - i * S;
        const float base = gx * (lin[j] - ex) + gy * (lin[i] - ey);
        for (int k = pl; k < S; k += npl) {
            const int p = row * S + k;
            const long long o = (long long)p * ld + c;
            const float l = divT(xb[o]);
            const float a = exp_v(l - m) * inv_s;
            float g = divT(a * (base + gz * (lin[k] - ez)));
            if (p == am) g += gm;
            if (accumulate) db[o] += g; else db[o] = g;
        }
    }
}

// Weight gradient of the input conv with the backward of SpatialSoftmax3D + max pool of its OUTPUT folded in: the gradient
// that reaches y is dy (conv paths) + the pooled-feature term of ss_bwd4_kernel (same formula per element), so that term
// never makes its own read-y / write-dy pass over the grid.  One workgroup per 4096 voxels of ONE sample (grid.y = b); x is
// staged through LDS in tiles of 256 voxels (the 16 threads of a voxel share its CIN inputs).
// LEAN (dy == nullptr and fold_src == nullptr: every conv path into y adds its share of dW / db itself, the shape the training step
// takes): only y is streamed, EIGHT voxels ahead instead of two -- the kernel moved 4.3 GB in 2.18 ms (2 TB/s) with 32 KB per CU in
// flight; it is bound by memory latency, not by its arithmetic (profiles/r05_*: 56 % of the wave time in s_waitcnt).
template <int CIN, int MODE>           // MODE 1: LEAN; 2: y and dy, no fold_src (the default training step): four voxels in flight; 0: generic
__global__ void __launch_bounds__(256) pw_wgrad4_ss_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, float* __restrict__ partW,
                                                           float* __restrict__ partB, int S, int vox_per_block, float slope,
                                                           const float* __restrict__ lin, const float* __restrict__ stats,
                                                           const float* __restrict__ out_ss, const int* __restrict__ argmax,
                                                           const float* __restrict__ g_ss, const float* __restrict__ g_max, float T,
                                                           const float* __restrict__ fold_src, int Sp, int pad) {
    __shared__ float red[4][64 * (CIN + 1)];
    __shared__ float sx[2][256 * CIN];
    // the coordinate table in LDS: read from global memory its three loads per voxel sit BEHIND the y prefetches on the in-order
    // vector-memory counter and every voxel waits for the newest prefetch (round 5)
    __shared__ float slin[1024];                  // (S <= 1024: checked by the entry point)
    for (int i = threadIdx.x; i < S; i += 256) slin[i] = lin[i];
    const DivT divT(T);
    const int b = blockIdx.y;
    const int c4 = (threadIdx.x & 15) * 4, gl = threadIdx.x >> 4, wid = threadIdx.x >> 6;
    const long long S3 = (long long)S * S * S;
    constexpr bool LEAN = MODE == 1, NOFOLD = MODE != 0;
    constexpr int NS = LEAN ? 8 : (MODE == 2 ? 4 : 2);   // voxels in flight per thread
    const bool has_dy = !LEAN && dy != nullptr;   // (uniform) nullptr: only the pooled-feature (and fold_src) terms reach y
    x += (long long)b * S3 * CIN; y += (long long)b * S3 * 64 + c4; dy += (long long)b * S3 * 64 + c4;
    if (!NOFOLD && fold_src) fold_src += (long long)b * Sp * Sp * Sp * 64 + c4;
    float m[4], inv_s[4], ex[4], ey[4], ez[4], gx[4], gy[4], gz[4], gm[4];
    int am[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int bc = b * 64 + c4 + e;
        m[e] = stats[2 * bc]; inv_s[e] = 1.0f / stats[2 * bc + 1];
        ex[e] = out_ss[3LL * bc]; ey[e] = out_ss[3LL * bc + 1]; ez[e] = out_ss[3LL * bc + 2];
        gx[e] = g_ss[3LL * bc]; gy[e] = g_ss[3LL * bc + 1]; gz[e] = g_ss[3LL * bc + 2];
        gm[e] = g_max[bc]; am[e] = argmax[bc];
    }
    float acc[4][CIN], accb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        accb[e] = 0.f;
#pragma unroll
        for (int i = 0; i < CIN; ++i) acc[e][i] = 0.f;
    }
    const long long v0 = (long long)blockIdx.x * vox_per_block;
    const long long v1 = min(S3, v0 + vox_per_block);
    float nxt[CIN];
    {
        const long long n = min((long long)256, v1 - v0) * CIN;
#pragma unroll
        for (int u = 0; u < CIN; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < n) sx[0][i] = x[v0 * CIN + i];
        }
    }
    __syncthreads();
    int cur = 0;
    for (long long t0 = v0; t0 < v1; t0 += 256) {
        const long long tn = t0 + 256;
        if (tn < v1) {
            const long long n = min((long long)256, v1 - tn) * CIN;
#pragma unroll
            for (int u = 0; u < CIN; ++u) {
                const int i = threadIdx.x + 256 * u;
                nxt[u] = i < n ? x[tn * CIN + i] : 0.f;
            }
        }
        const float* sxr = sx[cur];
        // 16 voxels per thread and tile (lv = gl, gl + 16, ...), one per iteration; the y / dy / fold_src loads of the voxel two
        // iterations ahead are in flight while one is consumed; (i, j, k) advance by 16 voxels without divisions
        int li_ = 0, lj_ = 0, lk_ = 0, ci_ = 0, cj_ = 0, ck_ = 0;
        {
            const int p = (int)min(t0 + gl, S3 - 1);
            const int row = p / S;
            lk_ = ck_ = p - row * S; li_ = ci_ = row / S; lj_ = cj_ = row - li_ * S;
        }
        float4 qy[NS], qd[LEAN ? 1 : NS], qf[NOFOLD ? 1 : NS];
        auto issue = [&](int slot, int lv) {
            const long long v = t0 + lv;
            const bool ok = v < v1;
            qy[slot] = ok ? *reinterpret_cast<const float4*>(y + v * 64) : make_float4(1.f, 1.f, 1.f, 1.f);
            if (LEAN) return;
            qd[slot] = (ok && has_dy) ? *reinterpret_cast<const float4*>(dy + v * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (NOFOLD) return;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fold_src && ok) {
                // one more gradient path into y, gathered in place: the adjoint of the replicate padding of a data gradient
                // that was computed on the padded domain (fold_kernel's sum; one source voxel except on the faces)
                const int i = li_, j = lj_, k = lk_;
                if (i > 0 && i < S - 1 && j > 0 && j < S - 1 && k > 0 && k < S - 1) {
                    a = *reinterpret_cast<const float4*>(fold_src + (((long long)(i + pad) * Sp + (j + pad)) * Sp + (k + pad)) * 64);
                } else {
                    const int d0 = i == 0 ? 0 : i + pad, d1 = i == S - 1 ? S - 1 + 2 * pad : i + pad;
                    const int h0 = j == 0 ? 0 : j + pad, h1 = j == S - 1 ? S - 1 + 2 * pad : j + pad;
                    const int x0 = k == 0 ? 0 : k + pad, x1 = k == S - 1 ? S - 1 + 2 * pad : k + pad;
                    for (int dz = d0; dz <= d1; ++dz)
                        for (int hh = h0; hh <= h1; ++hh)
                            for (int xx = x0; xx <= x1; ++xx) {
                                const float4 sv = *reinterpret_cast<const float4*>(fold_src + (((long long)dz * Sp + hh) * Sp + xx) * 64);
                                a.x += sv.x; a.y += sv.y; a.z += sv.z; a.w += sv.w;
                            }
                }
            }
            qf[slot] = a;
            lk_ += 16;
            while (lk_ >= S) { lk_ -= S; if (++lj_ >= S) { lj_ = 0; ++li_; } }
        };
#pragma unroll
        for (int slot = 0; slot < NS; ++slot) issue(slot, gl + 16 * slot);
#pragma unroll 1
        for (int it = 0; it < 16; it += NS) {
#pragma unroll
            for (int slot = 0; slot < NS; ++slot) {
                const int lv = gl + 16 * (it + slot);
                const float4 yy = qy[slot];
                const float4 d4 = LEAN ? make_float4(0.f, 0.f, 0.f, 0.f) : qd[LEAN ? 0 : slot], f4 = NOFOLD ? make_float4(0.f, 0.f, 0.f, 0.f) : qf[NOFOLD ? 0 : slot];
                if (it + slot + NS < 16) issue(slot, lv + 16 * NS);
                if (t0 + lv < v1) {
                    const int p = (int)(t0 + lv);
                    const float li = slin[ci_], lj = slin[cj_], lk = slin[ck_];
                    const float ys[4] = {yy.x, yy.y, yy.z, yy.w};
                    const float ds[4] = {d4.x + f4.x, d4.y + f4.y, d4.z + f4.z, d4.w + f4.w};
                    float xr[CIN];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) xr[ci] = sxr[lv * CIN + ci];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float l = divT(ys[e]);
                        const float a = exp_v(l - m[e]) * inv_s[e];
                        float g = divT(a * ((gx[e] * (lj - ex[e]) + gy[e] * (li - ey[e])) + gz[e] * (lk - ez[e])));
                        if (p == am[e]) g += gm[e];
                        float d = ds[e] + g;
                        d = ys[e] > 0.f ? d : d * slope;
                        accb[e] += d;
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) acc[e][ci] = fmaf(d, xr[ci], acc[e][ci]);
                    }
                }
                ck_ += 16;
                while (ck_ >= S) { ck_ -= S; if (++cj_ >= S) { cj_ = 0; ++ci_; } }
            }
        }
        if (tn < v1) {
#pragma unroll
            for (int u = 0; u < CIN; ++u) sx[cur ^ 1][threadIdx.x + 256 * u] = nxt[u];
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int ci = 0; ci < CIN + 1; ++ci) {
            float v = ci < CIN ? acc[e][ci < CIN ? ci : 0] : accb[e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if ((threadIdx.x & 63) < 16) red[wid][(c4 + e) * (CIN + 1) + ci] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int co = threadIdx.x;
        const long long blk = (long long)b * gridDim.x + blockIdx.x;
        constexpr int R = CIN + 1;
        for (int ci = 0; ci < CIN; ++ci)
            partW[blk * 64 * CIN + co * CIN + ci] =
                (red[0][co * R + ci] + red[1][co * R + ci]) + (red[2][co * R + ci] + red[3][co * R + ci]);
        partB[blk * 64 + co] = (red[0][co * R + CIN] + red[1][co * R + CIN]) + (red[2][co * R + CIN] + red[3][co * R + CIN]);
    }
}

// =====================================================================================================
// 3x3x3 conv, ONE output channel, replicate padding; lane = input channel (C == 64)
// =====================================================================================================
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// q[b, p] = bias + sum_t sum_c u[clamp(p + t - 1)][c] * w[c][t];  one wave handles runs of voxels along w
__global__ void __launch_bounds__(256) c1_fwd_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ q, int B, int S) {
    const int lane = threadIdx.x & 63;
    float wr[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wr[t] = w[lane * 27 + t];
    const float bv = bias[0];
    const long long nrows = (long long)B * S * S;
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += (long long)gridDim.x * 4) {
        const int h = (int)(row % S);
        const int d = (int)((row / S) % S);
        const int b = (int)(row / ((long long)S * S));
        const float* ub = u + (long long)b * S * S * S * 64;
        const float* rp[9];
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
#pragma unroll
            for (int hh = 0; hh < 3; ++hh)
                rp[dd * 3 + hh] = ub + ((long long)clampi(d + dd - 1, 0, S - 1) * S + clampi(h + hh - 1, 0, S - 1)) * S * 64 + lane;
        float win[9][3];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            win[r][1] = rp[r][0];                                  // w = 0 (and its replicate copy at w = -1)
            win[r][0] = win[r][1];
            win[r][2] = rp[r][(long long)clampi(1, 0, S - 1) * 64];
        }
        for (int x = 0; x < S; ++x) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                acc = fmaf(win[r][0], wr[r * 3 + 0], acc);
                acc = fmaf(win[r][1], wr[r * 3 + 1], acc);
                acc = fmaf(win[r][2], wr[r * 3 + 2], acc);
            }
            acc = wave_sum(acc);
            if (lane == 0) q[row * S + x] = acc + bv;
            const int nx = clampi(x + 2, 0, S - 1);
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                win[r][0] = win[r][1];
                win[r][1] = win[r][2];
                win[r][2] = rp[r][(long long)nx * 64];
            }
        }
    }
}

// du[j][c] = (du_in[j][c] + sum_{27 (o,t) pairs} dq[o] * w[c][t]) * (mask ? lrelu'(u[j][c]) : 1)
// per axis, pair k in {0,1,2}: o = j + 1 - k, t = k; o < 0 -> (0, t=0); o > S-1 -> (S-1, t=2)   (replicate adjoint)
__global__ void __launch_bounds__(256) c1_dgrad_kernel(const float* __restrict__ dq, const float* __restrict__ w,
                                                       const float* __restrict__ u, float* __restrict__ du, int B, int S,
                                                       int accumulate, int mask, float slope) {
    const int lane = threadIdx.x & 63;
    __shared__ float sw[27 * 64];          // [t][c]: runtime tap index at the borders -> LDS, not registers
    for (int i = threadIdx.x; i < 27 * 64; i += 256) sw[(i % 27) * 64 + i / 27] = w[i];
    __syncthreads();
    const long long nvox = (long long)B * S * S * S;
    for (long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); v < nvox; v += (long long)gridDim.x * 4) {
        const int x = (int)(v % S);
        const int h = (int)((v / S) % S);
        const int d = (int)((v / ((long long)S * S)) % S);
        const long long b = v / ((long long)S * S * S);
        const float* dqb = dq + b * S * S * S;
        int od[3], td[3], oh[3], th[3], ox[3], tx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int o = d + 1 - k; td[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); od[k] = clampi(o, 0, S - 1);
            o = h + 1 - k;     th[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); oh[k] = clampi(o, 0, S - 1);
            o = x + 1 - k;     tx[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); ox[k] = clampi(o, 0, S - 1);
        }
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const float g = dqb[((long long)od[a] * S + oh[bb]) * S + ox[cc]];
                    const int t = (td[a] * 3 + th[bb]) * 3 + tx[cc];
                    acc = fmaf(g, sw[t * 64 + lane], acc);
                }
        const long long o = v * 64 + lane;
        float r = acc + (accumulate ? du[o] : 0.f);
        if (mask) r = u[o] > 0.f ? r : r * slope;
        du[o] = r;
    }
}

// part[blk][c*27 + t] = sum over the block's rows of dq[o] * u[clamp(o + t - 1)][c];  partB[blk] = sum dq
__global__ void __launch_bounds__(256) c1_wgrad_kernel(const float* __restrict__ u, const float* __restrict__ dq,
                                                       float* __restrict__ part, float* __restrict__ partB, int B, int S,
                                                       int rows_per_block) {
    __shared__ float red[4][64 * 27];
    __shared__ float redb[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = 0.f;
    float accb = 0.f;
    const long long nrows = (long long)B * S * S;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    for (long long row = r0 + wid; row < r1; row += 4) {
        const int h = (int)(row % S);
        const int d = (int)((row / S) % S);
        const int b = (int)(row / ((long long)S * S));
        const float* ub = u + (long long)b * S * S * S * 64;
        const float* rp[9];
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
#pragma unroll
            for (int hh = 0; hh < 3; ++hh)
                rp[dd * 3 + hh] = ub + ((long long)clampi(d + dd - 1, 0, S - 1) * S + clampi(h + hh - 1, 0, S - 1)) * S * 64 + lane;
        float win[9][3];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            win[r][1] = rp[r][0];
            win[r][0] = win[r][1];
            win[r][2] = rp[r][(long long)clampi(1, 0, S - 1) * 64];
        }
        for (int x = 0; x < S; ++x) {
            const float g = dq[row * S + x];
            accb += g;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                acc[r * 3 + 0] = fmaf(g, win[r][0], acc[r * 3 + 0]);
                acc[r * 3 + 1] = fmaf(g, win[r][1], acc[r * 3 + 1]);
                acc[r * 3 + 2] = fmaf(g, win[r][2], acc[r * 3 + 2]);
            }
            const int nx = clampi(x + 2, 0, S - 1);
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                win[r][0] = win[r][1];
                win[r][1] = win[r][2];
                win[r][2] = rp[r][(long long)nx * 64];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 27; ++t) red[wid][lane * 27 + t] = acc[t];
    if (lane == 0) redb[wid] = accb;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 27; i += 256)
        part[(long long)blockIdx.x * 64 * 27 + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (threadIdx.x == 0) partB[blockIdx.x] = redb[0] + redb[1] + redb[2] + redb[3];
}

// =====================================================================================================
// fold: adjoint of replicate padding.  src [B, Sp^3, Cs] (valid extent S + 2*pad per axis), channels [c0, c0+C)
// dst[b, j, c] (+)= sum_{i : clamp(i - pad, 0, S-1) == j} src[b, i, c0 + c]   (* lrelu'(y[b,j,c]) if y)
// =====================================================================================================
__global__ void __launch_bounds__(256) fold_kernel(const float* __restrict__ src, int Sp, int Cs, int c0,
                                                   float* __restrict__ dst, const float* __restrict__ y, int B, int S, int C,
                                                   int pad, int accumulate, float slope) {
    const int q = C >> 2;
    const long long total = (long long)B * S * S * S * q;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % q) * 4;
        long long v = i / q;
        const int x = (int)(v % S); v /= S;
        const int h = (int)(v % S); v /= S;
        const int d = (int)(v % S); v /= S;
        const long long b = v;
        const int d0 = d == 0 ? 0 : d + pad, d1 = d == S - 1 ? S - 1 + 2 * pad : d + pad;
        const int h0 = h == 0 ? 0 : h + pad, h1 = h == S - 1 ? S - 1 + 2 * pad : h + pad;
        const int x0 = x == 0 ? 0 : x + pad, x1 = x == S - 1 ? S - 1 + 2 * pad : x + pad;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int dd = d0; dd <= d1; ++dd)
            for (int hh = h0; hh <= h1; ++hh)
                for (int xx = x0; xx <= x1; ++xx) {
                    const float4 s = *reinterpret_cast<const float4*>(
                        src + ((((long long)b * Sp + dd) * Sp + hh) * Sp + xx) * Cs + c0 + c4);
                    a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
                }
        const long long o = ((((long long)b * S + d) * S + h) * S + x) * C + c4;
        if (accumulate) {
            const float4 p = *reinterpret_cast<const float4*>(dst + o);
            a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        }
        if (y) {
            const float4 yy = *reinterpret_cast<const float4*>(y + o);
            a.x = yy.x > 0.f ? a.x : a.x * slope; a.y = yy.y > 0.f ? a.y : a.y * slope;
            a.z = yy.z > 0.f ? a.z : a.z * slope; a.w = yy.w > 0.f ? a.w : a.w * slope;
        }
        *reinterpret_cast<float4*>(dst + o) = a;
    }
}

// =====================================================================================================
// cross entropy over one huge class axis (q_trans: [B, P], P = V^3) + argmax
// =====================================================================================================
struct CePart { float m, s, xmax; int arg; };

__global__ void __launch_bounds__(256) ce_part_kernel(const float* __restrict__ x, long long P, int per_chunk, int nchunk,
                                                      CePart* __restrict__ part) {
    __shared__ CePart red[256];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float* xb = x + (long long)b * P;
    const long long p0 = (long long)chunk * per_chunk, p1 = min(P, p0 + per_chunk);
    float m = -INFINITY, s = 0.f;
    int arg = 0x7fffffff;
    for (long long p = p0 + threadIdx.x; p < p1; p += 256) {
        const float v = xb[p];
        if (v > m) { s = s * expf(m - v) + 1.0f; m = v; arg = (int)p; }
        else s += expf(v - m);
    }
    CePart a; a.m = m; a.s = s; a.xmax = m; a.arg = arg;
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            CePart l = red[threadIdx.x], r = red[threadIdx.x + o];
            const float mm = fmaxf(l.m, r.m);
            const float fl = l.m > -INFINITY ? expf(l.m - mm) : 0.f, fr = r.m > -INFINITY ? expf(r.m - mm) : 0.f;
            CePart z;
            z.m = mm; z.s = l.s * fl + r.s * fr;
            if (r.xmax > l.xmax || (r.xmax == l.xmax && r.arg < l.arg)) { z.xmax = r.xmax; z.arg = r.arg; }
            else { z.xmax = l.xmax; z.arg = l.arg; }
            red[threadIdx.x] = z;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(long long)b * nchunk + chunk] = red[0];
}

// one thread per b: lse, loss = lse - x[label], argmax
__global__ void ce_final_kernel(const float* __restrict__ x, long long P, const CePart* __restrict__ part, int nchunk,
                                const int* __restrict__ label, int B, float* __restrict__ lse, float* __restrict__ loss,
                                int* __restrict__ argmax) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float m = -INFINITY, s = 0.f, xm = -INFINITY;
    int arg = 0x7fffffff;
    for (int k = 0; k < nchunk; ++k) {
        const CePart r = part[(long long)b * nchunk + k];
        const float mm = fmaxf(m, r.m);
        s = s * (m > -INFINITY ? expf(m - mm) : 0.f) + r.s * (r.m > -INFINITY ? expf(r.m - mm) : 0.f);
        m = mm;
        if (r.xmax > xm || (r.xmax == xm && r.arg < arg)) { xm = r.xmax; arg = r.arg; }
    }
    const float l = m + logf(s);
    lse[b] = l;
    // a label outside [0, P) (a corrupt replay entry) must not become an out-of-bounds read: the loss of that sample is NaN,
    // which surfaces at the runner's `.item()` (the reference raises an indexing error at the same place, agent :519-545)
    const int lab = label[b];
    loss[b] = ((unsigned)lab < (unsigned long long)P) ? l - x[(long long)b * P + lab] : NAN;
    argmax[b] = arg;
}

__global__ void __launch_bounds__(256) ce_grad_kernel(const float* __restrict__ x, long long P, const float* __restrict__ lse,
                                                      const int* __restrict__ label, float* __restrict__ dx, int B, float gscale) {
    const long long total = (long long)B * P;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / P);
        const long long p = i - (long long)b * P;
        float g = expf(x[i] - lse[b]);
        const int lab = label[b];
        if (p == lab) g -= 1.0f;
        if ((unsigned)lab >= (unsigned long long)P) g = NAN;          // invalid label: poison the row, see ce_final_kernel
        dx[i] = g * gscale;
    }
}

// small heads: one wave per (row, segment); logits [rows, ld]; segment s covers columns [col0[s], col0[s]+ncls[s])
struct CeSegs { int n; int col0[8]; int ncls[8]; };
__global__ void __launch_bounds__(64) ce_rows_kernel(const float* __restrict__ logits, long long ld, CeSegs segs,
                                                     const int* __restrict__ labels, float* __restrict__ loss,
                                                     int* __restrict__ pred, float* __restrict__ dlogits, float gscale) {
    const int row = blockIdx.x, sg = blockIdx.y, lane = threadIdx.x;
    const float* x = logits + (long long)row * ld + segs.col0[sg];
    const int n = segs.ncls[sg];
    float m = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < n; c += 64) {
        const float v = x[c];
        if (v > m) { m = v; arg = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (om > m || (om == m && oa < arg)) { m = om; arg = oa; }
    }
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += expf(x[c] - m);
    s = wave_sum(s);
    const float l = m + logf(s);
    const int lab = labels[row * segs.n + sg];
    const bool lab_ok = (unsigned)lab < (unsigned)n;                  // invalid label -> NaN loss / gradient, never an OOB read
    if (lane == 0) {
        loss[row * segs.n + sg] = lab_ok ? l - x[lab] : NAN;
        pred[row * segs.n + sg] = arg;
    }
    if (dlogits) {
        float* d = dlogits + (long long)row * ld + segs.col0[sg];
        for (int c = lane; c < n; c += 64) d[c] = lab_ok ? (expf(x[c] - l) - (c == lab ? 1.0f : 0.0f)) * gscale : NAN;
    }
}

// =====================================================================================================
// polyphase weights:  Weff[(j3*Cin + ci)][(r3*Cout + co)] = sum_t W[co][ci][t3] * L[rd][td][jd] L[rh][th][jh] L[rw][tw][jw]
// =====================================================================================================
// one workgroup per output row (j3, ci): the 27 x Cout weights W[:, ci, :] it needs are staged once in LDS ([t][co], so the
// Cout-contiguous threads read consecutive words) instead of being fetched 27 times per element with a Cin*27-float stride
__global__ void __launch_bounds__(256) weff_fwd_kernel(const float* __restrict__ W, const float* __restrict__ L, float* __restrict__ Weff,
                                                       int Cin, int Cout, int k, int s, int kl) {
    extern __shared__ float sm[];          // sL [s][k][kl], then sW [T][Cout]
    float* sL = sm;
    float* sW = sm + s * k * kl;
    const int T = k * k * k;
    const int row = blockIdx.x;            // j3 * Cin + ci
    const int ci = row % Cin, j3 = row / Cin;
    for (int i = threadIdx.x; i < s * k * kl; i += 256) sL[i] = L[i];
    for (int i = threadIdx.x; i < T * Cout; i += 256) {
        const int co = i / T, t = i - co * T;
        sW[t * Cout + co] = W[((long long)co * Cin + ci) * T + t];
    }
    __syncthreads();
    const int jw = j3 % kl, jh = (j3 / kl) % kl, jd = j3 / (kl * kl);
    const int ncol = s * s * s * Cout;
    for (int col = threadIdx.x; col < ncol; col += 256) {
        const int co = col % Cout, r3 = col / Cout;
        const int rw = r3 % s, rh = (r3 / s) % s, rd = r3 / (s * s);
        float acc = 0.f;
        for (int td = 0; td < k; ++td) {
            const float ld = sL[(rd * k + td) * kl + jd];
            if (ld == 0.f) continue;
            for (int th = 0; th < k; ++th) {
                const float lh = ld * sL[(rh * k + th) * kl + jh];
                if (lh == 0.f) continue;
                for (int tw = 0; tw < k; ++tw)
                    acc = fmaf(sW[((td * k + th) * k + tw) * Cout + co], lh * sL[(rw * k + tw) * kl + jw], acc);
            }
        }
        Weff[(long long)row * ncol + col] = acc;
    }
}
// adjoint: dW[co][ci][t3] += sum_{r3, j3} dWeff[(j3*Cin+ci)][(r3*Cout+co)] * L3
__global__ void __launch_bounds__(256) weff_bwd_kernel(const float* __restrict__ dWeff, const float* __restrict__ L, float* __restrict__ dW,
                                                       int Cin, int Cout, int k, int s, int kl) {
    extern __shared__ float sL[];
    for (int i = threadIdx.x; i < s * k * kl; i += 256) sL[i] = L[i];
    __syncthreads();
    const int T = k * k * k;
    const long long ncol = (long long)s * s * s * Cout;
    const long long total = (long long)Cout * Cin * T;
    // thread order: co fastest so that dWeff reads are coalesced
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        const int ci = (int)((i / Cout) % Cin);
        const int t3 = (int)(i / ((long long)Cout * Cin));
        const int tw = t3 % k, th = (t3 / k) % k, td = t3 / (k * k);
        float acc = 0.f;
        for (int rd = 0; rd < s; ++rd)
            for (int jd = 0; jd < kl; ++jd) {
                const float ld = sL[(rd * k + td) * kl + jd];
                if (ld == 0.f) continue;
                for (int rh = 0; rh < s; ++rh)
                    for (int jh = 0; jh < kl; ++jh) {
                        const float lh = ld * sL[(rh * k + th) * kl + jh];
                        if (lh == 0.f) continue;
                        for (int rw = 0; rw < s; ++rw)
                            for (int jw = 0; jw < kl; ++jw) {
                                const float lw = lh * sL[(rw * k + tw) * kl + jw];
                                if (lw == 0.f) continue;
                                const long long row = (long long)((jd * kl + jh) * kl + jw) * Cin + ci;
                                const long long col = (long long)((rd * s + rh) * s + rw) * Cout + co;
                                acc = fmaf(dWeff[row * ncol + col], lw, acc);
                            }
                    }
            }
        dW[((long long)co * Cin + ci) * T + t3] += acc;
    }
}

// =====================================================================================================
// LAMB (lamb.py:94-122), multi-tensor over flat buffers.  chunk table: {tensor id, start, len} per chunk
// =====================================================================================================
__global__ void __launch_bounds__(256) lamb_stage1_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, float* __restrict__ upd,
                                                          const int* __restrict__ chunks, float* __restrict__ part, float beta1,
                                                          float beta2, float omb1, float omb2, float eps, float wd,
                                                          const int* __restrict__ skip) {
    __shared__ float r1[4], r2[4];
    if (skip && *skip < 0) return;             // a step whose inputs were flagged invalid on the device is a no-op (see the C entry)
    const int* ch = chunks + 3 * blockIdx.x;
    const long long start = ch[1];
    const int len = ch[2];
    float sw = 0.f, su = 0.f;
    for (int i = threadIdx.x; i < len; i += 256) {
        const long long o = start + i;
        const float gv = g[o], wv = w[o];
        // omb = 1 - beta evaluated in DOUBLE on the host and then rounded, as Python does for `alpha=1 - beta1` /
        // `value=1 - beta2` (lamb.py:99-101): 1.0f - 0.999f is 1.3e-5 away from float(0.001)
        const float mv = m[o] * beta1 + omb1 * gv;
        const float vv = v[o] * beta2 + omb2 * gv * gv;
        m[o] = mv; v[o] = vv;
        float u = mv / (sqrtf(vv) + eps);
        if (wd != 0.f) u += wd * wv;
        upd[o] = u;
        sw = fmaf(wv, wv, sw);
        su = fmaf(u, u, su);
    }
    sw = wave_sum(sw); su = wave_sum(su);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { r1[wid] = sw; r2[wid] = su; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = r1[0] + r1[1] + r1[2] + r1[3];
        part[2 * blockIdx.x + 1] = r2[0] + r2[1] + r2[2] + r2[3];
    }
}
// one thread per tensor: trust ratio from the partial sums of its chunks [first[t], first[t+1])
__global__ void lamb_stage2_kernel(const float* __restrict__ part, const int* __restrict__ first, int ntensors,
                                   float* __restrict__ trust) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntensors) return;
    float sw = 0.f, su = 0.f;
    for (int c = first[t]; c < first[t + 1]; ++c) { sw += part[2 * c]; su += part[2 * c + 1]; }
    float wn = sqrtf(sw);
    wn = fminf(fmaxf(wn, 0.f), 10.f);
    const float an = sqrtf(su);
    trust[t] = (wn == 0.f || an == 0.f) ? 1.0f : wn / an;
}
__global__ void __launch_bounds__(256) lamb_stage3_kernel(float* __restrict__ w, const float* __restrict__ upd,
                                                          const int* __restrict__ chunks, const float* __restrict__ trust, float lr,
                                                          const int* __restrict__ skip) {
    if (skip && *skip < 0) return;
    const int* ch = chunks + 3 * blockIdx.x;
    const long long start = ch[1];
    const int len = ch[2];
    const float alpha = -(lr * trust[ch[0]]);
    // p.data.add_(adam_step, alpha=...) rounds the product and the sum separately on the CPU path (lamb.py:122)
    for (int i = threadIdx.x; i < len; i += 256) w[start + i] = __fadd_rn(w[start + i], __fmul_rn(alpha, upd[start + i]));
}

// torch.optim.Adam (the reference's alternative optimizer, agent :263-268), single-tensor formulas of torch/optim/adam.py
// applied to the flat arena: g += wd * w; m = lerp(m, g, 1 - beta1); v = v * beta2 + (1 - beta2) g g;
// w -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float beta2, float omb1, float omb2, float eps,
                                                   float wd, float step_size, float bc2_sqrt, const int* __restrict__ skip) {
    if (skip && *skip < 0) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float gv = g[i];
        const float wv = w[i];
        if (wd != 0.f) gv = __fadd_rn(gv, __fmul_rn(wd, wv));
        const float mv = __fadd_rn(m[i], __fmul_rn(omb1, __fsub_rn(gv, m[i])));          // exp_avg.lerp_(grad, 1 - beta1)
        const float vv = __fadd_rn(__fmul_rn(v[i], beta2), __fmul_rn(__fmul_rn(omb2, gv), gv));
        m[i] = mv; v[i] = vv;
        const float denom = __fadd_rn(__fdiv_rn(sqrtf(vv), bc2_sqrt), eps);
        w[i] = __fadd_rn(wv, __fmul_rn(-step_size, __fdiv_rn(mv, denom)));              // addcdiv_(exp_avg, denom, value=-step_size)
    }
}

}  // namespace

extern "C" int vxb_adam_step_f32(float* w, const float* g, float* m, float* v, int64_t n, float lr, double beta1, double beta2,
                                 float eps, float weight_decay, int64_t step, const int32_t* skip_if_negative, vxb_stream_t stream) {
    if (!w || !g || !m || !v || n < 1 || step < 1) return VXB_EARG;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, (long long)n, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, (float)((double)lr / bc1), (float)sqrt(bc2), skip_if_negative);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_pointwise_fwd_f32(const float* x, const float* W, const float* bias, float* y, int64_t nvox, int Cin,
                                     int Cout, float slope, vxb_stream_t stream) {
    if (!x || !W || !bias || !y || nvox < 1 || Cin < 1 || Cin > 16 || Cout < 4 || Cout > 128 || (Cout & 3)) return VXB_EARG;
    hipLaunchKernelGGL(pw_fwd_kernel, dim3(grid_for(nvox * (Cout / 4))), dim3(256), 0, (hipStream_t)stream, x, W, bias, y,
                       (long long)nvox, Cin, Cout, slope);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
// part_ws: nblk*(64*Cin + 64) floats with nblk = ceil(nvox / 4096).  dW [64][Cin] and db [64] are ACCUMULATED.
extern "C" int vxb_pointwise_wgrad_f32(const float* x, const float* y, const float* dy, float* dW, float* db, float* part_ws,
                                       int64_t nvox, int Cin, int Cout, float slope, vxb_stream_t stream);
extern "C" int vxb_sum_splits_f32(const float* part, int nsplit, int64_t n, float* dst, int accumulate, float alpha, vxb_stream_t stream);

extern "C" int vxb_pointwise_wgrad_f32(const float* x, const float* y, const float* dy, float* dW, float* db, float* part_ws,
                                       int64_t nvox, int Cin, int Cout, float slope, vxb_stream_t stream) {
    if (!x || !y || !dy || !dW || !db || !part_ws || nvox < 1 || Cin < 1 || Cin > 16 || Cout != 64) return VXB_EARG;
    const int vpb = 4096;
    const int nb = vxb_cdiv(nvox, vpb);
    float* pW = part_ws;
    float* pB = part_ws + (size_t)nb * 64 * Cin;
    if (((((uintptr_t)y) | ((uintptr_t)dy)) & 15) == 0)
        hipLaunchKernelGGL(pw_wgrad4_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, dy, pW, pB, (long long)nvox, Cin, vpb, slope);
    else
        hipLaunchKernelGGL(pw_wgrad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, dy, pW, pB, (long long)nvox, Cin, vpb, slope);
    VXB_CHECK_LAUNCH();
    int rc = vxb_sum_splits_f32(pW, nb, 64 * Cin, dW, 1, 1.0f, stream);
    if (rc) return rc;
    return vxb_sum_splits_f32(pB, nb, 64, db, 1, 1.0f, stream);
}

// x: [B, S^3, C] with batch stride bs (elements).  part_ws: B*nchunk*C*7 floats, nchunk = ceil(S*S / rows_per_chunk),
// rows_per_chunk = max(1, S*S/64).  Outputs: out_ss [B,3C], out_max [B,C], stats [B,C,2], argmax [B,C] (int32).
static inline int vxb_ss3d_chunks(int B) { const int w = (1024 + B - 1) / B; return w < 64 ? 64 : w; }

extern "C" int vxb_ss3d_max_fwd_f32(const float* x, int64_t bs, int B, int S, int C, const float* lin, float* part_ws,
                                    float* out_ss, float* out_max, float* stats, int32_t* argmax, vxb_stream_t stream) {
    if (!x || !lin || !part_ws || !out_ss || !out_max || !stats || !argmax || B < 1 || S < 1) return VXB_EARG;
    if (C != 64 && C != 128 && C != 192) return VXB_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    // >= 1024 workgroups per launch: 64 chunks of (d, h) rows per sample at B >= 16, more for small batches (act(): B = 1)
    const int want = vxb_ss3d_chunks(B);
    const int rpc = (S * S / want) < 1 ? 1 : S * S / want;
    const int nchunk = vxb_cdiv(S * S, rpc);
    // the kernels spread 64 or 128 channels over a workgroup: a 192-wide tensor (2Robots context) runs as the slabs 128 + 64
    for (int c0 = 0; c0 < C; c0 += 128) {
        const int Cs = C - c0 < 128 ? C - c0 : 128;
        const float* xs = x + c0;
        if ((bs & 3) == 0 && (((uintptr_t)xs) & 15) == 0 && (C & 3) == 0)
            hipLaunchKernelGGL(ss_part4_kernel, dim3(nchunk, B), dim3(256), 0, st, xs, (long long)bs, S, Cs, C, lin, rpc, (SsPart*)part_ws, nchunk, 0.01f);
        else
            hipLaunchKernelGGL(ss_part_kernel, dim3(nchunk, B), dim3(256), 0, st, xs, (long long)bs, S, Cs, C, lin, rpc, (SsPart*)part_ws, nchunk, 0.01f);
        if (nchunk > 128)
            hipLaunchKernelGGL(ss_final_wide_kernel, dim3(B * Cs), dim3(256), 0, st, (const SsPart*)part_ws, nchunk, B, Cs, C, c0, out_ss,
                               out_max, stats, argmax, (long long)nchunk * Cs, (long long)Cs, 1LL);
        else
            hipLaunchKernelGGL(ss_final_kernel, dim3(vxb_cdiv(B * Cs, 256)), dim3(256), 0, st, (const SsPart*)part_ws, nchunk, B, Cs, C, c0,
                               out_ss, out_max, stats, argmax);
    }
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
// final merge of the partials the `final` conv's epilogue wrote (conv_halo_bf16.hip): part [B][C][ntiles]
int vxb_ss3d_final_tiles_launch(const float* part, int ntiles, int B, int C, float* out_ss, float* out_max, float* stats, int32_t* argmax,
                                hipStream_t st) {
    hipLaunchKernelGGL(ss_final_wide_kernel, dim3(B * C), dim3(256), 0, st, (const SsPart*)part, ntiles, B, C, C, 0, out_ss, out_max, stats,
                       argmax, (long long)C * ntiles, 1LL, (long long)ntiles);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_ss3d_max_bwd_f32(const float* x, int64_t bs, int B, int S, int C, const float* lin, const float* stats,
                                    const float* out_ss, const int32_t* argmax, const float* g_ss, const float* g_max,
                                    float* dx, int64_t dbs, int accumulate, vxb_stream_t stream) {
    if (!x || !lin || !stats || !out_ss || !argmax || !g_ss || !g_max || !dx || B < 1 || S < 1) return VXB_EARG;
    if (C != 64 && C != 128 && C != 192) return VXB_ESIZE;
    const int rpc = (S * S / 256) < 1 ? 1 : S * S / 256;
    const int nchunk = vxb_cdiv(S * S, rpc);
    for (int c0 = 0; c0 < C; c0 += 128) {
        const int Cs = C - c0 < 128 ? C - c0 : 128;
        const float* xs = x + c0;
        float* ds = dx + c0;
        if ((bs & 3) == 0 && (dbs & 3) == 0 && (((uintptr_t)xs) & 15) == 0 && (((uintptr_t)ds) & 15) == 0)
            hipLaunchKernelGGL(ss_bwd4_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, xs, (long long)bs, S, Cs, C, C, c0, lin, stats,
                               out_ss, argmax, g_ss, g_max, ds, (long long)dbs, rpc, 0.01f, accumulate);
        else
            hipLaunchKernelGGL(ss_bwd_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, xs, (long long)bs, S, Cs, C, C, c0, lin, stats,
                               out_ss, argmax, g_ss, g_max, ds, (long long)dbs, rpc, 0.01f, accumulate);
    }
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// Fused forms for the first layer (C = 64 output channels): the input conv + LeakyReLU with the SpatialSoftmax3D / max-pool
// statistics of its output taken while the output is written, and the conv's parameter gradients with the pooled features'
// backward term added to dy on the fly.  Same results as the two-kernel paths (forward: bit-identical).
extern "C" int vxb_pointwise_ss3d_fwd_f32(const float* x, const float* W, const float* bias, float* y, int B, int S, int Cin, int Cout,
                                          float slope, const float* lin, float* part_ws, float* out_ss, float* out_max, float* stats,
                                          int32_t* argmax, vxb_stream_t stream) {
    if (!x || !W || !bias || !y || !lin || !part_ws || !out_ss || !out_max || !stats || !argmax || B < 1 || S < 1 || Cin < 1 ||
        Cin > 16) return VXB_EARG;
    if (Cout != 64 || (((uintptr_t)y) & 15)) return VXB_ESIZE;
    if (Cin != 10 || S > PWSS_MAX_S) {          // other shapes: the two kernels one after the other (same results)
        const int rc = vxb_pointwise_fwd_f32(x, W, bias, y, (int64_t)B * S * S * S, Cin, Cout, slope, stream);
        if (rc) return rc;
        return vxb_ss3d_max_fwd_f32(y, (int64_t)S * S * S * 64, B, S, 64, lin, part_ws, out_ss, out_max, stats, argmax, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    const int want = vxb_ss3d_chunks(B);
    const int rpc = (S * S / want) < 1 ? 1 : S * S / want;
    const int nchunk = vxb_cdiv(S * S, rpc);
    hipLaunchKernelGGL(pw_fwd_ss_kernel<10>, dim3(nchunk, B), dim3(256), 0, st, x, W, bias, y, S, slope, lin, rpc, (SsPart*)part_ws,
                       nchunk, 0.01f);
    if (nchunk > 128)
        hipLaunchKernelGGL(ss_final_wide_kernel, dim3(B * 64), dim3(256), 0, st, (const SsPart*)part_ws, nchunk, B, 64, 64, 0, out_ss,
                           out_max, stats, argmax, (long long)nchunk * 64, (long long)64, 1LL);
    else
        hipLaunchKernelGGL(ss_final_kernel, dim3(vxb_cdiv(B * 64, 256)), dim3(256), 0, st, (const SsPart*)part_ws, nchunk, B, 64, 64, 0,
                           out_ss, out_max, stats, argmax);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
// part_ws: B * ceil(S^3 / 4096) * (64*Cin + 64) floats.  dW [64][Cin] and db [64] are ACCUMULATED.  Cin = 10 (the network's
// voxel features: 3 + 3 + 3 + 1) -- other widths return VXB_ESIZE and the caller uses the two separate kernels.
// fold_src (or null): [B, Sp^3, 64], a data gradient on the replicate-padded domain (S + 2 pad valid per axis) whose padding
// adjoint (vxb_fold_pad_f32) is gathered and added to dy on the fly as well.
extern "C" int vxb_pointwise_wgrad_ss3d_f32(const float* x, const float* y, const float* dy, float* dW, float* db, float* part_ws, int B,
                                            int S, int Cin, int Cout, float slope, const float* lin, const float* stats,
                                            const float* out_ss, const int32_t* argmax, const float* g_ss, const float* g_max,
                                            const float* fold_src, int Sp, int pad, vxb_stream_t stream) {
    if (!x || !y || !dW || !db || !part_ws || !lin || !stats || !out_ss || !argmax || !g_ss || !g_max || B < 1 || S < 1 ||
        Cin < 1 || Cin > 16 || (fold_src && (pad < 0 || Sp < S + 2 * pad))) return VXB_EARG;
    if (Cout != 64 || Cin != 10 || S > 1024 || ((((uintptr_t)y) | ((uintptr_t)dy) | ((uintptr_t)fold_src)) & 15)) return VXB_ESIZE;
    const int vpb = 4096;
    const int nbs = vxb_cdiv((long long)S * S * S, vpb);
    const int nb = nbs * B;
    float* pW = part_ws;
    float* pB = part_ws + (size_t)nb * 64 * Cin;
#define PW_WG(MODE_) hipLaunchKernelGGL((pw_wgrad4_ss_kernel<10, MODE_>), dim3(nbs, B), dim3(256), 0, (hipStream_t)stream, x, y, dy, pW, pB, S, vpb, slope, \
                                        lin, stats, out_ss, argmax, g_ss, g_max, 0.01f, fold_src, Sp, pad)
    if (!dy && !fold_src) PW_WG(1);
    else if (!fold_src) PW_WG(2);
    else PW_WG(0);
#undef PW_WG
    VXB_CHECK_LAUNCH();
    int rc = vxb_sum_splits_f32(pW, nb, 64 * Cin, dW, 1, 1.0f, stream);
    if (rc) return rc;
    return vxb_sum_splits_f32(pB, nb, 64, db, 1, 1.0f, stream);
}

extern "C" int vxb_conv3_c1_fwd_f32(const float* u, const float* w, const float* bias, float* q, int B, int S, int C,
                                    vxb_stream_t stream) {
    if (!u || !w || !bias || !q || B < 1 || S < 1) return VXB_EARG;
    if (C != 64) return VXB_ESIZE;
    if ((((uintptr_t)u) & 15) == 0) return vxb_c1_fwd4_launch(u, w, bias, q, B, S, (hipStream_t)stream);
    const long long nrows = (long long)B * S * S;
    const int grid = (int)((nrows + 3) / 4 > 8192 ? 8192 : (nrows + 3) / 4);
    hipLaunchKernelGGL(c1_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u, w, bias, q, B, S);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
extern "C" int vxb_conv3_c1_dgrad_f32(const float* dq, const float* w, const float* u, float* du, int B, int S, int C,
                                      int accumulate, int apply_lrelu_mask, float slope, vxb_stream_t stream) {
    if (!dq || !w || !du || (apply_lrelu_mask && !u) || B < 1 || S < 1) return VXB_EARG;
    if (C != 64) return VXB_ESIZE;
    if ((S & 3) == 0) return vxb_c1_dgrad4_launch(dq, w, u, du, B, S, accumulate, apply_lrelu_mask, slope, (hipStream_t)stream);
    const long long nvox = (long long)B * S * S * S;
    const int grid = (int)((nvox + 3) / 4 > 32768 ? 32768 : (nvox + 3) / 4);
    hipLaunchKernelGGL(c1_dgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dq, w, u, du, B, S, accumulate, apply_lrelu_mask, slope);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
// part_ws: nblk*(64*27 + 1) floats, nblk = ceil(B*S*S / 64).  dw [1][64][27] and db [1] are ACCUMULATED.
// du = lrelu'(u) * ([du] + conv3_c1 data gradient of dq + the pooled-feature term of vxb_ss3d_max_bwd_f32 on u), and
// dbias[64] += column sums of the finished du -- one pass over u / du instead of three (c1_conv.hip).  S % 4 == 0, C = 64.
extern "C" size_t vxb_conv3_c1_dgrad_ss3d_ws_floats(int B, int S) {
    return ((size_t)vxb_c1_dgrad_ss_blocks_per_sample(S) * (size_t)B + 64) * 65;      // (+ one |du| maximum per block)
}
extern "C" int vxb_conv3_c1_dgrad_ss3d_f32(const float* dq, const float* w, const float* u, float* du, int B, int S, int C,
                                           int accumulate, float slope, const float* lin, const float* stats, const float* out_ss,
                                           const int32_t* argmax, const float* g_ss, const float* g_max, float* dbias,
                                           float* part_ws, float* du_scale, vxb_stream_t stream) {
    if (!dq || !w || !u || !du || !lin || !stats || !out_ss || !argmax || !g_ss || !g_max || !dbias || !part_ws || B < 1 || S < 4)
        return VXB_EARG;
    if (C != 64 || (S & 3) || ((((uintptr_t)u) | ((uintptr_t)du)) & 15)) return VXB_ESIZE;
    // du_scale (optional, [2]): the fp16 operand scale of du (as vxb_absmax_scale_f32 would compute it), taken while du is written
    const int nb = vxb_c1_dgrad_ss_blocks_per_sample(S) * B;
    unsigned* part_amax = du_scale ? reinterpret_cast<unsigned*>(part_ws + ((size_t)nb + 64) * 64) : nullptr;
    int rc = vxb_c1_dgrad4_ss_launch(dq, w, u, du, B, S, accumulate, slope, lin, stats, out_ss, argmax, g_ss, g_max, dbias, part_ws,
                                     part_amax, (hipStream_t)stream);
    if (rc || !du_scale) return rc;
    return vxb_absmax_finish_launch(part_amax, nb, du_scale, (hipStream_t)stream);
}
extern "C" int vxb_conv3_c1_wgrad_f32(const float* u, const float* dq, float* dw, float* db, float* part_ws, int B, int S, int C,
                                      vxb_stream_t stream) {
    if (!u || !dq || !dw || !db || !part_ws || B < 1 || S < 1) return VXB_EARG;
    if (C != 64) return VXB_ESIZE;
    if ((S & 3) == 0) return vxb_c1_wgrad4_launch(u, dq, dw, db, part_ws, B, S, (hipStream_t)stream);
    const long long nrows = (long long)B * S * S;
    const int rpb = 64;
    const int nb = vxb_cdiv(nrows, rpb);
    float* pW = part_ws;
    float* pB = part_ws + (size_t)nb * 64 * 27;
    hipLaunchKernelGGL(c1_wgrad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, u, dq, pW, pB, B, S, rpb);
    VXB_CHECK_LAUNCH();
    int rc = vxb_sum_splits_f32(pW, nb, 64 * 27, dw, 1, 1.0f, stream);
    if (rc) return rc;
    return vxb_sum_splits_f32(pB, nb, 1, db, 1, 1.0f, stream);
}

extern "C" int vxb_fold_pad_f32(const float* src, int Sp, int Cs, int c0, float* dst, const float* lrelu_of, int B, int S,
                                int C, int pad, int accumulate, float slope, vxb_stream_t stream) {
    if (!src || !dst || B < 1 || S < 1 || C < 4 || (C & 3) || (Cs & 3) || (c0 & 3) || pad < 0 || Sp < S + 2 * pad) return VXB_EARG;
    hipLaunchKernelGGL(fold_kernel, dim3(grid_for((long long)B * S * S * S * (C / 4))), dim3(256), 0, (hipStream_t)stream, src, Sp, Cs,
                       c0, dst, lrelu_of, B, S, C, pad, accumulate, slope);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// part_ws: B*nchunk*4 floats (nchunk = ceil(P / 65536)).  Outputs per sample: lse, loss, argmax; dx = gscale*(softmax - onehot).
extern "C" int vxb_ce_big_f32(const float* x, int64_t P, int B, const int32_t* label, float* part_ws, float* lse, float* loss,
                              int32_t* argmax, float* dx, float gscale, vxb_stream_t stream) {
    if (!x || !label || !part_ws || !lse || !loss || !argmax || B < 1 || P < 1 || P >= INT32_MAX) return VXB_EARG;
    hipStream_t st = (hipStream_t)stream;
    const int per = 65536;
    const int nchunk = vxb_cdiv(P, per);
    hipLaunchKernelGGL(ce_part_kernel, dim3(nchunk, B), dim3(256), 0, st, x, (long long)P, per, nchunk, (CePart*)part_ws);
    hipLaunchKernelGGL(ce_final_kernel, dim3(vxb_cdiv(B, 64)), dim3(64), 0, st, x, (long long)P, (const CePart*)part_ws, nchunk, label, B,
                       lse, loss, argmax);
    if (dx) hipLaunchKernelGGL(ce_grad_kernel, dim3(grid_for((long long)B * P)), dim3(256), 0, st, x, (long long)P, lse, label, dx, B, gscale);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
// logits [rows, ld]; nseg <= 8 segments (col0, ncls); labels [rows, nseg] int32; loss, pred [rows, nseg]; dlogits may be null.
extern "C" int vxb_ce_rows_f32(const float* logits, int64_t ld, int rows, int nseg, const int32_t* col0, const int32_t* ncls,
                               const int32_t* labels, float* loss, int32_t* pred, float* dlogits, float gscale,
                               vxb_stream_t stream) {
    if (!logits || !col0 || !ncls || !labels || !loss || !pred || rows < 1 || nseg < 1 || nseg > 8) return VXB_EARG;
    CeSegs s;
    s.n = nseg;
    for (int i = 0; i < 8; ++i) { s.col0[i] = i < nseg ? col0[i] : 0; s.ncls[i] = i < nseg ? ncls[i] : 0; }
    hipLaunchKernelGGL(ce_rows_kernel, dim3(rows, nseg), dim3(64), 0, (hipStream_t)stream, logits, (long long)ld, s, labels, loss, pred,
                       dlogits, gscale);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_polyphase_weights_f32(const float* W, const float* L, float* Weff, int Cin, int Cout, int k, int s, int kl,
                                         vxb_stream_t stream) {
    if (!W || !L || !Weff || Cin < 1 || Cout < 1 || k < 1 || s < 1 || kl < 1 || s * k * kl > 8192) return VXB_EARG;
    const size_t lds = (size_t)(s * k * kl + k * k * k * Cout) * sizeof(float);
    if (lds > 64 * 1024) return VXB_ESIZE;
    hipLaunchKernelGGL(weff_fwd_kernel, dim3(kl * kl * kl * Cin), dim3(256), lds, (hipStream_t)stream, W, L, Weff, Cin, Cout, k, s, kl);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
extern "C" int vxb_polyphase_weights_bwd_f32(const float* dWeff, const float* L, float* dW, int Cin, int Cout, int k, int s, int kl,
                                             vxb_stream_t stream) {
    if (!dWeff || !L || !dW || Cin < 1 || Cout < 1 || k < 1 || s < 1 || kl < 1 || s * k * kl > 8192) return VXB_EARG;
    const long long total = (long long)Cout * Cin * k * k * k;
    hipLaunchKernelGGL(weff_bwd_kernel, dim3(grid_for(total)), dim3(256), s * k * kl * sizeof(float), (hipStream_t)stream, dWeff, L, dW, Cin,
                       Cout, k, s, kl);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// chunks: int32 [nchunks][3] = {tensor id, start, len} (device); first: int32 [ntensors+1] (device);
// upd: scratch of the same length as w; part: 2*nchunks floats; trust: ntensors floats (kept for inspection).
// skip_if_negative (optional, device): *skip < 0 makes the whole step a no-op -- weights and moments untouched.  The SE(3)
// relabel kernel sets its status word negative when the retry budget is exhausted (augmentation.py:119-120 raises BEFORE the
// forward pass upstream); that step's loss and gradients are NaN and must not reach the optimizer state.
extern "C" int vxb_lamb_step_f32(float* w, const float* g, float* m, float* v, float* upd, const int32_t* chunks, int nchunks,
                                 const int32_t* first, int ntensors, float* part, float* trust, float lr, double beta1, double beta2,
                                 float eps, float weight_decay, const int32_t* skip_if_negative, vxb_stream_t stream) {
    if (!w || !g || !m || !v || !upd || !chunks || !first || !part || !trust || nchunks < 1 || ntensors < 1) return VXB_EARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lamb_stage1_kernel, dim3(nchunks), dim3(256), 0, st, w, g, m, v, upd, chunks, part, (float)beta1, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, skip_if_negative);
    hipLaunchKernelGGL(lamb_stage2_kernel, dim3(vxb_cdiv(ntensors, 64)), dim3(64), 0, st, part, first, ntensors, trust);
    hipLaunchKernelGGL(lamb_stage3_kernel, dim3(nchunks), dim3(256), 0, st, w, upd, chunks, trust, lr, skip_if_negative);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

// ---- rigid transform of a channels-first point cloud (SE(3) augmentation, peract/voxel/augmentation.py:36-57):
//      dst[b, j, n] = sum_k (src[b, k, n] - xf[b][9 + k]) * xf[b][3 k + j] + xf[b][12 + j]
// xf [B][15] = rotation R (row-major; points are rotated as ROW vectors, p' = p R), gripper position, new centre.
// Replaces the reshape / subtract / bmm / transpose / add chain (five kernels and a K = 3 batched GEMM per camera).
namespace {
__global__ void __launch_bounds__(256) se3_points_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         const float* __restrict__ xf, long long n) {
    const int b = blockIdx.y;
    const float* x = xf + b * 15;
    const float r00 = x[0], r01 = x[1], r02 = x[2], r10 = x[3], r11 = x[4], r12 = x[5], r20 = x[6], r21 = x[7], r22 = x[8];
    const float t0 = x[9], t1 = x[10], t2 = x[11], c0 = x[12], c1 = x[13], c2 = x[14];
    const float* s = src + (long long)b * 3 * n;
    float* d = dst + (long long)b * 3 * n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p0 = s[i] - t0, p1 = s[n + i] - t1, p2 = s[2 * n + i] - t2;
        d[i] = fmaf(p2, r20, fmaf(p1, r10, p0 * r00)) + c0;
        d[n + i] = fmaf(p2, r21, fmaf(p1, r11, p0 * r01)) + c1;
        d[2 * n + i] = fmaf(p2, r22, fmaf(p1, r12, p0 * r02)) + c2;
    }
}
}  // namespace

extern "C" int vxb_se3_points_f32(const float* src, float* dst, const float* xf, int B, int64_t n, vxb_stream_t stream) {
    if (!src || !dst || !xf || B < 1 || n < 1) return VXB_EARG;
    const int gx = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(se3_points_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, src, dst, xf, (long long)n);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}
