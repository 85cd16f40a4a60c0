// float4 versions of the Cout = 1 3x3x3 conv of the translation head (trans_decoder, perceiver_lang_io.py:316-321 /
// :466) for S % 4 == 0: a thread owns 4 of the 64 channels (16 threads per voxel -> 256-byte coalesced rows).
// Same summation order as the scalar kernels in vox_ops.hip (taps in (d, h, w) order, fmaf chain), so the data
// gradient is bit-identical to them; the weight gradient only differs in how rows are grouped into partial sums.
#include "common.h"

namespace {

__device__ __forceinline__ int c1_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ void c1_fma4(float4& a, float g, const float4 w) {
    a.x = fmaf(g, w.x, a.x); a.y = fmaf(g, w.y, a.y); a.z = fmaf(g, w.z, a.z); a.w = fmaf(g, w.w, a.w);
}

// du[j][c] = (du_in[j][c] + sum_{27 (o,t) pairs} dq[o] * w[c][t]) * (mask ? lrelu'(u[j][c]) : 1)
// per axis, pair k in {0,1,2}: o = j + 1 - k, t = k; o < 0 -> (0, t=0); o > S-1 -> (S-1, t=2)   (replicate adjoint)
// A thread handles 4 consecutive x voxels: interior groups read a 3x3x6 dq window with the weights in registers,
// groups touching a face take the table-driven path with the weights in LDS.
__global__ void __launch_bounds__(256, 4) c1_dgrad4_kernel(const float* __restrict__ dq, const float* __restrict__ w,
                                                        const float* __restrict__ u, float* __restrict__ du, int B, int S,
                                                        int accumulate, int mask, float slope) {
    __shared__ float sw[27 * 64];          // [t][c]
    for (int i = threadIdx.x; i < 27 * 64; i += 256) sw[(i % 27) * 64 + i / 27] = w[i];
    const int c4 = (threadIdx.x & 15) * 4, gl = threadIdx.x >> 4;
    // (weights stay in LDS: 27 float4 in registers would cost 108 VGPRs and leave one wave per SIMD)
    __syncthreads();
    const int XG = S >> 2;
    const long long ngroups = (long long)B * S * S * XG;
    for (long long gi = (long long)blockIdx.x * 16 + gl; gi < ngroups; gi += (long long)gridDim.x * 16) {
        const int xg = (int)(gi % XG);
        const long long row = gi / XG;
        const int h = (int)(row % S);
        const int d = (int)((row / S) % S);
        const long long b = row / ((long long)S * S);
        const int x0 = xg * 4;
        const float* dqb = dq + b * S * S * S;
        float4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d >= 1 && d <= S - 2 && h >= 1 && h <= S - 2 && x0 >= 1 && x0 + 4 <= S - 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) {
                    const float* rp = dqb + ((long long)(d + 1 - a) * S + (h + 1 - bb)) * S + x0 - 1;
                    float g[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) g[j] = rp[j];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc)
                        {
                        const float4 wv = *reinterpret_cast<const float4*>(&sw[((a * 3 + bb) * 3 + cc) * 64 + c4]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) c1_fma4(acc[i], g[i + 2 - cc], wv);   // o_x = x0+i+1-cc
                    }
                }
        } else {
            int od[3], td[3], oh[3], th[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int o = d + 1 - k; td[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); od[k] = c1_clampi(o, 0, S - 1);
                o = h + 1 - k;     th[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); oh[k] = c1_clampi(o, 0, S - 1);
            }
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + i;
                int ox[3], tx[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { const int o = x + 1 - k; tx[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); ox[k] = c1_clampi(o, 0, S - 1); }
                float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            const float g = dqb[((long long)od[a] * S + oh[bb]) * S + ox[cc]];
                            const int t = (td[a] * 3 + th[bb]) * 3 + tx[cc];
                            c1_fma4(a4, g, *reinterpret_cast<const float4*>(&sw[t * 64 + c4]));
                        }
                acc[i] = a4;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long o = (row * S + x0 + i) * 64 + c4;
            float4 r = acc[i];
            if (accumulate) {
                const float4 p = *reinterpret_cast<const float4*>(du + o);
                r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w;
            }
            if (mask) {
                const float4 uu = *reinterpret_cast<const float4*>(u + o);
                r.x = uu.x > 0.f ? r.x : r.x * slope; r.y = uu.y > 0.f ? r.y : r.y * slope;
                r.z = uu.z > 0.f ? r.z : r.z * slope; r.w = uu.w > 0.f ? r.w : r.w * slope;
            }
            *reinterpret_cast<float4*>(du + o) = r;
        }
    }
}

// The data gradient above with the two other things that happen to the same tensor around it folded in (perceiver :465-470,
// backward): the pooled-feature term of SpatialSoftmax3D + max pool of u (the formula of ss_bwd4_kernel in vox_ops.hip, same
// arithmetic per element) is ADDED here instead of being written to du by its own read-u / write-du pass, and the column sums
// of the finished du (= the bias gradient of the conv that produced u) are taken while du is in registers instead of by a
// colsum pass.  du = lrelu'(u) * ([du_in] + sum_t dq w + ss term); partB[block][64] = sum over the block's voxels of du.
// One sample per blockIdx.y, so the per-(sample, channel) softmax constants stay in registers.
__global__ void __launch_bounds__(256, 2) c1_dgrad4_ss_kernel(const float* __restrict__ dq, const float* __restrict__ w,
                                                           const float* __restrict__ u, float* __restrict__ du, int S,
                                                           int accumulate, float slope, const float* __restrict__ lin,
                                                           const float* __restrict__ stats, const float* __restrict__ out_ss,
                                                           const int* __restrict__ argmax, const float* __restrict__ g_ss,
                                                           const float* __restrict__ g_max, float* __restrict__ partB,
                                                           unsigned* __restrict__ part_amax) {
    __shared__ float sw[27 * 64];          // [t][c]
    __shared__ float red[16][64];
    // the 3 x 3 x 6 dq window of each of the 16 voxel groups in flight, through LDS: read straight from global memory its 54 loads sit
    // BEHIND the group's u loads on the in-order vector-memory counter, so the tap loop waited for the HBM round trip it was meant to
    // hide (round 5).  The 16 lanes of a group are in one wave: wave-level ordering is enough.
    __shared__ float sdq[16][56];
    __shared__ unsigned ramx[4];
    float amxf = 0.f;                       // largest |du| this thread wrote (one v_max_f32 with an |x| modifier per element)
    for (int i = threadIdx.x; i < 27 * 64; i += 256) sw[(i % 27) * 64 + i / 27] = w[i];
    const DivT divT(0.01f);
    const int c4 = (threadIdx.x & 15) * 4, gl = threadIdx.x >> 4;
    const int b = blockIdx.y;
    float m[4], inv_s[4], ex[4], ey[4], ez[4], gx[4], gy[4], gz[4], gm[4], bsum[4];
    int am[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int bc = b * 64 + c4 + e;
        m[e] = stats[2 * bc]; inv_s[e] = 1.0f / stats[2 * bc + 1];
        ex[e] = out_ss[3LL * bc]; ey[e] = out_ss[3LL * bc + 1]; ez[e] = out_ss[3LL * bc + 2];
        gx[e] = g_ss[3LL * bc]; gy[e] = g_ss[3LL * bc + 1]; gz[e] = g_ss[3LL * bc + 2];
        gm[e] = g_max[bc]; am[e] = argmax[bc];
        bsum[e] = 0.f;
    }
    __syncthreads();
    const int XG = S >> 2;
    const int ngroups = S * S * XG;
    const long long S3 = (long long)S * S * S;
    const float* dqb = dq + b * S3;
    const float* ub = u + b * S3 * 64 + c4;
    float* dub = du + b * S3 * 64 + c4;
    for (int gi = blockIdx.x * 16 + gl; gi < ngroups; gi += gridDim.x * 16) {
        const int xg = gi % XG;
        const int row = gi / XG;
        const int h = row % S;
        const int d = row / S;
        const int x0 = xg * 4;
        const bool interior = d >= 1 && d <= S - 2 && h >= 1 && h <= S - 2 && x0 >= 1 && x0 + 4 <= S - 1;
        float wv4[4];
        if (interior) {
            const int l16 = threadIdx.x & 15;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = l16 + 16 * q;                   // (a, bb, j) = (idx / 18, (idx / 6) % 3, idx % 6)
                wv4[q] = idx < 54 ? dqb[((long long)(d + 1 - idx / 18) * S + (h + 1 - (idx / 6) % 3)) * S + x0 - 1 + idx % 6] : 0.f;
            }
        }
        asm volatile("" ::: "memory");                      // the window loads are issued BEFORE the u loads (in-order return)
        float4 uu[4], old[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                       // (issued before the tap loop: in flight under its 432 FMAs)
            const long long o = ((long long)row * S + x0 + i) * 64;
            uu[i] = *reinterpret_cast<const float4*>(ub + o);
            old[i] = accumulate ? *reinterpret_cast<const float4*>(dub + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (interior) {
            {
                const int l16 = threadIdx.x & 15;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (l16 + 16 * q < 54) sdq[gl][l16 + 16 * q] = wv4[q];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) {
                    float g[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) g[j] = sdq[gl][(a * 3 + bb) * 6 + j];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const float4 wv = *reinterpret_cast<const float4*>(&sw[((a * 3 + bb) * 3 + cc) * 64 + c4]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) c1_fma4(acc[i], g[i + 2 - cc], wv);   // o_x = x0+i+1-cc
                    }
                }
        } else {
            int od[3], td[3], oh[3], th[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int o = d + 1 - k; td[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); od[k] = c1_clampi(o, 0, S - 1);
                o = h + 1 - k;     th[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); oh[k] = c1_clampi(o, 0, S - 1);
            }
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + i;
                int ox[3], tx[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { const int o = x + 1 - k; tx[k] = o < 0 ? 0 : (o > S - 1 ? 2 : k); ox[k] = c1_clampi(o, 0, S - 1); }
                float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            const float g = dqb[((long long)od[a] * S + oh[bb]) * S + ox[cc]];
                            const int t = (td[a] * 3 + th[bb]) * 3 + tx[cc];
                            c1_fma4(a4, g, *reinterpret_cast<const float4*>(&sw[t * 64 + c4]));
                        }
                acc[i] = a4;
            }
        }
        const float li = lin[d], lj = lin[h];
        float base[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) base[e] = gx[e] * (lj - ex[e]) + gy[e] * (li - ey[e]);     // meshgrid 'xy' quirk, as ss_bwd4
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = row * S + x0 + i;
            const float lk = lin[x0 + i];
            const float xs[4] = {uu[i].x, uu[i].y, uu[i].z, uu[i].w};
            const float od4[4] = {old[i].x, old[i].y, old[i].z, old[i].w};
            const float ac[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
            float r[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float l = divT(xs[e]);
                const float a = exp_v(l - m[e]) * inv_s[e];
                float g = divT(a * (base[e] + gz[e] * (lk - ez[e])));
                if (p == am[e]) g += gm[e];
                if (accumulate) g = od4[e] + g;
                float v = ac[e] + g;                        // (same two addends as c1_dgrad4 after ss_bwd4: bit-identical)
                v = xs[e] > 0.f ? v : v * slope;
                bsum[e] += v;
                r[e] = v;
                amxf = fmaxf(amxf, fabsf(v));
            }
            *reinterpret_cast<float4*>(dub + (long long)p * 64) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[gl][c4 + e] = bsum[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float sacc = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) sacc += red[g][threadIdx.x];
        partB[((long long)b * gridDim.x + blockIdx.x) * 64 + threadIdx.x] = sacc;
    }
    if (part_amax) {                        // (uniform) per-block maximum of |du| for the fp16 kernels that read du next
        unsigned amx = vxb_amax_word(amxf, (bsum[0] + bsum[1]) + (bsum[2] + bsum[3]));      // (fmaxf drops NaN: the column sums keep it)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
        if ((threadIdx.x & 63) == 0) ramx[threadIdx.x >> 6] = amx;
        __syncthreads();
        if (threadIdx.x == 0) part_amax[b * gridDim.x + blockIdx.x] = max(max(ramx[0], ramx[1]), max(ramx[2], ramx[3]));
    }
}

// part[blk][c*27 + t] = sum over the block's rows of dq[o] * u[clamp(o + t - 1)][c];  partB[blk] = sum dq
// 16 threads (4 channels each) walk one (b, d, h) row with a sliding 3x3x3 window of float4; 16 rows per pass.
__global__ void __launch_bounds__(256) c1_wgrad4_kernel(const float* __restrict__ u, const float* __restrict__ dq,
                                                        float* __restrict__ part, float* __restrict__ partB, int B, int S,
                                                        int rows_per_block) {
    __shared__ float red[4][64 * 27];
    __shared__ float redb[16];
    const int c4 = (threadIdx.x & 15) * 4, gl = threadIdx.x >> 4;
    float4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float accb = 0.f;
    const long long nrows = (long long)B * S * S;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    for (long long row = r0 + gl; row < r1; row += 16) {
        const int h = (int)(row % S);
        const int d = (int)((row / S) % S);
        const long long b = row / ((long long)S * S);
        const float* ub = u + b * S * S * S * 64 + c4;
        long long ro[9];
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
#pragma unroll
            for (int hh = 0; hh < 3; ++hh)
                ro[dd * 3 + hh] = ((long long)c1_clampi(d + dd - 1, 0, S - 1) * S + c1_clampi(h + hh - 1, 0, S - 1)) * S * 64;
        float4 win[9][3];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            win[r][1] = *reinterpret_cast<const float4*>(ub + ro[r]);
            win[r][0] = win[r][1];
            win[r][2] = *reinterpret_cast<const float4*>(ub + ro[r] + (long long)c1_clampi(1, 0, S - 1) * 64);
        }
        const float* dqr = dq + row * S;
        for (int x = 0; x < S; ++x) {
            const float g = dqr[x];
            accb += g;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                c1_fma4(acc[r * 3 + 0], g, win[r][0]);
                c1_fma4(acc[r * 3 + 1], g, win[r][1]);
                c1_fma4(acc[r * 3 + 2], g, win[r][2]);
            }
            const int nx = c1_clampi(x + 2, 0, S - 1);
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                win[r][0] = win[r][1];
                win[r][1] = win[r][2];
                win[r][2] = *reinterpret_cast<const float4*>(ub + ro[r] + (long long)nx * 64);
            }
        }
    }
    // fold the 4 row groups of a wave with two cross-lane steps, then the 4 waves through LDS (fixed order)
    const int wid = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        float4 v = acc[t];
        v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64); v.z += __shfl_xor(v.z, 16, 64); v.w += __shfl_xor(v.w, 16, 64);
        v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
        if ((threadIdx.x & 63) < 16) {
            red[wid][(c4 + 0) * 27 + t] = v.x; red[wid][(c4 + 1) * 27 + t] = v.y;
            red[wid][(c4 + 2) * 27 + t] = v.z; red[wid][(c4 + 3) * 27 + t] = v.w;
        }
    }
    if ((threadIdx.x & 15) == 0) redb[gl] = accb;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 27; i += 256)
        part[(long long)blockIdx.x * 64 * 27 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += redb[k];
        partB[blockIdx.x] = s;
    }
}

// q[o] = bias + sum_t sum_c u[clamp(o + t - 1)][c] * w[c][t]: 16 threads (4 channels each) walk one (b, d, h) row with a
// sliding 3x3x3 window of float4, weights as float4 from LDS, 4 cross-lane steps fold the 16 partial dots per voxel.
__global__ void __launch_bounds__(256) c1_fwd4_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ q, int B, int S) {
    __shared__ float sw[27 * 64];          // [t][c]
    for (int i = threadIdx.x; i < 27 * 64; i += 256) sw[(i % 27) * 64 + i / 27] = w[i];
    __syncthreads();
    const int c4 = (threadIdx.x & 15) * 4, gl = threadIdx.x >> 4;
    const float bv = bias[0];
    const long long nrows = (long long)B * S * S;
    for (long long row = (long long)blockIdx.x * 16 + gl; row < nrows; row += (long long)gridDim.x * 16) {
        const int h = (int)(row % S);
        const int d = (int)((row / S) % S);
        const long long b = row / ((long long)S * S);
        const float* ub = u + b * S * S * S * 64 + c4;
        long long ro[9];
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
#pragma unroll
            for (int hh = 0; hh < 3; ++hh)
                ro[dd * 3 + hh] = ((long long)c1_clampi(d + dd - 1, 0, S - 1) * S + c1_clampi(h + hh - 1, 0, S - 1)) * S * 64;
        float4 win[9][3];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            win[r][1] = *reinterpret_cast<const float4*>(ub + ro[r]);
            win[r][0] = win[r][1];
            win[r][2] = *reinterpret_cast<const float4*>(ub + ro[r] + (long long)c1_clampi(1, 0, S - 1) * 64);
        }
        for (int x = 0; x < S; ++x) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 9; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 wv = *reinterpret_cast<const float4*>(&sw[(r * 3 + k) * 64 + c4]);
                    const float4 a = win[r][k];
                    acc = fmaf(a.x, wv.x, acc); acc = fmaf(a.y, wv.y, acc); acc = fmaf(a.z, wv.z, acc); acc = fmaf(a.w, wv.w, acc);
                }
            acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
            if ((threadIdx.x & 15) == 0) q[row * S + x] = acc + bv;
            const int nx = c1_clampi(x + 2, 0, S - 1);
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                win[r][0] = win[r][1];
                win[r][1] = win[r][2];
                win[r][2] = *reinterpret_cast<const float4*>(ub + ro[r] + (long long)nx * 64);
            }
        }
    }
}

// dst[i] += sum_k part[k][i] over nb partial rows, i < n: 64 columns x 4 row lanes per block (fixed order)
__global__ void __launch_bounds__(256) c1_reduce_kernel(const float* __restrict__ part, int nb, int n, float* __restrict__ dst) {
    __shared__ float red[256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < n) {
        int r = rl;
        for (; r + 4 < nb; r += 8) { s0 += part[(long long)r * n + c]; s1 += part[(long long)(r + 4) * n + c]; }
        for (; r < nb; r += 4) s0 += part[(long long)r * n + c];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < n) dst[c] += (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
}

// out[blockIdx.x][64] = sum of rows [blockIdx.x * per, +per) of part [nb][64] (zeros for an empty range)
__global__ void __launch_bounds__(256) c1_rows_kernel(const float* __restrict__ part, int nb, int per, float* __restrict__ out) {
    __shared__ float red[256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int r0 = blockIdx.x * per, r1 = min(nb, r0 + per);
    float s0 = 0.f;
    for (int r = r0 + rl; r < r1; r += 4) s0 += part[(long long)r * 64 + cl];
    red[threadIdx.x] = s0;
    __syncthreads();
    if (rl == 0) out[blockIdx.x * 64 + cl] = (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
}

}  // namespace

int vxb_c1_fwd4_launch(const float* u, const float* w, const float* bias, float* q, int B, int S, hipStream_t st) {
    const long long nrows = (long long)B * S * S;
    const int grid = (int)((nrows + 15) / 16 > 8192 ? 8192 : (nrows + 15) / 16);
    hipLaunchKernelGGL(c1_fwd4_kernel, dim3(grid), dim3(256), 0, st, u, w, bias, q, B, S);
    return hipGetLastError() == hipSuccess ? VXB_OK : VXB_ELAUNCH;
}

// launchers used by the C entry points in vox_ops.hip when S % 4 == 0
int vxb_c1_dgrad4_launch(const float* dq, const float* w, const float* u, float* du, int B, int S, int accumulate, int mask,
                         float slope, hipStream_t st) {
    const long long ngroups = (long long)B * S * S * (S >> 2);
    const int grid = (int)((ngroups + 15) / 16 > 16384 ? 16384 : (ngroups + 15) / 16);
    hipLaunchKernelGGL(c1_dgrad4_kernel, dim3(grid), dim3(256), 0, st, dq, w, u, du, B, S, accumulate, mask, slope);
    return hipGetLastError() == hipSuccess ? VXB_OK : VXB_ELAUNCH;
}

// fused data gradient (see c1_dgrad4_ss_kernel); part_ws: (vxb_c1_dgrad_ss_blocks_per_sample(S) * B + 64) * 64 floats; dbias [64] ACCUMULATED
int vxb_c1_dgrad_ss_blocks_per_sample(int S) {
    const long long g = (long long)S * S * (S >> 2);
    return (int)((g + 15) / 16 > 1024 ? 1024 : (g + 15) / 16);
}
int vxb_c1_dgrad4_ss_launch(const float* dq, const float* w, const float* u, float* du, int B, int S, int accumulate, float slope,
                            const float* lin, const float* stats, const float* out_ss, const int* argmax, const float* g_ss,
                            const float* g_max, float* dbias, float* part_ws, unsigned* part_amax, hipStream_t st) {
    const int nbx = vxb_c1_dgrad_ss_blocks_per_sample(S);
    hipLaunchKernelGGL(c1_dgrad4_ss_kernel, dim3(nbx, B), dim3(256), 0, st, dq, w, u, du, S, accumulate, slope, lin, stats, out_ss,
                       argmax, g_ss, g_max, part_ws, part_amax);
    // two stages (fixed order): 64 workgroups fold nb / 64 partial rows each into the head of a second buffer, one more folds those
    const int nb = nbx * B, per = (nb + 63) / 64;
    float* part2 = part_ws + (size_t)nb * 64;
    hipLaunchKernelGGL(c1_rows_kernel, dim3(64), dim3(256), 0, st, part_ws, nb, per, part2);
    hipLaunchKernelGGL(c1_reduce_kernel, dim3(1), dim3(256), 0, st, part2, 64, 64, dbias);
    return hipGetLastError() == hipSuccess ? VXB_OK : VXB_ELAUNCH;
}

// out[64] += column sums of part[nrows][64], two fixed-order stages (64 workgroups fold nrows / 64 rows each into the 64 rows behind
// the partials, one more folds those)
int vxb_rows64_sum_launch(float* part, int nrows, float* out, hipStream_t st) {
    if (!part || !out || nrows < 1) return VXB_EARG;
    const int per = (nrows + 63) / 64;
    float* part2 = part + (size_t)nrows * 64;
    hipLaunchKernelGGL(c1_rows_kernel, dim3(64), dim3(256), 0, st, part, nrows, per, part2);
    hipLaunchKernelGGL(c1_reduce_kernel, dim3(1), dim3(256), 0, st, part2, 64, 64, out);
    return hipGetLastError() == hipSuccess ? VXB_OK : VXB_ELAUNCH;
}

// part_ws: (ceil(B*S*S / 128) * (64*27 + 1)) floats (smaller than the scalar kernel's requirement)
int vxb_c1_wgrad4_launch(const float* u, const float* dq, float* dw, float* db, float* part_ws, int B, int S, hipStream_t st) {
    const long long nrows = (long long)B * S * S;
    const int rpb = 128;
    const int nb = vxb_cdiv(nrows, rpb);
    float* pW = part_ws;
    float* pB = part_ws + (size_t)nb * 64 * 27;
    hipLaunchKernelGGL(c1_wgrad4_kernel, dim3(nb), dim3(256), 0, st, u, dq, pW, pB, B, S, rpb);
    hipLaunchKernelGGL(c1_reduce_kernel, dim3(27), dim3(256), 0, st, pW, nb, 64 * 27, dw);
    hipLaunchKernelGGL(c1_reduce_kernel, dim3(1), dim3(256), 0, st, pB, nb, 1, db);
    return hipGetLastError() == hipSuccess ? VXB_OK : VXB_ELAUNCH;
}
