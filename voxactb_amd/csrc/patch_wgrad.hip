// The patchify data gradient folded straight into the input conv's weight gradient (perceiver_lang_io.py:357-371 forward;
// backward of patchify = Conv3DBlock(k = stride = 5, replicate padding 2) into input_preprocess = Conv3DBlock(10 -> 64, k = 1)).
//
// d(d0) receives three terms: `final`'s data gradient, the pooled-feature term of d0's SpatialSoftmax3D, and the patchify data
// gradient.  The voxel grid is a detached input (agent :100), so d(d0) feeds NOTHING but dW_in [64][10] and db_in [64] -- and those
// are linear in d(d0):   dW_in[c][j] = sum_v lrelu'(d0[v][c]) * d(d0)[v][c] * vox[v][j].
// The patchify term used to travel as a tensor: a 1^3 GEMM with 8000 output columns wrote it on the padded 105^3 grid (4.7 GB, 1.7 ms)
// and vxb_pointwise_wgrad_ss3d_f32 gathered it back through the padding adjoint (4.7 of its 13.6 GB).  With non-overlapping patches
// every (patch p, tap t) pair is ONE voxel v = clamp(5 p + t - 2) (the replicate padding's adjoint is that clamp), so the term is
//     sum over (p, t) of  lrelu'(d0[v][c]) * G[p][t][c] * vox[v][j],      G[p][t][c] = sum_k dpatch[p][k] * Wp[k][c][t],
// and G never needs to exist outside registers: a workgroup owns one tap (its 64 x 64 weight slice stays in registers as MFMA B
// fragments) and streams over patches, 32 per wave and step: G tile by v_mfma_f32_32x32x16_f16 (a leaf of the backward pass: single
// fp16 products, dpatch scaled by a power of two taken on the device), then per accumulator row the voxel's LeakyReLU' mask
// and its 10 inputs.  Deterministic: per-workgroup partials in a fixed order, a second kernel adds them up.
// Round 5: the sum over voxels of (masked G) x (vox | 1) is itself a product with K = patches -- the G accumulator already holds a
// lane's 16 patch rows of one channel, i.e. an A fragment in k-major order -- so it runs on the matrix cores as well (single fp16
// products, G x 2^-4; it was 320 fp32 FMAs and 160 live input registers per lane and step: 113 VALU instructions per MFMA).
//
// Round 5: the same launch also takes the patchify WEIGHT gradient  dWp[tap][c][k] = sum_p d0[v(p, tap)][c] * dpatch[p][k]  (it was a
// second full read of the 4.1 GB d0 by the generic transposed-read kernel, 1.29 ms).  The d0 values a lane fetches for the LeakyReLU'
// mask -- patch rows r = 8 s .. 8 s + 7 of its half, channel 32 i + lm -- ARE the A fragment (M = c, K = patches) of
// v_mfma_f32_32x32x16_f16 for k-step s under the k-order "slot e of lane half h = patch (r & 3) + 8 (r >> 2) + 4 h, r = 8 s + e"; the B
// fragment is dpatch in the same order, read as two 8-byte columns of the wave's dpatch tile kept transposed in LDS (fetching it with 32
// more dword loads per lane and step made the launch 2.9 x slower: it is bound by the number of vector-memory instructions in flight).
// Single fp16 products like every weight gradient (a leaf), dpatch under the same device-side scale.  3.13 -> 1.32 ms per step for
// the two gradients together (round 4: 1.84 + 1.29).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int PCIN = 10;             // channels of the voxel grid

struct PgArgs {
    const float* dpatch;     // [B, G^3, 64] fp32 (after the patchify block's LeakyReLU')
    const u16* wt;           // fp16 [k^3][64 c][64 kout]: Wp[kout][c][tap] transposed per tap (ops.patch_dgrad_weights)
    const float* d0;         // [B, V^3, 64] output of the input conv (its sign = LeakyReLU')
    const float* vox;        // [B, V^3, 10]
    const float* scale;      // {s, 1 / s}: dpatch is multiplied by s before the conversion to half
    float* part;             // [k^3 * Z][64][11] partial sums (scaled by s)
    float* wpart;            // optional: [k^3][Z][64 c][64 kout] partial sums of the patchify weight gradient (scaled by s)
    int B, V, G, k, pad, Z;
    float slope;
    unsigned magic;          // floor(2^32 / G) + 1: p / G == __umulhi(p, magic) for p * G < 2^32 (host-checked)
};

constexpr int TLD = 36;             // u16 per row of the transposed dpatch tile (32 patches + 4 pad: 18 dwords, conflict-free b64 column reads)

__global__ void __launch_bounds__(256, 2) patch_wgrad_kernel(PgArgs g) {
    __shared__ float wred[4][32][33];
    __shared__ __attribute__((aligned(8))) u16 tdp[4][64 * TLD];      // per wave: this step's dpatch tile as fp16 [kout][patch]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, half = lane >> 5;
    const int tap = blockIdx.x, z = blockIdx.y;
    const int tw = tap % g.k, th = (tap / g.k) % g.k, td = tap / (g.k * g.k);
    const int G = g.G, V = g.V;
    const long long P = (long long)g.B * G * G * G;
    // B fragments of this tap: lane (col n = lm, k half) holds Wt[tap][32 j + n][16 ks + 8 half .. + 7]
    f16x8 bf[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            bf[j][ks] = *reinterpret_cast<const f16x8*>(g.wt + ((long long)(tap * 64 + 32 * j + lm)) * 64 + 16 * ks + 8 * half);
    const float sc = g.scale[0];
    const bool wg_on = g.wpart != nullptr;                          // (uniform)
    // running sums, all three as accumulators of v_mfma_f32_32x32x16_f16 with K = patches:
    //   iacc[i]      [c = 32 i + row][col j]    input conv: sum of lrelu'(d0) G x vox[j] (j < 10), of lrelu'(d0) G (j = 10: the bias), 2^-4 scaled
    //   wacc[i][jn]  [c = 32 i + row][kout = 32 jn + col]    patchify weight gradient: sum of d0 x dpatch
    // k-order of a 16-deep step s: slot e of lane half h = patch row r = 8 s + e of that half, i.e. patch (r & 3) + 8 (r >> 2) + 4 h -- the
    // order in which the G accumulator holds its 16 rows, so the masked G values go from accumulator to A fragment without a transposition
    f32x16 iacc[2], wacc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) iacc[i][r] = 0.f;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) wacc[i][jn][r] = 0.f;
    }
    // this workgroup's slice of the patches, 128 per step (32 per wave)
    const long long per = (P + g.Z - 1) / g.Z;
    const long long p_begin = (long long)z * per, p_end = min(P, p_begin + per);
    for (long long p0 = p_begin + wid * 32; p0 < p_end; p0 += 128) {
        // A fragments: lane (row = patch p0 + lm, k half)
        const long long pr = min(p0 + lm, P - 1);
        f16x8 af[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 u = *reinterpret_cast<const float4*>(g.dpatch + pr * 64 + 16 * ks + 8 * half);
            const float4 w = *reinterpret_cast<const float4*>(g.dpatch + pr * 64 + 16 * ks + 8 * half + 4);
            union { unsigned x[4]; f16x8 v; } t;
            t.x[0] = vxb_pack_f16(u.x * sc, u.y * sc); t.x[1] = vxb_pack_f16(u.z * sc, u.w * sc);
            t.x[2] = vxb_pack_f16(w.x * sc, w.y * sc); t.x[3] = vxb_pack_f16(w.z * sc, w.w * sc);
            af[ks] = t.v;
        }
        if (wg_on) {
            // the same tile transposed into this wave's LDS slab ([kout][patch], zero rows past the slice): the patchify weight gradient's
            // B operand (K = patches) is then two 8-byte column reads per fragment instead of 8 more global loads
            u16* tw_ = tdp[wid];
            const bool rowlive = p0 + lm < p_end;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { f16x8 v; u16 h[8]; } t;
                t.v = af[ks];
#pragma unroll
                for (int e = 0; e < 8; ++e) tw_[(16 * ks + 8 * half + e) * TLD + lm] = rowlive ? t.h[e] : (u16)0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bf[j][ks], acc[j], 0, 0, 0);
        }
        // acc[j][r] = G[patch p0 + (r & 3) + 8 (r >> 2) + 4 half][c = 32 j + lm]
        const unsigned pbase = (unsigned)p0 + 4u * (unsigned)half, pend = (unsigned)p_end;
        float yk[2][16], xk[16];                                    // d0 of the lane's 16 patch rows (channel 32 j + lm); vox column lm of the same voxels
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (no branch per row: rows past the slice re-read its last patch and contribute zero, so that the loads of all 16 rows
            // can be in flight together)
            const unsigned pu = pbase + (unsigned)((r & 3) + 8 * (r >> 2));
            const bool live = pu < pend;
            const unsigned p = live ? pu : pend - 1u;
            // (b, pd, ph, pw) of patch p by multiply-high divisions (64-bit % and / cost ~100 instructions each: the first version of
            // this kernel spent 3.7 ms on them)
            const unsigned q1 = __umulhi(p, g.magic), q2 = __umulhi(q1, g.magic), q3 = __umulhi(q2, g.magic);
            const int pw = (int)(p - q1 * (unsigned)G), ph = (int)(q1 - q2 * (unsigned)G), pd = (int)(q2 - q3 * (unsigned)G), b = (int)q3;
            const int vw = min(max(pw * g.k + tw - g.pad, 0), V - 1);
            const int vh = min(max(ph * g.k + th - g.pad, 0), V - 1);
            const int vd = min(max(pd * g.k + td - g.pad, 0), V - 1);
            const long long v = (long long)(((b * V + vd) * V + vh) * V + vw);          // < 2^31 (host-checked)
            // B operand of the input conv's product: column lm = vox channel lm (< 10), column 10 = 1 (the bias gradient), else 0
            xk[r] = lm < PCIN ? g.vox[v * PCIN + lm] : (lm == PCIN ? 1.0f : 0.0f);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float y = g.d0[v * 64 + 32 * j + lm];
                yk[j][r] = y;
                float d = live ? acc[j][r] : 0.f;
                d = y > 0.f ? d : d * g.slope;
                // 2^-4 of headroom: dpatch arrives scaled to max |dpatch| < 2^15 (ops: the operand scale of this launch) and G sums 64 products
                // with patchify weights, so |G| < 2^15 * 64 * max|W| -- below 65504 * 16 for max|W| < 0.5 (the released weights: 0.09); beyond
                // that vxb_sat_f16 clips the operand of dW_in | db (tests/test_patch_wgrad_gpu.py provokes it and checks the bound stated there)
                acc[j][r] = d * 0.0625f;
            }
        }
        // input conv: iacc[i] += d[:, c = 32 i + ..]^T vox  (two 16-deep k-steps over the 32 patches)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { unsigned x[4]; f16x8 v; } tx;
#pragma unroll
            for (int e = 0; e < 4; ++e) tx.x[e] = vxb_pack_f16(xk[8 * ks + 2 * e], xk[8 * ks + 2 * e + 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                union { unsigned x[4]; f16x8 v; } td_;
#pragma unroll
                for (int e = 0; e < 4; ++e) td_.x[e] = vxb_pack_f16(vxb_sat_f16(acc[i][8 * ks + 2 * e]), vxb_sat_f16(acc[i][8 * ks + 2 * e + 1]));
                iacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(td_.v, tx.v, iacc[i], 0, 0, 0);
            }
        }
        if (wg_on) {
            // patchify weight gradient of these 32 patches: A = d0 (the values fetched for the mask), B = dpatch in the same k-order: slots
            // 0-3 / 4-7 of lane half h in k-step s are the patches 16 s + 4 h + 0..3 / 16 s + 8 + 4 h + 0..3
            const u16* tr_ = tdp[wid];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 xa[2], gb[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { unsigned x[4]; f16x8 v; } ta;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ta.x[e] = vxb_pack_f16(vxb_sat_f16(yk[j][8 * ks + 2 * e]), vxb_sat_f16(yk[j][8 * ks + 2 * e + 1]));      // (NaN / inf stay non-finite: common.h)
                    xa[j] = ta.v;
                    union { uint2 q[2]; f16x8 v; } tb;
                    const u16* col = tr_ + (32 * j + lm) * TLD + 16 * ks + 4 * half;
                    tb.q[0] = *reinterpret_cast<const uint2*>(col);
                    tb.q[1] = *reinterpret_cast<const uint2*>(col + 8);
                    gb[j] = tb.v;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        wacc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[i], gb[jn], wacc[i][jn], 0, 0, 0);
            }
        }
    }
    // the four waves' sums in a fixed order through LDS, one 32 x 32 tile at a time
    float* out = g.part + ((long long)tap * g.Z + z) * 64 * (PCIN + 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) wred[wid][(r & 3) + 8 * (r >> 2) + 4 * half][lm] = iacc[i][r];
        __syncthreads();
        for (int q = tid; q < 32 * (PCIN + 1); q += 256) {
            const int row = q / (PCIN + 1), e = q - row * (PCIN + 1);
            out[(32 * i + row) * (PCIN + 1) + e] = 16.0f * ((wred[0][row][e] + wred[1][row][e]) + (wred[2][row][e] + wred[3][row][e]));
        }
    }
    if (wg_on) {
        float* wout = g.wpart + ((long long)tap * g.Z + z) * 64 * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) wred[wid][(r & 3) + 8 * (r >> 2) + 4 * half][lm] = wacc[i][jn][r];
                __syncthreads();
                for (int q = tid; q < 32 * 32; q += 256) {
                    const int row = q >> 5, col = q & 31;
                    wout[(32 * i + row) * 64 + 32 * jn + col] = (wred[0][row][col] + wred[1][row][col]) + (wred[2][row][col] + wred[3][row][col]);
                }
            }
    }
}

// dW[c][j] += inv * sum_n part[n][c][j], db[c] += inv * sum_n part[n][c][10]: one thread per output, partials in index order
__global__ void __launch_bounds__(256) patch_wgrad_finish_kernel(const float* __restrict__ part, int n, const float* __restrict__ scale,
                                                                 float* __restrict__ dW, float* __restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * (PCIN + 1)) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < n; k += 4) {
        s0 += part[(long long)k * 704 + i]; s1 += part[(long long)(k + 1) * 704 + i];
        s2 += part[(long long)(k + 2) * 704 + i]; s3 += part[(long long)(k + 3) * 704 + i];
    }
    for (; k < n; ++k) s0 += part[(long long)k * 704 + i];
    const float s = ((s0 + s1) + (s2 + s3)) * scale[1];
    const int c = i / (PCIN + 1), e = i - c * (PCIN + 1);
    if (e < PCIN) dW[c * PCIN + e] += s; else db[c] += s;
}

// dWp [kout][c][tap] (the parameter's layout, k^3 taps innermost) += inv * sum_z wpart[tap][z][c][kout], partials in index order
__global__ void __launch_bounds__(256) patch_wgrad_weight_finish_kernel(const float* __restrict__ wpart, int ntap, int Z, const float* __restrict__ scale,
                                                                        float* __restrict__ dWp) {
    const int i = blockIdx.x * 256 + threadIdx.x;                   // (tap, c, kout)
    if (i >= ntap * 64 * 64) return;
    const int kout = i & 63, c = (i >> 6) & 63, tap = i >> 12;
    float s = 0.f;
    for (int z = 0; z < Z; ++z) s += wpart[((long long)tap * Z + z) * 4096 + c * 64 + kout];
    dWp[((long long)kout * 64 + c) * ntap + tap] += s * scale[1];
}

// first level of a long partial list: workgroup (r, y) sums rows [r chunk, (r + 1) chunk) in a fixed order into out[r]
__global__ void __launch_bounds__(256) wgin_reduce_kernel(const float* __restrict__ part, int n, int chunk, float* __restrict__ out) {
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= 704) return;
    const int k1 = min(n, ((int)blockIdx.x + 1) * chunk);
    int k = blockIdx.x * chunk;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; k + 3 < k1; k += 4) {
        s0 += part[(long long)k * 704 + i]; s1 += part[(long long)(k + 1) * 704 + i];
        s2 += part[(long long)(k + 2) * 704 + i]; s3 += part[(long long)(k + 3) * 704 + i];
    }
    for (; k < k1; ++k) s0 += part[(long long)k * 704 + i];
    out[(long long)blockIdx.x * 704 + i] = (s0 + s1) + (s2 + s3);
}

}  // namespace

// part: n rows of 704 partial sums, followed by room for VXB_WGIN_FINISH_ROWS more (the first-level sums of a long list)
int vxb_wgin_finish_launch(float* part, int n, const float* scale, float* dW, float* db, hipStream_t st) {
    if (!part || !scale || !dW || !db || n < 1) return VXB_EARG;
    if (n > 64) {                           // two levels of about sqrt(n) rows each (a single thread per entry walks a level serially)
        int chunk = 8;
        while ((long long)chunk * chunk < n) ++chunk;
        chunk = max(chunk, vxb_cdiv(n, VXB_WGIN_FINISH_ROWS));
        const int rows = vxb_cdiv(n, chunk);
        float* lvl1 = part + (long long)n * 704;
        hipLaunchKernelGGL(wgin_reduce_kernel, dim3(rows, vxb_cdiv(704, 256)), dim3(256), 0, st, part, n, chunk, lvl1);
        part = lvl1;
        n = rows;
    }
    hipLaunchKernelGGL(patch_wgrad_finish_kernel, dim3(vxb_cdiv(704, 256)), dim3(256), 0, st, part, n, scale, dW, db);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" size_t vxb_patch_dgrad_input_wgrad_ws_floats(int k, int nsplit) {
    return ((size_t)k * k * k * (nsplit > 0 ? nsplit : 1) + VXB_WGIN_FINISH_ROWS) * 704;
}

// ... plus the patchify weight gradient's partial sums when dWp is asked for (4096 floats per (tap, split))
extern "C" size_t vxb_patch_wgrad_weight_ws_floats(int k, int nsplit) { return (size_t)k * k * k * (nsplit > 0 ? nsplit : 1) * 4096; }

// The patchify data gradient's contribution to the input conv's weight / bias gradient, without the data gradient tensor (see the head
// of this file): dpatch [B, G^3, 64] (gradient of the patchify block's pre-activation), wt_f16 = fp16 [k^3][64][64] with
// wt[t][c][kout] = Wp[kout][c][t] (t = (kd k + kh) k + kw), d0 [B, V^3, 64] = output of the input conv, vox [B, V^3, 10] its input,
// patches of size k at stride k with replicate padding `pad` (G = ceil((V + 2 pad - k) / k) + 1 patches per axis);
// scale = {s, 1 / s} on the device (vxb_absmax_scale_f32 of dpatch).  dW [64][10] and db [64] are ACCUMULATED.
// ws: vxb_patch_dgrad_input_wgrad_ws_floats(k, nsplit) floats.  dWp (optional, with ws_wp of vxb_patch_wgrad_weight_ws_floats(k, nsplit)
// floats): the patchify weight gradient [64 kout][64 c][k^3] in the parameter's own layout, ACCUMULATED, from the same pass over d0
// (network_utils.py:128-170 backward; single fp16 products).
extern "C" int vxb_patch_dgrad_input_wgrad_f32(const float* dpatch, const void* wt_f16, const float* d0, const float* vox, int B, int V,
                                               int G, int k, int pad, float slope, const float* scale, float* ws, int nsplit,
                                               float* dW, float* db, float* dWp, float* ws_wp, vxb_stream_t stream) {
    if (!dpatch || !wt_f16 || !d0 || !vox || !scale || !ws || !dW || !db || B < 1 || V < 1 || G < 1 || k < 1 || nsplit < 1) return VXB_EARG;
    if ((((uintptr_t)dpatch | (uintptr_t)wt_f16) & 15) || (((uintptr_t)vox) & 7) || nsplit > 65535) return VXB_ESIZE;
    if ((long long)B * V * V * V >= INT32_MAX || (long long)B * G * G * G * G >= (1ll << 32) || G < 2) return VXB_ESIZE;
    PgArgs g;
    g.magic = (unsigned)((1ull << 32) / (unsigned)G) + 1u;
    g.dpatch = dpatch; g.wt = (const u16*)wt_f16; g.d0 = d0; g.vox = vox; g.scale = scale; g.part = ws;
    g.B = B; g.V = V; g.G = G; g.k = k; g.pad = pad; g.Z = nsplit; g.slope = slope;
    if ((dWp != nullptr) != (ws_wp != nullptr)) return VXB_EARG;
    g.wpart = ws_wp;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(patch_wgrad_kernel, dim3(k * k * k, nsplit), dim3(256), 0, st, g);
    if (dWp)
        hipLaunchKernelGGL(patch_wgrad_weight_finish_kernel, dim3(vxb_cdiv((long long)k * k * k * 4096, 256)), dim3(256), 0, st, ws_wp, k * k * k, nsplit,
                           scale, dWp);
    return vxb_wgin_finish_launch(ws, k * k * k * nsplit, scale, dW, db, st);
}
