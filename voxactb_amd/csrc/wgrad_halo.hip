// LDS-halo weight gradient of a 3x3x3 stride-1 conv3d on the 16-bit matrix cores ('bf16', 'bf16x3' and 'fp16' products):
//     part[z][(tap, ci)][n] = sum over the z-th slice of voxel tiles of  x[clamp(pos + tap + off)][ci] * dY[pos][n]
// (the `final` conv of the Q-function, perceiver_lang_io.py:462, and the polyphase form of the decoder's up-conv,
// network_utils.py:245-250, whose dY is gathered from the fine grid by space-to-depth).
//
// The generic transposed-read kernel (wgrad_bf16.hip) gathers x once per tap (27x through L2) and sits at ~100 TF/s in
// bf16x3.  Here a workgroup owns 16 input channels x 64 output channels x ALL 27 taps and streams over 2x8x8 voxel
// tiles: per tile the 4x10x10 halo of x and the 128 dY rows are staged once (fp32 -> bf16 hi[/lo] planes), and every
// tap reads its A operand from the halo at a shifted address.  The reduction runs over voxels, so both operands go
// through ds_read_b64_tr_b16 (voxel-major LDS -> k-major fragments) into v_mfma_f32_16x16x32_bf16.
//   * wave w owns taps w, w+4, ... (7/7/7/6): 7 x 4 accumulator tiles of 16x16 = 112 VGPRs; the four dY fragments of
//     a k-step are read once and reused by all of a wave's taps.
//   * k-step = 32 voxels = 4 h-rows x 8 w of one d-plane; lane group q (16 lanes) holds h = hb + 2(q>>1) + r,
//     w = 4(q&1) + 0..3 for read r -- the 32 lanes served per LDS cycle then touch 8 consecutive voxels, which with
//     32-byte halo voxels and 160-byte dY rows hit 64 distinct banks (no padding of the halo needed).
//   * the next tile's global loads are issued before the current tile's MFMAs (15 float4 per thread in flight) --
//     there are no other global loads in the loop, so in-order retirement does not get in the way.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// voxel tile WTD x WTH x 8 = 128 voxels: 2x8x8 or 4x4x8 (the host picks the one that wastes fewer voxels on the grid edge:
// S = 100 -> 100x100x104 instead of 100x104x104, S = 20 -> 20x20x24 instead of 20x24x24)
constexpr int WTW = 8, XW = WTW + 2;
constexpr int DLD = 80;                               // u16 per dY row (64 + 16 pad): 40 dwords, 8 rows -> 8 bank octets
constexpr int DPL = 128 * DLD;                        // u16 per dY plane

}  // namespace

// (shared by the two translation units of this kernel: wgrad_halo.hip holds the 2x8x8 tiles and the C entries,
// wgrad_halo_t44.hip -- this file again under WH_T44_UNIT -- the 4x4x8 tiles; each is compiled with the instruction
// scheduler that measured faster for it, see build.py)
struct VxbWhArgs {
    const float* src0;
    const float* src1;
    const float* dy;
    float* part;
    int C0, C1, B, S_in, S_out, off, replicate;
    int N, Krows;
    long long ldy;
    int d2s_s, d2s_C;
    int ntd, nth, ntw;
    long long ntiles;
    int tiles_per_split;
    int dbg;                      // timing experiments only (vxb_debug_set_wgrad_halo_experiment); results are WRONG when != 0
    const unsigned* phase_mask;   // d2s: bit t of phase_mask[column block] clear -> that (tap, phase) weight block is
                                  // structurally zero (polyphase up-conv) and is neither computed nor stored
    const float* dy_scale;        // 'fp16' products: dY is multiplied by *dy_scale (a power of two on the device, chosen from
                                  // the tensor's largest magnitude: vxb_absmax_scale_f32) before the conversion; part = scale * dW
    // 5x5x5 kernels as eight shifted 3x3x3 blocks (nshift = 8, no d2s): grid.y = shift * (N / 64) + column block; shift bit a
    // (4: d, 2: h, 1: w) adds 2 to that axis' offset, and the shift's taps go to the rows shift_rows[shift * 27 + tap] of a
    // [out_taps][Ct][N] gradient (-1: the tap belongs to another shift and is neither computed nor stored)
    int nshift;
    const int* shift_rows;
    int tabn;                     // fp16 variants: entries per axis of the clamp tables (max(S_in, S_out) + 16), 0 = tables not usable
};
typedef VxbWhArgs WhArgs;

namespace {

// PM = product mode: 0 plain bf16, 1 bf16x3 (hi/lo planes, three MFMAs), 2 plain fp16 (11-bit mantissa, 5-bit exponent: the
// gradient operand is pre-scaled, the activation operand saturates at the largest half instead of becoming inf)
template <int PM>
__device__ __forceinline__ unsigned wh_pack2(float lo, float hi) {
    if (PM == 2) return vxb_pack_f16(__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f));
    return vxb_pack_bf16(lo, hi);
}

template <int PM>
__device__ __forceinline__ f32x4 wh_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    if (PM == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 wh_frag(const u16* p0, const u16* p1) {
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p1));
    return u.v;
}

// NCH = 16-channel chunks per workgroup.  1: 4 waves, two workgroups per CU.  2: 8 waves = two 4-wave groups that own one
// chunk each and SHARE the dY tile -- per MFMA 29 % less staging work (10 instead of 14 loads + conversions per thread and
// tile), one workgroup per CU.  (Predicted from the timing experiments below: - 9 %.  Measured: - 2.7 % dense, + 5 % with tap
// masks -- the second resident workgroup does hide part of the staging after all.)
template <int PM, int WTD, int WTH, int NCH>
__global__ void __launch_bounds__(256 * NCH, NCH == 1 ? 2 : 1) wgrad_halo_kernel(WhArgs g) {
    constexpr int X3 = PM == 1;
    constexpr int NTH = 256 * NCH;
    constexpr int XH = WTH + 2;                           // halo h extent; d extent WTD + 2, w extent 10
    constexpr int XSLOTS = (WTD + 2) * XH * XW;           // 400 / 360
    constexpr int XPL = XSLOTS * 16;                      // u16 per x plane
    constexpr int XF4 = XSLOTS * 4;                       // float4 per chunk halo
    constexpr int NXL = (NCH * XF4 + NTH - 1) / NTH;      // 7 / 6 float4 x loads per thread per tile
    constexpr int NDL = 128 * 16 / NTH;                   // 8 / 4 float4 dY loads per thread per tile
    constexpr int HB = WTH / 4;                           // k-steps (4 h-rows x 8 w) per d-plane
    constexpr bool FASTADDR = PM == 2;                    // table-driven load addresses (below): the issue-bound fp16 variants only
    // DB (fp16, two chunks per workgroup -- the shape every big launch of the step takes): TWO LDS stages.  With one stage all eight
    // waves convert and store a tile together (VALU, matrix pipe idle), then multiply it together (matrix pipe, VALU idle): measured at
    // B = 8 on the `final` shape the staging alone takes 3.41 ms, the MFMA loop alone 3.12 ms and the kernel 5.47 ms -- the two hardly
    // overlap.  With two stages the tile t + 1 is staged while tile t is multiplied, ONE barrier per tile, and the two waves of a SIMD
    // (wave w of chunk 0, wave w + 4 of chunk 1) run the two halves of a period in opposite order: one converts while the other multiplies.
    constexpr bool DB = PM == 2 && NCH == 2;
    constexpr int BUFU = (1 + X3) * (NCH * XPL + DPL);     // u16 per LDS stage
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* xs = smem;                                   // [NCH][1 + X3][XSLOTS][16]
    u16* ds = smem + NCH * (1 + X3) * XPL;            // [1 + X3][128][DLD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid8 = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform (scalar register)
    const int wid = wid8 & 3, wch = wid8 >> 2;        // tap group of this wave / which of the workgroup's chunks it owns
    const int Ct = g.C0 + g.C1;
    // Workgroup -> (channel chunk bx, column block by, tile slice bz).  The chunk blocks of one (by, bz) read the SAME dY
    // tiles; hardware places linear workgroup id L on XCD L % 8, so when the number of (by, bz) pairs is a multiple of 8
    // the ids are permuted to put all chunks of a pair on one XCD (same L2) and next to each other in launch order:
    // L = xcd + 8 * (chunk + G * pj)  <->  pair = pj * 8 + xcd.  Bijective; only speed depends on the placement.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int G = gridDim.x, P = gridDim.y * gridDim.z;
        if ((P & 7) == 0) {
            const int L = blockIdx.x + G * (blockIdx.y + gridDim.y * blockIdx.z);
            const int xcd = L & 7, j = L >> 3;
            bx = j % G;
            const int pair = (j / G) * 8 + xcd;
            by = pair % gridDim.y;
            bz = pair / gridDim.y;
        }
    }
    const int cb = (bx * NCH + wch) * 16;             // this wave's 16 input channels
    int shift = 0;
    if (g.nshift > 1) { const int ncb = g.N / 64; shift = by / ncb; by -= shift * ncb; }
    const int offd = g.off + ((shift & 4) ? 2 : 0), offh = g.off + ((shift & 2) ? 2 : 0), offw = g.off + ((shift & 1) ? 2 : 0);
    const int n0 = by * 64;                           // ... and the workgroup's 64 output channels
    const int S = g.S_out;
    // dY of a depth-to-space output: column block nb is one phase (d2s_C == 64) of the fine grid
    int rd = 0, rh = 0, rw = 0;
    if (g.d2s_s > 0) { const int ph = by; rw = ph % g.d2s_s; rh = (ph / g.d2s_s) % g.d2s_s; rd = ph / (g.d2s_s * g.d2s_s); }

    f32x4 acc[7][4];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int t_begin = bz * g.tiles_per_split;
    const int t_end = (int)min((long long)t_begin + g.tiles_per_split, g.ntiles);

    // the loads of a tile in flight: px / pd and okm (bit i: x load i is real data, not zero padding / a slot past the halo; bit 8 + i:
    // dY load i).  PF2 (fp16, two chunks per workgroup: 10 float4 per thread and tile): TWO sets, the loads of tiles t+1 and t+2 are in
    // flight while tile t is multiplied -- a tile's MFMAs take ~0.9 us, a load ~2.5 us to arrive, and one workgroup per CU has nobody
    // else to hide that (33.9 % pipe-busy, 36 % of the wave time waiting: profiles/r03_v4_sq_summary.txt)
    struct LoadSet { float4 px[NXL]; float4 pd[NDL]; unsigned okm; };
    constexpr bool PF2 = PM == 2 && NCH == 2;
    LoadSet ls0, ls1;
    // dY addressing without a branch on the layout: fine-grid extent, voxel step and phase offsets (1 / 0 without d2s)
    const int ds_ = g.d2s_s > 0 ? g.d2s_s : 1;
    const int Vf = S * ds_;
    const long long dy_row = g.d2s_s > 0 ? (long long)g.d2s_C : g.ldy;
    const float* __restrict__ dyb = g.dy + (g.d2s_s > 0 ? 0 : n0);
    const int Sm = g.S_in - 1;
    // Every load is unconditional (clamped address + a mask bit) and the per-thread slot decomposition is recomputed from an
    // opaque copy of the thread id on every call.  As loop invariants the ~40 values were hoisted out of the tile loop, kept
    // live across it and spilled -- and every scratch reload in here is an s_waitcnt vmcnt(0) that drains the prefetch
    // loads issued before it: the 15 loads of a tile arrived one at a time (half of the kernel's time was that wait).
    // ---- fast path of issue() for tiles whose halo and dY block lie inside the grid (no clamping, no padding, no ragged edge):
    // the address of load slot i is  (per-tile scalar base) + (per-slot constant offset).  The per-slot constants are computed
    // ONCE and parked in LDS (as registers they would be hoisted values that spill, see above); a tile then costs one LDS read
    // and one load per slot instead of ~35 VALU instructions of index arithmetic -- the fp16 variants of this kernel are
    // issue-bound on exactly that arithmetic (profiles/r03_v1_sq_summary.txt: 42-48 % of wave time issuing, matrix pipe 23-33 %).
    int* tabx = reinterpret_cast<int*>(smem + (DB ? 2 : 1) * BUFU);              // [NXL][NTH] element offset, -1 = slot past the halo
    int* tabd = tabx + NXL * NTH;                                                  // [NDL][NTH]
    // ... and for the tiles that DO touch the border of the grid (at S = 20 that is 66 of 75 tiles, at S = 100 a quarter): clamping and
    // padding are separable per axis, so three small tables per operand hold  clamp(j) * stride | invalid << 31  for every coordinate
    // j a halo / tile voxel can have, and a slot's address is the sum of three entries -- ~14 VALU instructions and 4 LDS reads per
    // slot instead of the ~35 of issue_slow().  tabp / tabq: the (d, h, w) position of a slot inside the halo / tile, packed.
    int* tabp = tabd + NDL * NTH;                                                  // [NXL][NTH]  pd << 27 | ph << 23 | pw << 19 | channel offset; -1 = no slot
    int* tabq = tabp + NXL * NTH;                                                  // [NDL][NTH]  od << 27 | oh << 23 | ow << 19 | n4
    int* axx = tabq + NDL * NTH;                                                   // [3][tabn] x: index j + 4
    int* ayy = axx + 3 * g.tabn;                                                   // [3][tabn] dY: index j
    // all chunks of this workgroup in one source (always, unless a chunk pair straddles the concatenation point)
    const bool one_src = ((bx * NCH) * 16 >= g.C0) == ((bx * NCH + NCH - 1) * 16 >= g.C0);
    const bool src_second = (bx * NCH) * 16 >= g.C0;
    const int Cs_f = src_second ? g.C1 : g.C0;
    const float* __restrict__ src_f = src_second ? g.src1 : g.src0;
    if (FASTADDR) {
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int e = tid + NTH * i;
            const int lch = (e >> 2) % NCH;             // (the NCH chunks of a voxel on adjacent lanes: 64 NCH contiguous bytes per voxel)
            int p = e / (4 * NCH);
            const int c4 = (e & 3) * 4;
            const bool ok = p < XSLOTS;
            p = min(p, XSLOTS - 1);
            const int cbl = (bx * NCH + lch) * 16;
            const int c0 = src_second ? cbl - g.C0 : cbl;
            const int hw = p % XW; p /= XW;
            const int hh = p % XH; p /= XH;
            tabx[i * NTH + tid] = ok ? ((p * g.S_in + hh) * g.S_in + hw) * Cs_f + c0 + c4 : -1;
        }
#pragma unroll
        for (int i = 0; i < NDL; ++i) {
            const int e = tid + NTH * i;
            const int pos = e >> 4, n4 = (e & 15) * 4;
            const int od = pos / (WTH * 8), oh = (pos >> 3) % WTH, ow = pos & 7;
            tabd[i * NTH + tid] = (int)((((long long)od * ds_ * Vf + oh * ds_) * Vf + ow * ds_) * dy_row) + n4;
            tabq[i * NTH + tid] = (od << 27) | (oh << 23) | (ow << 19) | n4;
        }
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int e = tid + NTH * i;
            const int lch = (e >> 2) % NCH;             // (the NCH chunks of a voxel on adjacent lanes: 64 NCH contiguous bytes per voxel)
            int p = e / (4 * NCH);
            const int c4 = (e & 3) * 4;
            const bool ok = p < XSLOTS;
            p = min(p, XSLOTS - 1);
            const int cbl = (bx * NCH + lch) * 16;
            const int c0 = src_second ? cbl - g.C0 : cbl;
            const int hw = p % XW; p /= XW;
            const int hh = p % XH; p /= XH;
            tabp[i * NTH + tid] = ok ? ((p << 27) | (hh << 23) | (hw << 19) | (c0 + c4)) : -1;
        }
        if (g.tabn > 0) {
            for (int j = tid; j < 3 * g.tabn; j += NTH) {
                const int ax = j / g.tabn, jj = j - ax * g.tabn;
                // x: coordinate jj - 4 (halo voxels reach from off to S + tile overhang + 1 + off)
                const int cx = jj - 4, cc = min(max(cx, 0), Sm);
                const int sx = ax == 0 ? g.S_in * g.S_in * Cs_f : (ax == 1 ? g.S_in * Cs_f : Cs_f);
                axx[j] = cc * sx | ((g.replicate || cc == cx) ? 0 : (int)0x80000000);
                // dY: coordinate jj (tile voxels reach from 0 to S + tile overhang)
                const int cy = min(jj, S - 1);
                const long long sy = ax == 0 ? (long long)ds_ * Vf * Vf * dy_row : (ax == 1 ? (long long)ds_ * Vf * dy_row : (long long)ds_ * dy_row);
                ayy[j] = (int)(cy * sy) | (jj < S ? 0 : (int)0x80000000);
            }
        }
        __syncthreads();           // (the axis tables are shared; tabx / tabd / tabp / tabq entries are read back only by their writer)
    }
    auto issue_fast = [&](LoadSet& L, int tile) __attribute__((always_inline)) -> bool {
        int t = tile;
        const int tw = t % g.ntw; t /= g.ntw;
        const int th = t % g.nth; t /= g.nth;
        const int td = t % g.ntd; t /= g.ntd;
        const int b = t;
        const int d0 = td * WTD, h0 = th * WTH, w0 = tw * WTW;
        // uniform: everything in range?
        const bool in = one_src && d0 + offd >= 0 && d0 + WTD + 1 + offd <= Sm && h0 + offh >= 0 && h0 + WTH + 1 + offh <= Sm &&
                        w0 + offw >= 0 && w0 + WTW + 1 + offw <= Sm && d0 + WTD <= S && h0 + WTH <= S && w0 + WTW <= S;
        if (!in) return false;
        const long long xb = ((((long long)b * g.S_in + d0 + offd) * g.S_in + h0 + offh) * g.S_in + w0 + offw) * Cs_f;
        const float* __restrict__ xp = src_f + xb;
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int o = tabx[i * NTH + tid];
            m |= (o >= 0 ? 1u : 0u) << i;
            L.px[i] = *reinterpret_cast<const float4*>(xp + max(o, 0));
        }
        const long long db = ((((long long)b * Vf + (long long)d0 * ds_ + rd) * Vf + h0 * ds_ + rh) * Vf + w0 * ds_ + rw) * dy_row;
        const float* __restrict__ dp = dyb + db;
#pragma unroll
        for (int i = 0; i < NDL; ++i) {
            L.pd[i] = *reinterpret_cast<const float4*>(dp + tabd[i * NTH + tid]);
            m |= 1u << (8 + i);
        }
        L.okm = m;
        return true;
    };
    auto issue_mid = [&](LoadSet& L, int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tw = t % g.ntw; t /= g.ntw;
        const int th = t % g.nth; t /= g.nth;
        const int td = t % g.ntd; t /= g.ntd;
        const int b = t;
        const int d0 = td * WTD, h0 = th * WTH, w0 = tw * WTW;
        const float* __restrict__ xp = src_f + (long long)b * g.S_in * g.S_in * g.S_in * Cs_f;
        const int* ad = axx + d0 + offd + 4;
        const int* ah = axx + g.tabn + h0 + offh + 4;
        const int* aw = axx + 2 * g.tabn + w0 + offw + 4;
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int pk = tabp[i * NTH + tid];
            const int pu = max(pk, 0);
            const int a = ad[(pu >> 27) & 15], bb = ah[(pu >> 23) & 15], c = aw[(pu >> 19) & 15];
            m |= (((unsigned)~(a | bb | c | pk)) >> 31) << i;
            L.px[i] = *reinterpret_cast<const float4*>(xp + (((a + bb + c) & 0x7fffffff) + (pu & 0x7ffff)));
        }
        const float* __restrict__ dp = dyb + ((((long long)b * Vf + rd) * Vf + rh) * Vf + rw) * dy_row;
        const int* yd = ayy + d0;
        const int* yh = ayy + g.tabn + h0;
        const int* yw = ayy + 2 * g.tabn + w0;
#pragma unroll
        for (int i = 0; i < NDL; ++i) {
            const int pk = tabq[i * NTH + tid];
            const int a = yd[(pk >> 27) & 15], bb = yh[(pk >> 23) & 15], c = yw[(pk >> 19) & 15];
            m |= (((unsigned)~(a | bb | c)) >> 31) << (8 + i);
            L.pd[i] = *reinterpret_cast<const float4*>(dp + (((a + bb + c) & 0x7fffffff) + (pk & 0x7ffff)));
        }
        L.okm = m;
    };
    auto issue_slow = [&](LoadSet& L, int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tw = t % g.ntw; t /= g.ntw;
        const int th = t % g.nth; t /= g.nth;
        const int td = t % g.ntd; t /= g.ntd;
        const int b = t;
        const int d0 = td * WTD, h0 = th * WTH, w0 = tw * WTW;
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int e = tid_ + NTH * i;
            const int lch = (e >> 2) % NCH;                              // which chunk of the workgroup this slot belongs to
            int p = e / (4 * NCH);
            const int c4 = (e & 3) * 4;
            bool ok = p < XSLOTS;
            p = min(p, XSLOTS - 1);
            const int cbl = (bx * NCH + lch) * 16;                       // first channel of that chunk: source 0 or source 1
            const bool second = cbl >= g.C0;
            const float* __restrict__ src = second ? g.src1 : g.src0;
            const int Cs = second ? g.C1 : g.C0;
            const int c0 = second ? cbl - g.C0 : cbl;
            const int hw = p % XW; p /= XW;
            const int hh = p % XH; p /= XH;
            const int id = d0 + p + offd, ih = h0 + hh + offh, iw = w0 + hw + offw;
            const int cd = min(max(id, 0), Sm), ch = min(max(ih, 0), Sm), cw = min(max(iw, 0), Sm);
            ok = ok && (g.replicate || (cd == id && ch == ih && cw == iw));
            m |= (ok ? 1u : 0u) << i;
            const int vox = ((b * g.S_in + cd) * g.S_in + ch) * g.S_in + cw;
            L.px[i] = *reinterpret_cast<const float4*>(src + (long long)vox * Cs + c0 + c4);
        }
#pragma unroll
        for (int i = 0; i < NDL; ++i) {
            const int e = tid_ + NTH * i;
            const int pos = e >> 4, n4 = (e & 15) * 4;
            const int od = d0 + pos / (WTH * 8), oh = h0 + ((pos >> 3) % WTH), ow = w0 + (pos & 7);
            const bool ok = od < S && oh < S && ow < S;
            m |= (ok ? 1u : 0u) << (8 + i);
            const int vox = ((b * Vf + min(od, S - 1) * ds_ + rd) * Vf + min(oh, S - 1) * ds_ + rh) * Vf + min(ow, S - 1) * ds_ + rw;
            L.pd[i] = *reinterpret_cast<const float4*>(dyb + (long long)vox * dy_row + n4);
        }
        L.okm = m;
    };
    const bool mid_ok = FASTADDR && one_src && g.tabn > 0;           // (uniform)
    auto issue = [&](LoadSet& L, int tile) __attribute__((always_inline)) {
        if (FASTADDR && issue_fast(L, tile)) return;
        if (mid_ok) issue_mid(L, tile);
        else issue_slow(L, tile);
    };
    const float dysc = (PM == 2 && g.dy_scale) ? *g.dy_scale : 1.0f;
    auto stage = [&](LoadSet& L, const int bo) __attribute__((always_inline)) {      // bo: u16 offset of the LDS stage
        int tid_s = tid;                  // (opaque copy: the slot arithmetic below is NOT to be hoisted out of the tile loop and kept live
        asm volatile("" : "+v"(tid_s));   //  across the MFMAs -- with two LDS stages that cost 43-75 spilled VGPRs)
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int e0 = tid_s + NTH * i;
            if (e0 < NCH * XF4 && !(g.dbg & 4)) {          // (experiment bit 4: no x conversion / LDS stores)
                const int lch = (e0 >> 2) % NCH;
                const int e = (e0 / (4 * NCH)) * 4 + (e0 & 3) + lch * (1 + X3) * XPL / 4;     // slot index inside the chunk's plane pair
                if (!((L.okm >> i) & 1u)) L.px[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 pk;
                pk.x = wh_pack2<PM>(L.px[i].x, L.px[i].y); pk.y = wh_pack2<PM>(L.px[i].z, L.px[i].w);
                *reinterpret_cast<uint2*>(&xs[bo + (e >> 2) * 16 + (e & 3) * 4]) = pk;
                if (X3) {
                    uint2 q;
                    q.x = wh_pack2<PM>(L.px[i].x - __uint_as_float(pk.x << 16), L.px[i].y - __uint_as_float(pk.x & 0xffff0000u));
                    q.y = wh_pack2<PM>(L.px[i].z - __uint_as_float(pk.y << 16), L.px[i].w - __uint_as_float(pk.y & 0xffff0000u));
                    *reinterpret_cast<uint2*>(&xs[bo + XPL + (e >> 2) * 16 + (e & 3) * 4]) = q;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (((g.dbg & 8) != 0) ? 0 : NDL); ++i) {      // (experiment bit 8: no dY conversion / LDS stores)
            const int e = tid_s + NTH * i;
            if (!((L.okm >> (8 + i)) & 1u)) L.pd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PM == 2) { L.pd[i].x *= dysc; L.pd[i].y *= dysc; L.pd[i].z *= dysc; L.pd[i].w *= dysc; }
            uint2 pk;
            pk.x = wh_pack2<PM>(L.pd[i].x, L.pd[i].y); pk.y = wh_pack2<PM>(L.pd[i].z, L.pd[i].w);
            *reinterpret_cast<uint2*>(&ds[bo + (e >> 4) * DLD + (e & 15) * 4]) = pk;
            if (X3) {
                uint2 q;
                q.x = wh_pack2<PM>(L.pd[i].x - __uint_as_float(pk.x << 16), L.pd[i].y - __uint_as_float(pk.x & 0xffff0000u));
                q.y = wh_pack2<PM>(L.pd[i].z - __uint_as_float(pk.y << 16), L.pd[i].w - __uint_as_float(pk.y & 0xffff0000u));
                *reinterpret_cast<uint2*>(&ds[bo + DPL + (e >> 4) * DLD + (e & 15) * 4]) = q;
            }
        }
    };

    // fragment addressing: lane group q = lane >> 4, t = lane & 15 -> voxel (h = hb + 2(q>>1) + r, w = 4(q&1) + (t>>2)),
    // channel quad t & 3 (ds_read_b64_tr_b16 then hands lane i channel i at the group's 4 voxels)
    const int q = lane >> 4, tl = lane & 15;
    const int fh = 2 * (q >> 1), fw = 4 * (q & 1) + (tl >> 2), fc = 4 * (tl & 3);

    // this wave's taps: the (wid + 4 ti)-th ACTIVE tap of the column block (all 27, or the phase's footprint); slots past
    // the end repeat the last active tap and are never stored
    unsigned tmask = (g.phase_mask && g.d2s_s > 0) ? (g.phase_mask[by] & 0x7ffffffu) : 0x7ffffffu;
    if (g.nshift > 1) {
        tmask = 0;
        for (int tp = 0; tp < 27; ++tp) tmask |= (g.shift_rows[shift * 27 + tp] >= 0 ? 1u : 0u) << tp;     // (uniform scalar loads)
    }
    const int nact = __builtin_popcount(tmask);
    const int nti = (nact + 3) >> 2;     // tap slots every wave runs (workgroup-uniform)
    int toffs[7];                        // halo offsets (u16) of this wave's taps; wave-uniform
    int wtap[7];
#pragma unroll
    for (int ti = 0; ti < 7; ++ti) {
        int idx = min(wid + 4 * ti, nact - 1);
        unsigned m = tmask;
        for (int z = 0; z < idx; ++z) m &= m - 1;
        const int tap = __builtin_ctz(m);
        wtap[ti] = (wid + 4 * ti < nact) ? tap : -1;
        toffs[ti] = (((tap / 9) * XH + (tap / 3) % 3) * XW + tap % 3) * 16;
    }

    // (Timing experiments, round 2, B = 4, S = 100, 4x4x8 tiles, 'bf16x3': 4.75 ms as is; 3.25 ms with the staging of all but the
    // first tile removed (g.dbg & 1: 545 TF/s, the ceiling of this MFMA loop); 1.70 ms with the MFMA loop removed (g.dbg & 2).
    // The VALU work of issue() + stage() (~650 instructions per thread and tile) is not hidden behind the other resident
    // workgroup's MFMAs -- the times add.  Delaying every second workgroup by 1-8 thousand cycles to break a possible
    // lockstep of the two changed nothing.)
    // one tile: stage the loads of set L into LDS, issue the loads of tile + AHEAD into the freed set, multiply
    // the MFMAs of one tile, operands in the LDS stage at u16 offset bo
    auto mma = [&](const int bo) __attribute__((always_inline)) {
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            const int dd = ks / HB, hb = (ks % HB) * 4;
            const u16* xa0 = xs + bo + wch * (1 + X3) * XPL + ((dd * XH + hb + fh) * XW + fw) * 16 + fc;   // read r = 0 (tap offset added later)
            const u16* xa1 = xa0 + XW * 16;                                          // r = 1: next h row
            const u16* db0 = ds + bo + ((dd * WTH + hb + fh) * WTW + fw) * DLD + fc;
            const u16* db1 = db0 + WTW * DLD;
            bf16x8 bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = wh_frag(db0 + 16 * j, db1 + 16 * j);
                if (X3) bl[j] = wh_frag(db0 + DPL + 16 * j, db1 + DPL + 16 * j);
            }
            // The A fragments of tap ti+1 are read while the MFMAs of tap ti run (a wave's slot past the end re-reads the last
            // active tap and its accumulator is never stored): the hi fragment into a second register set before the tap's
            // first MFMA, the lo fragment back into its own registers as soon as the tap's four lo * hi products have issued
            // -- 12 / 8 MFMAs of cover for the LDS round trip at 4 extra VGPRs (a full second set of both does not fit in 256
            // without spilling; without the barriers the scheduler sinks the reads below the MFMAs and every tap starts on
            // s_waitcnt lgkmcnt(0)).
            bf16x8 ah = wh_frag(xa0 + toffs[0], xa1 + toffs[0]);
            bf16x8 al = X3 ? wh_frag(xa0 + XPL + toffs[0], xa1 + XPL + toffs[0]) : ah;
#pragma unroll
            for (int ti = 0; ti < 7; ++ti) {
                if (ti >= nti) break;    // uniform: a phase with 8 / 12 / 18 active taps runs 2 / 3 / 5 slots
                bf16x8 ahn = ah;
                if (ti + 1 < 7) ahn = wh_frag(xa0 + toffs[ti + 1], xa1 + toffs[ti + 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (X3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ti][j] = wh_mfma<PM>(al, bh[j], acc[ti][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ti + 1 < 7) al = wh_frag(xa0 + XPL + toffs[ti + 1], xa1 + XPL + toffs[ti + 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ti][j] = wh_mfma<PM>(ah, bl[j], acc[ti][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ti][j] = wh_mfma<PM>(ah, bh[j], acc[ti][j]);
                ah = ahn;
            }
        }
    };
    auto tile_body = [&](LoadSet& L, int tile, int ahead) __attribute__((always_inline)) {
        __syncthreads();                 // every wave is done reading the previous tile
        if (!(g.dbg & 1) || tile == t_begin) stage(L, 0);
        __syncthreads();
        if (tile + ahead < t_end && !(g.dbg & 1)) issue(L, tile + ahead);
        if (g.dbg & 2) return;
        mma(0);
    };
    if (DB) {
        // One loop body for all waves -- multiply tile t, then convert + store a later tile, one s_barrier per tile -- but the chunk-0
        // waves ("lead") run one tile further ahead in the staging and take their barrier BETWEEN the two halves, the chunk-1 waves
        // after both: between two barriers a lead wave converts tile t + 2 and then multiplies tile t + 1 while its SIMD partner
        // multiplies tile t + 1 and then converts tile t + 2.  Stage t & 1 of the LDS holds tile t; a lead wave overwrites it with tile
        // t + 2 right after the barrier that every wave reaches only after its MFMAs of tile t.  Set X (ls1 in even iterations) holds
        // the loads of the tile staged next: t + 1 (chunk 1) / t + 2 (lead), refilled with the tile two further on.
        const int lead = wch == 0 ? 1 : 0;      // wave-uniform
        if (t_begin < t_end) issue(ls0, t_begin);
        if (t_begin + 1 < t_end) issue(ls1, t_begin + 1);
        if (t_begin < t_end) stage(ls0, 0);
        if (lead) {
            if (t_begin + 1 < t_end) stage(ls1, BUFU);
            if (!(g.dbg & 1)) {
                if (t_begin + 2 < t_end) issue(ls1, t_begin + 2);
                if (t_begin + 3 < t_end) issue(ls0, t_begin + 3);
            }
        } else if (t_begin + 2 < t_end && !(g.dbg & 1)) {
            issue(ls0, t_begin + 2);
        }
        __syncthreads();
        auto period = [&](LoadSet& X, int tile, const int par) __attribute__((always_inline)) {      // par = (tile - t_begin) & 1
            if (!(g.dbg & 2)) mma(par * BUFU);
            if (lead) vxb_raw_barrier_lds();
            const int nt = tile + 1 + lead;
            if (nt < t_end && !(g.dbg & 1)) {
                stage(X, ((par + 1 + lead) & 1) * BUFU);
                if (nt + 2 < t_end && !(g.dbg & 16)) issue(X, nt + 2);      // (experiment bit 16: no loads after the prologue's)
            }
            if (!lead) vxb_raw_barrier_lds();
        };
        for (int tile = t_begin; tile < t_end; tile += 2) {
            period(ls1, tile, 0);
            if (tile + 1 < t_end) period(ls0, tile + 1, 1);
        }
    } else if (PF2) {
        if (t_begin < t_end) issue(ls0, t_begin);
        if (t_begin + 1 < t_end) issue(ls1, t_begin + 1);
        for (int tile = t_begin; tile < t_end; tile += 2) {
            tile_body(ls0, tile, 2);
            if (tile + 1 < t_end) tile_body(ls1, tile + 1, 2);
        }
    } else {
        if (t_begin < t_end) issue(ls0, t_begin);
        for (int tile = t_begin; tile < t_end; ++tile) tile_body(ls0, tile, 1);
    }

    // D tile (16 ci x 16 n): lane l holds column n = l & 15, rows 4 (l >> 4) + r
    float* __restrict__ C = g.part + (long long)bz * g.Krows * g.N;
#pragma unroll
    for (int ti = 0; ti < 7; ++ti) {
        const int tap = wtap[ti];
        if (tap >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (g.nshift > 1 ? g.shift_rows[shift * 27 + tap] : tap) * Ct + cb + 4 * q + r;
                    C[(long long)row * g.N + n0 + 16 * j + tl] = acc[ti][j][r];
                }
        }
    }
}

}  // namespace

template <int PM, int TD, int TH, int NCH>
static int wgrad_halo_launch(WhArgs& g, int nsplit, hipStream_t st) {
    constexpr int X3 = PM == 1;
    g.ntd = vxb_cdiv(g.S_out, TD); g.nth = vxb_cdiv(g.S_out, TH); g.ntw = vxb_cdiv(g.S_out, WTW);
    g.ntiles = (long long)g.B * g.ntd * g.nth * g.ntw;
    if (g.ntiles >= INT32_MAX) return VXB_ESIZE;
    g.tiles_per_split = (int)((g.ntiles + nsplit - 1) / nsplit);
    constexpr int NTHR = 256 * NCH;
    constexpr int XF4_ = (TD + 2) * (TH + 2) * XW * 4;
    // fp16 variants: per-slot tables (offsets + packed positions) and the per-axis clamp tables (wgrad_halo_kernel: issue_fast / issue_mid)
    g.tabn = 0;
    if (PM == 2) {
        const long long vf = (long long)g.S_out * (g.d2s_s > 0 ? g.d2s_s : 1);
        const long long xspan = (long long)g.S_in * g.S_in * g.S_in * (g.C0 > g.C1 ? g.C0 : g.C1);
        const long long yspan = vf * vf * vf * (g.d2s_s > 0 ? g.d2s_C : g.ldy);
        const int off_lo = g.off, off_hi = g.off + (g.nshift > 1 ? 2 : 0);
        // coordinates off_lo .. S_out + 7 + 1 + off_hi must fit the table (index + 4) and the packed fields (< 16 per axis), sums < 2^30
        if (xspan < (1ll << 30) && yspan < (1ll << 30) && off_lo >= -4 && off_hi <= 4 && (g.C0 + g.C1) < (1 << 19) && g.N < (1 << 19))
            g.tabn = (g.S_in > g.S_out ? g.S_in : g.S_out) + 20;
    }
    const size_t lds = (size_t)((PM == 2 && NCH == 2) ? 2 : 1) * (1 + X3) * (NCH * (TD + 2) * (TH + 2) * XW * 16 + DPL) * sizeof(u16) +
                       (PM == 2 ? (size_t)2 * ((NCH * XF4_ + NTHR - 1) / NTHR + 128 * 16 / NTHR) * NTHR * sizeof(int) + (size_t)6 * g.tabn * sizeof(int) : 0);
    dim3 grid((g.C0 + g.C1) / (16 * NCH), (g.N / 64) * (g.nshift > 1 ? g.nshift : 1), nsplit);
    if (hipFuncSetAttribute((const void*)wgrad_halo_kernel<PM, TD, TH, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VXB_ELAUNCH;
    hipLaunchKernelGGL((wgrad_halo_kernel<PM, TD, TH, NCH>), grid, dim3(256 * NCH), lds, st, g);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

#ifdef WH_T44_UNIT
int vxb_wgrad_halo_launch_t44(VxbWhArgs& g, int pm, int nch, int nsplit, hipStream_t st) {
    if (pm == 2) return nch == 2 ? wgrad_halo_launch<2, 4, 4, 2>(g, nsplit, st) : wgrad_halo_launch<2, 4, 4, 1>(g, nsplit, st);
    if (nch == 2) return pm ? wgrad_halo_launch<1, 4, 4, 2>(g, nsplit, st) : wgrad_halo_launch<0, 4, 4, 2>(g, nsplit, st);
    return pm ? wgrad_halo_launch<1, 4, 4, 1>(g, nsplit, st) : wgrad_halo_launch<0, 4, 4, 1>(g, nsplit, st);
}
#else
int vxb_wgrad_halo_launch_t44(VxbWhArgs& g, int pm, int nch, int nsplit, hipStream_t st);
static int g_wh_nch = 0;          // experiment knob (vxb_debug_set_wgrad_halo_chunks): 0 = default, 1 / 2 = force

static int g_wh_dbg = 0;
static int g_wh_shape = -1;       // experiment knob (vxb_debug_set_wgrad_halo_shape): -1 = choose per grid, 0 / 1 = force

// tile shape for a grid of extent S: 1 -> 4x4x8, 0 -> 2x8x8.  Per tile the 4x4x8 kernel is the faster one in both precisions
// (its halo is 360 instead of 400 voxels; B = 4, S = 100, tools/bench_wgrad_halo.py: 378 vs 348 TF/s in 'bf16x3', 694 vs 601 in
// 'bf16'), so it is taken unless it pads the grid more (S = 100: 100x100x104 vs 100x104x104, S = 20: 20x20x24 vs 20x24x24).
static inline int wgrad_halo_shape(int S, int x3) {
    (void)x3;
    if (g_wh_shape >= 0) return g_wh_shape;
    const long long a = (long long)vxb_cdiv(S, 2) * 2 * vxb_cdiv(S, 8) * 8, b = (long long)vxb_cdiv(S, 4) * 4 * vxb_cdiv(S, 4) * 4;
    return b <= a;
}

static int wgrad_halo_impl(int pm, const float* dy_scale, const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out, int off,
                           int replicate, const float* dy, int N, int64_t ldy, int d2s_s, int d2s_C, float* part, int nsplit,
                           const uint32_t* phase_mask, vxb_stream_t stream, const int32_t* shift_rows = nullptr, int out_taps = 27) {
    if (!src0 || !dy || !part || B < 1 || S_in < 1 || S_out < 1 || N < 1 || nsplit < 1) return VXB_EARG;
    if ((C0 & 15) || (C1 & 15) || C0 < 16 || (C1 > 0 && !src1) || (N & 63)) return VXB_ESIZE;
    if (d2s_s > 0 && d2s_C != 64) return VXB_ESIZE;
    if (d2s_s <= 0 && ((ldy & 3) || phase_mask)) return VXB_ESIZE;
    WhArgs g;
    g.src0 = src0; g.src1 = src1; g.dy = dy; g.part = part; g.C0 = C0; g.C1 = C1; g.B = B; g.S_in = S_in; g.S_out = S_out;
    g.off = off; g.replicate = replicate; g.N = N; g.Krows = out_taps * (C0 + C1); g.ldy = ldy; g.d2s_s = d2s_s; g.d2s_C = d2s_C;
    g.phase_mask = phase_mask;
    g.nshift = shift_rows ? 8 : 1; g.shift_rows = shift_rows;
    if (shift_rows && (d2s_s > 0 || phase_mask || (long long)(N / 64) * 8 > 65535)) return VXB_EARG;
    g.dy_scale = dy_scale;
    g.dbg = g_wh_dbg;
    if (nsplit > 65535 || N / 64 > 65535) return VXB_ESIZE;
    {   // voxel indices are 32-bit inside the kernel
        const long long vf = (long long)S_out * (d2s_s > 0 ? d2s_s : 1);
        if ((long long)B * S_in * S_in * S_in >= INT32_MAX || (long long)B * vf * vf * vf >= INT32_MAX) return VXB_ESIZE;
    }
    hipStream_t st = (hipStream_t)stream;
    // two chunks per workgroup (8 waves sharing the dY tile): measured at B = 4 in 'bf16x3' (tools/bench_wgrad_halo.py, WH_NCH):
    // + 2.7 % on the dense 128 -> 64 gradient at S = 100 (4.83 -> 4.70 ms), - 5 % on the tap-masked depth-to-space one (2.20 -> 2.31)
    // (depth-to-space dY -- the up-conv, one phase per 64-column block -- in fp16: 5.52 -> 4.72 ms in the training step, B = 16)
    int nch = ((C0 + C1) % 32 == 0 && (d2s_s <= 0 || pm == 2)) ? 2 : 1;
    if (g_wh_nch) nch = (g_wh_nch == 2 && (C0 + C1) % 32 == 0) ? 2 : 1;
    if (wgrad_halo_shape(S_out, pm == 1)) return vxb_wgrad_halo_launch_t44(g, pm, nch, nsplit, st);
    if (pm == 2) return nch == 2 ? wgrad_halo_launch<2, 2, 8, 2>(g, nsplit, st) : wgrad_halo_launch<2, 2, 8, 1>(g, nsplit, st);
    if (nch == 2) return pm ? wgrad_halo_launch<1, 2, 8, 2>(g, nsplit, st) : wgrad_halo_launch<0, 2, 8, 2>(g, nsplit, st);
    return pm ? wgrad_halo_launch<1, 2, 8, 1>(g, nsplit, st) : wgrad_halo_launch<0, 2, 8, 1>(g, nsplit, st);
}

// 3x3x3 stride-1 specialisation of vxb_conv3d_wgrad_bf16_f32 / _bf16x3_f32 (same contract and part[z][K][N] layout;
// the z slices are runs of 128-voxel tiles, 2x8x8 or 4x4x8 -- vxb_conv3_wgrad_halo_tiles gives their number).
// C0, C1 % 16 == 0, N % 64 == 0; d2s: d2s_C == 64.  phase_mask (optional, d2s only): one word per 64-column block, bit t
// set <=> the (tap t, phase) weight block is not structurally zero; the other blocks of `part` are left untouched.
extern "C" int vxb_conv3_wgrad_halo_bf16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                             int off, int replicate, const float* dy, int N, int64_t ldy, int d2s_s,
                                             int d2s_C, float* part, int nsplit, const uint32_t* phase_mask, vxb_stream_t stream) {
    return wgrad_halo_impl(0, nullptr, src0, src1, C0, C1, B, S_in, S_out, off, replicate, dy, N, ldy, d2s_s, d2s_C, part, nsplit, phase_mask, stream);
}

extern "C" int vxb_conv3_wgrad_halo_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                               int off, int replicate, const float* dy, int N, int64_t ldy, int d2s_s,
                                               int d2s_C, float* part, int nsplit, const uint32_t* phase_mask, vxb_stream_t stream) {
    return wgrad_halo_impl(1, nullptr, src0, src1, C0, C1, B, S_in, S_out, off, replicate, dy, N, ldy, d2s_s, d2s_C, part, nsplit, phase_mask, stream);
}

// The same gradient with ONE fp16 product per term (v_mfma_f32_16x16x32_f16, fp32 accumulate) instead of the bf16x3 triple:
// a weight gradient is a leaf of the backward pass -- its rounding errors (2^-12 per operand, averaged over >= 10^5 voxels) do not
// propagate, and against the reference's gradients at configs[1] / [2] size it is indistinguishable from the bf16x3 kernel
// (tools/experiments/emu_precision.py).  Range: x saturates at +-65504; dY is multiplied by *dy_scale (device
// float, a power of two, see vxb_absmax_scale_f32; NULL = 1) before the conversion and `part` comes out as dy_scale * dW.
extern "C" int vxb_conv3_wgrad_halo_f16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                            int off, int replicate, const float* dy, int N, int64_t ldy, int d2s_s,
                                            int d2s_C, float* part, int nsplit, const uint32_t* phase_mask, const float* dy_scale,
                                            vxb_stream_t stream) {
    return wgrad_halo_impl(2, dy_scale, src0, src1, C0, C1, B, S_in, S_out, off, replicate, dy, N, ldy, d2s_s, d2s_C, part, nsplit, phase_mask, stream);
}

// Weight gradient of a 5x5x5 stride-1 conv (the first conv of the decoder's up-block, network_utils.py:236-244 with kernel 5) on the
// same kernel: the 125 taps are eight shifted 3x3x3 blocks -- per axis the offsets {-2, -1, 0} (shift bit clear, all three taps) and
// {+1, +2} (bit set: the block starts at 0, whose first tap belongs to the other block) -- run as ONE launch with the shift in
// grid.y.  shift_rows: device int32 [8][27], the row block (0 .. 124, = (kd * 5 + kh) * 5 + kw) of every (shift, tap) or -1
// (ops.k5_shift_rows).  part [nsplit][125 * (C0 + C1)][N]: every row is written exactly once; off = -2 for 'same' padding.
// Against the generic gather kernel (vxb_conv3d_wgrad_f16_f32: one workgroup per tap re-reads x through the L2 125 times and spends
// 92 VALU instructions per MFMA on addresses): 2.27 -> 0.8 ms at B = 16, 20^3, 128 -> 64.
extern "C" int vxb_conv3_wgrad_halo5_f16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S, int off,
                                             int replicate, const float* dy, int N, int64_t ldy, float* part, int nsplit,
                                             const int32_t* shift_rows, const float* dy_scale, vxb_stream_t stream) {
    if (!shift_rows) return VXB_EARG;
    return wgrad_halo_impl(2, dy_scale, src0, src1, C0, C1, B, S, S, off, replicate, dy, N, ldy, 0, 0, part, nsplit, nullptr, stream,
                           shift_rows, 125);
}

// number of voxel tiles the entries above split into z slices (for choosing nsplit)
extern "C" size_t vxb_conv3_wgrad_halo_tiles(int B, int S_out, int x3) {
    if (B < 1 || S_out < 1) return 0;
    const int sh = wgrad_halo_shape(S_out, x3);
    return (size_t)B * vxb_cdiv(S_out, sh ? 4 : 2) * vxb_cdiv(S_out, sh ? 4 : 8) * vxb_cdiv(S_out, 8);
}

extern "C" void vxb_debug_set_wgrad_halo_experiment(int bits) { g_wh_dbg = bits; }
extern "C" void vxb_debug_set_wgrad_halo_chunks(int nch) { g_wh_nch = (nch == 1 || nch == 2) ? nch : 0; }
extern "C" void vxb_debug_set_wgrad_halo_shape(int shape) { g_wh_shape = shape < 0 ? -1 : (shape ? 1 : 0); }
#endif   // WH_T44_UNIT
