// LDS-halo 3x3x3 stride-1 conv3d on the bf16 matrix cores -- the `final` conv of the Q-function
// (perceiver_lang_io.py:462, 442 GF/sample forward) and its data gradient, in the 'bf16' and 'bf16x3' precisions.
//
// The generic implicit-GEMM kernel (gemm_conv.hip) re-fetches every input voxel once per tap (27x) through L2, which
// bounds it at ~250-400 TF/s in bf16.  Here one workgroup owns a 4x8x8 block of output voxels:
//   * the 6x10x10 input halo of one channel chunk is staged ONCE into LDS (fp32 -> bf16 on the way, replicate padding
//     by clamping / zero padding by predication): 2.3x amplification instead of 27x;
//   * the A operand of v_mfma_f32_32x32x16_bf16 is read straight out of the halo at a shifted address per tap;
//   * the weights of a tap (B operand) come either from global memory in MFMA fragment order (pre-shuffled by
//     ops.halo_wfrag; every wave loads its own fragments two taps ahead, no barrier in the 27-tap loop -- the default)
//     or, without that copy, as the [N][chunk] slice staged per tap through a double-buffered LDS tile.
// Variants: two concatenated sources, zero padding (data gradients), space-to-depth input, depth-to-space output, and a
// "fold" epilogue that applies the adjoint of the replicate padding in place of the padded-domain store.
// A chunk is 32 bf16 per voxel: 32 channels ('bf16') or 16 channels as hi|lo halves ('bf16x3': hi = bf16(a),
// lo = bf16(a - hi), products hi*hi + hi*lo + lo*hi).  That keeps a workgroup at 68-78 KB of LDS and <= 256 VGPRs, so
// TWO workgroups share a CU and one's halo fetch hides behind the other's MFMAs (global loads retire in order, so
// prefetching the next halo inside one wave would stall on the per-tap weight loads instead).
// LDS layout: halo[(d*10 + h)*12 + w][40 bf16]: with 80-byte voxels and a padded row length of 12 the 8(h) x 4(w)
// patch of voxels that forms one 32-row MFMA tile is conflict-free for ds_read_b128 (brute-forced over the four
// 16-lane service groups of that instruction).  Output: C[row = voxel][col = channel] -> 128-byte stores per voxel.
#include "common.h"
#include "ss3d.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int TD = 4, TH = 8, TW = 8;                 // output tile
constexpr int HDp = TD + 2, HHp = TH + 2, HWp = 12;   // halo extents (w padded 10 -> 12)
constexpr int HW_USED = TW + 2;
constexpr int NPOS = HDp * HHp * HW_USED;             // 600 staged voxels
constexpr int SP = 40;                                // bf16 per halo voxel (32 + 8 pad)
constexpr int LDW = 40;                               // bf16 per weight row (32 + 8 pad)
constexpr int HALO_SLOTS = HDp * HHp * HWp;           // 720

struct HaloArgs {
    const float* src0;
    const float* src1;
    int C0, C1;
    int B, S_in, S_out, off, replicate;
    const u16* wb;          // bf16 [N][K] (x3: planes [2][N][K]), K = 27 * (C0 + C1), k = tap * (C0 + C1) + channel
    int N, K;
    const float* bias;
    float* out;             // [B, S_out^3, N]
    int act;
    float slope;
    int ntd, nth, ntw;
    int s2d_s, s2d_C;       // > 0: src0 is a fine grid [B, (S_in*s)^3, s2d_C], input channel = (phase, co)
    int d2s_s;              // > 0: out is a fine grid [B, (S_out*s)^3, 64], output column = (phase, co)
    // fold mode (fold_pad > 0): the conv is a data gradient on the padded domain S_out = fold_S + 2 fold_pad; instead of
    // storing it, column block nb adds the replicate-padding adjoint into fold_dst[nb] [B, fold_S^3, 64]
    const u16* wfrag;        // WD kernels: weights pre-shuffled into fragment order (ops.halo_wfrag)
    // TL kernels (space-to-depth input with block-sparse weights, the polyphase up-conv's data gradient): every phase of
    // the fine grid belongs to a footprint class with its own list of non-zero taps; wfrag then holds only those taps
    // ([column block][tap_total][...], a chunk's taps consecutive).  taptab = ncls lists of 32 ints (entries 0..26: LDS
    // offset of the n-th listed tap, entry 31: list length, a multiple of 3 -- padded with zero-weight taps) followed by
    // (class, taps listed before this phase's first chunk) per phase.
    const int* taptab;
    int ncls, nphase, tap_total;
    int fold_pad, fold_S;
    float* fold_dst[2];
    const float* fold_y[2];  // != nullptr: multiply by LeakyReLU'(y) (the producer's activation)
    int fold_acc[2];         // 1: dst += ..., 0: dst = ...
    int dbg;                 // timing experiments only (vxb_debug_set_halo_experiment): results are WRONG when != 0
    // 'fp16' products (PM = 2): the input is multiplied by scale[0] before the conversion to half (a power of two taken from
    // its largest magnitude, vxb_absmax_scale_f32) and the result by scale[1] = 1 / scale[0] in the epilogue; nullptr = 1
    const float* scale;
    unsigned* amax_part;     // fold mode, optional: word [workgroup] = largest magnitude (bits) this workgroup wrote to its destination
    float* colsum_part;      // fold mode, optional: [workgroup][64] = column sums of what this workgroup wrote (the bias gradient of the
                             // conv that produced the destination's activation, taken on the way instead of by a pass over 4 GB)
    // fold mode, 64-column fp16 launches, fold_acc[1] == 2 ("wgin"; the second block's fields are free there and carry the operands,
    // so that the argument block -- and with it the register allocation of every other variant -- stays as it was): the destination
    // is the output of a 1x1x1 conv of x = fold_y[1] [B, S^3, 10] whose data gradient nothing else reads (the input conv of the
    // Q-function: the voxel grid is a detached input, agent :100) -- instead of storing the folded gradient, its share of that conv's
    // weight / bias gradient is accumulated on the spot: fold_dst[1] [workgroup][64][11] = sum over the workgroup's voxels of
    // g[c] * {x[0..9], 1}, g after fold_y[0]'s LeakyReLU' (still multiplied by the fp16 operand scale: the finishing kernel undoes it)
    // TL kernels, split K (ksplit > 1): the launch has ksplit workgroups per tile; part p walks the chunks kparts[p] .. kparts[p + 1] - 1
    // (device array, balanced by listed taps on the host) and writes its partial sums to out + p * part_stride -- the up-conv's data
    // gradient has only B * 54 tiles of ~5 ms each for the chip's 512 workgroup slots
    int ksplit;
    const int* kparts;
    long long part_stride;
    // plain launches with 64 columns per tile (N = 64), optional: SpatialSoftmax3D + max statistics of the OUTPUT (after bias and
    // activation) taken in the epilogue -- the online-softmax partial of every (tile, channel) goes to ss_part [B][N][ntiles] (SsPart),
    // ss_lin = the S_out coordinate values (ops.lin_table); vxb_ss3d_final_tiles_launch merges them (perceiver_lang_io.py:470 on the
    // output of :462, without the 4.1 GB pass over u)
    float* ss_part;
    const float* ss_lin;
};

// bf16 pair from two fp32 (RNE).  Both forms give identical bits; which one is FASTER was measured per precision on the
// 128->64 forward (4 waves x 2 M tiles): the integer form 765 TF/s vs v_cvt_pk_bf16_f32 590 TF/s in 'bf16', but 351 vs 365
// TF/s in 'bf16x3' -- the cheaper conversion bunches the two resident workgroups' ds_write bursts together.
// PM = product mode: 0 plain bf16, 1 bf16x3 (hi | lo halves, three MFMAs per product), 2 plain fp16 (pre-scaled input,
// saturating conversion), 3 "fp16x2": the input as an fp16 hi | lo pair (pre-scaled: 22 bits), the weights as ONE fp16 value, two
// MFMAs per product -- the data gradients that propagate (DESIGN 4a, round 5 of tools/experiments/emu_precision.py: rounding the
// WEIGHTS of a data gradient to 11 bits moves no reference gate, rounding its dY does)
template <int PM>
__device__ __forceinline__ f32x16 hb_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    if (PM >= 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int PM>
__device__ __forceinline__ unsigned hb_pack2(float lo, float hi) {
    if (PM >= 2) return vxb_pack_f16(__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f));
    if (PM == 1) return vxb_pack_bf16(lo, hi);
    unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7fffu + ((a >> 16) & 1u);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xffff0000u);
}

// the fp32 values of a packed pair as the matrix cores will see them (bf16: the upper halves; fp16: converted back)
template <int PM>
__device__ __forceinline__ float hb_unpack_lo(unsigned pk) {
    if (PM >= 2) return (float)__builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffffu));
    return __uint_as_float(pk << 16);
}
template <int PM>
__device__ __forceinline__ float hb_unpack_hi(unsigned pk) {
    if (PM >= 2) return (float)__builtin_bit_cast(_Float16, (unsigned short)(pk >> 16));
    return __uint_as_float(pk & 0xffff0000u);
}

__device__ __forceinline__ bool g_dbg_all_waves(const HaloArgs& g) { return (g.dbg & 4) != 0; }    // experiment bit 4: no wave skipping

// Tile index (inside one sample) -> tile coordinates.  Launch order walks 4 x 4 x 4 BLOCKS of tiles (ragged at the far faces), not rows:
// the ~64 workgroups that are resident on an XCD at one time then cover a compact 16 x 32 x 32 voxel block, whose halo overlaps meet in
// that XCD's L2 -- in row order (w fastest) they were a 4 x 40 x 104 slab and depth neighbours, the largest overlap (2 of 6 planes),
// ran 169 tiles apart.  Any bijection gives the same results; experiment bit 0x80 restores row order.
__device__ __forceinline__ void halo_tile_coords(const HaloArgs& g, int t, int& td, int& th, int& tw) {
    if (g.dbg & 0x80) { tw = t % g.ntw; t /= g.ntw; th = t % g.nth; td = t / g.nth; return; }
    vxb_tile_block_coords(t, g.ntd, g.nth, g.ntw, td, th, tw);
}

template <int NTG, int PM, int NW, int WD, int TL, int WN, int HALF, int WG = 0, int BLN = 0>
                                              // NTG = N / 32 column tiles per workgroup; the NW waves form a (NW / WN) x WN grid over
                                              // (8 M tiles) x (NTG column tiles): 4 x 1 -> 2 M tiles x 2 column tiles per wave,
                                              // 2 x 2 -> 4 x 1 (half the B-fragment traffic per MFMA, twice the A reads from LDS),
                                              // 8 x 1 -> 1 x 2; WD: B fragments straight from global (pre-shuffled weights); TL:
                                              // per-chunk tap lists (block-sparse weights, WD only)
                                              // HALF = 1: an edge tile whose odd M tiles hold no output voxel (see the kernel below)
                                              // WG = 1: Winograd F(2, 3) along the depth axis (see WG_* below)
__device__ __forceinline__ void conv3_halo_body(const HaloArgs& g) {
    // ---- WG: the two output depths 2p, 2p + 1 of a tile (p = 0, 1: the wave row wm) come from FOUR products per (kh, kw) instead of six:
    //   m0 = (x0 - x2) g0,  m1 = (x1 + x2) (g0 + g1 + g2) / 2,  m2 = (x2 - x1) (g0 - g1 + g2) / 2,  m3 = (x1 - x3) g2
    //   y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3                                   (x: input depths 2p .. 2p + 3, g: the three depth taps)
    // The input transform is applied on the way into LDS (fp32, before the hi | lo split; the image holds the 2 x 4 transformed planes
    // instead of the 6 raw ones), the weight transform on the host (ops.halo_wfrag_wg: 36 "taps" (xi, kh, kw)), the output transform on
    // the accumulators (a wave's M tiles are (depth, w half), so y0 / y1 are its own registers).  36 x 2 instead of 27 x 4 MFMA groups
    // per wave and chunk: two thirds of the matrix work, which is what bounds the direct kernel (73 % of its time).
    static_assert(WG != 2 || PM == 1 || PM == 3, "BL: the two-halves-per-voxel variants");
    static_assert(!WG || (WD && !TL && WN == 2 && NW == 4 && NTG == 2 && (HALF == 0 || HALF == 1) && PM >= 1),
                  "WG: the final conv's forward (bf16x3) and its data gradients (fp16x2; fp16 with 16-channel chunks)");
    constexpr int NTAP = WG ? 36 : 27;
    // ---- BL (WG = 2; round 6): the weight fragments of a tap reach the four waves through LDS instead of every wave fetching its own.
    // Measured on the bf16x3 forward: the fragment loads cost 21 % of the launch and do not hide; two 1 KB fragments per
    // wave and tap, fetched twice per workgroup (the two waves of a column tile), ~16 B / clk / CU through the vector L1.  Here a ROUND
    // = the 4 KB of fragments of one tap (two weight planes) or two taps (one plane) is fetched ONCE per workgroup -- wave w brings
    // fragment w with one `global_load_lds_dwordx4`, no VGPRs -- into a ring of BL_NR rounds; one s_barrier per round publishes it.
    // The room comes from the halo image: 64-byte voxels and 10-voxel rows (51 KB) instead of 80-byte voxels and 12-voxel rows (77 KB),
    // the bank conflicts of the fragment reads avoided by XOR-ing the 16-byte slot of a voxel (k half | hi / lo) with its row & 3
    // instead of by padding: the 16 lanes of a ds_read_b128 service group are 4 rows x 4 voxels; the voxels of a row fall on the four
    // 64-byte quarters of the bank space (10 voxels per row: (2 h + w) mod 4), the four rows {0, 3, 5, 6} / {1, 2, 4, 7} on different slots.
    constexpr bool BL = WG == 2;
    constexpr int SPW = BL ? 32 : SP;               // u16 per voxel of the WG image
    constexpr int HWW = BL ? HW_USED : HWp;         // voxels per row
    constexpr int PLANE = HHp * HWW * SPW;          // u16 per depth plane of the LDS image
    constexpr int BL_NR = 6;                        // rounds in the ring (24 KB)
    constexpr int BL_TPR = (PM == 1) ? 1 : 2;       // taps per round: a round is four 1 KB fragments
    constexpr int BL_ROUNDS = 36 / BL_TPR;
    constexpr int BL_RING = 8 * PLANE;              // u16 offset of the ring
    // BN (BLN = 1; no Winograd, every wave multiplies BOTH column tiles -- the d(d0) data gradient on fp16 products and the direct forward):
    // there the four waves fetch the SAME four 1 KB fragments per tap (16 KB through the vector L1 for 4 KB of weights, at 8 MFMAs per
    // wave and tap that is the L1's peak rate).  Same ring -- a round is a tap -- behind the padded halo image of this variant, in the
    // space of the weight tiles the register-staged variant keeps there (5 rounds: 57.6 + 20 KB, two workgroups per CU).
    constexpr bool BN = BLN != 0;
    static_assert(!BN || (WD && !TL && !WG && WN == 1 && NW == 4 && NTG == 2 && (PM == 0 || PM == 1 || PM == 2)), "BN: four fragments per tap, no Winograd");
    constexpr int RING_OFF = BL ? BL_RING : HALO_SLOTS * SP;
    constexpr int RING_NR = BL ? BL_NR : 5;
    constexpr int X3 = PM == 1;                     // three products: input hi | lo, weight planes hi / lo
    constexpr int X2 = PM == 3;                     // two products: input hi | lo, one weight plane
    constexpr int HL = X3 || X2;                    // a chunk is 16 channels as hi | lo halves
    constexpr int W1 = WG && PM == 2;               // WG with single fp16 products: 16-channel chunks too (one plane, one MFMA per tap and tile)
    constexpr int NF = (X2 || W1) ? 1 : 2;          // weight fragments per (tap, column tile): k halves (PM 0, 2) / planes (1) / one (3; W1)
    constexpr int ST = HALF == 1 ? 2 : 1;           // M-tile stride of the tap loops
    constexpr int MLIM = HALF == 2 ? 3 : (HALF == 3 ? 2 : 8 / (NW / WN));     // ... and their end: HALF = 2 runs three of a wave's four M tiles, 3 two
    constexpr int NTH = NW * 64, MTW = 8 / (NW / WN), NT = NTG / WN;
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16* halo = smem;                               // [HALO_SLOTS][SP]
    u16* wsm = smem + (WG ? 8 : HDp) * PLANE;       // [2][N][LDW]   (WG: 8 planes, and no weight tiles -- WD)
    constexpr int N = NTG * 32;
    constexpr int CPC = (HL || W1) ? 16 : 32;       // channels per chunk
    constexpr int F4P = CPC / 4;                    // float4 per voxel per chunk
    constexpr int NLD = (NPOS * F4P + NTH - 1) / NTH;   // halo float4 loads per thread per chunk (10 / 19)
    constexpr int W_V8 = (N * 4 + NTH - 1) / NTH;              // 16-byte weight loads per thread per tap (1 / 2)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hi = lane >> 5, lq = lane & 31;
    const int wm = wid / WN, wn = wid % WN;          // this wave's row of M tiles / its first column tile is wn * NT
    // XCD-aware order (workgroup id b runs on XCD b % 8, each with a private L2): give every XCD a contiguous run of
    // tiles so that neighbouring tiles share their halo overlap through one L2.  Bijective for any grid size.
    int t;
    {
        const int nwg = gridDim.x, lid = blockIdx.x;
        const int xcd = lid & 7, slot = lid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int nnb = g.N / N;                       // column blocks of one voxel tile are neighbours in launch order
    const int n0 = (t % nnb) * N;                   // this workgroup's 64 output channels
    t /= nnb;
    int kpart = 0;
    if (TL) {
        // the part is the SLOWEST index of the launch order: workgroups that run together are neighbouring tiles of one part, which read
        // the same lines of the fine grid through their halos.  (Part fastest -- the ksplit parts of a tile side by side -- fetched
        // 2.75 x the bytes: profiles/r03_v8_pmc_FETCH_SIZE_summary.txt, 21.5 against 7.8 M KiB per launch.)
        const int ntile_ = g.B * g.ntd * g.nth * g.ntw;
        kpart = t / ntile_; t -= kpart * ntile_;
    }
    const int per_b = g.ntd * g.nth * g.ntw;
    const int b = t / per_b;
    int td, th, tw;
    halo_tile_coords(g, t - b * per_b, td, th, tw);
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int Ct = g.C0 + g.C1;
    // A wave whose M tiles all lie beyond the last depth slice (the last depth tile of S_out = 22 or 102: two of its four depths
    // do not exist) has nothing to compute: in the WD kernels (no barrier inside the tap loop) it only helps staging the halo.
    // 1/6 of the workgroups of the up-conv data gradient (22^3 grid in 24^3 of tiles) and 1/26 of the final conv's run at half
    // their matrix work this way.
    const bool wave_on = HALF == 3 || __builtin_amdgcn_readfirstlane((int)(!WD || g_dbg_all_waves(g) || d0 + (((wid / WN) * (8 / (NW / WN))) >> 1) < g.S_out)) != 0;

    // HALF: edge tiles whose last rows / columns lie beyond the output grid run a part of their matrix work, with the same products
    // in the same order for every output voxel (bit-identical results).
    //   HALF = 1 (at most 4 of the 8 rows or columns exist; S_out = 100: the 13th tile covers 96..103): along w the odd M tiles
    //     (w half 1) are simply left out; along h the M tile of such a workgroup is a 4(h) x 8(w) patch instead of 8(h) x 4(w), so
    //     that again the odd M tiles (h half 1) are the empty ones.
    //   HALF = 2 (6 of 8 exist: S_out = 22 -- the up-conv's data gradient -- and 102): a wave (two depths) runs three M tiles: the
    //     two patches of the existing half, one per depth, and the 2-wide strip of rows / columns 4..5 of both depths.
    // The lane order inside the 4 x 8 patch and the h strip keeps ds_read_b128 conflict-free: its 16-lane service groups {0-3, 12-15,
    // 20-27} / {4-11, 16-19, 28-31} take the rows {0, 2} / {1, 3} of the patch (h = 4 / h = 5 of the strip), whose 16 voxels fall on
    // 16 different 16-byte slots with 80-byte voxels and 12-voxel rows; the w strip (8 distinct slots for 32 voxels) is read with
    // 2-way conflicts (tools/experiments/halo_rowmap_check.py).
    //   HALF = 3 (the last depth tile holds two depths: S_out = 22, 102): instead of two waves doing both depths and two idling (the
    //     wave_on rule of the other modes), every wave row takes ONE depth, its two M tiles -- the workgroup is done in half the time.
    const bool edge_h = HALF && HALF != 3 && __builtin_amdgcn_readfirstlane((int)(w0 + (HALF == 1 ? 4 : 6) < g.S_out)) != 0;     // (else: the w edge)
    // tile-local voxel (dd, hh, ww) of row l (0..31) of this wave's i-th M tile
    auto rowmap = [&](int i, int l, int& dd, int& hh, int& ww) {
        const int mt = wm * MTW + i;
        const int par = ((l >> 2) ^ (l >> 3) ^ (l >> 4)) & 1;          // service group of lane l
        const int aw = ((l >> 3) & 1) * 4 + (l & 3), k = (l >> 3) * 4 + (l & 3);
        dd = mt >> 1; hh = l >> 2; ww = (mt & 1) * 4 + (l & 3);
        if (HALF == 1 && edge_h) { hh = (mt & 1) * 4 + par + 2 * (l >> 4); ww = aw; }
        if (HALF == 3) { dd = wm; ww = (i & 1) * 4 + (l & 3); }
        if (HALF == 2) {
            dd = 2 * wm + i; ww = l & 3;
            if (edge_h) { hh = par + 2 * (l >> 4); ww = aw; }
            if (i == 2) {
                if (edge_h) { dd = 2 * wm + (l >> 4); hh = 4 + par; ww = aw; }
                else { dd = 2 * wm + par; hh = k >> 1; ww = 4 + (k & 1); }
            }
        }
    };

    f32x16 acc[MTW][NT];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A-operand base slots of this wave's two M tiles: tile mt = wid*2 + i -> d = mt >> 1, w half = mt & 1;
    // lane row: h = lq >> 2, w = (mt & 1) * 4 + (lq & 3)
    int abase[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        int dd, hh, ww;
        rowmap(i, lq, dd, hh, ww);
        abase[i] = ((dd * HHp + hh) * HWp + ww) * SP + 8 * hi;
    }
    const int wrow = lq * LDW + 8 * hi;            // B-operand row of this lane inside a weight tile (+ nt*32*LDW)
    // WG: m0..m3 of this wave's two (w half) tiles; the A operand of product xi is transformed plane 4 wm + xi
    constexpr int WLIM = HALF == 1 ? 1 : 2;
    f32x16 wacc[WG ? 2 : 1][WG ? 4 : 1];
    int wabase[2] = {0, 0};
    int wab[2][3] = {{0, 0, 0}, {0, 0, 0}};      // BL: the hi-half fragment address of tile wh at kernel row kh (the slot depends on the row)
    if constexpr (WG) {
#pragma unroll
        for (int wh = 0; wh < 2; ++wh) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) wacc[wh][x][r] = 0.f;
            int dd, hh, ww;
            rowmap(wh, lq, dd, hh, ww);
            wabase[wh] = ((4 * wm * HHp + hh) * HWp + ww) * SP + 8 * hi;
            if (BL) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) wab[wh][kh] = 4 * wm * PLANE + ((hh + kh) * HWW + ww) * SPW + ((hi ^ ((hh + kh) & 3)) << 3);
            }
        }
    }

    // halo staging slots of this thread: element e = tid + 256 i -> voxel e / F4P, channel quad e % F4P
    // space-to-depth source (s2d_s > 0): the input "channels" are (phase, co) of a fine grid [B, (S_in*s)^3, s2d_C];
    // voxel (id, ih, iw), phase (rd, rh, rw) lives at fine voxel (id*s + rd, ...).
    const int sm = g.s2d_s > 0 ? g.s2d_s : 1;
    const int Vin = g.S_in * sm;
    int st_goff[NLD];      // voxel offset in the source cube (in voxels, phase 0), -1 = unused, -2 = zero fill
    int st_soff[NLD];      // LDS offset (u16)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + NTH * i;
        int p = e / F4P;
        const int c4 = (e % F4P) * 4;
        st_goff[i] = -1;
        st_soff[i] = 0;
        if (p < NPOS) {
            const int pw = p % HW_USED; p /= HW_USED;
            const int ph = p % HHp; p /= HHp;
            const int pd = p;
            int id = d0 + pd + g.off, ih = h0 + ph + g.off, iw = w0 + pw + g.off;
            bool ok = true;
            if (g.replicate) {
                id = min(max(id, 0), g.S_in - 1); ih = min(max(ih, 0), g.S_in - 1); iw = min(max(iw, 0), g.S_in - 1);
            } else {
                ok = id >= 0 && id < g.S_in && ih >= 0 && ih < g.S_in && iw >= 0 && iw < g.S_in;
            }
            st_soff[i] = ((pd * HHp + ph) * HWp + pw) * SP + c4;
            if (ok) st_goff[i] = ((id * sm) * Vin + ih * sm) * Vin + iw * sm;
            else st_goff[i] = -2;                   // staged as zeros
            if (HALF && HALF != 3 && (edge_h ? ph : pw) >= (HALF == 1 ? 6 : 8)) st_goff[i] = -1;      // a part tile reads 6 / 8 of the 10 rows or columns
            if (HALF == 3 && pd >= 4) st_goff[i] = -1;                                           // ... two depths: 4 of the 6 halo slices
        }
    }
    const long long bvox = (long long)b * Vin * Vin * Vin;
    // WG staging: a thread owns whole depth COLUMNS (h, w, channel quad) of the halo -- 10 x 10 x 4 = 400 over 256 threads -- because the
    // transform mixes the six depths of a column
    constexpr int NCOL = WG ? 2 : 1;
    int wg_hw[NCOL];        // (h, w) part of the source voxel offset, -1 = unused, -2 = zero fill
    int wg_soff[NCOL];      // LDS offset inside plane 0
    int wg_d[HDp];          // depth part of the source voxel offset (uniform), -2 = zero fill
    if constexpr (WG) {
#pragma unroll
        for (int i = 0; i < NCOL; ++i) {
            const int e = tid + NTH * i;
            const int p = e / F4P, c4 = (e % F4P) * 4;
            wg_hw[i] = -1; wg_soff[i] = 0;
            if (p < HHp * HW_USED) {
                const int pw = p % HW_USED, ph = p / HW_USED;
                int ih = h0 + ph + g.off, iw = w0 + pw + g.off;
                bool ok = true;
                if (g.replicate) { ih = min(max(ih, 0), g.S_in - 1); iw = min(max(iw, 0), g.S_in - 1); }
                else ok = ih >= 0 && ih < g.S_in && iw >= 0 && iw < g.S_in;
                wg_soff[i] = BL ? (ph * HWW + pw) * SPW + (((c4 >> 3) ^ (ph & 3)) << 3) + (c4 & 4) : (ph * HWp + pw) * SP + c4;
                wg_hw[i] = ok ? ih * Vin + iw : -2;
                if (HALF == 1 && (edge_h ? ph : pw) >= 6) wg_hw[i] = -1;
            }
        }
#pragma unroll
        for (int pd = 0; pd < HDp; ++pd) {
            int id = d0 + pd + g.off;
            if (g.replicate) id = min(max(id, 0), g.S_in - 1);
            wg_d[pd] = (id >= 0 && id < g.S_in) ? id * Vin * Vin : -2;
        }
    }

    // Weight tiles travel global -> registers -> LDS two taps ahead of their use (two register sets, two LDS buffers),
    // so an L2 round trip has two taps of MFMAs to hide behind; the A fragments of the next tap are read from the halo
    // before the barrier that publishes its weights.
    uint4 rw0[W_V8], rw1[W_V8];
    // A fragments, two register sets (current tap / next tap): [M tile][k half]; bf16 -> channels 0-15 | 16-31 of the
    // chunk, x3 -> hi | lo of its 16 channels
    bf16x8 afa[MTW][2], afb[MTW][2];
#define HB_LOAD_W(R, tap_)                                                                                           \
    _Pragma("unroll") for (int i = 0; i < W_V8; ++i) {                                                                \
        const int e = tid + NTH * i;                                                                                 \
        const int n = e >> 2, q = e & 3;                                                                             \
        const u16* wp = X3 ? g.wb + ((long long)(q >> 1) * g.N + n0 + n) * g.K + (tap_) * Ct + cb + (q & 1) * 8       \
                           : g.wb + (long long)(n0 + n) * g.K + (tap_) * Ct + cb + q * 8;                            \
        if (N * 4 >= NTH || e < N * 4) R[i] = *reinterpret_cast<const uint4*>(wp);                                    \
    }
#define HB_STORE_W(R, buf_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < W_V8; ++i) {                                                                \
        const int e = tid + NTH * i;                                                                                 \
        if (N * 4 >= NTH || e < N * 4) *reinterpret_cast<uint4*>(&wsm[(buf_) * N * LDW + (e >> 2) * LDW + (e & 3) * 8]) = R[i]; \
    }
#define HB_READ_A(AF, tap_)                                                                                          \
    {                                                                                                                \
        const int tp_ = (tap_);                                                                                      \
        const int toff_ = (((tp_ / 9) * HHp + (tp_ / 3) % 3) * HWp + tp_ % 3) * SP;                                  \
        _Pragma("unroll") for (int i = 0; i < MLIM; i += ST) {                                                       \
            AF[i][0] = *reinterpret_cast<const bf16x8*>(&halo[abase[i] + toff_]);                                    \
            AF[i][1] = *reinterpret_cast<const bf16x8*>(&halo[abase[i] + toff_ + 16]);                               \
        }                                                                                                            \
    }
    // one tap: issue everything the NEXT taps need (weight load two taps ahead, weight tile of tap+1 into the other LDS
    // buffer, A fragments of tap+1) before this tap's MFMAs, so that it all runs under them; one barrier per tap.
#define HB_TAP(tap_, RL, RS, AC, AN)                                                                                 \
    {                                                                                                                \
        const int tap = (tap_);                                                                                      \
        if (tap + 2 < 27) { HB_LOAD_W(RL, tap + 2) }                                                                 \
        __builtin_amdgcn_sched_barrier(0);   /* keep the load at the top of the tap: it must stay two taps ahead */  \
        const u16* wcur = wsm + (tap & 1) * N * LDW;                                                                 \
        bf16x8 bfr[NT][2];                                                                                           \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                              \
            bfr[j][0] = *reinterpret_cast<const bf16x8*>(&wcur[(wn * NT + j) * 32 * LDW + wrow]);                                \
            bfr[j][1] = *reinterpret_cast<const bf16x8*>(&wcur[(wn * NT + j) * 32 * LDW + wrow + 16]);                           \
        }                                                                                                            \
        if (tap + 1 < 27) {                                                                                          \
            HB_STORE_W(RS, (tap + 1) & 1)                                                                            \
            HB_READ_A(AN, tap + 1)                                                                                   \
        }                                                                                                            \
        /* term-major order: consecutive MFMAs hit different accumulators (no back-to-back dependent issue) */       \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                                 \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
            acc[i][j] = hb_mfma<PM>(AC[i][1], bfr[j][HL ? 0 : 1], acc[i][j]);   \
        if (X3) {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                            \
                acc[i][j] = hb_mfma<PM>(AC[i][0], bfr[j][1], acc[i][j]);        \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                                 \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
            acc[i][j] = hb_mfma<PM>(AC[i][0], bfr[j][0], acc[i][j]);            \
        __syncthreads();                                                                                             \
    }
    // ---- WD variant: the B fragments of a tap come straight from global memory (weights pre-shuffled on the host into
    // fragment order: 1 KB per (tap, column tile, k half / plane), lane-contiguous -> fully coalesced, L1-shared by the
    // waves of the CU), two taps ahead in three rotating register sets.  No weight traffic through LDS and NO barrier
    // inside the 27-tap loop.
#ifndef WG_BD
#define WG_BD 4
#endif
#ifndef WG_BD_X2
#define WG_BD_X2 6
#endif
    // B-fragment prefetch distance in taps.  WG: as deep as the registers allow without spilling -- a set is 8 VGPRs with two weight
    // planes (bf16x3: 4 sets; 5 spill 23 registers), 4 with one (fp16x2 / fp16: 6 sets; 8 spill)
    constexpr int BD = WG ? (PM == 1 ? WG_BD : WG_BD_X2) : (WD && !TL && WN == 2) ? 4 : 2;
#ifndef WG_BD_PRE
#define WG_BD_PRE 4
#endif
    constexpr int BD_PRE = WG_BD_PRE;     // ... of which this many sets are requested BEFORE the halo is converted (the staging registers are live then)
    bf16x8 bq0[NT][2], bq1[NT][2], bq2[NT][2];
    bf16x8 bqr[BD + 1][NT][2];
    const int nchunk = Ct / CPC;
    int tapbase = 0, ntap = 27, taplist = 0;      // TL: first row of this chunk in wfrag, its tap count, lane n = n-th tap's LDS offset
    const long long wf_rows = TL ? g.tap_total : (long long)nchunk * NTAP;
#define HD_LOADB(BQ, tap_)                                                                                           \
    if (!(g.dbg & 2) || (tap_) < 3)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                    \
    _Pragma("unroll") for (int f = 0; f < NF; ++f)                                                                    \
        BQ[j][f] = *reinterpret_cast<const bf16x8*>(g.wfrag + (((long long)(n0 / N) * wf_rows + (TL ? tapbase : ch * NTAP) + (tap_)) * (NTG * NF) + (wn * NT + j) * NF + f) * 512 + lane * 8);
#define HB_READ_A_OFF(AF, off_)                                                                                      \
    {                                                                                                                \
        const int toff_ = (off_);                                                                                    \
        _Pragma("unroll") for (int i = 0; i < MLIM; i += ST) {                                                       \
            AF[i][0] = *reinterpret_cast<const bf16x8*>(&halo[abase[i] + toff_]);                                    \
            AF[i][1] = *reinterpret_cast<const bf16x8*>(&halo[abase[i] + toff_ + 16]);                               \
        }                                                                                                            \
    }
    // TL tap n of the chunk's list: same pipeline as HD_TAP, with the list position instead of the tap number
#define HT_TAP(n_, BC, BL, AC, AN)                                                                                   \
    {                                                                                                                \
        const int ln_ = (n_);                                                                                        \
        if (ln_ + 2 < ntap) { HD_LOADB(BL, ln_ + 2) }                                                                \
        if (ln_ + 1 < ntap) { HB_READ_A_OFF(AN, __builtin_amdgcn_readlane(taplist, ln_ + 1)) }                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int i = 0; i < MLIM; i += ST)                                                         \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
            acc[i][j] = hb_mfma<PM>(AC[i][1], BC[j][HL ? 0 : 1], acc[i][j]);    \
        if (X3) {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < MLIM; i += ST)                                                     \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                            \
                acc[i][j] = hb_mfma<PM>(AC[i][0], BC[j][1], acc[i][j]);         \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MLIM; i += ST)                                                         \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
            acc[i][j] = hb_mfma<PM>(AC[i][0], BC[j][0], acc[i][j]);             \
    }
    // the S_out coordinate values of the SpatialSoftmax3D statistics (ss_part launches; S_out <= 256): staged here, read in the
    // epilogue -- published by the chunk loop's barriers
    __shared__ float slin[256];
    if (WD && !TL && WN == 2 && NW == 4 && g.ss_part != nullptr && tid < g.S_out) slin[tid] = g.ss_lin[tid];
    int* ttab = reinterpret_cast<int*>(wsm);        // TL: the tap table lives in the (otherwise unused) weight buffers
    if (TL) {
        for (int i = tid; i < g.ncls * 32 + g.nphase * 2; i += NTH) ttab[i] = g.taptab[i];
        __syncthreads();
    }
#define HD_MFMA(AC, BC)                                                                                              \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < MLIM; i += ST)                                                         \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
            acc[i][j] = hb_mfma<PM>(AC[i][1], BC[j][HL ? 0 : 1], acc[i][j]);    \
        if (X3) {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < MLIM; i += ST)                                                     \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                            \
                acc[i][j] = hb_mfma<PM>(AC[i][0], BC[j][1], acc[i][j]);         \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MLIM; i += ST)                                                         \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
            acc[i][j] = hb_mfma<PM>(AC[i][0], BC[j][0], acc[i][j]);             \
    }
#define HD_TAP(tap_, BC, BL, AC, AN)                                                                                 \
    {                                                                                                                \
        const int tap = (tap_);                                                                                      \
        if (tap + 2 < 27) { HD_LOADB(BL, tap + 2) }                                                                  \
        if (tap + 1 < 27) { HB_READ_A(AN, tap + 1) }                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
            acc[i][j] = hb_mfma<PM>(AC[i][1], BC[j][HL ? 0 : 1], acc[i][j]);    \
        if (X3) {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                           \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                            \
                acc[i][j] = hb_mfma<PM>(AC[i][0], BC[j][1], acc[i][j]);         \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
            acc[i][j] = hb_mfma<PM>(AC[i][0], BC[j][0], acc[i][j]);             \
    }
    // (Measured and rejected, round 2: fetching the halo of chunk ch + 1 into registers during the taps of chunk ch -- 40 more
    // VGPRs in 'bf16x3', no spills -- is 3.6 % SLOWER.  The vector L1 returns data in request order for the whole CU, so HBM-latency
    // halo loads in the middle of a tap loop hold up the L2-hit B-fragment loads of both resident workgroups.)
    constexpr bool PF = false;
    // (WG: the transformed inputs are sums of two values -- one more bit of headroom under the largest half; a power of two, exact)
    const float in_sc = ((PM >= 2 && g.scale) ? g.scale[0] : 1.0f) * ((WG && PM >= 2) ? 0.5f : 1.0f);
    const float out_sc = ((PM >= 2 && g.scale) ? g.scale[1] : 1.0f) * ((WG && PM >= 2) ? 2.0f : 1.0f);
    float4 hv[WG ? NCOL * HDp : NLD];
    auto halo_issue = [&](int ch_) {
        const int cb_ = ch_ * CPC;
        const bool second = cb_ >= g.C0;
        const float* src = second ? g.src1 : g.src0;
        int Cs = second ? g.C1 : g.C0;
        int c0 = second ? cb_ - g.C0 : cb_;
        long long vbase = bvox;
        if constexpr (WG) {
#pragma unroll
            for (int i = 0; i < NCOL; ++i)
#pragma unroll
                for (int pd = 0; pd < HDp; ++pd) {
                    hv[i * HDp + pd] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (wg_hw[i] >= 0 && wg_d[pd] >= 0 && !((g.dbg & 0x20) && ch_ > 0))
                        hv[i * HDp + pd] = *reinterpret_cast<const float4*>(src + (vbase + wg_d[pd] + wg_hw[i]) * Cs + c0 + (((tid + NTH * i) % F4P) * 4));
                }
            return;
        }
        if (g.s2d_s > 0) {
            const int ph = cb_ / g.s2d_C;
            c0 = cb_ - ph * g.s2d_C;
            Cs = g.s2d_C;
            vbase += ((long long)(ph / (sm * sm)) * Vin + (ph / sm) % sm) * Vin + ph % sm;
        }
        const bool dbg_skip_stage = (g.dbg & 1) && ch_ > 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (st_goff[i] >= 0 && !dbg_skip_stage && !((g.dbg & 0x20) && ch_ > 0))      // (0x20: no halo loads after the first chunk)
                hv[i] = *reinterpret_cast<const float4*>(src + (vbase + st_goff[i]) * Cs + c0 + (((tid + NTH * i) % F4P) * 4));
        }
    };
    // BL: this wave's quarter (fragment `wid`) of round r_ of chunk ch_: 1 KB, lane l's 16 bytes to LDS[ring slot + wid KB + 16 l].  Issued as
    // inline asm: the compiler's counter model does not see it (it would put s_waitcnt vmcnt(0) in front of every LDS read that may alias
    // the destination); the counted waits are placed by hand (bl_wait).  A visible load's compiler-placed wait only gets stricter by it.
    const int widu = __builtin_amdgcn_readfirstlane(wid);
    auto bl_issue = [&](int ch_, int r_) __attribute__((always_inline)) {
        const unsigned long long sp_ = (unsigned long long)(g.wfrag + (((long long)(n0 / N) * wf_rows + (long long)ch_ * NTAP) * (NTG * NF) + (long long)r_ * 4 + widu) * 512);
        const unsigned slo_ = __builtin_amdgcn_readfirstlane((unsigned)sp_), shi_ = __builtin_amdgcn_readfirstlane((unsigned)(sp_ >> 32));
        const u16* src_ = reinterpret_cast<const u16*>(((unsigned long long)shi_ << 32) | slo_);
        const unsigned ldsb_ = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)(halo + RING_OFF + (r_ % RING_NR) * 2048 + widu * 512));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((unsigned)(lane * 16)), "s"(src_), "s"(ldsb_) : "memory", "m0");
    };
    auto bl_wait = [&](int n_) __attribute__((always_inline)) {
        switch (n_) {          // (n_ is a constant after unrolling)
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        }
    };
    if (PF) halo_issue(0);
    if (g.dbg & 0xf00) {
        // start-up stagger (experiment bits 8-11, results stay correct): the two workgroups that share a CU start together, stage together
        // and compete for the matrix pipe together; the one in the odd hardware wave slot waits ((dbg >> 8) & 15) x 1024 cycles once
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, WAVE_ID field
        if (blockIdx.x < 512 && (slot & 1))
            for (int i = 0; i < ((g.dbg >> 8) & 15); ++i) __builtin_amdgcn_s_sleep(16);
    }
    int ch_begin = 0, ch_end = nchunk;
    if (TL && g.ksplit > 1) {
        ch_begin = __builtin_amdgcn_readfirstlane(g.kparts[kpart]);
        ch_end = __builtin_amdgcn_readfirstlane(g.kparts[kpart + 1]);
    }
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int cb = ch * CPC;
        const bool dbg_skip_stage = (g.dbg & 1) && ch > 0;
        if (TL) {
            const int ph = cb / g.s2d_C;
            const int cls = __builtin_amdgcn_readfirstlane(ttab[g.ncls * 32 + 2 * ph]);
            const int pre = __builtin_amdgcn_readfirstlane(ttab[g.ncls * 32 + 2 * ph + 1]);
            taplist = ttab[cls * 32 + (lane & 31)];
            ntap = __builtin_amdgcn_readlane(taplist, 31);
            tapbase = pre + (ch - ph * (g.s2d_C / CPC)) * ntap;
        }
        // ---- fetch the halo of this chunk (all loads in flight together), then convert + store
        if (!PF) halo_issue(ch);
        if (!WD) {
            HB_LOAD_W(rw0, 0)
            HB_LOAD_W(rw1, 1)
            // (the barrier that ended the previous chunk's last tap already freed the halo and both weight buffers)
        } else {
            if (wave_on) {
                if (BN) {
                } else if (BD == 2) {
                    HD_LOADB(bq0, 0)
                    HD_LOADB(bq1, 1)
                } else {
#pragma unroll
                    for (int t0 = 0; t0 < (BL ? 0 : (BD < BD_PRE ? BD : BD_PRE)); ++t0) { HD_LOADB(bqr[t0], t0) }
                }
            }
            __syncthreads();                        // no per-tap barriers here: every wave must be done with the old halo
            if constexpr (BL || BN) {
                // every wave has left the previous chunk's taps: the whole ring is free -- rounds 0 .. RING_NR - 1 of this chunk
#pragma unroll
                for (int r0 = 0; r0 < RING_NR; ++r0) bl_issue(ch, r0);
            }
        }
        if constexpr (WG) {
#pragma unroll
            for (int i = 0; i < NCOL; ++i) {
                if (wg_hw[i] == -1 || ((g.dbg & 0x10) && ch > 0)) continue;
                float4* x = &hv[i * HDp];
                if (PM >= 2) {
#pragma unroll
                    for (int pd = 0; pd < HDp; ++pd) { x[pd].x *= in_sc; x[pd].y *= in_sc; x[pd].z *= in_sc; x[pd].w *= in_sc; }
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const float4 x0 = x[2 * pr], x1 = x[2 * pr + 1], x2 = x[2 * pr + 2], x3 = x[2 * pr + 3];
                    float4 t[4];
                    t[0] = make_float4(x0.x - x2.x, x0.y - x2.y, x0.z - x2.z, x0.w - x2.w);
                    t[1] = make_float4(x1.x + x2.x, x1.y + x2.y, x1.z + x2.z, x1.w + x2.w);
                    t[2] = make_float4(x2.x - x1.x, x2.y - x1.y, x2.z - x1.z, x2.w - x1.w);
                    t[3] = make_float4(x1.x - x3.x, x1.y - x3.y, x1.z - x3.z, x1.w - x3.w);
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
                        uint2 pk, q;
                        pk.x = hb_pack2<PM>(t[xi].x, t[xi].y); pk.y = hb_pack2<PM>(t[xi].z, t[xi].w);
                        u16* dstp = &halo[wg_soff[i] + (4 * pr + xi) * PLANE];
                        *reinterpret_cast<uint2*>(dstp) = pk;
                        if (HL) {
                            q.x = hb_pack2<PM>(t[xi].x - hb_unpack_lo<PM>(pk.x), t[xi].y - hb_unpack_hi<PM>(pk.x));
                            q.y = hb_pack2<PM>(t[xi].z - hb_unpack_lo<PM>(pk.y), t[xi].w - hb_unpack_hi<PM>(pk.y));
                            if (BL) *reinterpret_cast<uint2*>(&halo[(wg_soff[i] ^ 16) + (4 * pr + xi) * PLANE]) = q;      // (slot ^ 2: the lo half)
                            else *reinterpret_cast<uint2*>(dstp + 16) = q;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (WG ? 0 : NLD); ++i) {
            if (st_goff[i] != -1 && !dbg_skip_stage && !((g.dbg & 0x10) && ch > 0)) {       // (0x10: loads, but no conversion / LDS stores after the first chunk)
                if (PM >= 2) { hv[i].x *= in_sc; hv[i].y *= in_sc; hv[i].z *= in_sc; hv[i].w *= in_sc; }
                uint2 pk;
                pk.x = hb_pack2<PM>(hv[i].x, hv[i].y); pk.y = hb_pack2<PM>(hv[i].z, hv[i].w);
                *reinterpret_cast<uint2*>(&halo[st_soff[i]]) = pk;
                if (HL) {
                    uint2 q;
                    q.x = hb_pack2<PM>(hv[i].x - hb_unpack_lo<PM>(pk.x), hv[i].y - hb_unpack_hi<PM>(pk.x));
                    q.y = hb_pack2<PM>(hv[i].z - hb_unpack_lo<PM>(pk.y), hv[i].w - hb_unpack_hi<PM>(pk.y));
                    *reinterpret_cast<uint2*>(&halo[st_soff[i] + 16]) = q;
                }
            }
        }
        if (!WD) { HB_STORE_W(rw0, 0) }
        if constexpr (BL || BN) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING_NR - 1) : "memory");      // this wave's fragment of round 0 has landed
            vxb_raw_barrier_lds();                  // (LDS stores of the staging done, then the barrier: no fence -- the rounds in flight stay in flight)
        } else __syncthreads();
        if (!BL && !BN && !wave_on) continue;              // (WD only: the two barriers above are the chunk's only ones; BL: every wave takes the rounds' barriers)
        if constexpr (WG && !BL && BD > BD_PRE) {
#pragma unroll
            for (int t0 = BD_PRE; t0 < BD; ++t0) { HD_LOADB(bqr[t0], t0) }
        }
        if (TL) {
            HB_READ_A_OFF(afa, __builtin_amdgcn_readlane(taplist, 0))
            int n = 0;
#pragma unroll 1
            for (; n + 6 <= ntap; n += 6) {
                HT_TAP(n, bq0, bq2, afa, afb)
                HT_TAP(n + 1, bq1, bq0, afb, afa)
                HT_TAP(n + 2, bq2, bq1, afa, afb)
                HT_TAP(n + 3, bq0, bq2, afb, afa)
                HT_TAP(n + 4, bq1, bq0, afa, afb)
                HT_TAP(n + 5, bq2, bq1, afb, afa)
            }
            if (n < ntap) {          // list lengths are multiples of 3
                HT_TAP(n, bq0, bq2, afa, afb)
                HT_TAP(n + 1, bq1, bq0, afb, afa)
                HT_TAP(n + 2, bq2, bq1, afa, afb)
            }
            continue;
        }
        if constexpr (WG) {
            // tap = (xi, kh, kw): A from transformed plane 4 wm + xi at (kh, kw), accumulator m_xi; same pipeline as the BD = 4 loop below
#define WG_READ_A(AF, tap_)                                                                                          \
    {                                                                                                                \
        const int tp_ = (tap_);                                                                                      \
        const int toff_ = (tp_ / 9) * PLANE + (((tp_ / 3) % 3) * HWp + tp_ % 3) * SP;                                \
        _Pragma("unroll") for (int wh = 0; wh < WLIM; ++wh) {                                                        \
            AF[wh][0] = *reinterpret_cast<const bf16x8*>(&halo[wabase[wh] + toff_]);                                 \
            if (HL) AF[wh][1] = *reinterpret_cast<const bf16x8*>(&halo[wabase[wh] + toff_ + 16]);                    \
        }                                                                                                            \
    }
#define WG_MFMA(AC, BC, XI)                                                                                          \
    {                                                                                                                \
        if (HL) {                                                                                                    \
            _Pragma("unroll") for (int wh = 0; wh < WLIM; ++wh) wacc[wh][XI] = hb_mfma<PM>(AC[wh][1], BC[0][0], wacc[wh][XI]); \
        }                                                                                                            \
        if (X3) {                                                                                                    \
            _Pragma("unroll") for (int wh = 0; wh < WLIM; ++wh) wacc[wh][XI] = hb_mfma<PM>(AC[wh][0], BC[0][1], wacc[wh][XI]); \
        }                                                                                                            \
        _Pragma("unroll") for (int wh = 0; wh < WLIM; ++wh) wacc[wh][XI] = hb_mfma<PM>(AC[wh][0], BC[0][0], wacc[wh][XI]); \
    }
            if constexpr (BL) {
                // A of tap tp: tile wh at (kh, kw) of transformed plane 4 wm + xi; B: fragments (wn, plane) of the tap's round out of the ring
#define BL_READ_A(AF, tap_)                                                                                          \
    {                                                                                                                \
        const int tp_ = (tap_);                                                                                      \
        const int c_ = (tp_ / 9) * PLANE + (tp_ % 3) * SPW;                                                          \
        _Pragma("unroll") for (int wh = 0; wh < WLIM; ++wh) {                                                        \
            AF[wh][0] = *reinterpret_cast<const bf16x8*>(&halo[wab[wh][(tp_ / 3) % 3] + c_]);                        \
            AF[wh][1] = *reinterpret_cast<const bf16x8*>(&halo[(wab[wh][(tp_ / 3) % 3] ^ 16) + c_]);                 \
        }                                                                                                            \
    }
#define BL_READ_B(BF, tap_)                                                                                          \
    {                                                                                                                \
        const int tp_ = (tap_);                                                                                      \
        const u16* rb_ = halo + BL_RING + ((tp_ / BL_TPR) % BL_NR) * 2048 + lane * 8;                                \
        if (PM == 1) {                                                                                               \
            BF[0][0] = *reinterpret_cast<const bf16x8*>(rb_ + (wn * 2 + 0) * 512);                                   \
            BF[0][1] = *reinterpret_cast<const bf16x8*>(rb_ + (wn * 2 + 1) * 512);                                   \
        } else {                                                                                                     \
            BF[0][0] = *reinterpret_cast<const bf16x8*>(rb_ + ((tp_ % BL_TPR) * 2 + wn) * 512);                      \
        }                                                                                                            \
    }
                // Round r >= 1 is opened one tap BEFORE its first tap (its fragments are read a tap ahead, like A): at the top of tap
                // r * BL_TPR - 1 every wave has issued -- and (lgkmcnt(0)) finished -- its reads of round r - 1, whose ring slot then
                // takes round r - 1 + BL_NR.  A wave waits for ITS fragment of round r (the rounds r + 1 .. r + BL_NR - 2 stay in flight).
                bf16x8 bfa[1][2], bfb[1][2];
                BL_READ_A(afa, 0)
                BL_READ_B(bfa, 0)
#pragma unroll
                for (int tp = 0; tp < 36; ++tp) {
                    if ((tp + 1) % BL_TPR == 0 && tp + 1 < 36) {
                        constexpr int dummy_ = 0; (void)dummy_;
                        const int r = (tp + 1) / BL_TPR;
                        const int young = (BL_ROUNDS - 1 - r) < (BL_NR - 2) ? (BL_ROUNDS - 1 - r) : (BL_NR - 2);
                        bl_wait(young);
                        vxb_raw_barrier_lds();
                        if (r - 1 + BL_NR < BL_ROUNDS) bl_issue(ch, r - 1 + BL_NR);
                    }
                    if (tp & 1) {
                        if (tp + 1 < 36) { BL_READ_A(afa, tp + 1) BL_READ_B(bfa, tp + 1) }
                        __builtin_amdgcn_sched_barrier(0);
                        WG_MFMA(afb, bfb, tp / 9)
                    } else {
                        if (tp + 1 < 36) { BL_READ_A(afb, tp + 1) BL_READ_B(bfb, tp + 1) }
                        __builtin_amdgcn_sched_barrier(0);
                        WG_MFMA(afa, bfa, tp / 9)
                    }
                }
                continue;
            }
            WG_READ_A(afa, 0)
#pragma unroll
            for (int tp = 0; tp < 36; ++tp) {
                if (tp + BD < 36) { HD_LOADB(bqr[(tp + BD) % (BD + 1)], tp + BD) }
                if (tp & 1) {
                    if (tp + 1 < 36) { WG_READ_A(afa, tp + 1) }
                    __builtin_amdgcn_sched_barrier(0);
                    WG_MFMA(afb, bqr[tp % (BD + 1)], tp / 9)
                } else {
                    if (tp + 1 < 36) { WG_READ_A(afb, tp + 1) }
                    __builtin_amdgcn_sched_barrier(0);
                    WG_MFMA(afa, bqr[tp % (BD + 1)], tp / 9)
                }
            }
            continue;
        }
        if constexpr (BN) {
#define BN_READ_B(BF, tap_)                                                                                          \
    {                                                                                                                \
        const u16* rb_ = halo + RING_OFF + ((tap_) % RING_NR) * 2048 + lane * 8;                                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                \
        _Pragma("unroll") for (int f = 0; f < NF; ++f) BF[j][f] = *reinterpret_cast<const bf16x8*>(rb_ + (j * NF + f) * 512); \
    }
            bf16x8 bfa[NT][2], bfb[NT][2];
            HB_READ_A(afa, 0)
            BN_READ_B(bfa, 0)
#pragma unroll
            for (int tp = 0; tp < 27; ++tp) {
                if (tp + 1 < 27) {          // open round tp + 1 (see the BL loop)
                    const int r = tp + 1;
                    const int young = (26 - r) < (RING_NR - 2) ? (26 - r) : (RING_NR - 2);
                    bl_wait(young);
                    vxb_raw_barrier_lds();
                    if (r - 1 + RING_NR < 27) bl_issue(ch, r - 1 + RING_NR);
                }
                if (tp & 1) {
                    if (tp + 1 < 27) { HB_READ_A(afa, tp + 1) BN_READ_B(bfa, tp + 1) }
                    __builtin_amdgcn_sched_barrier(0);
                    HD_MFMA(afb, bfb)
                } else {
                    if (tp + 1 < 27) { HB_READ_A(afb, tp + 1) BN_READ_B(bfb, tp + 1) }
                    __builtin_amdgcn_sched_barrier(0);
                    HD_MFMA(afa, bfa)
                }
            }
            continue;
        }
        HB_READ_A(afa, 0)
        if (!WD) {
            for (int tp = 0; tp < 26; tp += 2) {
                HB_TAP(tp, rw0, rw1, afa, afb)
                HB_TAP(tp + 1, rw1, rw0, afb, afa)
            }
            HB_TAP(26, rw0, rw1, afa, afb)
        } else if (BD == 2) {
            for (int tp = 0; tp < 24; tp += 6) {
                HD_TAP(tp, bq0, bq2, afa, afb)
                HD_TAP(tp + 1, bq1, bq0, afb, afa)
                HD_TAP(tp + 2, bq2, bq1, afa, afb)
                HD_TAP(tp + 3, bq0, bq2, afb, afa)
                HD_TAP(tp + 4, bq1, bq0, afa, afb)
                HD_TAP(tp + 5, bq2, bq1, afb, afa)
            }
            HD_TAP(24, bq0, bq2, afa, afb)
            HD_TAP(25, bq1, bq0, afb, afa)
            HD_TAP(26, bq2, bq1, afa, afb)
        } else {
            // deeper B pipeline (2 x 2 wave layout: a set is 8 VGPRs): fragments arrive BD taps ahead in BD + 1 rotating sets
#pragma unroll
            for (int tp = 0; tp < 27; ++tp) {
                if (tp + BD < 27) { HD_LOADB(bqr[(tp + BD) % (BD + 1)], tp + BD) }
                if (tp & 1) {
                    if (tp + 1 < 27) { HB_READ_A(afa, tp + 1) }
                    __builtin_amdgcn_sched_barrier(0);
                    HD_MFMA(afb, bqr[tp % (BD + 1)])
                } else {
                    if (tp + 1 < 27) { HB_READ_A(afb, tp + 1) }
                    __builtin_amdgcn_sched_barrier(0);
                    HD_MFMA(afa, bqr[tp % (BD + 1)])
                }
            }
        }
    }
    if constexpr (WG) {
        // output transform: M tile i = 2 (depth parity) + w half
#pragma unroll
        for (int wh = 0; wh < WLIM; ++wh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[wh][0][r] = (wacc[wh][0][r] + wacc[wh][1][r]) + wacc[wh][2][r];
                acc[2 + wh][0][r] = (wacc[wh][1][r] - wacc[wh][2][r]) - wacc[wh][3][r];
            }
    }
    if (g.fold_pad > 0) {
        // ---- fused adjoint of the replicate padding (vxb_fold_pad_f32 without the round trip through HBM): the tile goes
        // to LDS as fp32 [256 voxels][64 ch] (the halo / weight buffers are free now); the voxel that is the first of the
        // padded voxels clamping to an output voxel sums its (pad+1)-wide border group -- the host checked that every
        // group lies inside one tile -- and updates the destination
        float* ft = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MLIM; i += ST) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
                int dd, hh, ww;
                rowmap(i, m, dd, hh, ww);
                const int pos = (dd * TH + hh) * TW + ww;
#pragma unroll
                for (int j = 0; j < NT; ++j) ft[pos * 64 + (wn * NT + j) * 32 + lq] = acc[i][j][r];
            }
        }
        __syncthreads();
        if constexpr (PM == 2) {
            if (g.fold_acc[1] == 2) {           // "wgin" (uniform): see HaloArgs
                const int P = g.fold_pad, S = g.fold_S;
                const float* __restrict__ yv = g.fold_y[0];
                const float* __restrict__ wgin_x = g.fold_y[1];
                float* __restrict__ wgin_part = g.fold_dst[1];
                float wgx[4][10], wgb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wgb[e] = 0.f;
#pragma unroll
                    for (int jx = 0; jx < 10; ++jx) wgx[e][jx] = 0.f;
                }
                for (int item = tid; item < TD * TH * TW * 16; item += NTH) {
                    const int c4 = (item & 15) * 4, pos = item >> 4;
                    const int wl = pos % TW, hl = (pos / TW) % TH, dl = pos / (TW * TH);
                    const int id = d0 + dl, ih = h0 + hl, iw = w0 + wl;
                    if (id >= g.S_out || ih >= g.S_out || iw >= g.S_out) continue;
                    if ((id >= 1 && id <= P) || id > S - 1 + P || (ih >= 1 && ih <= P) || ih > S - 1 + P || (iw >= 1 && iw <= P) || iw > S - 1 + P) continue;
                    const int nd = (id == 0 || id == S - 1 + P) ? P + 1 : 1;
                    const int nh = (ih == 0 || ih == S - 1 + P) ? P + 1 : 1;
                    const int nw = (iw == 0 || iw == S - 1 + P) ? P + 1 : 1;
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int dd = 0; dd < nd; ++dd)
                        for (int hh = 0; hh < nh; ++hh)
                            for (int ww = 0; ww < nw; ++ww) {
                                const float4 v = *reinterpret_cast<const float4*>(&ft[(((dl + dd) * TH + hl + hh) * TW + wl + ww) * 64 + c4]);
                                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                            }
                    const int jd = min(max(id - P, 0), S - 1), jh = min(max(ih - P, 0), S - 1), jw = min(max(iw - P, 0), S - 1);
                    const long long vox = (((long long)b * S + jd) * S + jh) * S + jw;
                    const float4 yy = *reinterpret_cast<const float4*>(yv + vox * 64 + c4);
                    a.x = yy.x > 0.f ? a.x : a.x * g.slope; a.y = yy.y > 0.f ? a.y : a.y * g.slope;
                    a.z = yy.z > 0.f ? a.z : a.z * g.slope; a.w = yy.w > 0.f ? a.w : a.w * g.slope;
                    const float* xv = wgin_x + vox * 10;
                    float xr[10];
#pragma unroll
                    for (int jx = 0; jx < 10; jx += 2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(xv + jx);
                        xr[jx] = t2.x; xr[jx + 1] = t2.y;
                    }
                    // every input value in a register of its own, so that the packed fp32 FMAs below take it as the low half of their
                    // operand.  With the odd values picked out of a loaded register pair instead (v_pk_fma_f32 ... op_sel:[0,1,0], what
                    // the compiler emits without this) the LOW lane of the result missed a term in about one workgroup in 15, differently
                    // from run to run; build with -DWGIN_NO_OPAQUE to see it (tools/experiments/README.md, wgin_pk_fma_repro.py)
#ifndef WGIN_NO_OPAQUE
#pragma unroll
                    for (int jx = 0; jx < 10; ++jx) asm volatile("" : "+v"(xr[jx]));
#endif
                    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        wgb[e] += av[e];
#pragma unroll
                        for (int jx = 0; jx < 10; ++jx) wgx[e][jx] = fmaf(av[e], xr[jx], wgx[e][jx]);
                    }
                }
                // fixed-order fold of the NTH / 16 threads that share a channel quad: 11 values (10 inputs + bias) per channel
                __syncthreads();                // everyone is done reading the fp32 tile in ft
                float4* wred = reinterpret_cast<float4*>(ft);          // [11][NTH] float4 (45 KB at 256 threads)
#pragma unroll
                for (int jx = 0; jx < 10; ++jx) wred[jx * NTH + tid] = make_float4(wgx[0][jx], wgx[1][jx], wgx[2][jx], wgx[3][jx]);
                wred[10 * NTH + tid] = make_float4(wgb[0], wgb[1], wgb[2], wgb[3]);
                __syncthreads();
                for (int t2 = tid; t2 < 16 * 11; t2 += NTH) {
                    const int quad = t2 & 15, jx = t2 >> 4;
                    float4 acc4 = wred[jx * NTH + quad];
                    for (int j2 = 1; j2 < NTH / 16; ++j2) {
                        const float4 b4 = wred[jx * NTH + quad + 16 * j2];
                        acc4.x += b4.x; acc4.y += b4.y; acc4.z += b4.z; acc4.w += b4.w;
                    }
                    float* po = wgin_part + (long long)blockIdx.x * 704 + (quad * 4) * 11 + jx;
                    po[0] = acc4.x; po[11] = acc4.y; po[22] = acc4.z; po[33] = acc4.w;
                }
                return;
            }
        }
        const int nb = n0 / N, P = g.fold_pad, S = g.fold_S;
        float* __restrict__ dst = g.fold_dst[nb];
        const float* __restrict__ yv = g.fold_y[nb];
        const int facc = g.fold_acc[nb];
        float amxf = 0.f;
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int item = tid; item < TD * TH * TW * 16; item += NTH) {
            const int c4 = (item & 15) * 4, pos = item >> 4;
            const int wl = pos % TW, hl = (pos / TW) % TH, dl = pos / (TW * TH);
            const int id = d0 + dl, ih = h0 + hl, iw = w0 + wl;
            if (id >= g.S_out || ih >= g.S_out || iw >= g.S_out) continue;
            // owner test and group span per axis: i in [1, P] belongs to the group of i = 0; i > S-1+P to that of S-1+P
            if ((id >= 1 && id <= P) || id > S - 1 + P || (ih >= 1 && ih <= P) || ih > S - 1 + P || (iw >= 1 && iw <= P) || iw > S - 1 + P) continue;
            const int nd = (id == 0 || id == S - 1 + P) ? P + 1 : 1;
            const int nh = (ih == 0 || ih == S - 1 + P) ? P + 1 : 1;
            const int nw = (iw == 0 || iw == S - 1 + P) ? P + 1 : 1;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int dd = 0; dd < nd; ++dd)
                for (int hh = 0; hh < nh; ++hh)
                    for (int ww = 0; ww < nw; ++ww) {
                        const float4 v = *reinterpret_cast<const float4*>(&ft[(((dl + dd) * TH + hl + hh) * TW + wl + ww) * 64 + c4]);
                        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                    }
            if (PM >= 2) { a.x *= out_sc; a.y *= out_sc; a.z *= out_sc; a.w *= out_sc; }
            const int jd = min(max(id - P, 0), S - 1), jh = min(max(ih - P, 0), S - 1), jw = min(max(iw - P, 0), S - 1);
            const long long o = ((((long long)b * S + jd) * S + jh) * S + jw) * 64 + c4;
            if (facc) {
                const float4 pv = *reinterpret_cast<const float4*>(dst + o);
                a.x += pv.x; a.y += pv.y; a.z += pv.z; a.w += pv.w;
            }
            if (yv) {
                const float4 yy = *reinterpret_cast<const float4*>(yv + o);
                a.x = yy.x > 0.f ? a.x : a.x * g.slope; a.y = yy.y > 0.f ? a.y : a.y * g.slope;
                a.z = yy.z > 0.f ? a.z : a.z * g.slope; a.w = yy.w > 0.f ? a.w : a.w * g.slope;
            }
            *reinterpret_cast<float4*>(dst + o) = a;
            amxf = fmaxf(fmaxf(amxf, fabsf(a.x)), fmaxf(fmaxf(fabsf(a.y), fabsf(a.z)), fabsf(a.w)));
            csum.x += a.x; csum.y += a.y; csum.z += a.z; csum.w += a.w;      // (item & 15 is the same for all of a thread's items)
        }
        if (g.colsum_part) {                // (uniform) fixed-order fold of the 16 threads that share a channel quad
            __syncthreads();                // everyone is done reading the fp32 tile in ft
            float4* cred = reinterpret_cast<float4*>(ft);
            cred[tid] = csum;
            __syncthreads();
            if (tid < 16) {
                float4 a = cred[tid];
                for (int j = 1; j < NTH / 16; ++j) { const float4 b = cred[tid + 16 * j]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
                *reinterpret_cast<float4*>(g.colsum_part + (long long)blockIdx.x * 64 + 4 * tid) = a;
            }
        }
        if (g.amax_part) {                  // (uniform) the fp16 operand scale of the tensor just written is taken on the way
            __shared__ unsigned famx[8];
            unsigned amx = vxb_amax_word(amxf, (csum.x + csum.y) + (csum.z + csum.w));      // (fmaxf drops NaN: the column sums keep it)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amx = max(amx, (unsigned)__shfl_xor((int)amx, o, 64));
            if (lane == 0) famx[wid] = amx;
            __syncthreads();
            if (tid == 0) {
                unsigned m = 0;
                for (int w = 0; w < NW; ++w) m = max(m, famx[w]);
                g.amax_part[blockIdx.x] = m;
            }
        }
        return;
    }
    // ---- epilogue: acc[i][j][r] = C[voxel row (r&3) + 8*(r>>2) + 4*hi of M tile i][channel j*32 + lq]
    constexpr bool SSOK = WD && !TL && WN == 2 && NW == 4 && NT == 1;
    const bool ss_on = SSOK && g.ss_part != nullptr;              // (uniform)
    SsPart sa;
    sa.m = -INFINITY; sa.s = 0.f; sa.sx = 0.f; sa.sy = 0.f; sa.sz = 0.f; sa.xmax = -INFINITY; sa.arg = 0x7fffffff;
    const DivT divT(0.01f);
    // one group = the four rows r4 .. r4 + 3 of M tile i: bias + activation + store, and (ss_on) the group's term of this lane's
    // channel statistics exactly as ss_update4 forms it (vox_ops.hip): one rescale of the running sums, then four terms.  ASC: the
    // caller visits this lane's voxels in ascending index order, so a strict > keeps the lowest index among equal maxima.
    auto do_group = [&](int i, int r4, bool asc) {
        float xv[4], lv[4], wxv[4], wyv[4], wzv[4];
        bool okv[4];
        float mn = sa.m;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = u + 8 * (r4 >> 2) + 4 * hi;
            int dd, hh, ww;
            rowmap(i, m, dd, hh, ww);
            const int od = d0 + dd, oh = h0 + hh, ow = w0 + ww;
            okv[u] = od < g.S_out && oh < g.S_out && ow < g.S_out;
            float vj[NT];
            if (okv[u]) {
                float* op;
                if (g.d2s_s > 0) {
                    // depth-to-space output: this workgroup's 64 columns are one phase of the fine grid
                    const int s = g.d2s_s, ph = n0 / N;
                    const long long Vf = (long long)g.S_out * s;
                    op = g.out + ((((long long)b * Vf + od * s + ph / (s * s)) * Vf + oh * s + (ph / s) % s) * Vf + ow * s + ph % s) * N - n0;
                } else {
                    op = g.out + ((((long long)b * g.S_out + od) * g.S_out + oh) * g.S_out + ow) * g.N;
                    if (TL) op += kpart * g.part_stride;
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = n0 + (wn * NT + j) * 32 + lq;
                    float v = (PM >= 2 ? acc[i][j][r4 + u] * out_sc : acc[i][j][r4 + u]) + (g.bias ? g.bias[n] : 0.f);
                    if (g.act == 1) v = v > 0.f ? v : v * g.slope;
                    if (!(g.dbg & 0x40)) op[n] = v;                  // (0x40: no output stores)
                    vj[j] = v;
                }
            }
            if (SSOK && ss_on) {
                const float v = okv[u] ? vj[0] : 0.f;
                const int pidx = (od * g.S_out + oh) * g.S_out + ow;
                xv[u] = v;
                lv[u] = divT(v);
                mn = okv[u] ? fmaxf(mn, lv[u]) : mn;
                const bool gt = okv[u] && (v > sa.xmax || (!asc && v == sa.xmax && pidx < sa.arg));
                sa.xmax = gt ? v : sa.xmax;
                sa.arg = gt ? pidx : sa.arg;
                wyv[u] = slin[okv[u] ? od : 0]; wxv[u] = slin[okv[u] ? oh : 0]; wzv[u] = slin[okv[u] ? ow : 0];   // meshgrid 'xy' quirk
            }
        }
        if (SSOK && ss_on) {
            const float f = sa.m > -INFINITY ? exp_v(sa.m - mn) : 0.f;
            sa.s *= f; sa.sx *= f; sa.sy *= f; sa.sz *= f;
            sa.m = mn;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float e = okv[u] ? exp_v(lv[u] - mn) : 0.f;
                sa.s += e; sa.sx = fmaf(e, wxv[u], sa.sx); sa.sy = fmaf(e, wyv[u], sa.sy); sa.sz = fmaf(e, wzv[u], sa.sz);
            }
        }
    };
    if (HALF == 0 && MTW == 4) {
        // a wave's voxels in ascending index order: depth (i >> 1), then the row pair of the group (h = 2 (r4 >> 2) + hi), then the w half
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4)
#pragma unroll
                for (int iw = 0; iw < 2; ++iw) do_group(2 * ip + iw, r4, true);
    } else {
#pragma unroll
        for (int i = 0; i < MLIM; i += ST)
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) do_group(i, r4, false);
    }
    if (SSOK && ss_on) {
        // fixed-order merge: the two row halves of a wave (lanes l, l + 32), then the two waves that share the column tile
        SsPart ob;
        ob.m = __shfl_xor(sa.m, 32, 64); ob.s = __shfl_xor(sa.s, 32, 64); ob.sx = __shfl_xor(sa.sx, 32, 64); ob.sy = __shfl_xor(sa.sy, 32, 64);
        ob.sz = __shfl_xor(sa.sz, 32, 64); ob.xmax = __shfl_xor(sa.xmax, 32, 64); ob.arg = __shfl_xor(sa.arg, 32, 64);
        if (hi == 0) ss_merge(sa, ob);
        SsPart* sred = reinterpret_cast<SsPart*>(smem);          // (the halo is free once every wave has left the tap loop)
        __syncthreads();
        if (wm == 1 && hi == 0) sred[wn * 32 + lq] = sa;
        __syncthreads();
        if (wm == 0 && hi == 0) {
            ss_merge(sa, sred[wn * 32 + lq]);
            const int ntile = g.ntd * g.nth * g.ntw;
            const int tile = (td * g.nth + th) * g.ntw + tw;
            reinterpret_cast<SsPart*>(g.ss_part)[((long long)b * g.N + n0 + wn * 32 + lq) * ntile + tile] = sa;
        }
    }
}

template <int NTG, int PM, int NW, int WD, int TL = 0, int WN = 1, int WG = 0, int BLN = 0>
__global__ void __launch_bounds__(NW * 64, 2) conv3_halo_kernel(HaloArgs g) {
    constexpr bool EDGE = WD && WN == 2 && NW == 4;
    if constexpr (WG) {
        // (the host checked S_out % 2 == 0: a depth tile holds two or four outputs -- the idle wave row of a two-deep tile only helps
        // staging; a last row / column tile of 6 runs in full, of 4 or less as a half tile)
        const int nwg = gridDim.x, lid = blockIdx.x;
        const int xcd = lid & 7, slot = lid >> 3, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        t /= g.N / (NTG * 32);
        int td, th, tw;
        halo_tile_coords(g, t % (g.ntd * g.nth * g.ntw), td, th, tw);
        if (min(g.S_out - tw * TW, g.S_out - th * TH) <= 4 && !g_dbg_all_waves(g)) conv3_halo_body<NTG, PM, NW, WD, TL, WN, 1, WG>(g);
        else conv3_halo_body<NTG, PM, NW, WD, TL, WN, 0, WG>(g);
        return;
    } else
    if constexpr (EDGE) {
        // this workgroup's tile (the body decodes it again): how many of its 8 columns (else: rows) lie inside the output grid?
        const int nwg = gridDim.x, lid = blockIdx.x;
        const int xcd = lid & 7, slot = lid >> 3, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        t /= g.N / (NTG * 32);
        if (TL) t %= g.B * g.ntd * g.nth * g.ntw;
        int td, th, tw;
        halo_tile_coords(g, t % (g.ntd * g.nth * g.ntw), td, th, tw);
        if (g.S_out - td * TD <= 2 && !g_dbg_all_waves(g)) {
            conv3_halo_body<NTG, PM, NW, WD, TL, WN, 3>(g);
            return;
        }
        const int rem = min(g.S_out - tw * TW, g.S_out - th * TH);
        if (rem <= 6 && !g_dbg_all_waves(g)) {
            if (rem <= 4) conv3_halo_body<NTG, PM, NW, WD, TL, WN, 1>(g);
            else conv3_halo_body<NTG, PM, NW, WD, TL, WN, 2>(g);
            return;
        }
    }
    conv3_halo_body<NTG, PM, NW, WD, TL, WN, 0, 0, BLN>(g);
}

inline bool hb_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

int g_halo_waves = 4;
int g_halo_dbg = 0;
int g_halo_wn = 0;         // experiment knob (vxb_debug_set_halo_wn): waves along N in the WD kernels, 1 or 2; 0 = default (2 in
                          // 'bf16x3'; 1 in 'bf16': no difference).  In the training step on one box (VOXACTB_HALO_WN=1 vs 2, two
                          // repetitions each, B = 16, S = 100): forward 20.23 -> 19.43 ms, data gradient + padding adjoint
                          // 21.89 -> 20.85 ms; the tap-list variant (up-conv data gradient) is unchanged, 11.0 ms either way        // experiment knob (vxb_debug_set_halo_waves): 4 waves x 2 M tiles or 8 waves x 1 M tile per workgroup

template <int NT, int PM, int NW, int WD, int TL = 0, int WN = 1, int WG = 0, int BLN = 0>
int hb_launch(const HaloArgs& g, long long nblk, hipStream_t st) {
    // (the fold epilogue re-uses the buffer as an fp32 [256][64] tile: keep the full size in every variant)
    const size_t lds = WG == 2 ? (size_t)(8 * HHp * HW_USED * 32 + 6 * 2048) * sizeof(u16)      // compact image + the fragment ring (conv3_halo_body: BL)
                     : WG ? (size_t)8 * HHp * HWp * SP * sizeof(u16)
                     : BLN ? (size_t)(HALO_SLOTS * SP + 5 * 2048) * sizeof(u16)                   // padded image + the fragment ring (BN)
                     : (size_t)(HALO_SLOTS * SP + 2 * NT * 32 * LDW) * sizeof(u16);
    if (TL && (size_t)(g.ncls * 32 + g.nphase * 2) * sizeof(int) > (size_t)2 * NT * 32 * LDW * sizeof(u16)) return VXB_ESIZE;
    if (hipFuncSetAttribute((const void*)conv3_halo_kernel<NT, PM, NW, WD, TL, WN, WG, BLN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return VXB_ELAUNCH;
    hipLaunchKernelGGL((conv3_halo_kernel<NT, PM, NW, WD, TL, WN, WG, BLN>), dim3((unsigned)(nblk * (g.N / (NT * 32)) * (TL ? g.ksplit : 1))), dim3(NW * 64), lds, st, g);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

int hb_impl(int x3 /* product mode: 0 bf16, 1 bf16x3, 2 fp16 (WD kernels only, `scale` = {in, 1 / in} on the device) */, const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out, int off, int replicate,
            const void* wt_bf16, int N, const float* bias, float* out, int act, float slope, int s2d_s, int s2d_C,
            int d2s_s, vxb_stream_t stream, const HaloArgs* fold = nullptr, const void* wfrag = nullptr,
            const int32_t* taptab = nullptr, int ncls = 0, int nphase = 0, int tap_total = 0, const float* scale = nullptr,
            int ksplit = 1, const int32_t* kparts = nullptr, float* ss_part = nullptr, const float* ss_lin = nullptr, int wino = 0) {
    // wino: Winograd F(2, 3) along depth (see conv3_halo_body): bf16x3, fragment-order weights of the 36 transformed taps, whole depth tiles
    if (wino && (x3 < 1 || x3 > 3 || !wfrag || taptab || s2d_s > 0 || d2s_s > 0 || N != 64 || (S_out & 1) || (g_halo_wn && g_halo_wn != 2)))
        return VXB_EARG;
    if (x3 == 2 && (!wfrag || taptab)) return VXB_EARG;
    if (x3 == 3 && (!wfrag || !scale || src1 || replicate || d2s_s > 0)) return VXB_EARG;     // data gradients only
    if (!src0 || !wt_bf16 || (!out && !fold) || B < 1 || S_in < 1 || S_out < 1) return VXB_EARG;
    if ((C0 & 31) || (C1 & 31) || C0 < 32 || (C1 > 0 && !src1) || N < 64 || (N & 63)) return VXB_ESIZE;
    if (!hb_aligned16(src0) || !hb_aligned16(wt_bf16) || (src1 && !hb_aligned16(src1))) return VXB_ESIZE;
    if (s2d_s > 0 && (C1 != 0 || s2d_C < 32 || (s2d_C & 31) || C0 != s2d_s * s2d_s * s2d_s * s2d_C)) return VXB_EARG;
    const long long Vin = (long long)S_in * (s2d_s > 0 ? s2d_s : 1);
    if (Vin * Vin * Vin >= INT32_MAX) return VXB_ESIZE;
    if (taptab && (!wfrag || s2d_s <= 0 || fold || ncls < 1 || tap_total < 3 || nphase != s2d_s * s2d_s * s2d_s)) return VXB_EARG;
    if (ksplit < 1 || ksplit > 16 || (ksplit > 1 && (!taptab || !kparts || fold))) return VXB_EARG;
    HaloArgs g;
    g.ksplit = ksplit; g.kparts = kparts; g.part_stride = (long long)B * S_out * S_out * S_out * N;
    g.ss_part = ss_part; g.ss_lin = ss_lin;
    if (ss_part && (x3 != 1 || !wfrag || taptab || fold || N != 64 || !ss_lin || d2s_s > 0 || S_out > 256 || (g_halo_wn && g_halo_wn != 2))) return VXB_EARG;
    g.taptab = taptab; g.ncls = ncls; g.nphase = nphase; g.tap_total = tap_total;
    g.s2d_s = s2d_s; g.s2d_C = s2d_C; g.d2s_s = d2s_s; g.wfrag = (const u16*)wfrag;
    g.dbg = g_halo_dbg;
    g.scale = scale;
    g.amax_part = fold ? fold->amax_part : nullptr;
    g.colsum_part = fold ? fold->colsum_part : nullptr;
    g.fold_pad = 0; g.fold_S = 0; g.fold_dst[0] = g.fold_dst[1] = nullptr; g.fold_y[0] = g.fold_y[1] = nullptr;
    g.fold_acc[0] = g.fold_acc[1] = 0;
    if (fold) {
        g.fold_pad = fold->fold_pad; g.fold_S = fold->fold_S;
        for (int i = 0; i < 2; ++i) { g.fold_dst[i] = fold->fold_dst[i]; g.fold_y[i] = fold->fold_y[i]; g.fold_acc[i] = fold->fold_acc[i]; }
    }
    g.src0 = src0; g.src1 = src1; g.C0 = C0; g.C1 = C1; g.B = B; g.S_in = S_in; g.S_out = S_out; g.off = off;
    g.replicate = replicate; g.wb = (const u16*)wt_bf16; g.N = N; g.K = 27 * (C0 + C1); g.bias = bias; g.out = out;
    g.act = act; g.slope = slope;
    g.ntd = vxb_cdiv(S_out, TD); g.nth = vxb_cdiv(S_out, TH); g.ntw = vxb_cdiv(S_out, TW);
    const long long nblk = (long long)B * g.ntd * g.nth * g.ntw;
    if (nblk >= INT32_MAX) return VXB_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    // 64 output channels per workgroup (162-225 VGPRs -> two workgroups per CU); N = 128 runs two column blocks that each
    // stage the halo -- cheaper than the register spills of a 128-wide accumulator tile.
    if (wino && x3 != 2 && !(g.dbg & 0x1000))          // (experiment bit 0x1000: the fragments from global memory per wave, rounds 5 - 6)
        return x3 == 3 ? hb_launch<2, 3, 4, 1, 0, 2, 2>(g, nblk, st) : hb_launch<2, 1, 4, 1, 0, 2, 2>(g, nblk, st);
    if (wino) return x3 == 3 ? hb_launch<2, 3, 4, 1, 0, 2, 1>(g, nblk, st) : x3 == 2 ? hb_launch<2, 2, 4, 1, 0, 2, 1>(g, nblk, st)
                                                                               : hb_launch<2, 1, 4, 1, 0, 2, 1>(g, nblk, st);
    if (x3 == 2) return (g.dbg & 0x1000) ? hb_launch<2, 2, 4, 1>(g, nblk, st) : hb_launch<2, 2, 4, 1, 0, 1, 0, 1>(g, nblk, st);
    if (x3 == 3) return g.taptab ? hb_launch<2, 3, 4, 1, 1, 2>(g, nblk, st) : hb_launch<2, 3, 4, 1, 0, 2>(g, nblk, st);
    const int wn = g_halo_wn ? g_halo_wn : (x3 ? 2 : 1);
    if (g.taptab && wn == 2) return x3 ? hb_launch<2, 1, 4, 1, 1, 2>(g, nblk, st) : hb_launch<2, 0, 4, 1, 1, 2>(g, nblk, st);
    if (g.taptab) return x3 ? hb_launch<2, 1, 4, 1, 1>(g, nblk, st) : hb_launch<2, 0, 4, 1, 1>(g, nblk, st);
    if (g.wfrag && wn == 2) return x3 ? hb_launch<2, 1, 4, 1, 0, 2>(g, nblk, st) : hb_launch<2, 0, 4, 1, 0, 2>(g, nblk, st);
    if (g.wfrag && !(g.dbg & 0x1000)) return x3 ? hb_launch<2, 1, 4, 1, 0, 1, 0, 1>(g, nblk, st) : hb_launch<2, 0, 4, 1, 0, 1, 0, 1>(g, nblk, st);
    if (g.wfrag) return x3 ? hb_launch<2, 1, 4, 1>(g, nblk, st) : hb_launch<2, 0, 4, 1>(g, nblk, st);
    if (g_halo_waves == 8) return x3 ? hb_launch<2, 1, 8, 0>(g, nblk, st) : hb_launch<2, 0, 8, 0>(g, nblk, st);
    return x3 ? hb_launch<2, 1, 4, 0>(g, nblk, st) : hb_launch<2, 0, 4, 0>(g, nblk, st);
}

}  // namespace

extern "C" void vxb_debug_set_halo_waves(int nw) { g_halo_waves = nw == 8 ? 8 : 4; }
extern "C" void vxb_debug_set_halo_experiment(int bits) { g_halo_dbg = bits; }
extern "C" void vxb_debug_set_halo_wn(int wn) { g_halo_wn = (wn == 1 || wn == 2) ? wn : 0; }

// 3x3x3, stride-1 twin of vxb_conv3d_bf16w_f32 (same weights layout bf16 [N][27*(C0+C1)], same padding semantics:
// src voxel = out + tap + off per axis); C0, C1 multiples of 32, N a multiple of 64.  out [B, S_out^3, N] is overwritten.
// s2d_s > 0: src0 is a fine grid [B, (S_in*s2d_s)^3, s2d_C] read by space-to-depth (input channel = (phase, co), C0 =
// s^3 * s2d_C) -- the data gradient of the polyphase up-conv.  With taptab (s2d + wfrag only) the weights are block-sparse:
// see HaloArgs::taptab; wfrag then lists only the non-zero taps of every chunk (ops.halo_wfrag_sparse).  d2s_s > 0: depth-to-space output with 64 channels per
// phase (N = d2s_s^3 * 64), out = fine grid [B, (S_out*d2s_s)^3, 64] -- the polyphase up-conv forward.
extern "C" int vxb_conv3_halo_bf16w_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                        int off, int replicate, const void* wt_bf16, int N, const float* bias, float* out,
                                        int act, float slope, int s2d_s, int s2d_C, int d2s_s, const void* wfrag, const int32_t* taptab,
                                         int ncls, int tap_total, vxb_stream_t stream) {
    return hb_impl(0, src0, src1, C0, C1, B, S_in, S_out, off, replicate, wt_bf16, N, bias, out, act, slope, s2d_s, s2d_C, d2s_s,
                   stream, nullptr, wfrag, taptab, ncls, s2d_s * s2d_s * s2d_s, tap_total);
}

// 'bf16x3' twin (weights = planes [2][N][K], see vxb_conv3d_bf16x3_f32).
extern "C" int vxb_conv3_halo_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                         int off, int replicate, const void* wt_bf16, int N, const float* bias, float* out,
                                         int act, float slope, int s2d_s, int s2d_C, int d2s_s, const void* wfrag, const int32_t* taptab,
                                         int ncls, int tap_total, vxb_stream_t stream) {
    return hb_impl(1, src0, src1, C0, C1, B, S_in, S_out, off, replicate, wt_bf16, N, bias, out, act, slope, s2d_s, s2d_C, d2s_s,
                   stream, nullptr, wfrag, taptab, ncls, s2d_s * s2d_s * s2d_s, tap_total);
}

// vxb_conv3_halo_bf16x3_f32 for a 64-column conv (two concatenated sources, replicate padding, S_in = S_out = S, fragment-order
// weights) followed by SpatialSoftmax3D + global max pool of its OUTPUT (vxb_ss3d_max_fwd_f32 on `out`: same outputs, same argmax
// rule) -- the statistics of every (tile, channel) are taken in the conv's epilogue while the tile is in registers and merged by one
// small launch: `final` + ss_final of the Q-function (perceiver_lang_io.py:462, :470) without the 4.1 GB statistics pass over u.
// part_ws: vxb_conv3_halo_ss3d_ws(B, S) floats.  The conv output is bit-identical to the plain entry; the pooled features differ
// from the statistics kernel's by the association of the partial sums (~1e-7 relative).
extern "C" size_t vxb_conv3_halo_ss3d_ws(int B, int S) {
    if (B < 1 || S < 1) return 0;
    return (size_t)B * 64 * vxb_cdiv(S, TD) * vxb_cdiv(S, TH) * vxb_cdiv(S, TW) * 7;
}
extern "C" int vxb_conv3_halo_ss3d_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S, const void* wt_bf16,
                                              const float* bias, float* out, int act, float slope, const void* wfrag, const float* lin,
                                              float* part_ws, float* out_ss, float* out_max, float* stats, int32_t* argmax,
                                              vxb_stream_t stream) {
    if (!wfrag || !lin || !part_ws || !out_ss || !out_max || !stats || !argmax) return VXB_EARG;
    const int rc = hb_impl(1, src0, src1, C0, C1, B, S, S, -1, 1, wt_bf16, 64, bias, out, act, slope, 0, 0, 0, stream, nullptr, wfrag,
                           nullptr, 0, 0, 0, nullptr, 1, nullptr, part_ws, lin);
    if (rc) return rc;
    return vxb_ss3d_final_tiles_launch(part_ws, vxb_cdiv(S, TD) * vxb_cdiv(S, TH) * vxb_cdiv(S, TW), B, 64, out_ss, out_max, stats, argmax,
                                       (hipStream_t)stream);
}

// vxb_conv3_halo_ss3d_bf16x3_f32 with the depth axis of the 3x3x3 filter evaluated by Winograd's F(2, 3) (four products per two output
// depths instead of six: 2/3 of the matrix work; conv3_halo_body, WG): wfrag_wg = ops.halo_wfrag_wg of the weights -- the 36 transformed
// taps (xi, kh, kw) in fragment order.  S even (a two-deep last depth tile idles one wave row).  The transforms are exact up to fp32 rounding of sums of two inputs / three
// weights; results agree with the direct kernel to ~1e-6 relative (tests/test_halo_winograd_gpu.py), not bit for bit.
extern "C" int vxb_conv3_halo_ss3d_wg_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S, const float* bias,
                                                 float* out, int act, float slope, const void* wfrag_wg, const float* lin,
                                                 float* part_ws, float* out_ss, float* out_max, float* stats, int32_t* argmax,
                                                 vxb_stream_t stream) {
    if (!wfrag_wg || !lin || !part_ws || !out_ss || !out_max || !stats || !argmax || (S & 1)) return VXB_EARG;
    const int rc = hb_impl(1, src0, src1, C0, C1, B, S, S, -1, 1, wfrag_wg, 64, bias, out, act, slope, 0, 0, 0, stream, nullptr, wfrag_wg,
                           nullptr, 0, 0, 0, nullptr, 1, nullptr, part_ws, lin, 1);
    if (rc) return rc;
    return vxb_ss3d_final_tiles_launch(part_ws, vxb_cdiv(S, TD) * vxb_cdiv(S, TH) * vxb_cdiv(S, TW), B, 64, out_ss, out_max, stats, argmax,
                                       (hipStream_t)stream);
}

// The tap-list launch of vxb_conv3_halo_bf16x3_f32 / _bf16w_f32 (space-to-depth input, block-sparse weights: the polyphase up-conv's
// data gradient) with the reduction split over ksplit workgroups per tile: part p accumulates the chunks kparts[p] .. kparts[p + 1] - 1
// (kparts: DEVICE int32 [ksplit + 1], kparts[0] = 0, kparts[ksplit] = C0 / (16 or 32 channels per chunk)) into out_parts[p] [B, S_out^3, N];
// the caller sums the parts (vxb_sum_splits_f32).  B * ceil(S_out/4) * ceil(S_out/8)^2 tiles of 8000 input channels are 864 workgroups
// of ~5 ms at the step's size -- 1.7 rounds of the chip's 512 slots; with ksplit parts the tail is a ksplit-th as long.
// x3 = 3 ("fp16x2", see the product modes in the kernel): src_fine is multiplied by scale[0] (device, a power of two:
// vxb_absmax_scale_f32) and carried as an fp16 hi + lo pair, the weights are ONE fp16 value each -- wfrag = ops.halo_wfrag_x2 of the
// fp16 [N][27 C0] matrix, listed taps only; wt_bf16 is then only checked for alignment (pass wfrag) -- and every part is multiplied
// by scale[1]: two MFMAs per product instead of three.
extern "C" int vxb_conv3_s2d_splitk_f32(const float* src_fine, int C0, int B, int S_in, int S_out, int off, const void* wt_bf16, int x3,
                                        int N, float* out_parts, int s2d_s, int s2d_C, const void* wfrag, const int32_t* taptab,
                                        int ncls, int tap_total, int ksplit, const int32_t* kparts, const float* scale,
                                        vxb_stream_t stream) {
    if (!taptab || !wfrag || !kparts || ksplit < 1 || x3 < 0 || x3 == 2 || x3 > 3 || (x3 == 3) != (scale != nullptr)) return VXB_EARG;
    return hb_impl(x3, src_fine, nullptr, C0, 0, B, S_in, S_out, off, 0, wt_bf16, N, nullptr, out_parts, 0, 0.f, s2d_s, s2d_C, 0,
                   stream, nullptr, wfrag, taptab, ncls, s2d_s * s2d_s * s2d_s, tap_total, scale, ksplit, kparts);
}

// Data gradient of a 3x3x3 replicate-padded conv fused with the adjoint of its padding (vxb_conv3_halo_* followed by
// vxb_fold_pad_f32, without writing the padded-domain gradient): dy [B, S^3, C0] -> for each 64-column block nb of the
// N <= 128 input channels, dst_nb [B, S^3, 64] (+)= fold(conv_zero_pad(dy, wt_dgrad)) (* LeakyReLU'(y_nb) when y_nb is
// given).  x3 != 0: weights are the [2][N][K] planes ('bf16x3').  pad = 1.
extern "C" int vxb_conv3_dgrad_fold_f32(const float* dy, int C0, int B, int S, const void* wt_bf16, int x3, int N, float* dst0,
                                        float* dst1, const float* y0, const float* y1, int acc0, int acc1, float slope,
                                        const void* wfrag, float* dst_scale, float* scale_ws, float* dst_colsum, float* colsum_ws,
                                        vxb_stream_t stream) {
    // dst_colsum (optional, [64], ACCUMULATED; needs colsum_ws of 64 * vxb_conv3_dgrad_fold_blocks + 64 * 64 floats; N = 64 only): column
    // sums of dst0 as written (after the LeakyReLU' factor) = the bias gradient of the conv whose activation y0 is
    // dst_scale (optional, [2], needs scale_ws of vxb_conv3_dgrad_fold_blocks words; N = 64 only): the fp16 operand scale of dst0 as
    // vxb_absmax_scale_f32 would compute it, taken while dst0 is written
    if (!dy || !wt_bf16 || !dst0 || (N > 64 && !dst1) || N > 128 || S < 2) return VXB_EARG;
    if (dst_scale && (!scale_ws || N != 64)) return VXB_EARG;
    if (dst_colsum && (!colsum_ws || N != 64)) return VXB_EARG;
    const int pad = 1, S_out = S + 2 * pad;
    // every border group {0..pad} / {S-1+pad..S-1+2 pad} must lie inside one tile
    if ((S_out - 1 - pad) / TD != (S_out - 1) / TD || (S_out - 1 - pad) / TH != (S_out - 1) / TH || (S_out - 1 - pad) / TW != (S_out - 1) / TW)
        return VXB_ESIZE;
    HaloArgs f;
    f.fold_pad = pad; f.fold_S = S; f.fold_dst[0] = dst0; f.fold_dst[1] = dst1; f.fold_y[0] = y0; f.fold_y[1] = y1;
    f.fold_acc[0] = acc0; f.fold_acc[1] = acc1;
    f.amax_part = dst_scale ? reinterpret_cast<unsigned*>(scale_ws) : nullptr;
    f.colsum_part = dst_colsum ? colsum_ws : nullptr;
    int rc = hb_impl(x3 ? 1 : 0, dy, nullptr, C0, 0, B, S, S_out, -2 * pad, 0, wt_bf16, N, nullptr, nullptr, 0, slope, 0, 0, 0, stream, &f, wfrag);
    if (rc) return rc;
    const int nblk = (int)vxb_conv3_dgrad_fold_blocks(B, S, N);
    if (dst_colsum) {
        rc = vxb_rows64_sum_launch(colsum_ws, nblk, dst_colsum, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (!dst_scale) return VXB_OK;
    return vxb_absmax_finish_launch(f.amax_part, nblk, dst_scale, (hipStream_t)stream);
}

extern "C" size_t vxb_conv3_dgrad_fold_blocks(int B, int S, int N) {
    if (B < 1 || S < 2 || N < 64) return 0;
    const int So = S + 2;
    return (size_t)B * vxb_cdiv(So, TD) * vxb_cdiv(So, TH) * vxb_cdiv(So, TW) * (N / 64);
}

// The same data gradient + padding adjoint for ONE 64-column block with a single fp16 product per term: dy is multiplied by
// scale[0] (device, a power of two: vxb_absmax_scale_f32) before the conversion to half, the result by scale[1]; weights only in
// fragment order (ops.halo_wfrag of the fp16 [64][27 C0] matrix).  Used for the d(d0) half of `final`'s data gradient
// (perceiver_lang_io.py:462): that tensor only feeds the weight gradient of the 1x1x1 input conv -- a leaf: against the reference's
// gradients it is indistinguishable from the bf16x3 evaluation (tools/experiments/emu_precision.py --round4), while the d(u0)
// half, which propagates through the whole decoder and trunk, stays bf16x3.
static int dgrad_fold_f16(const float* dy, int C0, int B, int S, const void* wfrag_f16, float* dst, const float* y,
                          int acc, float slope, const float* scale, vxb_stream_t stream, int wino) {
    if (!dy || !wfrag_f16 || !dst || S < 2) return VXB_EARG;
    const int pad = 1, S_out = S + 2 * pad;
    if ((S_out - 1 - pad) / TD != (S_out - 1) / TD || (S_out - 1 - pad) / TH != (S_out - 1) / TH || (S_out - 1 - pad) / TW != (S_out - 1) / TW)
        return VXB_ESIZE;
    HaloArgs f;
    f.fold_pad = pad; f.fold_S = S; f.fold_dst[0] = dst; f.fold_dst[1] = nullptr; f.fold_y[0] = y; f.fold_y[1] = nullptr;
    f.fold_acc[0] = acc; f.fold_acc[1] = 0;
    f.amax_part = nullptr; f.colsum_part = nullptr;
    return hb_impl(2, dy, nullptr, C0, 0, B, S, S_out, -2 * pad, 0, wfrag_f16, 64, nullptr, nullptr, 0, slope, 0, 0, 0, stream, &f,
                   wfrag_f16, nullptr, 0, 0, 0, scale, 1, nullptr, nullptr, nullptr, wino);
}

extern "C" int vxb_conv3_dgrad_fold_f16_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16, float* dst, const float* y,
                                            int acc, float slope, const float* scale, vxb_stream_t stream) {
    return dgrad_fold_f16(dy, C0, B, S, wfrag_f16, dst, y, acc, slope, scale, stream, 0);
}
// ... with the depth axis by Winograd's F(2, 3): wfrag = ops.halo_wfrag_x2_wg (the 36 transformed taps in fp16, 16-channel chunks);
// S even.  See vxb_conv3_dgrad_fold_f16x2_wg_f32.
extern "C" int vxb_conv3_dgrad_fold_f16_wg_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16_wg, float* dst, const float* y,
                                               int acc, float slope, const float* scale, vxb_stream_t stream) {
    if ((S & 1) || !scale) return VXB_EARG;
    return dgrad_fold_f16(dy, C0, B, S, wfrag_f16_wg, dst, y, acc, slope, scale, stream, 1);
}

// One 64-column block of the data gradient + padding adjoint on TWO fp16 products per term ("fp16x2": dy * scale[0] as an fp16
// hi + lo pair, the weights as one fp16 value; wfrag_f16x2 = ops.halo_wfrag_x2 of the fp16 [64][27 C0] matrix): for the blocks that
// PROPAGATE -- the d(u0) half of `final`'s data gradient (perceiver_lang_io.py:462) -- with the optional by-products of
// vxb_conv3_dgrad_fold_f32 (operand scale of dst, column sums of dst).  Against the reference's gradients the weight rounding of a
// data gradient moves no gate (tools/experiments/emu_precision.py --round5), and a third of the MFMAs is gone.
static int dgrad_fold_f16x2(const float* dy, int C0, int B, int S, const void* wfrag_f16x2, float* dst, const float* y,
                            int acc, float slope, const float* scale, float* dst_scale, float* scale_ws,
                            float* dst_colsum, float* colsum_ws, vxb_stream_t stream, int wino) {
    if (!dy || !wfrag_f16x2 || !dst || !scale || S < 2) return VXB_EARG;
    if ((dst_scale && !scale_ws) || (dst_colsum && !colsum_ws)) return VXB_EARG;
    const int pad = 1, S_out = S + 2 * pad;
    if ((S_out - 1 - pad) / TD != (S_out - 1) / TD || (S_out - 1 - pad) / TH != (S_out - 1) / TH || (S_out - 1 - pad) / TW != (S_out - 1) / TW)
        return VXB_ESIZE;
    HaloArgs f;
    f.fold_pad = pad; f.fold_S = S; f.fold_dst[0] = dst; f.fold_dst[1] = nullptr; f.fold_y[0] = y; f.fold_y[1] = nullptr;
    f.fold_acc[0] = acc; f.fold_acc[1] = 0;
    f.amax_part = dst_scale ? reinterpret_cast<unsigned*>(scale_ws) : nullptr;
    f.colsum_part = dst_colsum ? colsum_ws : nullptr;
    int rc = hb_impl(3, dy, nullptr, C0, 0, B, S, S_out, -2 * pad, 0, wfrag_f16x2, 64, nullptr, nullptr, 0, slope, 0, 0, 0, stream, &f,
                     wfrag_f16x2, nullptr, 0, 0, 0, scale, 1, nullptr, nullptr, nullptr, wino);
    if (rc) return rc;
    const int nblk = (int)vxb_conv3_dgrad_fold_blocks(B, S, 64);
    if (dst_colsum) {
        rc = vxb_rows64_sum_launch(colsum_ws, nblk, dst_colsum, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (!dst_scale) return VXB_OK;
    return vxb_absmax_finish_launch(f.amax_part, nblk, dst_scale, (hipStream_t)stream);
}

extern "C" int vxb_conv3_dgrad_fold_f16x2_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16x2, float* dst, const float* y,
                                              int acc, float slope, const float* scale, float* dst_scale, float* scale_ws,
                                              float* dst_colsum, float* colsum_ws, vxb_stream_t stream) {
    return dgrad_fold_f16x2(dy, C0, B, S, wfrag_f16x2, dst, y, acc, slope, scale, dst_scale, scale_ws, dst_colsum, colsum_ws, stream, 0);
}
// ... with the filter's depth axis by Winograd's F(2, 3) (see vxb_conv3_halo_ss3d_wg_bf16x3_f32): wfrag = ops.halo_wfrag_x2_wg, the 36
// transformed taps rounded to fp16 AFTER the transform; S even.
extern "C" int vxb_conv3_dgrad_fold_f16x2_wg_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16x2_wg, float* dst,
                                                 const float* y, int acc, float slope, const float* scale, float* dst_scale,
                                                 float* scale_ws, float* dst_colsum, float* colsum_ws, vxb_stream_t stream) {
    if (S & 1) return VXB_EARG;
    return dgrad_fold_f16x2(dy, C0, B, S, wfrag_f16x2_wg, dst, y, acc, slope, scale, dst_scale, scale_ws, dst_colsum, colsum_ws, stream, 1);
}

// The same launch when the 64-column block is the data gradient of a 1x1x1 conv's OUTPUT y [B, S^3, 64] = lrelu(W_in x + b_in) whose
// input x [B, S^3, 10] is a detached tensor (the input conv of the Q-function, perceiver_lang_io.py:357; agent :100): that gradient only
// feeds dW_in [64][10] / db_in [64], so it is not stored -- the epilogue multiplies it with LeakyReLU'(y) and the voxel's inputs and the
// sums are ACCUMULATED into dW / db (4.1 GB less written here, 4.1 GB less read by vxb_pointwise_wgrad_ss3d_f32).
// ws: 704 * (vxb_conv3_dgrad_fold_blocks(B, S, 64) + 512) floats.
extern "C" int vxb_conv3_dgrad_fold_f16_wgin_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16, const float* y,
                                                 const float* x, float slope, const float* scale, float* ws, float* dW, float* db,
                                                 vxb_stream_t stream) {
    if (!dy || !wfrag_f16 || !y || !x || !scale || !ws || !dW || !db || S < 2) return VXB_EARG;
    if ((((uintptr_t)x) & 7) || (((uintptr_t)y) & 15)) return VXB_ESIZE;
    const int pad = 1, S_out = S + 2 * pad;
    if ((S_out - 1 - pad) / TD != (S_out - 1) / TD || (S_out - 1 - pad) / TH != (S_out - 1) / TH || (S_out - 1 - pad) / TW != (S_out - 1) / TW)
        return VXB_ESIZE;
    HaloArgs f;
    f.fold_pad = pad; f.fold_S = S; f.fold_dst[0] = ws /* (never written in this mode) */; f.fold_dst[1] = ws;
    f.fold_y[0] = y; f.fold_y[1] = x; f.fold_acc[0] = 0; f.fold_acc[1] = 2;          // "wgin": see HaloArgs
    f.amax_part = nullptr; f.colsum_part = nullptr;
    int rc = hb_impl(2, dy, nullptr, C0, 0, B, S, S_out, -2 * pad, 0, wfrag_f16, 64, nullptr, nullptr, 0, slope, 0, 0, 0, stream, &f,
                     wfrag_f16, nullptr, 0, 0, 0, scale);
    if (rc) return rc;
    return vxb_wgin_finish_launch(ws, (int)vxb_conv3_dgrad_fold_blocks(B, S, 64), scale, dW, db, (hipStream_t)stream);
}
