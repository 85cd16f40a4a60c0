// Tile-routed point chain of the voxelizer (gfx950): everything irregular between the point clouds and the dense grid,
// without a dense table, without float atomics and without a sort.
//
// Replaces the scatter half of VoxelGrid.coords_to_bounding_voxel_grid (reference peract/voxel/voxel_grid.py:106-146,
// :159-182): sums per cell are taken in ASCENDING POINT ID, the order scatter_add_ uses on the CPU, so all channels are
// bit-identical to the reference and run-to-run deterministic.
//
// The V^3 cells of a sample are cut into tiles of 8 x 8 x 16 cells (1024 cells: a tile's per-cell counters fit in a few KB
// of LDS).  Four kernels, the first three on a high-priority side stream next to the 640 MB empty-grid store stream:
//
//   route   one workgroup per 2048-point chunk (ids ascending inside a wave): cell index with the reference's fp32
//           arithmetic (optionally after the SE(3) transform of the augmentation), border points dropped, then a STABLE
//           partition of the chunk by tile -- ranks from wave ballots ("which lower lanes hold the same tile"), wave-
//           private LDS histograms, no atomics -- writes 32-byte point records (xyz, features, cell | id key) sorted by
//           tile and a tile-major [tile][chunk] offset table (uint16).
//   light   ONE WAVE per tile, no barriers: reads the tile's two offset rows, leaves if the tile is empty, hands tiles
//           with more than 256 points to the heavy list; otherwise holds the tile's records in registers (chunk order = id
//           order), ranks them per cell with the same ballots (stable counting sort), parks them in LDS in (cell, id) order
//           and lets each lane add the records of its cells front to back -- ascending id by construction.
//   heavy   one 256-thread workgroup per listed tile: the same algorithm with four wave-private counter rows and the
//           sorted records in an L2-resident scratch.
//   patch   after the fill has finished: a flat pass over the compact (means, cell address) records of every sample
//           overwrites the occupied cells' 40 bytes of the dense grid.
//
// All intermediate buffers are a few bytes per point (L2 / MALL resident); a per-call memset clears B + 1 counters.
#include "voxelize.h"
#include <limits.h>

using namespace vox;

#ifndef VT_ABL
#define VT_ABL 0          // timing-experiment builds of the route kernel (-DVT_ABL=bits through VXB_EXTRA_FLAGS; results WRONG when != 0)
#endif

namespace {

constexpr int TX = 8, TY = 8, TZ = 16;          // cells per tile and axis
constexpr int CELLS = TX * TY * TZ;             // 1024: cell-in-tile = (x & 7) << 7 | (y & 7) << 4 | (z & 15)
constexpr int CHUNK = 1024;                     // points per route workgroup (256 threads x 4)
constexpr int ID_BITS = 20;                     // key = cell-in-tile << 20 | point id
constexpr int MAX_TILES = 8191;                 // 13-bit tile field in the route kernel's per-point word
constexpr int MAX_NC = 512;                     // chunks per sample

constexpr int LIGHT = 256;                      // tiles with at most this many points are reduced by a single wave

struct TileWs {
    int* ctr;                // [64 + 32 B]  ctr[0] = light-list length, ctr[32] = heavy-list length, ctr[64 + 32 b] = occupied
                             // cells of sample b emitted so far (one 128-byte line per counter: same-line atomics serialise
                             // at ~12 ns each)
    const int* ctr_prev;     // the same counters of the previous call on this workspace (incremental mode), or null
    int* light;              // [B * NT]  b * NT + tile of every tile with 1 .. LIGHT points
    int* heavy;              // [B * NT]  ... with more than LIGHT points
    unsigned short* off;     // [B][NT + 1][NC]  start of tile t inside chunk c's sorted run (row NT = points kept of chunk c)
    float4* recs;            // [B][NC * CHUNK][2]  xyz, features (<= 4), key bits; sorted by tile inside a chunk
    float4* sorted;          // same shape: heavy tiles' records in (cell, id) order
    float4* res;             // same shape: per occupied cell: means (<= 7), cell address bits; dense from slot 0 per sample
    const float4* res_prev;  // the cell list the previous call on this workspace left (incremental mode), or null
    int NT, Tx, Ty, Tz, NC, tile_bits;
    long long NP;            // NC * CHUNK slots per sample
};

// lanes (among the valid ones) whose key equals this lane's key: `nbits` ballots
__device__ __forceinline__ unsigned long long match_bits(unsigned key, int nbits, bool valid) {
    unsigned long long m = __ballot(valid);
    for (int i = 0; i < nbits; ++i) {
        const bool bit = (key >> i) & 1u;
        const unsigned long long bal = __ballot(bit && valid);
        m &= bit ? bal : ~bal;
    }
    return m;
}

__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
}

// exclusive prefix sum over the 256 threads of a workgroup (two barriers); *total = sum over the workgroup
__device__ __forceinline__ int block_excl_scan(int v, int* s_red, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) s_red[wv] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = s_red[k];
        if (k < wv) base += x;
        tot += x;
    }
    *total = tot;
    return base + incl - v;
}

// ------------------------------------------------------------------------------------------------------------ route
// (s_hist: [4][NT] unsigned short of dynamic LDS: per wave, points of this chunk per tile)
template <int F>
__device__ __forceinline__ void route_chunk(const Src& src, const Geom& g, const float* __restrict__ bounds, const TileWs& w,
                                            const int chunk, const int b, unsigned short* s_hist) {
    __shared__ int s_red[4];
    const int NT = w.NT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 2 * NT; i += 256) reinterpret_cast<unsigned*>(s_hist)[i] = 0u;
    __syncthreads();
    const float* bd = bounds + (g.bounds_rows > 1 ? b * 6 : 0);
    float bmn[3], bmx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { bmn[a] = bd[a]; bmx[a] = bd[3 + a]; }

    constexpr int PT = CHUNK / 256;                     // points per thread
    unsigned tilecell[PT];                              // tile | cell << 13
    float pv[PT][3 + F];
    unsigned validmask = 0;
    const int n0 = chunk * CHUNK + wv * (CHUNK / 4) + lane;
    // A chunk that lies inside ONE source (camera) -- always, when the points per source are a multiple of the chunk -- takes that
    // source's base pointers once, as scalars.  Indexing the by-value pointer arrays of `src` with a per-lane source index costs a
    // dependent (vector) load of the pointer in front of every coordinate load: 11 of the kernel's 29 us (profiles/r05_voxel_route_ablation.log).
    const int cbeg = chunk * CHUNK;
    const int s_u = __builtin_amdgcn_readfirstlane(cbeg / g.pps);
    const bool one_src = !g.proj && (cbeg - s_u * g.pps) + CHUNK <= g.pps;          // (uniform)
    const float* __restrict__ cbase = one_src ? src.c[s_u] + (long long)b * g.cb - (long long)s_u * g.pps * g.cp : nullptr;
    const float* __restrict__ fbase = (one_src && F > 0) ? src.f[s_u] + (long long)b * g.fb - (long long)s_u * g.pps * g.fp : nullptr;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int n = n0 + j * 64;
        tilecell[j] = 0;
        if (n < g.N) {
            float p[3];
            if (VT_ABL & 16) { p[0] = bmn[0] + 1e-3f * (n & 255); p[1] = bmn[1] + 1e-3f * ((n >> 8) & 255); p[2] = bmn[2] + 1e-3f * (n >> 16); }
            else if (one_src) {
                const float* cp = cbase + (long long)n * g.cp;
#pragma unroll
                for (int a = 0; a < 3; ++a) p[a] = cp[a * g.cc];
                if (g.xf) xform_point(g.xf + b * 15, p);
            } else load_coords(src, g, b, n, p);
            int ix[3];
            bool inside = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int iv = axis_index(p[a], bmn[a], bmx[a], g.V);
                inside = inside && (iv >= 1) && (iv <= g.V);
                ix[a] = iv - 1;
            }
            if (inside) {
                validmask |= 1u << j;
                const unsigned tile = (unsigned)(((ix[0] >> 3) * w.Ty + (ix[1] >> 3)) * w.Tz + (ix[2] >> 4));
                const unsigned cell = (unsigned)(((ix[0] & 7) << 7) | ((ix[1] & 7) << 4) | (ix[2] & 15));
                tilecell[j] = tile | (cell << 13);
#pragma unroll
                for (int a = 0; a < 3; ++a) pv[j][a] = p[a];
                if (F > 0 && !(VT_ABL & 8)) {
                    const float* fp = one_src ? fbase + (long long)n * g.fp : point_ptr(src.f, n, g.pps, b, g.fb, g.fp);
#pragma unroll
                    for (int c = 0; c < F; ++c) pv[j][3 + c] = fp[c * g.fc];
                }
            }
        }
    }
    // stable position of every point inside its wave's run of the tile (wave w holds ids [w*512, w*512+512) of the chunk,
    // batch j the next 64 of them, a lane's rank = equal-tile lanes below it)
    unsigned short* myh = s_hist + wv * NT;
    unsigned pos_in_wave[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const bool valid = (validmask >> j) & 1u;
        const unsigned tile = tilecell[j] & 0x1FFFu;
        const unsigned long long m = (VT_ABL & 4) ? (1ull << lane) : match_bits(tile, w.tile_bits, valid);
        const int rank = lanes_below(m), cnt = __popcll(m);
        const unsigned before = valid ? myh[tile] : 0u;
        if (valid && rank == 0) myh[tile] = (unsigned short)(before + cnt);
        pos_in_wave[j] = before + rank;
    }
    __syncthreads();
    // offsets: tiles in order, waves in order inside a tile
    const int per = (NT + 255) / 256;
    const int t0 = min(NT, tid * per), t1 = min(NT, t0 + per);
    int sum = 0;
    for (int t = t0; t < t1; ++t) sum += s_hist[t] + s_hist[NT + t] + s_hist[2 * NT + t] + s_hist[3 * NT + t];
    int total;
    int run = block_excl_scan(sum, s_red, &total);
    unsigned short* offp = w.off + (size_t)b * (NT + 1) * w.NC + chunk;       // [tile][chunk]: a tile's row is contiguous
    for (int t = t0; t < t1; ++t) {
        const int h0 = s_hist[t], h1 = s_hist[NT + t], h2 = s_hist[2 * NT + t], h3 = s_hist[3 * NT + t];
        if (!(VT_ABL & 2)) offp[(size_t)t * w.NC] = (unsigned short)run;
        s_hist[t] = (unsigned short)run;
        s_hist[NT + t] = (unsigned short)(run + h0);
        s_hist[2 * NT + t] = (unsigned short)(run + h0 + h1);
        s_hist[3 * NT + t] = (unsigned short)(run + h0 + h1 + h2);
        run += h0 + h1 + h2 + h3;
    }
    if (tid == 255) offp[(size_t)NT * w.NC] = (unsigned short)total;
    __syncthreads();
    const size_t slot0 = (size_t)b * w.NP + (size_t)chunk * CHUNK;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        if ((validmask >> j) & 1u) {
            const unsigned tile = tilecell[j] & 0x1FFFu, cell = tilecell[j] >> 13;
            const size_t dst = slot0 + myh[tile] + pos_in_wave[j];
            const unsigned key = (cell << ID_BITS) | (unsigned)(n0 + j * 64);
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3 + F; ++c) v[c] = pv[j][c];
            v[7] = __uint_as_float(key);
            if (!(VT_ABL & 1)) {
                w.recs[dst * 2] = make_float4(v[0], v[1], v[2], v[3]);
                w.recs[dst * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
}

template <int F>
__global__ void __launch_bounds__(256) vt_route_kernel(Src src, Geom g, const float* __restrict__ bounds, TileWs w, int clear_ctr) {
    extern __shared__ unsigned short s_dyn[];
    if (clear_ctr && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < 64 + 32 * g.B; i += 256) w.ctr[i] = 0;
    route_chunk<F>(src, g, bounds, w, blockIdx.x, blockIdx.y, s_dyn);
}

// one compact record per occupied cell: means of the 3 + F channels, address of the cell inside the sample; with `out` the
// cell's 3 + F + 4 floats of the dense grid are written here as well (voxel_grid.py:184-198 for an occupied cell) and no
// patch pass follows -- the grid already holds the empty pattern everywhere else (fill / unpatch ran before)
template <int F>
__device__ __forceinline__ void emit_cell(float4* __restrict__ res, size_t slot, const float (&acc)[7], int cnt, int gc,
                                          float* __restrict__ out_b, int V) {
    constexpr int C = 3 + F + 4;
    const float Lf = (float)cnt;                                    // clamp_(1) is a no-op for occupied cells (:121)
    float mean[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3 + F; ++c) mean[c] = __fdiv_rn(acc[c], Lf);
    res[slot * 2] = make_float4(mean[0], mean[1], mean[2], mean[3]);
    res[slot * 2 + 1] = make_float4(mean[4], mean[5], mean[6], __int_as_float(gc));
    if (out_b) {
        const float Vf = (float)V;                                  // self._voxel_d (:197)
        const int x = gc / (V * V), y = (gc / V) % V, z = gc % V;
        float v[C];
#pragma unroll
        for (int c = 0; c < 3 + F; ++c) v[c] = mean[c];
        v[3 + F + 0] = __fdiv_rn((float)x, Vf);
        v[3 + F + 1] = __fdiv_rn((float)y, Vf);
        v[3 + F + 2] = __fdiv_rn((float)z, Vf);
        v[3 + F + 3] = 1.0f;                                        // (count/count > 0).float()  (:192)
        float* o = out_b + (size_t)gc * C;
        if ((C & 1) == 0 && ((reinterpret_cast<uintptr_t>(out_b) & 7) == 0)) {
#pragma unroll
            for (int c = 0; c < C; c += 2) *reinterpret_cast<float2*>(o + c) = make_float2(v[c], v[c + 1]);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) o[c] = v[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ light
// One wave per (sample, tile); wave-synchronous (LDS traffic of one wave is processed in program order, the fences only
// stop the compiler from moving accesses across them).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one thread per (sample, tile): points in the tile = sum over the chunks of the segment lengths (two contiguous rows of the
// tile-major offset table) -> compact lists of the light and of the heavy tiles (empty tiles are never visited again)
__global__ void __launch_bounds__(256) vt_classify_kernel(Geom g, TileWs w) {
    __shared__ int s_red[4];
    __shared__ int s_base[2];
    const int NT = w.NT, NC = w.NC;
    const int bt = blockIdx.x * 256 + threadIdx.x;
    int n = 0;
    if (bt < g.B * NT) {
        const int b = bt / NT, tile = bt - b * NT;
        const unsigned short* row = w.off + ((size_t)b * (NT + 1) + tile) * NC;
        for (int c = 0; c < NC; ++c) n += (int)row[NC + c] - (int)row[c];
    }
    const bool is_heavy = n > LIGHT || (n > 0 && NC > 64);
    const bool is_light = n > 0 && !is_heavy;
    int tot;
    const int ex = block_excl_scan((is_light ? 1 : 0) | (is_heavy ? 1 << 16 : 0), s_red, &tot);
    if (threadIdx.x == 0) {
        s_base[0] = (tot & 0xFFFF) ? atomicAdd(&w.ctr[0], tot & 0xFFFF) : 0;
        s_base[1] = (tot >> 16) ? atomicAdd(&w.ctr[32], tot >> 16) : 0;
    }
    __syncthreads();
    if (is_light) w.light[s_base[0] + (ex & 0xFFFF)] = bt;
    if (is_heavy) w.heavy[s_base[1] + (ex >> 16)] = bt;
}

// slot (inside the sample's record array) of stream position s of a tile: chunk by binary search over the prefix of the
// segment lengths, then the offset inside that chunk's segment
__device__ __forceinline__ size_t locate(const int* s_segp, const int* s_segs, int NC, int s) {
    int lo = 0, hi = NC;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_segp[mid] <= s) lo = mid; else hi = mid;
    }
    return (size_t)lo * CHUNK + s_segs[lo] + (s - s_segp[lo]);
}

// ------------------------------------------------------------------------------------------------------------ heavy
constexpr int HB = 4;                                   // 64-record batches a wave keeps in flight in the heavy kernel
// (keeping the sorted records of tiles up to 3584 points in 112 KB of LDS, with 8 batches in flight, was measured: 64 us
// instead of 53 for the kernel -- the reduction is not what it waits for)

// ------------------------------------------------------------------------------------------------------------ merged tile kernel
// Round 5.  One launch instead of heavy -> light: workgroups [0, heavy_blocks) take the heavy tiles of the classify list (they are
// dispatched first, and their ~60 us run under everything else), the other workgroups are four independent waves that walk the light
// list, one tile each.  Both kinds are latency-bound chains of dependent round trips, so -- unlike every pairing with the store-bound
// fill (below) -- they overlap: 110 -> 55 us for the pair, the incremental call 162 -> 120 us.
//
// Measured and NOT kept (profiles/r05_v1_voxel_bench.log, r05_v2_*): a fresh grid without the fill kernel, every tile (or every column of
// tiles, as aligned 32 KB streams) writing its own block of the grid before patching its occupied cells -- 262 / 287 us against
// 225 us for fill -> route -> classify -> this kernel.  The store phase of a workgroup is bandwidth-bound and its chain phase
// latency-bound; workgroups that start together stay in step (all storing, then all waiting), so the launch takes the SUM of the two
// phases, and a loaded memory system stretches every dependent round trip of the chains by what the overlap would have hidden.
struct LightLds {
    unsigned short cnt[CELLS];
    int segs[64], segp[66];
    float4 rec[LIGHT * 2];
};
struct HeavyLds {
    unsigned hist[4 * CELLS];
    unsigned cell[CELLS + 4];
    int segs[MAX_NC];
    int segp[MAX_NC + 4];
    int red[4];
    int rbase, pad[3];
};
static_assert(sizeof(LightLds) % 16 == 0 && sizeof(HeavyLds) % 16 == 0, "16-byte LDS carve");
constexpr size_t TILES_LDS = 4 * sizeof(LightLds) > sizeof(HeavyLds) ? 4 * sizeof(LightLds) : sizeof(HeavyLds);

// one wave, one listed light tile (the algorithm of vt_light_kernel): its occupied cells into the compact list and, with `out`, the grid
template <int F>
__device__ __forceinline__ void wave_tile(const Geom& g, const TileWs& w, float* __restrict__ out, LightLds& L, int bt, int lane) {
    const int NT = w.NT, NC = w.NC;
    const int b = bt / NT, tile = bt - b * NT;
    const int tz = tile % w.Tz, ty = (tile / w.Tz) % w.Ty, tx = tile / (w.Tz * w.Ty);
    float* out_b = out ? out + (size_t)b * g.V * g.V * g.V * (3 + F + 4) : nullptr;
    const unsigned short* row = w.off + ((size_t)b * (NT + 1) + tile) * NC;
    int s = 0, len = 0;
    if (lane < NC) {
        s = row[lane];
        len = (int)row[NC + lane] - s;                      // next tile's row: where this tile's segment ends
    }
    int incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int n = __shfl(incl, 63, 64);
    wave_sync();                                            // (LDS of the previous tile of this wave is no longer read)
    L.segs[lane] = s;
    L.segp[lane] = incl - len;
    if (lane == 63) L.segp[64] = n;
    for (int i = lane; i < CELLS / 2; i += 64) reinterpret_cast<unsigned*>(L.cnt)[i] = 0u;
    wave_sync();
    // the tile's records, stream position j = lane + 64 k (chunk order = ascending point id)
    constexpr int K = LIGHT / 64;
    float4 ra[K], rb[K];
    unsigned cell[K], pos[K];
    const size_t rec0 = (size_t)b * w.NP;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int j = lane + 64 * k;
        cell[k] = 0;
        ra[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        rb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < n) {
            int lo = 0;                                     // largest chunk whose prefix is <= j (prefixes beyond NC equal n)
#pragma unroll
            for (int st = 32; st > 0; st >>= 1)
                if (L.segp[lo + st] <= j) lo += st;
            const size_t idx = rec0 + (size_t)lo * CHUNK + L.segs[lo] + (j - L.segp[lo]);
            ra[k] = w.recs[idx * 2];
            rb[k] = w.recs[idx * 2 + 1];
            cell[k] = __float_as_uint(rb[k].w) >> ID_BITS;
        }
    }
    // stable rank inside the cell: earlier batches first, lower lanes first
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool valid = lane + 64 * k < n;
        const unsigned long long m = match_bits(cell[k], 10, valid);
        const int rank = lanes_below(m), cnt = __popcll(m);
        const unsigned before = valid ? L.cnt[cell[k]] : 0u;
        if (valid && rank == 0) L.cnt[cell[k]] = (unsigned short)(before + cnt);
        pos[k] = before + rank;
        wave_sync();
    }
    // cell starts: lane L owns the z-column of cells 16 L .. 16 L + 15
    int tot = 0, occ = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = L.cnt[16 * lane + i];
        tot += c;
        occ += c > 0 ? 1 : 0;
    }
    int packed = (occ << 16) | tot, pincl = packed;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(pincl, o, 64);
        if (lane >= o) pincl += t;
    }
    const int nocc = __shfl(pincl, 63, 64) >> 16;
    const int lane_start = (pincl - packed) & 0xFFFF;
    const int occ_before = (pincl - packed) >> 16;
    wave_sync();
    {
        int run = lane_start;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = L.cnt[16 * lane + i];
            L.cnt[16 * lane + i] = (unsigned short)run;
            run += c;
        }
    }
    int rbase = 0;
    if (lane == 0) rbase = atomicAdd(&w.ctr[64 + 32 * b], nocc);
    wave_sync();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (lane + 64 * k < n) {
            const int p = L.cnt[cell[k]] + pos[k];
            L.rec[2 * p] = ra[k];
            L.rec[2 * p + 1] = rb[k];
        }
    }
    wave_sync();
    rbase = __shfl(rbase, 0, 64);
    float4* res = w.res + (size_t)b * w.NP * 2;
    int slot = rbase + occ_before;
    const int lane_end = lane_start + tot;
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const int st = L.cnt[16 * lane + i];
        const int cnt = (i < 15 ? (int)L.cnt[16 * lane + i + 1] : lane_end) - st;
        if (cnt > 0) {
            float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                // zeros_like(self._flat_output) (:145)
            for (int k = 0; k < cnt; ++k) {
                const float4 a = L.rec[2 * (st + k)], c4 = L.rec[2 * (st + k) + 1];
                const float r[7] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z};
#pragma unroll
                for (int c = 0; c < 3 + F; ++c) acc[c] = __fadd_rn(acc[c], r[c]);
            }
            const int cellid = 16 * lane + i;
            const int X = tx * TX + (cellid >> 7), Y = ty * TY + ((cellid >> 4) & 7), Z = tz * TZ + (cellid & 15);
            emit_cell<F>(res, (size_t)slot, acc, cnt, (X * g.V + Y) * g.V + Z, out_b, g.V);
            ++slot;
        }
    }
}

// one workgroup, one heavy tile (the algorithm of vt_heavy_kernel)
template <int F>
__device__ __forceinline__ void group_tile(const Geom& g, const TileWs& w, float* __restrict__ out, HeavyLds& H, int bt, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    const int NT = w.NT, NC = w.NC;
    const int b = bt / NT, tile = bt - b * NT;
    const int tz = tile % w.Tz, ty = (tile / w.Tz) % w.Ty, tx = tile / (w.Tz * w.Ty);
    float* out_b = out ? out + (size_t)b * g.V * g.V * g.V * (3 + F + 4) : nullptr;
    const unsigned short* row = w.off + ((size_t)b * (NT + 1) + tile) * NC;
    int n = 0, base_part = 0;
    for (int c0 = 0; c0 < NC; c0 += 256) {
        const int c = c0 + tid;
        int s = 0, len = 0;
        if (c < NC) {
            s = row[c];
            len = (int)row[NC + c] - s;
            H.segs[c] = s;
        }
        int tot;
        const int ex = block_excl_scan(len, H.red, &tot);
        if (c < NC) H.segp[c] = n + ex;
        n += tot;
        base_part += s;
    }
    if (n == 0) { __syncthreads(); return; }            // (uniform; only reachable when every tile is visited: more than 64 chunks per sample)
    if (tid == 0) H.segp[NC] = n;
    int base;                                           // points of this sample in lower-numbered tiles
    block_excl_scan(base_part, H.red, &base);
    for (int i = tid; i < 4 * CELLS; i += 256) H.hist[i] = 0u;
    __syncthreads();
    const int q = (((n + 3) >> 2) + 63) & ~63;
    const int r0 = min(n, wv * q), r1 = min(n, r0 + q);
    unsigned* myh = H.hist + wv * CELLS;
    const size_t rec0 = (size_t)b * w.NP;
    for (int s0 = r0; s0 < r1; s0 += 64 * HB) {
        unsigned cell[HB];
        bool valid[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const int s = s0 + 64 * u + lane;
            valid[u] = s < r1;
            cell[u] = 0;
            if (valid[u]) cell[u] = __float_as_uint(w.recs[(rec0 + locate(H.segp, H.segs, NC, s)) * 2 + 1].w) >> ID_BITS;
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            if (s0 + 64 * u >= r1) break;                   // (wave-uniform)
            const unsigned long long m = match_bits(cell[u], 10, valid[u]);
            const int rank = lanes_below(m), cnt = __popcll(m);
            if (valid[u] && rank == 0) myh[cell[u]] += (unsigned)cnt;
        }
    }
    __syncthreads();
    {
        unsigned h[4][4];
        int cs[4], tot = 0, occ = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cellid = 4 * tid + i;
            cs[i] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { h[i][k] = H.hist[k * CELLS + cellid]; cs[i] += (int)h[i][k]; }
            tot += cs[i];
            occ += cs[i] > 0 ? 1 : 0;
        }
        int dummy;
        unsigned run = (unsigned)block_excl_scan((occ << 20) | tot, H.red, &dummy);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cellid = 4 * tid + i;
            H.cell[cellid] = run;
            const unsigned start = run & 0xFFFFFu;
            H.hist[cellid] = start;
            H.hist[CELLS + cellid] = start + h[i][0];
            H.hist[2 * CELLS + cellid] = start + h[i][0] + h[i][1];
            H.hist[3 * CELLS + cellid] = start + h[i][0] + h[i][1] + h[i][2];
            run += (cs[i] > 0 ? (1u << 20) : 0u) + (unsigned)cs[i];
        }
        if (tid == 255) {
            H.cell[CELLS] = run;
            H.rbase = atomicAdd(&w.ctr[64 + 32 * b], (int)(run >> 20));
        }
    }
    __syncthreads();
    float4* sorted = w.sorted + ((size_t)b * w.NP + base) * 2;
    for (int s0 = r0; s0 < r1; s0 += 64 * HB) {
        float4 ra[HB], rb[HB];
        bool valid[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const int s = s0 + 64 * u + lane;
            valid[u] = s < r1;
            ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            rb[u] = ra[u];
            if (valid[u]) {
                const size_t idx = rec0 + locate(H.segp, H.segs, NC, s);
                ra[u] = w.recs[idx * 2];
                rb[u] = w.recs[idx * 2 + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            if (s0 + 64 * u >= r1) break;                   // (wave-uniform)
            const unsigned cell = __float_as_uint(rb[u].w) >> ID_BITS;
            const unsigned long long m = match_bits(cell, 10, valid[u]);
            const int rank = lanes_below(m), cnt = __popcll(m);
            const unsigned before = valid[u] ? myh[cell] : 0u;
            if (valid[u] && rank == 0) myh[cell] = before + (unsigned)cnt;
            if (valid[u]) {
                const size_t pos = before + rank;
                sorted[pos * 2] = ra[u];
                sorted[pos * 2 + 1] = rb[u];
            }
        }
    }
    __syncthreads();
    float4* res = w.res + (size_t)b * w.NP * 2;
    const int rbase = H.rbase;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const int cellid = i * 256 + tid;
        const unsigned info = H.cell[cellid], next = H.cell[cellid + 1];
        const int start = (int)(info & 0xFFFFFu), cnt = (int)(next & 0xFFFFFu) - start;
        if (cnt > 0) {
            float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // zeros_like(self._flat_output) (:145)
            const float4* sp = sorted + (size_t)start * 2;
            int k = 0;
            for (; k + 8 <= cnt; k += 8) {                              // eight records in flight, added in order
                float4 a[8], c4[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { a[u] = sp[(k + u) * 2]; c4[u] = sp[(k + u) * 2 + 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float r[7] = {a[u].x, a[u].y, a[u].z, a[u].w, c4[u].x, c4[u].y, c4[u].z};
#pragma unroll
                    for (int c = 0; c < 3 + F; ++c) acc[c] = __fadd_rn(acc[c], r[c]);
                }
            }
            for (; k < cnt; ++k) {
                const float4 a = sp[k * 2], c4 = sp[k * 2 + 1];
                const float r[7] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z};
#pragma unroll
                for (int c = 0; c < 3 + F; ++c) acc[c] = __fadd_rn(acc[c], r[c]);
            }
            const int X = tx * TX + (cellid >> 7), Y = ty * TY + ((cellid >> 4) & 7), Z = tz * TZ + (cellid & 15);
            emit_cell<F>(res, (size_t)(rbase + (int)(info >> 20)), acc, cnt, (X * g.V + Y) * g.V + Z, out_b, g.V);
        }
    }
    __syncthreads();
}

template <int F>
__global__ void __launch_bounds__(256) vt_tiles_kernel(Geom g, TileWs w, float* __restrict__ out, int heavy_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[TILES_LDS];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < heavy_blocks) {
        HeavyLds& H = *reinterpret_cast<HeavyLds*>(smem);
        const int nheavy = w.ctr[32];
        for (int e = blockIdx.x; e < nheavy; e += heavy_blocks) group_tile<F>(g, w, out, H, w.heavy[e], tid);
        return;
    }
    LightLds& L = reinterpret_cast<LightLds*>(smem)[tid >> 6];
    const int wave = ((int)blockIdx.x - heavy_blocks) * 4 + (tid >> 6), nwaves = ((int)gridDim.x - heavy_blocks) * 4;
    const int nlight = w.ctr[0];
    for (int e = wave; e < nlight; e += nwaves) wave_tile<F>(g, w, out, L, w.light[e], tid & 63);
}

// the same two routines as separate launches: the chain of rounds 2-4 (vxb_voxelize_select_chain(3), A/B measurements) and, heavy
// routine only over EVERY tile, samples of more than 64 chunks
template <int F>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4))) vt_light_kernel(Geom g, TileWs w, float* __restrict__ out) {
    __shared__ LightLds L;
    const int nlight = w.ctr[0];
    for (int e = blockIdx.x; e < nlight; e += gridDim.x) wave_tile<F>(g, w, out, L, w.light[e], threadIdx.x);
}

template <int F>
__global__ void __launch_bounds__(256) vt_heavy_kernel(Geom g, TileWs w, int all_tiles, float* __restrict__ out) {
    __shared__ HeavyLds H;
    const int nwork = all_tiles ? g.B * w.NT : w.ctr[32];
    for (int e = blockIdx.x; e < nwork; e += gridDim.x) group_tile<F>(g, w, out, H, all_tiles ? e : w.heavy[e], threadIdx.x);
}

// ------------------------------------------------------------------------------------------------------------ unpatch
// Incremental mode: `out` still holds the grid of the previous call that used this workspace, whose compact cell list is
// still in `res` / `ctr`.  The empty-cell pattern does not depend on the input, so instead of re-writing all V^3 cells the
// previously occupied ones are reset to "empty" (zeros | idx/V | 0, voxel_grid.py:192-198 for count == 0) before the new
// ones are patched in: ~2 x 40 bytes per occupied cell of traffic instead of 40 bytes per cell of the grid.
template <int F>
__device__ __forceinline__ void unpatch_cells(const Geom& g, const TileWs& w, float* __restrict__ out, int bx, int b) {
    constexpr int C = 3 + F + 4;
    const int V = g.V;
    const int nocc = w.ctr_prev[64 + 32 * b];
    const int j = bx * 256 + threadIdx.x;
    if (j >= nocc) return;
    const size_t V3 = (size_t)V * V * V;
    const float Vf = (float)V;
    const int gc = __float_as_int(w.res_prev[((size_t)b * w.NP + j) * 2 + 1].w);
    const int x = gc / (V * V), y = (gc / V) % V, z = gc % V;
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = 0.0f;
    v[3 + F + 0] = __fdiv_rn((float)x, Vf);
    v[3 + F + 1] = __fdiv_rn((float)y, Vf);
    v[3 + F + 2] = __fdiv_rn((float)z, Vf);
    float* o = out + ((size_t)b * V3 + gc) * C;
    if ((C & 1) == 0 && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)) {
#pragma unroll
        for (int c = 0; c < C; c += 2) *reinterpret_cast<float2*>(o + c) = make_float2(v[c], v[c + 1]);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = v[c];
    }
}

template <int F>
__global__ void __launch_bounds__(256) vt_unpatch_kernel(Geom g, TileWs w, float* __restrict__ out) {
    if (blockIdx.x == 0 && blockIdx.y == 0)              // this call's counters (the other set than ctr_prev): no memset node
        for (int i = threadIdx.x; i < 64 + 32 * g.B; i += 256) w.ctr[i] = 0;
    unpatch_cells<F>(g, w, out, blockIdx.x, blockIdx.y);
}

// Round 6: the reset of the previously occupied cells IN the route launch.  The two are independent (unpatch writes the grid, route the
// record arrays; only the tile kernel needs both) and each is a chain of dependent round trips that leaves the memory system idle: the
// route workgroups (blockIdx.x < NC) are dispatched first and all of them are resident at once (NC x B = 1024 of 2048 slots at
// configs[1]), the unpatch workgroups take the free slots beside them -- one launch and one kernel boundary fewer, the 17 us of the
// unpatch kernel off the critical path.  (Two streams for the same pair ended later than back to back in round 2: an event wait and a
// join cost more than the pair overlaps.)
template <int F>
__global__ void __launch_bounds__(256) vt_unpatch_route_kernel(Src src, Geom g, const float* __restrict__ bounds, TileWs w, float* __restrict__ out) {
    extern __shared__ unsigned short s_dyn[];
    if ((int)blockIdx.x >= w.NC) {
        if ((int)blockIdx.x == w.NC && blockIdx.y == 0)
            for (int i = threadIdx.x; i < 64 + 32 * g.B; i += 256) w.ctr[i] = 0;
        unpatch_cells<F>(g, w, out, (int)blockIdx.x - w.NC, blockIdx.y);
        return;
    }
    route_chunk<F>(src, g, bounds, w, blockIdx.x, blockIdx.y, s_dyn);
}

// ------------------------------------------------------------------------------------------------------------ patch
template <int F>
__global__ void __launch_bounds__(256) vt_patch_kernel(Geom g, TileWs w, float* __restrict__ out) {
    constexpr int C = 3 + F + 4;
    const int b = blockIdx.y, V = g.V;
    const int nocc = w.ctr[64 + 32 * b];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nocc) return;
    const size_t V3 = (size_t)V * V * V;
    const float Vf = (float)V;                           // self._voxel_d (:197)
    const float4* res = w.res + (size_t)b * w.NP * 2;
    const float4 ra = res[(size_t)j * 2], rb = res[(size_t)j * 2 + 1];
    const int gc = __float_as_int(rb.w);
    const int x = gc / (V * V), y = (gc / V) % V, z = gc % V;
    const float m[7] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z};
    float v[C];
#pragma unroll
    for (int c = 0; c < 3 + F; ++c) v[c] = m[c];
    v[3 + F + 0] = __fdiv_rn((float)x, Vf);
    v[3 + F + 1] = __fdiv_rn((float)y, Vf);
    v[3 + F + 2] = __fdiv_rn((float)z, Vf);
    v[3 + F + 3] = 1.0f;                                 // (count/count > 0).float()  (:192)
    float* o = out + ((size_t)b * V3 + gc) * C;
    if ((C & 1) == 0 && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)) {
#pragma unroll
        for (int c = 0; c < C; c += 2) *reinterpret_cast<float2*>(o + c) = make_float2(v[c], v[c + 1]);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = v[c];
    }
}

struct Layout {
    size_t ctr[2], light, heavy, off, recs, sorted, res[2], total;
    int NT, Tx, Ty, Tz, NC;
};
constexpr size_t CTR_INTS(long long B) { return 64 + 32 * (size_t)B; }

Layout vt_layout(long long B, long long N, int V) {
    Layout L;
    L.Tx = (V + TX - 1) / TX; L.Ty = (V + TY - 1) / TY; L.Tz = (V + TZ - 1) / TZ;
    L.NT = L.Tx * L.Ty * L.Tz;
    L.NC = (int)((N + CHUNK - 1) / CHUNK);
    size_t p = 0;
    auto take = [&p](size_t bytes) { const size_t at = p; p += (bytes + 255) & ~(size_t)255; return at; };
    L.ctr[0] = take(CTR_INTS(B) * sizeof(int));
    L.ctr[1] = take(CTR_INTS(B) * sizeof(int));
    L.light = take((size_t)B * L.NT * sizeof(int));
    L.heavy = take((size_t)B * L.NT * sizeof(int));
    L.off = take((size_t)B * (L.NT + 1) * L.NC * sizeof(unsigned short));
    const size_t slots = (size_t)B * L.NC * CHUNK;
    L.recs = take(slots * 32);
    L.sorted = take(slots * 32);
    L.res[0] = take(slots * 32);            // the compact cell lists are double-buffered: in incremental mode the previous
    L.res[1] = take(slots * 32);            // call's list is still being read (unpatch) while the new one is written
    L.total = p;
    return L;
}

template <int F>
int vt_launch(const Src& src, const Geom& g, const float* bounds, float* out, const TileWs& w, hipStream_t st, hipStream_t side,
              hipEvent_t ev_fork, hipEvent_t ev_mid, hipEvent_t ev_join, int order_in, int C) {
    int order = order_in;
    // order 2: fresh buffer.   fill, route, classify, heavy, light           -- all in order on `st`: no side stream, no
    // order 4: incremental.    unpatch (+ counter reset), route, classify, heavy, light     events, capturable in a hipGraph
    // orders 0 / 3 (A/B measurements): the fill on the side stream, from the start / after the classify kernel.
    // Every attempt to run two of these kernels side by side (fill || chain, light || heavy, unpatch || route) ended LATER
    // than running them back to back on MI355X: they are all bound by memory latency, and sharing the memory system
    // stretches each by about what the overlap would have hidden (profiles/r02_voxel_*.txt).
    const long long cap = (long long)g.N < (long long)g.V * g.V * g.V ? g.N : (long long)g.V * g.V * g.V;   // occupied cells per sample
    const bool forked = order == 0 || order == 3;       // only the A/B orders use the side stream
    if (forked && (hipEventRecord(ev_fork, st) != hipSuccess || hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess)) return VXB_ELAUNCH;
    // orders 2 and 4 (the shipped ones): the reduce kernels write the occupied cells straight into the grid, so the dense part
    // (fill, or the reset of the previously occupied cells) runs FIRST and there is no patch kernel
    // orders 6 / 7 (round 5, the shipped ones when a sample has at most 64 chunks): fresh buffer / incremental with heavy and light
    // tiles in ONE launch (vt_tiles_kernel): fill | unpatch, route, classify, tiles
    // order 9 (round 6, the shipped incremental one): order 7 with the unpatch workgroups inside the route launch
    if (order == 6 && w.NC > 64) order = 2;
    if ((order == 7 || order == 9) && w.NC > 64) order = 4;
    const bool merged_unpatch = order == 9;
    if (merged_unpatch) order = 7;
    const bool direct = order == 2 || order == 4 || order == 6 || order == 7;
    float* dout = direct ? out : nullptr;
    if ((order == 4 || order == 7) && !merged_unpatch) hipLaunchKernelGGL(vt_unpatch_kernel<F>, dim3((unsigned)((cap + 255) / 256), g.B), dim3(256), 0, st, g, w, out);
    if (order == 0) vox_launch_fill(out, g.B, g.V, C, side);
    if (order == 2 || order == 6) vox_launch_fill(out, g.B, g.V, C, st);
    const size_t lds = (size_t)4 * w.NT * sizeof(unsigned short);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)vt_route_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VXB_ELAUNCH;
    }
    // (this call's counters are cleared by the route kernel's first workgroup unless the unpatch kernel already did: no memset node)
    if (merged_unpatch) {
        if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)vt_unpatch_route_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VXB_ELAUNCH;
        hipLaunchKernelGGL(vt_unpatch_route_kernel<F>, dim3((unsigned)(w.NC + (cap + 255) / 256), g.B), dim3(256), lds, st, src, g, bounds, w, out);
    } else
        hipLaunchKernelGGL(vt_route_kernel<F>, dim3(w.NC, g.B), dim3(256), lds, st, src, g, bounds, w, (order != 4 && order != 7) ? 1 : 0);
    const long long tiles = (long long)g.B * w.NT;
    const size_t heavy_lds = 0;
    if (order == 6 || order == 7) {
        hipLaunchKernelGGL(vt_classify_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, st, g, w);
        const int hb = (int)(tiles < 512 ? tiles : 512);
        hipLaunchKernelGGL((vt_tiles_kernel<F>), dim3((unsigned)(hb + (tiles < 4096 ? (tiles + 3) / 4 : 1024))), dim3(256), 0, st, g, w, out, hb);
    } else if (w.NC <= 64) {
        hipLaunchKernelGGL(vt_classify_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, st, g, w);
        // (light and heavy tiles on two streams were measured: the heavy kernel stretches from 53 to 75 us next to the
        // light one and the pair ends later than back to back)
        if (order == 3) vox_launch_fill(out, g.B, g.V, C, side);
        hipLaunchKernelGGL(vt_heavy_kernel<F>, dim3(512), dim3(256), heavy_lds, st, g, w, 0, dout);
        hipLaunchKernelGGL(vt_light_kernel<F>, dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(64), 0, st, g, w, dout);
    } else {                                                    // more than 64 chunks per sample: every tile by the workgroup kernel
        if (order == 3) vox_launch_fill(out, g.B, g.V, C, side);
        hipLaunchKernelGGL(vt_heavy_kernel<F>, dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(256), heavy_lds, st, g, w, 1, dout);
    }
    if (forked && (hipEventRecord(ev_join, side) != hipSuccess || hipStreamWaitEvent(st, ev_join, 0) != hipSuccess)) return VXB_ELAUNCH;
    if (!direct) hipLaunchKernelGGL(vt_patch_kernel<F>, dim3((unsigned)((cap + 255) / 256), g.B), dim3(256), 0, st, g, w, out);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

}  // namespace

bool vox_tiles_supported(long long B, long long N, int V, int F) {
    if (F < 0 || F > 4 || V < 1 || N < 1 || N >= (1ll << ID_BITS)) return false;
    const Layout L = vt_layout(B, N, V);
    if (L.NT > MAX_TILES || L.NC > MAX_NC) return false;
    if ((long long)B * L.NT >= INT_MAX || (long long)B * L.NC * CHUNK >= (1ll << 30)) return false;
    return true;
}

size_t vox_tiles_ws_bytes(long long B, long long N, int V) {
    if (!vox_tiles_supported(B, N, V, 0)) return 0;
    return vt_layout(B, N, V).total;
}

int vox_tiles_launch(const Src& src, const Geom& g, const float* bounds, float* out, void* ws, hipStream_t st, hipStream_t side,
                     hipEvent_t ev_fork, hipEvent_t ev_mid, hipEvent_t ev_join, int order, int list_in, int list_out) {
    const int C = 3 + g.F + 4;
    const Layout L = vt_layout(g.B, g.N, g.V);
    char* p = (char*)ws;
    TileWs w;
    w.ctr = (int*)(p + L.ctr[list_out]);
    w.ctr_prev = list_in >= 0 ? (const int*)(p + L.ctr[list_in]) : nullptr;
    w.light = (int*)(p + L.light);
    w.heavy = (int*)(p + L.heavy);
    w.off = (unsigned short*)(p + L.off);
    w.recs = (float4*)(p + L.recs);
    w.sorted = (float4*)(p + L.sorted);
    w.res = (float4*)(p + L.res[list_out]);
    w.res_prev = list_in >= 0 ? (const float4*)(p + L.res[list_in]) : nullptr;
    w.NT = L.NT; w.Tx = L.Tx; w.Ty = L.Ty; w.Tz = L.Tz; w.NC = L.NC;
    w.NP = (long long)L.NC * CHUNK;
    int bits = 1;
    while ((1 << bits) < L.NT) ++bits;
    w.tile_bits = bits;
    if ((order == 4 || order == 7 || order == 9) && list_in < 0) return VXB_EARG;
    switch (g.F) {
        case 0: return vt_launch<0>(src, g, bounds, out, w, st, side, ev_fork, ev_mid, ev_join, order, C);
        case 1: return vt_launch<1>(src, g, bounds, out, w, st, side, ev_fork, ev_mid, ev_join, order, C);
        case 2: return vt_launch<2>(src, g, bounds, out, w, st, side, ev_fork, ev_mid, ev_join, order, C);
        case 3: return vt_launch<3>(src, g, bounds, out, w, st, side, ev_fork, ev_mid, ev_join, order, C);
        case 4: return vt_launch<4>(src, g, bounds, out, w, st, side, ev_fork, ev_mid, ev_join, order, C);
        default: return VXB_EARG;
    }
}
