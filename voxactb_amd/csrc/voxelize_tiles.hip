// Tile-routed point chain of the voxelizer (gfx950): everything irregular between the point clouds and the dense grid,
// without a dense table, without float atomics and without a sort.
//
// Replaces the scatter half of VoxelGrid.coords_to_bounding_voxel_grid (reference peract/voxel/voxel_grid.py:106-146,
// :159-182): sums per cell are taken in ASCENDING POINT ID, the order scatter_add_ uses on the CPU, so all channels are
// bit-identical to the reference and run-to-run deterministic.
//
// The V^3 cells of a sample are cut into tiles of 8 x 8 x 16 cells (1024 cells: a tile's per-cell counters fit in 16 KB
// of LDS).  Three kernels, the first two on a side stream next to the 640 MB empty-grid store stream:
//
//   route   one workgroup per 2048-point chunk (ids ascending inside a wave): cell index with the reference's fp32
//           arithmetic (optionally after the SE(3) transform of the augmentation), border points dropped, then a STABLE
//           partition of the chunk by tile -- ranks from wave ballots ("which lower lanes hold the same tile"), wave-
//           private LDS histograms, no atomics on the data path -- writes (cell | id) keys and 32-byte point records
//           sorted by tile, a [chunk][tile] offset table (uint16) and a worklist of non-empty (sample, tile) pairs.
//   tile    one workgroup per non-empty tile: walks the tile's segments of all chunks in chunk order (= id order),
//           stable counting sort by cell (same ballot ranks, 4 wave-private counter rows), records moved to their cell
//           segment, then one thread per cell adds its records front to back -- ascending id by construction -- and emits
//           a compact (means, cell address) record per occupied cell.
//   patch   after the fill has finished: occupied cells overwrite their 40 bytes of the dense grid.
//
// All intermediate buffers are a few bytes per point (L2 / MALL resident); nothing needs zero-initialising except a
// B * tiles counter array (one small memset per call).
#include "voxelize.h"
#include <limits.h>

using namespace vox;

namespace {

constexpr int TX = 8, TY = 8, TZ = 16;          // cells per tile and axis
constexpr int CELLS = TX * TY * TZ;             // 1024: cell-in-tile = (x & 7) << 7 | (y & 7) << 4 | (z & 15)
constexpr int CHUNK = 2048;                     // points per route workgroup (256 threads x 8)
constexpr int ID_BITS = 20;                     // key = cell-in-tile << 20 | point id
constexpr int MAX_TILES = 8191;                 // 13-bit tile field in the route kernel's per-point word
constexpr int MAX_NC = 512;                     // chunks per sample

struct TileWs {
    int* ctr;                // [16]   ctr[0] = worklist length
    int* tile_total;         // [B * NT] points per tile (only its zero / non-zero transition is used: worklist append)
    int* worklist;           // [B * NT] b * NT + tile of every non-empty tile
    int2* tinfo;             // [B * NT] (first slot of the tile inside its sample's NP-slot scratch, occupied cells)
    unsigned short* off;     // [B][NC][NT + 1] start of tile t inside chunk c's sorted run
    unsigned* keys;          // [B][NC * CHUNK]
    float4* recs;            // [B][NC * CHUNK][2]  xyz, features (<= 4), key bits
    float4* sorted;          // same shape: records in (tile, cell, id) order
    float4* res;             // same shape: per occupied cell: means (<= 7), cell address bits
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
template <int F>
__global__ void __launch_bounds__(256) vt_route_kernel(Src src, Geom g, const float* __restrict__ bounds, TileWs w) {
    extern __shared__ unsigned short s_hist[];          // [4][NT]: per wave, points of this chunk per tile
    __shared__ int s_red[4];
    const int NT = w.NT;
    const int chunk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
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
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int n = n0 + j * 64;
        tilecell[j] = 0;
        if (n < g.N) {
            float p[3];
            load_coords(src, g, b, n, p);
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
                if (F > 0) {
                    const float* fp = point_ptr(src.f, n, g.pps, b, g.fb, g.fp);
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
        const unsigned long long m = match_bits(tile, w.tile_bits, valid);
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
    unsigned short* offp = w.off + ((size_t)b * w.NC + chunk) * (NT + 1);
    for (int t = t0; t < t1; ++t) {
        const int h0 = s_hist[t], h1 = s_hist[NT + t], h2 = s_hist[2 * NT + t], h3 = s_hist[3 * NT + t];
        offp[t] = (unsigned short)run;
        s_hist[t] = (unsigned short)run;
        s_hist[NT + t] = (unsigned short)(run + h0);
        s_hist[2 * NT + t] = (unsigned short)(run + h0 + h1);
        s_hist[3 * NT + t] = (unsigned short)(run + h0 + h1 + h2);
        const int s = h0 + h1 + h2 + h3;
        run += s;
        if (s > 0) {
            const int old = atomicAdd(&w.tile_total[b * NT + t], s);
            if (old == 0) w.worklist[atomicAdd(&w.ctr[0], 1)] = b * NT + t;
        }
    }
    if (tid == 255) offp[NT] = (unsigned short)total;
    __syncthreads();
    const size_t slot0 = ((size_t)b * w.NC + chunk) * CHUNK;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        if ((validmask >> j) & 1u) {
            const unsigned tile = tilecell[j] & 0x1FFFu, cell = tilecell[j] >> 13;
            const size_t dst = slot0 + myh[tile] + pos_in_wave[j];
            const unsigned key = (cell << ID_BITS) | (unsigned)(n0 + j * 64);
            w.keys[dst] = key;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3 + F; ++c) v[c] = pv[j][c];
            v[7] = __uint_as_float(key);
            w.recs[dst * 2] = make_float4(v[0], v[1], v[2], v[3]);
            w.recs[dst * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ tile
template <int F>
__global__ void __launch_bounds__(256) vt_tile_kernel(Geom g, TileWs w) {
    __shared__ unsigned s_hist[4 * CELLS];              // per wave: count, then write cursor, of every cell
    __shared__ unsigned s_cell[CELLS + 1];              // occupied cells before this one << 20 | first slot of the cell
    __shared__ int s_segs[MAX_NC];                      // start of the tile's segment inside chunk c
    __shared__ int s_segp[MAX_NC + 1];                  // exclusive prefix of the segment lengths
    __shared__ int s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NT = w.NT, NC = w.NC;
    const int nwork = w.ctr[0];
    for (int e = blockIdx.x; e < nwork; e += gridDim.x) {
        const int bt = w.worklist[e];
        const int b = bt / NT, tile = bt - b * NT;
        // 1. the tile's segments, chunk by chunk (= ascending point id)
        int n = 0, base_part = 0;
        for (int c0 = 0; c0 < NC; c0 += 256) {
            const int c = c0 + tid;
            int s = 0, len = 0;
            if (c < NC) {
                const unsigned short* offp = w.off + ((size_t)b * NC + c) * (NT + 1) + tile;
                s = offp[0];
                len = (int)offp[1] - s;
                s_segs[c] = s;
            }
            int tot;
            const int ex = block_excl_scan(len, s_red, &tot);
            if (c < NC) s_segp[c] = n + ex;
            n += tot;
            base_part += s;
        }
        if (tid == 0) s_segp[NC] = n;
        int base;                                       // points of this sample in lower-numbered tiles
        block_excl_scan(base_part, s_red, &base);
        for (int i = tid; i < 4 * CELLS; i += 256) s_hist[i] = 0u;
        __syncthreads();
        // 2. count per (wave, cell); wave w owns the w-th quarter of the stream
        const int q = (((n + 3) >> 2) + 63) & ~63;
        const int r0 = min(n, wv * q), r1 = min(n, r0 + q);
        unsigned* myh = s_hist + wv * CELLS;
        const size_t key0 = (size_t)b * w.NP;
        for (int s0 = r0; s0 < r1; s0 += 64) {
            const int s = s0 + lane;
            const bool valid = s < r1;
            unsigned cell = 0;
            if (valid) {
                int lo = 0, hi = NC;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_segp[mid] <= s) lo = mid; else hi = mid;
                }
                cell = w.keys[key0 + (size_t)lo * CHUNK + s_segs[lo] + (s - s_segp[lo])] >> ID_BITS;
            }
            const unsigned long long m = match_bits(cell, 10, valid);
            const int rank = lanes_below(m), cnt = __popcll(m);
            if (valid && rank == 0) myh[cell] += (unsigned)cnt;
        }
        __syncthreads();
        // 3. cell starts: cells in order, waves in order inside a cell
        {
            unsigned h[4][4];
            int cs[4], tot = 0, occ = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cellid = 4 * tid + i;
                cs[i] = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) { h[i][k] = s_hist[k * CELLS + cellid]; cs[i] += (int)h[i][k]; }
                tot += cs[i];
                occ += cs[i] > 0 ? 1 : 0;
            }
            int dummy;
            unsigned run = (unsigned)block_excl_scan((occ << 20) | tot, s_red, &dummy);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cellid = 4 * tid + i;
                s_cell[cellid] = run;
                const unsigned start = run & 0xFFFFFu;
                s_hist[cellid] = start;
                s_hist[CELLS + cellid] = start + h[i][0];
                s_hist[2 * CELLS + cellid] = start + h[i][0] + h[i][1];
                s_hist[3 * CELLS + cellid] = start + h[i][0] + h[i][1] + h[i][2];
                run += (cs[i] > 0 ? (1u << 20) : 0u) + (unsigned)cs[i];
            }
            if (tid == 255) s_cell[CELLS] = run;
        }
        __syncthreads();
        // 4. move every record to its cell segment (stable: stream order = id order)
        float4* sorted = w.sorted + ((size_t)b * w.NP + base) * 2;
        for (int s0 = r0; s0 < r1; s0 += 64) {
            const int s = s0 + lane;
            const bool valid = s < r1;
            unsigned cell = 0;
            float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
            if (valid) {
                int lo = 0, hi = NC;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_segp[mid] <= s) lo = mid; else hi = mid;
                }
                const size_t idx = key0 + (size_t)lo * CHUNK + s_segs[lo] + (s - s_segp[lo]);
                ra = w.recs[idx * 2];
                rb = w.recs[idx * 2 + 1];
                cell = __float_as_uint(rb.w) >> ID_BITS;
            }
            const unsigned long long m = match_bits(cell, 10, valid);
            const int rank = lanes_below(m), cnt = __popcll(m);
            const unsigned before = valid ? myh[cell] : 0u;
            if (valid && rank == 0) myh[cell] = before + (unsigned)cnt;
            if (valid) {
                const size_t pos = before + rank;
                sorted[pos * 2] = ra;
                sorted[pos * 2 + 1] = rb;
            }
        }
        __syncthreads();
        // 5. one thread per cell: add the records front to back (ascending id), emit a compact record per occupied cell
        const int tz = tile % w.Tz, ty = (tile / w.Tz) % w.Ty, tx = tile / (w.Tz * w.Ty);
        float4* res = w.res + ((size_t)b * w.NP + base) * 2;
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int cellid = i * 256 + tid;
            const unsigned info = s_cell[cellid], next = s_cell[cellid + 1];
            const int start = (int)(info & 0xFFFFFu), cnt = (int)(next & 0xFFFFFu) - start;
            if (cnt > 0) {
                float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // zeros_like(self._flat_output) (:145)
                const float4* sp = sorted + (size_t)start * 2;
                int k = 0;
                for (; k + 4 <= cnt; k += 4) {                              // four records in flight, added in order
                    float4 a[4], c4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = sp[(k + u) * 2]; c4[u] = sp[(k + u) * 2 + 1]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
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
                const float Lf = (float)cnt;                                // clamp_(1) is a no-op for occupied cells (:121)
                float mean[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 3 + F; ++c) mean[c] = __fdiv_rn(acc[c], Lf);
                const int X = tx * TX + (cellid >> 7), Y = ty * TY + ((cellid >> 4) & 7), Z = tz * TZ + (cellid & 15);
                const int gc = (X * g.V + Y) * g.V + Z;
                const size_t o = (size_t)(info >> 20) * 2;
                res[o] = make_float4(mean[0], mean[1], mean[2], mean[3]);
                res[o + 1] = make_float4(mean[4], mean[5], mean[6], __int_as_float(gc));
            }
        }
        if (tid == 0) w.tinfo[bt] = make_int2(base, (int)(s_cell[CELLS] >> 20));
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------ patch
template <int F>
__global__ void __launch_bounds__(256) vt_patch_kernel(Geom g, TileWs w, float* __restrict__ out) {
    constexpr int C = 3 + F + 4;
    const int NT = w.NT, V = g.V;
    const size_t V3 = (size_t)V * V * V;
    const float Vf = (float)V;                           // self._voxel_d (:197)
    const int nwork = w.ctr[0];
    for (int e = blockIdx.x; e < nwork; e += gridDim.x) {
        const int bt = w.worklist[e];
        const int b = bt / NT;
        const int2 ti = w.tinfo[bt];
        const float4* res = w.res + ((size_t)b * w.NP + ti.x) * 2;
        for (int j = threadIdx.x; j < ti.y; j += 256) {
            const float4 ra = res[j * 2], rb = res[j * 2 + 1];
            const int gc = __float_as_int(rb.w);
            const int x = gc / (V * V), y = (gc / V) % V, z = gc % V;
            const float m[7] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z};
            float v[C];
#pragma unroll
            for (int c = 0; c < 3 + F; ++c) v[c] = m[c];
            v[3 + F + 0] = __fdiv_rn((float)x, Vf);
            v[3 + F + 1] = __fdiv_rn((float)y, Vf);
            v[3 + F + 2] = __fdiv_rn((float)z, Vf);
            v[3 + F + 3] = 1.0f;                         // (count/count > 0).float()  (:192)
            float* o = out + ((size_t)b * V3 + gc) * C;
            if ((C & 1) == 0 && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)) {
#pragma unroll
                for (int c = 0; c < C; c += 2) *reinterpret_cast<float2*>(o + c) = make_float2(v[c], v[c + 1]);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) o[c] = v[c];
            }
        }
    }
}

struct Layout {
    size_t ctr, tile_total, worklist, tinfo, off, keys, recs, sorted, res, total;
    int NT, Tx, Ty, Tz, NC;
};

Layout vt_layout(long long B, long long N, int V) {
    Layout L;
    L.Tx = (V + TX - 1) / TX; L.Ty = (V + TY - 1) / TY; L.Tz = (V + TZ - 1) / TZ;
    L.NT = L.Tx * L.Ty * L.Tz;
    L.NC = (int)((N + CHUNK - 1) / CHUNK);
    size_t p = 0;
    auto take = [&p](size_t bytes) { const size_t at = p; p += (bytes + 255) & ~(size_t)255; return at; };
    L.ctr = take(16 * sizeof(int));
    L.tile_total = take((size_t)B * L.NT * sizeof(int));      // (contiguous with ctr: one memset covers both)
    L.worklist = take((size_t)B * L.NT * sizeof(int));
    L.tinfo = take((size_t)B * L.NT * sizeof(int2));
    L.off = take((size_t)B * L.NC * (L.NT + 1) * sizeof(unsigned short));
    const size_t slots = (size_t)B * L.NC * CHUNK;
    L.keys = take(slots * sizeof(unsigned));
    L.recs = take(slots * 32);
    L.sorted = take(slots * 32);
    L.res = take(slots * 32);
    L.total = p;
    return L;
}

template <int F>
int vt_launch(const Src& src, const Geom& g, const float* bounds, float* out, const TileWs& w, hipStream_t st, hipStream_t side,
              hipEvent_t ev_join, size_t zero_bytes) {
    if (hipMemsetAsync(w.ctr, 0, zero_bytes, side) != hipSuccess) return VXB_ELAUNCH;
    const size_t lds = (size_t)4 * w.NT * sizeof(unsigned short);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)vt_route_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VXB_ELAUNCH;
    }
    hipLaunchKernelGGL(vt_route_kernel<F>, dim3(w.NC, g.B), dim3(256), lds, side, src, g, bounds, w);
    const long long tiles = (long long)g.B * w.NT;
    const int grid = (int)(tiles < 2048 ? tiles : 2048);
    hipLaunchKernelGGL(vt_tile_kernel<F>, dim3(grid), dim3(256), 0, side, g, w);
    if (hipEventRecord(ev_join, side) != hipSuccess || hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) return VXB_ELAUNCH;
    hipLaunchKernelGGL(vt_patch_kernel<F>, dim3(grid), dim3(256), 0, st, g, w, out);
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
                     hipEvent_t ev_join) {
    const Layout L = vt_layout(g.B, g.N, g.V);
    char* p = (char*)ws;
    TileWs w;
    w.ctr = (int*)(p + L.ctr);
    w.tile_total = (int*)(p + L.tile_total);
    w.worklist = (int*)(p + L.worklist);
    w.tinfo = (int2*)(p + L.tinfo);
    w.off = (unsigned short*)(p + L.off);
    w.keys = (unsigned*)(p + L.keys);
    w.recs = (float4*)(p + L.recs);
    w.sorted = (float4*)(p + L.sorted);
    w.res = (float4*)(p + L.res);
    w.NT = L.NT; w.Tx = L.Tx; w.Ty = L.Ty; w.Tz = L.Tz; w.NC = L.NC;
    w.NP = (long long)L.NC * CHUNK;
    int bits = 1;
    while ((1 << bits) < L.NT) ++bits;
    w.tile_bits = bits;
    const size_t zero_bytes = L.worklist;               // ctr + tile_total
    switch (g.F) {
        case 0: return vt_launch<0>(src, g, bounds, out, w, st, side, ev_join, zero_bytes);
        case 1: return vt_launch<1>(src, g, bounds, out, w, st, side, ev_join, zero_bytes);
        case 2: return vt_launch<2>(src, g, bounds, out, w, st, side, ev_join, zero_bytes);
        case 3: return vt_launch<3>(src, g, bounds, out, w, st, side, ev_join, zero_bytes);
        case 4: return vt_launch<4>(src, g, bounds, out, w, st, side, ev_join, zero_bytes);
        default: return VXB_EARG;
    }
}
