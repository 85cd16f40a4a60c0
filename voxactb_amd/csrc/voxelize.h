// Internal declarations shared by the two voxelizer translation units (voxelize.hip: C entry point, empty-grid store
// stream, table-based fallback chain; voxelize_tiles.hip: the tile-routed point chain).  Not part of the public C ABI.
#pragma once
#include "common.h"

namespace vox {

constexpr int MAX_SRC = 8;
constexpr int MAX_F = 8;

struct Src {
    const float* c[MAX_SRC];
    const float* f[MAX_SRC];
};
struct Geom {
    int B, n_src, pps, N, F, V, bounds_rows;
    long long cb, cc, cp, fb, fc, fp;
    const float* xf;      // optional [B, 15] rigid transform applied to every point as it is loaded (R row-major 9, t 3, c 3)
    // depth mode (proj != nullptr): source s holds a DEPTH image [B][H * W] instead of coordinates and the world point of
    // pixel (h, w) is computed here.  proj [B][n_src][14] doubles: rows of the inverse camera projection (3 x 4), near, far
    const double* proj;
    int img_w, depth_norm;
};

// voxel_grid.py:153-163 for one axis; returns the index in the (V+2)-grid, clamped to [0, V+1].
__device__ __forceinline__ int axis_index(float p, float mn, float mx, int V) {
    const float Vf = __fadd_rn((float)V, 1e-12f);          // dims_orig.float() + MIN_DENOMINATOR
    const float res = __fdiv_rn(__fsub_rn(mx, mn), Vf);    // :157
    const float den = __fadd_rn(res, 1e-12f);              // :158
    const float org = __fsub_rn(mn, res);                  // :160
    const float q = floorf(__fdiv_rn(__fsub_rn(p, org), den));
    // .int() then min/max (:161-164).  NaN / huge values end in a border cell (0 or V+1) on CPU
    // and here alike; border cells are cropped (:184), so only "is it a border" matters for them.
    int iv;
    if (!(q >= 0.0f)) iv = 0;
    else if (q > (float)(V + 1)) iv = V + 1;
    else iv = (int)q;
    return iv;
}

__device__ __forceinline__ const float* point_ptr(const float* const* src, int n, int pps, int b, long long bs, long long ps) {
    const int s = n / pps;
    const int i = n - s * pps;
    return src[s] + (long long)b * bs + (long long)i * ps;
}

// SE(3) augmentation folded into the point load (reference peract/voxel/augmentation.py:36-62): p' = (p - t) R + c with
// the points as ROW vectors.  Same operation order as vxb_se3_points_f32, so a cloud transformed by that kernel and
// voxelized lands in the same cells, bit for bit, as the fused load.
__device__ __forceinline__ void xform_point(const float* __restrict__ x, float (&p)[3]) {
    const float p0 = p[0] - x[9], p1 = p[1] - x[10], p2 = p[2] - x[11];
    p[0] = fmaf(p2, x[6], fmaf(p1, x[3], p0 * x[0])) + x[12];
    p[1] = fmaf(p2, x[7], fmaf(p1, x[4], p0 * x[1])) + x[13];
    p[2] = fmaf(p2, x[8], fmaf(p1, x[5], p0 * x[2])) + x[14];
}

// RGB-D input: world point of pixel (h, w) from its depth, as PyRep computes the stored clouds
// (PyRep/pyrep/objects/vision_sensor.py:155-175: pc = (w d, h d, d) in fp32, world = inv(K [R^T | -R^T C])[0:3] (pc, 1) in
// float64, stored as fp32; RLBench/rlbench/utils.py:205-207 for a normalised depth: d = near + depth (far - near) in fp32)
__device__ __forceinline__ void depth_to_point(const Src& src, const Geom& g, int b, int n, float (&p)[3]) {
    const int s = n / g.pps, i = n - s * g.pps;
    const float d = src.c[s][(long long)b * g.cb + (long long)i * g.cp];
    const double* P = g.proj + ((long long)b * g.n_src + s) * 14;
    float dm = d;
    if (g.depth_norm) dm = __fadd_rn((float)P[12], __fmul_rn(d, (float)(P[13] - P[12])));
    const int h = i / g.img_w, w = i - h * g.img_w;
    const double px = (double)__fmul_rn((float)w, dm), py = (double)__fmul_rn((float)h, dm), pz = (double)dm;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        p[a] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(P[4 * a], px), __dmul_rn(P[4 * a + 1], py)), __dmul_rn(P[4 * a + 2], pz)),
                                P[4 * a + 3]);
}

// coordinates of point n of sample b, after the optional rigid transform
__device__ __forceinline__ void load_coords(const Src& src, const Geom& g, int b, int n, float (&p)[3]) {
    if (g.proj) {
        depth_to_point(src, g, b, n, p);
    } else {
        const float* cp = point_ptr(src.c, n, g.pps, b, g.cb, g.cp);
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = cp[a * g.cc];
    }
    if (g.xf) xform_point(g.xf + b * 15, p);
}

}  // namespace vox

// tile-routed chain (voxelize_tiles.hip)
bool vox_tiles_supported(long long B, long long N, int V, int F);
size_t vox_tiles_ws_bytes(long long B, long long N, int V);
// enqueue the empty-grid fill (or, order 4, the reset of the previously occupied cells) and the point chain; `order`: see
// vt_launch.  The compact cell lists are double-buffered inside the workspace: this call writes list `list_out` (0 / 1) and,
// in incremental mode, resets the cells of list `list_in`.
int vox_tiles_launch(const vox::Src& src, const vox::Geom& g, const float* bounds, float* out, void* ws, hipStream_t st,
                     hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_mid, hipEvent_t ev_join, int order, int list_in,
                     int list_out);
void vox_launch_fill(float* out, int B, int V, int C, hipStream_t fs);
