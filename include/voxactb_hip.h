/* voxactb_hip.h -- C ABI of the MI355X (gfx950) kernels behind the VoxAct-B voxel hot path.
 *
 * The reference (VoxAct-B/voxactb @ 2024-10-22) has NO native code on this path: every op is a
 * stock ATen call from Python.  The functions below are therefore the kernels the Python mirror
 * classes in voxactb_amd/ call (through ctypes) where the reference calls ATen; each entry names
 * the reference code it replaces.  INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - plain C, device pointers + sizes, no torch types; `stream` is a hipStream_t passed as void*.
 *   - the caller owns every buffer (inputs, outputs, workspaces); nothing is allocated or freed here.
 *   - return 0 on success; <0 on error: -1 bad argument, -2 unsupported size, -3 workspace too
 *     small, -4 HIP launch error (hipGetLastError != hipSuccess).  Python raises RuntimeError.
 *   - all tensors fp32 unless the name says otherwise; activations are channels-last
 *     ([B, D, H, W, C] / [B, N, C]); one host thread per process drives one stream.
 */
#ifndef VOXACTB_HIP_H
#define VOXACTB_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* vxb_stream_t;

int vxb_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Voxelizer: replaces VoxelGrid.coords_to_bounding_voxel_grid (peract/voxel/voxel_grid.py:148-198,
 * _scatter_nd :127-146, _scatter_mean :106-125) and the camera flatten in QFunction.forward
 * (peract/agents/peract_bc/qattention_peract_bc_agent.py:85-93).
 *
 * Points of sample b: for source s (camera) in order, point i in [0, pts_per_src): coordinate c at
 *   coord_src[s][b*coord_bstride + c*coord_cstride + i*coord_pstride]   (same for feat_src, F chans)
 * so planar [B,3,H,W] camera images (cstride=H*W, pstride=1) and interleaved [B,N,3] clouds
 * (cstride=1, pstride=3) are both read in place.  Point id = s*pts_per_src + i (reference order).
 * bounds: [bounds_rows (1 or B), 6] = (min xyz, max xyz).  out: [B, V, V, V, 3+F+3+1].
 * Results equal the reference's CPU path bit-for-bit in all channels (sums are accumulated in
 * ascending point id, as scatter_add_ does on CPU).
 * Workspace: vxb_voxelize_workspace_bytes(); its first B*V^3 int32 must be ZERO on entry (zero the
 * whole workspace once after allocation); every successful call leaves them zero again.
 */
size_t vxb_voxelize_workspace_bytes(int B, int n_points, int V);
int vxb_voxelize_f32(const float* const* coord_src, const float* const* feat_src, int n_src,
                     int B, int pts_per_src, int F,
                     int64_t coord_bstride, int64_t coord_cstride, int64_t coord_pstride,
                     int64_t feat_bstride, int64_t feat_cstride, int64_t feat_pstride,
                     const float* bounds, int bounds_rows, int V,
                     float* out, void* workspace, size_t workspace_bytes, vxb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VOXACTB_HIP_H */
