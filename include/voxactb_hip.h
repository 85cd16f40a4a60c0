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
 *   - "ACCUMULATED" outputs are += (parameter gradients); everything else is overwritten unless an
 *     `accumulate` flag says otherwise.
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
 * xform: NULL, or [B][15] = R_b (row-major 3x3), t_b, c_b: every point is replaced by (p - t_b) R_b + c_b (row-vector
 * convention) as it is loaded -- the SE(3) augmentation's perturb_se3 (peract/voxel/augmentation.py:36-62) folded into
 * the voxelizer, same operation order as vxb_se3_points_f32 (so both routes give the same grid, bit for bit).
 * out_state: 0 = `out` holds anything (every cell is written).  1 / 2 = INCREMENTAL: `out` and `workspace` are untouched
 * since the previous successful call with the same B, N, V, F that used this very pair, and that call was made with
 * out_state 0 or 2 (-> pass 1) or with out_state 1 (-> pass 2): the workspace keeps two lists of occupied cells and the
 * value says which one the previous call wrote.  The empty-cell pattern does not depend on the input, so only the cells
 * occupied then are reset and the cells occupied now are written (~80 bytes per occupied cell instead of 40 bytes per cell
 * of the grid).  Same result, bit for bit.  Ignored (treated as 0) by the table chain.
 * Workspace: vxb_voxelize_workspace_bytes() bytes, contents arbitrary on entry for out_state 0 (nothing to pre-zero).
 */
size_t vxb_voxelize_workspace_bytes(int B, int n_points, int V);
/* Point chain used by vxb_voxelize_f32: 0 = automatic (tile-routed chain when F <= 4, V <= 200 and N <= 2^19 points per sample (512 route
 * chunks of 1024), all kernels in order on the caller's stream; heavy and light tiles in ONE launch and -- incremental calls, round 6 -- the
 * reset of the old cells inside the route launch while N <= 65 536 (64 chunks: the headline's 4 x 128 x 128 exactly), every tile through the
 * workgroup kernel above that: correct, ~1.5 x slower per point; otherwise the table-based chain), 1 = always the table-based
 * chain, 3 = the tile-routed chain of rounds 2-4 (fill, route, classify, heavy, light as separate launches), 5 = that chain with
 * the empty-grid fill on a side stream, 7 = round 5's incremental chain (the reset as its own launch) -- kept for A/B measurements.  All
 * produce identical grids. */
int vxb_voxelize_select_chain(int which);
int vxb_voxelize_f32(const float* const* coord_src, const float* const* feat_src, int n_src,
                     int B, int pts_per_src, int F,
                     int64_t coord_bstride, int64_t coord_cstride, int64_t coord_pstride,
                     int64_t feat_bstride, int64_t feat_cstride, int64_t feat_pstride,
                     const float* bounds, int bounds_rows, int V, const float* xform,
                     float* out, int out_state, void* workspace, size_t workspace_bytes, vxb_stream_t stream);

/* RGB-D input: the same voxelizer fed with DEPTH images [B][H][W] instead of stored point clouds; the
 * world point of pixel (h, w) of camera s is computed in the point load exactly as PyRep builds the stored clouds
 * (PyRep/pyrep/objects/vision_sensor.py:155-175, RLBench/rlbench/utils.py:205-256): pc = (w d, h d, d) in fp32,
 * world = M (pc, 1) in float64, rounded to fp32.  proj [B][n_src][14] doubles = the three rows of
 * M = inv([K [R^T | -R^T C]; 0 0 0 1])[0:3], then near, far; depth_normalised != 0: d = near + depth (far - near) in fp32
 * first (a 0..1 depth buffer), else the images are in metres.  Point id = s*H*W + h*W + w.  Everything else as above. */
int vxb_voxelize_depth_f32(const float* const* depth_src, const float* const* feat_src, int n_src, int B, int H, int W, int F,
                           int64_t feat_bstride, int64_t feat_cstride, int64_t feat_pstride, const double* proj,
                           int depth_normalised, const float* bounds, int bounds_rows, int V, const float* xform, float* out,
                           int out_state, void* workspace, size_t workspace_bytes, vxb_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 matrix-core GEMM (v_mfma_f32_32x32x2_f32).  Replaces nn.Linear / einsum in Attention,
 * FeedForward, lang_preprocess (perceiver_lang_io.py:80-132, :417).
 *   C[z] (+)= act(alpha * A[z] @ B[z] + bias) (+ residual[z]),  z in [0,batch): offsets (z/H)*b?1 + (z%H)*b?2
 * A(m,k) at A[m*sAm + k*sAk], B(k,n) at B[k*sBk + n*sBn]; one stride of each operand must be 1 and the
 * other a multiple of 4 (16-byte loads).  act: 0 none, 1 LeakyReLU(slope). */
int vxb_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                 int M, int N, int K, int64_t sAm, int64_t sAk, int64_t sBk, int64_t sBn, int64_t ldc,
                 int batch, int H, int64_t bA1, int64_t bA2, int64_t bB1, int64_t bB2, int64_t bC1,
                 int64_t bC2, float alpha, int act, float slope, int accumulate, vxb_stream_t stream);
/* same contract, one thread per output element: for the tiny layers (proprio 4|7|8 -> 64, MLP heads). */
int vxb_naive_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                       int64_t sAm, int64_t sAk, int64_t sBk, int64_t sBn, int64_t ldc, int act, float slope,
                       int accumulate, vxb_stream_t stream);

/* Implicit-GEMM conv3d over channels-last cubes.  Replaces Conv3DBlock.forward
 * (peract/helpers/network_utils.py:166-170; replicate padding :135-137) and, with flipped weights and
 * zero padding, its data gradient.  rows m=(b,d,h,w) over S_out^3; src voxel = o*stride + tap + off per axis
 * (replicate: clamped, else zero outside); K = kext^3*(C0+C1) ordered (tap, channel) with the channels of
 * src0 then src1 (torch.cat([d0,u0],1) without the cat, perceiver_lang_io.py:462); wt: [K][N].
 * d2s_s > 0: depth-to-space output -- column n=(phase, co), written to a (S_out*d2s_s)^3 x d2s_C grid. */
int vxb_conv3d_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                   int stride, int kext, int off, int replicate, const float* wt, int N,
                   const float* bias, float* out, int64_t ldc, int act, float slope, int accumulate,
                   int d2s_s, int d2s_C, vxb_stream_t stream);
/* Weight gradient of the same conv: part[z][K][N] = gather(src)^T @ dY over the z-th slice of positions
 * (reduce with vxb_sum_splits_f32).  d2s_s > 0: dY is the fine grid of a depth-to-space output. */
int vxb_conv3d_wgrad_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                         int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                         int d2s_s, int d2s_C, float* part, int nsplit, vxb_stream_t stream);
/* Adjoint of replicate padding after a data-gradient conv: dst[b,j,c] (+)= sum of src[b,i,c0+c] over the padded
 * positions i that clamp to j; optionally times LeakyReLU'(lrelu_of[b,j,c]) (the producer's activation). */
int vxb_fold_pad_f32(const float* src, int Sp, int Cs, int c0, float* dst, const float* lrelu_of, int B, int S,
                     int C, int pad, int accumulate, float slope, vxb_stream_t stream);
/* Rigid transform of a channels-first point cloud [B, 3, n] (apply_se3_augmentation / perturb_se3,
 * peract/voxel/augmentation.py:36-57): dst[b, :, i] = (src[b, :, i] - t_b) R_b + c_b with points as ROW vectors;
 * xf [B][15] = R_b (row-major 3x3), t_b (gripper position), c_b (new, clamped centre). */
int vxb_se3_points_f32(const float* src, float* dst, const float* xf, int B, int64_t n, vxb_stream_t stream);
/* Pose / label half of apply_se3_augmentation (peract/voxel/augmentation.py:98-177 with helpers/utils.py:63-64, 92-97,
 * 104-116), on the device and without a host round trip.  For attempt k = 0 .. K-1 and sample b: shift = (bounds_max -
 * bounds_min) * aug_xyz * shift_unit[k][b] (float64, as the reference's float64 `trans_aug_range` makes it), rotation
 * Rx Ry Rz of rpy_steps[k][b] * rot_aug_resolution degrees, T' = T_grip R with T'[:3,3] += shift, translation index =
 * min(floor((t' - min) / (res + 1e-12)), V - 1) in float64 with the bounds row of sample b (row 0 when layer == 0,
 * :161-162), rotation index = round((euler_xyz(T') + 180) / rot_resolution) mod (360 / rot_resolution).  The FIRST
 * attempt whose translation indices are >= 0 for the whole batch wins (:116); none in K: status[0] = -1 and the labels
 * are -1 (the caller raises, :119-120).  Outputs of the winning attempt: trans_idx [B][3], rot_grip_idx [B][4] (grip bit
 * copied from rot_grip_in), xf [B][15] = R (row-major), t_grip, clamp(t_grip + shift, batch-wide bounds) -- the operand of
 * vxb_voxelize_f32 / vxb_se3_points_f32; status[0] = winning attempt.  pose [B][7] = xyz + quaternion (x, y, z, w). */
int vxb_se3_relabel_f32(const float* pose, const int32_t* rot_grip_in, const float* bounds, int bounds_rows, int layer,
                        const float* shift_unit, const int32_t* rpy_steps, int K, int B, double aug_x, double aug_y,
                        double aug_z, float rot_aug_resolution, int V, float rot_resolution, int32_t* trans_idx,
                        int32_t* rot_grip_idx, float* xf, int32_t* status, vxb_stream_t stream);
/* Two arms under ONE perturbation (apply_se3_augmentation_2Robots, peract/voxel/augmentation.py:187-348, the
 * one_policy_more_heads baseline): both poses use the same draws, an attempt wins only when BOTH arms' translation indices
 * are >= 0 for the whole batch (:237), xf is centred on the RIGHT arm's pose (:346).  Otherwise as vxb_se3_relabel_f32. */
int vxb_se3_relabel_pair_f32(const float* pose_right, const int32_t* rot_grip_right, const float* pose_left,
                             const int32_t* rot_grip_left, const float* bounds, int bounds_rows, int layer,
                             const float* shift_unit, const int32_t* rpy_steps, int K, int B, double aug_x, double aug_y,
                             double aug_z, float rot_aug_resolution, int V, float rot_resolution, int32_t* trans_idx_right,
                             int32_t* rot_grip_idx_right, int32_t* trans_idx_left, int32_t* rot_grip_idx_left, float* xf,
                             int32_t* status, vxb_stream_t stream);
/* bf16 matrix-core twins ("throughput mode", v_mfma_f32_32x32x16_bf16, fp32 accumulate): the fp32 A operand is rounded
 * to bf16 (RNE) while it is staged into LDS; weights come as bf16 [N][K] (K contiguous, K % 8 == 0; C0, C1 % 32 == 0). */
int vxb_gemm_bf16w_f32(const float* A, int64_t lda, const void* Bw, float* C, int64_t ldc, const float* bias,
                       const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                       vxb_stream_t stream);
int vxb_conv3d_bf16w_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                         int stride, int kext, int off, int replicate, const void* wt_bf16, int N,
                         const float* bias, float* out, int64_t ldc, int act, float slope, int accumulate,
                         int d2s_s, int d2s_C, vxb_stream_t stream);
/* "bf16x3" split-product twins of the three entries above and below: each fp32 operand value a is used as
 * hi = bf16(a), lo = bf16(a - hi) and a product is evaluated as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32
 * accumulation -- dropped terms <= 2^-16 relative per product, 3 MFMAs at 16x the fp32-MFMA rate.  Weights come
 * pre-split as bf16 planes [2][N][K] (hi plane, then lo plane). */
int vxb_gemm_bf16x3_f32(const float* A, int64_t lda, const void* Bw, float* C, int64_t ldc, const float* bias,
                        const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                        vxb_stream_t stream);
int vxb_conv3d_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                          int stride, int kext, int off, int replicate, const void* wt_bf16, int N,
                          const float* bias, float* out, int64_t ldc, int act, float slope, int accumulate,
                          int d2s_s, int d2s_C, vxb_stream_t stream);
int vxb_conv3d_wgrad_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                                int d2s_s, int d2s_C, float* part, int nsplit, float* possum, vxb_stream_t stream);
/* ... with ONE fp16 product per term: the gradient operand (src0 when grad_is_src0 != 0: the plain-GEMM form of a linear layer's
 * weight gradient, src0 = its dY; else dy) times scale[0] (device, power of two) before the conversion to half, `part` = scale[0] * dW,
 * the other operand saturating at +-65504.  next_scale (optional, [2]; amax_ws of vxb_conv3d_wgrad_f16_amax_words words): the scale
 * vxb_absmax_scale_f32 would give for the gradient operand this launch read -- for the next step (delayed scaling).  sum_dst (optional,
 * with next_scale): [(tap, ci)][N] = (sum_accumulate ? sum_dst : 0) + scale[1] * the sum of the nsplit partial results, written by the
 * same finishing launch that computes next_scale (the caller then skips vxb_sum_splits_dev_f32). */
int vxb_conv3d_wgrad_f16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                             int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                             int d2s_s, int d2s_C, float* part, int nsplit, float* possum, const float* scale,
                             int grad_is_src0, float* next_scale, float* amax_ws, float* sum_dst, int sum_accumulate,
                             vxb_stream_t stream);
size_t vxb_conv3d_wgrad_f16_amax_words(int C0, int C1, int kext, int N, int nsplit, int grad_is_src0);
void vxb_debug_set_wgrad_bm256(int on);       /* experiment knob: 256-row tiles of the fp16 weight-gradient kernel (measured slower: default off) */
void vxb_debug_set_gemm_wide_waves(int waves);  /* wide linear-layer GEMMs: 8 = one 128 x 512 workgroup of 8 waves per CU, 4 = two 128 x 256 workgroups of 4 waves per CU */
void vxb_debug_set_gemm_wide_experiment(int bits);  /* timing experiments of gemm_wide.hip (WRONG results): 1 no weight-fragment loads in the loop, 2 no A loads, 4 no A staging / barrier, 8 no epilogue; 32 = row blocks fastest in the grid (right results); 64 = the epilogue of rounds 3 - 5 (stores straight out of the accumulators; correct results) in gemm_wide_kernel and conv_poly_wide_x3_kernel */
void vxb_debug_set_wide_min_rows(int rows);  /* rows from which the wide weight-gradient kernel is dispatched (default 16384; tests lower it) */
void vxb_debug_set_wgrad_lin(int mode);       /* A/B switch of the linear layers' fp16 weight gradients: 2 (default) wide kernel where one operand has 512 channels, 1 pipelined 128x128 kernel only, 0 generic kernel; + 16: the wide kernel's workgroups in the plain launch order; + 32: 32-position tiles; + 64: every step through the guarded form of rounds 3 - 5 (all: correct results, A/B of round 6's changes) */
/* experiment knobs of the LDS-halo conv kernels (conv_halo_bf16.hip, wgrad_halo.hip; tools/bench_halo.py, tools/bench_wgrad_halo*.py).
   Everything the library exports is declared in this header: libvoxactb_hip.so is linked with csrc/exports.map (vxb_* only). */
void vxb_debug_set_halo_waves(int nw);              /* 4 (default) or 8 waves per workgroup */
void vxb_debug_set_halo_wn(int wn);                 /* waves along the output channels in the fragment-from-global kernels: 1 or 2; 0 = default */
void vxb_debug_set_halo_experiment(int bits);       /* timing experiments (1, 2, 16, 32, 64: WRONG results -- no halo staging / weight-fragment loads / conversion / halo loads / output stores after the first chunk; 4: no skipping of empty waves; 0x100..0xf00: start-up stagger, right results): profiles/r05_final_conv_ablation.log */
void vxb_debug_set_wgrad_halo_experiment(int bits); /* timing experiments of wgrad_halo.hip (WRONG results): staging only / matrix loop only */
void vxb_debug_set_wgrad_halo_chunks(int nch);      /* input-channel chunks per workgroup: 1 or 2; 0 = default */
void vxb_debug_set_wgrad_halo_shape(int shape);     /* voxel tile: 0 = 2x8x8, 1 = 4x4x8, -1 = choose by the grid edge (default) */
/* dst[i] = convert(src[idx[i]]), i < n: a weight tensor re-laid out through a cached index table in one pass (ops.gather_cvt: the
   polyphase up-conv's effective weight, network_utils.py:245-250, into the fragment orders its forward and data-gradient kernels
   read).  mode 0: dst fp16 (round to nearest even).  mode 1: dst bf16 planes: entry j < plane_off is the hi half of src[j],
   j >= plane_off the lo half bf16(v - hi) of src[j - plane_off] (the values of vxb_split_bf16_f32).  idx < 0 gives 0.
   n % 8 == 0; idx, dst 16-byte aligned. */
int vxb_gather_cvt_f32(const float* src, const int32_t* idx, int64_t n, void* dst, int mode, int32_t plane_off, vxb_stream_t stream);

/* Direct-to-LDS variants (global_load_lds_dwordx4, no register round trip): BOTH operands are bf16 planes in HBM.
 * vxb_split_bf16_f32 makes the activation planes [nplanes][rows][cols] (plane 0 = bf16(x), plane 1 = bf16(x - plane 0));
 * weights are the same [nplanes][N][K] planes as above.  nplanes = 1 ('bf16') or 2 ('bf16x3').  K % 32 == 0 (conv:
 * Cin % 32 == 0, one source).  zeros: >= 16 bytes of device zeros, fetched for zero-padded taps. */
int vxb_split_bf16_f32(const float* src, int64_t ld, int64_t rows, int cols, void* dst_planes, int nplanes,
                       vxb_stream_t stream);
/* Wide form of vxb_gemm_bf16x3_f32 for N % 512 == 0 (the linear layers of the Perceiver trunk, perceiver_lang_io.py:74-132, forward
 * and data gradient): one workgroup owns 128 rows x 512 columns (grid.y = N / 512 column groups), A is read three k-tiles ahead, the
 * weights come ONLY in MFMA fragment order (Bw_frag = [N / 32][K / 16][2][64][8] bf16, see vxb_gemm_dl_f32) and never touch LDS.
 * K % 32 == 0, K >= 64.  Bit-identical to vxb_gemm_bf16x3_f32. */
int vxb_gemm_wide_bf16x3_f32(const float* A, int64_t lda, const void* Bw_frag, float* C, int64_t ldc, const float* bias,
                             const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                             vxb_stream_t stream);
/* ... with the result also as ONE fp16 plane f16_out [M][N] (saturating round-to-nearest: the bits of vxb_split_f16_f32 applied to C): the
 * k | v operand plane of vxb_flash2_attn_fwd / _bwd out of the to_kv projection's epilogue (perceiver_lang_io.py:112-113) instead of by a
 * pass over kv. */
int vxb_gemm_wide_bf16x3_f16out_f32(const float* A, int64_t lda, const void* Bw_frag, float* C, int64_t ldc, const float* bias,
                                    const float* residual, int M, int N, int K, int act, float slope, int accumulate, void* f16_out,
                                    vxb_stream_t stream);
/* ... on TWO fp16 products per term for the data gradients of the linear layers (dX = dY @ W, perceiver_lang_io.py:74-132 backward):
 * A = dY times scale[0] / 16 as an fp16 hi + lo pair, the weights as one fp16 value (Bw_frag16: single-plane fragment order of the fp16
 * [N][K] matrix, vxb_split_bf16_batch_f32 flag bit 3), the sums times 16 scale[1]; scale = device {2^k, 2^-k}. */
int vxb_gemm_wide_f16x2_f32(const float* A, int64_t lda, const void* Bw_frag16, float* C, int64_t ldc, const float* residual,
                            int M, int N, int K, int accumulate, const float* scale, vxb_stream_t stream);
/* FeedForward with GEGLU (perceiver_lang_io.py:74-78, :100-106) on the wide kernel.  _fwd: the up-projection h [M][2 F] = A @ W^T + bias
 * and gg [M][F] = h[:, :F] * gelu(h[:, F:]) from one launch (Bw_frag: fragment order with the value / gate rows interleaved, see
 * vxb_split_bf16_batch_f32 flag bit 2); 2 F % 512 == 0.  _bwd: the data gradient of the down-projection d(gg) = dY [M][K] @ W2 (Bw_frag:
 * fragment order of the transposed weight's planes [2][F][K]) with GEGLU's backward applied in the epilogue: dh [M][2 F] = [d * gelu(h_gate)
 * | d * h_value * gelu'(h_gate)]; F % 512 == 0.  Same bits as vxb_gemm_bf16x3_f32 followed by vxb_geglu_fwd_f32 / vxb_geglu_bwd_f32. */
int vxb_gemm_wide_geglu_fwd_f32(const float* A, int64_t lda, const void* Bw_frag, const float* bias, float* h, float* gg, int M, int F,
                                int K, vxb_stream_t stream);
int vxb_gemm_wide_geglu_bwd_f32(const float* dY, int64_t lda, const void* Bw_frag, const float* h, float* dh, int M, int F, int K,
                                vxb_stream_t stream);
/* ... and the _bwd launch on the two-fp16-product arithmetic of vxb_gemm_wide_f16x2_f32 (Bw_frag16: ONE fp16 plane of the transposed weight
 * in fragment order; scale: device {2^k, 2^-k}, the operand scale of dY): d(gg) has the bits of that entry, dh those of vxb_geglu_bwd_f32
 * applied to it.  FeedForward's backward, perceiver_lang_io.py:74-78 / :100-106 through autograd. */
int vxb_gemm_wide_geglu_bwd_f16x2_f32(const float* dY, int64_t lda, const void* Bw_frag16, const float* h, float* dh, int M, int F, int K,
                                      const float* scale, vxb_stream_t stream);
/* Forward of the polyphase up-conv (network_utils.py:245-250 as ops.conv3_polyphase_fwd evaluates it: the low-res kext^3 conv with
 * s^3 * 64 phase columns and a depth-to-space store) on the same 128 x 512 workgroup tiles: z [B, S^3, Cin] fp32 is gathered and split
 * in the kernel (no activation planes), wt_frag = the [N][kext^3 * Cin] weights (column blocks in `perm` order) as hi / lo planes in
 * MFMA fragment order, blockmask [N / 64] = tap mask of every 64-column block (a wave skips the taps its phase does not reach:
 * exact zeros), perm [N / 64] = fine-grid phase of every block.  Bit-identical to vxb_conv3d_dl_f32 with the same masks. */
int vxb_conv3_poly_wide_bf16x3_f32(const float* z, int Cin, int B, int S, int kext, int off, int replicate, const void* wt_frag,
                                   int N, const float* bias, float* out, int act, float slope, int d2s_s,
                                   const int32_t* blockmask, const int32_t* perm, vxb_stream_t stream);
/* The same split for MANY matrices in one launch (all linear-layer weights of a training step, each also in transposed form
 * for the data-gradient GEMM): desc = device table of n x 6 int64 {src fp32 [rows][cols], dst planes, rows, cols,
 * flags, first tile}; dst receives [nplanes][rows][cols] or, transposed (flags bit 0), [nplanes][cols][rows]; with flags bit 1 the
 * (possibly transposed) [n][k] matrix goes out in MFMA fragment order [n / 32][k / 16][nplanes][64][8] instead (n % 32 == 0, k % 16
 * == 0: the Bw_frag operand of vxb_gemm_wide_bf16x3_f32 / vxb_gemm_dl_f32), with bit 2 as well its rows interleaved per 64-row block as
 * [32 value rows | their 32 gate rows] (the operand of vxb_gemm_wide_geglu_fwd_f32); tiles of an entry = ceil(rows/64) * ceil(cols/64),
 * first tiles ascending, total_tiles = their sum.  Bit-identical to vxb_split_bf16_f32. */
int vxb_split_bf16_batch_f32(const int64_t* desc, int n, int64_t total_tiles, int nplanes, vxb_stream_t stream);
int vxb_gemm_dl_f32(const void* A_planes, const void* Bw_planes, const void* Bw_frag, int nplanes, float* C, int64_t ldc, const float* bias,
                    const float* residual, int M, int N, int K, int act, float slope, int accumulate,
                    vxb_stream_t stream);
int vxb_conv3d_dl_f32(const void* src_planes, int Cin, int B, int S_in, int S_out, int stride, int kext, int off,
                      int replicate, const void* wt_planes, const void* wt_frag, int nplanes, int N, const float* bias, float* out,
                      int64_t ldc, int act, float slope, int accumulate, int d2s_s, int d2s_C, const void* zeros,
                      const uint32_t* tapmask, const int32_t* d2s_perm, vxb_stream_t stream);
/* (Bw_frag / wt_frag, optional: the same weights, rows zero-padded to a multiple of 128, in MFMA fragment order
 * [ceil(N/128)*4][K/16][nplanes][64 lanes][8]
 * -- lane (col = lane & 31, half = lane >> 5) of column tile t, k-step ks holds W[32 t + col][16 ks + 8 half .. + 7]; the
 * kernel then reads its B fragments straight from global memory, one k-tile ahead, and only A goes through LDS.) */
/* (tapmask / d2s_perm, both optional: block-sparse weights of the polyphase up-conv, network_utils.py:245-250 -- a fine
 * phase only sees the low-res taps its trilinear footprint reaches.  tapmask[column tile] bit t set <=> tap t has
 * non-zero weights in that 128-column tile; d2s_perm[p] = fine-grid phase held by 64-column block p.) */
/* LDS-halo specialisation of vxb_conv3d_bf16w_f32 for kext == 3, stride == 1 (the `final` conv of the Q-function,
 * perceiver_lang_io.py:462-466, and its data gradient): a 4x8x8 block of output voxels stages its 6x10x10 input halo
 * once instead of once per tap.  C0, C1 % 32 == 0, N % 64 == 0; out [B, S_out^3, N] is overwritten.  The x3 entry
 * takes the hi/lo weight planes [2][N][K] of the bf16x3 split.
 * s2d_s > 0: src0 is a fine grid [B, (S_in*s2d_s)^3, s2d_C] read by space-to-depth (input channel = (phase, co)) -- the
 * data gradient of the polyphase up-conv.  d2s_s > 0: depth-to-space output, 64 channels per phase (its forward). */
/* taptab / ncls / tap_total (optional; s2d input + wfrag only): block-sparse weights of the polyphase up-conv's data
 * gradient.  taptab = ncls footprint classes x 32 ints (entries 0..26: LDS offset, in bf16 units, of the n-th listed tap =
 * ((td*10 + th)*12 + tw)*40; entry 31: list length, a multiple of 3, padded with zero-weight taps) followed by
 * (class, number of taps listed before the phase's first chunk) for each of the s2d_s^3 phases; wfrag then holds only
 * the listed taps, [column block][tap_total][...]. */
int vxb_conv3_halo_bf16w_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                             int off, int replicate, const void* wt_bf16, int N, const float* bias, float* out,
                             int act, float slope, int s2d_s, int s2d_C, int d2s_s, const void* wfrag, const int32_t* taptab,
                              int ncls, int tap_total, vxb_stream_t stream);
int vxb_conv3_halo_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                              int off, int replicate, const void* wt_bf16, int N, const float* bias, float* out,
                              int act, float slope, int s2d_s, int s2d_C, int d2s_s, const void* wfrag, const int32_t* taptab,
                              int ncls, int tap_total, vxb_stream_t stream);
/* The tap-list launch above (space-to-depth input + taptab: the data gradient of the polyphase up-conv, network_utils.py:245-250 via
 * perceiver_lang_io.py:449-455) with the reduction split over ksplit workgroups per tile: part p accumulates the chunks
 * kparts[p] .. kparts[p + 1] - 1 (DEVICE int32 [ksplit + 1]; a chunk = 16 input channels in bf16x3, 32 in bf16) into
 * out_parts[p] [B, S_out^3, N]; the caller sums the parts in order (vxb_sum_splits_f32).  x3 = 1: wt_bf16 = hi/lo planes;
 * x3 = 3 ("fp16x2", two MFMAs per product): src_fine * scale[0] (device, power of two) as an fp16 hi + lo pair, the weights ONE fp16
 * value each (wfrag from the fp16 matrix, single plane), parts multiplied by scale[1]; scale = NULL otherwise. */
int vxb_conv3_s2d_splitk_f32(const float* src_fine, int C0, int B, int S_in, int S_out, int off, const void* wt_bf16, int x3,
                             int N, float* out_parts, int s2d_s, int s2d_C, const void* wfrag, const int32_t* taptab,
                             int ncls, int tap_total, int ksplit, const int32_t* kparts, const float* scale, vxb_stream_t stream);
/* `final` conv + SpatialSoftmax3D / global max pool of its output in one pass over the output (perceiver_lang_io.py:462 then :470;
 * network_utils.py:768-800): vxb_conv3_halo_bf16x3_f32 (N = 64, two sources, replicate padding, S_in = S_out = S, wfrag required)
 * whose epilogue also takes the online-softmax partial of every (tile, channel); one small launch merges them into the outputs of
 * vxb_ss3d_max_fwd_f32 (out_ss [B,192], out_max [B,64], stats [B,64,2], argmax [B,64]).  lin: the S coordinate values.
 * part_ws: vxb_conv3_halo_ss3d_ws(B, S) floats.  `out` is bit-identical to the plain entry's. */
size_t vxb_conv3_halo_ss3d_ws(int B, int S);
int vxb_conv3_halo_ss3d_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S, const void* wt_bf16,
                                   const float* bias, float* out, int act, float slope, const void* wfrag, const float* lin,
                                   float* part_ws, float* out_ss, float* out_max, float* stats, int32_t* argmax,
                                   vxb_stream_t stream);
/* The same launch with the filter's depth axis evaluated by Winograd's F(2, 3): per (kh, kw) four products for two output depths
 * instead of six -- two thirds of the matrix work (helpers/network_utils.py:128-170 `final`, the step's largest kernel).  wfrag_wg: the
 * 36 transformed taps (xi, kh, kw), xi = {g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2} over the depth taps, as bf16 hi/lo planes in
 * fragment order [1][chunk][36][column tile 2][plane 2][lane 64][8] (ops.halo_wfrag_wg).  S even.  Agrees with the direct entry to
 * fp32 rounding of the transforms (~1e-6 relative), not bit for bit. */
int vxb_conv3_halo_ss3d_wg_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S, const float* bias,
                                      float* out, int act, float slope, const void* wfrag_wg, const float* lin,
                                      float* part_ws, float* out_ss, float* out_max, float* stats, int32_t* argmax,
                                      vxb_stream_t stream);
/* wfrag (optional, NULL = weights staged through LDS per tap): the same weights pre-shuffled into MFMA fragment order,
 * [N/64][chunk][tap][column tile 2][k half or plane 2][lane 64][8 bf16] with chunk = 32 channels ('bf16') or 16 ('bf16x3');
 * the kernel then loads its B fragments straight from global memory and the 27-tap loop has no barrier.
 * Data gradient of a 3x3x3 replicate-padded conv fused with vxb_fold_pad_f32 (the padded-domain gradient never reaches
 * HBM): dy [B, S^3, C0]; wt_bf16 = data-gradient weights [N][27*C0] (x3 != 0: planes [2][N][K]); for each 64-column block
 * nb of the N <= 128 columns: dst_nb [B, S^3, 64] (+)= fold(...) (* LeakyReLU'(y_nb) when y_nb != NULL).
 * Replaces the pair at perceiver_lang_io.py:462 (backward of `final` into d0 and u0). */
int vxb_conv3_dgrad_fold_f32(const float* dy, int C0, int B, int S, const void* wt_bf16, int x3, int N, float* dst0,
                             float* dst1, const float* y0, const float* y1, int acc0, int acc1, float slope,
                             const void* wfrag, float* dst_scale, float* scale_ws, float* dst_colsum, float* colsum_ws,
                             vxb_stream_t stream);
/* workgroups of that launch = words of scale_ws needed when dst_scale is asked for; dst_colsum ([64], accumulated: column sums of
 * dst0 as written = the bias gradient of the conv whose activation y0 is; N = 64) needs colsum_ws of 64 * (that + 64) floats */
size_t vxb_conv3_dgrad_fold_blocks(int B, int S, int N);
/* ... one 64-column block of it with a single fp16 product per term (dy * scale[0] -> half, result * scale[1]; scale on the
 * device, vxb_absmax_scale_f32; weights in fragment order of the fp16 [64][27 C0] matrix): the d(d0) half of `final`'s data
 * gradient, which only feeds the weight gradient of the 1x1x1 input conv (a leaf of the backward pass). */
int vxb_conv3_dgrad_fold_f16_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16, float* dst, const float* y,
                                 int acc, float slope, const float* scale, vxb_stream_t stream);
/* ... with the filter's depth axis by Winograd's F(2, 3) (see vxb_conv3_dgrad_fold_f16x2_wg_f32): wfrag_f16_wg = ops.halo_wfrag_x2_wg; S even,
 * scale required. */
int vxb_conv3_dgrad_fold_f16_wg_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16_wg, float* dst, const float* y,
                                 int acc, float slope, const float* scale, vxb_stream_t stream);
/* ... on TWO fp16 products per term for a block that propagates (the d(u0) half): dy * scale[0] as an fp16 hi + lo pair, the weights
 * as one fp16 value (wfrag_f16x2: single-plane fragment order of the fp16 [64][27 C0] matrix); optional by-products as in
 * vxb_conv3_dgrad_fold_f32 (dst_scale [2] + scale_ws, dst_colsum [64] ACCUMULATED + colsum_ws). */
int vxb_conv3_dgrad_fold_f16x2_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16x2, float* dst, const float* y,
                                   int acc, float slope, const float* scale, float* dst_scale, float* scale_ws,
                                   float* dst_colsum, float* colsum_ws, vxb_stream_t stream);
/* ... with the filter's depth axis by Winograd's F(2, 3) (see vxb_conv3_halo_ss3d_wg_bf16x3_f32): wfrag_f16x2_wg = ops.halo_wfrag_x2_wg,
 * the 36 transformed taps rounded to fp16 after the transform; S even.  dy * scale[0] / 2 is carried (the transformed operand is a sum
 * of two values).  Against the direct entry: ~1e-6 of the largest output (tests/test_halo_winograd_gpu.py). */
int vxb_conv3_dgrad_fold_f16x2_wg_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16x2_wg, float* dst, const float* y,
                                   int acc, float slope, const float* scale, float* dst_scale, float* scale_ws,
                                   float* dst_colsum, float* colsum_ws, vxb_stream_t stream);
/* ... when that block is the data gradient of a 1x1x1 conv's output y = lrelu(W_in x + b_in) whose input x [B, S^3, 10] is a detached
 * tensor (the input conv of the Q-function, perceiver_lang_io.py:357; agent :100): it only feeds dW_in [64][10] / db_in [64], so it is
 * not stored -- the epilogue multiplies it with LeakyReLU'(y) and x, and the sums are ACCUMULATED into dW / db.
 * ws: 704 * (vxb_conv3_dgrad_fold_blocks(B, S, 64) + 512) floats. */
int vxb_conv3_dgrad_fold_f16_wgin_f32(const float* dy, int C0, int B, int S, const void* wfrag_f16, const float* y, const float* x,
                                      float slope, const float* scale, float* ws, float* dW, float* db, vxb_stream_t stream);
/* LDS-halo weight gradient of the same 3x3x3 stride-1 convs (contract of vxb_conv3d_wgrad_f32 with kext = 3, stride = 1;
 * the z slices of part[z][K][N] are runs of 128-voxel tiles, 2x8x8 or 4x4x8 -- chosen by the voxels wasted on the
 * edge of an S_out^3 grid; vxb_conv3_wgrad_halo_tiles returns their number).  C0, C1 % 16 == 0, N % 64 == 0; d2s needs
 * d2s_C == 64.  phase_mask (optional, d2s only: the polyphase up-conv, network_utils.py:245-250): one word per 64-column
 * block (= fine-grid phase), bit t set <=> weight block (tap t, phase) is not structurally zero; the other blocks are
 * neither computed nor stored (the caller zero-fills `part`). */
int vxb_conv3_wgrad_halo_bf16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                  int off, int replicate, const float* dy, int N, int64_t ldy, int d2s_s, int d2s_C,
                                  float* part, int nsplit, const uint32_t* phase_mask, vxb_stream_t stream);
int vxb_conv3_wgrad_halo_bf16x3_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                    int off, int replicate, const float* dy, int N, int64_t ldy, int d2s_s, int d2s_C,
                                    float* part, int nsplit, const uint32_t* phase_mask, vxb_stream_t stream);
/* ... with ONE fp16 product per term (v_mfma_f32_16x16x32_f16, fp32 accumulate): the weight gradients of `final`
 * (perceiver_lang_io.py:462) and of the up-conv (network_utils.py:245-250) in the default precision -- a weight gradient is a
 * leaf of the backward pass, its operand rounding (2^-12) averages out over >= 10^5 voxels and never propagates.  x saturates at
 * +-65504; dY is multiplied by *dy_scale (device float, a power of two from vxb_absmax_scale_f32; NULL = 1) before the
 * conversion, `part` holds dy_scale * dW (undo it with vxb_sum_splits_dev_f32). */
int vxb_conv3_wgrad_halo_f16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                                 int off, int replicate, const float* dy, int N, int64_t ldy, int d2s_s, int d2s_C,
                                 float* part, int nsplit, const uint32_t* phase_mask, const float* dy_scale,
                                 vxb_stream_t stream);
/* ... of a 5x5x5 stride-1 conv (first conv of the decoder's up-block: network_utils.py:236-244, perceiver_lang_io.py:381-389) on the
 * same kernel: the 125 taps as eight shifted 3x3x3 blocks in one launch.  shift_rows: device int32 [8][27] = the row block
 * (kd * 5 + kh) * 5 + kw of every (shift, tap), -1 where the tap belongs to another shift (shift bit 4 / 2 / 1 set: the block of
 * axis d / h / w covers the offsets {0, +1, +2} and owns the last two; clear: {-2, -1, 0}).  part [nsplit][125 * (C0 + C1)][N]. */
int vxb_conv3_wgrad_halo5_f16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S, int off, int replicate,
                                  const float* dy, int N, int64_t ldy, float* part, int nsplit, const int32_t* shift_rows,
                                  const float* dy_scale, vxb_stream_t stream);
size_t vxb_conv3_wgrad_halo_tiles(int B, int S_out, int x3);
/* bf16 matrix-core weight gradient (same contract as vxb_conv3d_wgrad_f32): both operands are staged position-major and
 * transposed for the matrix cores by ds_read_b64_tr_b16.  possum (optional; plain GEMM form only: kext = S_in = S_out = 1, one
 * source): possum[z][C0] = fp32 sums over the z-th slice of positions of src0's rows, i.e. with src0 = dY of a linear layer its
 * bias gradient (nn.Linear backward, perceiver_lang_io.py:80-132) from the same launch instead of a second pass over dY. */
int vxb_conv3d_wgrad_bf16_f32(const float* src0, const float* src1, int C0, int C1, int B, int S_in, int S_out,
                              int stride, int kext, int off, int replicate, const float* dy, int N, int64_t ldy,
                              int d2s_s, int d2s_C, float* part, int nsplit, float* possum, vxb_stream_t stream);
/* Polyphase form of Upsample(x s, trilinear, align_corners=False) followed by Conv3d(k) (network_utils.py:245-250):
 * Weff[(j3*Cin+ci)][(r3*Cout+co)] = sum_t W[co][ci][t] * L[r][t][j] per axis; and its adjoint (dW ACCUMULATED). */
int vxb_polyphase_weights_f32(const float* W, const float* L, float* Weff, int Cin, int Cout, int k, int s, int kl,
                              vxb_stream_t stream);
int vxb_polyphase_weights_bwd_f32(const float* dWeff, const float* L, float* dW, int Cin, int Cout, int k, int s, int kl,
                                  vxb_stream_t stream);

/* 1x1x1 input conv 10 -> 64 + LeakyReLU (perceiver_lang_io.py:357) and its parameter gradients (ACCUMULATED). */
int vxb_pointwise_fwd_f32(const float* x, const float* W, const float* bias, float* y, int64_t nvox, int Cin,
                          int Cout, float slope, vxb_stream_t stream);
int vxb_pointwise_wgrad_f32(const float* x, const float* y, const float* dy, float* dW, float* db, float* part_ws,
                            int64_t nvox, int Cin, int Cout, float slope, vxb_stream_t stream);

/* 3x3x3 conv with one output channel = trans_decoder (perceiver_lang_io.py:465): forward, data gradient
 * (replicate adjoint folded in, optional LeakyReLU' of the producer), parameter gradients (ACCUMULATED). */
int vxb_conv3_c1_fwd_f32(const float* u, const float* w, const float* bias, float* q, int B, int S, int C,
                         vxb_stream_t stream);
int vxb_conv3_c1_dgrad_f32(const float* dq, const float* w, const float* u, float* du, int B, int S, int C,
                           int accumulate, int apply_lrelu_mask, float slope, vxb_stream_t stream);
int vxb_conv3_c1_wgrad_f32(const float* u, const float* dq, float* dw, float* db, float* part_ws, int B, int S, int C,
                           vxb_stream_t stream);
/* Matrix-core versions of the two entries above for the 'bf16x3' / 'bf16' precisions (bf16x3 products, fp32 accumulate):
 * the 27 taps are the MFMA column dimension -- forward P[v'][t] = u[v'] . w[:, t] on the tile's halo, then
 * q[v] = bias + sum_t P[v + t - 1][t]; weight gradient dW = u^T Bq with Bq[v'][t] = the dq of the outputs whose tap t
 * reads u[v'] (trans_decoder, perceiver_lang_io.py:316-321 / :466).  C = 64.  wfrag_ws: 8 KB of device scratch;
 * part_ws: vxb_conv3_c1_wgrad_mfma_blocks(B, S) * (64*27 + 1) floats. */
int vxb_conv3_c1_fwd_mfma(const float* u, const float* w, const float* bias, float* q, int B, int S, void* wfrag_ws,
                          vxb_stream_t stream);
int vxb_conv3_c1_wgrad_mfma(const float* u, const float* dq, float* dw, float* db, float* part_ws, int B, int S,
                            vxb_stream_t stream);
/* vxb_conv3_c1_wgrad_mfma on ONE fp16 product per term (this gradient is a leaf of the backward pass, perceiver_lang_io.py:466 backward):
   dq * dq_scale[0] is the fp16 operand, the sums are multiplied by dq_scale[1]; dq_scale: device {2^k, 2^-k} with max |dq| * 2^k in
   [2^14, 2^15) (vxb_absmax_scale_f32).  Same workspace as vxb_conv3_c1_wgrad_mfma. */
int vxb_conv3_c1_wgrad_f16(const float* u, const float* dq, const float* dq_scale, float* dw, float* db, float* part_ws, int B, int S,
                           vxb_stream_t stream);
size_t vxb_conv3_c1_wgrad_mfma_blocks(int B, int S);

/* The same two layers fused with the statistics / backward of the pooled features of THEIR OUTPUT (perceiver_lang_io.py:357
 * + :360, network_utils.py:773-809): the forward stores y and folds it into the SpatialSoftmax3D / max partials while it is in
 * registers (bit-identical to vxb_pointwise_fwd_f32 + vxb_ss3d_max_fwd_f32, one pass less over 256 B per voxel); the
 * weight gradient adds the term vxb_ss3d_max_bwd_f32 would have written into dy on the fly, and (fold_src != NULL) the
 * padding adjoint vxb_fold_pad_f32 would have added to dy from a [B, Sp^3, 64] data gradient.  x [B,S,S,S,Cin], y / dy
 * [B,S,S,S,64] (dy may be NULL: only the pooled-feature and fold_src terms, when every conv path into y has added its share of
 * dW / db itself: vxb_conv3_dgrad_fold_f16_wgin_f32, vxb_patch_dgrad_input_wgrad_f32); part_ws as for vxb_ss3d_max_fwd_f32 (C = 64), resp. B * ceil(S^3 / 4096) * (64*Cin + 64) floats. */
int vxb_pointwise_ss3d_fwd_f32(const float* x, const float* W, const float* bias, float* y, int B, int S, int Cin, int Cout,
                               float slope, const float* lin, float* part_ws, float* out_ss, float* out_max, float* stats,
                               int32_t* argmax, vxb_stream_t stream);
int vxb_pointwise_wgrad_ss3d_f32(const float* x, const float* y, const float* dy, float* dW, float* db, float* part_ws, int B,
                                 int S, int Cin, int Cout, float slope, const float* lin, const float* stats,
                                 const float* out_ss, const int32_t* argmax, const float* g_ss, const float* g_max,
                                 const float* fold_src, int Sp, int pad, vxb_stream_t stream);

/* Backward of everything that reads u = final(...) (perceiver_lang_io.py:462-470) in one pass over u: du = lrelu'(u) *
 * ([du if accumulate] + data gradient of trans_decoder (Cout = 1 conv, vxb_conv3_c1_dgrad_f32) + the pooled-feature term of
 * vxb_ss3d_max_bwd_f32 on u), and dbias[64] += column sums of du (the bias gradient of `final`).  Bit-identical du to the
 * separate kernels.  S % 4 == 0, C = 64; part_ws: vxb_conv3_c1_dgrad_ss3d_ws_floats(B, S) floats. */
size_t vxb_conv3_c1_dgrad_ss3d_ws_floats(int B, int S);
int vxb_conv3_c1_dgrad_ss3d_f32(const float* dq, const float* w, const float* u, float* du, int B, int S, int C,
                                int accumulate, float slope, const float* lin, const float* stats, const float* out_ss,
                                const int32_t* argmax, const float* g_ss, const float* g_max, float* dbias,
                                float* part_ws, float* du_scale, vxb_stream_t stream);

/* SpatialSoftmax3D (T=0.01, meshgrid 'xy' quirk) + AdaptiveMaxPool3d(1) in one streaming pass
 * (network_utils.py:773-809; perceiver_lang_io.py:360,:451,:470), and the backward of both.
 * part_ws: B * nchunk * C * 7 floats with nchunk = ceil(S^2 / max(1, S^2 / want)), want = max(64, ceil(1024 / B))
 * (the (d, h) rows of a sample are cut into >= 1024 / B chunks so that small batches still fill the chip). */
int vxb_ss3d_max_fwd_f32(const float* x, int64_t bs, int B, int S, int C, const float* lin, float* part_ws,
                         float* out_ss, float* out_max, float* stats, int32_t* argmax, vxb_stream_t stream);
int vxb_ss3d_max_bwd_f32(const float* x, int64_t bs, int B, int S, int C, const float* lin, const float* stats,
                         const float* out_ss, const int32_t* argmax, const float* g_ss, const float* g_max,
                         float* dx, int64_t dbs, int accumulate, vxb_stream_t stream);

/* The patchify data gradient folded into the input conv's weight gradient (perceiver_lang_io.py:357-371: patchify = Conv3DBlock(64 -> 64,
 * k = stride, replicate padding) on d0 = input_preprocess(voxel grid); the grid is a detached input, agent :100, so d(d0) only feeds
 * dW_in / db_in): for every (patch p, tap t), v = clamp(k p + t - pad) per axis,
 *   dW[c][j] += lrelu'(d0[v][c]) * (sum_kout dpatch[p][kout] * Wp[kout][c][t]) * vox[v][j],   db[c] += the same without vox,
 * i.e. the padding adjoint and the data gradient tensor (4.7 GB at configs[1]) never exist.  Single fp16 products (a leaf): dpatch is
 * multiplied by scale[0] (device, power of two: vxb_absmax_scale_f32), the sums by scale[1].  wt_f16: fp16 [k^3][64][64] =
 * Wp[kout][c][t] transposed per tap (t = (kd k + kh) k + kw).  dW [64][10], db [64] ACCUMULATED; ws: ..._ws_floats(k, nsplit) floats.
 * dWp (optional; with ws_wp of vxb_patch_wgrad_weight_ws_floats(k, nsplit) floats, both or neither): the patchify conv's own WEIGHT
 * gradient (network_utils.py:128-170 backward), dWp[kout][c][t] += sum_p d0[v(p, t)][c] * dpatch[p][kout] in the parameter's layout --
 * from the d0 values the same launch fetches for lrelu', instead of a second pass over d0. */
size_t vxb_patch_dgrad_input_wgrad_ws_floats(int k, int nsplit);
size_t vxb_patch_wgrad_weight_ws_floats(int k, int nsplit);
int vxb_patch_dgrad_input_wgrad_f32(const float* dpatch, const void* wt_f16, const float* d0, const float* vox, int B, int V, int G,
                                    int k, int pad, float slope, const float* scale, float* ws, int nsplit, float* dW, float* db,
                                    float* dWp, float* ws_wp, vxb_stream_t stream);

/* PreNorm LayerNorm (perceiver_lang_io.py:56-71), eps 1e-5.  bwd: dgamma/dbeta ACCUMULATED; part_ws: ceil(rows / 32) * 2 * D floats. */
int vxb_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                          float* rstd, int64_t rows, int D, float eps, vxb_stream_t stream);
int vxb_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                          const float* rstd, float* dx, float* dgamma, float* dbeta, float* part_ws,
                          int64_t rows, int D, int accumulate_dx, vxb_stream_t stream);

/* attention softmax + dropout (perceiver_lang_io.py:124-128), counter-based keep mask (seed,row,col). */
int vxb_softmax_rows_f32(float* S, float* P_drop, int64_t rows, int cols, int64_t ld, float dropout_p,
                         uint32_t seed, vxb_stream_t stream);
/* In-place softmax of a few very long rows (the V^3-wide translation softmax of act(),
 * qattention_peract_bc_agent.py:705-707): rows are cut into 8192-column chunks over many workgroups.
 * ws: rows * ceil(cols / 8192) * 2 floats of scratch. */
int vxb_softmax_long_rows_f32(float* S, float* ws, int64_t rows, int cols, int64_t ld, vxb_stream_t stream);
int vxb_softmax_bwd_rows_f32(const float* P, float* dP_inout, int64_t rows, int cols, int64_t ld, float scale,
                             float dropout_p, uint32_t seed, vxb_stream_t stream);

/* Fused attention on the bf16 matrix cores (throughput mode): O = dropout(softmax(scale q k^T)) v per (b, h) without
 * materialising the score tensor (perceiver_lang_io.py:116-130).  q [B,Nq,H*64], kv [B,Nk,2*H*64] (k | v halves), fp32;
 * lse [B*H, Nq] = log-sum-exp of the scaled scores (kept for the backward pass).  head_dim must be 64. */
int vxb_flash_attn_fwd_bf16(const float* q, const float* kv, float* o, float* lse, int B, int H, int Nq, int Nk,
                            int head_dim, float scale, float dropout_p, uint32_t seed, vxb_stream_t stream);

int vxb_flash_attn_bwd_bf16(const float* q, const float* kv, const float* o, const float* d_o, const float* lse,
                            float* dq, float* dkv, float* dsum_ws, int B, int H, int Nq, int Nk, int head_dim,
                            float scale, float dropout_p, uint32_t seed, vxb_stream_t stream);
/* 'bf16x3' twins: q, k, v, dO and the probabilities / score gradients are all carried as hi + lo bf16 halves, three
 * MFMAs per product -- the fused kernels then stay inside the 1e-4 Q-value bound of the exact-fp32 attention path. */
int vxb_flash_attn_fwd_bf16x3(const float* q, const float* kv, float* o, float* lse, int B, int H, int Nq, int Nk,
                              int head_dim, float scale, float dropout_p, uint32_t seed, vxb_stream_t stream);
int vxb_flash_attn_bwd_bf16x3(const float* q, const float* kv, const float* o, const float* d_o, const float* lse,
                              float* dq, float* dkv, float* dsum_ws, int B, int H, int Nq, int Nk, int head_dim,
                              float scale, float dropout_p, uint32_t seed, vxb_stream_t stream);

/* fp16 twin of vxb_split_bf16_f32: plane 0 = RNE(x) saturated at +-65504, plane 1 (nplanes = 2) = RNE(x - plane 0). */
int vxb_split_f16_f32(const float* src, int64_t ld, int64_t rows, int cols, void* dst_planes, int nplanes, vxb_stream_t stream);

/* Attention.forward (perceiver_lang_io.py:107-132: sim = q k^T * scale, softmax, dropout, attn v), pipelined structure of round 4
 * (csrc/flash2_fwd.hip): per wave the scores of tile j+1, the exponentials of tile j and the P V product of tile j-1 are independent
 * instruction streams of one region; the row offset is subtracted inside the matrix product (no per-tile maximum / rescale).
 * mode 0: kv_planes = one bf16 plane [B*Nk][2*H*64] ('bf16'); 1: one fp16 plane ('f16'); one MFMA per product.  mode 2 ('bf16x3', the
 * default precision's forward): kv_planes = the hi | lo bf16 planes [2][B*Nk][2*H*64], three MFMAs per product, unit-pipelined kernel.
 * waves = 4 or 8 per workgroup, 0 = by grid size (mode 2: always 8).
 * Outputs and dropout mask as vxb_flash_attn_fwd_dl (o [B,Nq,H*64], lse [B*H,Nq], natural log). */
int vxb_flash2_attn_fwd(const float* q, const void* kv_planes, int mode, float* o, float* lse, int B, int H, int Nq, int Nk,
                        int head_dim, float scale, float dropout_p, uint32_t seed, int waves, vxb_stream_t stream);

/* Backward of Attention.forward (autograd of perceiver_lang_io.py:107-132), pipelined structure (csrc/flash2_bwd.hip): per wave the
 * scores / dP of unit s+1, the score gradients of unit s and the dQ (or dK | dV) products of unit s-1 overlap; -lse and -D are folded
 * into the matrix products.  kv_plane: the 16-bit plane of k | v the forward used (mode 0 bf16 / 1 fp16: vxb_split_bf16_f32 /
 * vxb_split_f16_f32, one plane); gx = 1: dO and dS as hi + lo pairs (two MFMAs per product).  dO is scaled by a power of two taken from
 * its largest magnitude on the device (fp16 range) and the results are scaled back.  which: 1 = dq, 2 = dkv, 3 = both (written, not
 * accumulated).  ws: vxb_flash2_attn_bwd_ws_bytes(B, H, Nq, gx) bytes, 256-byte aligned.  Same dropout mask as the forward. */
size_t vxb_flash2_attn_bwd_ws_bytes(int B, int H, int Nq, int gx);
int vxb_flash2_attn_bwd(const float* q, const float* kv, const float* o, const float* d_o, const float* lse, const void* kv_plane,
                        int mode, int gx, float* dq, float* dkv, void* ws, int B, int H, int Nq, int Nk, int head_dim, float scale,
                        float dropout_p, uint32_t seed, int which, vxb_stream_t stream);

/* Round 6: the dropout mask as DATA.  The forward's threshold compares leave one 64-bit lane mask per score register in an SGPR pair; with
 * dropout_p > 0 vxb_flash2_attn_fwd_mask also stores those pairs (scalar stores, no vector-ALU work) as "keep words" -- bit q of word
 * [b * H + h][32-row block][64-key tile][kb][r][half] = query row 32 block + q keeps key 64 tile + 32 kb + (r & 3) + 8 (r >> 2) + 4 half --
 * and vxb_flash2_attn_bwd_mask reads them (through LDS with the tile loads; one bit test per score in both kernels)
 * instead of hashing (seed, row, key) again (9 - 15 vector instructions per score pair less in the backward).  Because the mask is handed
 * on as data, this forward does not evaluate the (seed, row, key) hash of the entries above either: it draws the mask from a per-row 24-bit
 * linear congruential sequence seeded by (seed, row) -- one full-rate instruction per score pair, same Bernoulli(1 - p) statistics, same
 * mask for the same (seed, shape) -- so its backward MUST be vxb_flash2_attn_bwd_mask with the words it wrote.
 * drop_mask: vxb_flash2_drop_mask_bytes(B, H, Nq, Nk) bytes, 256-byte aligned, written by the forward call and read by the backward call of
 * the same (B, H, Nq, Nk).  modes 0 / 1 only.  Everything else as vxb_flash2_attn_fwd / vxb_flash2_attn_bwd (Attention.forward's dropout,
 * perceiver_lang_io.py:124-128). */
size_t vxb_flash2_drop_mask_bytes(int B, int H, int Nq, int Nk);
int vxb_flash2_attn_fwd_mask(const float* q, const void* kv_planes, int mode, float* o, float* lse, void* drop_mask, int B, int H, int Nq, int Nk,
                             int head_dim, float scale, float dropout_p, uint32_t seed, int waves, vxb_stream_t stream);
int vxb_flash2_attn_bwd_mask(const float* q, const float* kv, const float* o, const float* d_o, const float* lse, const void* kv_plane,
                             const void* drop_mask, int mode, int gx, float* dq, float* dkv, void* ws, int B, int H, int Nq, int Nk, int head_dim,
                             float scale, float dropout_p, uint32_t seed, int which, vxb_stream_t stream);

/* Forward with k | v as bf16 planes [nplanes][B*Nk][2*H*64] (vxb_split_bf16_f32 of the to_kv output): K/V tiles go
 * global -> LDS directly (double-buffered, one barrier per 64-key tile).  Same outputs / dropout mask as the entries above. */
int vxb_flash_attn_fwd_dl(const float* q, const void* kv_planes, int nplanes, float* o, float* lse, int B, int H, int Nq,
                          int Nk, int head_dim, float scale, float dropout_p, uint32_t seed, vxb_stream_t stream);

/* Backward twin: q, dO and k | v tiles from bf16 planes (kv_planes as above; q_planes / do_planes [nplanes][B*Nq][H*64]),
 * one LDS copy per tile serving row and transposed fragment reads.  dq / dkv WRITTEN; dsum_ws: B*H*Nq floats. */
int vxb_flash_attn_bwd_dl(const float* q, const float* kv, const float* o, const float* d_o, const float* lse,
                          const void* kv_planes, const void* q_planes, const void* do_planes, int nplanes, float* dq,
                          float* dkv, float* dsum_ws, int B, int H, int Nq, int Nk, int head_dim, float scale,
                          float dropout_p, uint32_t seed, vxb_stream_t stream);

/* GEGLU x * gelu_erf(gates) (perceiver_lang_io.py:74-77); LeakyReLU backward; y += alpha*x. */
int vxb_geglu_fwd_f32(const float* h, float* out, int64_t rows, int F, vxb_stream_t stream);
int vxb_geglu_bwd_f32(const float* h, const float* dout, float* dh, int64_t rows, int F, vxb_stream_t stream);
int vxb_lrelu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, float slope, vxb_stream_t stream);
int vxb_axpy_f32(float* dst, const float* src, int64_t n, float alpha, vxb_stream_t stream);

/* deterministic reductions: dst (+)= alpha * sum_s part[s][:];  out (+)= column sums of x[rows,N]. */
int vxb_sum_splits_f32(const float* part, int nsplit, int64_t n, float* dst, int accumulate, float alpha,
                       vxb_stream_t stream);
int vxb_colsum_f32(const float* x, int64_t rows, int N, int64_t ld, float* part_ws, float* out, int accumulate,
                   vxb_stream_t stream);
/* the same with the factor read from device memory (*alpha): no host round trip for a scale computed on the device. */
int vxb_sum_splits_dev_f32(const float* part, int nsplit, int64_t n, float* dst, int accumulate, const float* alpha,
                           vxb_stream_t stream);
/* scale[0] = the power of two that maps max |x[i]| into [2^14, 2^15) (1 if the array is all zero or holds inf / NaN),
 * scale[1] = 1 / scale[0]: the operand scale of the fp16 single-product kernels (fp16 keeps 11 bits from 6e-5 to 65504).
 * x dense, n elements; ws: 1024 words.  Two launches, no host sync. */
int vxb_absmax_scale_f32(const float* x, int64_t n, float* ws, float* scale, vxb_stream_t stream);

/* CLIP text transformer pieces of the act() path (reference peract/helpers/clip/core/clip.py:426-440
 * encode_text_with_embeddings, :224-245 ResidualAttentionBlock, :219-221 QuickGELU): token + positional embedding (or a
 * plain row gather with pos == NULL), x * sigmoid(1.702 x) in place, and the causal multi-head self-attention of n short
 * sequences (L <= 128, heads of 64; qkv = nn.MultiheadAttention's in_proj output q | k | v). */
int vxb_embed_rows_f32(const int32_t* idx, const float* table, const float* pos, float* out, int64_t rows, int L, int D,
                       int64_t table_rows, vxb_stream_t stream);
int vxb_quick_gelu_f32(float* x, int64_t n, vxb_stream_t stream);
int vxb_attn_causal_small_f32(const float* qkv, float* out, int n, int L, int H, vxb_stream_t stream);

/* context assembly cat(lang, cat(patch, proprio)) + pos_encoding (perceiver_lang_io.py:370-422) and its adjoint.  pp is
   [B, Cp]: Cp = C for one proprio vector, 2 C for the right | left pair of the 2Robots encoder (perceiver_lang_io.py:721-727);
   the context is C + Cp wide. */
int vxb_ctx_build_f32(const float* lang, const float* patch, const float* pp, const float* pos, float* ctx, int B,
                      int T0, int T1, int C, int Cp, vxb_stream_t stream);
int vxb_ctx_bwd_f32(const float* dctx, float* dlang, float* dpatch, float* dpp, float* dpos, float* part_ws, int B,
                    int T0, int T1, int C, int Cp, vxb_stream_t stream);

/* cross-entropy (agent :517-578) + argmax (agent :57-80): one 10^6-way head, and up to 8 small heads per row. */
int vxb_ce_big_f32(const float* x, int64_t P, int B, const int32_t* label, float* part_ws, float* lse, float* loss,
                   int32_t* argmax, float* dx, float gscale, vxb_stream_t stream);
int vxb_ce_rows_f32(const float* logits, int64_t ld, int rows, int nseg, const int32_t* col0, const int32_t* ncls,
                    const int32_t* labels, float* loss, int32_t* pred, float* dlogits, float gscale,
                    vxb_stream_t stream);

/* Fused multi-tensor LAMB (peract/helpers/optim/lamb.py:94-122; no bias correction, ||w|| clamp 10).  The betas come as
 * doubles: 1 - beta is evaluated in double and then rounded to fp32, as Python evaluates `alpha=1 - beta1` upstream.
 * skip_if_negative (optional device word): when it holds a negative value the step is a no-op (weights, moments untouched):
 * the status word of vxb_se3_relabel_f32 -- upstream raises before the forward pass (augmentation.py:119-120), here the
 * poisoned step must not reach the optimizer state. */
int vxb_lamb_step_f32(float* w, const float* g, float* m, float* v, float* upd, const int32_t* chunks, int nchunks,
                      const int32_t* first, int ntensors, float* part, float* trust, float lr, double beta1, double beta2,
                      float eps, float weight_decay, const int32_t* skip_if_negative, vxb_stream_t stream);

/* 256 x 256-tile GEMM for the big linear layers (nn.Linear forward / data gradient at M = B * 2048 latent rows,
 * perceiver_lang_io.py:80-132): C[M][N] (+)= act(A[M][K] @ W[N][K]^T + bias) (+ residual).  Both operands are bf16 planes
 * (nplanes = 1: 'bf16'; 2: hi then lo, 'bf16x3'): A_planes [nplanes][M][lda], W_planes [nplanes][N][K]; K % 32 == 0,
 * lda % 8 == 0, 16-byte aligned bases. */
int vxb_gemm256_f32(const void* A_planes, int64_t lda, const void* W_planes, int nplanes, float* C, int64_t ldc, const float* bias,
                    const float* residual, int M, int N, int K, int act, float slope, int accumulate, vxb_stream_t stream);

/* torch.optim.Adam step over a flat parameter buffer (the reference's `optimizer: adam` alternative, agent :263-268):
 * L2 weight decay folded into the gradient, bias correction with `step` (1-based), eps 1e-8 by default upstream. */
int vxb_adam_step_f32(float* w, const float* g, float* m, float* v, int64_t n, float lr, double beta1, double beta2, float eps,
                      float weight_decay, int64_t step, const int32_t* skip_if_negative, vxb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VOXACTB_HIP_H */
