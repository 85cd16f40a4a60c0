// Pose / label half of the SE(3) augmentation on the device (gfx950).
//
// Replaces the per-attempt host loop of apply_se3_augmentation (reference peract/voxel/augmentation.py:98-177): ~80 tiny
// ATen kernels, three device->host copies and B scipy / numpy calls per attempt upstream.  Here ONE workgroup evaluates up
// to K pre-drawn attempts (thread = sample, attempts in order, a workgroup-wide vote per attempt -- the reference re-draws the
// WHOLE batch while any sample's translation index is negative, :116) and leaves the labels and the rigid transform of the
// winning attempt in device memory; the point clouds are transformed inside the voxelizer's point load.
//
// Numeric types follow the reference line by line: fp32 for the pose matrices (torch), float64 for the translation shift
// (`trans_aug_range` is a float64 tensor, agent :186) and for both discretisations (numpy / scipy, helpers/utils.py:92-116).
// The three pytorch3d==0.3.0 helpers (quaternion_to_matrix, euler_angles_to_matrix, matrix_to_quaternion) are restated from
// their published definition -- upstream pins them only by version number (oracle/se3.py has the caveat).
#include "common.h"
#include <math.h>

namespace {

struct RelabelArgs {
    // arm 0 = the (right) arm whose pose is the centre of the point-cloud transform; arm 1 (optional, 2Robots baseline,
    // augmentation.py:187-348): shares every random draw, the retry vote covers both arms' translation indices (:237)
    int n_arms;
    const float* pose2;
    const int32_t* rot_grip_in2;
    int32_t* trans_idx2;
    int32_t* rot_grip_idx2;
    const float* pose;
    const int32_t* rot_grip_in;
    const float* bounds;
    int bounds_rows, layer;
    const float* shift_unit;
    const int32_t* rpy_steps;
    int K, B;
    double aug[3];
    float rot_aug_resolution;
    int V;
    float rot_resolution;
    int32_t* trans_idx;
    int32_t* rot_grip_idx;
    float* xf;
    int32_t* status;
};

__device__ __forceinline__ void mat3_mul(const float (&a)[9], const float (&b)[9], float (&c)[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[i * 3 + j] = __fadd_rn(__fadd_rn(__fmul_rn(a[i * 3], b[j]), __fmul_rn(a[i * 3 + 1], b[3 + j])),
                                     __fmul_rn(a[i * 3 + 2], b[6 + j]));
}

// helpers/utils.py:92-97 on a unit quaternion (x, y, z, w), float64: extrinsic x-y-z Euler angles in degrees + 180,
// divided by the resolution, rounded half-to-even (np.around), the top bin folded onto 0.
__device__ __forceinline__ void discrete_euler(double x, double y, double z, double w, double resolution, int (&disc)[3]) {
    const double nrm = sqrt(x * x + y * y + z * z + w * w);          // Rotation.from_quat normalises again
    x /= nrm; y /= nrm; z /= nrm; w /= nrm;
    const double r00 = 1.0 - 2.0 * (y * y + z * z), r10 = 2.0 * (x * y + w * z);
    const double r20 = 2.0 * (x * z - w * y), r21 = 2.0 * (y * z + w * x), r22 = 1.0 - 2.0 * (x * x + y * y);
    double e[3];
    const double sb = fmin(1.0, fmax(-1.0, -r20));
    e[1] = asin(sb);
    if (fabs(sb) > 1.0 - 1e-14) {                                    // gimbal lock: scipy puts the whole twist in the first angle
        const double r01 = 2.0 * (x * y - w * z), r11 = 1.0 - 2.0 * (x * x + z * z);
        e[0] = atan2(-r01 * (sb > 0 ? -1.0 : 1.0), r11);
        e[2] = 0.0;
    } else {
        e[0] = atan2(r21, r22);
        e[2] = atan2(r10, r00);
    }
    const int top = (int)(360.0 / resolution);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double deg = e[a] * (180.0 / M_PI) + 180.0;
        int d = (int)rint(deg / resolution);
        if (d == top) d = 0;
        disc[a] = d;
    }
}

__global__ void se3_relabel_kernel(RelabelArgs A) {
    const int b = threadIdx.x;
    const bool live = b < A.B;
    // batch-wide bounds for the clamp of the new centre (augmentation.py:44-57)
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = A.bounds[a]; hi[a] = A.bounds[3 + a]; }
    for (int r = 1; r < A.bounds_rows; ++r)
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], A.bounds[r * 6 + a]); hi[a] = fmaxf(hi[a], A.bounds[r * 6 + 3 + a]); }

    float T[2][9] = {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {1, 0, 0, 0, 1, 0, 0, 0, 1}}, t[2][3] = {{0, 0, 0}, {0, 0, 0}};
    const float* bd_range = A.bounds;                               // row whose extent scales the shift (broadcast [1|B, 3])
    const float* bd_label = A.bounds;                               // row used for the translation label (:161-162)
    if (live) {
#pragma unroll
        for (int arm = 0; arm < 2; ++arm) {
            if (arm >= A.n_arms) break;
            const float* p = (arm ? A.pose2 : A.pose) + b * 7;
            t[arm][0] = p[0]; t[arm][1] = p[1]; t[arm][2] = p[2];
            const float i = p[3], j = p[4], k = p[5], r = p[6];     // pytorch3d order is (r, i, j, k) = (w, x, y, z)
            const float ss = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r, r), __fmul_rn(i, i)), __fmul_rn(j, j)), __fmul_rn(k, k));
            const float two_s = __fdiv_rn(2.0f, ss);
            float* Ta = T[arm];
            Ta[0] = 1.0f - two_s * (j * j + k * k); Ta[1] = two_s * (i * j - k * r);        Ta[2] = two_s * (i * k + j * r);
            Ta[3] = two_s * (i * j + k * r);        Ta[4] = 1.0f - two_s * (i * i + k * k); Ta[5] = two_s * (j * k - i * r);
            Ta[6] = two_s * (i * k - j * r);        Ta[7] = two_s * (j * k + i * r);        Ta[8] = 1.0f - two_s * (i * i + j * j);
        }
        if (A.bounds_rows > 1) bd_range = A.bounds + b * 6;
        if (A.bounds_rows > 1 && A.layer > 0) bd_label = A.bounds + b * 6;
    }
    const float step_rad = (float)(A.rot_aug_resolution * (M_PI / 180.0));          // np.deg2rad(...), cast by the f32 product
    for (int k = 0; k < A.K; ++k) {
        int tidx[2][3] = {{0, 0, 0}, {0, 0, 0}};
        float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, M[2][9], centre[3] = {0, 0, 0};
        bool ok = true;
        if (live) {
            const float* su = A.shift_unit + ((size_t)k * A.B + b) * 3;
            const int32_t* st = A.rpy_steps + ((size_t)k * A.B + b) * 3;
            float Rx[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Ry[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Rz[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Rxy[9];
            const float ax = (float)st[0] * step_rad, ay = (float)st[1] * step_rad, az = (float)st[2] * step_rad;
            Rx[4] = cosf(ax); Rx[5] = -sinf(ax); Rx[7] = sinf(ax); Rx[8] = cosf(ax);
            Ry[0] = cosf(ay); Ry[2] = sinf(ay); Ry[6] = -sinf(ay); Ry[8] = cosf(ay);
            Rz[0] = cosf(az); Rz[1] = -sinf(az); Rz[3] = sinf(az); Rz[4] = cosf(az);
            mat3_mul(Rx, Ry, Rxy);
            mat3_mul(Rxy, Rz, R);                                   // euler_angles_to_matrix(., "XYZ") (:142)
#pragma unroll
            for (int arm = 0; arm < 2; ++arm) {
                if (arm >= A.n_arms) break;
                mat3_mul(T[arm], R, M[arm]);                        // rotation block of bmm(T_grip, R) (:147)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double range = (double)__fsub_rn(bd_range[3 + a], bd_range[a]) * A.aug[a];    // f32 extent * f64 range (:123)
                    const double shift = range * (double)su[a];                                          // (:124)
                    const float tp = (float)((double)t[arm][a] + shift);                                 // in-place += (:148)
                    if (arm == 0) centre[a] = fminf(fmaxf(__fadd_rn(t[0][a], (float)shift), lo[a]), hi[a]);   // (:49-57)
                    // point_to_voxel_index (helpers/utils.py:104-116): float64, clipped from above only
                    const double res = (double)__fsub_rn(bd_label[3 + a], bd_label[a]) / ((double)A.V + 1e-12);
                    const double q = floor((double)__fsub_rn(tp, bd_label[a]) / (res + 1e-12));
                    int iv = (q >= -2147483648.0 && q <= 2147483647.0) ? (int)q : INT32_MIN;             // NaN / overflow -> invalid
                    iv = iv < A.V - 1 ? iv : A.V - 1;
                    tidx[arm][a] = iv;
                    ok = ok && iv >= 0;
                }
            }
        }
        const int all_ok = __syncthreads_and(ok ? 1 : 0);
        if (all_ok) {
            if (live) {
#pragma unroll
                for (int arm = 0; arm < 2; ++arm) {
                    if (arm >= A.n_arms) break;
                    const float* Ma = M[arm];
                    // matrix_to_quaternion (pytorch3d 0.3.0), fp32, (w, x, y, z)
                    const float m00 = Ma[0], m11 = Ma[4], m22 = Ma[8];
                    const float qw = 0.5f * sqrtf(fmaxf(0.0f, 1.0f + m00 + m11 + m22));
                    float qx = 0.5f * sqrtf(fmaxf(0.0f, 1.0f + m00 - m11 - m22));
                    float qy = 0.5f * sqrtf(fmaxf(0.0f, 1.0f - m00 + m11 - m22));
                    float qz = 0.5f * sqrtf(fmaxf(0.0f, 1.0f - m00 - m11 + m22));
                    qx = copysignf(qx, Ma[7] - Ma[5]);
                    qy = copysignf(qy, Ma[2] - Ma[6]);
                    qz = copysignf(qz, Ma[3] - Ma[1]);
                    // normalize_quaternion (utils.py:63-64) in fp32, then force w >= 0 (:167-171)
                    const float nrm = sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
                    float x = qx / nrm, y = qy / nrm, z = qz / nrm, w = qw / nrm;
                    if (w < 0.0f) { x = -x; y = -y; z = -z; w = -w; }
                    int disc[3];
                    discrete_euler((double)x, (double)y, (double)z, (double)w, (double)A.rot_resolution, disc);
                    int32_t* ti = arm ? A.trans_idx2 : A.trans_idx;
                    int32_t* ri = arm ? A.rot_grip_idx2 : A.rot_grip_idx;
                    const int32_t* rin = arm ? A.rot_grip_in2 : A.rot_grip_in;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        ti[b * 3 + a] = tidx[arm][a];
                        ri[b * 4 + a] = disc[a];
                    }
                    ri[b * 4 + 3] = rin[b * 4 + 3];                 // the gripper bit is carried over (:172)
                }
                float* x15 = A.xf + b * 15;
#pragma unroll
                for (int a = 0; a < 9; ++a) x15[a] = R[a];
#pragma unroll
                for (int a = 0; a < 3; ++a) { x15[9 + a] = t[0][a]; x15[12 + a] = centre[a]; }
            }
            if (b == 0) A.status[0] = k;
            return;
        }
    }
    // no attempt kept the whole batch inside the bounds: poison the labels (the CE kernels turn them into NaN losses) and
    // flag it; the caller raises 'Failing to perturb action and keep it within bounds.' (:119-120)
    if (live) {
#pragma unroll
        for (int a = 0; a < 3; ++a) A.trans_idx[b * 3 + a] = -1;
#pragma unroll
        for (int a = 0; a < 4; ++a) A.rot_grip_idx[b * 4 + a] = -1;
        if (A.n_arms > 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) A.trans_idx2[b * 3 + a] = -1;
#pragma unroll
            for (int a = 0; a < 4; ++a) A.rot_grip_idx2[b * 4 + a] = -1;
        }
        float* x15 = A.xf + b * 15;
        for (int a = 0; a < 15; ++a) x15[a] = (a == 0 || a == 4 || a == 8) ? 1.0f : 0.0f;
    }
    if (b == 0) A.status[0] = -1;
}

}  // namespace

static int se3_relabel_launch(int n_arms, const float* pose, const int32_t* rot_grip_in, const float* pose2, const int32_t* rot_grip_in2,
                              const float* bounds, int bounds_rows, int layer, const float* shift_unit, const int32_t* rpy_steps, int K,
                              int B, double aug_x, double aug_y, double aug_z, float rot_aug_resolution, int V, float rot_resolution,
                              int32_t* trans_idx, int32_t* rot_grip_idx, int32_t* trans_idx2, int32_t* rot_grip_idx2, float* xf,
                              int32_t* status, vxb_stream_t stream) {
    if (!pose || !rot_grip_in || !bounds || !shift_unit || !rpy_steps || !trans_idx || !rot_grip_idx || !xf || !status) return VXB_EARG;
    if (n_arms == 2 && (!pose2 || !rot_grip_in2 || !trans_idx2 || !rot_grip_idx2)) return VXB_EARG;
    if (K < 1 || B < 1 || V < 1 || (bounds_rows != 1 && bounds_rows != B) || !(rot_resolution > 0.f)) return VXB_EARG;
    if (B > 1024) return VXB_ESIZE;
    RelabelArgs A;
    A.n_arms = n_arms; A.pose2 = pose2; A.rot_grip_in2 = rot_grip_in2; A.trans_idx2 = trans_idx2; A.rot_grip_idx2 = rot_grip_idx2;
    A.pose = pose; A.rot_grip_in = rot_grip_in; A.bounds = bounds; A.bounds_rows = bounds_rows; A.layer = layer;
    A.shift_unit = shift_unit; A.rpy_steps = rpy_steps; A.K = K; A.B = B;
    A.aug[0] = aug_x; A.aug[1] = aug_y; A.aug[2] = aug_z;
    A.rot_aug_resolution = rot_aug_resolution; A.V = V; A.rot_resolution = rot_resolution;
    A.trans_idx = trans_idx; A.rot_grip_idx = rot_grip_idx; A.xf = xf; A.status = status;
    const int threads = (B + 63) & ~63;
    hipLaunchKernelGGL(se3_relabel_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, A);
    VXB_CHECK_LAUNCH();
    return VXB_OK;
}

extern "C" int vxb_se3_relabel_f32(const float* pose, const int32_t* rot_grip_in, const float* bounds, int bounds_rows, int layer,
                                   const float* shift_unit, const int32_t* rpy_steps, int K, int B, double aug_x, double aug_y,
                                   double aug_z, float rot_aug_resolution, int V, float rot_resolution, int32_t* trans_idx,
                                   int32_t* rot_grip_idx, float* xf, int32_t* status, vxb_stream_t stream) {
    return se3_relabel_launch(1, pose, rot_grip_in, nullptr, nullptr, bounds, bounds_rows, layer, shift_unit, rpy_steps, K, B, aug_x, aug_y,
                              aug_z, rot_aug_resolution, V, rot_resolution, trans_idx, rot_grip_idx, nullptr, nullptr, xf, status, stream);
}

// Two arms under ONE perturbation (apply_se3_augmentation_2Robots, augmentation.py:187-348): same draws for both poses, the
// attempt is kept only when BOTH arms' translation indices stay inside the grid (:237), and the point-cloud transform xf is
// centred on the RIGHT arm's pose (:346).
extern "C" int vxb_se3_relabel_pair_f32(const float* pose_right, const int32_t* rot_grip_right, const float* pose_left,
                                        const int32_t* rot_grip_left, const float* bounds, int bounds_rows, int layer,
                                        const float* shift_unit, const int32_t* rpy_steps, int K, int B, double aug_x, double aug_y,
                                        double aug_z, float rot_aug_resolution, int V, float rot_resolution, int32_t* trans_idx_right,
                                        int32_t* rot_grip_idx_right, int32_t* trans_idx_left, int32_t* rot_grip_idx_left, float* xf,
                                        int32_t* status, vxb_stream_t stream) {
    return se3_relabel_launch(2, pose_right, rot_grip_right, pose_left, rot_grip_left, bounds, bounds_rows, layer, shift_unit, rpy_steps,
                              K, B, aug_x, aug_y, aug_z, rot_aug_resolution, V, rot_resolution, trans_idx_right, rot_grip_idx_right,
                              trans_idx_left, rot_grip_idx_left, xf, status, stream);
}
