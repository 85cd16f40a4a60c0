// SpatialSoftmax3D (T = 0.01) + global max: the online-softmax partial of one (sample, channel) over a set of voxels and its merge
// (network_utils.py:768-800, perceiver_lang_io.py:360 / :451 / :470).  Shared by the statistics kernels (vox_ops.hip) and by the
// epilogue of the `final` conv, which takes the statistics of its output tile while it is in registers (conv_halo_bf16.hip).
#pragma once
#include "common.h"

namespace {

struct SsPart { float m, s, sx, sy, sz, xmax; int arg; };

__device__ __forceinline__ void ss_merge(SsPart& a, const SsPart& b) {
    if (b.s > 0.f || b.m > -INFINITY) {
        const float m = fmaxf(a.m, b.m);
        const float fa = a.m > -INFINITY ? expf(a.m - m) : 0.f;
        const float fb = b.m > -INFINITY ? expf(b.m - m) : 0.f;
        a.s = a.s * fa + b.s * fb;
        a.sx = a.sx * fa + b.sx * fb;
        a.sy = a.sy * fa + b.sy * fb;
        a.sz = a.sz * fa + b.sz * fb;
        a.m = m;
    }
    if (b.xmax > a.xmax || (b.xmax == a.xmax && b.arg < a.arg)) { a.xmax = b.xmax; a.arg = b.arg; }
}

}  // namespace
