// prep_job.h — the frame's camera-only preparation (get_proj_mats of every cascade level, utils.py:35-55, and level 0's
// get_depth_values, utils.py:98-111,148-150) as device code that can ride in ANOTHER kernel's launch.
//
// Level 0's depth hypotheses and all projection matrices depend on the batch's cameras and near_far only — not on a single
// feature — yet k_depth_values sat as a 5.8 us launch on the frame's critical chain between the FeatureNet trunk and the warp.
// enerf_forward now appends the job's blocks to the FIRST kernel of the frame (k_conv0_fused_cb): they run beside the
// convolution's blocks, and the launch (and with it the cross-stream bubble in front of it) leaves the chain.  The arithmetic
// is the same device code the stand-alone kernels (geometry.hip) call: identical bits either way.
#pragma once
#include "common.h"

namespace enerf {

// proj3x4 = K'(3x3, rows 0,1 scaled) * E[:3] (3x4)
__device__ __forceinline__ void k_times_e(const float* K, const float* E, float scale, double* out34) {
    for (int r = 0; r < 3; ++r) {
        double sc = r < 2 ? (double)scale : 1.0;
        for (int c = 0; c < 4; ++c) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += ((double)K[r * 3 + k] * sc) * (double)E[k * 4 + c];
            out34[r * 4 + c] = a;
        }
    }
}
// one (b, s) projection matrix (get_proj_mats utils.py:35-55), fp64 like torch.inverse's LU on these sizes
__device__ __forceinline__ void proj_one(int i, const float* __restrict__ src_ixts, const float* __restrict__ src_exts,
                                         const float* __restrict__ tar_ixt, const float* __restrict__ tar_ext, int S,
                                         float src_scale, float tar_scale, float* __restrict__ proj) {
    const int b = i / S;
    double t44[16], tinv[16], s34[12];
    k_times_e(tar_ixt + b * 9, tar_ext + b * 16, tar_scale, t44);
    t44[12] = t44[13] = t44[14] = 0.0;
    t44[15] = 1.0;
    if (!inv4x4(t44, tinv)) {
        for (int k = 0; k < 16; ++k) tinv[k] = NAN;
    }
    k_times_e(src_ixts + i * 9, src_exts + i * 16, src_scale, s34);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double a = 0;
            for (int k = 0; k < 4; ++k) a += s34[r * 4 + k] * tinv[k * 4 + c];
            proj[i * 12 + r * 4 + c] = (float)a;
        }
}

// Optional piggy-backed get_proj_mats of the same level (enerf_level_prep): the B*S fp64 inverses are a 4.7 us
// latency chain in a launch of their own; here the last block's first threads run them next to the plane writes.
struct ProjJob {
    const float *src_ixts, *src_exts, *tar_ixt, *tar_ext;
    float* proj;          // nullptr: no projection matrices in this launch
    int S;
    float src_scale, tar_scale;
};

// one element of get_depth_values' output (B, D, h, w) from the pixel's [nn, ff] (utils.py:104-111 level 0; 137-146 level > 0) and
// the level's near/far planes (utils.py:148-150)
__device__ __forceinline__ void depth_plane_value(float nn, float ff, int b, int k, int p, int D, int hw, int depth_inv,
                                                  float* __restrict__ dv_elem, float* __restrict__ nf_out) {
    const float inn = 1.f / nn, iff = 1.f / ff;
    const float t = linspace01(k, D);
    const float v = depth_inv ? 1.f / (inn + t * (iff - inn)) : nn + t * (ff - nn);
    *dv_elem = v;
    if (k == 0 || k == D - 1) {                  // utils.py:149-150 (k == 0 == D-1 writes both)
        const float e = depth_inv ? 1.f / clamp_min(v, 1e-6f) : v;
        if (k == 0) nf_out[((long long)b * 2 + 0) * hw + p] = e;
        if (k == D - 1) nf_out[((long long)b * 2 + 1) * hw + p] = e;
    }
}

struct PrepJob {
    const float* near_far;     // (B, 2); nullptr: no level-0 planes in this job
    float *dv, *nf;            // level 0: (B, D, h, w), (B, 2, h, w)
    int B, D, h, w, depth_inv;
    ProjJob pj[3];             // projection matrices of up to three cascade levels (proj == nullptr: none)
    int nblocks;               // blocks the job needs at `threads` threads each (dispatched in front of the carrying kernel's own)
};
inline int prep_job_blocks(int B, int D, int h, int w, int threads) { return (B * D * h * w + threads - 1) / threads; }
// block `vb` of the job (threads = blockDim.x of the carrying kernel)
__device__ __forceinline__ void prep_job_block(const PrepJob& J, int vb, int tid, int threads) {
    if (vb == 0) {
#pragma unroll 1
        for (int l = 0; l < 3; ++l)
            if (J.pj[l].proj != nullptr)
                for (int q = tid; q < J.B * J.pj[l].S; q += threads)
                    proj_one(q, J.pj[l].src_ixts, J.pj[l].src_exts, J.pj[l].tar_ixt, J.pj[l].tar_ext, J.pj[l].S, J.pj[l].src_scale,
                             J.pj[l].tar_scale, J.pj[l].proj);
    }
    if (J.near_far == nullptr) return;
    const int i = vb * threads + tid, hw = J.h * J.w;
    if (i >= J.B * J.D * hw) return;
    const int b = i / (J.D * hw), r = i - b * (J.D * hw);
    const int k = r / hw, p = r - k * hw;
    depth_plane_value(J.near_far[b * 2 + 0], J.near_far[b * 2 + 1], b, k, p, J.D, hw, J.depth_inv, J.dv + i, J.nf);
}

}  // namespace enerf
