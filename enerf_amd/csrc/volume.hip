// volume.hip — homography warp + variance aggregation (homo_warp utils.py:57-95,
// build_feature_volume utils.py:322-349) fused into one gather kernel.
//
// Layout: source features channels-last (B,S,Hs,Ws,C); output cost volume channels-last
// (B,D,h,w,C).  C/4 consecutive lanes own one voxel (one float4 of channels each), so every bilinear
// tap of a voxel is a single contiguous C*4-byte segment and the store of 64/(C/4) neighbouring
// voxels is one contiguous run.  The per-view warped features are never materialised: Σx and Σx² stay in
// registers (the reference writes 2 + S volumes).
//
// Algorithmic HBM bytes per launch: read S*Hs*Ws*C*4 (features, once) + D*h*w*4 (depth planes),
// write D*h*w*C*4.  The 4-tap gathers re-read features from L2/MALL, not HBM.
#include "kernels.h"


namespace enerf {

// The CQ lanes of a voxel share its geometry, so they also share the work: lane q projects the voxel into
// view s0+q (homography, perspective divide, bilinear taps: ~130 VALU with the IEEE divides the reference
// rounding needs) and the group then walks the views, each lane fetching the taps of the view in turn from
// its owner with one lane broadcast per value.  Before, every lane projected every view — 4x (level 1) and
// 8x (level 0) redundant VALU in a kernel that is VALU- not HBM-bound (48+37 us against a 16 us HBM floor).
// The group's broadcasts: DPP register permutes (default) or, for A/B, the ds_bpermute round trips of rounds 1-3.
// CONVERGENCE: group_bcast_i runs with bound_ctrl = true, i.e. an INACTIVE source lane yields 0 — all CQ lanes of a group must reach
// every broadcast together.  They do: dead lanes (beyond the volume) shadow a live voxel instead of exiting (see `live` below), and
// no broadcast sits under a lane-divergent branch.  (GPU check of the primitives: tools/micro/dpp_primitives.hip.)
#define ENERF_VOL_BCAST_I(v, k) group_bcast_i<CQ>((v), (k))
#define ENERF_VOL_BCAST_F(v, k) group_bcast_f<CQ>((v), (k))
template <int CQ>  // CQ = C/4 lanes per voxel
__global__ __launch_bounds__(256) void k_feature_volume(const float* __restrict__ feat, const float* __restrict__ proj,
                                                        const float* __restrict__ dv, int B, int S, int Hs, int Ws,
                                                        int D, int h, int w, float inv_w, float inv_d, int planar,
                                                        float* __restrict__ vol) {
    constexpr int C = CQ * 4;
    // Block -> voxels, XCD-aware: the dispatcher puts block i on XCD i % 8 and every XCD has a private 4 MiB L2, so
    // with the raster order every XCD gathers from the whole of every source image (PMC: 245 MB of fabric reads
    // per level-1 launch against 58 MB compulsory — at 7 TB/s that IS the kernel time).  Here XCD k owns the band
    // of rows [k*rb, (k+1)*rb) of every depth plane: its gathers stay inside a band of each source image that
    // fits its L2.  Within the band blocks walk plane-major.  Speed only; any placement is correct.
    // (launcher: B*D*h*w*CQ < 2^31, D*h*w < 2^24)
    // Round 4: the decomposition is carried by the GRID (x = XCD band, y = 256/CQ-voxel chunk of the band, z = depth plane of
    // a batch element), so nothing is divided per lane or per block: the round-3 kernel spent a third of its ~400 VALU
    // instructions (two emulated integer divisions + two carry loops) on turning a linear block id into (b, d, y, x), in a
    // kernel that is VALU-issue bound (40 waves per SIMD x ~400 VALU x 4 cycles = its 27 us).
    const int xcd = blockIdx.x;                                    // gridDim.x == 8: linear block id % 8 == blockIdx.x
    const int rb = (h + 7) >> 3;                                   // rows per band
    const int yb = xcd * rb, ny = min(rb, h - yb);
    if (ny <= 0) return;                                           // uniform
    const unsigned band = (unsigned)(ny * w);                      // voxels of the band in one plane
    const unsigned p0 = blockIdx.y * (256 / CQ);                   // uniform
    if (p0 >= band) return;                                        // uniform
    const int cq = threadIdx.x & (CQ - 1);
    const unsigned p_raw = p0 + (threadIdx.x / CQ);
    const bool live = p_raw < band;
    const unsigned p = live ? p_raw : band - 1;            // dead lanes shadow the last voxel (they take part in the broadcasts)
    const int lane = threadIdx.x & 63, lead = lane & ~(CQ - 1);
    const unsigned pl = blockIdx.z;                                // plane index b * D + d
    const int b = (int)(((float)pl + 0.5f) * inv_d);               // uniform; exact for B * D < 2^22
    // row = p / w through a float reciprocal with an exact fix-up
    int yr = (int)((float)p * inv_w), x = (int)p - mul24(yr, w);
    if (x < 0) { --yr; x += w; }
    if (x >= w) { ++yr; x -= w; }
    const int y = yb + yr;
    const unsigned vox = (unsigned)mul24((int)(pl * (unsigned)h) + y, w) + (unsigned)x;
    const float depth = dv[vox];                      // (B,D,h,w) has the same linear index as the voxel
    const float fx = (float)x, fy = (float)y;
    // utils.py:82-83 divide by the Python scalars (W_S-1)/2, (H_S-1)/2: ATen's GPU kernel multiplies by the
    // reciprocal for a scalar divisor (BinaryDivTrueKernel), so this is the reference's device arithmetic
    const float inv_half_w = 1.f / (float)((Ws - 1) / 2.0), inv_half_h = 1.f / (float)((Hs - 1) / 2.0);
    const unsigned img = (unsigned)(Hs * Ws * C);      // floats per source view (launcher: B*S*img < 2^32)
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < S; s0 += CQ) {
        // ---- this lane's view ----
        const int sv = min(s0 + cq, S - 1);
        const float* P = proj + (b * S + sv) * 12;
        // IEEE divisions, as the reference (utils.py:72,80-83).  Measured in round 3: two refined reciprocals instead (<= 1.5 ulp
        // off) save 0.7 us per launch but move the ill-conditioned BatchNorm-weight gradients of conv0 by 6e-3 relative in
        // the training path, which shares this arithmetic — not worth it.
#if defined(ENERF_VOL_FASTDIV) && ENERF_VOL_FASTDIV && !defined(ENERF_EMU)   /* A/B only: what the five IEEE divisions cost (1-ulp reciprocals) */
        const float rdep = __builtin_amdgcn_rcpf(depth);
        const float px = P[0] * fx + P[1] * fy + P[2] + P[3] * rdep;
        const float py = P[4] * fx + P[5] * fy + P[6] + P[7] * rdep;
        const float pz = P[8] * fx + P[9] * fy + P[10] + P[11] * rdep;
        const float z = clamp_min(pz, 1e-6f);
        const float rz = __builtin_amdgcn_rcpf(z);
        const float gx = (px * rz) * inv_half_w - 1.f, gy = (py * rz) * inv_half_h - 1.f;
#else
        const float px = P[0] * fx + P[1] * fy + P[2] + P[3] / depth;       // utils.py:72
        const float py = P[4] * fx + P[5] * fy + P[6] + P[7] / depth;
        const float pz = P[8] * fx + P[9] * fy + P[10] + P[11] / depth;
        const float z = clamp_min(pz, 1e-6f);                                // utils.py:80
        const float gx = (px / z) * inv_half_w - 1.f, gy = (py / z) * inv_half_h - 1.f;
#endif
        const Taps2 t = gs_taps2<false>(gs_unnorm(gx, Ws), gs_unnorm(gy, Hs), Ws, Hs);
        const unsigned vb = (unsigned)(b * S + sv) * img;
        const int r0 = mul24(t.y0, Ws), r1 = mul24(t.y1, Ws);
        const int my_o[4] = {(int)(vb + (unsigned)mul24(r0 + t.x0, C)), (int)(vb + (unsigned)mul24(r0 + t.x1, C)),
                             (int)(vb + (unsigned)mul24(r1 + t.x0, C)), (int)(vb + (unsigned)mul24(r1 + t.x1, C))};
        const float my_w[4] = {t.w00, t.w01, t.w10, t.w11};
        // ---- the group's views in turn ----
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            if (s0 + k >= S) break;                                          // uniform
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned o = (unsigned)ENERF_VOL_BCAST_I(my_o[c], k) + (unsigned)(cq * 4);
                const float wgt = ENERF_VOL_BCAST_F(my_w[c], k);
                const float4 v = *reinterpret_cast<const float4*>(feat + o);
                if (c == 0) { r.x = v.x * wgt; r.y = v.y * wgt; r.z = v.z * wgt; r.w = v.w * wgt; }
                else { r.x += v.x * wgt; r.y += v.y * wgt; r.z += v.z * wgt; r.w += v.w * wgt; }
            }
            s1.x += r.x; s1.y += r.y; s1.z += r.z; s1.w += r.w;
            s2.x += r.x * r.x; s2.y += r.y * r.y; s2.z += r.z * r.z; s2.w += r.w * r.w;
        }
    }
    // utils.py:345 `div_(S)`: with a Python scalar divisor ATen's GPU kernel multiplies by the reciprocal
    // (BinaryDivTrueKernel: is_cpu_scalar -> MulFunctor(1/b)), so this is the reference's device arithmetic;
    // the CPU oracle divides, which differs by <= 1 ulp of the mean.
    const float inv_s = 1.f / (float)S;
    float4 o;
    float m;
    m = s1.x * inv_s; o.x = s2.x * inv_s - m * m;
    m = s1.y * inv_s; o.y = s2.y * inv_s - m * m;
    m = s1.z * inv_s; o.z = s2.z * inv_s - m * m;
    m = s1.w * inv_s; o.w = s2.w * inv_s - m * m;
    // planar: channel-quad planes (B, C/4, D, h, w, 4) — what the asynchronously staged conv0 reads (one quad per pass: a
    // channels-last voxel would give it 16 useful bytes of every 64-128-byte line, re-fetched once per pass)
    const unsigned nvp = (unsigned)(D * h * w);
    const long long oidx = planar ? ((long long)((unsigned)b * CQ + cq) * nvp + (vox - (unsigned)b * nvp)) * 4 : (long long)vox * C + cq * 4;
    if (live) *reinterpret_cast<float4*>(vol + oidx) = o;
}

// Two depth planes per wave (the default when D is even): the same pixel at two consecutive hypotheses — one set of projection
// matrices, two independent chains `depth -> projection -> 4 gathers per view -> moments`, and the gathers of BOTH planes of a
// view are requested before either is consumed (eight 16-byte loads in flight per lane instead of four).  A wave of the
// one-plane form above lives for two dependent memory round trips and there are ~40 of them per SIMD: neither its VALU count
// (round 4: -30 % VALU, < 3 % time) nor its lane broadcasts (DPP instead of ds_bpermute: nothing) is what it waits for.
// Same arithmetic per voxel, bit-identical output.
template <int CQ>
__global__ __launch_bounds__(256) void k_feature_volume_mp(const float* __restrict__ feat, const float* __restrict__ proj,
                                                           const float* __restrict__ dv, int B, int S, int Hs, int Ws,
                                                           int D, int h, int w, float inv_w, float inv_d, int planar,
                                                           float* __restrict__ vol) {
    constexpr int C = CQ * 4, NPL = 2;
    const int xcd = blockIdx.x;                                    // gridDim.x == 8: one row band per XCD (see above)
    const int rb = (h + 7) >> 3;
    const int yb = xcd * rb, ny = min(rb, h - yb);
    if (ny <= 0) return;                                           // uniform
    const unsigned band = (unsigned)(ny * w);
    const unsigned p0 = blockIdx.y * (256 / CQ);
    if (p0 >= band) return;                                        // uniform
    const int cq = threadIdx.x & (CQ - 1);
    const unsigned p_raw = p0 + (threadIdx.x / CQ);
    const bool live = p_raw < band;
    const unsigned p = live ? p_raw : band - 1;                    // dead lanes shadow the last voxel (they take part in the broadcasts)
    const unsigned pl0 = blockIdx.z * NPL;                         // plane index b * D + d of the first plane (launcher: D even)
    const int b = (int)(((float)pl0 + 0.5f) * inv_d);              // uniform; exact for B * D < 2^22
    int yr = (int)((float)p * inv_w), x = (int)p - mul24(yr, w);
    if (x < 0) { --yr; x += w; }
    if (x >= w) { ++yr; x -= w; }
    const int y = yb + yr;
    unsigned vox[NPL];
    float depth[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        vox[q] = (unsigned)mul24((int)((pl0 + q) * (unsigned)h) + y, w) + (unsigned)x;
        depth[q] = dv[vox[q]];
    }
    const float fx = (float)x, fy = (float)y;
    const float inv_half_w = 1.f / (float)((Ws - 1) / 2.0), inv_half_h = 1.f / (float)((Hs - 1) / 2.0);
    const unsigned img = (unsigned)(Hs * Ws * C);
    float4 s1[NPL], s2[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) { s1[q] = make_float4(0.f, 0.f, 0.f, 0.f); s2[q] = s1[q]; }
    for (int s0 = 0; s0 < S; s0 += CQ) {
        // ---- this lane's view, both planes ----
        const int sv = min(s0 + cq, S - 1);
        const float* P = proj + (b * S + sv) * 12;
        const float P0 = P[0], P1 = P[1], P2 = P[2], P3 = P[3], P4 = P[4], P5 = P[5], P6 = P[6], P7 = P[7], P8 = P[8], P9 = P[9], P10 = P[10], P11 = P[11];
        const unsigned vb = (unsigned)(b * S + sv) * img;
        int my_o[NPL][4];
        float my_w[NPL][4];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const float px = P0 * fx + P1 * fy + P2 + P3 / depth[q];        // utils.py:72 (IEEE divisions, as the reference)
            const float py = P4 * fx + P5 * fy + P6 + P7 / depth[q];
            const float pz = P8 * fx + P9 * fy + P10 + P11 / depth[q];
            const float z = clamp_min(pz, 1e-6f);                           // utils.py:80
            const float gx = (px / z) * inv_half_w - 1.f, gy = (py / z) * inv_half_h - 1.f;
            const Taps2 t = gs_taps2<false>(gs_unnorm(gx, Ws), gs_unnorm(gy, Hs), Ws, Hs);
            const int r0 = mul24(t.y0, Ws), r1 = mul24(t.y1, Ws);
            my_o[q][0] = (int)(vb + (unsigned)mul24(r0 + t.x0, C)); my_o[q][1] = (int)(vb + (unsigned)mul24(r0 + t.x1, C));
            my_o[q][2] = (int)(vb + (unsigned)mul24(r1 + t.x0, C)); my_o[q][3] = (int)(vb + (unsigned)mul24(r1 + t.x1, C));
            my_w[q][0] = t.w00; my_w[q][1] = t.w01; my_w[q][2] = t.w10; my_w[q][3] = t.w11;
        }
        // ---- the group's views in turn: all eight gathers of a view first, then the blends ----
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            if (s0 + k >= S) break;                                          // uniform
            float4 v[NPL][4];
            float wgt[NPL][4];
#pragma unroll
            for (int q = 0; q < NPL; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned o = (unsigned)group_bcast_i<CQ>(my_o[q][c], k) + (unsigned)(cq * 4);
                    wgt[q][c] = group_bcast_f<CQ>(my_w[q][c], k);
                    v[q][c] = *reinterpret_cast<const float4*>(feat + o);
                }
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                float4 r;
                r.x = v[q][0].x * wgt[q][0]; r.y = v[q][0].y * wgt[q][0]; r.z = v[q][0].z * wgt[q][0]; r.w = v[q][0].w * wgt[q][0];
#pragma unroll
                for (int c = 1; c < 4; ++c) { r.x += v[q][c].x * wgt[q][c]; r.y += v[q][c].y * wgt[q][c]; r.z += v[q][c].z * wgt[q][c]; r.w += v[q][c].w * wgt[q][c]; }
                s1[q].x += r.x; s1[q].y += r.y; s1[q].z += r.z; s1[q].w += r.w;
                s2[q].x += r.x * r.x; s2[q].y += r.y * r.y; s2[q].z += r.z * r.z; s2[q].w += r.w * r.w;
            }
        }
    }
    const float inv_s = 1.f / (float)S;                                      // utils.py:345 (see the one-plane kernel)
    const unsigned nvp = (unsigned)(D * h * w);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        float4 o;
        float m;
        m = s1[q].x * inv_s; o.x = s2[q].x * inv_s - m * m;
        m = s1[q].y * inv_s; o.y = s2[q].y * inv_s - m * m;
        m = s1[q].z * inv_s; o.z = s2[q].z * inv_s - m * m;
        m = s1[q].w * inv_s; o.w = s2[q].w * inv_s - m * m;
        const long long oidx = planar ? ((long long)((unsigned)b * CQ + cq) * nvp + (vox[q] - (unsigned)b * nvp)) * 4 : (long long)vox[q] * C + cq * 4;
        if (live) *reinterpret_cast<float4*>(vol + oidx) = o;
    }
}

void launch_feature_volume(const float* feat_nhwc, const float* proj, const float* dv, int B, int S, int C, int Hs,
                           int Ws, int D, int h, int w, float* vol, hipStream_t st, int planar) {
    // grid = 8 row bands (one per XCD) x chunks of 256/CQ voxels of a band x (B * D) planes
    const int rb = (h + 7) / 8;
    const int vpb = 256 / (C / 4);
    const float inv_w = 1.f / (float)w, inv_d = 1.f / (float)D;
#ifndef ENERF_VOL_NPL
#define ENERF_VOL_NPL 2
#endif
    if (ENERF_VOL_NPL == 2 && D % 2 == 0) {                       // two planes per wave
        const dim3 grid2(8, (unsigned)cdiv(rb * w, vpb), (unsigned)(B * D / 2));
        switch (C) {
            case 32: ENERF_LAUNCH(k_feature_volume_mp<8>, grid2, 256, 0, st, feat_nhwc, proj, dv, B, S, Hs, Ws, D, h, w, inv_w, inv_d, planar, vol); return;
            case 16: ENERF_LAUNCH(k_feature_volume_mp<4>, grid2, 256, 0, st, feat_nhwc, proj, dv, B, S, Hs, Ws, D, h, w, inv_w, inv_d, planar, vol); return;
            case 8: ENERF_LAUNCH(k_feature_volume_mp<2>, grid2, 256, 0, st, feat_nhwc, proj, dv, B, S, Hs, Ws, D, h, w, inv_w, inv_d, planar, vol); return;
            default: return;   // validated by the C-ABI layer
        }
    }
    const dim3 grid(8, (unsigned)cdiv(rb * w, vpb), (unsigned)(B * D));
    switch (C) {
        case 32: ENERF_LAUNCH(k_feature_volume<8>, grid, 256, 0, st, feat_nhwc, proj, dv, B, S, Hs, Ws, D, h, w, inv_w, inv_d, planar, vol); break;
        case 16: ENERF_LAUNCH(k_feature_volume<4>, grid, 256, 0, st, feat_nhwc, proj, dv, B, S, Hs, Ws, D, h, w, inv_w, inv_d, planar, vol); break;
        case 8: ENERF_LAUNCH(k_feature_volume<2>, grid, 256, 0, st, feat_nhwc, proj, dv, B, S, Hs, Ws, D, h, w, inv_w, inv_d, planar, vol); break;
        default: break;   // validated by the C-ABI layer
    }
}

}  // namespace enerf
