// backward.hip — hand-written backward kernels of the training path, first batch: the stages
// around the dense layers whose PyTorch backward materialises large intermediates or runs as dozens of tiny launches:
//   * homo_warp + variance (utils.py:57-95, 322-349): gradient w.r.t. the source feature maps (scatter-add of the four
//     bilinear taps) AND w.r.t. the depth hypotheses (through the warp grid — this is how level 1 back-propagates into
//     level 0's depth/std).  The forward never materialises the S warped volumes, neither does this.
//   * depth_regression (utils.py:658-667): softmax over D, mean, std -> gradients of prob and depth_values.
//   * raw2outputs (utils.py:571-603): alpha compositing forward and backward (cumprod, the softmaxed weights quirk).
// All fp32, one thread (group) per voxel / pixel / ray, atomics only for the feature scatter.  Gradients follow torch
// autograd's formulas (grid_sampler_2d_backward, clamp_min, softmax, cumprod) and are checked against them and against
// the reference's own parameter gradients (tests/test_training.py).
#include "kernels.h"

namespace enerf {

// ---------------------------------------------------------------------------------------------------------------------
// build_feature_volume backward.  CQ = C/4 lanes own one voxel (one float4 of channels each), as in the forward.
//   var[c] = mean_s f_s[c]^2 - (mean_s f_s[c])^2          =>   d f_s[c] = (2/S) g[c] (f_s[c] - mean[c])
//   f_s = bilinear(feat_s, u, v) (zeros padding)          =>   d feat_s[tap] += w_tap d f_s ;  d u, d v from the tap values
//   (u, v) = p.xy / max(p.z, 1e-6), p = R [x,y,1] + T / d =>   d d = -(T . d p) / d^2
// ---------------------------------------------------------------------------------------------------------------------
template <int CQ>
__global__ __launch_bounds__(256) void k_feature_volume_bwd(const float* __restrict__ feat, const float* __restrict__ proj,
                                                            const float* __restrict__ dv, const float* __restrict__ gvol,
                                                            int B, int S, int Hs, int Ws, int D, int h, int w,
                                                            float* __restrict__ gfeat, float* __restrict__ gdv) {
    constexpr int C = CQ * 4;
    const long long nvox = (long long)B * D * h * w;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long vraw = t / CQ;
    const int cq = (int)(t - vraw * CQ);
    const bool live = vraw < nvox;
    const long long vox = live ? vraw : nvox - 1;
    const int x = (int)(vox % w), y = (int)((vox / w) % h);
    const int b = (int)(vox / ((long long)D * h * w));
    const float depth = dv[vox];
    const float fx = (float)x, fy = (float)y;
    const float4 g = *reinterpret_cast<const float4*>(gvol + vox * C + cq * 4);
    const long long img = (long long)Hs * Ws * C;
    // pass 1: mean over views of the warped features
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
        const float* P = proj + (b * S + s) * 12;
        const float px = P[0] * fx + P[1] * fy + P[2] + P[3] / depth;
        const float py = P[4] * fx + P[5] * fy + P[6] + P[7] / depth;
        const float pz = P[8] * fx + P[9] * fy + P[10] + P[11] / depth;
        const float z = clamp_min(pz, 1e-6f);
        const Taps2 tp = gs_taps2<false>(px / z, py / z, Ws, Hs);
        const float* base = feat + (long long)(b * S + s) * img + cq * 4;
        const float4 v00 = *reinterpret_cast<const float4*>(base + ((long long)tp.y0 * Ws + tp.x0) * C);
        const float4 v01 = *reinterpret_cast<const float4*>(base + ((long long)tp.y0 * Ws + tp.x1) * C);
        const float4 v10 = *reinterpret_cast<const float4*>(base + ((long long)tp.y1 * Ws + tp.x0) * C);
        const float4 v11 = *reinterpret_cast<const float4*>(base + ((long long)tp.y1 * Ws + tp.x1) * C);
        mean.x += v00.x * tp.w00 + v01.x * tp.w01 + v10.x * tp.w10 + v11.x * tp.w11;
        mean.y += v00.y * tp.w00 + v01.y * tp.w01 + v10.y * tp.w10 + v11.y * tp.w11;
        mean.z += v00.z * tp.w00 + v01.z * tp.w01 + v10.z * tp.w10 + v11.z * tp.w11;
        mean.w += v00.w * tp.w00 + v01.w * tp.w01 + v10.w * tp.w10 + v11.w * tp.w11;
    }
    const float inv_s = 1.f / (float)S;
    mean.x *= inv_s; mean.y *= inv_s; mean.z *= inv_s; mean.w *= inv_s;
    // pass 2: per view, d f_s -> scatter to the four taps, and the grid gradient -> d depth
    float gd = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* P = proj + (b * S + s) * 12;
        const float px = P[0] * fx + P[1] * fy + P[2] + P[3] / depth;
        const float py = P[4] * fx + P[5] * fy + P[6] + P[7] / depth;
        const float pz = P[8] * fx + P[9] * fy + P[10] + P[11] / depth;
        const float z = clamp_min(pz, 1e-6f);
        const float u = px / z, v = py / z;
        const Taps2 tp = gs_taps2<false>(u, v, Ws, Hs);
        const long long vb = (long long)(b * S + s) * img + cq * 4;
        const long long o00 = vb + ((long long)tp.y0 * Ws + tp.x0) * C, o01 = vb + ((long long)tp.y0 * Ws + tp.x1) * C;
        const long long o10 = vb + ((long long)tp.y1 * Ws + tp.x0) * C, o11 = vb + ((long long)tp.y1 * Ws + tp.x1) * C;
        const float4 v00 = *reinterpret_cast<const float4*>(feat + o00), v01 = *reinterpret_cast<const float4*>(feat + o01);
        const float4 v10 = *reinterpret_cast<const float4*>(feat + o10), v11 = *reinterpret_cast<const float4*>(feat + o11);
        float f[4] = {v00.x * tp.w00 + v01.x * tp.w01 + v10.x * tp.w10 + v11.x * tp.w11,
                      v00.y * tp.w00 + v01.y * tp.w01 + v10.y * tp.w10 + v11.y * tp.w11,
                      v00.z * tp.w00 + v01.z * tp.w01 + v10.z * tp.w10 + v11.z * tp.w11,
                      v00.w * tp.w00 + v01.w * tp.w01 + v10.w * tp.w10 + v11.w * tp.w11};
        const float gg[4] = {g.x, g.y, g.z, g.w}, mm[4] = {mean.x, mean.y, mean.z, mean.w};
        const float a00[4] = {v00.x, v00.y, v00.z, v00.w}, a01[4] = {v01.x, v01.y, v01.z, v01.w};
        const float a10[4] = {v10.x, v10.y, v10.z, v10.w}, a11[4] = {v11.x, v11.y, v11.z, v11.w};
        // validity of the taps (zeros padding): a weight of exactly 0 marks an out-of-image tap or a zero-area one
        const float fxu = floorf(u), fyv = floorf(v);
        const bool fin = (u > -1e8f) && (u < 1e8f) && (v > -1e8f) && (v < 1e8f);
        const float tx1 = u - fxu, ty1 = v - fyv, tx0 = 1.f - tx1, ty0 = 1.f - ty1;
        const int x0 = (int)fxu, y0 = (int)fyv;
        const bool vx0 = fin && x0 >= 0 && x0 < Ws, vx1 = fin && x0 + 1 >= 0 && x0 + 1 < Ws;
        const bool vy0 = fin && y0 >= 0 && y0 < Hs, vy1 = fin && y0 + 1 >= 0 && y0 + 1 < Hs;
        float gu = 0.f, gv = 0.f;
        // Neighbour merge: the voxel one to the right (lane + CQ, same channel quad) usually lands one texel to the right, so ITS
        // left taps are MY right taps.  When the offsets agree, its two left contributions ride on my right atomics and it
        // skips them: two atomics per voxel, view and channel instead of four where the warp is ~1:1 (level 1) — the kernel is
        // bound by the L2 atomic rate.  (The order of fp32 atomic sums is arbitrary anyway.)
        const int lane_ = threadIdx.x & 63;
        const bool has_r = lane_ + CQ < 64, has_l = lane_ >= CQ;
        const int rl = has_r ? lane_ + CQ : lane_, ll = has_l ? lane_ - CQ : lane_;
        // (offsets compared as two 32-bit halves: lane broadcasts are 32-bit; every lane executes every broadcast)
        const int r00lo = __shfl((int)(o00 & 0xffffffffLL), rl), r00hi = __shfl((int)(o00 >> 32), rl);
        const int r10lo = __shfl((int)(o10 & 0xffffffffLL), rl), r10hi = __shfl((int)(o10 >> 32), rl);
        const int r_ok0 = __shfl((int)(vx0 && vy0), rl), r_ok1 = __shfl((int)(vx0 && vy1), rl);
        const bool same0 = r00lo == (int)(o01 & 0xffffffffLL) && r00hi == (int)(o01 >> 32);
        const bool same1 = r10lo == (int)(o11 & 0xffffffffLL) && r10hi == (int)(o11 >> 32);
        const bool m0 = has_r && r_ok0 && vx1 && vy0 && same0;      // right neighbour's (y0,x0) == my (y0,x1)
        const bool m1 = has_r && r_ok1 && vx1 && vy1 && same1;      // right neighbour's (y1,x0) == my (y1,x1)
        const int l_m0 = __shfl((int)m0, ll), l_m1 = __shfl((int)m1, ll);
        const bool skip0 = has_l && l_m0, skip1 = has_l && l_m1;    // my left taps were taken by the left neighbour
        // The four taps' contributions of this lane's channel quad.  They are NOT added from here: the L2 executes fp32 atomics
        // per 64-byte LINE REQUEST of a wave instruction (tools/micro/atomic_rate.hip: 20.8 G requests/s whatever the lanes
        // carry — 334 G adds/s when 16 lanes share a line, 82 G/s in this kernel's natural layout, where an instruction holds one
        // channel of every quad: 16-byte stride, a texel's line requested four times per tap).  The wave transposes them through
        // LDS so that an instruction carries whole texels: lane = channel, 64 / C texels per instruction, one request per line.
        float4 t00 = make_float4(0.f, 0.f, 0.f, 0.f), t01 = t00, t10 = t00, t11 = t00;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float df = live ? (2.f * inv_s) * gg[c] * (f[c] - mm[c]) : 0.f;
            const float c00 = tp.w00 * df, c10 = tp.w10 * df;
            const float r00 = __shfl(c00, rl), r10 = __shfl(c10, rl);
            if (vx0 && vy0) { (&t00.x)[c] = c00; gu -= a00[c] * ty0 * df; gv -= a00[c] * tx0 * df; }
            if (vx1 && vy0) { (&t01.x)[c] = tp.w01 * df + (m0 ? r00 : 0.f); gu += a01[c] * ty0 * df; gv -= a01[c] * tx1 * df; }
            if (vx0 && vy1) { (&t10.x)[c] = c10; gu -= a10[c] * ty1 * df; gv += a10[c] * tx0 * df; }
            if (vx1 && vy1) { (&t11.x)[c] = tp.w11 * df + (m1 ? r10 : 0.f); gu += a11[c] * ty1 * df; gv += a11[c] * tx1 * df; }
        }
        {
            constexpr int NVW = 64 / CQ;                        // voxels per wave
            __shared__ __attribute__((aligned(16))) float xval[4][4][256];       // [wave][tap][voxel-in-wave * C + channel]
            __shared__ int xoff[4][4][NVW];                                      // [wave][tap][voxel-in-wave]: texel offset or -1
            const int wv = threadIdx.x >> 6, vin = lane_ / CQ;
            *reinterpret_cast<float4*>(&xval[wv][0][lane_ * 4]) = t00;
            *reinterpret_cast<float4*>(&xval[wv][1][lane_ * 4]) = t01;
            *reinterpret_cast<float4*>(&xval[wv][2][lane_ * 4]) = t10;
            *reinterpret_cast<float4*>(&xval[wv][3][lane_ * 4]) = t11;
            if (cq == 0) {                                      // (offsets fit 32 bits: the C entry rejects larger maps)
                xoff[wv][0][vin] = (live && vx0 && vy0 && !skip0) ? (int)(o00 - cq * 4) : -1;
                xoff[wv][1][vin] = (live && vx1 && vy0) ? (int)(o01 - cq * 4) : -1;
                xoff[wv][2][vin] = (live && vx0 && vy1 && !skip1) ? (int)(o10 - cq * 4) : -1;
                xoff[wv][3][vin] = (live && vx1 && vy1) ? (int)(o11 - cq * 4) : -1;
            }
            wave_sync();
#pragma unroll
            for (int tap = 0; tap < 4; ++tap)
#pragma unroll
                for (int i = 0; i < 4; ++i) {                   // 256 floats of the wave's voxels, 64 per instruction
                    const int e = i * 64 + lane_, v = e / C, ch = e - v * C;
                    const int off = xoff[wv][tap][v];
                    const float val = xval[wv][tap][e];
                    if (off >= 0) atomic_add_f32(gfeat + off + ch, val);
                }
            wave_sync();                                        // before the next view overwrites the staging arrays
        }
        // (u, v) = p.xy / z ; z = max(p.z, 1e-6)
        const float gpx = gu / z, gpy = gv / z;
        const float gpz = pz >= 1e-6f ? -(gu * px + gv * py) / (z * z) : 0.f;
        gd -= (P[3] * gpx + P[7] * gpy + P[11] * gpz) / (depth * depth);
    }
    // sum over the CQ channel lanes of the voxel
    for (int m = CQ >> 1; m >= 1; m >>= 1) gd += __shfl_xor(gd, m);
    if (live && cq == 0) gdv[vox] = gd;
}
bool launch_feature_volume_bwd(const float* feat, const float* proj, const float* dv, const float* gvol, int B, int S, int C,
                               int Hs, int Ws, int D, int h, int w, float* gfeat, float* gdv, hipStream_t st) {
    const long long threads = (long long)B * D * h * w * (C / 4);
    const unsigned grid = (unsigned)cdivl(threads, 256);
    switch (C) {
        case 32: ENERF_LAUNCH(k_feature_volume_bwd<8>, grid, 256, 0, st, feat, proj, dv, gvol, B, S, Hs, Ws, D, h, w, gfeat, gdv); return true;
        case 16: ENERF_LAUNCH(k_feature_volume_bwd<4>, grid, 256, 0, st, feat, proj, dv, gvol, B, S, Hs, Ws, D, h, w, gfeat, gdv); return true;
        case 8: ENERF_LAUNCH(k_feature_volume_bwd<2>, grid, 256, 0, st, feat, proj, dv, gvol, B, S, Hs, Ws, D, h, w, gfeat, gdv); return true;
        default: return false;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// depth_regression backward: p = softmax_D(prob), v = depth_inv ? 1/max(dv,1e-6) : dv, mu = sum p v,
// var = sum p (v - mu)^2, std = sqrt(max(var, 1e-10)).
// Round 6: the forward kernel's mapping (geometry.hip k_depth_regression: a wave = 16 pixels x 4 depth slices, lane slice sl takes
// planes sl, sl + 4, ...; every plane value loaded ONCE into registers, exp'd once, the slice sums combined with two lane swaps).
// One thread per pixel walked the D <= 64 planes six times with dependent loads and an expf per visit, on 20 blocks for level 0's
// 64 x 80 pixels: 71 us per launch, two launches per training step.
// ---------------------------------------------------------------------------------------------------------------------
template <int MK>
__global__ __launch_bounds__(256) void k_depth_regression_bwd(const float* __restrict__ prob, const float* __restrict__ dv,
                                                              const float* __restrict__ g_depth, const float* __restrict__ g_std,
                                                              int B, int D, int h, int w, int depth_inv,
                                                              float* __restrict__ g_prob, float* __restrict__ g_dv) {
    const int lane = threadIdx.x & 63, sl = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long hw = (long long)h * w;
    const long long i = wave * 16 + (lane & 15);
    const bool ok = i < (long long)B * hw;
    const long long ii = ok ? i : 0;                    // dead lanes shadow pixel 0 (they take part in the lane swaps)
    const long long b = ii / hw, p = ii - b * hw;
    const float* pr = prob + b * D * hw + p;
    const float* dp = dv + b * D * hw + p;
    float e[MK], v[MK], d[MK];
    float m = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < MK; ++kk) {
        const int k = sl + 4 * kk;
        const bool in = k < D;
        const long long o = (long long)(in ? k : 0) * hw;
        const float x = pr[o];
        d[kk] = dp[o];
        e[kk] = in ? x : -INFINITY;
        v[kk] = depth_inv ? 1.f / clamp_min(d[kk], 1e-6f) : d[kk];
        m = fmaxf(m, e[kk]);
    }
    m = group_max4(m);
    float se = 0.f;
#pragma unroll
    for (int kk = 0; kk < MK; ++kk)
        if (sl + 4 * kk < D) { e[kk] = expf(e[kk] - m); se += e[kk]; } else e[kk] = 0.f;
    se = group_sum4(se);
    float mu = 0.f;
#pragma unroll
    for (int kk = 0; kk < MK; ++kk) { e[kk] = e[kk] / se; mu += e[kk] * v[kk]; }          // e = p_k from here on (0 beyond D)
    mu = group_sum4(mu);
    float var = 0.f, s1 = 0.f;                          // s1 = sum p (v - mu): d var / d mu = -2 s1
#pragma unroll
    for (int kk = 0; kk < MK; ++kk) { const float c = v[kk] - mu; var += e[kk] * c * c; s1 += e[kk] * c; }
    var = group_sum4(var);
    s1 = group_sum4(s1);
    const float gvar = var >= 1e-10f ? g_std[ii] * 0.5f / sqrtf(var) : 0.f;        // clamp_min + sqrt
    const float gmu = g_depth[ii] + gvar * (-2.f * s1);
    float dot = 0.f;                                    // sum_j p_j dL/dp_j (softmax backward)
#pragma unroll
    for (int kk = 0; kk < MK; ++kk) { const float c = v[kk] - mu; dot += e[kk] * (gmu * v[kk] + gvar * c * c); }
    dot = group_sum4(dot);
    if (!ok) return;
#pragma unroll
    for (int kk = 0; kk < MK; ++kk) {
        const int k = sl + 4 * kk;
        if (k >= D) continue;
        const float c = v[kk] - mu, pk = e[kk];
        const float gp = gmu * v[kk] + gvar * c * c;
        g_prob[b * D * hw + k * hw + p] = pk * (gp - dot);
        const float gvk = gmu * pk + gvar * 2.f * pk * c;
        g_dv[b * D * hw + k * hw + p] = depth_inv ? (d[kk] >= 1e-6f ? -gvk / (d[kk] * d[kk]) : 0.f) : gvk;
    }
}
// one thread per pixel, any D (the round 1 - 5 kernel; kept for D > 64)
__global__ __launch_bounds__(256) void k_depth_regression_bwd_serial(const float* __restrict__ prob, const float* __restrict__ dv,
                                                                     const float* __restrict__ g_depth, const float* __restrict__ g_std,
                                                                     int B, int D, int h, int w, int depth_inv,
                                                                     float* __restrict__ g_prob, float* __restrict__ g_dv) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long hw = (long long)h * w;
    if (i >= (long long)B * hw) return;
    const long long b = i / hw, p = i - b * hw;
    const float* pr = prob + b * D * hw + p;
    const float* dp = dv + b * D * hw + p;
    float m = -INFINITY;
    for (int k = 0; k < D; ++k) m = fmaxf(m, pr[k * hw]);
    float se = 0.f;
    for (int k = 0; k < D; ++k) se += expf(pr[k * hw] - m);
    float mu = 0.f;
    for (int k = 0; k < D; ++k) {
        const float d = dp[k * hw], v = depth_inv ? 1.f / clamp_min(d, 1e-6f) : d;
        mu += (expf(pr[k * hw] - m) / se) * v;
    }
    float var = 0.f, s1 = 0.f;
    for (int k = 0; k < D; ++k) {
        const float d = dp[k * hw], v = depth_inv ? 1.f / clamp_min(d, 1e-6f) : d;
        const float pk = expf(pr[k * hw] - m) / se;
        var += pk * (v - mu) * (v - mu);
        s1 += pk * (v - mu);
    }
    const float gvar = var >= 1e-10f ? g_std[i] * 0.5f / sqrtf(var) : 0.f;
    const float gmu = g_depth[i] + gvar * (-2.f * s1);
    float dot = 0.f;
    for (int k = 0; k < D; ++k) {
        const float d = dp[k * hw], v = depth_inv ? 1.f / clamp_min(d, 1e-6f) : d;
        const float pk = expf(pr[k * hw] - m) / se;
        dot += pk * (gmu * v + gvar * (v - mu) * (v - mu));
    }
    for (int k = 0; k < D; ++k) {
        const float d = dp[k * hw], v = depth_inv ? 1.f / clamp_min(d, 1e-6f) : d;
        const float pk = expf(pr[k * hw] - m) / se;
        const float gp = gmu * v + gvar * (v - mu) * (v - mu);
        g_prob[b * D * hw + k * hw + p] = pk * (gp - dot);
        const float gvk = gmu * pk + gvar * 2.f * pk * (v - mu);
        g_dv[b * D * hw + k * hw + p] = depth_inv ? (d >= 1e-6f ? -gvk / (d * d) : 0.f) : gvk;
    }
}
void launch_depth_regression_bwd(const float* prob, const float* dv, const float* g_depth, const float* g_std, int B, int D,
                                 int h, int w, int depth_inv, float* g_prob, float* g_dv, hipStream_t st) {
    const unsigned grid4 = (unsigned)cdivl((long long)B * h * w, 64);               // 16 pixels per wave, 4 waves per block
    if (D <= 16)
        ENERF_LAUNCH(k_depth_regression_bwd<4>, grid4, 256, 0, st, prob, dv, g_depth, g_std, B, D, h, w, depth_inv, g_prob, g_dv);
    else if (D <= 64)
        ENERF_LAUNCH(k_depth_regression_bwd<16>, grid4, 256, 0, st, prob, dv, g_depth, g_std, B, D, h, w, depth_inv, g_prob, g_dv);
    else
        ENERF_LAUNCH_SIMPLE(k_depth_regression_bwd_serial, (unsigned)cdivl((long long)B * h * w, 256), 256, 0, st, prob, dv, g_depth, g_std,
                            B, D, h, w, depth_inv, g_prob, g_dv);
}

// ---------------------------------------------------------------------------------------------------------------------
// raw2outputs (utils.py:571-603) forward / backward, one thread per ray (Ns <= 8).  raw (n,Ns,4) = [rgb, sigma], z (n,Ns).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_composite_fwd(const float* __restrict__ raw, const float* __restrict__ z, long long n,
                                                       int Ns, int white_bkgd, float* __restrict__ rgb,
                                                       float* __restrict__ depth, float* __restrict__ weights) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float wt[8], T = 1.f, c[3] = {0.f, 0.f, 0.f}, m = -INFINITY;
    for (int k = 0; k < Ns; ++k) {
        const float* r = raw + (i * Ns + k) * 4;
        const float alpha = 1.f - expf(-r[3]);
        wt[k] = alpha * T;
        T *= (1.f - alpha + 1e-10f);
        c[0] += wt[k] * r[0]; c[1] += wt[k] * r[1]; c[2] += wt[k] * r[2];
        m = fmaxf(m, wt[k]);
    }
    float se = 0.f;
    for (int k = 0; k < Ns; ++k) { wt[k] = expf(wt[k] - m); se += wt[k]; }
    float d = 0.f, acc = 0.f;
    for (int k = 0; k < Ns; ++k) {
        const float wk = wt[k] / se;
        weights[i * Ns + k] = wk;
        d += wk * z[i * Ns + k];
        acc += wk;
    }
    depth[i] = d;
    for (int q = 0; q < 3; ++q) rgb[i * 3 + q] = c[q] + (white_bkgd ? 1.f - acc : 0.f);
}
__global__ __launch_bounds__(256) void k_composite_bwd(const float* __restrict__ raw, const float* __restrict__ z,
                                                       const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                       const float* __restrict__ g_weights, long long n, int Ns,
                                                       float* __restrict__ g_raw, float* __restrict__ g_z) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float alpha[8], tt[8], Tk[8], wt[8], ws[8];
    float T = 1.f, m = -INFINITY;
    for (int k = 0; k < Ns; ++k) {
        alpha[k] = 1.f - expf(-raw[(i * Ns + k) * 4 + 3]);
        tt[k] = 1.f - alpha[k] + 1e-10f;
        Tk[k] = T;
        wt[k] = alpha[k] * T;
        T *= tt[k];
        m = fmaxf(m, wt[k]);
    }
    float se = 0.f;
    for (int k = 0; k < Ns; ++k) { ws[k] = expf(wt[k] - m); se += ws[k]; }
    const bool has_rgb = g_rgb != nullptr;             // absent output gradients (nullptr) are zeros
    const float gr[3] = {has_rgb ? g_rgb[i * 3] : 0.f, has_rgb ? g_rgb[i * 3 + 1] : 0.f, has_rgb ? g_rgb[i * 3 + 2] : 0.f};
    const float gd = g_depth != nullptr ? g_depth[i] : 0.f;
    float gws[8], dot = 0.f;                            // gradient w.r.t. the softmaxed weights (white_bkgd adds 1 - sum = const)
    for (int k = 0; k < Ns; ++k) {
        ws[k] /= se;
        gws[k] = (g_weights != nullptr ? g_weights[i * Ns + k] : 0.f) + gd * z[i * Ns + k];
        g_z[i * Ns + k] = gd * ws[k];
        dot += ws[k] * gws[k];
    }
    float gT[8], galpha[8];
    for (int k = 0; k < Ns; ++k) {
        const float* r = raw + (i * Ns + k) * 4;
        const float gwt = ws[k] * (gws[k] - dot) + gr[0] * r[0] + gr[1] * r[1] + gr[2] * r[2];
        for (int q = 0; q < 3; ++q) g_raw[(i * Ns + k) * 4 + q] = wt[k] * gr[q];
        galpha[k] = gwt * Tk[k];
        gT[k] = gwt * alpha[k];
    }
    // T_k = prod_{j<k} t_j :  d t_j = sum_{k>j} gT_k T_k / t_j   (torch cumprod backward, no zeros: t >= 1e-10)
    float run = 0.f;
    for (int j = Ns - 1; j >= 0; --j) {
        const float gt = run / tt[j];
        galpha[j] -= gt;                                // t_j = 1 - alpha_j + 1e-10
        run += gT[j] * Tk[j];
        g_raw[(i * Ns + j) * 4 + 3] = galpha[j] * expf(-raw[(i * Ns + j) * 4 + 3]);   // alpha = 1 - exp(-sigma)
    }
}
void launch_composite_fwd(const float* raw, const float* z, long long n, int Ns, int white_bkgd, float* rgb, float* depth,
                          float* weights, hipStream_t st) {
    ENERF_LAUNCH_SIMPLE(k_composite_fwd, (unsigned)cdivl(n, 256), 256, 0, st, raw, z, n, Ns, white_bkgd, rgb, depth, weights);
}
void launch_composite_bwd(const float* raw, const float* z, const float* g_rgb, const float* g_depth, const float* g_weights,
                          long long n, int Ns, float* g_raw, float* g_z, hipStream_t st) {
    ENERF_LAUNCH_SIMPLE(k_composite_bwd, (unsigned)cdivl(n, 256), 256, 0, st, raw, z, g_rgb, g_depth, g_weights, n, Ns, g_raw, g_z);
}

}  // namespace enerf

using namespace enerf;
extern "C" {

int enerf_build_feature_volume_bwd(const float* feat, const float* proj, const float* depth_values, const float* grad_vol, int B,
                                   int S, int C, int Hs, int Ws, int D, int h, int w, float* grad_feat, float* grad_depth_values,
                                   enerf_stream_t stream) {
    REQUIRE(feat && proj && depth_values && grad_vol && grad_feat && grad_depth_values, "build_feature_volume_bwd: null pointer");
    REQUIRE(C == 8 || C == 16 || C == 32, "build_feature_volume_bwd: C=%d unsupported (8/16/32)", C);
    REQUIRE(B > 0 && S > 0 && Hs > 1 && Ws > 1 && D > 0 && h > 0 && w > 0, "build_feature_volume_bwd: bad shape");
    REQUIRE((long long)B * S * Hs * Ws * C < (1LL << 31), "build_feature_volume_bwd: feature maps of %lld floats (limit 2^31)",
            (long long)B * S * Hs * Ws * C);
    zero_async(grad_feat, (size_t)B * S * Hs * Ws * C * sizeof(float), (hipStream_t)stream);
    launch_feature_volume_bwd(feat, proj, depth_values, grad_vol, B, S, C, Hs, Ws, D, h, w, grad_feat, grad_depth_values,
                              (hipStream_t)stream);
    return check_launch("build_feature_volume_bwd");
}
int enerf_depth_regression_bwd(const float* prob, const float* depth_values, const float* grad_depth, const float* grad_std, int B,
                               int D, int h, int w, int depth_inv, float* grad_prob, float* grad_depth_values,
                               enerf_stream_t stream) {
    REQUIRE(prob && depth_values && grad_depth && grad_std && grad_prob && grad_depth_values && B > 0 && D > 0 && h > 0 && w > 0,
            "depth_regression_bwd: bad arguments");
    launch_depth_regression_bwd(prob, depth_values, grad_depth, grad_std, B, D, h, w, depth_inv, grad_prob, grad_depth_values,
                                (hipStream_t)stream);
    return check_launch("depth_regression_bwd");
}
int enerf_composite(const float* raw, const float* z, long long n, int n_samples, int white_bkgd, float* rgb, float* depth,
                    float* weights, enerf_stream_t stream) {
    REQUIRE(n >= 0 && n_samples >= 1 && n_samples <= 8, "composite: n_samples must be in [1,8]");
    if (n == 0) return ENERF_OK;
    REQUIRE(raw && z && rgb && depth && weights, "composite: null pointer");
    launch_composite_fwd(raw, z, n, n_samples, white_bkgd, rgb, depth, weights, (hipStream_t)stream);
    return check_launch("composite");
}
int enerf_composite_bwd(const float* raw, const float* z, const float* grad_rgb, const float* grad_depth, const float* grad_weights,
                        long long n, int n_samples, float* grad_raw, float* grad_z, enerf_stream_t stream) {
    REQUIRE(n >= 0 && n_samples >= 1 && n_samples <= 8, "composite_bwd: n_samples must be in [1,8]");
    if (n == 0) return ENERF_OK;
    REQUIRE(raw && z && grad_raw && grad_z, "composite_bwd: null pointer");        // grad_rgb / grad_depth / grad_weights: NULL = zeros
    launch_composite_bwd(raw, z, grad_rgb, grad_depth, grad_weights, n, n_samples, grad_raw, grad_z, (hipStream_t)stream);
    return check_launch("composite_bwd");
}

}  // extern "C"
