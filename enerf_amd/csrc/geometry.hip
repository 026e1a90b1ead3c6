// geometry.hip — layout packers and the small per-pixel / per-ray stages of the ENeRF hot path.
// HBM-bound elementwise kernels: one thread per output element, coalesced along the fastest axis.
#include "kernels.h"
#include "prep_job.h"

namespace enerf {

// -------------------------------------------------------------------------------------------------
// (n, C, P) -> (n, P, Cpad): FeatureNet / cost-volume tensors to channels-last (pad channels = 0).
// Reads are coalesced across threads for every channel; each thread writes Cpad contiguous floats.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_channels_last(const float* __restrict__ src, float* __restrict__ dst, int n,
                                                       int C, long long P, int Cpad) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * P) return;
    long long img = i / P, p = i - img * P;
    const float* s = src + img * C * P + p;
    float* d = dst + i * Cpad;
    for (int c = 0; c < Cpad; ++c) d[c] = c < C ? s[(long long)c * P] : 0.f;
}
__global__ __launch_bounds__(256) void k_channels_first(const float* __restrict__ src, float* __restrict__ dst, int n,
                                                        int C, long long P, int Cpad) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * P) return;
    long long img = i / P, p = i - img * P;
    const float* s = src + i * Cpad;
    float* d = dst + img * C * P + p;
    for (int c = 0; c < C; ++c) d[(long long)c * P] = s[c];
}
void launch_channels_last(const float* src, float* dst, int n, int C, long long P, int Cpad, hipStream_t st) {
    long long tot = (long long)n * P;
    ENERF_LAUNCH_SIMPLE(k_channels_last, (unsigned)cdivl(tot, 256), 256, 0, st, src, dst, n, C, P, Cpad);
}
void launch_channels_first(const float* src, float* dst, int n, int C, long long P, int Cpad, hipStream_t st) {
    long long tot = (long long)n * P;
    ENERF_LAUNCH_SIMPLE(k_channels_first, (unsigned)cdivl(tot, 256), 256, 0, st, src, dst, n, C, P, Cpad);
}

// -------------------------------------------------------------------------------------------------
// Source-view texture for the render stage: tex[img][y][x] = [im_feat(C) | rgb(3) | 0-pad] at the render
// resolution.  rgb = bilinear_ac(src*0.5+0.5, render_scale)  (unpreprocess, utils.py:605-612);
// im_feat is resized with the same align_corners rule when its resolution differs (network.py:29-32).
// NCHW -> texel transpose through LDS: a block reads 256 consecutive pixels channel by channel
// (coalesced), parks them as [pixel][tex+1] in LDS (odd stride: conflict-free) and streams the
// 256*tex contiguous output floats with coalesced float4 stores.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_img_feat_rgb(const float* __restrict__ im_feat, int C, int Hf, int Wf,
                                                           const float* __restrict__ src, int H, int W, int Hr, int Wr,
                                                           int tex, int n_img, float* __restrict__ out) {
    ENERF_DYN_SMEM(float, tile);                       // 256 * (tex + 1)
    const int ts = tex + 1;
    const long long npix = (long long)Hr * Wr, total = npix * n_img;
    const long long base = (long long)blockIdx.x * 256;
    const long long i = base + threadIdx.x;
    if (i < total) {
        int img = (int)(i / npix);
        int p = (int)(i - (long long)img * npix);
        int y = p / Wr, x = p - y * Wr;
        float* o = tile + threadIdx.x * ts;
        {   // image features
            Lerp1 ly = ac_lerp(y, ac_scale(Hf, Hr), Hf), lx = ac_lerp(x, ac_scale(Wf, Wr), Wf);
            const float* f = im_feat + (long long)img * C * Hf * Wf;
            bool same = (Hf == Hr) && (Wf == Wr);
            for (int c = 0; c < C; ++c) {
                const float* fc = f + (long long)c * Hf * Wf;
                o[c] = same ? fc[y * Wf + x]
                            : ac_blend(ly, lx, fc[ly.i0 * Wf + lx.i0], fc[ly.i0 * Wf + lx.i1], fc[ly.i1 * Wf + lx.i0],
                                       fc[ly.i1 * Wf + lx.i1]);
            }
        }
        {   // colours
            Lerp1 ly = ac_lerp(y, ac_scale(H, Hr), H), lx = ac_lerp(x, ac_scale(W, Wr), W);
            const float* s = src + (long long)img * 3 * H * W;
            for (int c = 0; c < 3; ++c) {
                const float* sc = s + (long long)c * H * W;
                float v00 = sc[ly.i0 * W + lx.i0] * 0.5f + 0.5f, v01 = sc[ly.i0 * W + lx.i1] * 0.5f + 0.5f;
                float v10 = sc[ly.i1 * W + lx.i0] * 0.5f + 0.5f, v11 = sc[ly.i1 * W + lx.i1] * 0.5f + 0.5f;
                o[C + c] = ac_blend(ly, lx, v00, v01, v10, v11);
            }
        }
        for (int c = C + 3; c < tex; ++c) o[c] = 0.f;
    }
    __syncthreads();
    const long long nvalid = total - base < 256 ? total - base : 256;      // pixels of this block
    const int nq = (int)(nvalid * tex / 4);                                // tex % 4 == 0
    float4* dst = reinterpret_cast<float4*>(out + base * tex);
    for (int q = threadIdx.x; q < nq; q += 256) {
        int e = q * 4, px = e / tex, c = e - px * tex;
        const float* t = tile + px * ts + c;
        dst[q] = make_float4(t[0], t[1], t[2], t[3]);
    }
}
void launch_pack_img_feat_rgb(const float* im_feat, int C, int Hf, int Wf, const float* src_inps, int H, int W, int Hr,
                              int Wr, int tex, int n_img, float* out, hipStream_t st) {
    long long tot = (long long)Hr * Wr * n_img;
    size_t shmem = (size_t)256 * (tex + 1) * sizeof(float);
    ENERF_LAUNCH(k_pack_img_feat_rgb, (unsigned)cdivl(tot, 256), 256, shmem, st, im_feat, C, Hf, Wf, src_inps, H, W, Hr,
                 Wr, tex, n_img, out);
}

// -------------------------------------------------------------------------------------------------
// Same texel image, from CHANNELS-LAST features (n,Hr,Wr,C) already at the render resolution (the HIP
// FeatureNet's output): tex = [feat | bilinear_ac(src*0.5+0.5) | 0].  Used for level-0 rendering, where
// the colours are a x0.25 resize; at level 1 the smooth0 conv writes texels directly (conv2d.hip).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_texels_cl(const float* __restrict__ feat, int C,
                                                        const float* __restrict__ src, int H, int W, int Hr, int Wr,
                                                        int tex, int n_img, float* __restrict__ out) {
    ENERF_DYN_SMEM(float, tile);                       // 256 * (tex + 1)
    const int ts = tex + 1;
    const long long npix = (long long)Hr * Wr, total = npix * n_img;
    const long long base = (long long)blockIdx.x * 256;
    const long long i = base + threadIdx.x;
    if (i < total) {
        int img = (int)(i / npix);
        int p = (int)(i - (long long)img * npix);
        int y = p / Wr, x = p - y * Wr;
        float* o = tile + threadIdx.x * ts;
        const float* f = feat + i * C;
        for (int c = 0; c < C; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(f + c);
            o[c] = v.x; o[c + 1] = v.y; o[c + 2] = v.z; o[c + 3] = v.w;
        }
        Lerp1 ly = ac_lerp(y, ac_scale(H, Hr), H), lx = ac_lerp(x, ac_scale(W, Wr), W);
        const float* s = src + (long long)img * 3 * H * W;
        for (int c = 0; c < 3; ++c) {
            const float* sc = s + (long long)c * H * W;
            float v00 = sc[ly.i0 * W + lx.i0] * 0.5f + 0.5f, v01 = sc[ly.i0 * W + lx.i1] * 0.5f + 0.5f;
            float v10 = sc[ly.i1 * W + lx.i0] * 0.5f + 0.5f, v11 = sc[ly.i1 * W + lx.i1] * 0.5f + 0.5f;
            o[C + c] = ac_blend(ly, lx, v00, v01, v10, v11);
        }
        for (int c = C + 3; c < tex; ++c) o[c] = 0.f;
    }
    __syncthreads();
    const long long nvalid = total - base < 256 ? total - base : 256;
    const int nq = (int)(nvalid * tex / 4);
    float4* dst = reinterpret_cast<float4*>(out + base * tex);
    for (int q = threadIdx.x; q < nq; q += 256) {
        int e = q * 4, px = e / tex, c = e - px * tex;
        const float* t = tile + px * ts + c;
        dst[q] = make_float4(t[0], t[1], t[2], t[3]);
    }
}
void launch_pack_texels_cl(const float* feat_cl, int C, const float* src_inps, int H, int W, int Hr, int Wr, int tex,
                           int n_img, float* out, hipStream_t st) {
    long long tot = (long long)Hr * Wr * n_img;
    size_t shmem = (size_t)256 * (tex + 1) * sizeof(float);
    ENERF_LAUNCH(k_pack_texels_cl, (unsigned)cdivl(tot, 256), 256, shmem, st, feat_cl, C, src_inps, H, W, Hr, Wr, tex,
                 n_img, out);
}

// -------------------------------------------------------------------------------------------------
// get_proj_mats (utils.py:35-55):  P[b,s] = (K_s' E_s[:3]) · inv([K_t' E_t[:3]; 0 0 0 1]),
// K' = K with rows 0,1 scaled.  One thread per (b,s); fp64 internally (a 4x4 inverse per frame),
// rounded to fp32 on store.
// -------------------------------------------------------------------------------------------------
// (k_times_e / proj_one: prep_job.h)
__global__ void k_proj_mats(const float* __restrict__ src_ixts, const float* __restrict__ src_exts,
                            const float* __restrict__ tar_ixt, const float* __restrict__ tar_ext, int B, int S,
                            float src_scale, float tar_scale, float* __restrict__ proj) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S) return;
    proj_one(i, src_ixts, src_exts, tar_ixt, tar_ext, S, src_scale, tar_scale, proj);
}
void launch_proj_mats(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int B,
                      int S, float src_scale, float tar_scale, float* proj, hipStream_t st) {
    ENERF_LAUNCH_SIMPLE(k_proj_mats, cdiv(B * S, 64), 64, 0, st, src_ixts, src_exts, tar_ixt, tar_ext, B, S, src_scale,
                        tar_scale, proj);
}

// -------------------------------------------------------------------------------------------------
// get_depth_values (utils.py:98-151).  One thread per (b, y, x); loops the D planes.
// Level 0 (pdepth == nullptr): planes uniform in disparity (depth_inv) or depth between batch near/far.
// Level >0: x(h/hp) align-corners upsample of the previous level's {depth, std, near_far} (all in
// disparity: only the depth_inv[level-1]==True branch is live, utils.py:122-130), then
// [1/min(d+s, nf0), 1/max(d-s, nf1)] and D planes in between.
// torch.linspace(0,1,D): step=1/(D-1); t_k = k<D/2 ? step*k : 1 - step*(D-1-k).
// -------------------------------------------------------------------------------------------------
// (ProjJob: prep_job.h)
__global__ __launch_bounds__(256) void k_depth_values(const float* __restrict__ near_far,
                                                      const float* __restrict__ pdepth, const float* __restrict__ pstd,
                                                      const float* __restrict__ pnf, int B, int D, int h, int w, int hp,
                                                      int wp, int depth_inv, float* __restrict__ dv,
                                                      float* __restrict__ nf_out, ProjJob pj) {
    if (pj.proj != nullptr && blockIdx.x == gridDim.x - 1)
        for (int q = threadIdx.x; q < B * pj.S; q += blockDim.x)
            proj_one(q, pj.src_ixts, pj.src_exts, pj.tar_ixt, pj.tar_ext, pj.S, pj.src_scale, pj.tar_scale, pj.proj);
    // one thread per (plane, pixel): level 0 has 64x80 pixels x 48 planes — a thread per pixel left the chip idle
    // behind a 48-step serial loop of IEEE divides (10 us); the per-pixel [near, far] is recomputed per plane
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = h * w;
    if (i >= B * D * hw) return;
    const int b = i / (D * hw), r = i - b * (D * hw);
    const int k = r / hw, p = r - k * hw;
    float nn, ff;
    if (pdepth == nullptr) {
        nn = near_far[b * 2 + 0];
        ff = near_far[b * 2 + 1];
    } else {
        int y = p / w, x = p - y * w;
        Lerp1 ly = ac_lerp(y, ac_scale(hp, h), hp), lx = ac_lerp(x, ac_scale(wp, w), wp);
        int o00 = ly.i0 * wp + lx.i0, o01 = ly.i0 * wp + lx.i1, o10 = ly.i1 * wp + lx.i0, o11 = ly.i1 * wp + lx.i1;
        const float* pd = pdepth + (long long)b * hp * wp;
        const float* ps = pstd + (long long)b * hp * wp;
        const float* n0 = pnf + (long long)b * 2 * hp * wp;
        const float* n1 = n0 + hp * wp;
        float d = ac_blend(ly, lx, pd[o00], pd[o01], pd[o10], pd[o11]);
        float s = ac_blend(ly, lx, ps[o00], ps[o01], ps[o10], ps[o11]);
        float a0 = ac_blend(ly, lx, n0[o00], n0[o01], n0[o10], n0[o11]);
        float a1 = ac_blend(ly, lx, n1[o00], n1[o01], n1[o10], n1[o11]);
        float lo = d + s, hi = d - s;
        if (lo > a0) lo = a0;     // utils.py:123-125
        if (hi < a1) hi = a1;     // utils.py:126-127
        nn = 1.f / lo;            // utils.py:128
        ff = 1.f / hi;
    }
    depth_plane_value(nn, ff, b, k, p, D, hw, depth_inv, dv + i, nf_out);      // (B,D,h,w): the thread index is the element index
}
void launch_depth_values(const float* near_far, const float* pdepth, const float* pstd, const float* pnf, int B, int D,
                         int h, int w, int hp, int wp, int depth_inv, float* dv, float* nf_out, hipStream_t st) {
    ProjJob none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f};
    ENERF_LAUNCH_SIMPLE(k_depth_values, cdiv(B * D * h * w, 256), 256, 0, st, near_far, pdepth, pstd, pnf, B, D, h, w, hp,
                        wp, depth_inv, dv, nf_out, none);
}
void launch_level_prep(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int S,
                       float src_scale, float tar_scale, float* proj, const float* near_far, const float* pdepth,
                       const float* pstd, const float* pnf, int B, int D, int h, int w, int hp, int wp, int depth_inv,
                       float* dv, float* nf_out, hipStream_t st) {
    ProjJob pj = {src_ixts, src_exts, tar_ixt, tar_ext, proj, S, src_scale, tar_scale};
    ENERF_LAUNCH_SIMPLE(k_depth_values, cdiv(B * D * h * w, 256), 256, 0, st, near_far, pdepth, pstd, pnf, B, D, h, w, hp,
                        wp, depth_inv, dv, nf_out, pj);
}

// -------------------------------------------------------------------------------------------------
// depth_regression (utils.py:658-667): softmax over D, E[v], sqrt(max(Var,1e-10)); v = 1/max(dv,1e-6)
// for disparity-space levels.  A wave handles 16 pixels x 4 depth slices (lane = slice*16 + pixel; slice s
// takes planes k = s, s+4, ...): level 0 has only 64x80 pixels, so one thread per pixel would leave the
// chip idle behind a 48-step serial loop.  Slices are combined with two xor-shuffles.  prob/dv are
// (B,D,h,w): each plane read is a contiguous 64-B run per slice.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float slice_sum(float v) { return group_sum4(v); }
__device__ __forceinline__ float slice_max(float v) { return group_max4(v); }
// Softmax moments of one pixel's D planes, register-resident: every plane value is loaded once (all loads in
// flight together) and exp'd once; lane slice sl handles planes sl, sl+4, ...  The sums run in the same order as
// the streaming form in k_depth_regression, so the results are bit-identical to it.
template <int MK>
__device__ __forceinline__ void depth_moments_regs(const float* pr, const float* dp, int D, int hw, int sl, int depth_inv,
                                                   float& mu, float& var) {
    float e[MK], v[MK];
    float m = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < MK; ++kk) {
        const int k = sl + 4 * kk;
        const bool in = k < D;
        const long long o = (long long)(in ? k : 0) * hw;
        const float x = pr[o], d = dp[o];
        e[kk] = in ? x : -INFINITY;
        v[kk] = depth_inv ? 1.f / clamp_min(d, 1e-6f) : d;
        m = fmaxf(m, e[kk]);
    }
    m = slice_max(m);
    float se = 0.f;
#pragma unroll
    for (int kk = 0; kk < MK; ++kk)
        if (sl + 4 * kk < D) { e[kk] = expf(e[kk] - m); se += e[kk]; }
    se = slice_sum(se);
    mu = 0.f;
#pragma unroll
    for (int kk = 0; kk < MK; ++kk)
        if (sl + 4 * kk < D) { e[kk] = e[kk] / se; mu += e[kk] * v[kk]; }
    mu = slice_sum(mu);
    var = 0.f;
#pragma unroll
    for (int kk = 0; kk < MK; ++kk)
        if (sl + 4 * kk < D) { const float dd = v[kk] - mu; var += e[kk] * (dd * dd); }
    var = slice_sum(var);
}
__global__ __launch_bounds__(256) void k_depth_regression(const float* __restrict__ prob, const float* __restrict__ dv,
                                                          int B, int D, int h, int w, int depth_inv,
                                                          float* __restrict__ depth, float* __restrict__ std,
                                                          float* __restrict__ depth_mvs) {
    const int lane = threadIdx.x & 63, sl = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int hw = h * w;
    const long long i = wave * 16 + (lane & 15);
    const bool ok = i < (long long)B * hw;
    const long long ii = ok ? i : 0;
    const int b = (int)(ii / hw), p = (int)(ii - (long long)b * hw);
    const float* pr = prob + (long long)b * D * hw + p;
    const float* dp = dv + (long long)b * D * hw + p;
    float mu, var;
    if (D <= 16) depth_moments_regs<4>(pr, dp, D, hw, sl, depth_inv, mu, var);
    else if (D <= 64) depth_moments_regs<16>(pr, dp, D, hw, sl, depth_inv, mu, var);
    else {
        float m = -INFINITY;
        for (int k = sl; k < D; k += 4) m = fmaxf(m, pr[(long long)k * hw]);
        m = slice_max(m);
        float se = 0.f;
        for (int k = sl; k < D; k += 4) se += expf(pr[(long long)k * hw] - m);
        se = slice_sum(se);
        mu = 0.f;
        for (int k = sl; k < D; k += 4) {
            float pk = expf(pr[(long long)k * hw] - m) / se;
            float v = dp[(long long)k * hw];
            if (depth_inv) v = 1.f / clamp_min(v, 1e-6f);
            mu += pk * v;
        }
        mu = slice_sum(mu);
        var = 0.f;
        for (int k = sl; k < D; k += 4) {
            float pk = expf(pr[(long long)k * hw] - m) / se;
            float v = dp[(long long)k * hw];
            if (depth_inv) v = 1.f / clamp_min(v, 1e-6f);
            float dd = v - mu;
            var += pk * (dd * dd);
        }
        var = slice_sum(var);
    }
    if (ok && sl == 0) {
        depth[i] = mu;
        std[i] = sqrtf(clamp_min(var, 1e-10f));
        if (depth_mvs != nullptr) depth_mvs[i] = depth_inv ? 1.f / mu : mu;      // network.py:105-108
    }
}
void launch_depth_regression(const float* prob, const float* dv, int B, int D, int h, int w, int depth_inv,
                             float* depth, float* std, float* depth_mvs, hipStream_t st) {
    long long waves = cdivl((long long)B * h * w, 16);
    ENERF_LAUNCH(k_depth_regression, (unsigned)cdivl(waves, 4), 256, 0, st, prob, dv, B, D, h, w, depth_inv, depth, std,
                 depth_mvs);
}

// -------------------------------------------------------------------------------------------------
// depth_regression of level i-1 + get_depth_values (+ get_proj_mats) of level i in ONE launch (round 3).  Between two cascade
// levels the chain was cost_reg -> k_depth_regression (6.7 us) -> k_depth_values (6.4 us) -> warp: two tiny dependent launches
// on the frame's critical path.  Here a block owns a 16 x 16 tile of level-i pixels: it first computes the softmax moments of
// the <= 12 x 12 coarse pixels its align-corners upsample taps touch (same lane mapping and arithmetic as
// k_depth_regression: bit-identical values; neighbouring blocks recompute shared coarse pixels and write identical bits),
// keeps them in LDS, and then runs k_depth_values' per-pixel arithmetic for its tile.  Used when level i-1 is not rendered.
// -------------------------------------------------------------------------------------------------
constexpr int kRvTile = 16, kRvPatch = 12;
__global__ __launch_bounds__(256) void k_regress_and_values(const float* __restrict__ prob_p, const float* __restrict__ dv_p,
                                                            const float* __restrict__ nf_p, int Dp, int hp, int wp,
                                                            int depth_inv_p, float* __restrict__ depth_p,
                                                            float* __restrict__ std_p, int B, int D, int h, int w,
                                                            int depth_inv, float* __restrict__ dv, float* __restrict__ nf_out,
                                                            int tiles_y, int tiles_x, ProjJob pj) {
    __shared__ float cd[kRvPatch * kRvPatch], cs[kRvPatch * kRvPatch];
    if (pj.proj != nullptr && blockIdx.x == gridDim.x - 1)
        for (int q = threadIdx.x; q < B * pj.S; q += blockDim.x)
            proj_one(q, pj.src_ixts, pj.src_exts, pj.tar_ixt, pj.tar_ext, pj.S, pj.src_scale, pj.tar_scale, pj.proj);
    int t = blockIdx.x;
    const int txi = t % tiles_x; t /= tiles_x;
    const int tyi = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = tyi * kRvTile, x0 = txi * kRvTile;
    const int y1 = min(y0 + kRvTile, h) - 1, x1 = min(x0 + kRvTile, w) - 1;
    const float sy = ac_scale(hp, h), sx = ac_scale(wp, w);
    const int cy0 = ac_lerp(y0, sy, hp).i0, cy1 = ac_lerp(y1, sy, hp).i1;
    const int cx0 = ac_lerp(x0, sx, wp).i0, cx1 = ac_lerp(x1, sx, wp).i1;
    const int prows = cy1 - cy0 + 1, pcols = cx1 - cx0 + 1, npatch = prows * pcols;      // <= kRvPatch^2 (launcher checks the scale)
    const int hwp = hp * wp;
    // ---- phase 1: softmax moments of the patch (k_depth_regression's mapping: 16 pixels x 4 depth slices per wave) ----
    const int lane = threadIdx.x & 63, sl = lane >> 4, wv = threadIdx.x >> 6;
    for (int base = 0; base < npatch; base += 64) {
        const int c = base + wv * 16 + (lane & 15);
        const bool okc = c < npatch;
        const int cc = okc ? c : npatch - 1;
        const int pr_ = cc / pcols, pc_ = cc - pr_ * pcols;
        const int pcoarse = (cy0 + pr_) * wp + (cx0 + pc_);
        const float* pr = prob_p + (long long)b * Dp * hwp + pcoarse;
        const float* dp = dv_p + (long long)b * Dp * hwp + pcoarse;
        float mu, var;
        if (Dp <= 16) depth_moments_regs<4>(pr, dp, Dp, hwp, sl, depth_inv_p, mu, var);
        else depth_moments_regs<16>(pr, dp, Dp, hwp, sl, depth_inv_p, mu, var);          // Dp <= 64 (launcher)
        if (okc && sl == 0) {
            const float sd = sqrtf(clamp_min(var, 1e-10f));
            cd[cc] = mu; cs[cc] = sd;
            depth_p[(long long)b * hwp + pcoarse] = mu;                                  // (identical bits from every block that
            std_p[(long long)b * hwp + pcoarse] = sd;                                    //  shares this coarse pixel)
        }
    }
    __syncthreads();
    // ---- phase 2: k_depth_values' arithmetic for this tile's pixels, all D planes per thread ----
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= h || x >= w) return;
    const Lerp1 ly = ac_lerp(y, sy, hp), lx = ac_lerp(x, sx, wp);
    const int q00 = (ly.i0 - cy0) * pcols + (lx.i0 - cx0), q01 = (ly.i0 - cy0) * pcols + (lx.i1 - cx0);
    const int q10 = (ly.i1 - cy0) * pcols + (lx.i0 - cx0), q11 = (ly.i1 - cy0) * pcols + (lx.i1 - cx0);
    const int o00 = ly.i0 * wp + lx.i0, o01 = ly.i0 * wp + lx.i1, o10 = ly.i1 * wp + lx.i0, o11 = ly.i1 * wp + lx.i1;
    const float* n0 = nf_p + (long long)b * 2 * hwp;
    const float* n1 = n0 + hwp;
    const float d = ac_blend(ly, lx, cd[q00], cd[q01], cd[q10], cd[q11]);
    const float s = ac_blend(ly, lx, cs[q00], cs[q01], cs[q10], cs[q11]);
    const float a0 = ac_blend(ly, lx, n0[o00], n0[o01], n0[o10], n0[o11]);
    const float a1 = ac_blend(ly, lx, n1[o00], n1[o01], n1[o10], n1[o11]);
    float lo = d + s, hi = d - s;
    if (lo > a0) lo = a0;         // utils.py:123-125
    if (hi < a1) hi = a1;         // utils.py:126-127
    const float nn = 1.f / lo, ff = 1.f / hi;                                            // utils.py:128
    const float inn = 1.f / nn, iff = 1.f / ff;
    const int hw = h * w, p = y * w + x;
    for (int k = 0; k < D; ++k) {
        const float tk = linspace01(k, D);
        const float v = depth_inv ? 1.f / (inn + tk * (iff - inn)) : nn + tk * (ff - nn);
        dv[((long long)b * D + k) * hw + p] = v;
        if (k == 0 || k == D - 1) {               // utils.py:149-150 (k == 0 == D-1 writes both)
            const float e = depth_inv ? 1.f / clamp_min(v, 1e-6f) : v;
            if (k == 0) nf_out[((long long)b * 2 + 0) * hw + p] = e;
            if (k == D - 1) nf_out[((long long)b * 2 + 1) * hw + p] = e;
        }
    }
}
// false: shape not handled (nothing launched) — the caller then uses k_depth_regression + k_depth_values
bool launch_regress_and_values(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int S,
                               float src_scale, float tar_scale, float* proj, const float* prob_p, const float* dv_p,
                               const float* nf_p, int Dp, int hp, int wp, int depth_inv_p, float* depth_p, float* std_p, int B,
                               int D, int h, int w, int depth_inv, float* dv, float* nf_out, hipStream_t st) {
    if (Dp > 64 || h < hp || w < wp) return false;
    // the coarse patch of a 16-pixel tile side: (16 - 1) * scale + 2 taps (+1 for rounding) must fit kRvPatch
    const float sy = ac_scale(hp, h), sx = ac_scale(wp, w);
    if ((int)(15.f * sy) + 3 > kRvPatch || (int)(15.f * sx) + 3 > kRvPatch) return false;
    const int tiles_y = cdiv(h, kRvTile), tiles_x = cdiv(w, kRvTile);
    ProjJob pj = {src_ixts, src_exts, tar_ixt, tar_ext, proj, S, src_scale, tar_scale};
    ENERF_LAUNCH(k_regress_and_values, (unsigned)(B * tiles_y * tiles_x), 256, 0, st, prob_p, dv_p, nf_p, Dp, hp, wp, depth_inv_p,
                 depth_p, std_p, B, D, h, w, depth_inv, dv, nf_out, tiles_y, tiles_x, pj);
    return true;
}

// -------------------------------------------------------------------------------------------------
// build_rays (utils.py:390-420): x(Hr/h) align-corners upsample of {depth, std, near_far}, per-ray
// [near, far] clamped into the volume bounds, gathered at the ray's integer (u, v); appended to the
// 8-float ray -> 12 floats [o(3), d(3), u, v, ray_near, ray_far, vol_near, vol_far].
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_rays(const float* __restrict__ rays8, const float* __restrict__ depth,
                                                    const float* __restrict__ std, const float* __restrict__ nf, int B,
                                                    int N, int h, int w, int Hr, int Wr, int depth_inv,
                                                    float* __restrict__ rays12) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * N) return;
    int b = (int)(i / N);
    const float* r = rays8 + i * 8;
    float rv[8];
    for (int k = 0; k < 8; ++k) rv[k] = r[k];
    const RayBounds rb = ray_bounds(rv[6], rv[7], depth + (long long)b * h * w, std + (long long)b * h * w,
                                    nf + (long long)b * 2 * h * w, h, w, Hr, Wr, depth_inv);
    float* o = rays12 + i * 12;
    for (int k = 0; k < 8; ++k) o[k] = rv[k];
    o[8] = rb.rn; o[9] = rb.rf; o[10] = rb.vn; o[11] = rb.vf;
}
void launch_build_rays(const float* rays8, const float* depth, const float* std, const float* nf, int B, int N, int h,
                       int w, int Hr, int Wr, int depth_inv, float* rays12, hipStream_t st) {
    long long tot = (long long)B * N;
    ENERF_LAUNCH_SIMPLE(k_build_rays, (unsigned)cdivl(tot, 256), 256, 0, st, rays8, depth, std, nf, B, N, h, w, Hr, Wr,
                        depth_inv, rays12);
}

}  // namespace enerf
