// render.hip — fused per-ray rendering: sample_along_depth (utils.py:422-441) + get_vox_feat
// (utils.py:456-458) + get_img_feat (utils.py:689-722) + Agg/NeRF MLP (nerf.py:29-89) + raw2outputs
// (utils.py:571-603), i.e. Network.render_rays (network.py:24-43) in ONE kernel.  Nothing between the
// 12-float rays and {rgb, depth, weights} ever touches HBM (the reference materialises ~118 MB of
// gathered features plus every MLP activation).
//
// Mapping.  A wave owns 16 rays; lane l = (g = l>>4, j = l&15) works on ray j.  For every sample the
// MLP runs on v_mfma_f32_16x16x4_f32 with the *weights* as the A operand (rows = output units) and the
// 16 points as the B/D columns:
//     A: lane holds W[row = j][k = g]      B: lane holds X[k = g][col = j]
//     D: lane holds rows 4g+r (r = 0..3) of column j
// A layer's D registers are therefore directly the next layer's B operands (k-step (tile,r) supplies
// unit 16*tile+4g+r from lane group g); the packed weight image (nerf_pack) is permuted to that K
// order, so activations never move between lanes.  Gathered per-view features use the same idea:
// lane group g fetches channels [gR, gR+R) of the texel (R = ceil((C+3)/4)), register r is k-step r.
// Width-1 heads (agg weight, sigma, colour logit) are in-lane dot products + a 2-step xor reduction over
// the four lane groups.  Softmaxes over views and the compositing scan over samples are in-lane.
//
// Algebra (fp re-association only): global_fc = W_a·a_s + W_vm·[var,mean] and
// color.0 = W_p·[h,vox,agg] + W_v·[x_s,dir_s] — the view-independent halves are evaluated once per
// point instead of once per view (201 instead of 369 MFMAs per 16 points at S=3, C=8).
//
// Roofline: MFMA-bound (fp32 157.3 TF).  Algorithmic FLOPs/point: SURVEY.md §8a (50,952 at level 1).
#include "kernels.h"

namespace enerf {

// ---- packed weight image ---------------------------------------------------------------------------
struct NerfLayout {
    int F, R, TR;
    int view, viewb, glob, globb, aggw, fc, fcb, lr0, lr0b, sigma, c0p, c0b, c0v, col2, total;  // float offsets
};
__host__ __device__ __forceinline__ NerfLayout nerf_layout(int F) {
    NerfLayout L;
    L.F = F; L.R = (F + 3) / 4; L.TR = (L.R + 3) / 4;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 63) / 64 * 64; return r; };
    L.view = take(L.TR * 64);
    L.viewb = take(L.TR * 16);
    L.glob = take(3 * L.R * 2 * 64);
    L.globb = take(32);
    L.aggw = take(33);
    L.fc = take(8 * 64);
    L.fcb = take(16);
    L.lr0 = take(6 * 4 * 64);
    L.lr0b = take(64);
    L.sigma = take(65);
    L.c0p = take(22 * 4 * 64);
    L.c0b = take(64);
    L.c0v = take((L.R + 1) * 4 * 64);
    L.col2 = take(65);
    L.total = o;
    return L;
}
long long nerf_packed_floats(int F) { return nerf_layout(F).total; }

// slot (g, r) of the channel layout -> channel index (or -1)
__host__ __device__ __forceinline__ int slot_channel(int g, int r, int R, int F) {
    int c = g * R + r;
    return (r < R && c < F) ? c : -1;
}

__global__ __launch_bounds__(256) void k_nerf_pack(NerfRaw w, int F, int viewdir_agg, float* __restrict__ out) {
    const NerfLayout L = nerf_layout(F);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.total) return;
    const int R = L.R;
    const int CI = 88 + F + 4;    // color.0 fan-in
    float v = 0.f;
    auto mfma_lane = [&](int rel, int& e, int& g, int& row) { e = rel >> 6; g = (rel & 63) >> 4; row = rel & 15; };
    int e, g, row;
    if (i >= L.col2) {
        int k = i - L.col2;
        if (k < 64) v = w.col2_w[k]; else if (k == 64) v = w.col2_b[0];
    } else if (i >= L.c0v) {
        mfma_lane(i - L.c0v, e, g, row);
        int ks = e >> 2, vt = e & 3, unit = 16 * vt + row;
        if (ks < R) { int c = slot_channel(g, ks, R, F); if (c >= 0) v = w.col0_w[unit * CI + 88 + c]; }
        else if (ks == R) v = w.col0_w[unit * CI + 88 + F + g];
    } else if (i >= L.c0b) {
        int k = i - L.c0b; if (k < 64) v = w.col0_b[k];
    } else if (i >= L.c0p) {
        mfma_lane(i - L.c0p, e, g, row);
        int ks = e >> 2, vt = e & 3, unit = 16 * vt + row;
        if (ks < 16) v = w.col0_w[unit * CI + 16 * (ks >> 2) + 4 * g + (ks & 3)];         // h
        else if (ks < 18) v = w.col0_w[unit * CI + 64 + 2 * g + (ks - 16)];                // vox
        else if (ks < 22) v = w.col0_w[unit * CI + 72 + 4 * g + (ks - 18)];                // agg
    } else if (i >= L.sigma) {
        int k = i - L.sigma;
        if (k < 64) v = w.sigma_w[k]; else if (k == 64) v = w.sigma_b[0];
    } else if (i >= L.lr0b) {
        int k = i - L.lr0b; if (k < 64) v = w.lr0_b[k];
    } else if (i >= L.lr0) {
        mfma_lane(i - L.lr0, e, g, row);
        int ks = e >> 2, vt = e & 3, unit = 16 * vt + row;
        if (ks < 2) v = w.lr0_w[unit * 24 + 2 * g + ks];                                   // vox
        else if (ks < 6) v = w.lr0_w[unit * 24 + 8 + 4 * g + (ks - 2)];                    // agg
    } else if (i >= L.fcb) {
        int k = i - L.fcb; if (k < 16) v = w.fc_b[k];
    } else if (i >= L.fc) {
        mfma_lane(i - L.fc, e, g, row);                       // e = u*4 + r'
        if (e < 8) v = w.fc_w[row * 32 + 16 * (e >> 2) + 4 * g + (e & 3)];
    } else if (i >= L.aggw) {
        int k = i - L.aggw;
        if (k < 32) v = w.aggw_w[k]; else if (k == 32) v = w.aggw_b[0];
    } else if (i >= L.globb) {
        int k = i - L.globb; if (k < 32) v = w.glob_b[k];
    } else if (i >= L.glob) {
        mfma_lane(i - L.glob, e, g, row);                     // e = (part*R + r)*2 + u
        int u = e & 1, pr = e >> 1, part = pr / R, r = pr - part * R;
        int c = slot_channel(g, r, R, F);
        if (part < 3 && c >= 0) v = w.glob_w[(16 * u + row) * 3 * F + part * F + c];
    } else if (i >= L.viewb) {
        int k = i - L.viewb;                                   // [t][16]: row 4g'+r' of tile t
        int t = k >> 4, rr = k & 15;
        int c = slot_channel(rr >> 2, 4 * t + (rr & 3), R, F);
        if (t < L.TR && c >= 0 && viewdir_agg) v = w.view_b[c];
    } else {
        mfma_lane(i - L.view, e, g, row);                      // e = t
        int c = slot_channel(row >> 2, 4 * e + (row & 3), R, F);
        if (e < L.TR && c >= 0 && viewdir_agg) v = w.view_w[c * 4 + g];
    }
    out[i] = v;
}
void launch_nerf_pack(const NerfRaw& raw, int F, int viewdir_agg, float* packed, hipStream_t st) {
    int total = nerf_layout(F).total;
    ENERF_LAUNCH_SIMPLE(k_nerf_pack, cdiv(total, 256), 256, 0, st, raw, F, viewdir_agg, packed);
}

// ---- device helpers ----------------------------------------------------------------------------------
__device__ __forceinline__ float group_sum(float v) {      // sum over the 4 lane groups (lanes j, j+16, j+32, j+48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ f32x4 lds4(const float* p) {     // 16-byte aligned LDS/global read
    float4 t = *reinterpret_cast<const float4*>(p);
    return f32x4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ f32x4 relu4(f32x4 a) {
    return f32x4{relu1(a[0]), relu1(a[1]), relu1(a[2]), relu1(a[3])};
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b, float acc) {
    acc += a[0] * b[0]; acc += a[1] * b[1]; acc += a[2] * b[2]; acc += a[3] * b[3];
    return acc;
}
#define ENERF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// camera table in LDS, per (b,s): E[:3] (12) | K' (9) | centre (3) ; per b: target centre (3)
constexpr int kCamStride = 24;

template <int R, int S, int OCC>   // OCC = resident 256-thread blocks per CU the register budget is sized for
__global__ __launch_bounds__(256, OCC) void k_render_rays(RenderArgs a) {
    constexpr int TR = (R + 3) / 4;
    const NerfLayout L = nerf_layout(a.F);
    ENERF_DYN_SMEM(float, smem);
    float* wl = smem;                                    // packed weights
    float* cam = smem + L.total;                         // B*S*24
    float* tcen = cam + a.B * S * kCamStride;            // B*3

    // ---- prologue: stage weights + camera table ----
    for (int i = threadIdx.x * 4; i < L.total; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(wl + i) = *reinterpret_cast<const float4*>(a.packed + i);
    for (int i = threadIdx.x; i < a.B * (S + 1); i += blockDim.x) {
        int b = i / (S + 1), s = i - b * (S + 1);
        const float* E = (s < S) ? a.src_exts + ((long long)b * S + s) * 16 : a.tar_ext + (long long)b * 16;
        double m[16], inv[16];
        for (int k = 0; k < 16; ++k) m[k] = (double)E[k];
        bool ok = inv4x4(m, inv);
        float c0 = ok ? (float)inv[3] : NAN, c1 = ok ? (float)inv[7] : NAN, c2 = ok ? (float)inv[11] : NAN;
        if (s < S) {
            float* c = cam + ((long long)b * S + s) * kCamStride;
            for (int k = 0; k < 12; ++k) c[k] = E[k];
            const float* K = a.src_ixts + ((long long)b * S + s) * 9;
            for (int k = 0; k < 9; ++k) c[12 + k] = (k < 6) ? K[k] * a.render_scale : K[k];     // utils.py:700-701
            c[21] = c0; c[22] = c1; c[23] = c2;
        } else {
            tcen[b * 3 + 0] = c0; tcen[b * 3 + 1] = c1; tcen[b * 3 + 2] = c2;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave_in_block = threadIdx.x >> 6, waves_per_block = blockDim.x >> 6;
    // device-side ray selection (network_human.py:90-93): the number of rays is read here, not known to the host
    const long long nrays = a.ray_index != nullptr ? (long long)a.ray_count[0] : (long long)a.B * a.N;
    const long long ntiles = cdivl(nrays, 16);
    const bool scatter = a.ray_index != nullptr && a.scatter_rgb != 0;
    const int Ns = a.n_samples;
    const int TEX = 4 * R;
    const float* wlane = wl + lane;

    // natural (round-robin) tile order: handing each XCD a contiguous run of tiles measured ~3 % slower
    for (long long tile = (long long)blockIdx.x * waves_per_block + wave_in_block; tile < ntiles;
         tile += (long long)gridDim.x * waves_per_block) {
        long long ray = tile * 16 + j;
        const bool rok = ray < nrays;
        const long long rc = rok ? ray : nrays - 1;
        const long long rr = a.ray_index != nullptr ? (long long)a.ray_index[rc] : rc;     // position in the ray list
        const int b = (int)(rr / a.N);
        float ox, oy, oz, dx, dy, dz, ru, rv, rn, rf, vn, vf;
        if (a.rays8 != nullptr) {      // fused build_rays (uniform branch): 8-float ray + the level's depth/std/near_far maps
            const float* rp = a.rays8 + rr * 8;
            const float4 q0 = *reinterpret_cast<const float4*>(rp), q1 = *reinterpret_cast<const float4*>(rp + 4);
            ox = q0.x; oy = q0.y; oz = q0.z; dx = q0.w; dy = q1.x; dz = q1.y; ru = q1.z; rv = q1.w;
            const long long mo = (long long)b * a.map_h * a.map_w;
            const RayBounds rb = ray_bounds(ru, rv, a.depth_map + mo, a.std_map + mo, a.nf_map + 2 * mo, a.map_h, a.map_w,
                                            a.Hr, a.Wr, a.depth_inv);
            rn = rb.rn; rf = rb.rf; vn = rb.vn; vf = rb.vf;
        } else {
            const float* rp = a.rays12 + rr * 12;
            const float4 q0 = *reinterpret_cast<const float4*>(rp), q1 = *reinterpret_cast<const float4*>(rp + 4),
                         q2 = *reinterpret_cast<const float4*>(rp + 8);
            ox = q0.x; oy = q0.y; oz = q0.z; dx = q0.w; dy = q1.x; dz = q1.y; ru = q1.z; rv = q1.w;
            rn = q2.x; rf = q2.y; vn = q2.z; vf = q2.w;
        }
        // normalised (x,y) of the ray inside the feature volume: network.py:37 then utils.py:457
        const float gxv = (ru / (float)(a.Wr - 1)) * 2.f - 1.f, gyv = (rv / (float)(a.Hr - 1)) * 2.f - 1.f;
        // 32-bit element offsets from the (uniform) tensor bases: one saddr+voffset load per tap instead of a
        // 64-bit pointer add per tap (the launcher checks both tensors hold < 2^32 floats)
        const unsigned voff = (unsigned)b * (unsigned)(a.D * a.h * a.w * 8) + 2u * g;
        const unsigned toff = (unsigned)b * (unsigned)(S * a.Hr * a.Wr * TEX) + (unsigned)(g * R);
        const float rcpW = fast_rcp((float)(a.Wr - 1)), rcpH = fast_rcp((float)(a.Hr - 1));
        const int view_stride = a.Hr * a.Wr * TEX;     // < 2^31 floats (checked by the launcher)
        const float* camb = cam + (long long)b * S * kCamStride;
        const float tcx = tcen[b * 3], tcy = tcen[b * 3 + 1], tcz = tcen[b * 3 + 2];

        float Tacc = 1.f;                 // transmittance, raw2outputs utils.py:588-589
        float wk[8];                      // per-sample weights (Ns <= 8)
        float rgbacc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rgbacc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) wk[k] = 0.f;

#pragma unroll 1
        for (int k = 0; k < Ns; ++k) {
            // Keep the MFMA A operands (weights) in LDS: without this barrier LICM hoists all ~155 LDS reads
            // out of the sample loop into registers and the kernel drops to 1 wave/SIMD with spills.
            asm volatile("" ::: "memory");
            // ---------- sample placement (utils.py:425-436) ----------
            float tk = (Ns == 1) ? 0.5f : linspace01(k, Ns);
            float z = rn + (rf - rn) * tk;
            float zz = a.depth_inv ? fast_rcp(clamp_min(z, 1e-6f)) : z;
            float X = ox + dx * zz, Y = oy + dy * zz, Z = oz + dz * zz;
            float dn = a.depth_inv ? (vn - z) * fast_rcp(clamp_min(vn - vf, 1e-6f))
                                   : (z - vn) * fast_rcp(clamp_min(vf - vn, 1e-6f));

            // ---------- voxel feature: trilinear, zeros padding (utils.py:457) ----------
            // Per axis: two corner indices clamped for addressing, corner weights zeroed outside the volume
            // (zeros padding); the 8 taps are products/sums of those.  Non-finite coordinates -> all-zero taps.
            float vox[2] = {0.f, 0.f};
            {
                float ix = gs_unnorm(gxv, a.w), iy = gs_unnorm(gyv, a.h), iz = gs_unnorm(dn * 2.f - 1.f, a.D);
                ix = fabsf(ix) < 1e8f ? ix : -10.f;
                iy = fabsf(iy) < 1e8f ? iy : -10.f;
                iz = fabsf(iz) < 1e8f ? iz : -10.f;
                const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
                const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
                float wx[2] = {(fx + 1.f) - ix, ix - fx}, wy[2] = {(fy + 1.f) - iy, iy - fy}, wz[2] = {(fz + 1.f) - iz, iz - fz};
                int xo[2], yo[2], zo[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int xx = x0 + c, yy = y0 + c, zc = z0 + c;
                    wx[c] = (unsigned)xx < (unsigned)a.w ? wx[c] : 0.f;
                    wy[c] = (unsigned)yy < (unsigned)a.h ? wy[c] : 0.f;
                    wz[c] = (unsigned)zc < (unsigned)a.D ? wz[c] : 0.f;
                    xo[c] = min(max(xx, 0), a.w - 1);
                    yo[c] = mul24(min(max(yy, 0), a.h - 1), a.w);
                    zo[c] = mul24(min(max(zc, 0), a.D - 1), a.h * a.w);
                }
                float wxy[4];                         // (wx*wy)*wz: ATen's association
#pragma unroll
                for (int c = 0; c < 4; ++c) wxy[c] = wx[c & 1] * wy[c >> 1];
#pragma unroll
                for (int c = 0; c < 8; ++c) {           // order tnw,tne,tsw,tse,bnw,bne,bsw,bse
                    const float wgt = wxy[c & 3] * wz[c >> 2];
                    const unsigned vo = (unsigned)(zo[c >> 2] + yo[(c >> 1) & 1] + xo[c & 1]) * 8u + voff;
                    const float2 t = *reinterpret_cast<const float2*>(a.vol + vo);
                    vox[0] += t.x * wgt;
                    vox[1] += t.y * wgt;
                }
            }

            // ---------- per-view image features + direction code (utils.py:698-720) ----------
            float x[S][R], dsel[S];
            // (Tried: each lane group projecting ONE view and handing taps/direction round with lane broadcasts —
            // 170 fewer VALU per 16 points but 36 ds_bpermute on the critical path: 201 -> 215 us.  Not kept.)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float* c = camb + s * kCamStride;
                float cx = X * c[0] + Y * c[1] + Z * c[2] + c[3];
                float cy = X * c[4] + Y * c[5] + Z * c[6] + c[7];
                float cz = X * c[8] + Y * c[9] + Z * c[10] + c[11];
                float px = cx * c[12] + cy * c[13] + cz * c[14];
                float py = cx * c[15] + cy * c[16] + cz * c[17];
                float pz = cx * c[18] + cy * c[19] + cz * c[20];
                float rz = fast_rcp(clamp_min(pz, 1e-6f));
                float gx = ((px * rz) * rcpW) * 2.f - 1.f, gy = ((py * rz) * rcpH) * 2.f - 1.f;
                Taps2 t = gs_taps2<true>(gs_unnorm(gx, a.Wr), gs_unnorm(gy, a.Hr), a.Wr, a.Hr);
                const unsigned tb = toff + (unsigned)(s * view_stride);
                const int r0 = mul24(t.y0, a.Wr), r1 = mul24(t.y1, a.Wr);
                const float* p00 = a.tex + (tb + (unsigned)mul24(r0 + t.x0, TEX));
                const float* p01 = a.tex + (tb + (unsigned)mul24(r0 + t.x1, TEX));
                const float* p10 = a.tex + (tb + (unsigned)mul24(r1 + t.x0, TEX));
                const float* p11 = a.tex + (tb + (unsigned)mul24(r1 + t.x1, TEX));
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float acc = p00[r] * t.w00;
                    acc += p01[r] * t.w01;
                    acc += p10[r] * t.w10;
                    acc += p11[r] * t.w11;
                    x[s][r] = acc;
                }
                // direction code
                float tx = X - tcx, ty = Y - tcy, tz = Z - tcz;
                float sx = X - c[21], sy = Y - c[22], sz = Z - c[23];
                float tir = fast_rcp(fast_sqrt(tx * tx + ty * ty + tz * tz) + 1e-6f);
                float sir = fast_rcp(fast_sqrt(sx * sx + sy * sy + sz * sz) + 1e-6f);
                tx *= tir; ty *= tir; tz *= tir;
                sx *= sir; sy *= sir; sz *= sir;
                float ex = tx - sx, ey = ty - sy, ez = tz - sz;
                float eir = fast_rcp(fmaxf(fast_sqrt(ex * ex + ey * ey + ez * ez), 1e-6f));
                float dot = tx * sx + ty * sy + tz * sz;
                dsel[s] = g == 0 ? ex * eir : (g == 1 ? ey * eir : (g == 2 ? ez * eir : dot));
            }


            // ---------- Agg (nerf.py:74-89) ----------
            float av[S][R];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f32x4 va[TR];
#pragma unroll
                for (int t = 0; t < TR; ++t) {
                    va[t] = lds4(wl + L.viewb + t * 16 + 4 * g);
                    va[t] = ENERF_MFMA(wlane[L.view + t * 64], dsel[s], va[t]);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) av[s][r] = x[s][r] + relu1(va[r >> 2][r & 3]);
            }
            float var[R], mean[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float m = 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s) m += av[s][r];
                m *= (1.f / (float)S);
                float q = 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s) { float d = av[s][r] - m; q += d * d; }
                mean[r] = m;
                var[r] = q * (1.f / (float)(S - 1));              // unbiased, nerf.py:82
            }
            f32x4 P[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) P[u] = lds4(wl + L.globb + u * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    P[u] = ENERF_MFMA(wlane[L.glob + (((1 * R + r) * 2 + u) << 6)], var[r], P[u]);
                    P[u] = ENERF_MFMA(wlane[L.glob + (((2 * R + r) * 2 + u) << 6)], mean[r], P[u]);
                }
            f32x4 gf[S][2];
            float aw[S];
            const f32x4 aggw0 = lds4(wl + L.aggw + 4 * g), aggw1 = lds4(wl + L.aggw + 16 + 4 * g);
            const float aggb = wl[L.aggw + 32];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                gf[s][0] = P[0]; gf[s][1] = P[1];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        gf[s][u] = ENERF_MFMA(wlane[L.glob + (((0 * R + r) * 2 + u) << 6)], av[s][r], gf[s][u]);
                gf[s][0] = relu4(gf[s][0]); gf[s][1] = relu4(gf[s][1]);
                float part = dot4(gf[s][1], aggw1, dot4(gf[s][0], aggw0, 0.f));
                aw[s] = relu1(group_sum(part) + aggb);
            }
            {   // softmax over views
                float m = aw[0];
#pragma unroll
                for (int s = 1; s < S; ++s) m = fmaxf(m, aw[s]);
                float se = 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s) { aw[s] = fast_exp(aw[s] - m); se += aw[s]; }
                se = fast_rcp(se);
#pragma unroll
                for (int s = 0; s < S; ++s) aw[s] *= se;
            }
            f32x4 G[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                G[u] = gf[0][u] * aw[0];
#pragma unroll
                for (int s = 1; s < S; ++s) G[u] += gf[s][u] * aw[s];
            }
            f32x4 agg = lds4(wl + L.fcb + 4 * g);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) agg = ENERF_MFMA(wlane[L.fc + ((u * 4 + r) << 6)], G[u][r], agg);
            agg = relu4(agg);

            // ---------- NeRF trunk (nerf.py:33-37) ----------
            f32x4 hid[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) hid[v] = lds4(wl + L.lr0b + v * 16 + 4 * g);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                float bop = ks < 2 ? vox[ks] : agg[ks - 2];
#pragma unroll
                for (int v = 0; v < 4; ++v) hid[v] = ENERF_MFMA(wlane[L.lr0 + ((ks * 4 + v) << 6)], bop, hid[v]);
            }
            float sig = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                hid[v] = relu4(hid[v]);
                sig = dot4(hid[v], lds4(wl + L.sigma + v * 16 + 4 * g), sig);
            }
            sig = group_sum(sig) + wl[L.sigma + 64];
            sig = sig > 20.f ? sig : log1pf(expf(sig));            // nn.Softplus(beta=1, threshold=20)

            // ---------- colour head (nerf.py:38-42) ----------
            f32x4 P2[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) P2[v] = lds4(wl + L.c0b + v * 16 + 4 * g);
#pragma unroll
            for (int ks = 0; ks < 22; ++ks) {
                float bop = ks < 16 ? hid[ks >> 2][ks & 3] : (ks < 18 ? vox[ks - 16] : agg[ks - 18]);
#pragma unroll
                for (int v = 0; v < 4; ++v) P2[v] = ENERF_MFMA(wlane[L.c0p + ((ks * 4 + v) << 6)], bop, P2[v]);
            }
            float cl[S];
            const float c2b = wl[L.col2 + 64];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f32x4 cc[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) cc[v] = P2[v];
#pragma unroll
                for (int ks = 0; ks <= R; ++ks) {
                    float bop = ks < R ? x[s][ks < R ? ks : 0] : dsel[s];
#pragma unroll
                    for (int v = 0; v < 4; ++v) cc[v] = ENERF_MFMA(wlane[L.c0v + ((ks * 4 + v) << 6)], bop, cc[v]);
                }
                float part = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) part = dot4(relu4(cc[v]), lds4(wl + L.col2 + v * 16 + 4 * g), part);
                cl[s] = relu1(group_sum(part) + c2b);
            }
            {   // softmax over views (nerf.py:41)
                float m = cl[0];
#pragma unroll
                for (int s = 1; s < S; ++s) m = fmaxf(m, cl[s]);
                float se = 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s) { cl[s] = fast_exp(cl[s] - m); se += cl[s]; }
                se = fast_rcp(se);
#pragma unroll
                for (int s = 0; s < S; ++s) cl[s] *= se;
            }

            // ---------- compositing step (utils.py:584-592) ----------
            float alpha = 1.f - fast_exp(-sig);
            float wgt = alpha * Tacc;
            Tacc *= (1.f - alpha + 1e-10f);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float col = x[0][r] * cl[0];
#pragma unroll
                for (int s = 1; s < S; ++s) col += x[s][r] * cl[s];
                rgbacc[r] += wgt * col;
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (kk == k) wk[kk] = wgt;
        }

        // ---------- depth from softmaxed weights (utils.py:593-595) + stores ----------
        float m = wk[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) if (k < Ns) m = fmaxf(m, wk[k]);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < Ns) { wk[k] = fast_exp(wk[k] - m); se += wk[k]; }
        se = fast_rcp(se);
        float depth = 0.f, accw = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < Ns) {
                wk[k] *= se;
                float tk = (Ns == 1) ? 0.5f : linspace01(k, Ns);
                depth += wk[k] * (rn + (rf - rn) * tk);
                accw += wk[k];
            }
        if (rok) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int c = g * R + r - (a.F - 3);
                if (c >= 0 && c < 3) {
                    float v = rgbacc[r];
                    if (a.white_bkgd) v += 1.f - accw;              // utils.py:599-601
                    if (!scatter) a.rgb[ray * 3 + c] = v;
                    else if (nrays > 1) a.rgb[rr * 3 + c] = v;      // network_human.py:105-106: rgb[mask] = ..., if mask.sum() > 1
                }
            }
            if (g == 0) {
                a.depth[ray] = depth;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < Ns) a.weights[ray * Ns + k] = wk[k];
            }
        }
    }
}

template <int R, int OCC>
static int dispatch_s(const RenderArgs& a, unsigned grid, size_t shmem, hipStream_t st) {
    switch (a.S) {
        case 2: ENERF_LAUNCH((k_render_rays<R, 2, OCC>), grid, 256, shmem, st, a); return 0;
        case 3: ENERF_LAUNCH((k_render_rays<R, 3, OCC>), grid, 256, shmem, st, a); return 0;
        case 4: ENERF_LAUNCH((k_render_rays<R, 4, OCC>), grid, 256, shmem, st, a); return 0;
        default: return -3;
    }
}
int launch_render_rays(const RenderArgs& a, hipStream_t st) {
    if (a.n_samples < 1 || a.n_samples > 8) return -1;
    const int R = (a.F + 3) / 4;
    size_t shmem = ((size_t)nerf_layout(a.F).total + (size_t)a.B * a.S * kCamStride + (size_t)a.B * 3) * sizeof(float);
    if (shmem > 64 * 1024) return -2;
    // 32-bit element offsets and 24-bit index multiplies inside the kernel
    if ((long long)a.B * a.S * a.Hr * a.Wr * 4 * R >= (1LL << 32) || (long long)a.B * a.D * a.h * a.w * 8 >= (1LL << 32) ||
        (long long)a.Hr * a.Wr >= (1LL << 23) || (long long)a.h * a.w >= (1LL << 23) || a.D >= (1 << 23))
        return -5;
    long long ntiles = cdivl((long long)a.B * a.N, 16);
    long long blocks = cdivl(ntiles, 4);
    // Persistent waves: the weight image (40-56 KB) is staged into LDS once per block, so launch only as
    // many blocks as are co-resident (OCC per CU x 256 CUs) and let each wave stride over ray tiles.
    // enerf_options_t.render_blocks_per_cu: 3 blocks/CU (168 VGPR, 12 spilled) measured 4 % faster than 2 for one frame
    // at a time; 2 (no spills) wins when several frames share the matrix pipes.
    const Options o = resolve_options(a.options);
    const int occ = o.render_blocks_per_cu == 2 ? 2 : 3;
    const long long resident = (long long)device_cu_count() * occ;
    unsigned grid = (unsigned)(blocks < resident ? blocks : resident);
    if (grid == 0) return 0;
    if (R == 3) return occ == 3 ? dispatch_s<3, 3>(a, grid, shmem, st) : dispatch_s<3, 2>(a, grid, shmem, st);   // C = 8
    if (R == 9) return dispatch_s<9, 2>(a, grid, shmem, st);                                                   // C = 32
    return -4;
}

}  // namespace enerf
