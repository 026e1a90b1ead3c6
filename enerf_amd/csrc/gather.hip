// gather.hip — the render-side feature fetches for TRAINING, forward and backward:
//   get_img_feat (utils.py:689-722): per (point, view) projection into the source view, bilinear sample (border padding,
//     align_corners) of the texel [features | rgb], and the 4-float direction code;
//   get_vox_feat (utils.py:456-458): trilinear sample (zeros padding, align_corners) of the 8-channel feature volume.
// The inference path does all this inside k_render_rays and never materialises it; in training the per-(point, view)
// rows x (B,P,S,F+4) and vox (B,P,8) are the inputs of the fused MLP kernels (mlp_train.hip), so they are written once
// here.  The backward takes d x, d vox and produces: the scatter-add into the texel / volume gradients (fp32 atomics) and
// the gradients w.r.t. the sample position xyz (through the bilinear coordinates AND the direction code) and w.r.t. the
// normalised depth coordinate — which is how the rendering loss reaches depth/std of the cascade level.
// Forward: one thread per (point, view); view 0's thread also fetches the point's voxel feature.  Backward: 16 lanes per
// (point, view), lane = channel.  Gather / L2-atomic bound.
#include "kernels.h"

namespace enerf {

struct GatherArgs {
    const float *xyz, *dn, *uv;          // (B,P,3), (B,P), (B,P,2) raw pixel coordinates of the ray (u, v)
    const float *tex, *vol;              // (B,S,Hr,Wr,F) channels-last texels (F = C+3), (B,D,h,w,8) channels-last volume
    const float *cam, *tcen;             // (B,S,16): M = K'E33 (9) | K't (3) | camera centre (3) | 0 ; (B,4) target centre
    float *x, *vox;                      // forward outputs (B,P,S,F+4), (B,P,8)
    const float *g_x, *g_vox;            // backward inputs
    float *g_tex, *g_vol, *g_xyz, *g_dn; // backward outputs (g_tex, g_vol zeroed by the C entry; g_xyz (B,P,3), g_dn (B,P))
    int B, S, F, Hr, Wr, D, h, w;
    long long P;
    int ray_w, n_samples;                // raster hints of the backward (enerf_gather_args_t)
};

struct ViewGeom {
    float px, py, pz, zc, ix, iy;        // projection, clamped z, pixel coordinates
    int x0, y0, x1, y1;
    float w00, w01, w10, w11, tx1, ty1;
    bool gx_on, gy_on;                   // border padding: coordinate gradient is zero when clipped
    float tdx, tdy, tdz, sdx, sdy, sdz, nt, ns, ex, ey, ez, ne;   // direction-code intermediates
    float dir[4];
};

__device__ __forceinline__ ViewGeom view_geom(const float* c, const float* tc, float X, float Y, float Z, int Wr, int Hr) {
    ViewGeom q;
    q.px = X * c[0] + Y * c[1] + Z * c[2] + c[9];
    q.py = X * c[3] + Y * c[4] + Z * c[5] + c[10];
    q.pz = X * c[6] + Y * c[7] + Z * c[8] + c[11];
    q.zc = clamp_min(q.pz, 1e-6f);
    float ix = q.px / q.zc, iy = q.py / q.zc;      // grid = ix/(W-1)*2-1 -> unnormalised back to ix (align_corners)
    q.gx_on = ix > 0.f && ix < (float)(Wr - 1);
    q.gy_on = iy > 0.f && iy < (float)(Hr - 1);
    ix = fminf((float)(Wr - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(Hr - 1), fmaxf(iy, 0.f));
    q.ix = ix; q.iy = iy;
    const float fx = floorf(ix), fy = floorf(iy);
    q.x0 = (int)fx; q.y0 = (int)fy;
    q.x1 = q.x0 + 1 < Wr ? q.x0 + 1 : Wr - 1;
    q.y1 = q.y0 + 1 < Hr ? q.y0 + 1 : Hr - 1;
    q.tx1 = ix - fx; q.ty1 = iy - fy;
    const float tx0 = (fx + 1.f) - ix, ty0 = (fy + 1.f) - iy;
    q.w00 = tx0 * ty0; q.w01 = q.tx1 * ty0; q.w10 = tx0 * q.ty1; q.w11 = q.tx1 * q.ty1;
    // direction code (utils.py:707-720)
    q.tdx = X - tc[0]; q.tdy = Y - tc[1]; q.tdz = Z - tc[2];
    q.sdx = X - c[12]; q.sdy = Y - c[13]; q.sdz = Z - c[14];
    q.nt = sqrtf(q.tdx * q.tdx + q.tdy * q.tdy + q.tdz * q.tdz);
    q.ns = sqrtf(q.sdx * q.sdx + q.sdy * q.sdy + q.sdz * q.sdz);
    const float it = 1.f / (q.nt + 1e-6f), is = 1.f / (q.ns + 1e-6f);
    const float tx = q.tdx * it, ty = q.tdy * it, tz = q.tdz * it, sx = q.sdx * is, sy = q.sdy * is, sz = q.sdz * is;
    q.ex = tx - sx; q.ey = ty - sy; q.ez = tz - sz;
    q.ne = sqrtf(q.ex * q.ex + q.ey * q.ey + q.ez * q.ez);
    const float im = 1.f / fmaxf(q.ne, 1e-6f);
    q.dir[0] = q.ex * im; q.dir[1] = q.ey * im; q.dir[2] = q.ez * im; q.dir[3] = tx * sx + ty * sy + tz * sz;
    return q;
}

struct VoxGeom {
    int xo[2], yo[2], zo[2];
    float wx[2], wy[2], wz[2];
    bool vx[2], vy[2], vz[2];
};
__device__ __forceinline__ VoxGeom vox_geom(float u, float v, float dn, int Wr, int Hr, int D, int h, int w) {
    VoxGeom q;
    // network.py:36-38 then utils.py:457: grid = (u/(Wr-1), v/(Hr-1), dn)*2-1, unnormalised with align_corners
    float ix = gs_unnorm((u / (float)(Wr - 1)) * 2.f - 1.f, w), iy = gs_unnorm((v / (float)(Hr - 1)) * 2.f - 1.f, h);
    float iz = gs_unnorm(dn * 2.f - 1.f, D);
    ix = fabsf(ix) < 1e8f ? ix : -10.f; iy = fabsf(iy) < 1e8f ? iy : -10.f; iz = fabsf(iz) < 1e8f ? iz : -10.f;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    q.wx[0] = (fx + 1.f) - ix; q.wx[1] = ix - fx; q.wy[0] = (fy + 1.f) - iy; q.wy[1] = iy - fy;
    q.wz[0] = (fz + 1.f) - iz; q.wz[1] = iz - fz;
    for (int c = 0; c < 2; ++c) {
        q.vx[c] = (unsigned)(x0 + c) < (unsigned)w; q.vy[c] = (unsigned)(y0 + c) < (unsigned)h; q.vz[c] = (unsigned)(z0 + c) < (unsigned)D;
        q.xo[c] = min(max(x0 + c, 0), w - 1); q.yo[c] = min(max(y0 + c, 0), h - 1); q.zo[c] = min(max(z0 + c, 0), D - 1);
    }
    return q;
}

__global__ __launch_bounds__(256) void k_gather_fwd(GatherArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.B * a.P * a.S;
    if (i >= total) return;
    const int s = (int)(i % a.S);
    const long long bp = i / a.S;                          // b * P + p
    const int b = (int)(bp / a.P);
    const int F = a.F, XW = F + 4;
    const float X = a.xyz[bp * 3], Y = a.xyz[bp * 3 + 1], Z = a.xyz[bp * 3 + 2];
    const float* c = a.cam + ((long long)b * a.S + s) * 16;
    const ViewGeom q = view_geom(c, a.tcen + b * 4, X, Y, Z, a.Wr, a.Hr);
    const long long img = ((long long)b * a.S + s) * a.Hr * a.Wr;
    const long long o00 = (img + (long long)q.y0 * a.Wr + q.x0) * F, o01 = (img + (long long)q.y0 * a.Wr + q.x1) * F;
    const long long o10 = (img + (long long)q.y1 * a.Wr + q.x0) * F, o11 = (img + (long long)q.y1 * a.Wr + q.x1) * F;
    float* xo = a.x + i * XW;
    for (int ch = 0; ch < F; ++ch)
        xo[ch] = a.tex[o00 + ch] * q.w00 + a.tex[o01 + ch] * q.w01 + a.tex[o10 + ch] * q.w10 + a.tex[o11 + ch] * q.w11;
    for (int k = 0; k < 4; ++k) xo[F + k] = q.dir[k];
    if (s != 0) return;
    const VoxGeom v = vox_geom(a.uv[bp * 2], a.uv[bp * 2 + 1], a.dn[bp], a.Wr, a.Hr, a.D, a.h, a.w);
    const long long vb = (long long)b * a.D * a.h * a.w;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 8; ++t) {
        const int cx = t & 1, cy = (t >> 1) & 1, cz = t >> 2;
        if (!(v.vx[cx] && v.vy[cy] && v.vz[cz])) continue;
        const long long o = (vb + ((long long)v.zo[cz] * a.h + v.yo[cy]) * a.w + v.xo[cx]) * 8;
        const float wgt = v.wx[cx] * v.wy[cy] * v.wz[cz];
        for (int ch = 0; ch < 8; ++ch) acc[ch] += a.vol[o + ch] * wgt;
    }
    for (int ch = 0; ch < 8; ++ch) a.vox[bp * 8 + ch] = acc[ch];
}

// Forward, the form the C entry launches: a wave owns 64 consecutive points.  Per view, lane l computes the geometry of point l
// ONCE (the thread-per-(point, view) kernel above then walks 4 F scattered 4-byte loads per thread: 64 different cache lines per
// load instruction); the 64 points are then fetched one per pass with lane = (tap, channel): 4 taps x 16 channels (64 lanes, the
// four taps' weighted values summed across the lane groups by register permutes) for the texel, 8 taps x 8 channels for the
// volume — every load and store instruction touches whole texels.  No LDS, no barriers.
#ifndef ENERF_GFW_U
#define ENERF_GFW_U 4                  // points fetched per iteration of the pass loops (all their loads in flight)
#endif
template <int NCH>                      // channel rounds: ceil(F / 16)
__global__ __launch_bounds__(256) void k_gather_fwd_w(GatherArgs a) {
    constexpr int GU = ENERF_GFW_U;
    const int F = a.F, XW = F + 4, l64 = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;     // first point (b * P + p) of the wave
    const long long total = (long long)a.B * a.P;
    if (wave0 >= total) return;
    const long long bp_raw = wave0 + l64;
    const bool live = bp_raw < total;
    const long long bp = live ? bp_raw : total - 1;
    const int b = (int)(bp / a.P);
    const float X = a.xyz[bp * 3], Y = a.xyz[bp * 3 + 1], Z = a.xyz[bp * 3 + 2];
    const int tap = l64 >> 4, chl = l64 & 15, vtap = l64 >> 3, vch = l64 & 7;
    const int npass = (int)(total - wave0 < 64 ? total - wave0 : 64);
    // Both pass loops below fetch ENERF_GFW_U points per iteration, all their loads requested before the first is reduced (round 6: one point
    // per iteration was a chain of L2 round trips — load, wait, reduce, store — 64 x (1 + S) times per wave: 300 us at level 1), with
    // 32-bit element offsets (the launcher checks the tensors are below 2^32 elements).
    // ---- the volume sample (once per point) ----
    {
        const VoxGeom v = vox_geom(a.uv[bp * 2], a.uv[bp * 2 + 1], a.dn[bp], a.Wr, a.Hr, a.D, a.h, a.w);
        const int valid = (int)v.vx[0] | ((int)v.vx[1] << 1) | ((int)v.vy[0] << 2) | ((int)v.vy[1] << 3) | ((int)v.vz[0] << 4) | ((int)v.vz[1] << 5);
        const int r0 = v.xo[0] | (v.xo[1] << 16), r1 = v.yo[0] | (v.yo[1] << 16), r2 = v.zo[0] | (v.zo[1] << 8) | (valid << 16), rb = b;
        const int r3 = __float_as_int(v.wx[0]), r4 = __float_as_int(v.wx[1]), r5 = __float_as_int(v.wy[0]), r6 = __float_as_int(v.wy[1]);
        const int r7 = __float_as_int(v.wz[0]), r8 = __float_as_int(v.wz[1]);
        const int cx = vtap & 1, cy = (vtap >> 1) & 1, cz = vtap >> 2;
        float* vrow = a.vox + wave0 * 8 + vch;
        for (int j0 = 0; j0 < npass; j0 += GU) {
            float val[GU], wgt[GU];
            bool onm[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int j = min(j0 + u, npass - 1);
                const int w0 = __builtin_amdgcn_readlane(r0, j), w1 = __builtin_amdgcn_readlane(r1, j), w2 = __builtin_amdgcn_readlane(r2, j);
                const int bj = __builtin_amdgcn_readlane(rb, j);
                // (every lane executes every v_readlane — they are wave-level operations; the selects come after)
                const int x3 = __builtin_amdgcn_readlane(r3, j), x4 = __builtin_amdgcn_readlane(r4, j), x5 = __builtin_amdgcn_readlane(r5, j);
                const int x6 = __builtin_amdgcn_readlane(r6, j), x7 = __builtin_amdgcn_readlane(r7, j), x8 = __builtin_amdgcn_readlane(r8, j);
                const float wx = __int_as_float(cx ? x4 : x3), wy = __int_as_float(cy ? x6 : x5), wz = __int_as_float(cz ? x8 : x7);
                const int xo = cx ? (w0 >> 16) & 0xffff : w0 & 0xffff, yo = cy ? (w1 >> 16) & 0xffff : w1 & 0xffff;
                const int zo = cz ? (w2 >> 8) & 0xff : w2 & 0xff, vld = w2 >> 16;
                const bool on = ((vld >> cx) & 1) && ((vld >> (2 + cy)) & 1) && ((vld >> (4 + cz)) & 1);
                val[u] = a.vol[(((unsigned)(bj * a.D + zo) * (unsigned)a.h + (unsigned)yo) * (unsigned)a.w + (unsigned)xo) * 8u + (unsigned)vch];
                wgt[u] = wx * wy * wz; onm[u] = on;
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                if (j0 + u >= npass) break;                       // (wave-uniform)
                const float acc = group_sum4(add_xor8(onm[u] ? val[u] * wgt[u] : 0.f));      // over the 8 taps: lane bits 3 | 4, 5
                if (l64 < 8) vrow[(j0 + u) * 8] = acc;
            }
        }
    }
    // ---- the texel of every view ----
    const unsigned img_texels = (unsigned)(a.Hr * a.Wr);
    for (int s = 0; s < a.S; ++s) {
        const float* c = a.cam + ((long long)b * a.S + s) * 16;
        const ViewGeom q = view_geom(c, a.tcen + b * 4, X, Y, Z, a.Wr, a.Hr);
        if (live) {
            float* xo = a.x + (bp * a.S + s) * XW + F;
            xo[0] = q.dir[0]; xo[1] = q.dir[1]; xo[2] = q.dir[2]; xo[3] = q.dir[3];
        }
        const int r0 = q.x0 | (q.x1 << 16), r1 = q.y0 | (q.y1 << 16), rb = (int)((unsigned)(b * a.S + s) * img_texels);   // first texel of the image
        const int r2 = __float_as_int(q.w00), r3 = __float_as_int(q.w01), r4 = __float_as_int(q.w10), r5 = __float_as_int(q.w11);
        float* xbase = a.x + (wave0 * a.S + s) * XW;
        const unsigned xstep = (unsigned)(a.S * XW);
        for (int j0 = 0; j0 < npass; j0 += GU) {
            float v[GU][NCH], w[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int j = min(j0 + u, npass - 1);
                const int gxw = __builtin_amdgcn_readlane(r0, j), gyw = __builtin_amdgcn_readlane(r1, j), tj = __builtin_amdgcn_readlane(rb, j);
                const int x2 = __builtin_amdgcn_readlane(r2, j), x3 = __builtin_amdgcn_readlane(r3, j);   // (all lanes, then select)
                const int x4 = __builtin_amdgcn_readlane(r4, j), x5 = __builtin_amdgcn_readlane(r5, j);
                w[u] = __int_as_float((tap & 2) ? ((tap & 1) ? x5 : x4) : ((tap & 1) ? x3 : x2));
                const int x = (tap & 1) ? (gxw >> 16) & 0xffff : gxw & 0xffff, y = (tap & 2) ? (gyw >> 16) & 0xffff : gyw & 0xffff;
                const unsigned t = ((unsigned)tj + (unsigned)y * (unsigned)a.Wr + (unsigned)x) * (unsigned)F;
#pragma unroll
                for (int k = 0; k < NCH; ++k) v[u][k] = a.tex[t + (unsigned)min(chl + 16 * k, F - 1)];         // clamped: always loads
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                if (j0 + u >= npass) break;                       // (wave-uniform)
                float* xrow = xbase + (unsigned)(j0 + u) * xstep;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int ch = chl + 16 * k;
                    const float acc = group_sum4(v[u][k] * w[u]);
                    if (l64 < 16 && ch < F) xrow[ch] = acc;
                }
            }
        }
    }
}

__device__ __forceinline__ float sum16(float v) {          // sum over the 16 lanes of a group (lane ^ 1, 2, 4, 8)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// Backward: 16 lanes per (point, view), lane = channel (c, c+16, c+32): the scatter-add of a tap is ONE coalesced run of F
// floats (the thread-per-point form issued 64 scattered cache lines per atomic instruction and ran 2.5x slower than the
// library grid_sampler backward).  The body is shared by the two kernels below, which differ in WHERE a contribution is added:
//   tex_add(y, x, element offset of the texel, channel, value), vol_add(z, y, x, element offset of the voxel + channel, value),
//   xyz_add(b * P + p, gX, gY, gZ).
template <class TexAdd, class VolAdd, class XyzAdd>
__device__ __forceinline__ void gather_bwd_pair(const GatherArgs& a, long long bp, int b, int s, bool live, int lane, TexAdd tex_add,
                                                VolAdd vol_add, XyzAdd xyz_add) {
    const int F = a.F, XW = F + 4;
    const long long i = bp * a.S + s;
    const float X = a.xyz[bp * 3], Y = a.xyz[bp * 3 + 1], Z = a.xyz[bp * 3 + 2];
    const float* c = a.cam + ((long long)b * a.S + s) * 16;
    const ViewGeom q = view_geom(c, a.tcen + b * 4, X, Y, Z, a.Wr, a.Hr);
    const long long img = ((long long)b * a.S + s) * a.Hr * a.Wr;
    const long long o00 = (img + (long long)q.y0 * a.Wr + q.x0) * F, o01 = (img + (long long)q.y0 * a.Wr + q.x1) * F;
    const long long o10 = (img + (long long)q.y1 * a.Wr + q.x0) * F, o11 = (img + (long long)q.y1 * a.Wr + q.x1) * F;
    const float* gx = a.g_x + i * XW;
    float gix = 0.f, giy = 0.f;
    const float tx0 = 1.f - q.tx1, ty0 = 1.f - q.ty1;
    for (int ch = lane; live && ch < F; ch += 16) {
        const float g = gx[ch];
        const float v00 = a.tex[o00 + ch], v01 = a.tex[o01 + ch], v10 = a.tex[o10 + ch], v11 = a.tex[o11 + ch];
        tex_add(q.y0, q.x0, o00, ch, q.w00 * g); tex_add(q.y0, q.x1, o01, ch, q.w01 * g);
        tex_add(q.y1, q.x0, o10, ch, q.w10 * g); tex_add(q.y1, q.x1, o11, ch, q.w11 * g);
        gix += g * ((v01 - v00) * ty0 + (v11 - v10) * q.ty1);
        giy += g * ((v10 - v00) * tx0 + (v11 - v01) * q.tx1);
    }
    gix = sum16(gix); giy = sum16(giy);
    // ---- voxel feature of the point: lanes 0..7 of view 0's group own one channel each ----
    float giz = 0.f;
    if (s == 0 && live) {
        const VoxGeom v = vox_geom(a.uv[bp * 2], a.uv[bp * 2 + 1], a.dn[bp], a.Wr, a.Hr, a.D, a.h, a.w);
        const long long vb = (long long)b * a.D * a.h * a.w;
        if (lane < 8) {
            const float g = a.g_vox[bp * 8 + lane];
            for (int t = 0; t < 8; ++t) {
                const int cx = t & 1, cy = (t >> 1) & 1, cz = t >> 2;
                if (!(v.vx[cx] && v.vy[cy] && v.vz[cz])) continue;
                const long long o = (vb + ((long long)v.zo[cz] * a.h + v.yo[cy]) * a.w + v.xo[cx]) * 8 + lane;
                const float wxy = v.wx[cx] * v.wy[cy];
                vol_add(v.zo[cz], v.yo[cy], v.xo[cx], o, wxy * v.wz[cz] * g);
                giz += (cz ? 1.f : -1.f) * a.vol[o] * wxy * g;
            }
        }
    }
    giz = sum16(giz);
    if (lane != 0 || !live) return;
    if (s == 0) a.g_dn[bp] = giz * (float)(a.D - 1);       // iz = dn (D-1): unnormalise (D-1)/2 x d(2 dn - 1)/d dn
    if (!q.gx_on) gix = 0.f;
    if (!q.gy_on) giy = 0.f;
    // (ix, iy) = p.xy / max(p.z, 1e-6);  p = M X + v
    const float gpx = gix / q.zc, gpy = giy / q.zc;
    const float gpz = q.pz >= 1e-6f ? -(gix * q.px + giy * q.py) / (q.zc * q.zc) : 0.f;
    float gX = c[0] * gpx + c[3] * gpy + c[6] * gpz, gY = c[1] * gpx + c[4] * gpy + c[7] * gpz, gZ = c[2] * gpx + c[5] * gpy + c[8] * gpz;
    // direction code backward
    const float gd0 = gx[F], gd1 = gx[F + 1], gd2 = gx[F + 2], gdot = gx[F + 3];
    const float it = 1.f / (q.nt + 1e-6f), is = 1.f / (q.ns + 1e-6f);
    const float tx = q.tdx * it, ty = q.tdy * it, tz = q.tdz * it, sx = q.sdx * is, sy = q.sdy * is, sz = q.sdz * is;
    float gex, gey, gez;
    if (q.ne > 1e-6f) {
        const float in = 1.f / q.ne, hx = q.ex * in, hy = q.ey * in, hz = q.ez * in, pr = hx * gd0 + hy * gd1 + hz * gd2;
        gex = (gd0 - hx * pr) * in; gey = (gd1 - hy * pr) * in; gez = (gd2 - hz * pr) * in;
    } else { gex = gd0 * 1e6f; gey = gd1 * 1e6f; gez = gd2 * 1e6f; }
    const float gtx = gex + gdot * sx, gty = gey + gdot * sy, gtz = gez + gdot * sz;
    const float gsx = -gex + gdot * tx, gsy = -gey + gdot * ty, gsz = -gez + gdot * tz;
    {   // t_hat = dt / (|dt| + eps)
        const float pr = (q.tdx * gtx + q.tdy * gty + q.tdz * gtz), k = q.nt > 0.f ? pr * it * it / q.nt : 0.f;
        gX += gtx * it - q.tdx * k; gY += gty * it - q.tdy * k; gZ += gtz * it - q.tdz * k;
    }
    {
        const float pr = (q.sdx * gsx + q.sdy * gsy + q.sdz * gsz), k = q.ns > 0.f ? pr * is * is / q.ns : 0.f;
        gX += gsx * is - q.sdx * k; gY += gsy * is - q.sdy * k; gZ += gsz * is - q.sdz * k;
    }
    xyz_add(bp, gX, gY, gZ);
}

// Points in any order: every contribution is a global atomic (bounded by the L2's atomic request rate).
__global__ __launch_bounds__(256) void k_gather_bwd(GatherArgs a) {
    const int lane = threadIdx.x & 15;
    const long long i_raw = (long long)blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
    const long long total = (long long)a.B * a.P * a.S;
    const bool live = i_raw < total;                        // dead groups follow along (uniform shuffles), write nothing
    const long long i = live ? i_raw : total - 1;
    const int s = (int)(i % a.S);
    const long long bp = i / a.S;
    const int b = (int)(bp / a.P);
    gather_bwd_pair(a, bp, b, s, live, lane,
                    [&](int, int, long long o, int ch, float v) { atomic_add_f32(a.g_tex + o + ch, v); },
                    [&](int, int, int, long long o, float v) { atomic_add_f32(a.g_vol + o, v); },
                    [&](long long q, float gX, float gY, float gZ) {
                        atomic_add_f32(a.g_xyz + q * 3, gX); atomic_add_f32(a.g_xyz + q * 3 + 1, gY); atomic_add_f32(a.g_xyz + q * 3 + 2, gZ);
                    });
}

// Raster-ordered rays (enerf_gather_args_t.ray_w / n_samples): a block owns a th x tw tile of rays (x kg samples: <= 256 points)
// of one batch element and walks the source views.  Per view the tile's texel taps land in a compact patch of the view (the
// image of the tile under the view's projection), so they are accumulated in LDS patches [PH][PW][F] whose rows are contiguous
// runs of the gradient map and flushed ONCE with lane = consecutive floats; the same for the trilinear volume taps of view 0
// ([VZ][VH][VW][8]).  A patch's origin is the minimum tap coordinate over its points; a tap that does not fit goes to the
// global map as before.  fp32 atomics cost one L2 request per 64-byte line and instruction (tools/micro/atomic_rate.hip:
// 20.8 G requests/s): ~90 line requests per 64 points and view instead of ~420.
// Three phases per view: (1) one THREAD per point: the view's geometry once (the any-order kernel computes it in all 16 lanes of
// a group), taps + fractions to LDS, patch origin by min-reduction; (2) the scatter and the coordinate gradients' channel sums;
// (3) one thread per point again: the chain rule through projection and direction code, accumulated over the views in
// registers — g_xyz is stored once, no atomics.
// Two forms of phase 2, because LDS float atomics are no way out: ds_add_f32 runs at ~3 cycles PER LANE per CU on gfx950
// (tools/micro/lds_atomic_rate.hip: 133-193 cycles per wave instruction against 12.7 for a plain read-add-write), and an LDS
// instruction costs its issue slot whatever its lane count, so taking turns between the four 16-lane point groups of an
// instruction is as slow (both measured: profiles/r04_gather_bwd_tiled.txt).
//   WP ("wave-private", F <= 16): a wave owns an 8-row x cw-column strip of the tile (64 points) and ITS OWN patches, and handles
//     ONE point per instruction: lane = (tap, channel) for the texel — the four taps of a bilinear footprint are four different
//     texels, duplicates created by the border clamp carry weight 0 and are skipped — and lane = (tap, channel) of the eight
//     trilinear taps for the volume: 64 different words, so a plain read-add-write is exact (a wave's LDS operations execute in
//     order).  The sums over the 64 lanes are register permutes (group_sum4 + a DPP row rotation), not LDS traffic.
//   block-shared (F = 35: four private patches of 35-float texels do not fit): 16 lanes per point, LDS atomics.
struct GatherTile { int th, tw, kg, PW, PH, VW, VH, VZ, tiles_x, tiles_y, groups; };

__device__ __forceinline__ int wave_min_i(int v) {
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) { return row_sum16(group_sum4(v)); }

#ifndef ENERF_GT_ABL
#define ENERF_GT_ABL 0                  // timing ablations (tools/build_variant.py): 1 no scatter passes, 2 passes without adds, 3 no flush, 4 no phase 3
#endif
template <int NCH, bool WP>             // NCH: channel rounds of a 16-lane group: ceil(F / 16)
__global__ __launch_bounds__(256) void k_gather_bwd_tiled(GatherArgs a, GatherTile T) {
    ENERF_DYN_SMEM(float, smem);
    __shared__ int mn[5];                                   // minimum tap x, y of the view; minimum voxel x, y, z (block-shared form)
    __shared__ int geo[WP ? 1 : 256][4];                    // x0 | x1 << 16, y0 | y1 << 16, tx1, ty1 (float bits); WP: in registers
    __shared__ int gpt[WP ? 1 : 256];                       // the point's index p inside its batch element, -1: outside the image
    __shared__ float gij[256][3];                           // d loss / d (ix, iy, iz) of the point for this view
    const int F = a.F, XW = F + 4, Ns = a.n_samples, tid = threadIdx.x, lane = tid & 15, grp = tid >> 4, wave = tid >> 6, l64 = tid & 63;
    const int ntex = T.PH * T.PW * F, nvol = T.VZ * T.VH * T.VW * 8;
    float* ptex = smem + (WP ? wave * (ntex + nvol) : 0);
    float* pvol = ptex + ntex;
    for (int i = tid; i < (ntex + nvol) * (WP ? 4 : 1); i += 256) smem[i] = 0.f;
    int bid = blockIdx.x;
    const int sg = bid % T.groups; bid /= T.groups;
    const int txi = bid % T.tiles_x, tyi = (bid / T.tiles_x) % T.tiles_y, b = bid / (T.tiles_x * T.tiles_y);
    const int rows = (int)(a.P / Ns / a.ray_w);
    const int ry0 = tyi * T.th, rx0 = txi * T.tw;
    const int npts = T.th * T.tw * T.kg;                    // <= 256
    auto point_of = [&](int pi, long long& bp) {            // point pi of the tile -> (inside the image, b * P + p)
        int kk, rx, ry;
        if (WP) {                                           // wave w = pi >> 6 owns columns [w cw, (w + 1) cw) of the tile: cw = tw / 4
            const int l = pi & 63, cw = T.tw >> 2, r = l / T.kg;
            kk = l - r * T.kg; ry = r / cw; rx = (pi >> 6) * cw + (r - ry * cw);
        } else {
            const int r = pi / T.kg;
            kk = pi - r * T.kg; ry = r / T.tw; rx = r - ry * T.tw;
        }
        const int gy = ry0 + ry, gx = rx0 + rx;
        const bool ok = pi < npts && gy < rows && gx < a.ray_w;
        bp = (long long)b * a.P + (ok ? ((long long)gy * a.ray_w + gx) * Ns + sg * T.kg + kk : 0);
        return ok;
    };
    long long bp;
    const bool live = point_of(tid, bp);
    const int rpp = live ? (int)(bp - (long long)b * a.P) : -1;  // (P < 2^31: the C entry)
    if (!WP) gpt[tid] = rpp;
    const float X = a.xyz[bp * 3], Y = a.xyz[bp * 3 + 1], Z = a.xyz[bp * 3 + 2];
    float gX = 0.f, gY = 0.f, gZ = 0.f;
    for (int s = 0; s < a.S; ++s) {
        // ---- (1) geometry of (point, view s), one thread per point ----
        __syncthreads();                                    // the previous view's flush is done
        if (tid < 5) mn[tid] = 0x7fffffff;
        __syncthreads();
        const float* c = a.cam + ((long long)b * a.S + s) * 16;
        int m0, m1, m2 = 0x7fffffff, m3 = m2, m4 = m2;
        int rg0, rg1, rg2, rg3, rv0 = 0, rv1 = 0, rv2 = 0, rv3 = 0, rv4 = 0, rv5 = 0;   // WP: this lane's point, read by v_readlane in phase 2
        {
            const ViewGeom q = view_geom(c, a.tcen + b * 4, X, Y, Z, a.Wr, a.Hr);
            rg0 = q.x0 | (q.x1 << 16); rg1 = q.y0 | (q.y1 << 16); rg2 = __float_as_int(q.tx1); rg3 = __float_as_int(q.ty1);
            if (!WP) { geo[tid][0] = rg0; geo[tid][1] = rg1; geo[tid][2] = rg2; geo[tid][3] = rg3; }
            m0 = live ? q.x0 : 0x7fffffff; m1 = live ? q.y0 : 0x7fffffff;
        }
        if (s == 0) {
            const VoxGeom v = vox_geom(a.uv[bp * 2], a.uv[bp * 2 + 1], a.dn[bp], a.Wr, a.Hr, a.D, a.h, a.w);
            if (live) { m2 = v.xo[0]; m3 = v.yo[0]; m4 = v.zo[0]; }
            if (WP) {
                const int valid = (int)v.vx[0] | ((int)v.vx[1] << 1) | ((int)v.vy[0] << 2) | ((int)v.vy[1] << 3) | ((int)v.vz[0] << 4) | ((int)v.vz[1] << 5);
                rv0 = v.xo[0] | (v.xo[1] << 16); rv1 = v.yo[0] | (v.yo[1] << 16);
                rv2 = v.zo[0] | (v.zo[1] << 8) | (valid << 16);
                rv3 = __float_as_int(v.wx[1]); rv4 = __float_as_int(v.wy[1]); rv5 = __float_as_int(v.wz[1]);
            }
        }
        m0 = wave_min_i(m0); m1 = wave_min_i(m1);
        if (s == 0) { m2 = wave_min_i(m2); m3 = wave_min_i(m3); m4 = wave_min_i(m4); }
        if (!WP && (tid & 63) == 0) {
            atomicMin(&mn[0], m0); atomicMin(&mn[1], m1);
            if (s == 0) { atomicMin(&mn[2], m2); atomicMin(&mn[3], m3); atomicMin(&mn[4], m4); }
        }
        __syncthreads();
        // (the butterfly leaves a wave's minimum in all of its lanes: the wave-private patches take that)
        const int x0p = WP ? m0 : mn[0], y0p = WP ? m1 : mn[1], vx0 = WP ? m2 : mn[2], vy0 = WP ? m3 : mn[3], vz0 = WP ? m4 : mn[4];
        const long long img = ((long long)b * a.S + s) * a.Hr * a.Wr;
        const long long vb = (long long)b * a.D * a.h * a.w;
        // ---- (2) scatter.  Two-stage software pipeline over two NAMED stages (no copies, no conditional fetch: hipcc otherwise
        //      waits vmcnt(0) before every pass' scatter, i.e. for the prefetch it has just issued; the last prefetch re-reads the
        //      last pass): the loads of pass it + 1 are in flight while pass it blends and scatters ----
        if constexpr (WP) {
            struct Stage {
                long long bq, goff, voff; bool ok, von; int slot, vslot; float w, cx, cy, g, v, vwxy, vwz, vsign, gvox, vv;
            };
            const int tap = l64 >> 4, chl = l64 & 15, vtap = l64 >> 3, vch = l64 & 7;
            auto fetch = [&](int j, Stage& P) {
                // (point j of the wave is lane j of phase 1: its geometry comes over v_readlane, not through LDS)
                const int pp = __builtin_amdgcn_readlane(rpp, j);
                P.ok = pp >= 0;
                P.bq = (long long)b * a.P + (P.ok ? pp : 0);
                const int gxw = __builtin_amdgcn_readlane(rg0, j), gyw = __builtin_amdgcn_readlane(rg1, j);
                const float tx1 = __int_as_float(__builtin_amdgcn_readlane(rg2, j)), ty1 = __int_as_float(__builtin_amdgcn_readlane(rg3, j));
                const float tx0 = 1.f - tx1, ty0 = 1.f - ty1;
                // (view_geom forms tx0 as (fx + 1) - ix; 1 - tx1 differs from that by one rounding at most)
                const int x = (tap & 1) ? (gxw >> 16) & 0xffff : gxw & 0xffff, y = (tap & 2) ? (gyw >> 16) & 0xffff : gyw & 0xffff;
                P.w = ((tap & 1) ? tx1 : tx0) * ((tap & 2) ? ty1 : ty0);
                P.cx = ((tap & 1) ? 1.f : -1.f) * ((tap & 2) ? ty1 : ty0);       // d ix: (v01 - v00) ty0 + (v11 - v10) ty1
                P.cy = ((tap & 2) ? 1.f : -1.f) * ((tap & 1) ? tx1 : tx0);       // d iy: (v10 - v00) tx0 + (v11 - v01) tx1
                const unsigned dy = (unsigned)(y - y0p), dx = (unsigned)(x - x0p);
                P.slot = (dy < (unsigned)T.PH && dx < (unsigned)T.PW) ? (int)((dy * T.PW + dx) * F) + chl : -1;
                const int ch = min(chl, F - 1);                                  // clamped: always loads, masked when used
                P.goff = (img + (P.ok ? y * a.Wr + x : 0)) * F + chl;
                P.g = a.g_x[(P.bq * a.S + s) * XW + ch];
                P.v = a.tex[(img + (P.ok ? y * a.Wr + x : 0)) * F + ch];
                if (s == 0) {
                    const int w0 = __builtin_amdgcn_readlane(rv0, j), w1 = __builtin_amdgcn_readlane(rv1, j), w2 = __builtin_amdgcn_readlane(rv2, j);
                    const float wx1 = __int_as_float(__builtin_amdgcn_readlane(rv3, j)), wy1 = __int_as_float(__builtin_amdgcn_readlane(rv4, j));
                    const float wz1 = __int_as_float(__builtin_amdgcn_readlane(rv5, j));
                    const int cx = vtap & 1, cy = (vtap >> 1) & 1, cz = vtap >> 2;
                    const int xo = cx ? (w0 >> 16) & 0xffff : w0 & 0xffff, yo = cy ? (w1 >> 16) & 0xffff : w1 & 0xffff;
                    const int zo = cz ? (w2 >> 8) & 0xff : w2 & 0xff, valid = w2 >> 16;
                    P.von = P.ok && ((valid >> cx) & 1) && ((valid >> (2 + cy)) & 1) && ((valid >> (4 + cz)) & 1);
                    P.vwxy = (cx ? wx1 : 1.f - wx1) * (cy ? wy1 : 1.f - wy1);
                    P.vwz = cz ? wz1 : 1.f - wz1;
                    P.vsign = cz ? 1.f : -1.f;
                    const unsigned dz = (unsigned)(zo - vz0), dyv = (unsigned)(yo - vy0), dxv = (unsigned)(xo - vx0);
                    P.vslot = (dz < (unsigned)T.VZ && dyv < (unsigned)T.VH && dxv < (unsigned)T.VW) ? (int)(((dz * T.VH + dyv) * T.VW + dxv) * 8) + vch : -1;
                    P.voff = (vb + (P.ok ? ((long long)zo * a.h + yo) * a.w + xo : 0)) * 8 + vch;
                    P.gvox = a.g_vox[P.bq * 8 + vch];
                    P.vv = a.vol[P.voff];
                }
            };
            auto process = [&](const Stage& P, int j) {
                const int pj = wave * 64 + j;
                const bool on = P.ok && chl < F;
                const float g = on ? P.g : 0.f;
                if (ENERF_GT_ABL != 2 && on && P.w != 0.f) {                     // (weight 0: among others the clamp's duplicate taps)
                    if (P.slot >= 0) { const float r = ptex[P.slot]; ptex[P.slot] = r + P.w * g; }
                    else atomic_add_f32(a.g_tex + P.goff, P.w * g);
                }
                const float gix = wave_sum64(g * P.v * P.cx), giy = wave_sum64(g * P.v * P.cy);
                float giz = 0.f;
                if (s == 0) {
                    const float val = P.vwxy * P.vwz * P.gvox;
                    if (ENERF_GT_ABL != 2 && P.von) {
                        if (P.vslot >= 0) { const float r = pvol[P.vslot]; pvol[P.vslot] = r + val; }
                        else atomic_add_f32(a.g_vol + P.voff, val);
                    }
                    giz = wave_sum64(P.von ? P.vsign * P.vv * P.vwxy * P.gvox : 0.f);
                }
                if (l64 == 0) { gij[pj][0] = gix; gij[pj][1] = giy; gij[pj][2] = giz; }
            };
            const int passes = ENERF_GT_ABL == 1 ? 0 : 64;
            Stage sa, sb;
            fetch(0, sa);
            for (int it = 0; it < passes; it += 2) {
                fetch(it + 1, sb);
                __builtin_amdgcn_sched_barrier(0);
                process(sa, it);
                fetch(min(it + 2, 63), sa);
                __builtin_amdgcn_sched_barrier(0);
                process(sb, it + 1);
            }
        } else {
            struct Stage {
                long long bq; bool ok; int x0, x1, y0, y1; float tx1, ty1;
                float g[NCH], v00[NCH], v01[NCH], v10[NCH], v11[NCH];
                VoxGeom vg; float gvox, vv[4];                      // view 0 only: the point's volume taps, its d vox channel, tap values
            };
            auto fetch = [&](int it, Stage& P) {
                const int pj = it * 16 + grp;
                const int pp = pj < 256 ? gpt[pj] : -1;
                P.ok = pp >= 0;
                P.bq = (long long)b * a.P + (P.ok ? pp : 0);
                if (s == 0) {
                    // lanes 0-7 / 8-15 of the group: channel = lane & 7, the taps with cz = 0 / 1 (the any-order kernel leaves half of
                    // the group idle); coordinates are clamped, so the four values are always loadable
                    P.gvox = a.g_vox[P.bq * 8 + (lane & 7)];
                    P.vg = vox_geom(a.uv[P.bq * 2], a.uv[P.bq * 2 + 1], a.dn[P.bq], a.Wr, a.Hr, a.D, a.h, a.w);
                    const int zoc = (lane >> 3) ? P.vg.zo[1] : P.vg.zo[0];  // (selects, not a runtime array index: that would be scratch)
#pragma unroll
                    for (int t = 0; t < 4; ++t) P.vv[t] = a.vol[(vb + ((long long)zoc * a.h + P.vg.yo[t >> 1]) * a.w + P.vg.xo[t & 1]) * 8 + (lane & 7)];
                }
                const int gxw = geo[pj & 255][0], gyw = geo[pj & 255][1];
                P.tx1 = __int_as_float(geo[pj & 255][2]); P.ty1 = __int_as_float(geo[pj & 255][3]);
                P.x0 = gxw & 0xffff; P.x1 = (gxw >> 16) & 0xffff; P.y0 = gyw & 0xffff; P.y1 = (gyw >> 16) & 0xffff;
                if (!P.ok) { P.x0 = P.x1 = P.y0 = P.y1 = 0; }
                const float* gx = a.g_x + (P.bq * a.S + s) * XW;
                const long long t00 = (img + P.y0 * a.Wr + P.x0) * F, t01 = (img + P.y0 * a.Wr + P.x1) * F;
                const long long t10 = (img + P.y1 * a.Wr + P.x0) * F, t11 = (img + P.y1 * a.Wr + P.x1) * F;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int ch = min(lane + 16 * k, F - 1);               // clamped: always loads, masked when used
                    P.g[k] = gx[ch];
                    P.v00[k] = a.tex[t00 + ch]; P.v01[k] = a.tex[t01 + ch]; P.v10[k] = a.tex[t10 + ch]; P.v11[k] = a.tex[t11 + ch];
                }
            };
            auto process = [&](const Stage& cur, int it) {
                const int pj = it * 16 + grp;
                const bool ok = cur.ok;
                const int x0 = cur.x0, x1 = cur.x1, y0 = cur.y0, y1 = cur.y1;
                const float tx1 = cur.tx1, ty1 = cur.ty1, tx0 = 1.f - tx1, ty0 = 1.f - ty1;
                const float w00 = tx0 * ty0, w01 = tx1 * ty0, w10 = tx0 * ty1, w11 = tx1 * ty1;
                const int t00 = y0 * a.Wr + x0, t01 = y0 * a.Wr + x1, t10 = y1 * a.Wr + x0, t11 = y1 * a.Wr + x1;
                auto tex_add = [&](int y, int x, int t, int ch, float v) {
                    if (ENERF_GT_ABL == 2) return;
                    const unsigned dy = (unsigned)(y - y0p), dx = (unsigned)(x - x0p);
                    if (dy < (unsigned)T.PH && dx < (unsigned)T.PW) lds_add_f32(ptex + (dy * T.PW + dx) * F + ch, v);
                    else atomic_add_f32(a.g_tex + (img + t) * F + ch, v);
                };
                float gix = 0.f, giy = 0.f, giz = 0.f;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int ch = lane + 16 * k;
                    if (!ok || ch >= F) continue;
                    const float g = cur.g[k], v00 = cur.v00[k], v01 = cur.v01[k], v10 = cur.v10[k], v11 = cur.v11[k];
                    tex_add(y0, x0, t00, ch, w00 * g); tex_add(y0, x1, t01, ch, w01 * g);
                    tex_add(y1, x0, t10, ch, w10 * g); tex_add(y1, x1, t11, ch, w11 * g);
                    gix += g * ((v01 - v00) * ty0 + (v11 - v10) * ty1);
                    giy += g * ((v10 - v00) * tx0 + (v11 - v01) * tx1);
                }
                if (s == 0) {
                    const VoxGeom& v = cur.vg;
                    const int cz = lane >> 3, chn = lane & 7;
                    const int zoc = cz ? v.zo[1] : v.zo[0];
                    const float wzc = cz ? v.wz[1] : v.wz[0];
                    const bool vzc = cz ? v.vz[1] : v.vz[0];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int cx = t & 1, cy = t >> 1;
                        if (!(ok && v.vx[cx] && v.vy[cy] && vzc) || ENERF_GT_ABL == 2) continue;
                        const float wxy = v.wx[cx] * v.wy[cy];
                        const unsigned dz = (unsigned)(zoc - vz0), dy = (unsigned)(v.yo[cy] - vy0), dx = (unsigned)(v.xo[cx] - vx0);
                        if (dz < (unsigned)T.VZ && dy < (unsigned)T.VH && dx < (unsigned)T.VW)
                            lds_add_f32(pvol + ((dz * T.VH + dy) * T.VW + dx) * 8 + chn, wxy * wzc * cur.gvox);
                        else atomic_add_f32(a.g_vol + (vb + ((long long)zoc * a.h + v.yo[cy]) * a.w + v.xo[cx]) * 8 + chn, wxy * wzc * cur.gvox);
                        giz += (cz ? 1.f : -1.f) * cur.vv[t] * wxy * cur.gvox;
                    }
                }
                gix = sum16(gix); giy = sum16(giy);
                if (s == 0) giz = sum16(giz);
                if (lane == 0 && pj < 256) { gij[pj][0] = gix; gij[pj][1] = giy; gij[pj][2] = giz; }
            };
            const int passes = ENERF_GT_ABL == 1 ? 0 : (npts + 15) / 16;
            Stage sa, sb;
            fetch(0, sa);
            for (int it = 0; it < passes; it += 2) {
                fetch(min(it + 1, passes - 1), sb);
                __builtin_amdgcn_sched_barrier(0);
                process(sa, it);
                fetch(min(it + 2, passes - 1), sa);
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < passes) process(sb, it + 1);
            }
        }
        __syncthreads();
        // ---- (3) chain rule of (point, view s), one thread per point (the geometry again: 16 x cheaper than in the any-order
        //      kernel, and not held in registers across the scatter) ----
        if (live && ENERF_GT_ABL != 4) {
            const ViewGeom q = view_geom(c, a.tcen + b * 4, X, Y, Z, a.Wr, a.Hr);
            float gix = gij[tid][0], giy = gij[tid][1];
            if (s == 0) a.g_dn[bp] = gij[tid][2] * (float)(a.D - 1);   // iz = dn (D-1): unnormalise (D-1)/2 x d(2 dn - 1)/d dn
            if (!q.gx_on) gix = 0.f;
            if (!q.gy_on) giy = 0.f;
            const float gpx = gix / q.zc, gpy = giy / q.zc;
            const float gpz = q.pz >= 1e-6f ? -(gix * q.px + giy * q.py) / (q.zc * q.zc) : 0.f;
            gX += c[0] * gpx + c[3] * gpy + c[6] * gpz; gY += c[1] * gpx + c[4] * gpy + c[7] * gpz; gZ += c[2] * gpx + c[5] * gpy + c[8] * gpz;
            const float* gx = a.g_x + (bp * a.S + s) * XW;
            const float gd0 = gx[F], gd1 = gx[F + 1], gd2 = gx[F + 2], gdot = gx[F + 3];
            const float it = 1.f / (q.nt + 1e-6f), is = 1.f / (q.ns + 1e-6f);
            const float tx = q.tdx * it, ty = q.tdy * it, tz = q.tdz * it, sx = q.sdx * is, sy = q.sdy * is, sz = q.sdz * is;
            float gex, gey, gez;
            if (q.ne > 1e-6f) {
                const float in = 1.f / q.ne, hx = q.ex * in, hy = q.ey * in, hz = q.ez * in, pr = hx * gd0 + hy * gd1 + hz * gd2;
                gex = (gd0 - hx * pr) * in; gey = (gd1 - hy * pr) * in; gez = (gd2 - hz * pr) * in;
            } else { gex = gd0 * 1e6f; gey = gd1 * 1e6f; gez = gd2 * 1e6f; }
            const float gtx = gex + gdot * sx, gty = gey + gdot * sy, gtz = gez + gdot * sz;
            const float gsx = -gex + gdot * tx, gsy = -gey + gdot * ty, gsz = -gez + gdot * tz;
            {   // t_hat = dt / (|dt| + eps)
                const float pr = (q.tdx * gtx + q.tdy * gty + q.tdz * gtz), k = q.nt > 0.f ? pr * it * it / q.nt : 0.f;
                gX += gtx * it - q.tdx * k; gY += gty * it - q.tdy * k; gZ += gtz * it - q.tdz * k;
            }
            {
                const float pr = (q.sdx * gsx + q.sdy * gsy + q.sdz * gsz), k = q.ns > 0.f ? pr * is * is / q.ns : 0.f;
                gX += gsx * is - q.sdx * k; gY += gsy * is - q.sdy * k; gZ += gsz * is - q.sdz * k;
            }
        }
        // ---- flush: a patch row is a contiguous run of the gradient map ----
        if (ENERF_GT_ABL != 3) {
            const int rowf = T.PW * F;
            for (int i = WP ? l64 : tid; i < ntex; i += WP ? 64 : 256) {
                const float v = ptex[i];
                if (v == 0.f) continue;
                ptex[i] = 0.f;
                const int row = i / rowf, e = i - row * rowf;              // (an entry is only ever non-zero inside the image)
                atomic_add_f32(a.g_tex + (img + (long long)(y0p + row) * a.Wr + x0p) * F + e, v);
            }
        }
        if (s == 0 && ENERF_GT_ABL != 3) {
            const int rowf = T.VW * 8;
            for (int i = WP ? l64 : tid; i < nvol; i += WP ? 64 : 256) {
                const float v = pvol[i];
                if (v == 0.f) continue;
                pvol[i] = 0.f;
                const int r = i / rowf, e = i - r * rowf, dz = r / T.VH, dy = r - dz * T.VH;
                atomic_add_f32(a.g_vol + (vb + ((long long)(vz0 + dz) * a.h + vy0 + dy) * a.w + vx0) * 8 + e, v);
            }
        }
    }
    if (live) { a.g_xyz[bp * 3] = gX; a.g_xyz[bp * 3 + 1] = gY; a.g_xyz[bp * 3 + 2] = gZ; }
}

}  // namespace enerf

using namespace enerf;
extern "C" {

static int gather_check(const enerf_gather_args_t* u, GatherArgs* a, const char* what) {
    REQUIRE(u, "%s: null args", what);
    REQUIRE(u->xyz && u->dn && u->uv && u->tex && u->vol && u->cam && u->tcen, "%s: null input", what);
    REQUIRE(u->B > 0 && u->P >= 0 && u->S >= 1 && u->F >= 4 && u->Hr > 1 && u->Wr > 1 && u->D > 0 && u->h > 0 && u->w > 0, "%s: bad shape", what);
    a->xyz = u->xyz; a->dn = u->dn; a->uv = u->uv; a->tex = u->tex; a->vol = u->vol; a->cam = u->cam; a->tcen = u->tcen;
    a->x = u->x; a->vox = u->vox; a->g_x = u->g_x; a->g_vox = u->g_vox; a->g_tex = u->g_tex; a->g_vol = u->g_vol;
    a->g_xyz = u->g_xyz; a->g_dn = u->g_dn;
    a->B = u->B; a->S = u->S; a->F = u->F; a->Hr = u->Hr; a->Wr = u->Wr; a->D = u->D; a->h = u->h; a->w = u->w; a->P = u->P;
    a->ray_w = 0; a->n_samples = 0;
    return ENERF_OK;
}
int enerf_gather_fwd(const enerf_gather_args_t* u, enerf_stream_t stream) {
    GatherArgs a;
    int rc = gather_check(u, &a, "gather_fwd");
    if (rc != ENERF_OK) return rc;
    REQUIRE(u->x && u->vox, "gather_fwd: null output");
    if (u->P == 0) return ENERF_OK;
    const long long waves = cdivl((long long)a.B * a.P, 64);
    // k_gather_fwd_w packs the tap coordinates (x, y in 16 bits, z in 8): larger maps take the thread-per-(point, view) kernel
    const bool packable = a.Wr < 65535 && a.Hr < 65535 && a.w < 65535 && a.h < 65535 && a.D < 256 &&
                          (long long)a.B * a.S * a.Hr * a.Wr * a.F < (1LL << 32) && (long long)a.B * a.D * a.h * a.w * 8 < (1LL << 32);   // 32-bit offsets
    if (packable && a.F <= 16) ENERF_LAUNCH(k_gather_fwd_w<1>, (unsigned)cdivl(waves, 4), 256, 0, (hipStream_t)stream, a);
    else if (packable && a.F <= 48) ENERF_LAUNCH(k_gather_fwd_w<3>, (unsigned)cdivl(waves, 4), 256, 0, (hipStream_t)stream, a);
    else {
        const long long total = (long long)a.B * a.P * a.S;
        ENERF_LAUNCH_SIMPLE(k_gather_fwd, (unsigned)cdivl(total, 256), 256, 0, (hipStream_t)stream, a);
    }
    return check_launch("gather_fwd");
}
int enerf_gather_bwd(const enerf_gather_args_t* u, enerf_stream_t stream) {
    GatherArgs a;
    int rc = gather_check(u, &a, "gather_bwd");
    if (rc != ENERF_OK) return rc;
    REQUIRE(u->g_x && u->g_vox && u->g_tex && u->g_vol && u->g_xyz && u->g_dn, "gather_bwd: null gradient buffer");
    hipStream_t st = (hipStream_t)stream;
    zero_async2(a.g_tex, (size_t)a.B * a.S * a.Hr * a.Wr * a.F * sizeof(float), a.g_vol, (size_t)a.B * a.D * a.h * a.w * 8 * sizeof(float), st);
    if (u->P == 0) return ENERF_OK;
    const long long total = (long long)a.B * a.P * a.S;
    a.ray_w = u->ray_w; a.n_samples = u->n_samples;
    REQUIRE(a.ray_w >= 0 && a.n_samples >= 0, "gather_bwd: negative raster hint");
    const bool raster = a.ray_w > 0 && a.n_samples > 0 && a.n_samples <= 8 && a.P % ((long long)a.ray_w * a.n_samples) == 0 &&
                        (long long)a.B * a.S * a.Hr * a.Wr * a.F < (1LL << 31) && a.F <= 48 && a.Wr < 65535 && a.Hr < 65535 &&
                        a.w < 65535 && a.h < 65535 && a.D < 256 &&      // the tiled kernel packs volume taps as (x, y: 16 bits, z: 8)
                        a.Wr > 1 && a.Hr > 1;                           // its volume patch extent divides by (Wr - 1), (Hr - 1)
    if (raster) {
        GatherTile T;
        const bool wp = a.F <= 16;                            // wave-private patches (see the kernel)
        T.kg = a.n_samples <= 2 ? a.n_samples : 1;            // the fine level's samples share a patch; the coarse level's span the
        T.groups = a.n_samples / T.kg;                        // whole depth range: one block per sample index
        T.th = 8; T.tw = wp ? 4 * (64 / (8 * T.kg)) : 16;     // x kg <= 256 points: one thread per point in phases 1 and 3
        const int pw = wp ? T.tw / 4 : T.tw;                  // columns of rays behind one patch
        T.PW = pw + (wp ? 6 : 12); T.PH = T.th + (wp ? 4 : 6);
        while (T.PH > T.th + 2 && (size_t)T.PH * T.PW * a.F * sizeof(float) > 40 * 1024) --T.PH;
        T.VW = (int)((float)pw * (float)(a.w - 1) / (float)(a.Wr - 1)) + 3;
        T.VH = (int)((float)T.th * (float)(a.h - 1) / (float)(a.Hr - 1)) + 3;
        T.VZ = a.D < 4 ? a.D : 4;
        const int rows = (int)(a.P / a.n_samples / a.ray_w);
        T.tiles_x = cdiv(a.ray_w, T.tw); T.tiles_y = cdiv(rows, T.th);
        const size_t shmem = ((size_t)T.PH * T.PW * a.F + (size_t)T.VZ * T.VH * T.VW * 8) * sizeof(float) * (wp ? 4 : 1);
        const long long blocks = (long long)a.B * T.tiles_x * T.tiles_y * T.groups;
        // dynamic + the kernel's static __shared__ (geometry / point tables: <= 8 KB) inside the 64 KB every launch may use
        if (shmem + 8 * 1024 <= 64 * 1024 && blocks < (1LL << 31)) {
            if (wp) ENERF_LAUNCH((k_gather_bwd_tiled<1, true>), (unsigned)blocks, 256, shmem, st, a, T);
            else ENERF_LAUNCH((k_gather_bwd_tiled<3, false>), (unsigned)blocks, 256, shmem, st, a, T);
            return check_launch("gather_bwd");
        }
    }
    zero_async(a.g_xyz, (size_t)a.B * a.P * 3 * sizeof(float), st);     // (the tiled kernel stores every point's g_xyz itself)
    ENERF_LAUNCH(k_gather_bwd, (unsigned)cdivl(total, 16), 256, 0, st, a);
    return check_launch("gather_bwd");
}

}  // extern "C"
