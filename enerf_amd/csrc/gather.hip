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

__device__ __forceinline__ float sum16(float v) {          // sum over the 16 lanes of a group (lane ^ 1, 2, 4, 8)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// Backward: 16 lanes per (point, view), lane = channel (c, c+16, c+32): the scatter-add of a tap is ONE coalesced run of F
// floats (the thread-per-point form issued 64 scattered cache lines per atomic instruction and ran 2.5x slower than the
// library grid_sampler backward; this form is bounded by the L2 atomic rate of 4*F*P*S adds).
__global__ __launch_bounds__(256) void k_gather_bwd(GatherArgs a) {
    const int lane = threadIdx.x & 15;
    const long long i_raw = (long long)blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
    const long long total = (long long)a.B * a.P * a.S;
    const bool live = i_raw < total;                        // dead groups follow along (uniform shuffles), write nothing
    const long long i = live ? i_raw : total - 1;
    const int s = (int)(i % a.S);
    const long long bp = i / a.S;
    const int b = (int)(bp / a.P);
    const int F = a.F, XW = F + 4;
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
        atomic_add_f32(a.g_tex + o00 + ch, q.w00 * g); atomic_add_f32(a.g_tex + o01 + ch, q.w01 * g);
        atomic_add_f32(a.g_tex + o10 + ch, q.w10 * g); atomic_add_f32(a.g_tex + o11 + ch, q.w11 * g);
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
                atomic_add_f32(a.g_vol + o, wxy * v.wz[cz] * g);
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
    atomic_add_f32(a.g_xyz + bp * 3, gX); atomic_add_f32(a.g_xyz + bp * 3 + 1, gY); atomic_add_f32(a.g_xyz + bp * 3 + 2, gZ);
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
    return ENERF_OK;
}
int enerf_gather_fwd(const enerf_gather_args_t* u, enerf_stream_t stream) {
    GatherArgs a;
    int rc = gather_check(u, &a, "gather_fwd");
    if (rc != ENERF_OK) return rc;
    REQUIRE(u->x && u->vox, "gather_fwd: null output");
    if (u->P == 0) return ENERF_OK;
    const long long total = (long long)a.B * a.P * a.S;
    ENERF_LAUNCH_SIMPLE(k_gather_fwd, (unsigned)cdivl(total, 256), 256, 0, (hipStream_t)stream, a);
    return check_launch("gather_fwd");
}
int enerf_gather_bwd(const enerf_gather_args_t* u, enerf_stream_t stream) {
    GatherArgs a;
    int rc = gather_check(u, &a, "gather_bwd");
    if (rc != ENERF_OK) return rc;
    REQUIRE(u->g_x && u->g_vox && u->g_tex && u->g_vol && u->g_xyz && u->g_dn, "gather_bwd: null gradient buffer");
    hipStream_t st = (hipStream_t)stream;
    zero_async(a.g_tex, (size_t)a.B * a.S * a.Hr * a.Wr * a.F * sizeof(float), st);
    zero_async(a.g_vol, (size_t)a.B * a.D * a.h * a.w * 8 * sizeof(float), st);
    if (u->P == 0) return ENERF_OK;
    zero_async(a.g_xyz, (size_t)a.B * a.P * 3 * sizeof(float), st);
    const long long total = (long long)a.B * a.P * a.S;
    ENERF_LAUNCH(k_gather_bwd, (unsigned)cdivl(total, 16), 256, 0, st, a);
    return check_launch("gather_bwd");
}

}  // extern "C"
