// train_glue.hip — the pieces of the TRAINING step that were still eager PyTorch ops after round 3:
//   * input gradient of the FeatureNet's two stride-2 5x5 convolutions (feature_net.py:11,14): a transposed 5x5 convolution
//     = four parity classes, each a <= 3x3 stride-1 convolution of the output gradient -> ONE stride-1 3x3 launch of the
//     inference path's MFMA kernel with 4*cin output channels (sub-kernels zero-padded to 3x3) + a depth-to-space kernel;
//   * get_depth_values (utils.py:98-151) backward: d dv -> d depth, d std of the previous level through the clamps
//     (utils.py:122-127: the clamped branch carries no gradient), the reciprocals and the align-corners upsampling;
//   * build_rays + sample_along_depth (utils.py:390-441) forward and backward: per-ray [near, far] from the upsampled
//     {depth, std, near_far} maps, sample depths, world positions, normalised depth coordinate;
//   * the camera tables (get_proj_mats lives in geometry.hip; here the per-view constants of the render-side fetches:
//     K'E, K't, camera centres — 4x4 inverses in fp64 on the device, no host synchronisation);
//   * small layout kernels: dgrad weight images (flip + channel transpose), the fused-heads weight stack, the render texels of
//     the training path and their channel slice, the index gathers that build the MLP backward's transposed-weight images.
// All HBM/latency-bound elementwise or gather work: one thread per output element, coalesced along the fastest axis.
#include "kernels.h"

namespace enerf {

// ---------------------------------------------------------------------------------------------------------------------
// dgrad of Conv2d(cin -> cout, k5, s2, p2):  gx[y][x][ci] = sum_{co,ky,kx} dz[(y+2-ky)/2][(x+2-kx)/2][co] w[co][ci][ky][kx]
// over the taps with (y+2-ky), (x+2-kx) even.  Output parity py = y&1 selects ky in {0,2,4} (py = 0) or {1,3} (py = 1):
//   py = 0: gx[2a]   = sum_{dy=-1..1} dz[a+dy] w[ky = 2(1-dy)]         py = 1: gx[2a+1] = dz[a] w[ky=3] + dz[a+1] w[ky=1]
// i.e. a 3-tap correlation with zero padding 1 (dy = -1 unused for py = 1).  w3 (4*cin, cout, 3, 3): row = cls*cin + ci,
// cls = 2*py + px, the "input channel" of that stride-1 conv is co.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_t5_subkernels(const float* __restrict__ w, int cout, int cin, float* __restrict__ w3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 4 * cin * cout * 9;
    if (i >= total) return;
    const int tx = i % 3, ty = (i / 3) % 3, co = (i / 9) % cout, r = i / (9 * cout);
    const int ci = r % cin, cls = r / cin, py = cls >> 1, px = cls & 1;
    const int dy = ty - 1, dx = tx - 1;
    const int ky = py == 0 ? 2 * (1 - dy) : (dy == 1 ? 1 : (dy == 0 ? 3 : -1));
    const int kx = px == 0 ? 2 * (1 - dx) : (dx == 1 ? 1 : (dx == 0 ? 3 : -1));
    w3[i] = (ky < 0 || kx < 0) ? 0.f : w[((co * cin + ci) * 5 + ky) * 5 + kx];
}
// src_a (N,h,w,nclsA*C): classes [0, nclsA); src_b (N,h,w,(4-nclsA)*C) or nullptr: the rest  ->  dst (N,2h,2w,C) (+ add)
__global__ __launch_bounds__(256) void k_depth_to_space2(const float* __restrict__ src_a, const float* __restrict__ src_b, int ncls_a,
                                                         const float* __restrict__ add, int N, int h, int w, int CQ,
                                                         float* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * 2 * h * 2 * w * CQ;
    if (i >= total) return;
    const int cq = (int)(i % CQ);
    long long r = i / CQ;
    const int x = (int)(r % (2 * w)); r /= 2 * w;
    const int y = (int)(r % (2 * h));
    const int n = (int)(r / (2 * h));
    const int cls = 2 * (y & 1) + (x & 1), C = CQ * 4;
    const long long pix = ((long long)n * h + (y >> 1)) * w + (x >> 1);
    const float* s = cls < ncls_a ? src_a + pix * (ncls_a * C) + cls * C : src_b + pix * ((4 - ncls_a) * C) + (cls - ncls_a) * C;
    float4 v = *reinterpret_cast<const float4*>(s + cq * 4);
    if (add != nullptr) {
        const float4 a = *reinterpret_cast<const float4*>(add + i * 4);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    *reinterpret_cast<float4*>(dst + i * 4) = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// adjoint of F.interpolate(bilinear, align_corners=True) on planar maps (n_maps, Hf, Wf) -> (n_maps, Hc, Wc), GATHER form
// (k_up2_adjoint's scheme for any scale >= 1): a coarse pixel sums w_y w_x g over the fine pixels whose taps touch it.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_ac_adjoint(const float* __restrict__ g, const float* __restrict__ add, int n_maps,
                                                           int Hf, int Wf, int Hc, int Wc, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_maps * Hc * Wc) return;
    const int xc = (int)(i % Wc), yc = (int)((i / Wc) % Hc), m = (int)(i / ((long long)Wc * Hc));
    const float sy = ac_scale(Hc, Hf), sx = ac_scale(Wc, Wf);
    float acc = 0.f;
    const float* gm = g + (long long)m * Hf * Wf;
    if (Hf == Hc && Wf == Wc) {
        acc = gm[yc * Wf + xc];
    } else {
        const int ylo = sy > 0.f ? max(0, (int)floorf((float)(yc - 1) / sy)) : 0, yhi = sy > 0.f ? min(Hf - 1, (int)ceilf((float)(yc + 1) / sy)) : Hf - 1;
        const int xlo = sx > 0.f ? max(0, (int)floorf((float)(xc - 1) / sx)) : 0, xhi = sx > 0.f ? min(Wf - 1, (int)ceilf((float)(xc + 1) / sx)) : Wf - 1;
        for (int y = ylo; y <= yhi; ++y) {
            const Lerp1 ly = ac_lerp(y, sy, Hc);
            const float wy = (ly.i0 == yc ? ly.l0 : 0.f) + (ly.i1 == yc ? ly.l1 : 0.f);
            if (wy == 0.f) continue;
            for (int x = xlo; x <= xhi; ++x) {
                const Lerp1 lx = ac_lerp(x, sx, Wc);
                const float wx = (lx.i0 == xc ? lx.l0 : 0.f) + (lx.i1 == xc ? lx.l1 : 0.f);
                if (wx != 0.f) acc += wy * wx * gm[y * Wf + x];
            }
        }
    }
    if (add != nullptr) acc += add[i];
    out[i] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// get_depth_values backward, per fine pixel (b, y, x): recompute the upsampled {d, s, a0, a1} and the clamps of
// k_depth_values, reduce g_dv over the D planes to (g_lo, g_hi) and write g_d = g_lo + g_hi, g_s = g_lo - g_hi as two
// planar maps (2, B, h, w); k_resize_ac_adjoint then takes them to the previous level's resolution.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_depth_values_bwd_fine(const float* __restrict__ pdepth, const float* __restrict__ pstd,
                                                               const float* __restrict__ pnf, const float* __restrict__ g_dv, int B,
                                                               int D, int h, int w, int hp, int wp, int depth_inv,
                                                               float* __restrict__ g_fine) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = h * w;
    if (i >= B * hw) return;
    const int b = i / hw, p = i - b * hw, y = p / w, x = p - y * w;
    const Lerp1 ly = ac_lerp(y, ac_scale(hp, h), hp), lx = ac_lerp(x, ac_scale(wp, w), wp);
    const int o00 = ly.i0 * wp + lx.i0, o01 = ly.i0 * wp + lx.i1, o10 = ly.i1 * wp + lx.i0, o11 = ly.i1 * wp + lx.i1;
    const float* pd = pdepth + (long long)b * hp * wp;
    const float* ps = pstd + (long long)b * hp * wp;
    const float* n0 = pnf + (long long)b * 2 * hp * wp;
    const float* n1 = n0 + hp * wp;
    const float d = ac_blend(ly, lx, pd[o00], pd[o01], pd[o10], pd[o11]);
    const float s = ac_blend(ly, lx, ps[o00], ps[o01], ps[o10], ps[o11]);
    const float a0 = ac_blend(ly, lx, n0[o00], n0[o01], n0[o10], n0[o11]);
    const float a1 = ac_blend(ly, lx, n1[o00], n1[o01], n1[o10], n1[o11]);
    float lo = d + s, hi = d - s;
    const bool lo_free = !(lo > a0), hi_free = !(hi < a1);          // utils.py:123-127: the replaced entries are constants
    if (!lo_free) lo = a0;
    if (!hi_free) hi = a1;
    const float nn = 1.f / lo, ff = 1.f / hi;
    const float inn = 1.f / nn, iff = 1.f / ff;
    float g_nn = 0.f, g_ff = 0.f;
    const float* gp = g_dv + (long long)b * D * hw + p;
    if (depth_inv) {                                                // dv_k = 1 / (inn + t (iff - inn))
        float g_inn = 0.f, g_iff = 0.f;
        for (int k = 0; k < D; ++k) {
            const float t = linspace01(k, D), u = inn + t * (iff - inn);
            const float gu = -gp[(long long)k * hw] / (u * u);
            g_inn += gu * (1.f - t);
            g_iff += gu * t;
        }
        g_nn = -g_inn / (nn * nn);
        g_ff = -g_iff / (ff * ff);
    } else {                                                        // dv_k = nn + t (ff - nn)
        for (int k = 0; k < D; ++k) {
            const float t = linspace01(k, D), gk = gp[(long long)k * hw];
            g_nn += gk * (1.f - t);
            g_ff += gk * t;
        }
    }
    const float g_lo = lo_free ? -g_nn / (lo * lo) : 0.f, g_hi = hi_free ? -g_ff / (hi * hi) : 0.f;
    g_fine[i] = g_lo + g_hi;
    g_fine[(long long)B * hw + i] = g_lo - g_hi;
}

// ---------------------------------------------------------------------------------------------------------------------
// build_rays + sample_along_depth (utils.py:390-441).  One thread per ray.
//   fwd: z (B,N,Ns), xyz (B,N,Ns,3), dn (B,N,Ns), uv (B,N,Ns,2) (the ray's pixel repeated per sample), rays12 (optional)
//   bwd: g_xyz, g_dn -> g_z -> (g_rn, g_rf) -> through the clamps -> scatter-add into g_depth / g_std (B,h,w) with the
//        ray's bilinear taps (atomics: ray lists are arbitrary pixel sets; the maps are zeroed by the caller)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sample_t(int j, int Ns) { return Ns == 1 ? 0.5f : linspace01(j, Ns); }
__global__ __launch_bounds__(256) void k_ray_samples_fwd(const float* __restrict__ rays8, const float* __restrict__ depth,
                                                         const float* __restrict__ std, const float* __restrict__ nf, int B, int N,
                                                         int Ns, int h, int w, int Hr, int Wr, int depth_inv, float* __restrict__ z,
                                                         float* __restrict__ xyz, float* __restrict__ dn, float* __restrict__ uv,
                                                         float* __restrict__ rays12) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * N) return;
    const int b = (int)(i / N);
    float rv[8];
    for (int k = 0; k < 8; ++k) rv[k] = rays8[i * 8 + k];
    const RayBounds rb = ray_bounds(rv[6], rv[7], depth + (long long)b * h * w, std + (long long)b * h * w,
                                    nf + (long long)b * 2 * h * w, h, w, Hr, Wr, depth_inv);
    if (rays12 != nullptr) {
        float* o = rays12 + i * 12;
        for (int k = 0; k < 8; ++k) o[k] = rv[k];
        o[8] = rb.rn; o[9] = rb.rf; o[10] = rb.vn; o[11] = rb.vf;
    }
    const float den = depth_inv ? clamp_min(rb.vn - rb.vf, 1e-6f) : clamp_min(rb.vf - rb.vn, 1e-6f);
    for (int j = 0; j < Ns; ++j) {
        const float zz = rb.rn + (rb.rf - rb.rn) * sample_t(j, Ns);
        const long long o = i * Ns + j;
        z[o] = zz;
        const float m = depth_inv ? 1.f / clamp_min(zz, 1e-6f) : zz;
        xyz[o * 3 + 0] = rv[0] + rv[3] * m;
        xyz[o * 3 + 1] = rv[1] + rv[4] * m;
        xyz[o * 3 + 2] = rv[2] + rv[5] * m;
        dn[o] = depth_inv ? (rb.vn - zz) / den : (zz - rb.vn) / den;
        uv[o * 2 + 0] = rv[6];
        uv[o * 2 + 1] = rv[7];
    }
}
__global__ __launch_bounds__(256) void k_ray_samples_bwd(const float* __restrict__ rays8, const float* __restrict__ depth,
                                                         const float* __restrict__ std, const float* __restrict__ nf,
                                                         const float* __restrict__ g_xyz, const float* __restrict__ g_dn, int B,
                                                         int N, int Ns, int h, int w, int Hr, int Wr, int depth_inv,
                                                         float* __restrict__ g_depth, float* __restrict__ g_std) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * N) return;
    const int b = (int)(i / N);
    float rv[8];
    for (int k = 0; k < 8; ++k) rv[k] = rays8[i * 8 + k];
    const float* pd = depth + (long long)b * h * w;
    const float* ps = std + (long long)b * h * w;
    const float* n0 = nf + (long long)b * 2 * h * w;
    const RayBounds rb = ray_bounds(rv[6], rv[7], pd, ps, n0, h, w, Hr, Wr, depth_inv);
    const float den = depth_inv ? clamp_min(rb.vn - rb.vf, 1e-6f) : clamp_min(rb.vf - rb.vn, 1e-6f);
    float g_rn = 0.f, g_rf = 0.f;
    for (int j = 0; j < Ns; ++j) {
        const float t = sample_t(j, Ns), zz = rb.rn + (rb.rf - rb.rn) * t;
        const long long o = i * Ns + j;
        const float gd = g_xyz[o * 3 + 0] * rv[3] + g_xyz[o * 3 + 1] * rv[4] + g_xyz[o * 3 + 2] * rv[5];
        float gz;
        if (depth_inv) gz = (zz > 1e-6f ? -gd / (zz * zz) : 0.f) - g_dn[o] / den;      // clamp_min(z, 1e-6): no gradient below
        else gz = gd + g_dn[o] / den;
        g_rn += gz * (1.f - t);
        g_rf += gz * t;
    }
    // the clamps of ray_bounds, re-derived: which of rn / rf is the free expression d +- s
    int u = (int)rv[6], v = (int)rv[7];
    u = u < 0 ? u + Wr : u; v = v < 0 ? v + Hr : v;
    u = u < 0 ? 0 : (u > Wr - 1 ? Wr - 1 : u); v = v < 0 ? 0 : (v > Hr - 1 ? Hr - 1 : v);
    const Lerp1 ly = ac_lerp(v, ac_scale(h, Hr), h), lx = ac_lerp(u, ac_scale(w, Wr), w);
    const int o00 = ly.i0 * w + lx.i0, o01 = ly.i0 * w + lx.i1, o10 = ly.i1 * w + lx.i0, o11 = ly.i1 * w + lx.i1;
    const bool same = (h == Hr) && (w == Wr);
    const float d = same ? pd[o00] : ac_blend(ly, lx, pd[o00], pd[o01], pd[o10], pd[o11]);
    const float s = same ? ps[o00] : ac_blend(ly, lx, ps[o00], ps[o01], ps[o10], ps[o11]);
    float gd_, gs_;
    if (depth_inv) {                      // rn = d + s unless > a0 ; rf = d - s unless < a1
        const float fn = !(d + s > rb.vn) ? g_rn : 0.f, ff = !(d - s < rb.vf) ? g_rf : 0.f;
        gd_ = fn + ff; gs_ = fn - ff;
    } else {                              // rn = d - s unless < a0 ; rf = d + s unless > a1
        const float fn = !(d - s < rb.vn) ? g_rn : 0.f, ff = !(d + s > rb.vf) ? g_rf : 0.f;
        gd_ = fn + ff; gs_ = ff - fn;
    }
    float* gdm = g_depth + (long long)b * h * w;
    float* gsm = g_std + (long long)b * h * w;
    if (same) {
        atomicAdd(gdm + o00, gd_); atomicAdd(gsm + o00, gs_);
    } else {
        const float w00 = ly.l0 * lx.l0, w01 = ly.l0 * lx.l1, w10 = ly.l1 * lx.l0, w11 = ly.l1 * lx.l1;
        atomicAdd(gdm + o00, gd_ * w00); atomicAdd(gsm + o00, gs_ * w00);
        if (w01 != 0.f) { atomicAdd(gdm + o01, gd_ * w01); atomicAdd(gsm + o01, gs_ * w01); }
        if (w10 != 0.f) { atomicAdd(gdm + o10, gd_ * w10); atomicAdd(gsm + o10, gs_ * w10); }
        if (w11 != 0.f) { atomicAdd(gdm + o11, gd_ * w11); atomicAdd(gsm + o11, gs_ * w11); }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// camera tables of the render-side fetches (utils.py:697-704, 712-715): cam (B,S,16) = K'E[:3,:3] | K't | source centre | 0
// and tcen (B,4) = target centre | 0; K' = K with rows 0,1 scaled; fp64 products and 4x4 inverses, stored fp32.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_camera_tables(const float* __restrict__ src_ixts, const float* __restrict__ src_exts, const float* __restrict__ tar_ext,
                                int B, int S, float scale, float* __restrict__ cam, float* __restrict__ tcen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S + B) return;
    double m[16], inv[16];
    if (i < B * S) {
        const float* K = src_ixts + i * 9;
        const float* E = src_exts + i * 16;
        float* o = cam + i * 16;
        for (int r = 0; r < 3; ++r) {
            const double sc = r < 2 ? (double)scale : 1.0;
            for (int c = 0; c < 4; ++c) {
                double a = 0;
                for (int k = 0; k < 3; ++k) a += ((double)K[r * 3 + k] * sc) * (double)E[k * 4 + c];
                if (c < 3) o[r * 3 + c] = (float)a; else o[9 + r] = (float)a;
            }
        }
        for (int k = 0; k < 16; ++k) m[k] = (double)E[k];
        const bool ok = inv4x4(m, inv);
        for (int r = 0; r < 3; ++r) o[12 + r] = ok ? (float)inv[r * 4 + 3] : NAN;
        o[15] = 0.f;
    } else {
        const int b = i - B * S;
        for (int k = 0; k < 16; ++k) m[k] = (double)tar_ext[b * 16 + k];
        const bool ok = inv4x4(m, inv);
        for (int r = 0; r < 3; ++r) tcen[b * 4 + r] = ok ? (float)inv[r * 4 + 3] : NAN;
        tcen[b * 4 + 3] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// layout kernels
// ---------------------------------------------------------------------------------------------------------------------
// dgrad weights of a stride-1 convolution: w (cout, cin, taps) -> wd (cin, cout, taps) with the taps reversed
__global__ __launch_bounds__(256) void k_weights_flip_transpose(const float* __restrict__ w, int cout, int cin, int taps,
                                                                float* __restrict__ wd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cout * cin * taps) return;
    const int t = i % taps, co = (i / taps) % cout, ci = i / (taps * cout);
    wd[i] = w[(co * cin + ci) * taps + (taps - 1 - t)];
}
// out[i] = (i < na ? a[i] : i < na + nb ? b[i - na] : 0)   (stacked weight tensors, e.g. feat_conv ++ depth_conv ++ 0)
__global__ __launch_bounds__(256) void k_concat2_pad(const float* __restrict__ a, long long na, const float* __restrict__ b,
                                                     long long nb, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = i < na ? a[i] : (i < na + nb ? b[i - na] : 0.f);
}
// texels of the training path: tex (n,Hr,Wr,F) = [feat (n,Hr,Wr,C) | bilinear_ac(src*0.5+0.5) (3)], F = C + 3, unpadded
__global__ __launch_bounds__(256) void k_pack_texels_train(const float* __restrict__ feat, int C, const float* __restrict__ src, int H,
                                                           int W, int Hr, int Wr, long long npix_total, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int F = C + 3;
    if (i >= npix_total * F) return;
    const long long pix = i / F;
    const int c = (int)(i - pix * F);
    if (c < C) { out[i] = feat[pix * C + c]; return; }
    const long long npix = (long long)Hr * Wr;
    const int img = (int)(pix / npix), p = (int)(pix - (long long)img * npix), y = p / Wr, x = p - y * Wr;
    const Lerp1 ly = ac_lerp(y, ac_scale(H, Hr), H), lx = ac_lerp(x, ac_scale(W, Wr), W);
    const float* sc = src + ((long long)img * 3 + (c - C)) * H * W;
    const float v00 = sc[ly.i0 * W + lx.i0] * 0.5f + 0.5f, v01 = sc[ly.i0 * W + lx.i1] * 0.5f + 0.5f;
    const float v10 = sc[ly.i1 * W + lx.i0] * 0.5f + 0.5f, v11 = sc[ly.i1 * W + lx.i1] * 0.5f + 0.5f;
    out[i] = (H == Hr && W == Wr) ? v00 : ac_blend(ly, lx, v00, v01, v10, v11);
}
// dst (n, C) = src (n, F)[:, c0 : c0 + C]
__global__ __launch_bounds__(256) void k_slice_channels(const float* __restrict__ src, long long n, int F, int c0, int C,
                                                        float* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const long long p = i / C;
    dst[i] = src[p * F + c0 + (int)(i - p * C)];
}
// out (n, C) = [a (n, Ca) | b (n, Cb) | 0]
__global__ __launch_bounds__(256) void k_concat_channels(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb,
                                                         long long n, int C, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const long long p = i / C;
    const int c = (int)(i - p * C);
    out[i] = c < Ca ? a[p * Ca + c] : (c < Ca + Cb ? b[p * Cb + (c - Ca)] : 0.f);
}
// out[i] = idx[i] >= 0 ? srcs[which[i]][idx[i]] : 0     (<= 64 source tensors: the MLP backward's transposed-weight images, and
// every packed convolution-weight image of a network's training step in one launch — enerf_amd/pack_plan.py)
constexpr int kGatherSrcs = 64;
struct GatherSrcs { const float* p[kGatherSrcs]; };
__global__ __launch_bounds__(256) void k_gather_images(GatherSrcs s, const int* __restrict__ which, const int* __restrict__ idx,
                                                       long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx[i];
    out[i] = j >= 0 ? s.p[which[i] & (kGatherSrcs - 1)][j] : 0.f;
}
// sum_i w_i * mean((a_i - b_i)^2) is the trainer's; this is its plain building block: out = a (+ b), any length
__global__ __launch_bounds__(256) void k_add2(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void k_cast_f64_f32(const double* __restrict__ in, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
__global__ __launch_bounds__(256) void k_reciprocal(const float* __restrict__ x, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.f / x[i];
}

}  // namespace enerf

using namespace enerf;
extern "C" {

// scratch layout of enerf_conv2d_s2k5_dgrad: [w3 plain | packed image of each part | class outputs], 64-float aligned pieces
struct S2k5Plan { int parts, cout3; long long w3, pk, cls, total; };
static S2k5Plan s2k5_plan(int cin, int cout, int N, int Ho, int Wo) {
    S2k5Plan p;
    p.cout3 = 4 * cin > 32 ? 2 * cin : 4 * cin;                    // the stride-1 kernel's widest layer is 32 output channels
    p.parts = 4 * cin / p.cout3;
    auto al = [](long long v) { return (v + 63) / 64 * 64; };
    p.w3 = al((long long)4 * cin * cout * 9);
    p.pk = al(enerf_conv2d_layer_packed_floats(cout, p.cout3, 3));
    p.cls = al((long long)N * Ho * Wo * p.cout3);
    p.total = p.w3 + p.parts * (p.pk + p.cls);
    return p;
}
size_t enerf_conv2d_s2k5_dgrad_workspace_bytes(int cin, int cout, int N, int Ho, int Wo) {
    return (size_t)s2k5_plan(cin, cout, N, Ho, Wo).total * sizeof(float);
}
// the packed 3x3 sub-kernel images alone (parts x enerf_conv2d_layer_packed_floats(cout, cout3, 3), 64-float aligned pieces):
// what a caller that prepares all of a step's weight images in one launch (enerf_amd/pack_plan.py) traces and then hands to
// enerf_conv2d_s2k5_dgrad_packed.  `w3_scratch`: 4*cin*cout*9 floats.
long long enerf_conv2d_s2k5_dgrad_packed_floats(int cin, int cout) {
    const S2k5Plan pl = s2k5_plan(cin, cout, 1, 1, 1);
    return pl.parts * pl.pk;
}
int enerf_conv2d_s2k5_dgrad_pack(const float* w, int cin, int cout, float* w3_scratch, float* packed, enerf_stream_t stream) {
    REQUIRE(w && w3_scratch && packed, "conv2d_s2k5_dgrad_pack: null pointer");
    REQUIRE((cin == 8 && cout == 16) || (cin == 16 && cout == 32), "conv2d_s2k5_dgrad_pack: the FeatureNet's layers are 8 -> 16 and 16 -> 32 (got %d -> %d)", cin, cout);
    const S2k5Plan pl = s2k5_plan(cin, cout, 1, 1, 1);
    const long long w3n = (long long)4 * cin * cout * 9;
    ENERF_LAUNCH_SIMPLE(k_t5_subkernels, (unsigned)cdivl(w3n, 256), 256, 0, (hipStream_t)stream, w, cout, cin, w3_scratch);
    for (int q = 0; q < pl.parts; ++q) {
        int rc = enerf_conv2d_layer_pack(w3_scratch + (long long)q * pl.cout3 * cout * 9, nullptr, cout, pl.cout3, 3, packed + q * pl.pk, stream);
        if (rc != ENERF_OK) return rc;
    }
    return check_launch("conv2d_s2k5_dgrad_pack");
}
// workspace: the class outputs only (enerf_conv2d_s2k5_dgrad_workspace_bytes covers it)
int enerf_conv2d_s2k5_dgrad_packed(const float* packed, int cin, int cout, const float* dz, const float* add, float* gx, int N, int Ho,
                                   int Wo, void* workspace, size_t workspace_bytes, enerf_stream_t stream) {
    REQUIRE(packed && dz && gx && workspace && N > 0 && Ho > 0 && Wo > 0, "conv2d_s2k5_dgrad_packed: bad arguments");
    REQUIRE((cin == 8 && cout == 16) || (cin == 16 && cout == 32), "conv2d_s2k5_dgrad_packed: the FeatureNet's layers are 8 -> 16 and 16 -> 32 (got %d -> %d)", cin, cout);
    const S2k5Plan pl = s2k5_plan(cin, cout, N, Ho, Wo);
    REQUIRE(workspace_bytes >= (size_t)pl.parts * pl.cls * sizeof(float), "conv2d_s2k5_dgrad_packed: workspace too small");
    float* cls = (float*)workspace;
    for (int q = 0; q < pl.parts; ++q) {
        int rc = enerf_conv2d_layer(packed + q * pl.pk, cout, pl.cout3, 3, 1, dz, nullptr, cls + q * pl.cls, N, Ho, Wo, stream);
        if (rc != ENERF_OK) return rc;
    }
    const long long total = (long long)N * 2 * Ho * 2 * Wo * (cin / 4);
    ENERF_LAUNCH_SIMPLE(k_depth_to_space2, (unsigned)cdivl(total, 256), 256, 0, (hipStream_t)stream, cls, pl.parts == 2 ? cls + pl.cls : nullptr,
                        4 / pl.parts, add, N, Ho, Wo, cin / 4, gx);
    return check_launch("conv2d_s2k5_dgrad_packed");
}
int enerf_conv2d_s2k5_dgrad(const float* w, int cin, int cout, const float* dz, const float* add, float* gx, int N, int Ho, int Wo,
                            void* workspace, size_t workspace_bytes, enerf_stream_t stream) {
    REQUIRE(w && dz && gx && workspace && N > 0 && Ho > 0 && Wo > 0, "conv2d_s2k5_dgrad: bad arguments");
    REQUIRE((cin == 8 && cout == 16) || (cin == 16 && cout == 32), "conv2d_s2k5_dgrad: the FeatureNet's layers are 8 -> 16 and 16 -> 32 (got %d -> %d)", cin, cout);
    const S2k5Plan pl = s2k5_plan(cin, cout, N, Ho, Wo);
    REQUIRE(workspace_bytes >= (size_t)pl.total * sizeof(float), "conv2d_s2k5_dgrad: workspace too small");
    float* w3 = (float*)workspace;
    float* packed = w3 + pl.w3;
    int rc = enerf_conv2d_s2k5_dgrad_pack(w, cin, cout, w3, packed, stream);
    if (rc != ENERF_OK) return rc;
    return enerf_conv2d_s2k5_dgrad_packed(packed, cin, cout, dz, add, gx, N, Ho, Wo, packed + pl.parts * pl.pk,
                                          (size_t)pl.parts * pl.cls * sizeof(float), stream);
}

int enerf_resize_ac_adjoint(const float* grad_fine, const float* add, int n_maps, int Hf, int Wf, int Hc, int Wc, float* grad_coarse,
                            enerf_stream_t stream) {
    REQUIRE(grad_fine && grad_coarse && n_maps > 0 && Hc > 0 && Wc > 0 && Hf >= Hc && Wf >= Wc, "resize_ac_adjoint: bad arguments");
    const long long total = (long long)n_maps * Hc * Wc;
    ENERF_LAUNCH_SIMPLE(k_resize_ac_adjoint, (unsigned)cdivl(total, 256), 256, 0, (hipStream_t)stream, grad_fine, add, n_maps, Hf, Wf, Hc,
                        Wc, grad_coarse);
    return check_launch("resize_ac_adjoint");
}

int enerf_get_depth_values_bwd(const float* prev_depth, const float* prev_std, const float* prev_near_far, const float* grad_dv, int B,
                               int D, int h, int w, int hp, int wp, int depth_inv, float* grad_depth, float* grad_std, float* scratch,
                               enerf_stream_t stream) {
    REQUIRE(prev_depth && prev_std && prev_near_far && grad_dv && grad_depth && grad_std && scratch, "get_depth_values_bwd: null pointer");
    REQUIRE(B > 0 && D > 0 && h >= hp && w >= wp && hp > 0 && wp > 0, "get_depth_values_bwd: bad shape");
    REQUIRE(grad_std == grad_depth + (long long)B * hp * wp, "get_depth_values_bwd: grad_depth and grad_std are the two halves of one (2,B,hp,wp) buffer");
    hipStream_t st = (hipStream_t)stream;
    ENERF_LAUNCH_SIMPLE(k_depth_values_bwd_fine, (unsigned)cdiv(B * h * w, 256), 256, 0, st, prev_depth, prev_std, prev_near_far, grad_dv,
                        B, D, h, w, hp, wp, depth_inv, scratch);
    const long long total = (long long)2 * B * hp * wp;
    ENERF_LAUNCH_SIMPLE(k_resize_ac_adjoint, (unsigned)cdivl(total, 256), 256, 0, st, scratch, (const float*)nullptr, 2 * B, h, w, hp, wp,
                        grad_depth);
    return check_launch("get_depth_values_bwd");
}

int enerf_ray_samples_fwd(const float* rays8, const float* depth, const float* std, const float* near_far, int B, int N, int n_samples,
                          int h, int w, int Hr, int Wr, int depth_inv, float* z, float* xyz, float* dn, float* uv, float* rays12,
                          enerf_stream_t stream) {
    REQUIRE(rays8 && depth && std && near_far && z && xyz && dn && uv && B > 0 && N > 0 && n_samples > 0 && n_samples <= 64,
            "ray_samples_fwd: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_ray_samples_fwd, (unsigned)cdivl((long long)B * N, 256), 256, 0, (hipStream_t)stream, rays8, depth, std, near_far,
                        B, N, n_samples, h, w, Hr, Wr, depth_inv, z, xyz, dn, uv, rays12);
    return check_launch("ray_samples_fwd");
}
int enerf_ray_samples_bwd(const float* rays8, const float* depth, const float* std, const float* near_far, const float* grad_xyz,
                          const float* grad_dn, int B, int N, int n_samples, int h, int w, int Hr, int Wr, int depth_inv,
                          float* grad_depth, float* grad_std, enerf_stream_t stream) {
    REQUIRE(rays8 && depth && std && near_far && grad_xyz && grad_dn && grad_depth && grad_std && B > 0 && N > 0 && n_samples > 0,
            "ray_samples_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    zero_async2(grad_depth, (size_t)B * h * w * sizeof(float), grad_std, (size_t)B * h * w * sizeof(float), st);
    ENERF_LAUNCH_SIMPLE(k_ray_samples_bwd, (unsigned)cdivl((long long)B * N, 256), 256, 0, st, rays8, depth, std, near_far, grad_xyz, grad_dn,
                        B, N, n_samples, h, w, Hr, Wr, depth_inv, grad_depth, grad_std);
    return check_launch("ray_samples_bwd");
}

int enerf_camera_tables(const float* src_ixts, const float* src_exts, const float* tar_ext, int B, int S, float render_scale, float* cam,
                        float* tcen, enerf_stream_t stream) {
    REQUIRE(src_ixts && src_exts && tar_ext && cam && tcen && B > 0 && S > 0, "camera_tables: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_camera_tables, (unsigned)cdiv(B * S + B, 64), 64, 0, (hipStream_t)stream, src_ixts, src_exts, tar_ext, B, S,
                        render_scale, cam, tcen);
    return check_launch("camera_tables");
}

int enerf_weights_flip_transpose(const float* w, int cout, int cin, int taps, float* out, enerf_stream_t stream) {
    REQUIRE(w && out && cout > 0 && cin > 0 && taps > 0, "weights_flip_transpose: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_weights_flip_transpose, (unsigned)cdiv(cout * cin * taps, 256), 256, 0, (hipStream_t)stream, w, cout, cin, taps, out);
    return check_launch("weights_flip_transpose");
}
int enerf_concat2_pad(const float* a, long long na, const float* b, long long nb, long long n, float* out, enerf_stream_t stream) {
    REQUIRE(a && out && na > 0 && nb >= 0 && n >= na + nb && (b || nb == 0), "concat2_pad: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_concat2_pad, (unsigned)cdivl(n, 256), 256, 0, (hipStream_t)stream, a, na, b, nb, n, out);
    return check_launch("concat2_pad");
}
int enerf_pack_texels_train(const float* feat_cl, int C, const float* src_inps, int H, int W, int Hr, int Wr, int n_img, float* tex,
                            enerf_stream_t stream) {
    REQUIRE(feat_cl && src_inps && tex && C > 0 && n_img > 0 && Hr > 0 && Wr > 0 && H >= Hr && W >= Wr, "pack_texels_train: bad arguments");
    const long long npix = (long long)n_img * Hr * Wr;
    ENERF_LAUNCH_SIMPLE(k_pack_texels_train, (unsigned)cdivl(npix * (C + 3), 256), 256, 0, (hipStream_t)stream, feat_cl, C, src_inps, H, W, Hr,
                        Wr, npix, tex);
    return check_launch("pack_texels_train");
}
int enerf_slice_channels(const float* src, long long n, int F, int c0, int C, float* dst, enerf_stream_t stream) {
    REQUIRE(src && dst && n > 0 && c0 >= 0 && C > 0 && c0 + C <= F, "slice_channels: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_slice_channels, (unsigned)cdivl(n * C, 256), 256, 0, (hipStream_t)stream, src, n, F, c0, C, dst);
    return check_launch("slice_channels");
}
int enerf_concat_channels(const float* a, int Ca, const float* b, int Cb, long long n, int C, float* out, enerf_stream_t stream) {
    REQUIRE(a && out && n > 0 && Ca > 0 && Cb >= 0 && C >= Ca + Cb && (b || Cb == 0), "concat_channels: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_concat_channels, (unsigned)cdivl(n * C, 256), 256, 0, (hipStream_t)stream, a, Ca, b, Cb, n, C, out);
    return check_launch("concat_channels");
}
int enerf_gather_images(const float* const* srcs, int n_srcs, const int* which, const int* idx, long long n, float* out,
                        enerf_stream_t stream) {
    REQUIRE(srcs && which && idx && out && n > 0 && n_srcs >= 1 && n_srcs <= kGatherSrcs, "gather_images: bad arguments");
    GatherSrcs s;
    for (int k = 0; k < kGatherSrcs; ++k) s.p[k] = k < n_srcs ? srcs[k] : srcs[0];
    ENERF_LAUNCH_SIMPLE(k_gather_images, (unsigned)cdivl(n, 256), 256, 0, (hipStream_t)stream, s, which, idx, n, out);
    return check_launch("gather_images");
}
int enerf_cast_f64_f32(const double* in, long long n, float* out, enerf_stream_t stream) {
    REQUIRE(in && out && n > 0, "cast_f64_f32: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_cast_f64_f32, (unsigned)cdivl(n, 256), 256, 0, (hipStream_t)stream, in, n, out);
    return check_launch("cast_f64_f32");
}
int enerf_reciprocal(const float* x, long long n, float* out, enerf_stream_t stream) {
    REQUIRE(x && out && n > 0, "reciprocal: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_reciprocal, (unsigned)cdivl(n, 256), 256, 0, (hipStream_t)stream, x, n, out);
    return check_launch("reciprocal");
}
int enerf_add(const float* a, const float* b, long long n, float* out, enerf_stream_t stream) {
    REQUIRE(a && b && out && n > 0, "add: bad arguments");
    ENERF_LAUNCH_SIMPLE(k_add2, (unsigned)cdivl(n, 256), 256, 0, (hipStream_t)stream, a, b, n, out);
    return check_launch("add");
}

}  // extern "C"
