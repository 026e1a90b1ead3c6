// mlp_train.hip — backward of the Agg + NeRF MLP (nerf.py:29-89) for training, fused per point.
//
// PyTorch's backward of this MLP materialises every per-(point, view) activation (up to 103 floats x 2 M rows per layer) and is
// memory-bound: 21 ms of `aten::mm` + 7 ms of reductions per training step at 512x640.  Here a wave takes 16 points,
// RECOMPUTES the forward in registers exactly like the render kernel (same packed weight image, same MFMA operand chaining)
// and back-propagates through it with the TRANSPOSED weights as MFMA A operands: a layer's output gradient sits in the D
// layout (rows 4g+r of column j), which is the B layout of the transposed product, so gradients never move between lanes
// either.  It writes (i) the gradients of the inputs (voxel feature, per-view texel features + direction code) and (ii), per
// layer, the pre-activation gradient and the layer input as channels-last rows — the weight gradients are then plain
// position-reductions on the matrix cores (enerf_gemm_wgrad, the 1x1 case of wgrad.hip).
//
// Activation layouts inside a wave (lane l = (g = l>>4, j = l&15), point j):
//   "unit" layout: f32x4 tile t holds units 16t + 4g + r;   "slot" layout: register r of lane group g holds channel g*R + r
//   (r < R) and the direction-code component g (r == R) — one 16-row tile t covers slots 4t..4t+3.
// The transposed weight images (built on the host, enerf_amd/autograd.py:mlp_backward_images) follow those two layouts.
#include "nerf_layout.h"

namespace enerf {

struct MlpBwdArgs {
    const float *vox, *x, *g_raw;        // (P,8), (P,S,F+4), (P,4)
    const float *packed, *bimg;          // forward weight image (nerf_pack), backward images
    float *g_vox, *g_x;                  // (P,8), (P,S,F+4)
    // saved layer inputs / pre-activation gradients (channels-last rows)
    float *sv_hv, *sv_G, *sv_q, *sv_g, *sv_a, *sv_vm;                       // (P,88) (P,32) (P,S,64) (P,S,32) (P,S,F) (P,2F)
    float *d_cpre, *d_qpre, *d_p2, *d_spre, *d_hpre, *d_aggpre, *d_upre, *d_gpre, *d_gsum, *d_vpre;
    long long P;
    int F, n_bimg;                       // floats of the backward images (staged in LDS behind the forward image)
    int o_b1, o_b2, o_b3, o_b4, o_b5, o_b6v, o_b6m, o_b7;                  // float offsets of the backward images
    float *wg_q, *wg_g, *wg_v, *wg_rows; // WG: per-wave partial weight gradients ([wave][4 | 2 | 1][256] tiles and [wave][128] rows, see below)
};

// A/B switches (tools/build_variant.py): backward images from LDS (default) or from global memory as until round 3; the
// per-layer saves switched off (timing ablation only: the weight gradients are then garbage).
#ifndef ENERF_MLPB_LDS_BIMG
#define ENERF_MLPB_LDS_BIMG 1
#endif
#ifndef ENERF_MLPB_NOSTORE
#define ENERF_MLPB_NOSTORE 0
#endif
// WG (round 6): the weight gradients of the PER-VIEW colour branch are accumulated inside the kernel instead of through
// saved rows.  q = relu(P2 + W_v [x_s | dir_s]) and its pre-activation gradient d_q are the two widest per-(point, view) tensors the
// kernel used to write (2 x 64 of its 967 floats per point at S = 3: 40 % of the 2.8 GB) for two consumers only:
//   color.0's per-view columns  dW[c][i] = sum_{p,s} d_q[p,s,c] [x_s | dir_s][p,s,i]   (64 x (F + 4))
//   color.2                     dW[c]    = sum_{p,s} d_c[p,s] q[p,s,c]  (+ bias sum d_c)
// The first is a 16-point product on the matrix cores per view: d_q (D layout: unit rows, point columns) and [x_s | dir_s] (slot layout) are
// transposed through a per-wave LDS tile (point-major rows, conflict-free pitches) into A / B operands with k = point — 16 MFMAs per view
// into four accumulator tiles that live in registers for the whole kernel (R = 3 runs at 354 of 512 registers).  The second is 16 fused
// multiply-adds per view into per-lane partial sums, reduced over the 16 point lanes once at the end.
// WG = 2 adds the per-view AGGREGATION branch the same way (g = relu(global_fc ...), d_g, a, d_u, d_v: another 27 % of the saved floats):
//   global_fc's `a` columns  dW[o][i] = sum d_g[p,s,o] a[p,s,i]  (32 x F: 8 MFMAs per view),  agg_w_fc  dW[o] = sum d_u g[o] (+ bias),
//   view_fc  dW[o][i] = sum d_v[p,s,o] dir[p,s,i]  (F x 4: 4 MFMAs per view, + bias sum d_v).
// Every wave ends with one row of partials: tiles wg_q[wave][4][256], wg_g[wave][2][256], wg_v[wave][256] (the layout a
// enerf_gemm_wgrad_group member with `partials` reduces) and wg_rows[wave][128] = [color.2 w 64 | b | 0 x 15 | agg_w w 32 | b | view_fc b 11 |
// 0 x 4] (enerf_colsum).
template <int R, int S, int WG = 0>
__global__ __launch_bounds__(256) void k_mlp_bwd(MlpBwdArgs a) {
    constexpr int TR = (R + 3) / 4;             // slot tiles of the F channels
    constexpr int TX = (R + 1 + 3) / 4;         // slot tiles of [channels | direction code]
    const int F = a.F, XW = F + 4;
    const NerfLayout L = nerf_layout(F);
    ENERF_DYN_SMEM(float, smem);
    float* wl = smem;
    for (int i = threadIdx.x * 4; i < L.total; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(wl + i) = *reinterpret_cast<const float4*>(a.packed + i);
#if ENERF_MLPB_LDS_BIMG
    // the transposed-weight images too (R = 3: 46 KB, R = 9: 72 KB; with the forward image 88 / 130 KB, one block per CU — which
    // the 512 registers per lane dictate anyway): a backward A operand was a 256-B global load per MFMA, 233 of them per 16
    // points with one wave per SIMD to hide them behind
    float* wb = wl + L.total;
    for (int i = threadIdx.x * 4; i < a.n_bimg; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(wb + i) = *reinterpret_cast<const float4*>(a.bimg + i);
#endif
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const float* wlane = wl + lane;
#if ENERF_MLPB_LDS_BIMG
    const float* bl = wb + lane;
#else
    const float* bl = a.bimg + lane;
#endif
    const long long ntiles = cdivl(a.P, 16);
    const int waves = blockDim.x >> 6;
    // WG: this wave's transpose tiles behind the images: d_q as [point][80] (64 units; pitch 80: the four point rows of a k-block sit 16
    // banks apart) and [x_s | dir_s] as [point][16] (column 15 = 0)
    constexpr int kTq = 16 * 80, kTx = 16 * 16, kWaveLds = kTq + 4 * kTx;
    float *tq = nullptr, *tx = nullptr, *ta = nullptr, *tv = nullptr, *td = nullptr;
    f32x4 wq[4], wc2[4], wga[2], wag[2], wvf = f32x4{0.f, 0.f, 0.f, 0.f};
    float bc2 = 0.f, bag = 0.f, bv[R];
    if (WG) {
        tq = wl + L.total + (ENERF_MLPB_LDS_BIMG ? a.n_bimg : 0) + (threadIdx.x >> 6) * kWaveLds;
        tx = tq + kTq; ta = tx + kTx; tv = ta + kTx; td = tv + kTx;
        tx[j * 16 + 12 + g] = 0.f;                          // (columns 12..14 are rewritten per view; 15 stays 0)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) { ta[j * 16 + 4 * q4 + g] = 0.f; tv[j * 16 + 4 * q4 + g] = 0.f; td[j * 16 + 4 * q4 + g] = 0.f; }   // columns >= F / 4 stay 0
#pragma unroll
        for (int v = 0; v < 4; ++v) { wq[v] = f32x4{0.f, 0.f, 0.f, 0.f}; wc2[v] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < 2; ++u) { wga[u] = f32x4{0.f, 0.f, 0.f, 0.f}; wag[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int r = 0; r < R; ++r) bv[r] = 0.f;
        wave_sync();
    }
    // The NEXT tile's inputs are requested at the top of a tile and used one iteration later (round 6): with one wave per SIMD nothing
    // else hides the ~2 us the (vox, x, dir, d raw) loads of a tile take — 40 tiles per wave, a tenth of the kernel.  PF: R = 3 only
    // (the F = 35 kernel has no registers for a second set of 36 inputs).
    constexpr bool PF = R == 3;
    struct Inputs { float vox[2], x[S][R], dsel[S]; float4 graw; };
    auto load_inputs = [&](long long t, Inputs& I) {
        const long long pr_ = t * 16 + j, p_ = pr_ < a.P ? pr_ : a.P - 1;
        I.vox[0] = a.vox[p_ * 8 + 2 * g]; I.vox[1] = a.vox[p_ * 8 + 2 * g + 1];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float* xp = a.x + (p_ * S + s) * XW;
#pragma unroll
            for (int r = 0; r < R; ++r) I.x[s][r] = xp[min(g * R + r, F - 1)];    // clamped index: always loads (masked when used)
            I.dsel[s] = xp[F + g];
        }
        I.graw = *reinterpret_cast<const float4*>(a.g_raw + p_ * 4);
    };
    const long long tile0 = (long long)blockIdx.x * waves + (threadIdx.x >> 6), tstride = (long long)gridDim.x * waves;
    Inputs nxt;
    if (PF && tile0 < ntiles) load_inputs(tile0, nxt);
    for (long long tile = tile0; tile < ntiles; tile += tstride) {
        const long long pr = tile * 16 + j;
        const bool ok = pr < a.P;
        const bool sv = ok && !ENERF_MLPB_NOSTORE;          // write the per-layer saves
        const long long p = ok ? pr : a.P - 1;
        // ---------------- inputs ----------------
        Inputs cur;
        if (PF) {
            cur = nxt;
            load_inputs(tile + tstride < ntiles ? tile + tstride : tile, nxt);   // (the last tile re-requests itself: never used)
            __builtin_amdgcn_sched_barrier(0);
        } else load_inputs(tile, cur);
        float vox[2], x[S][R], dsel[S];
        vox[0] = cur.vox[0]; vox[1] = cur.vox[1];
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int r = 0; r < R; ++r) x[s][r] = (g * R + r < F) ? cur.x[s][r] : 0.f;
            dsel[s] = cur.dsel[s];
        }
        // ---------------- forward recompute (the render kernel's MLP phase) ----------------
        float aview[TR];
        f32x4 vb[TR];
#pragma unroll
        for (int t = 0; t < TR; ++t) { aview[t] = wlane[L.view + t * 64]; vb[t] = lds4(wl + L.viewb + t * 16 + 4 * g); }
        float av[S][R];
        bool vmask[S][R];                                   // view_fc pre-activation > 0
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f32x4 va[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) va[t] = ENERF_MFMA(aview[t], dsel[s], vb[t]);
#pragma unroll
            for (int r = 0; r < R; ++r) { const float v = va[r >> 2][r & 3]; vmask[s][r] = v > 0.f; av[s][r] = x[s][r] + relu1(v); }
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
            var[r] = q * (1.f / (float)(S - 1));
        }
        f32x4 Pg[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) Pg[u] = lds4(wl + L.globb + u * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                Pg[u] = ENERF_MFMA(wlane[L.glob + ((1 * R + r) * 2 + u) * 64], var[r], Pg[u]);
                Pg[u] = ENERF_MFMA(wlane[L.glob + ((2 * R + r) * 2 + u) * 64], mean[r], Pg[u]);
            }
        f32x4 gf[S][2];
        float upre[S], aw[S];
        const f32x4 aggw0 = lds4(wl + L.aggw + 4 * g), aggw1 = lds4(wl + L.aggw + 16 + 4 * g);
        const float aggb = wl[L.aggw + 32];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            gf[s][0] = Pg[0]; gf[s][1] = Pg[1];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u) gf[s][u] = ENERF_MFMA(wlane[L.glob + ((0 * R + r) * 2 + u) * 64], av[s][r], gf[s][u]);
            gf[s][0] = relu4(gf[s][0]); gf[s][1] = relu4(gf[s][1]);
            upre[s] = group_sum(dot4(gf[s][1], aggw1, dot4(gf[s][0], aggw0, 0.f))) + aggb;
            aw[s] = relu1(upre[s]);
        }
        {
            float m = aw[0];
#pragma unroll
            for (int s = 1; s < S; ++s) m = fmaxf(m, aw[s]);
            float se = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) { aw[s] = expf(aw[s] - m); se += aw[s]; }
#pragma unroll
            for (int s = 0; s < S; ++s) aw[s] /= se;
        }
        f32x4 G[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            G[u] = gf[0][u] * aw[0];
#pragma unroll
            for (int s = 1; s < S; ++s) G[u] += gf[s][u] * aw[s];
        }
        f32x4 aggv = lds4(wl + L.fcb + 4 * g);
#pragma unroll
        for (int e = 0; e < 8; ++e) aggv = ENERF_MFMA(wlane[L.fc + e * 64], G[e >> 2][e & 3], aggv);
        const f32x4 agg = relu4(aggv);
        f32x4 hid[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) hid[v] = lds4(wl + L.lr0b + v * 16 + 4 * g);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const float bop = ks < 2 ? vox[ks < 2 ? ks : 0] : agg[ks >= 2 ? ks - 2 : 0];
#pragma unroll
            for (int v = 0; v < 4; ++v) hid[v] = ENERF_MFMA(wlane[L.lr0 + (ks * 4 + v) * 64], bop, hid[v]);
        }
        f32x4 sigw[4];
        float spre = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            hid[v] = relu4(hid[v]);
            sigw[v] = lds4(wl + L.sigma + v * 16 + 4 * g);
            spre = dot4(hid[v], sigw[v], spre);
        }
        spre = group_sum(spre) + wl[L.sigma + 64];
        f32x4 P2[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) P2[v] = lds4(wl + L.c0b + v * 16 + 4 * g);
#pragma unroll
        for (int ks = 0; ks < 22; ++ks) {
            const float bop = ks < 16 ? hid[(ks < 16 ? ks : 0) >> 2][ks & 3] : (ks < 18 ? vox[ks < 18 ? ks - 16 : 0] : agg[ks >= 18 ? ks - 18 : 0]);
#pragma unroll
            for (int v = 0; v < 4; ++v) P2[v] = ENERF_MFMA(wlane[L.c0p + (ks * 4 + v) * 64], bop, P2[v]);
        }
        f32x4 c2w[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) c2w[v] = lds4(wl + L.col2 + v * 16 + 4 * g);
        const float c2b = wl[L.col2 + 64];
        // colour logits of every view (needed before any view's gradient: softmax over views)
        float cpre[S], cl[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f32x4 cc[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) cc[v] = P2[v];
#pragma unroll
            for (int ks = 0; ks <= R; ++ks) {
                const float bop = ks < R ? x[s][ks < R ? ks : 0] : dsel[s];
#pragma unroll
                for (int v = 0; v < 4; ++v) cc[v] = ENERF_MFMA(wlane[L.c0v + (ks * 4 + v) * 64], bop, cc[v]);
            }
            float part = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) part = dot4(relu4(cc[v]), c2w[v], part);
            cpre[s] = group_sum(part) + c2b;
            cl[s] = relu1(cpre[s]);
        }
        {
            float m = cl[0];
#pragma unroll
            for (int s = 1; s < S; ++s) m = fmaxf(m, cl[s]);
            float se = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) { cl[s] = expf(cl[s] - m); se += cl[s]; }
#pragma unroll
            for (int s = 0; s < S; ++s) cl[s] /= se;
        }
        // ---------------- save the layer inputs ----------------
        if (sv) {
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(a.sv_hv + p * 88 + 16 * v + 4 * g) = hid[v];
            a.sv_hv[p * 88 + 64 + 2 * g] = vox[0]; a.sv_hv[p * 88 + 64 + 2 * g + 1] = vox[1];
            *reinterpret_cast<f32x4*>(a.sv_hv + p * 88 + 72 + 4 * g) = agg;
#pragma unroll
            for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(a.sv_G + p * 32 + 16 * u + 4 * g) = G[u];
            if (WG < 2) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(a.sv_g + (p * S + s) * 32 + 16 * u + 4 * g) = gf[s][u];
#pragma unroll
                    for (int r = 0; r < R; ++r) if (g * R + r < F) a.sv_a[(p * S + s) * F + g * R + r] = av[s][r];
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (g * R + r < F) { a.sv_vm[p * 2 * F + g * R + r] = var[r]; a.sv_vm[p * 2 * F + F + g * R + r] = mean[r]; }
        }
        // ---------------- backward ----------------
        const float4 graw = cur.graw;
        const float gcol[3] = {graw.x, graw.y, graw.z};
        const float gsig = ok ? graw.w : 0.f;
        // col = sum_s cw_s rgb_s (rgb_s = channels F-3..F-1 of x_s): d cw_s, softmax over views, ReLU of the logit
        float gcw[S], gcpre[S];
        float dotc = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int c = g * R + r - (F - 3);
                if (c >= 0 && c < 3) part += gcol[c] * x[s][r];
            }
            gcw[s] = group_sum(part);
            dotc += cl[s] * gcw[s];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float gc = cl[s] * (gcw[s] - dotc);
            gcpre[s] = (ok && cpre[s] > 0.f) ? gc : 0.f;
        }
        f32x4 dP2[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        f32x4 dxd[S][TX];                                   // gradient of [x_s | dir_s] in slot layout
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // recompute q_s = relu(P2 + W_v [x_s, dir_s])
            f32x4 cc[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) cc[v] = P2[v];
#pragma unroll
            for (int ks = 0; ks <= R; ++ks) {
                const float bop = ks < R ? x[s][ks < R ? ks : 0] : dsel[s];
#pragma unroll
                for (int v = 0; v < 4; ++v) cc[v] = ENERF_MFMA(wlane[L.c0v + (ks * 4 + v) * 64], bop, cc[v]);
            }
            f32x4 gq[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gq[v][r] = cc[v][r] > 0.f ? c2w[v][r] * gcpre[s] : 0.f;
                dP2[v] += gq[v];
                if (WG) {                                   // color.2: d_c q into the per-lane partial sums
                    const f32x4 qv = relu4(cc[v]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) wc2[v][r] += qv[r] * gcpre[s];
                } else if (sv) {
                    *reinterpret_cast<f32x4*>(a.sv_q + (p * S + s) * 64 + 16 * v + 4 * g) = relu4(cc[v]);
                    *reinterpret_cast<f32x4*>(a.d_qpre + (p * S + s) * 64 + 16 * v + 4 * g) = gq[v];
                }
            }
            if (WG) {
                bc2 += gcpre[s];
                // color.0's per-view columns: d_q and [x_s | dir_s] to point-major LDS rows, then k = point on the matrix cores
#pragma unroll
                for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(tq + j * 80 + 16 * v + 4 * g) = gq[v];
#pragma unroll
                for (int r = 0; r < R; ++r) if (g * R + r < F) tx[j * 16 + g * R + r] = x[s][r];
                tx[j * 16 + F + g] = dsel[s];
                wave_sync();
                float bx[4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) bx[kb] = tx[(4 * kb + g) * 16 + j];
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) wq[v] = ENERF_MFMA(tq[(4 * kb + g) * 80 + 16 * v + j], bx[kb], wq[v]);
                wave_sync();
            } else if (sv && g == 0) a.d_cpre[p * S + s] = gcpre[s];
            // B1: d [x_s | dir_s] = W_v^T gq
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc = ENERF_MFMA(bl[a.o_b1 + (t * 16 + kk) * 64], gq[kk >> 2][kk & 3], acc);
                dxd[s][t] = acc;
            }
            // park the slot-layout gradient in g_x (this lane's own entries) instead of holding S x TX accumulators
            // across the rest of the backward pass
            if (ok) {
                float* gx = a.g_x + (p * S + s) * XW;
#pragma unroll
                for (int r = 0; r < R; ++r) if (g * R + r < F) gx[g * R + r] = dxd[s][r >> 2][r & 3];
                gx[F + g] = dxd[s][R >> 2][R & 3];
            }
        }
        // B2: d [h | vox | agg] = W_p^T dP2
        f32x4 dh[4], dvox = f32x4{0, 0, 0, 0}, dagg = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = ENERF_MFMA(bl[a.o_b2 + (t * 16 + kk) * 64], dP2[kk >> 2][kk & 3], acc);
            if (t < 4) dh[t] = acc; else if (t == 4) dvox = acc; else dagg = acc;
        }
        if (sv) {
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(a.d_p2 + p * 64 + 16 * v + 4 * g) = dP2[v];
        }
        // sigma = softplus(spre) (beta 1, threshold 20)
        const float gspre = gsig * (spre > 20.f ? 1.f : 1.f / (1.f + expf(-spre)));
        if (sv && g == 0) a.d_spre[p] = gspre;
        f32x4 ghpre[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ghpre[v][r] = hid[v][r] > 0.f ? dh[v][r] + sigw[v][r] * gspre : 0.f;
            if (sv) *reinterpret_cast<f32x4*>(a.d_hpre + p * 64 + 16 * v + 4 * g) = ghpre[v];
        }
        // B3: d [vox | agg] += W_0^T ghpre
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc = t == 0 ? dvox : dagg;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = ENERF_MFMA(bl[a.o_b3 + (t * 16 + kk) * 64], ghpre[kk >> 2][kk & 3], acc);
            if (t == 0) dvox = acc; else dagg = acc;
        }
        f32x4 gaggpre;
#pragma unroll
        for (int r = 0; r < 4; ++r) gaggpre[r] = agg[r] > 0.f ? dagg[r] : 0.f;
        if (sv) *reinterpret_cast<f32x4*>(a.d_aggpre + p * 16 + 4 * g) = gaggpre;
        // B4: dG = W_f^T gaggpre
        f32x4 dG[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = ENERF_MFMA(bl[a.o_b4 + (u * 4 + kk) * 64], gaggpre[kk], acc);
            dG[u] = acc;
        }
        // G = sum_s w_s g_s ; w = softmax_s(relu(upre))
        float daw[S], dot2 = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            daw[s] = group_sum(dot4(dG[1], gf[s][1], dot4(dG[0], gf[s][0], 0.f)));
            dot2 += aw[s] * daw[s];
        }
        f32x4 dgsum[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        f32x4 dgp[S][2];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float du = aw[s] * (daw[s] - dot2);
            const float dup = (ok && upre[s] > 0.f) ? du : 0.f;
            if (WG < 2 && sv && g == 0) a.d_upre[p * S + s] = dup;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 wv = u == 0 ? aggw0 : aggw1;
#pragma unroll
                for (int r = 0; r < 4; ++r) dgp[s][u][r] = gf[s][u][r] > 0.f ? aw[s] * dG[u][r] + wv[r] * dup : 0.f;
                dgsum[u] += dgp[s][u];
                if (WG < 2 && sv) *reinterpret_cast<f32x4*>(a.d_gpre + (p * S + s) * 32 + 16 * u + 4 * g) = dgp[s][u];
            }
            if (WG >= 2) {      // agg_w_fc: d_u g into per-lane partial sums; global_fc's `a` columns: d_g (x) a on the matrix cores, k = point
                bag += dup;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) wag[u][r] += gf[s][u][r] * dup;
                    *reinterpret_cast<f32x4*>(tq + j * 80 + 16 * u + 4 * g) = dgp[s][u];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) if (g * R + r < F) ta[j * 16 + g * R + r] = av[s][r];
                wave_sync();
                float ba[4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) ba[kb] = ta[(4 * kb + g) * 16 + j];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) wga[u] = ENERF_MFMA(tq[(4 * kb + g) * 80 + 16 * u + j], ba[kb], wga[u]);
                wave_sync();
            }
        }
        if (sv) {
#pragma unroll
            for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(a.d_gsum + p * 32 + 16 * u + 4 * g) = dgsum[u];
        }
        // B6: d var, d mean (slot layout) = W_var^T dgsum, W_mean^T dgsum ; B5: d a_s = W_a^T dgp_s
        f32x4 dvar[TR], dmean[TR];
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            f32x4 a1 = f32x4{0, 0, 0, 0}, a2 = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                a1 = ENERF_MFMA(bl[a.o_b6v + (t * 8 + kk) * 64], dgsum[kk >> 2][kk & 3], a1);
                a2 = ENERF_MFMA(bl[a.o_b6m + (t * 8 + kk) * 64], dgsum[kk >> 2][kk & 3], a2);
            }
            dvar[t] = a1; dmean[t] = a2;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f32x4 da[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc = ENERF_MFMA(bl[a.o_b5 + (t * 8 + kk) * 64], dgp[s][kk >> 2][kk & 3], acc);
                da[t] = acc;
            }
            float dvp[R];
            f32x4 dx2[TX];
#pragma unroll
            for (int t = 0; t < TX; ++t) dx2[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float d = da[r >> 2][r & 3] + dmean[r >> 2][r & 3] * (1.f / (float)S) +
                                dvar[r >> 2][r & 3] * (2.f / (float)(S - 1)) * (av[s][r] - mean[r]);
                dx2[r >> 2][r & 3] += d;                                     // a_s = x_s + relu(view_fc(dir_s))
                dvp[r] = vmask[s][r] ? d : 0.f;
                if (WG < 2 && sv && g * R + r < F) a.d_vpre[(p * S + s) * F + g * R + r] = dvp[r];
                // the rgb channels also feed the colour blend directly: col = sum_s cw_s rgb_s
                const int c = g * R + r - (F - 3);
                if (c >= 0 && c < 3) dx2[r >> 2][r & 3] += cl[s] * gcol[c];
            }
            if (WG >= 2) {      // view_fc: d_v (x) dir_s on the matrix cores (k = point), its bias as per-lane partial sums
#pragma unroll
                for (int r = 0; r < R; ++r) { bv[r] += dvp[r]; if (g * R + r < F) tv[j * 16 + g * R + r] = dvp[r]; }
                td[j * 16 + g] = dsel[s];
                wave_sync();
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) wvf = ENERF_MFMA(tv[(4 * kb + g) * 16 + j], td[(4 * kb + g) * 16 + j], wvf);
                wave_sync();
            }
            // B7: d dir_s += W_view^T dvp  (k-steps over the view_fc outputs in slot layout)
#pragma unroll
            for (int t = 0; t < TX; ++t)
#pragma unroll
                for (int r = 0; r < R; ++r) dx2[t] = ENERF_MFMA(bl[a.o_b7 + (t * R + r) * 64], dvp[r], dx2[t]);
            if (ok) {
                float* gx = a.g_x + (p * S + s) * XW;
#pragma unroll
                for (int r = 0; r < R; ++r) if (g * R + r < F) gx[g * R + r] += dx2[r >> 2][r & 3];
                gx[F + g] += dx2[R >> 2][R & 3];
            }
        }
        if (ok) { a.g_vox[p * 8 + 2 * g] = dvox[0]; a.g_vox[p * 8 + 2 * g + 1] = dvox[1]; }
    }
    if (WG) {       // this wave's row of partial weight gradients (every wave writes one, also the waves that had no tile)
        const long long gwv = (long long)blockIdx.x * waves + (threadIdx.x >> 6);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a.wg_q[(gwv * 4 + v) * 256 + r * 64 + lane] = wq[v][r];
                const float t = row_sum16(wc2[v][r]);                          // over the 16 points of the row
                if (j == 0) a.wg_rows[gwv * 128 + 16 * v + 4 * g + r] = t;
            }
        const float tb = row_sum16(bc2);                                      // (the same value in all four rows)
        if (lane < 16) a.wg_rows[gwv * 128 + 64 + lane] = lane == 0 ? tb : 0.f;
        if (WG >= 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a.wg_g[(gwv * 2 + u) * 256 + r * 64 + lane] = wga[u][r];
                    const float t = row_sum16(wag[u][r]);
                    if (j == 0) a.wg_rows[gwv * 128 + 80 + 16 * u + 4 * g + r] = t;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) a.wg_v[gwv * 256 + r * 64 + lane] = wvf[r];
            const float tg = row_sum16(bag);
            if (lane < 16) a.wg_rows[gwv * 128 + 112 + lane] = lane == 0 ? tg : 0.f;   // [112] agg_w bias; [113..127] zeroed, then view_fc's bias:
            wave_sync();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float t = row_sum16(bv[r]);
                if (j == 0 && g * R + r < F) a.wg_rows[gwv * 128 + 113 + g * R + r] = t;
            }
        } else if (lane < 48) a.wg_rows[gwv * 128 + 80 + lane] = 0.f;
    }
}


// Forward only (training-mode forward of NerfMlpFn): raw (P,4) = [sum_s softmax_s(c_s) rgb_s | softplus(sigma)] — the same
// register-resident evaluation as the first half of k_mlp_bwd / the render kernel's MLP phase, nothing materialised.
template <int R, int S>
__global__ __launch_bounds__(256) void k_mlp_fwd(const float* __restrict__ voxp, const float* __restrict__ xin,
                                                 const float* __restrict__ packed, long long P, int Fin, float* __restrict__ raw) {
    constexpr int TR = (R + 3) / 4;
    struct { const float *vox, *x; long long P; int F; } a = {voxp, xin, P, Fin};
    const int F = a.F, XW = F + 4;
    const NerfLayout L = nerf_layout(F);
    ENERF_DYN_SMEM(float, smem);
    float* wl = smem;
    for (int i = threadIdx.x * 4; i < L.total; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(wl + i) = *reinterpret_cast<const float4*>(packed + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const float* wlane = wl + lane;
    const long long ntiles = cdivl(a.P, 16);
    const int waves = blockDim.x >> 6;
    // (the next tile's inputs are requested one iteration ahead, as in k_mlp_bwd)
    struct Inputs { float vox[2], x[S][R], dsel[S]; };
    auto load_inputs = [&](long long t, Inputs& I) {
        const long long pr_ = t * 16 + j, p_ = pr_ < a.P ? pr_ : a.P - 1;
        I.vox[0] = a.vox[p_ * 8 + 2 * g]; I.vox[1] = a.vox[p_ * 8 + 2 * g + 1];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float* xp = a.x + (p_ * S + s) * XW;
#pragma unroll
            for (int r = 0; r < R; ++r) I.x[s][r] = xp[min(g * R + r, F - 1)];    // clamped index: always loads (masked when used)
            I.dsel[s] = xp[F + g];
        }
    };
    const long long tile0 = (long long)blockIdx.x * waves + (threadIdx.x >> 6), tstride = (long long)gridDim.x * waves;
    Inputs nxt;
    if (tile0 < ntiles) load_inputs(tile0, nxt);
    for (long long tile = tile0; tile < ntiles; tile += tstride) {
        const long long pr = tile * 16 + j;
        const bool ok = pr < a.P;
        const long long p = ok ? pr : a.P - 1;
        // ---------------- inputs ----------------
        const Inputs cur = nxt;
        load_inputs(tile + tstride < ntiles ? tile + tstride : tile, nxt);       // (the last tile re-requests itself: never used)
        __builtin_amdgcn_sched_barrier(0);
        float vox[2], x[S][R], dsel[S];
        vox[0] = cur.vox[0]; vox[1] = cur.vox[1];
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int r = 0; r < R; ++r) x[s][r] = (g * R + r < F) ? cur.x[s][r] : 0.f;
            dsel[s] = cur.dsel[s];
        }
        // ---------------- forward recompute (the render kernel's MLP phase) ----------------
        float aview[TR];
        f32x4 vb[TR];
#pragma unroll
        for (int t = 0; t < TR; ++t) { aview[t] = wlane[L.view + t * 64]; vb[t] = lds4(wl + L.viewb + t * 16 + 4 * g); }
        float av[S][R];
        bool vmask[S][R];                                   // view_fc pre-activation > 0
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f32x4 va[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) va[t] = ENERF_MFMA(aview[t], dsel[s], vb[t]);
#pragma unroll
            for (int r = 0; r < R; ++r) { const float v = va[r >> 2][r & 3]; vmask[s][r] = v > 0.f; av[s][r] = x[s][r] + relu1(v); }
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
            var[r] = q * (1.f / (float)(S - 1));
        }
        f32x4 Pg[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) Pg[u] = lds4(wl + L.globb + u * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                Pg[u] = ENERF_MFMA(wlane[L.glob + ((1 * R + r) * 2 + u) * 64], var[r], Pg[u]);
                Pg[u] = ENERF_MFMA(wlane[L.glob + ((2 * R + r) * 2 + u) * 64], mean[r], Pg[u]);
            }
        f32x4 gf[S][2];
        float upre[S], aw[S];
        const f32x4 aggw0 = lds4(wl + L.aggw + 4 * g), aggw1 = lds4(wl + L.aggw + 16 + 4 * g);
        const float aggb = wl[L.aggw + 32];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            gf[s][0] = Pg[0]; gf[s][1] = Pg[1];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u) gf[s][u] = ENERF_MFMA(wlane[L.glob + ((0 * R + r) * 2 + u) * 64], av[s][r], gf[s][u]);
            gf[s][0] = relu4(gf[s][0]); gf[s][1] = relu4(gf[s][1]);
            upre[s] = group_sum(dot4(gf[s][1], aggw1, dot4(gf[s][0], aggw0, 0.f))) + aggb;
            aw[s] = relu1(upre[s]);
        }
        {
            float m = aw[0];
#pragma unroll
            for (int s = 1; s < S; ++s) m = fmaxf(m, aw[s]);
            float se = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) { aw[s] = expf(aw[s] - m); se += aw[s]; }
#pragma unroll
            for (int s = 0; s < S; ++s) aw[s] /= se;
        }
        f32x4 G[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            G[u] = gf[0][u] * aw[0];
#pragma unroll
            for (int s = 1; s < S; ++s) G[u] += gf[s][u] * aw[s];
        }
        f32x4 aggv = lds4(wl + L.fcb + 4 * g);
#pragma unroll
        for (int e = 0; e < 8; ++e) aggv = ENERF_MFMA(wlane[L.fc + e * 64], G[e >> 2][e & 3], aggv);
        const f32x4 agg = relu4(aggv);
        f32x4 hid[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) hid[v] = lds4(wl + L.lr0b + v * 16 + 4 * g);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const float bop = ks < 2 ? vox[ks < 2 ? ks : 0] : agg[ks >= 2 ? ks - 2 : 0];
#pragma unroll
            for (int v = 0; v < 4; ++v) hid[v] = ENERF_MFMA(wlane[L.lr0 + (ks * 4 + v) * 64], bop, hid[v]);
        }
        f32x4 sigw[4];
        float spre = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            hid[v] = relu4(hid[v]);
            sigw[v] = lds4(wl + L.sigma + v * 16 + 4 * g);
            spre = dot4(hid[v], sigw[v], spre);
        }
        spre = group_sum(spre) + wl[L.sigma + 64];
        f32x4 P2[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) P2[v] = lds4(wl + L.c0b + v * 16 + 4 * g);
#pragma unroll
        for (int ks = 0; ks < 22; ++ks) {
            const float bop = ks < 16 ? hid[(ks < 16 ? ks : 0) >> 2][ks & 3] : (ks < 18 ? vox[ks < 18 ? ks - 16 : 0] : agg[ks >= 18 ? ks - 18 : 0]);
#pragma unroll
            for (int v = 0; v < 4; ++v) P2[v] = ENERF_MFMA(wlane[L.c0p + (ks * 4 + v) * 64], bop, P2[v]);
        }
        f32x4 c2w[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) c2w[v] = lds4(wl + L.col2 + v * 16 + 4 * g);
        const float c2b = wl[L.col2 + 64];
        // colour logits of every view (needed before any view's gradient: softmax over views)
        float cpre[S], cl[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f32x4 cc[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) cc[v] = P2[v];
#pragma unroll
            for (int ks = 0; ks <= R; ++ks) {
                const float bop = ks < R ? x[s][ks < R ? ks : 0] : dsel[s];
#pragma unroll
                for (int v = 0; v < 4; ++v) cc[v] = ENERF_MFMA(wlane[L.c0v + (ks * 4 + v) * 64], bop, cc[v]);
            }
            float part = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) part = dot4(relu4(cc[v]), c2w[v], part);
            cpre[s] = group_sum(part) + c2b;
            cl[s] = relu1(cpre[s]);
        }
        {
            float m = cl[0];
#pragma unroll
            for (int s = 1; s < S; ++s) m = fmaxf(m, cl[s]);
            float se = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) { cl[s] = expf(cl[s] - m); se += cl[s]; }
#pragma unroll
            for (int s = 0; s < S; ++s) cl[s] /= se;
        }
        const float sig = spre > 20.f ? spre : log1pf(expf(spre));             // nn.Softplus(beta=1, threshold=20)
        if (ok) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int c = g * R + r - (F - 3);
                if (c >= 0 && c < 3) {
                    float col = x[0][r] * cl[0];
#pragma unroll
                    for (int s = 1; s < S; ++s) col += x[s][r] * cl[s];
                    raw[p * 4 + c] = col;
                }
            }
            if (g == 0) raw[p * 4 + 3] = sig;
        }
        (void)upre; (void)cpre; (void)vmask;
    }
}

}  // namespace enerf

using namespace enerf;
static long long mlp_bwd_blocks(long long P) {
    long long blocks = cdivl(cdivl(P, 16), 4);
#ifndef ENERF_MLPB_RESIDENT
#define ENERF_MLPB_RESIDENT 2
#endif
    const long long resident = (long long)device_cu_count() * ENERF_MLPB_RESIDENT;
    return blocks > resident ? resident : (blocks < 1 ? 1 : blocks);
}
static int mlp_bwd_launch(const enerf_mlp_bwd_args_t* u, int level, float* wg_q, float* wg_g, float* wg_v, float* wg_rows, enerf_stream_t stream) {
    const bool wg = level > 0;
    REQUIRE(u, "nerf_mlp_bwd: null args");
    REQUIRE(u->F == 11 || u->F == 35, "nerf_mlp_bwd: F=%d unsupported (11 or 35)", u->F);
    REQUIRE(u->S >= 2 && u->S <= 4 && u->P >= 0, "nerf_mlp_bwd: bad shape");
    if (wg) REQUIRE(u->F == 11 && wg_q && wg_rows && u->P > 0 && (level == 1 || (level == 2 && wg_g && wg_v && u->S <= 3)),
                    "nerf_mlp_bwd_partials: F = 11 only (the F = 35 kernel has neither the registers nor the LDS), P > 0, level 1 or 2 (2: S <= 3)");
    if (u->P == 0) return ENERF_OK;
    REQUIRE(u->vox && u->x && u->g_raw && u->packed && u->bimg && u->g_vox && u->g_x, "nerf_mlp_bwd: null pointer");
    for (int i = 0; i < 16; ++i)
        REQUIRE(u->save[i] || (wg && (i == 2 || i == 6 || i == 7)) || (level == 2 && (i == 3 || i == 4 || i == 12 || i == 13 || i == 15)),
                "nerf_mlp_bwd: save buffer %d missing", i);
    MlpBwdArgs a;
    a.vox = u->vox; a.x = u->x; a.g_raw = u->g_raw; a.packed = u->packed; a.bimg = u->bimg; a.g_vox = u->g_vox; a.g_x = u->g_x;
    a.sv_hv = u->save[0]; a.sv_G = u->save[1]; a.sv_q = u->save[2]; a.sv_g = u->save[3]; a.sv_a = u->save[4]; a.sv_vm = u->save[5];
    a.d_cpre = u->save[6]; a.d_qpre = u->save[7]; a.d_p2 = u->save[8]; a.d_spre = u->save[9]; a.d_hpre = u->save[10];
    a.d_aggpre = u->save[11]; a.d_upre = u->save[12]; a.d_gpre = u->save[13]; a.d_gsum = u->save[14]; a.d_vpre = u->save[15];
    a.P = u->P; a.F = u->F;
    a.wg_q = wg_q; a.wg_g = wg_g; a.wg_v = wg_v; a.wg_rows = wg_rows;
    {
        const int Rr = (u->F + 3) / 4, TXr = (Rr + 1 + 3) / 4;
        a.n_bimg = u->image_offsets[7] + TXr * Rr * 64;      // b7 is the last image (autograd.py:mlp_backward_images)
    }
    a.o_b1 = u->image_offsets[0]; a.o_b2 = u->image_offsets[1]; a.o_b3 = u->image_offsets[2]; a.o_b4 = u->image_offsets[3];
    a.o_b5 = u->image_offsets[4]; a.o_b6v = u->image_offsets[5]; a.o_b6m = u->image_offsets[6]; a.o_b7 = u->image_offsets[7];
    REQUIRE(a.n_bimg % 4 == 0 && nerf_layout(u->F).total % 4 == 0, "nerf_mlp_bwd: image sizes not float4-aligned");
    const size_t shmem = ((size_t)nerf_layout(u->F).total + (ENERF_MLPB_LDS_BIMG ? a.n_bimg : 0) + (wg ? 4 * (16 * 80 + 4 * 16 * 16) : 0)) * sizeof(float);
    REQUIRE(shmem <= 160 * 1024, "nerf_mlp_bwd: images do not fit LDS");
    const unsigned grid = (unsigned)mlp_bwd_blocks(u->P);
    hipStream_t st = (hipStream_t)stream;
    const int R = (u->F + 3) / 4;
#define ENERF_MLPB(RR, SS) ENERF_LAUNCH((k_mlp_bwd<RR, SS>), grid, 256, shmem, st, a)
#define ENERF_MLPB_WG(SS, LV) ENERF_LAUNCH((k_mlp_bwd<3, SS, LV>), grid, 256, shmem, st, a)
    if (level == 2) { if (u->S == 2) ENERF_MLPB_WG(2, 2); else ENERF_MLPB_WG(3, 2); }
    else if (level == 1) { if (u->S == 2) ENERF_MLPB_WG(2, 1); else if (u->S == 3) ENERF_MLPB_WG(3, 1); else ENERF_MLPB_WG(4, 1); }
    else if (R == 3) { if (u->S == 2) ENERF_MLPB(3, 2); else if (u->S == 3) ENERF_MLPB(3, 3); else ENERF_MLPB(3, 4); }
    else { if (u->S == 2) ENERF_MLPB(9, 2); else if (u->S == 3) ENERF_MLPB(9, 3); else ENERF_MLPB(9, 4); }
#undef ENERF_MLPB
#undef ENERF_MLPB_WG
    return check_launch("nerf_mlp_bwd");
}
extern "C" int enerf_nerf_mlp_bwd(const enerf_mlp_bwd_args_t* u, enerf_stream_t stream) { return mlp_bwd_launch(u, 0, nullptr, nullptr, nullptr, nullptr, stream); }
// partial rows (waves) the kernel writes for P points: the chunk count of the reductions that follow
extern "C" long long enerf_nerf_mlp_bwd_chunks(long long P) { return P > 0 ? mlp_bwd_blocks(P) * 4 : 0; }
extern "C" int enerf_nerf_mlp_bwd_partials(const enerf_mlp_bwd_args_t* u, int level, float* wg_q, float* wg_g, float* wg_v, float* wg_rows,
                                           enerf_stream_t stream) {
    REQUIRE(level == 1 || level == 2, "nerf_mlp_bwd_partials: level 1 (colour branch) or 2 (+ aggregation branch)");
    return mlp_bwd_launch(u, level, wg_q, wg_g, wg_v, wg_rows, stream);
}

extern "C" int enerf_nerf_mlp_fwd(const float* vox, const float* x, const float* packed, long long P, int S, int F, float* raw,
                                  enerf_stream_t stream) {
    REQUIRE(F == 11 || F == 35, "nerf_mlp_fwd: F=%d unsupported (11 or 35)", F);
    REQUIRE(S >= 2 && S <= 4 && P >= 0, "nerf_mlp_fwd: bad shape");
    if (P == 0) return ENERF_OK;
    REQUIRE(vox && x && packed && raw, "nerf_mlp_fwd: null pointer");
    const size_t shmem = (size_t)nerf_layout(F).total * sizeof(float);
    long long blocks = cdivl(cdivl(P, 16), 4);
    const long long resident = (long long)device_cu_count() * 2;
    if (blocks > resident) blocks = resident;
    const unsigned grid = (unsigned)blocks;
    hipStream_t st = (hipStream_t)stream;
    const int R = (F + 3) / 4;
#define ENERF_MLPF(RR, SS) ENERF_LAUNCH((k_mlp_fwd<RR, SS>), grid, 256, shmem, st, vox, x, packed, P, F, raw)
    if (R == 3) { if (S == 2) ENERF_MLPF(3, 2); else if (S == 3) ENERF_MLPF(3, 3); else ENERF_MLPF(3, 4); }
    else { if (S == 2) ENERF_MLPF(9, 2); else if (S == 3) ENERF_MLPF(9, 3); else ENERF_MLPF(9, 4); }
#undef ENERF_MLPF
    return check_launch("nerf_mlp_fwd");
}
