// render.hip — fused per-ray rendering: sample_along_depth (utils.py:422-441) + get_vox_feat
// (utils.py:456-458) + get_img_feat (utils.py:689-722) + Agg/NeRF MLP (nerf.py:29-89) + raw2outputs
// (utils.py:571-603), i.e. Network.render_rays (network.py:24-43) in ONE kernel.  Nothing between the
// 12-float rays and {rgb, depth, weights} ever touches HBM (the reference materialises ~118 MB of
// gathered features plus every MLP activation).
//
// Mapping.  A wave owns 16 rays at a time; lane l = (g = l>>4, j = l&15).
//   Gather phase (per sample): lane group g < S owns SOURCE VIEW g of point j — projection, bilinear taps, 16-byte gathers
//   of the whole texel, blend, direction code — and writes a (view, point) record to LDS; every group also fetches its
//   channel pair of the trilinear voxel feature.  All addresses are computed first, then all gathers are issued together
//   (pinned with sched_barrier), then the arithmetic that does not depend on them, then the blends.
//   MLP phase: the MLP runs on v_mfma_f32_16x16x4_f32 with the *weights* as the A operand (rows = output units) and the
//   16 points as the B/D columns:
//     A: lane holds W[row = j][k = g]      B: lane holds X[k = g][col = j]
//     D: lane holds rows 4g+r (r = 0..3) of column j
//   A layer's D registers are therefore directly the next layer's B operands (k-step (tile,r) supplies unit 16*tile+4g+r
//   from lane group g); the packed weight image (nerf_pack) is permuted to that K order, so activations never move between
//   lanes.  Gathered features use the same idea: lane group g reads channels [gR, gR+R) of each view's LDS record
//   (R = ceil((C+3)/4)), register r is k-step r.  Width-1 heads (agg weight, sigma, colour logit) are in-lane dot products +
//   a sum over the four lane groups (v_permlane32_swap / v_permlane16_swap).  Softmaxes over views and the compositing scan
//   over samples are in-lane.  A operands stream from LDS through a two-deep register ring one k-step ahead of the MFMAs.
//
// Algebra (fp re-association only): global_fc = W_a·a_s + W_vm·[var,mean], color.0 = W_p·[h,vox,agg] + W_v·[x_s,dir_s] — the
// view-independent halves are evaluated once per point instead of once per view (201 instead of 369 MFMAs per 16 points at
// S=3, C=8) — and pix = (K'·E)·X + K'·t with the 3x4 product formed once per view in fp64.
//
// Roofline: fp32 MFMA (157.3 TF).  On gfx950 VALU and MFMA instructions of a SIMD do NOT overlap, within a wave or across
// waves (tools/micro/mfma_valu_overlap.hip: 4.50 ms MFMA-only + 1.68 ms VALU-only = 6.5 ms interleaved in one wave, 5.7 ms
// from two waves): the kernel's time is (201 MFMA x 32 cycles + ~700 VALU x ~3-4 cycles) per 16 samples plus what latency the
// two waves per SIMD fail to hide.  Algorithmic FLOPs/point: SURVEY.md §8a (50,952 at level 1).
#include "nerf_layout.h"

#ifndef ENERF_RENDER_PREFETCH
#define ENERF_RENDER_PREFETCH 0      // 1: the level-1 kernel gathers sample k+1 during sample k's Agg phase (see PF below)
#endif

namespace enerf {

// fp32 image (nerf_layout) + for F = 11 the bf16x3 tiles of the level-1 render (bx_layout: 44 tiles x 64 lanes x 16 bytes)
// three planes (hi, mid, lo) of [pair-tile][lane] 16 bytes = bx_layout().tiles * 256 floats each
__host__ __device__ __forceinline__ int bx_image_floats(int F, int pieces = 3) { return F == 11 ? bx_layout().tiles * 256 * pieces : 0; }
long long nerf_packed_floats(int F) { return nerf_layout(F).total + bx_image_floats(F); }

// slot (g, r) of the channel layout -> channel index (or -1)
__host__ __device__ __forceinline__ int slot_channel(int g, int r, int R, int F) {
    int c = g * R + r;
    return (r < R && c < F) ? c : -1;
}

// one weight of the bf16 image: pair-tile e, lane (g, j), k-slot q = 0..7 (q < 4: first 16-unit group of the pair, slot r = q;
// q >= 4: second group, r = q - 4); see bx_layout / the BX branches of k_render_rays
__device__ __forceinline__ float bx_weight(const NerfRaw& w, int F, int e, int g, int j, int q) {
    const int R = (F + 3) / 4, CI = 88 + F + 4;
    const BxLayout B = bx_layout();
    const int half = q >> 2, r = q & 3;
    const int c = (r < R && g * R + r < F) ? g * R + r : -1;          // feature channel of slot (g, r)
    if (e < B.ga) return c >= 0 ? w.glob_w[(16 * e + j) * 3 * F + (1 + half) * F + c] : 0.f;       // [variance | mean] columns, tile u = e
    if (e < B.fc) return (half == 0 && c >= 0) ? w.glob_w[(16 * (e - B.ga) + j) * 3 * F + c] : 0.f;  // per-view a_s columns | 0
    if (e < B.lr0) return w.fc_w[j * 32 + 16 * half + 4 * g + r];                                    // agg.fc: [G tile 0 | G tile 1]
    if (e < B.c0p) {                         // lr0: [voxel pair (slots 0, 1) | agg]; tile v
        const int unit = 16 * (e - B.lr0) + j;
        if (half == 0) return r < 2 ? w.lr0_w[unit * 24 + 2 * g + r] : 0.f;
        return w.lr0_w[unit * 24 + 8 + 4 * g + r];
    }
    if (e < B.c0v) {                         // color.0, shared columns: pair 0 [h0 | h1], 1 [h2 | h3], 2 [voxel pair | agg]; tile = pair*4 + v
        const int pr = (e - B.c0p) >> 2, unit = 16 * ((e - B.c0p) & 3) + j;
        if (pr < 2) return w.col0_w[unit * CI + 16 * (2 * pr + half) + 4 * g + r];
        if (half == 0) return r < 2 ? w.col0_w[unit * CI + 64 + 2 * g + r] : 0.f;
        return w.col0_w[unit * CI + 72 + 4 * g + r];
    }
    if (half) return 0.f;                    // color.0, per-view columns | 0: slots r < R = texel channels, slot R = direction code g
    const int unit = 16 * (e - B.c0v) + j;
    if (c >= 0) return w.col0_w[unit * CI + 88 + c];
    return r == R ? w.col0_w[unit * CI + 88 + F + g] : 0.f;
}
__device__ __forceinline__ unsigned bf16_bits_rne(float f) {
    unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ __launch_bounds__(256) void k_nerf_pack(NerfRaw w, int F, int viewdir_agg, float* __restrict__ out) {
    const NerfLayout L = nerf_layout(F);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.total + bx_image_floats(F)) return;
    if (i >= L.total) {                      // bf16 image: plane (hi, mid, lo) x [pair-tile][lane] x 4 words, word = slots (2w, 2w+1)
        const int idx = i - L.total, n_plane = bx_layout().tiles * 256;
        const int piece = idx / n_plane, k = idx - piece * n_plane;
        const int e = k >> 8, lane = (k >> 2) & 63, wq = k & 3;
        unsigned word = 0;
        for (int h = 0; h < 2; ++h) {
            float rem = bx_weight(w, F, e, lane >> 4, lane & 15, 2 * wq + h);
            unsigned bits = 0;
            for (int pc = 0; pc <= piece; ++pc) { bits = bf16_bits_rne(rem); rem -= __uint_as_float(bits << 16); }
            word |= (bits & 0xffffu) << (16 * h);
        }
        out[i] = __uint_as_float(word);
        return;
    }
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
    int total = (int)nerf_packed_floats(F);
    ENERF_LAUNCH_SIMPLE(k_nerf_pack, cdiv(total, 256), 256, 0, st, raw, F, viewdir_agg, packed);
}

// camera table in LDS, per (b,s): M = K'·E[:3,:3] (9) | v = K'·E[:3,3] (3) | camera centre (3) | pad ; per b: target centre
constexpr int kCamStride = 16;

// Staging stride (floats) of one (view, point) record in LDS: TEX blended texel channels + 4 direction-code values,
// padded so that 16 consecutive points land on distinct 16-byte bank groups (20 = 4*5, 44 = 4*11: j*stride mod 64 hits
// every multiple of 4 once) — conflict-free ds_write_b128 by the geometry lanes and ds_read_b32 by the MLP lanes.
template <int R> struct Stage { static constexpr int kTex = 4 * R, kStride = 4 * R + 8; };

// k_render_rays — one wave renders 16 rays ("points" per sample) at a time.
//   Geometry phase, lane (g, j): lane group g < S owns SOURCE VIEW g of point j: projection, bilinear taps, the gather
//     of ALL texel channels of its view (16-byte loads), the blend and the view's direction code, written to an LDS
//     record; every lane group also fetches its channel pair of the trilinear voxel feature.  (Before: all four lane
//     groups projected all S views — the same ~390 VALU instructions four times over, 552 of the 930 per tile.)
//   MLP phase, lane (g, j): channels [gR, gR+R) of every view come back from the LDS records as MFMA B operands; the
//     rest is the k-ordered MFMA chain described at the top of this file.
// WPE = waves per SIMD the register budget is sized for (512 / WPE VGPRs): blocks per CU x WAVES / 4
// BX = 3 / 6 (R = 3 only): the dense layers of the MLP on the bf16 matrix cores with operands split into 2 / 3 bf16 pieces
// (nerf_layout.h "bf16x3" / "bf16x6"); view_fc (k = 4) and the width-1 heads stay as they are.  BX = 0: fp32 MFMAs.
template <int R, int S, int WAVES, int WPE, bool PFK = false, bool LEANK = false, int BX = 0>
__global__ __launch_bounds__(64 * WAVES)
#ifndef ENERF_EMU
__attribute__((amdgpu_waves_per_eu(WPE, WPE)))
#endif
void k_render_rays(RenderArgs a) {
    static_assert(BX == 0 || R == 3, "the bf16 image exists for F = 11 only");
    constexpr bool BX3 = BX != 0;                        // (name kept: "the MLP runs on bf16 pieces")
    constexpr int NP = BX == 6 ? 3 : 2;                  // bf16 pieces per operand
    constexpr int TR = (R + 3) / 4;
    constexpr int TEX = Stage<R>::kTex, SST = Stage<R>::kStride;
    const NerfLayout LF = nerf_layout(a.F);              // the image in global memory
    const NerfLayout L = nerf_layout(a.F, BX3);          // what is staged: without the fp32 MFMA tiles in the bf16 variants
    constexpr int BXP = 25 * 256;                        // floats of one bf16 plane (bx_layout().tiles pair-tiles x 64 lanes x 16 B)
    constexpr int BXF = BX3 ? BXP * NP : 0;              // the planes staged behind it (hi, mid[, lo])
    ENERF_DYN_SMEM(float, smem);
    float* wl = smem;                                    // packed weights
    const float* bx = smem + L.total;                    // bf16 planes
    auto bxplane = [&](int q) { return bx + q * BXP; };
    float* cam = smem + L.total + BXF;                   // B*S*16
    float* tcen = cam + a.B * S * kCamStride;            // B*4
    float* stage_all = tcen + ((a.B * 4 + 15) & ~15);    // WAVES * S * 16 * SST

    // ---- prologue: stage weights + camera table ----
    {   // weight image (42-56 KB) -> LDS with every load in flight before the first store: the plain copy loop compiles to
        // load -> s_waitcnt vmcnt(0) -> ds_write per iteration (11-14 serial L2 round trips at the head of every block)
        constexpr int MAXW4 = (BX3 ? BXF : (R == 3 ? 10560 : 14528)) / 4;    // nerf_layout(11 | 35).total / 4 (the launcher checks)
        constexpr int NWI = (MAXW4 + 64 * WAVES - 1) / (64 * WAVES);
        float4 wq[NWI];
        const int n4 = (BX3 ? BXF : L.total) / 4;
        const float* src = a.packed + (BX3 ? LF.total : 0);
        float* dst = BX3 ? smem + L.total : wl;
#pragma unroll
        for (int it = 0; it < NWI; ++it) {
            const int i = (int)threadIdx.x + it * 64 * WAVES;
            wq[it] = *reinterpret_cast<const float4*>(src + (i < n4 ? i : n4 - 1) * 4);
        }
#pragma unroll
        for (int it = 0; it < NWI; ++it) {
            const int i = (int)threadIdx.x + it * 64 * WAVES;
            if (i < n4) *reinterpret_cast<float4*>(dst + i * 4) = wq[it];
        }
        if (BX3) {   // the small fp32 arrays (view_fc, biases, width-1 heads): nine short copies from their places in the full image
            const int so[9] = {LF.view, LF.viewb, LF.globb, LF.aggw, LF.fcb, LF.lr0b, LF.sigma, LF.c0b, LF.col2};
            const int dd[9] = {L.view, L.viewb, L.globb, L.aggw, L.fcb, L.lr0b, L.sigma, L.c0b, L.col2};
            const int nn[9] = {L.TR * 64, L.TR * 16, 32, 33, 16, 64, 65, 64, 65};
            for (int q = 0; q < 9; ++q)
                for (int i = threadIdx.x; i < nn[q]; i += blockDim.x) wl[dd[q] + i] = a.packed[so[q] + i];
        }
    }
    for (int i = threadIdx.x; i < a.B * (S + 1); i += blockDim.x) {
        int b = i / (S + 1), s = i - b * (S + 1);
        const float* E = (s < S) ? a.src_exts + ((long long)b * S + s) * 16 : a.tar_ext + (long long)b * 16;
        double m[16], inv[16];
        for (int k = 0; k < 16; ++k) m[k] = (double)E[k];
        bool ok = inv4x4(m, inv);
        float c0 = ok ? (float)inv[3] : NAN, c1 = ok ? (float)inv[7] : NAN, c2 = ok ? (float)inv[11] : NAN;
        if (s < S) {
            float* c = cam + ((long long)b * S + s) * kCamStride;
            const float* K = a.src_ixts + ((long long)b * S + s) * 9;
            // pix = K'·(E·[X,1]) (utils.py:698-703) evaluated as (K'·E33)·X + K'·t with the 3x4 product formed in fp64
            for (int r = 0; r < 3; ++r) {
                double kr[3];
                for (int q = 0; q < 3; ++q) kr[q] = (double)K[r * 3 + q] * (r < 2 ? (double)a.render_scale : 1.0);   // utils.py:700-701
                for (int q = 0; q < 4; ++q) {
                    const double v = kr[0] * m[q] + kr[1] * m[4 + q] + kr[2] * m[8 + q];
                    if (q < 3) c[r * 3 + q] = (float)v; else c[9 + r] = (float)v;
                }
            }
            c[12] = c0; c[13] = c1; c[14] = c2; c[15] = 0.f;
        } else {
            tcen[b * 4 + 0] = c0; tcen[b * 4 + 1] = c1; tcen[b * 4 + 2] = c2; tcen[b * 4 + 3] = 0.f;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave_in_block = threadIdx.x >> 6;
    // device-side ray selection (network_human.py:90-93): the number of rays is read here, not known to the host
    const long long nrays = a.ray_index != nullptr ? (long long)a.ray_count[0] : (long long)a.B * a.N;
    const long long ntiles = cdivl(nrays, 16);
    const bool scatter = a.ray_index != nullptr && a.scatter_rgb != 0;
    const int Ns = a.n_samples;
    const float* wlane = wl + lane;
#define A_VIEW(e) wlane[L.view + (e) * 64]
#define A_GLOB(e) wlane[L.glob + (e) * 64]
#define A_FC(e) wlane[L.fc + (e) * 64]
#define A_LR0(e) wlane[L.lr0 + (e) * 64]
#define A_C0P(e) wlane[L.c0p + (e) * 64]
#define A_C0V(e) wlane[L.c0v + (e) * 64]
    float* stage = stage_all + wave_in_block * ((PFK ? 2 : 1) * S * 16 * SST);
    const int sv = g < S ? g : S - 1;                    // the source view this lane's geometry works on
    float* my_rec = stage + (sv * 16 + j) * SST;         // record this lane writes (lane groups >= S write nothing)
    const float* rd_rec = stage + j * SST + g * R;       // + s*16*SST: channels [gR, gR+R) of view s, point j
    const float* rd_dir = stage + j * SST + TEX + g;     // + s*16*SST: direction-code component g of view s

    // Tile order (round 4, profiles/r04_ab_warp_variants_render_xcd_band.txt): XCD k (= block id % 8) renders the k-th eighth of
    // the tile list = an image row band, round-robin over ITS blocks, so the texels its gathers touch stay in its own L2:
    // FETCH_SIZE 241 -> 159 MB per launch at the same kernel time (190.2 vs 190.6 us).  (Round 3's variant — one contiguous run
    // of tiles per BLOCK — measured ~3 % slower; 1=0 is the plain round-robin order.)
    const long long nbands = gridDim.x < 8 ? gridDim.x : 8;                      // (small launches: fewer blocks than XCDs)
    const long long xcd_ = blockIdx.x % nbands, nb8 = ((long long)gridDim.x + nbands - 1 - xcd_) / nbands, bi_ = blockIdx.x / nbands;
    const long long t_hi = ntiles * (xcd_ + 1) / nbands;
// The level-0 kernel (R = 9; lego: 2500 tiles for 2048 waves) gains even more ALONE from the balanced deal (266 -> 201 us = 0.50 of
// the fp32-MFMA peak: a SIMD carries 3 tiles instead of 4).  In the frame it runs FORKED beside level 1; as one persistent block per CU
// the balanced form held every CU to the end and delayed level 1 (lego 543.9 -> 535.3 frames/s, profiles/r05_ab_render_balance.txt).
// Round 6: the forked launch is capped at half of the CUs (enerf_render_args_t.max_blocks, frame.hip), which turns the balanced deal
// into the better one there too (lego 543 -> 558 frames/s; unbalanced on half of the CUs 553): default on.
#ifndef ENERF_RENDER_BALANCE_R9
#define ENERF_RENDER_BALANCE_R9 1
#endif
    // Round 5: the band's PARTIAL last round of tiles is dealt out evenly.  Round-robin, it went to the band's first blocks, 12
    // tiles each: at dtu (2560 tiles per band = 6 full rounds of 32 x 12 + 256) 21 of a band's 32 CUs rendered 84 tiles and 10
    // rendered 72 — the launch ends with the 84s, 5 % above the 80-tile average.  Now every block takes rem / nb8 of them
    // (+- 1) on its FIRST waves: consecutive waves of a block sit on different SIMDs, so a SIMD's three waves end up with
    // 7 + 7 + 6 tiles everywhere.  The full rounds keep the interleaved order (the band's blocks walk the same image rows).
    const long long t_lo = ntiles * xcd_ / nbands, per_round = nb8 * WAVES, full = (t_hi - t_lo) / per_round;
    const long long rem = (t_hi - t_lo) - full * per_round, rb = rem / nb8, rx = rem - rb * nb8;
    const long long my_rem = rb + (bi_ < rx ? 1 : 0), my_rem_lo = bi_ * rb + (bi_ < rx ? bi_ : rx);
    constexpr bool kBal = R == 3 || ENERF_RENDER_BALANCE_R9;
    const long long n_it = kBal ? full + (wave_in_block < my_rem ? 1 : 0)
                                : full + (full * per_round + bi_ * WAVES + wave_in_block < t_hi - t_lo ? 1 : 0);
    for (long long it = 0; it < n_it; ++it) {
        const long long tile = (it < full || !kBal) ? t_lo + it * per_round + bi_ * WAVES + wave_in_block
                                                    : t_lo + full * per_round + my_rem_lo + wave_in_block;
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
        // ---- per-ray part of the voxel fetch (network.py:37 then utils.py:457): the (x, y) taps do not depend on the sample
        // 32-bit element offsets from the (uniform) tensor bases (the launcher checks both tensors hold < 2^30 floats: byte offsets fit 32 bits)
        const unsigned voff = (unsigned)b * (unsigned)(a.D * a.h * a.w * 8) + 2u * g;
        float wxy[4];
        int oxy[4];
        {
            const float gxv = (ru / (float)(a.Wr - 1)) * 2.f - 1.f, gyv = (rv / (float)(a.Hr - 1)) * 2.f - 1.f;
            float ix = gs_unnorm(gxv, a.w), iy = gs_unnorm(gyv, a.h);
            ix = fabsf(ix) < 1e8f ? ix : -10.f;
            iy = fabsf(iy) < 1e8f ? iy : -10.f;
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            float wx[2] = {(fx + 1.f) - ix, ix - fx}, wy[2] = {(fy + 1.f) - iy, iy - fy};
            int xo[2], yo[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int xx = x0 + c, yy = y0 + c;
                wx[c] = (unsigned)xx < (unsigned)a.w ? wx[c] : 0.f;      // zeros padding: weight 0 outside
                wy[c] = (unsigned)yy < (unsigned)a.h ? wy[c] : 0.f;
                xo[c] = min(max(xx, 0), a.w - 1);
                yo[c] = mul24(min(max(yy, 0), a.h - 1), a.w);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { wxy[c] = wx[c & 1] * wy[c >> 1]; oxy[c] = yo[c >> 1] + xo[c & 1]; }   // (wx*wy)*wz: ATen's association
        }
        // ---- this lane's view: camera constants of (b, sv) ----
        const float* cb = cam + ((long long)b * S + sv) * kCamStride;
        // LEANK (the level-1 build: 157 instead of 184 VGPRs -> three waves per SIMD): the 20 camera constants are re-read from
        // LDS per sample instead of living in registers, and two per-sample weights are kept instead of eight
        f32x4 cm0, cm1, cm2, cm3, tc4;
        if (!LEANK) { cm0 = lds4(cb); cm1 = lds4(cb + 4); cm2 = lds4(cb + 8); cm3 = lds4(cb + 12); tc4 = lds4(tcen + b * 4); }
        const unsigned toff = (unsigned)b * (unsigned)(S * a.Hr * a.Wr * TEX) + (unsigned)sv * (unsigned)(a.Hr * a.Wr * TEX);
        const float rcpW = fast_rcp((float)(a.Wr - 1)), rcpH = fast_rcp((float)(a.Hr - 1));

        float Tacc = 1.f;                 // transmittance, raw2outputs utils.py:588-589
        constexpr int NSM = (LEANK && R == 3) ? 2 : 8;      // samples kept per ray (lean level-1 build: Ns <= 2, the launcher checks)
        float wk[NSM];                    // per-sample weights
        float rgbacc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rgbacc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NSM; ++k) wk[k] = 0.f;

        // ---------- the gather of one sample, in two halves so that a sample's loads can be in flight during the PREVIOUS
        // sample's Agg MFMAs (PF, below): issue = placement, projection, taps, ALL loads, direction code;
        // finish = trilinear weights, blends, the (view, point) record into LDS, the voxel feature pair ----------
        struct GatherRegs {
            float2 vt[8];
            f32x4 tq[(R <= 3 ? 1 : 2)][4][(R <= 3 ? R : 3)];
            unsigned tp[4];                 // BYTE offsets from a.tex: `uniform base + 32-bit lane offset` is the saddr form of
                                            // global_load (no 64-bit VALU address per gather)
            float tw[4], wz[2];
            f32x4 dirc;
        };
        constexpr int QB = R <= 3 ? R : 3, NRND = R / QB, NBUF = NRND > 1 ? 2 : 1;
        static_assert(R % QB == 0, "texel chunk rounds");
        const char* const texb = reinterpret_cast<const char*>(a.tex);
        const char* const volb = reinterpret_cast<const char*>(a.vol);
        auto gather_issue = [&](int k, GatherRegs& G) {
            if (LEANK) { cm0 = lds4(cb); cm1 = lds4(cb + 4); cm2 = lds4(cb + 8); cm3 = lds4(cb + 12); tc4 = lds4(tcen + b * 4); }
            // ---------- sample placement (utils.py:425-436) ----------
            float tk = (Ns == 1) ? 0.5f : linspace01(k, Ns);
            float z = rn + (rf - rn) * tk;
            float zz = a.depth_inv ? fast_rcp(clamp_min(z, 1e-6f)) : z;
            float X = ox + dx * zz, Y = oy + dy * zz, Z = oz + dz * zz;
            float dn = a.depth_inv ? (vn - z) * fast_rcp(clamp_min(vn - vf, 1e-6f))
                                   : (z - vn) * fast_rcp(clamp_min(vf - vn, 1e-6f));
            // ---------- addresses first, then ALL gathers of the sample in flight together, then the arithmetic that does
            // not need them, then the blends.  The sched_barriers pin that order: left alone, hipcc sinks every load to
            // its first use under register pressure — eight voxel taps became eight serial memory round trips.
            // this lane's view of the point: projection + bilinear taps, border padding (utils.py:698-706)
            const float px = X * cm0[0] + Y * cm0[1] + Z * cm0[2] + cm2[1];
            const float py = X * cm0[3] + Y * cm1[0] + Z * cm1[1] + cm2[2];
            const float pz = X * cm1[2] + Y * cm1[3] + Z * cm2[0] + cm2[3];
            const float rz = fast_rcp(clamp_min(pz, 1e-6f));
            const float gx = ((px * rz) * rcpW) * 2.f - 1.f, gy = ((py * rz) * rcpH) * 2.f - 1.f;
            const Taps2 t = gs_taps2<true>(gs_unnorm(gx, a.Wr), gs_unnorm(gy, a.Hr), a.Wr, a.Hr);
            const int r0 = mul24(t.y0, a.Wr), r1 = mul24(t.y1, a.Wr);
            G.tp[0] = (toff + (unsigned)mul24(r0 + t.x0, TEX)) * 4u; G.tp[1] = (toff + (unsigned)mul24(r0 + t.x1, TEX)) * 4u;
            G.tp[2] = (toff + (unsigned)mul24(r1 + t.x0, TEX)) * 4u; G.tp[3] = (toff + (unsigned)mul24(r1 + t.x1, TEX)) * 4u;
            G.tw[0] = t.w00; G.tw[1] = t.w01; G.tw[2] = t.w10; G.tw[3] = t.w11;
            // voxel feature: trilinear, zeros padding (utils.py:457); lane group g fetches channels 2g, 2g+1
            float iz = gs_unnorm(dn * 2.f - 1.f, a.D);
            iz = fabsf(iz) < 1e8f ? iz : -10.f;
            const float fz = floorf(iz);
            const int z0 = (int)fz;
            G.wz[0] = (fz + 1.f) - iz; G.wz[1] = iz - fz;
            int zo[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int zc = z0 + c;
                G.wz[c] = (unsigned)zc < (unsigned)a.D ? G.wz[c] : 0.f;
                zo[c] = mul24(min(max(zc, 0), a.D - 1), a.h * a.w);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 8; ++c)                 // order tnw,tne,tsw,tse,bnw,bne,bsw,bse
                G.vt[c] = *reinterpret_cast<const float2*>(volb + ((unsigned)(zo[c >> 2] + oxy[c & 3]) * 8u + voff) * 4u);
            // texel channels are gathered QB float4 chunks per tap at a time; R = 9 takes three rounds, two of them in flight
#pragma unroll
            for (int rd = 0; rd < NBUF; ++rd)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int q = 0; q < QB; ++q) G.tq[rd][c][q] = lds4(reinterpret_cast<const float*>(texb + G.tp[c]) + 4 * (rd * QB + q));
            __builtin_amdgcn_sched_barrier(0);
            // direction code (utils.py:707-720) while the gathers are in flight
            float tx = X - tc4[0], ty = Y - tc4[1], tz = Z - tc4[2];
            float sx = X - cm3[0], sy = Y - cm3[1], sz = Z - cm3[2];
            const float tir = fast_rcp(fast_sqrt(tx * tx + ty * ty + tz * tz) + 1e-6f);
            const float sir = fast_rcp(fast_sqrt(sx * sx + sy * sy + sz * sz) + 1e-6f);
            tx *= tir; ty *= tir; tz *= tir;
            sx *= sir; sy *= sir; sz *= sir;
            const float ex = tx - sx, ey = ty - sy, ez = tz - sz;
            const float eir = fast_rcp(fmaxf(fast_sqrt(ex * ex + ey * ey + ez * ez), 1e-6f));
            G.dirc = f32x4{ex * eir, ey * eir, ez * eir, tx * sx + ty * sy + tz * sz};
        };
        auto gather_finish = [&](GatherRegs& G, float* rec, float (&vox)[2]) {
            float vw[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) vw[c] = wxy[c & 3] * G.wz[c >> 2];
            __builtin_amdgcn_sched_barrier(0);
            f32x4 blend[R];
#pragma unroll
            for (int rd = 0; rd < NRND; ++rd) {
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    f32x4 acc = G.tq[rd % NBUF][0][q] * G.tw[0];
                    acc += G.tq[rd % NBUF][1][q] * G.tw[1];
                    acc += G.tq[rd % NBUF][2][q] * G.tw[2];
                    acc += G.tq[rd % NBUF][3][q] * G.tw[3];
                    blend[rd * QB + q] = acc;
                }
                if (rd + NBUF < NRND) {                 // refill the buffer just consumed (R = 9 only)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int q = 0; q < QB; ++q) G.tq[rd % NBUF][c][q] = lds4(reinterpret_cast<const float*>(texb + G.tp[c]) + 4 * ((rd + NBUF) * QB + q));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (g < S) {
#pragma unroll
                for (int q = 0; q < R; ++q) *reinterpret_cast<f32x4*>(rec + 4 * q) = blend[q];
                *reinterpret_cast<f32x4*>(rec + TEX) = G.dirc;
            }
            vox[0] = 0.f; vox[1] = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                vox[0] += G.vt[c].x * vw[c];
                vox[1] += G.vt[c].y * vw[c];
            }
        };
        // PF: software pipelining across the samples of a ray tile (R = 3 only: the 74 registers of a sample in flight fit
        // beside the Agg phase, not beside the wider level-0 MLP).  Records are double-buffered in LDS.
        constexpr bool PF = PFK;
        constexpr int RECBUF = S * 16 * SST;              // floats of one record buffer of a wave
        GatherRegs GR;
        float vox_next[2] = {0.f, 0.f};
        if (PF) {
            gather_issue(0, GR);
            gather_finish(GR, my_rec, vox_next);
        }
#pragma unroll 1
        for (int k = 0; k < Ns; ++k) {
            // Keep the MFMA A operands (weights) in LDS (measured: holding all 155 tiles in registers at one wave per SIMD,
            // 512-entry budget, is 40 % slower — hipcc parks them in AGPRs and copies each back with v_accvgpr_read).
            asm volatile("" ::: "memory");
            float vox[2];
            const int rb = PF ? (k & 1) * RECBUF : 0;     // this sample's record buffer
            if (!PF) {
                gather_issue(k, GR);
                gather_finish(GR, my_rec, vox);
            } else {
                vox[0] = vox_next[0]; vox[1] = vox_next[1];
            }
            wave_sync();                              // the records of this tile are written: read them back as B operands
            float x[S][R], dsel[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int r = 0; r < R; ++r) x[s][r] = rd_rec[rb + s * 16 * SST + r];
                dsel[s] = rd_dir[rb + s * 16 * SST];
            }
            wave_sync();                              // ... before the next sample overwrites them
            const bool more = PF && k + 1 < Ns;           // uniform
            if (more) gather_issue(k + 1, GR);            // the next sample's loads fly during this sample's Agg phase

            // ---------- Agg (nerf.py:74-89) ----------
            float av[S][R];
            {
                float aview[TR];
#pragma unroll
                for (int t = 0; t < TR; ++t) aview[t] = A_VIEW(t);
                f32x4 vb[TR];
#pragma unroll
                for (int t = 0; t < TR; ++t) vb[t] = lds4(wl + L.viewb + t * 16 + 4 * g);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    f32x4 va[TR];
#pragma unroll
                    for (int t = 0; t < TR; ++t) va[t] = ENERF_MFMA(aview[t], dsel[s], vb[t]);
#pragma unroll
                    for (int r = 0; r < R; ++r) av[s][r] = x[s][r] + relu1(va[r >> 2][r & 3]);
                }
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
            f32x4 gf[S][2];
            float aw[S];
            const f32x4 aggw0 = lds4(wl + L.aggw + 4 * g), aggw1 = lds4(wl + L.aggw + 16 + 4 * g);
            const float aggb = wl[L.aggw + 32];
            if constexpr (BX3) {
                // global_fc on the bf16 cores: one 16-deep chunk each for the variance, the mean and a view's a_s slots
                const BxLayout BL = bx_layout();
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                {
                    bfx8 vm[NP];
                    bx_split<NP>(f32x4{var[0], var[1], var[2], 0.f}, f32x4{mean[0], mean[1], mean[2], 0.f}, vm);
                    bx_mma_n<NP, 2>(bxplane, BL.gvm, lane, vm, P);
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    bfx8 ap[NP];
                    bx_split<NP>(f32x4{av[s][0], av[s][1], av[s][2], 0.f}, zero4, ap);
                    gf[s][0] = P[0]; gf[s][1] = P[1];
                    bx_mma_n<NP, 2>(bxplane, BL.ga, lane, ap, gf[s]);
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    gf[s][0] = relu4(gf[s][0]); gf[s][1] = relu4(gf[s][1]);
                    float part = dot4(gf[s][1], aggw1, dot4(gf[s][0], aggw0, 0.f));
                    aw[s] = relu1(group_sum(part) + aggb);
                }
            } else {
            // k-steps 0..R-1: variance slots, R..2R-1: mean slots (tiles (1*R + r)*2 + u and (2*R + r)*2 + u)
            mfma_chain<2 * R, 2>(P, [&](int e) { return A_GLOB(2 * R + e); },
                                 [&](int ks) { return ks < R ? var[ks < R ? ks : 0] : mean[ks >= R ? ks - R : 0]; });
            {
                float ag[R * 2];                          // the per-view tiles are the same for every view: read them once
#pragma unroll
                for (int e = 0; e < R * 2; ++e) ag[e] = A_GLOB(e);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    gf[s][0] = P[0]; gf[s][1] = P[1];
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int u = 0; u < 2; ++u) gf[s][u] = ENERF_MFMA(ag[r * 2 + u], av[s][r], gf[s][u]);
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    gf[s][0] = relu4(gf[s][0]); gf[s][1] = relu4(gf[s][1]);
                    float part = dot4(gf[s][1], aggw1, dot4(gf[s][0], aggw0, 0.f));
                    aw[s] = relu1(group_sum(part) + aggb);
                }
            }
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
            f32x4 aggv[1] = {lds4(wl + L.fcb + 4 * g)};
            if constexpr (BX3) {
                const BxLayout BL = bx_layout();
                bfx8 gp[NP];
                bx_split<NP>(G[0], G[1], gp);
                bx_mma_n<NP, 1>(bxplane, BL.fc, lane, gp, aggv);
            } else {
                float afc[8];                             // one accumulator, eight dependent k-steps: all A values up front
#pragma unroll
                for (int e = 0; e < 8; ++e) afc[e] = A_FC(e);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 8; ++e) aggv[0] = ENERF_MFMA(afc[e], G[e >> 2][e & 3], aggv[0]);
            }
            const f32x4 agg = relu4(aggv[0]);

            if (more) gather_finish(GR, my_rec + ((k + 1) & 1) * RECBUF, vox_next);   // blends + record of sample k+1 (other buffer)

            // ---------- NeRF trunk (nerf.py:33-37) ----------
            f32x4 hid[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) hid[v] = lds4(wl + L.lr0b + v * 16 + 4 * g);
            bfx8 vap[NP];                                // (bf16 variants) [voxel pair | agg] feeds lr0 AND color.0: split once
            if constexpr (BX3) {
                const BxLayout BL = bx_layout();
                bx_split<NP>(f32x4{vox[0], vox[1], 0.f, 0.f}, agg, vap);
                bx_mma_n<NP, 4>(bxplane, BL.lr0, lane, vap, hid);
            } else
            mfma_chain<6, 4>(hid, [&](int e) { return A_LR0(e); }, [&](int ks) { return ks < 2 ? vox[ks < 2 ? ks : 0] : agg[ks >= 2 ? ks - 2 : 0]; });
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
            if constexpr (BX3) {
                const BxLayout BL = bx_layout();
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    bfx8 hp[NP];
                    bx_split<NP>(hid[2 * pr], hid[2 * pr + 1], hp);
                    bx_mma_n<NP, 4>(bxplane, BL.c0p + 4 * pr, lane, hp, P2);
                }
                bx_mma_n<NP, 4>(bxplane, BL.c0p + 8, lane, vap, P2);
            } else
            mfma_chain<22, 4>(P2, [&](int e) { return A_C0P(e); },
                              [&](int ks) { return ks < 16 ? hid[(ks < 16 ? ks : 0) >> 2][ks & 3]
                                                           : (ks < 18 ? vox[ks < 18 ? ks - 16 : 0] : agg[ks >= 18 ? ks - 18 : 0]); });
            float cl[S];
            const float c2b = wl[L.col2 + 64];
            if constexpr (BX3) {
                const BxLayout BL = bx_layout();
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                f32x4 c2w[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) c2w[v] = lds4(wl + L.col2 + v * 16 + 4 * g);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    bfx8 xp[NP];
                    bx_split<NP>(f32x4{x[s][0], x[s][1], x[s][2], dsel[s]}, zero4, xp);
                    f32x4 cc[4] = {P2[0], P2[1], P2[2], P2[3]};
                    bx_mma_n<NP, 4>(bxplane, BL.c0v, lane, xp, cc);
                    float part = 0.f;
#pragma unroll
                    for (int v = 0; v < 4; ++v) part = dot4(relu4(cc[v]), c2w[v], part);
                    cl[s] = relu1(group_sum(part) + c2b);
                }
            } else
            {
                float acv[(R + 1) * 4];                   // the per-view tiles of color.0, shared by the S views
#pragma unroll
                for (int e = 0; e < (R + 1) * 4; ++e) acv[e] = A_C0V(e);
                f32x4 c2w[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) c2w[v] = lds4(wl + L.col2 + v * 16 + 4 * g);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    f32x4 cc[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) cc[v] = P2[v];
#pragma unroll
                    for (int ks = 0; ks <= R; ++ks) {
                        float bop = ks < R ? x[s][ks < R ? ks : 0] : dsel[s];
#pragma unroll
                        for (int v = 0; v < 4; ++v) cc[v] = ENERF_MFMA(acv[ks * 4 + v], bop, cc[v]);
                    }
                    float part = 0.f;
#pragma unroll
                    for (int v = 0; v < 4; ++v) part = dot4(relu4(cc[v]), c2w[v], part);
                    cl[s] = relu1(group_sum(part) + c2b);
                }
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
            for (int kk = 0; kk < NSM; ++kk)
                if (kk == k) wk[kk] = wgt;
        }

        // ---------- depth from softmaxed weights (utils.py:593-595) + stores ----------
        float m = wk[0];
#pragma unroll
        for (int k = 1; k < NSM; ++k) if (k < Ns) m = fmaxf(m, wk[k]);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < NSM; ++k) if (k < Ns) { wk[k] = fast_exp(wk[k] - m); se += wk[k]; }
        se = fast_rcp(se);
        float depth = 0.f, accw = 0.f;
#pragma unroll
        for (int k = 0; k < NSM; ++k)
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
                for (int k = 0; k < NSM; ++k) if (k < Ns) a.weights[ray * Ns + k] = wk[k];
            }
        }
    }
}

template <int R, int WAVES, int OCC, bool PFK = false, bool LEANK = false, int BX3 = 0>
static int dispatch_s(const RenderArgs& a, unsigned grid, size_t shmem, hipStream_t st) {
    constexpr int WPE = (WAVES * OCC + 3) / 4;
    switch (a.S) {
        case 2: ENERF_LAUNCH((k_render_rays<R, 2, WAVES, WPE, PFK, LEANK, BX3>), grid, 64 * WAVES, shmem, st, a); return 0;
        case 3: ENERF_LAUNCH((k_render_rays<R, 3, WAVES, WPE, PFK, LEANK, BX3>), grid, 64 * WAVES, shmem, st, a); return 0;
        case 4: ENERF_LAUNCH((k_render_rays<R, 4, WAVES, WPE, PFK, LEANK, BX3>), grid, 64 * WAVES, shmem, st, a); return 0;
        default: return -3;
    }
}
template <int R, int WAVES>
static size_t render_shmem(const RenderArgs& a, int record_buffers = 1, int bx = 0) {
    return ((size_t)nerf_layout(a.F, bx != 0).total + (size_t)(bx ? bx_image_floats(a.F, bx == 6 ? 3 : 2) : 0) + (size_t)a.B * a.S * kCamStride + (size_t)((a.B * 4 + 15) & ~15) +
            (size_t)WAVES * record_buffers * a.S * 16 * Stage<R>::kStride) * sizeof(float);
}
int launch_render_rays(const RenderArgs& a, hipStream_t st) {
    if (a.n_samples < 1 || a.n_samples > 8) return -1;
    const int R = (a.F + 3) / 4;
    // 32-bit BYTE offsets and 24-bit index multiplies inside the kernel
    if ((long long)a.B * a.S * a.Hr * a.Wr * 4 * R >= (1LL << 30) || (long long)a.B * a.D * a.h * a.w * 8 >= (1LL << 30) ||
        (long long)a.Hr * a.Wr >= (1LL << 23) || (long long)a.h * a.w >= (1LL << 23) || a.D >= (1 << 23))
        return -5;
    const long long ntiles = cdivl((long long)a.B * a.N, 16);
    // Persistent waves: the weight image (41-56 KB) is staged into LDS once per block, so launch only as many blocks as
    // are co-resident (blocks per CU x CUs) and let each wave stride over ray tiles.
    // Measured on MI355X (512x640, S = 3): 4-wave blocks x 2 per CU (2 waves per SIMD, 187 VGPRs) 210 us; 12-wave blocks
    // (3 per SIMD, 168 VGPRs + spills) 215 us; 16-wave blocks (4 per SIMD) 247 us; 1 wave per SIMD 350 us.
    const int cus = device_cu_count();
    auto grid_for = [&](int waves, int occ) {
        const long long blocks = cdivl(ntiles, waves);
        long long resident = (long long)cus * occ;
        if (a.max_blocks > 0 && a.max_blocks < resident) resident = a.max_blocks;       // a render that shares the device (frame.hip)
        return (unsigned)(blocks < resident ? blocks : resident);
    };
#ifndef ENERF_R9_LEAN
#define ENERF_R9_LEAN 1              // level-0 kernel (C = 32): camera constants re-read from LDS per sample (fewer spills)
#endif
#ifndef ENERF_RENDER_LEAN12
#define ENERF_RENDER_LEAN12 1        // 0: the round-2 launch shape for every level-1 render (A/B)
#endif
#ifndef ENERF_RENDER_WAVES
#define ENERF_RENDER_WAVES 4
#endif
#ifndef ENERF_RENDER_OCC
#define ENERF_RENDER_OCC 2
#endif
    if (nerf_layout(a.F).total > (R == 3 ? 10560 : 14528)) return -6;      // the kernel's weight-staging bound (F = 11 / 35)
    if (R == 3 && a.n_samples <= 2 && ENERF_RENDER_LEAN12) {        // C = 8, the cascade's last level: 12-wave blocks, one per CU
        // Measured (profiles/r03_render_waves.txt, S = 3): 4-wave blocks x 2 per CU at 184 VGPRs 199.5 us; the same with the lean
        // register set (157) 196.5; 12-wave blocks x 1 (THREE waves per SIMD, 159 VGPRs, no spills) 190.7; 6-wave blocks x 2 254.8
        // (uneven over the SIMDs).  Round 2's 12-wave attempt spilled at 168 VGPRs (215 us).
        // enerf_options_t.render_precision: 0 / 1 = exact fp32 MFMAs (default), 2 = bf16x3 (opt-in fast mode), 3 = bf16x6 (A/B)
        const int prec = a.options != nullptr ? a.options->render_precision : 0;
        const int bx = prec == 2 ? 3 : (prec == 3 ? 6 : 0);
        const size_t shmem = render_shmem<3, 12>(a, 1, bx);
        if (shmem <= 160 * 1024) {
            const unsigned grid = grid_for(12, 1);
            if (grid == 0) return 0;
            if (bx == 6) return dispatch_s<3, 12, 1, false, true, 6>(a, grid, shmem, st);
            if (bx == 3) return dispatch_s<3, 12, 1, false, true, 3>(a, grid, shmem, st);
            return dispatch_s<3, 12, 1, false, true>(a, grid, shmem, st);
        }
    }
    if (R == 3) {                                                                                              // C = 8
        const size_t shmem = render_shmem<3, ENERF_RENDER_WAVES>(a);
        if (shmem > 160 * 1024 / ENERF_RENDER_OCC) return -2;
        const unsigned grid = grid_for(ENERF_RENDER_WAVES, ENERF_RENDER_OCC);
        if (grid == 0) return 0;
        // sample-pipelined variant (two record buffers per wave) when it still fits two blocks per CU and there is a next sample
        const size_t shmem_pf = render_shmem<3, ENERF_RENDER_WAVES>(a, 2);
        if (ENERF_RENDER_PREFETCH != 0 && a.n_samples > 1 && shmem_pf <= 160 * 1024 / ENERF_RENDER_OCC)
            return dispatch_s<3, ENERF_RENDER_WAVES, ENERF_RENDER_OCC, true>(a, grid, shmem_pf, st);
        return dispatch_s<3, ENERF_RENDER_WAVES, ENERF_RENDER_OCC>(a, grid, shmem, st);
    }
    if (R == 9) {                                                                   // C = 32: one 8-wave block per CU
        const size_t shmem = render_shmem<9, 8>(a);
        if (shmem > 160 * 1024) return -2;
        unsigned grid = grid_for(8, 1);
        if (0 && ENERF_RENDER_BALANCE_R9 && ntiles > 0) {
            const long long t = cdivl(ntiles, 4LL * cus), tight = cdivl(ntiles, 4 * t);
            if (tight < (long long)grid) grid = (unsigned)tight;
        }
        if (grid == 0) return 0;
        return dispatch_s<9, 8, 1, false, ENERF_R9_LEAN != 0>(a, grid, shmem, st);
    }
    return -4;
}

}  // namespace enerf
