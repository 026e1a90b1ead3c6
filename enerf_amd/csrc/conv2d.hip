// conv2d.hip — the 2-D FPN feature extractor (FeatureNet, feature_net.py:4-36; ConvBnReLU utils.py:10-20)
// as fp32-MFMA implicit GEMMs with LDS-staged input tiles, channels-last activations.
//
// Same scheme as conv3d.hip's V2 kernel, specialised to 2-D and generalised to k in {1,3,5}, stride in
// {1,2}:  D[cout][pixel] += W[cout][k]·X[k][pixel] on v_mfma_f32_16x16x4_f32, weights = A operand (packed
// once, streamed from L1/L2 with one-row-of-taps-ahead prefetch), pixels = B/D columns (16 consecutive x
// per column tile), the haloed input tile of one 16-channel block staged once per block in LDS so every
// tap is one ds_read_b128.  Epilogue fuses BN (scale/shift) or bias, ReLU, and — for the FPN lateral
// convs — the bilinear x2 (align_corners) upsample-add of the coarser map (feature_net.py:24-25).
// Outputs are channels-last, i.e. exactly the layout the warp/variance and render kernels gather from,
// so the NCHW->NHWC adapter kernels disappear when this path is used.
//
// First layer (Cin=3): input is the NCHW image batch; it is staged as [pixel][4] (4th channel 0) so a
// tap is one k-step with lane group g supplying channel g.

#include "kernels.h"
#include "prep_job.h"
#include <cstring>

namespace enerf {

__host__ __device__ __forceinline__ int c2_cinp(int cin) { return cin <= 4 ? 4 : cin; }             // padded Cin
__host__ __device__ __forceinline__ int c2_cb(int cinp) { return cinp >= 16 ? 16 : cinp; }         // channels / LDS pass
long long conv2d_packed_floats(int cin, int cout, int k) {
    return (long long)k * k * (c2_cinp(cin) / 4) * cdiv(cout, 16) * 64;
}

// packed[((tap*KS + ks)*RT + rt)*64 + lane], lane=(g,i): W[cout = rt*16+i][cin = chan(ks,g)][kh][kw]
__global__ __launch_bounds__(256) void k_conv2d_pack(const float* __restrict__ w, const float* __restrict__ bias,
                                                     const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                     const float* __restrict__ bn_mean, const float* __restrict__ bn_var,
                                                     float eps, int cin, int cout, int k, float* __restrict__ packed,
                                                     float* __restrict__ scale, float* __restrict__ shift) {
    const int cinp = c2_cinp(cin), CB = c2_cb(cinp), CPL = CB / 4, KS = cinp / 4, RT = cdiv(cout, 16);
    const long long total = (long long)k * k * KS * RT * 64;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < RT * 16) {
        int co = (int)i;
        float sc = 1.f, sh = 0.f;
        if (co < cout) {
            if (bn_w != nullptr) {
                sc = bn_w[co] / sqrtf(bn_var[co] + eps);
                sh = bn_b[co] - bn_mean[co] * sc;
            } else if (bias != nullptr) {
                sh = bias[co];
            }
        }
        scale[co] = sc;
        shift[co] = sh;
    }
    if (i >= total) return;
    const int lane = (int)(i & 63);
    long long q = i >> 6;
    const int rt = (int)(q % RT); q /= RT;
    const int ks = (int)(q % KS);
    const int tap = (int)(q / KS);
    const int g = lane >> 4, co = rt * 16 + (lane & 15);
    const int cb = ks / CPL, r = ks - cb * CPL;
    const int ci = cb * CB + g * CPL + r;
    float v = 0.f;
    if (co < cout && ci < cin) v = w[((long long)co * cin + ci) * k * k + tap];
    packed[i] = v;
}
void launch_conv2d_pack(const float* w, const float* bias, const float* bn_w, const float* bn_b, const float* bn_mean,
                        const float* bn_var, float eps, int cin, int cout, int k, float* packed, float* scale,
                        float* shift, hipStream_t st) {
    long long total = conv2d_packed_floats(cin, cout, k);
    ENERF_LAUNCH_SIMPLE(k_conv2d_pack, (unsigned)cdivl(total, 256), 256, 0, st, w, bias, bn_w, bn_b, bn_mean, bn_var, eps,
                        cin, cout, k, packed, scale, shift);
}

// CINP: padded input channels (4,8,16,32); RT: cout tiles of 16; K: 1/3/5; STR: 1/2; TH: output rows per block
// (x TW=32 columns = 2 column tiles per row); NCHW3: input is the (n,3,H,W) image batch.
// CHAIN: a following 1x1 convolution (RT*16 -> 32 channels, bias, no activation: FeatureNet.toplayer) is applied in the
// epilogue.  This layer's D registers (rows 16rt+4g+r of pixel j) are exactly the B operands of the next layer's
// k-steps in the standard packed-weight order (ci = 16cb + 4g + r), so the chained layer is 8*RT more MFMAs per
// column tile on values that never leave the registers; this layer's own output is not stored.
template <int CINP, int RT, int K, int STR, int TH, bool NCHW3, bool CHAIN = false>
__global__ __launch_bounds__(256) void k_conv2d(const float* __restrict__ wpk, const float* __restrict__ scale,
                                                const float* __restrict__ shift, const float* __restrict__ in,
                                                float* __restrict__ out, const float* __restrict__ up,
                                                const float* __restrict__ rgb_src, int out_stride, int cout,
                                                int relu, int N, int Hi, int Wi, int Ho, int Wo, int Hc, int Wc,
                                                int tiles_y, int tiles_x, const float* __restrict__ chain_w,
                                                const float* __restrict__ chain_shift) {
    constexpr int TW = 32, P = (K - 1) / 2;
    constexpr int CB = CINP >= 16 ? 16 : CINP, CPL = CB / 4, NCB = CINP / CB, KS = CINP / 4;
    constexpr int IH = (TH - 1) * STR + K, IW = (TW - 1) * STR + K;
    constexpr int NT = TH * (TW / 16), CTW = NT / 4;      // column tiles per block / per wave
    constexpr int QV = CB / 4 > 0 ? CB / 4 : 1;
    constexpr int NPX = IH * IW;
    // Stride-2 layers read every other tile pixel per lane: with the tile stored pixel-major, lane j sits 2 pixels
    // (64 / 128 B) after lane j-1 and a wave touches half of the LDS banks (PMC: bank-conflict cycles 0.85 of the LDS-active
    // cycles in conv1.0).  So their rows are stored de-interleaved — even pixels, then odd pixels — and a tap (kw) reads
    // consecutive slots of one parity.
    constexpr int IWH = (IW + 1) / 2;
    auto slot = [](int ly, int lx) { return STR == 2 ? (ly * 2 + (lx & 1)) * IWH + (lx >> 1) : ly * IW + lx; };
    // (round 5 measured the tile as channel-quad PLANES — conflict-free ds_read_b128 / ds_write_b128 groups — with no change on any
    // layer: the bank conflicts PMC counts do not bound these kernels; profiles/r05_ab_conv2d_planar.txt, tools/patches/)
    ENERF_DYN_SMEM(float, lds);

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    int bid = blockIdx.x;
    {   // XCD-contiguous block order (bijective), see conv3d.hip
        const int nblk = gridDim.x, q = nblk / 8, r = nblk % 8, xcd = bid % 8, kk = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
    }
    const int tx = bid % tiles_x;
    const int ty = (bid / tiles_x) % tiles_y;
    const int n = bid / (tiles_x * tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * STR - P, ix0 = ox0 * STR - P;

    f32x4 acc[CTW][RT];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wl = wpk + lane;
    constexpr int NAQ = K * CPL * RT;
    // PRE: the whole pass's weights are requested before the tile is staged and stay in registers (k <= 3 always; round 5: also the
    // 5x5 stride-2 layer with few channels, conv1.0: 50 registers — its per-row operand ring cost 11 of 79 us at zju sizes, timing
    // ablation profiles/r05_conv2d_ablation.txt).  conv2.0 (200 registers) keeps the two-row ring.
#ifndef ENERF_C2_PRE_REGS
#define ENERF_C2_PRE_REGS 64
#endif
    constexpr bool PRE = K <= 3 || K * NAQ <= ENERF_C2_PRE_REGS;

#pragma unroll 1
    for (int cb = 0; cb < NCB; ++cb) {
        auto issue_a = [&](int kh, float (&aq)[NAQ]) {         // weights of the taps (kh, 0..K-1) for this pass
            const float* wt = wl + ((long long)(kh * K) * KS + cb * CPL) * RT * 64;
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int r = 0; r < CPL; ++r)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        aq[(kw * CPL + r) * RT + rt] = wt[((long long)(kw * KS + r) * RT + rt) * 64];
        };
        // k<=3: the whole pass's weights are requested BEFORE the tile is staged, so their L2 latency hides
        // behind the staging traffic; sched_barrier pins the loads here (hipcc otherwise sinks each load to
        // just before its MFMA and waits vmcnt(0) on it).
        float aq_all[PRE ? K : 1][NAQ];
        if (PRE) {
#pragma unroll
            for (int kh = 0; kh < K; ++kh) issue_a(kh, aq_all[kh]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cb > 0) __syncthreads();
        if (NCHW3) {
            // image batch (n,3,Hi,Wi): one thread per tile pixel, three coalesced plane reads
            constexpr int NIT = (NPX + 255) / 256;
            float v0[NIT], v1[NIT], v2[NIT];
            bool sk[NIT];
            const float* base = in + (long long)n * 3 * Hi * Wi;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * 256, ic = i < NPX ? i : NPX - 1;
                const int ly = ic / IW, lx = ic - ly * IW, gy = iy0 + ly, gx = ix0 + lx;
                sk[it] = gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
                const int off = sk[it] ? gy * Wi + gx : 0;
                v0[it] = base[off]; v1[it] = base[Hi * Wi + off]; v2[it] = base[2 * Hi * Wi + off];
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * 256;
                if (i < NPX)
                    *reinterpret_cast<float4*>(lds + i * 4) =
                        sk[it] ? make_float4(v0[it], v1[it], v2[it], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            constexpr int NITEM = NPX * QV;
            constexpr int NIT = (NITEM + 255) / 256;
            float4 sv[NIT];
            bool sk[NIT];
            int so[NIT];
            const float* base = in + (long long)n * Hi * Wi * CINP + cb * CB;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * 256, ic = i < NITEM ? i : NITEM - 1;
                const int px = ic / QV, q = ic - px * QV;
                const int ly = px / IW, lx = px - ly * IW, gy = iy0 + ly, gx = ix0 + lx;
                sk[it] = gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
                so[it] = (slot(ly, lx) * QV + q) * 4;
                const int off = sk[it] ? gy * Wi + gx : 0;
                sv[it] = *reinterpret_cast<const float4*>(base + (long long)off * CINP + q * 4);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * 256;
                if (i < NITEM)
                    *reinterpret_cast<float4*>(lds + so[it]) = sk[it] ? sv[it] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();

        // B operands (tile pixels from LDS) of tap (kh, kw) for this wave's column tiles
        auto read_b = [&](int kh, int kw, float (&bv)[CTW][4]) {
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int tile = wv * CTW + c, tr = tile / (TW / 16), tc = tile - tr * (TW / 16);
                const int sl = slot(tr * STR + kh, (tc * 16 + j) * STR + kw);
                const float* p = lds + sl * CB + g * CPL;
                if (CPL == 4) {
                    const float4 tq = *reinterpret_cast<const float4*>(p);
                    bv[c][0] = tq.x; bv[c][1] = tq.y; bv[c][2] = tq.z; bv[c][3] = tq.w;
                } else if (CPL == 2) {
                    const float2 tq = *reinterpret_cast<const float2*>(p);
                    bv[c][0] = tq.x; bv[c][1] = tq.y; bv[c][2] = 0.f; bv[c][3] = 0.f;
                } else {
                    bv[c][0] = p[0]; bv[c][1] = 0.f; bv[c][2] = 0.f; bv[c][3] = 0.f;
                }
            }
        };
        // One tap row; the B operands of tap kw+1 are read while the MFMAs of tap kw run (two-deep register ring pinned with
        // sched_barrier).  Measured (profiles/r03_conv2d_ablation.txt, r03_kv3): conv1.0 28.0 -> 25.9 us, the others unchanged.
        auto compute = [&](int kh, const float (&aq)[NAQ]) {
            float bq[2][CTW][4];
            read_b(kh, 0, bq[0]);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                if (kw + 1 < K) read_b(kh, kw + 1, bq[(kw + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < CPL; ++r)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int c = 0; c < CTW; ++c)
                            acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[(kw * CPL + r) * RT + rt], bq[kw & 1][c][r],
                                                                              acc[c][rt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (PRE) {
#pragma unroll
            for (int kh = 0; kh < K; ++kh) compute(kh, aq_all[kh]);
        } else {
            float a0[NAQ], a1[NAQ];
            issue_a(0, a0);
#pragma unroll 1
            for (int kh = 0; kh < K; kh += 2) {
                issue_a(kh + 1 < K ? kh + 1 : K - 1, a1);
                __builtin_amdgcn_sched_barrier(0);
                compute(kh, a0);
                issue_a(kh + 2 < K ? kh + 2 : K - 1, a0);
                __builtin_amdgcn_sched_barrier(0);
                if (kh + 1 < K) compute(kh + 1, a1);
            }
        }
    }

    if (CHAIN) {      // ---- epilogue with the chained 1x1 layer (see above); all lanes run the MFMAs, stores are guarded ----
        float ca[RT * 4][2], sc[RT][4], sh[RT][4];
#pragma unroll
        for (int ks = 0; ks < RT * 4; ++ks) {
            ca[ks][0] = chain_w[(ks * 2 + 0) * 64 + lane];
            ca[ks][1] = chain_w[(ks * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[rt][r] = scale[rt * 16 + 4 * g + r]; sh[rt][r] = shift[rt * 16 + 4 * g + r]; }
        const float4 b0 = *reinterpret_cast<const float4*>(chain_shift + 4 * g), b1 = *reinterpret_cast<const float4*>(chain_shift + 16 + 4 * g);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            f32x4 t0 = f32x4{b0.x, b0.y, b0.z, b0.w}, t1 = f32x4{b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float yv = acc[c][rt][r] * sc[rt][r] + sh[rt][r];
                    if (relu) yv = relu1(yv);
                    t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[rt * 4 + r][0], yv, t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[rt * 4 + r][1], yv, t1, 0, 0, 0);
                }
            const int tile = wv * CTW + c, tr = tile / (TW / 16), tc = tile - tr * (TW / 16);
            const int oy = oy0 + tr, ox = ox0 + tc * 16 + j;
            if (oy >= Ho || ox >= Wo) continue;
            float* op = out + (((long long)n * Ho + oy) * Wo + ox) * 32 + 4 * g;
            *reinterpret_cast<float4*>(op) = make_float4(t0[0], t0[1], t0[2], t0[3]);
            *reinterpret_cast<float4*>(op + 16) = make_float4(t1[0], t1[1], t1[2], t1[3]);
        }
        return;
    }
    // ---- epilogue: BN/bias, optional x2 bilinear upsample-add of the coarser FPN map, ReLU ----
    const float sy = ac_scale(Hc, Ho), sx = ac_scale(Wc, Wo);
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int tile = wv * CTW + c, tr = tile / (TW / 16), tc = tile - tr * (TW / 16);
        const int oy = oy0 + tr, ox = ox0 + tc * 16 + j;
        if (oy >= Ho || ox >= Wo) continue;
        const long long o = ((long long)n * Ho + oy) * Wo + ox;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int c0 = rt * 16 + 4 * g;
            if (c0 == cout && rgb_src != nullptr) {
                // texel mode (smooth0 -> render gather source): the first idle lane group appends
                // [rgb*0.5+0.5 | 0] (unpreprocess utils.py:608 + cat network.py:34) behind the features
                const float* sp = rgb_src + (long long)n * 3 * Ho * Wo + (long long)oy * Wo + ox;
                *reinterpret_cast<float4*>(out + o * out_stride + c0) =
                    make_float4(sp[0] * 0.5f + 0.5f, sp[(long long)Ho * Wo] * 0.5f + 0.5f,
                                sp[2LL * Ho * Wo] * 0.5f + 0.5f, 0.f);
            }
            if (c0 >= cout) continue;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = acc[c][rt][r] * scale[c0 + r] + shift[c0 + r];
            if (up != nullptr) {            // feature_net.py:24-25: F.interpolate(x, 2, bilinear, align_corners) + y
                const Lerp1 ly = ac_lerp(oy, sy, Hc), lx = ac_lerp(ox, sx, Wc);
                const float* ub = up + (long long)n * Hc * Wc * cout + c0;
                const float4 u00 = *reinterpret_cast<const float4*>(ub + ((long long)ly.i0 * Wc + lx.i0) * cout);
                const float4 u01 = *reinterpret_cast<const float4*>(ub + ((long long)ly.i0 * Wc + lx.i1) * cout);
                const float4 u10 = *reinterpret_cast<const float4*>(ub + ((long long)ly.i1 * Wc + lx.i0) * cout);
                const float4 u11 = *reinterpret_cast<const float4*>(ub + ((long long)ly.i1 * Wc + lx.i1) * cout);
                y[0] = ac_blend(ly, lx, u00.x, u01.x, u10.x, u11.x) + y[0];
                y[1] = ac_blend(ly, lx, u00.y, u01.y, u10.y, u11.y) + y[1];
                y[2] = ac_blend(ly, lx, u00.z, u01.z, u10.z, u11.z) + y[2];
                y[3] = ac_blend(ly, lx, u00.w, u01.w, u10.w, u11.w) + y[3];
            }
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
            }
            *reinterpret_cast<float4*>(out + o * out_stride + c0) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

template <int CINP, int RT, int K, int STR, int TH, bool NCHW3, bool CHAIN = false>
static void launch_c2(const Conv2dDesc& L, const float* in, float* out, const float* up, int N, int Hi, int Wi, int Hc,
                      int Wc, hipStream_t st) {
    const float* rgb_src = L.rgb_src;
    const int out_stride = L.out_stride > 0 ? L.out_stride : L.cout;
    constexpr int P = (K - 1) / 2, CB = CINP >= 16 ? 16 : CINP;
    const int Ho = (Hi + 2 * P - K) / STR + 1, Wo = (Wi + 2 * P - K) / STR + 1;
    const int tiles_y = cdiv(Ho, TH), tiles_x = cdiv(Wo, 32);
    constexpr int IH = (TH - 1) * STR + K, IW = 31 * STR + K;
    const int nslot = IH * (STR == 2 ? 2 * ((IW + 1) / 2) : IW);                                    // stride 2: de-interleaved rows
    const size_t shmem = (size_t)nslot * CB * sizeof(float);
    const unsigned grid = (unsigned)((long long)N * tiles_y * tiles_x);
    ENERF_LAUNCH((k_conv2d<CINP, RT, K, STR, TH, NCHW3, CHAIN>), grid, 256, shmem, st, L.w, L.scale, L.shift, in, out, up,
                 rgb_src, out_stride, L.cout, L.relu, N, Hi, Wi, Ho, Wo, Hc, Wc, tiles_y, tiles_x, L.chain_w, L.chain_shift);
}

// =====================================================================================================
// conv0.1( conv0.0( image ) )  — feature_net.py:7-9 (two ConvBnReLU 3x3, 3 -> 8 -> 8 at full resolution) in ONE
// kernel.  Unfused, the 8-channel intermediate (31 MB at 3x512x640) is written and read back with halos and the
// first layer is a 23 us launch against a 10 us floor.  Here a block stages the 12x36 image patch of its 8x32
// output tile (NCHW planes -> [pixel][4] texels), evaluates conv0.0 + BN + ReLU on the matrix cores for the
// 10x34 haloed tile (22 column tiles, 1.33x halo recompute of a 1-k-step layer) straight into LDS — zero outside
// the image, which is conv0.1's padding — and runs conv0.1 from there exactly like k_conv2d<8,...>.
// =====================================================================================================
// (rounds 1-2 ran both layers on 16x16x4 MFMAs — k_conv0_fused — and rounds 3-4 on the batched 4x4x1 instruction with the weights
// re-laid-out in LDS per block — k_conv0_fused_b4; both forms left the source in round 6: tools/patches/r06_pruned_conv2d_variants.diff)
//
// The same two layers on the batched 4x4x1 matrix instruction (see conv3d_b4.hip): both have Cout = 8, i.e. half of every
// 16x16x4 tile multiplied zeros (and conv0.0's 3 input channels were padded to a 4-wide k-step).  LANE = PIXEL: a lane
// feeds its own pixel's input value and receives its own pixel's four output channels; two instructions per input channel
// cover the 8 outputs.  Stage 1 walks the 340 haloed pixels in flat order (its texel reads and its two plane stores are
// conflict free that way), stage 2 gives each ds_read_b128 service group 16 consecutive pixels of a row.  The weights are
// re-laid-out from the 16x16x4 operand images the library already packs (no new pack kernel): per (tap, quad, half) one
// broadcast float4 per lane (row i = lane & 3).  198 + 288 16x16x4 MFMAs (15.5 k matrix cycles per block) become
// 324 + 576 4x4x1 MFMAs (8.5 k).
// Round 5: the same two layers with their weights in REGISTERS (A-operand broadcast, common.h mfma4_bc): 54 + 144 weight columns =
// 4 + 9 VGPRs, loaded with 13 coalesced loads from images packed once (k_conv2d_cb_pack).  The per-block re-layout of both
// layers' weights into LDS (four gather loads + stores per thread, in front of the first barrier) and two of the three
// ds_read_b128 per 6-8 MFMAs go away; a block is 17.8 KB of LDS.  Same arithmetic as k_conv0_fused_b4: bit-identical outputs.
__global__ __launch_bounds__(256) void k_conv0_fused_cb(const float* __restrict__ wc0, const float* __restrict__ scale0,
                                                        const float* __restrict__ shift0, const float* __restrict__ wc1,
                                                        const float* __restrict__ scale1, const float* __restrict__ shift1,
                                                        const float* __restrict__ img, float* __restrict__ out, int N, int H,
                                                        int W, int tiles_y, int tiles_x, PrepJob job) {
    constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, NPX = IH * IW;      // conv0.1 input tile (10 x 34)
    constexpr int PH = IH + 2, PW = IW + 2, NPP = PH * PW;                        // image patch (12 x 36)
    ENERF_DYN_SMEM(float, lds);
    float* pat = lds;                   // [NPP] float4 image texels (4th channel 0)
    float* til = pat + NPP * 4;         // [2 quads][NPX] float4: conv0.0 output planes

    const int tid = threadIdx.x, lane = tid & 63;
    const int nconv = N * tiles_y * tiles_x;
    // the blocks beside the convolution's own carry the frame's camera-only preparation (prep_job.h): first kernel of the frame,
    // so level 0's depth planes and every level's projection matrices are ready long before the warp asks for them
    // (in FRONT of them in dispatch order: the one block with the fp64 inverse chain starts at once and ends under conv0's blocks;
    // behind them it was the kernel's tail — 25.4 -> 32.2 us, the whole launch it was meant to save)
    if ((int)blockIdx.x < job.nblocks) { prep_job_block(job, (int)blockIdx.x, tid, 256); return; }
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = (int)xcd_contiguous(blockIdx.x - (unsigned)job.nblocks, (unsigned)nconv);
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;

    float w0r[4], w1r[9];
#pragma unroll
    for (int r = 0; r < 4; ++r) w0r[r] = wc0[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < 9; ++r) w1r[r] = wc1[r * 64 + lane];
    {   // ---- image patch -> LDS: one thread per patch pixel, three coalesced plane reads, zero outside ----
        constexpr int NIT = (NPP + 255) / 256;
        float v0[NIT], v1[NIT], v2[NIT];
        bool sk[NIT];
        const float* base = img + (long long)n * 3 * H * W;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256, ic = i < NPP ? i : NPP - 1;
            const int ly = ic / PW, lx = ic - ly * PW, gy = oy0 - 2 + ly, gx = ox0 - 2 + lx;
            sk[it] = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int off = sk[it] ? gy * W + gx : 0;
            v0[it] = base[off]; v1[it] = base[H * W + off]; v2[it] = base[2 * H * W + off];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i < NPP)
                *reinterpret_cast<float4*>(pat + i * 4) =
                    sk[it] ? make_float4(v0[it], v1[it], v2[it], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    // ---- stage 1: conv0.0 + BN + ReLU for the haloed tile, lane = haloed pixel (flat order) -> LDS planes ----
#pragma unroll 1
    for (int base = wv * 64; base < NPX; base += 256) {                    // wave-uniform
        const int p = base + lane, pc = p < NPX ? p : NPX - 1;
        const int ly = pc / IW, lx = pc - ly * IW;
        const float* pb = pat + (ly * PW + lx) * 4;
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 tex = *reinterpret_cast<const float4*>(pb + ((t / 3) * PW + (t % 3)) * 4);
            const int c = t * 6;                                            // column (t*3 + ch)*2 + half
            acc0 = mfma4_bc(w0r[(c + 0) >> 4], tex.x, acc0, (c + 0) & 15);
            acc1 = mfma4_bc(w0r[(c + 1) >> 4], tex.x, acc1, (c + 1) & 15);
            acc0 = mfma4_bc(w0r[(c + 2) >> 4], tex.y, acc0, (c + 2) & 15);
            acc1 = mfma4_bc(w0r[(c + 3) >> 4], tex.y, acc1, (c + 3) & 15);
            acc0 = mfma4_bc(w0r[(c + 4) >> 4], tex.z, acc0, (c + 4) & 15);
            acc1 = mfma4_bc(w0r[(c + 5) >> 4], tex.z, acc1, (c + 5) & 15);
        }
        const int gy = oy0 - 1 + ly, gx = ox0 - 1 + lx;
        const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
        if (p < NPX) {
            float4 o0, o1;
            o0.x = inside ? relu1(acc0[0] * scale0[0] + shift0[0]) : 0.f; o0.y = inside ? relu1(acc0[1] * scale0[1] + shift0[1]) : 0.f;
            o0.z = inside ? relu1(acc0[2] * scale0[2] + shift0[2]) : 0.f; o0.w = inside ? relu1(acc0[3] * scale0[3] + shift0[3]) : 0.f;
            o1.x = inside ? relu1(acc1[0] * scale0[4] + shift0[4]) : 0.f; o1.y = inside ? relu1(acc1[1] * scale0[5] + shift0[5]) : 0.f;
            o1.z = inside ? relu1(acc1[2] * scale0[6] + shift0[6]) : 0.f; o1.w = inside ? relu1(acc1[3] * scale0[7] + shift0[7]) : 0.f;
            *reinterpret_cast<float4*>(til + p * 4) = o0;
            *reinterpret_cast<float4*>(til + (NPX + p) * 4) = o1;
        }
    }
    __syncthreads();

    // ---- stage 2: conv0.1 (8 -> 8), lane = output pixel; service group k of a wave = 16 consecutive pixels of one row ----
    const int m = lane & 31;
    const bool g1 = (m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28;
    const int pos = g1 ? (m < 12 ? m - 4 : m < 20 ? m - 8 : m - 16) : (m < 4 ? m : m < 16 ? m - 8 : m - 12);
    const int kgrp = 2 * (lane >> 5) + (g1 ? 1 : 0);
    const int row = 2 * wv + (kgrp >> 1), col = (kgrp & 1) * 16 + pos;
    const float* tb = til + (row * IW + col) * 4;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bq[2];
    bq[0] = *reinterpret_cast<const float4*>(tb);
#pragma unroll
    for (int tq = 0; tq < 18; ++tq) {                                       // (tap, quad)
        if (tq + 1 < 18) {
            const int t1 = (tq + 1) >> 1, q1 = (tq + 1) & 1;
            bq[(tq + 1) & 1] = *reinterpret_cast<const float4*>(tb + q1 * NPX * 4 + ((t1 / 3) * IW + (t1 % 3)) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float4 bb = bq[tq & 1];
        const float a = w1r[tq >> 1];                                       // columns tq*8 .. tq*8 + 7
        const int kb = (tq & 1) * 8;
        acc0 = mfma4_bc(a, bb.x, acc0, kb + 0); acc1 = mfma4_bc(a, bb.x, acc1, kb + 1);
        acc0 = mfma4_bc(a, bb.y, acc0, kb + 2); acc1 = mfma4_bc(a, bb.y, acc1, kb + 3);
        acc0 = mfma4_bc(a, bb.z, acc0, kb + 4); acc1 = mfma4_bc(a, bb.z, acc1, kb + 5);
        acc0 = mfma4_bc(a, bb.w, acc0, kb + 6); acc1 = mfma4_bc(a, bb.w, acc1, kb + 7);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: BN + ReLU, channels-last store (cout = 8: 32 contiguous bytes per pixel) ----
    const int oy = oy0 + row, ox = ox0 + col;
    if (oy >= H || ox >= W) return;
    const long long o = ((long long)n * H + oy) * W + ox;
    *reinterpret_cast<float4*>(out + o * 8) =
        make_float4(relu1(acc0[0] * scale1[0] + shift1[0]), relu1(acc0[1] * scale1[1] + shift1[1]),
                    relu1(acc0[2] * scale1[2] + shift1[2]), relu1(acc0[3] * scale1[3] + shift1[3]));
    *reinterpret_cast<float4*>(out + o * 8 + 4) =
        make_float4(relu1(acc1[0] * scale1[4] + shift1[4]), relu1(acc1[1] * scale1[5] + shift1[5]),
                    relu1(acc1[2] * scale1[6] + shift1[6]), relu1(acc1[3] * scale1[7] + shift1[7]));
}

bool launch_conv0_fused(const Conv2dDesc& L0, const Conv2dDesc& L1, const float* w_cb0, const float* w_cb1, const float* img,
                        float* out, int N, int H, int W, hipStream_t st, const PrepJob* job) {
    const int tiles_y = cdiv(H, 8), tiles_x = cdiv(W, 32);
    const size_t shmem = (size_t)(12 * 36 * 4 + 2 * 10 * 34 * 4) * sizeof(float);
    PrepJob none;
    memset(&none, 0, sizeof(none));
    const PrepJob& J = job != nullptr ? *job : none;
    ENERF_LAUNCH(k_conv0_fused_cb, (unsigned)(N * tiles_y * tiles_x + J.nblocks), 256, shmem, st, w_cb0, L0.scale, L0.shift, w_cb1,
                 L1.scale, L1.shift, img, out, N, H, W, tiles_y, tiles_x, J);
    return job != nullptr;
}

// =====================================================================================================
// smooth0( up2(f1pre) + lat0(c0) )  — feature_net.py:32-35 — in ONE kernel.
// The unfused chain writes the 32-channel full-resolution FPN sum (126 MB at 3x512x640) and reads it back
// with halos; here each block rebuilds its haloed 10x34 tile of that sum in LDS from (a) the 8-channel c0
// tile and (b) the half-resolution f1pre patch under it, 16 channels per pass:
//     f0[px][m] = bilinear_ac(f1pre)[px][m] + (b_lat0[m] + sum_ci W_lat0[m][ci] * c0[px][ci]),  0 outside the image
// and then runs the 3x3 32->8 convolution on the matrix cores exactly like k_conv2d (same packed weights,
// same texel-mode epilogue).  HBM traffic per frame drops by ~290 MB and one launch disappears.
// =====================================================================================================
// (rounds 1-2: k_smooth0_fused, the convolution on 16x16x4 MFMAs, plain or tap-packed; left the source in round 6 with its P/Q weight
// image: tools/patches/r06_pruned_conv2d_variants.diff.  enerf_options_t.featnet_smooth0_plain now selects the two unfused launches.)
//
// The same fusion with the 3x3 32 -> 8 convolution on the batched 4x4x1 matrix instruction (conv3d_b4.hip's idea): lane =
// output pixel, two instructions per input channel cover the 8 outputs, so nothing of the 16x16x4 tile's 16 rows is wasted
// (the tap-packed variant above still wastes a third) and the tile is a full 8 x 32.  The FPN-sum tile lives in LDS as
// [channel quad][pixel] float4 planes (the build phase's D layout stores straight into them), every ds_read_b128 service
// group reads 16 consecutive pixels of a row, and the pass's weights are re-laid-out in LDS from the plain 16x16x4 operand
// image (W[o][16cb + 4q + r][t] = w[((8t + 4cb + r)*64 + 16q + o], k_conv2d_pack) as broadcast float4s.
// (k_smooth0_b4 — this form with the pass's weights re-laid-out in LDS — was rounds 3-4's kernel: tools/patches/r06_pruned_conv2d_variants.diff)

// ---- round 5: the same fusion with the weights in REGISTERS (A-operand broadcast, common.h mfma4_bc) -------------------------
// k_smooth0_b4's three co-resident blocks per CU spent half of the CU's LDS cycles (PMC r04: LDS busy 0.51, matrix pipe 0.38)
// and two of every three ds_read_b128 of the tap loop were broadcast reads of weights that a per-pass re-layout had put there.
// With cbsz:4 abid:K ONE VGPR holds sixteen 4x1 weight columns: a 16-channel pass of the 3x3 32 -> 8 layer is 288 columns = 18
// VGPRs, loaded with 18 coalesced 256-byte loads from an image packed once (k_conv2d_cb_pack) — the tap loop is one ds_read_b128
// (the lane's own pixel, 4 channels) per 8 MFMAs.  What else left LDS: the conv0 tile (a lane's six float2 build operands are
// loaded ONCE into registers and serve both passes), the lat0 weights (one float2 + one float4 per lane and pass, from global),
// the re-laid-out conv weights; the f1pre patch arrives by LDS-DMA (no VGPRs; pass 1's copy lands during pass 0's MFMAs).
// 31 KB of LDS per block instead of 47.6: FIVE blocks per CU, so that one block's build phase (VALU + LDS) runs beside four
// others' matrix phases.  Arithmetic (operation order included) is k_smooth0_b4's: the outputs are bit-identical.
// Broadcast-A image of a 3x3 layer with Cout = 8 for `cinp` of its input channels (ci0 .. ci0 + cinp - 1):
// packed[r*64 + l]: column c = 16r + (l >> 2) = (t*cinp + cil)*2 + half  ->  W[4*half + (l & 3)][ci0 + cil][tap t]; 0 beyond 18*cinp
// columns.  ceil(18*cinp / 16) registers: 4 (conv0.0, cinp = 3), 9 (conv0.1, 8), 18 per 16-channel pass of smooth0.
__host__ __device__ __forceinline__ int cb_regs(int cinp) { return (18 * cinp + 15) / 16; }
__global__ __launch_bounds__(256) void k_conv2d_cb_pack(const float* __restrict__ w, int cin, int ci0, int cinp,
                                                        float* __restrict__ packed) {
    const int total = cb_regs(cinp) * 64, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int l = i & 63, c = (i >> 6) * 16 + (l >> 2);
    const int half = c & 1, cc = c >> 1, cil = cc % cinp, t = cc / cinp;
    packed[i] = c < 18 * cinp ? w[((4 * half + (l & 3)) * cin + ci0 + cil) * 9 + t] : 0.f;
}
void launch_conv2d_cb_pack(const float* w, int cin, int ci0, int cinp, float* packed, hipStream_t st) {
    ENERF_LAUNCH_SIMPLE(k_conv2d_cb_pack, (unsigned)cdiv(cb_regs(cinp) * 64, 256), 256, 0, st, w, cin, ci0, cinp, packed);
}

#ifndef ENERF_S0_CB_BLOCKS
#define ENERF_S0_CB_BLOCKS 5         // co-resident blocks per CU the register budget is set for (LDS allows 5)
#endif
__global__ __launch_bounds__(256, ENERF_S0_CB_BLOCKS) void k_smooth0_cb(
    const float* __restrict__ wcb, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ c0,
    const float* __restrict__ f1pre, const float* __restrict__ lat_w, const float* __restrict__ lat_b, float* __restrict__ out,
    const float* __restrict__ rgb_src, int out_stride, int N, int H, int W, int tiles_y, int tiles_x) {
    constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, NPX = IH * IW;             // 10 x 34 halo tile
    constexpr int PH = 7, PW = 20, NPP = PH * PW;                                       // f1pre patch (half res)
    constexpr int NPCH = (NPP * 4 + 63) / 64;                                           // 64-float4 chunks of a pass's patch (9)
    constexpr int NT16 = (NPX + 15) / 16, NBI = (NT16 + 3) / 4;                         // build items (16 px x 16 ch): 22, <= 6 per wave
    ENERF_DYN_SMEM(float, lds);
    float* pat = lds;                       // [NPCH * 64] float4: [patch pixel][16 channels of the pass], DMA destination
    float* til = pat + NPCH * 256;          // [4 quads][NPX] float4 planes (16 channels of the current pass)
    float* tabs = til + 4 * NPX * 4;        // bilinear tables of the x2 upsample (see k_smooth0_b4)

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int H1 = H / 2, W1 = W / 2;
    const float sy = ac_scale(H1, H), sx = ac_scale(W1, W);
    const int py0 = (int)(sy * (float)max(iy0, 0)), px0 = (int)(sx * (float)max(ix0, 0));

    // ---- everything the block reads from memory is requested here, before the first wait ----
    float wr[2][18];                                                  // the passes' weight columns (16 per register)
#pragma unroll
    for (int r = 0; r < 18; ++r) wr[0][r] = wcb[r * 64 + lane];
    float2 a_lat[2];
    float4 bias4[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        a_lat[cb] = *reinterpret_cast<const float2*>(lat_w + (cb * 16 + j) * 8 + 2 * g);    // lat0 A operand: row j, k = channel 2g + r
        bias4[cb] = *reinterpret_cast<const float4*>(lat_b + cb * 16 + 4 * g);
    }
    float2 cvr[NBI];                                                  // conv0 values of this lane's build items (both passes)
#pragma unroll
    for (int it = 0; it < NBI; ++it) {
        const int t = wv + 4 * it;
        cvr[it] = make_float2(0.f, 0.f);
        if (t < NT16) {                                               // wave-uniform
            const int px = t * 16 + j, pxc = px < NPX ? px : NPX - 1;
            const int ly = (pxc * 241) >> 13, lx = pxc - ly * IW, gy = iy0 + ly, gx = ix0 + lx;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const long long off = ok ? ((long long)n * H + gy) * W + gx : 0;
            const float2 v = *reinterpret_cast<const float2*>(c0 + off * 8 + 2 * g);
            cvr[it] = ok ? v : make_float2(0.f, 0.f);
        }
    }
    unsigned psrc[(NPCH + 3) / 4];                                    // this lane's source (float offset, channel 0) per patch chunk
#pragma unroll
    for (int k = 0; k < (NPCH + 3) / 4; ++k) {
        const int i = (wv + 4 * k) * 64 + lane, ic = i < NPP * 4 ? i : NPP * 4 - 1;
        const int pp = ic >> 2, q = ic & 3, pr = pp / PW, pc = pp - pr * PW;
        const int gy = min(py0 + pr, H1 - 1), gx = min(px0 + pc, W1 - 1);
        psrc[k] = (unsigned)((((long long)n * H1 + gy) * W1 + gx) * 32 + q * 4);
    }
    auto issue_pat = [&](int cb) {
#pragma unroll
        for (int k = 0; k < (NPCH + 3) / 4; ++k)
            if (wv + 4 * k < NPCH) glds16(f1pre + psrc[k] + cb * 16, pat + (wv + 4 * k) * 256, lane);
    };
    issue_pat(0);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < IH + IW) {                                              // upsample tables, once per block (see k_smooth0_b4)
        const bool isrow = tid < IH;
        const int k = isrow ? tid : tid - IH, gq = (isrow ? iy0 : ix0) + k, lim = isrow ? H : W;
        const Lerp1 v = ac_lerp(min(max(gq, 0), lim - 1), isrow ? sy : sx, isrow ? H1 : W1);
        const int unit = isrow ? PW * 16 : 16, org = isrow ? py0 : px0;
        const bool ok = gq >= 0 && gq < lim;
        float4 e;
        e.x = __int_as_float(ok ? (v.i0 - org) * unit : -1);
        e.y = __int_as_float(ok ? (v.i1 - org) * unit : -1);
        e.z = v.l0; e.w = v.l1;
        *reinterpret_cast<float4*>(tabs + tid * 4) = e;
    }
    // stage-2 lane -> pixel map: service group k of a wave = 16 consecutive pixels of one tile row
    const int m = lane & 31;
    const bool g1 = (m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28;
    const int pos = g1 ? (m < 12 ? m - 4 : m < 20 ? m - 8 : m - 16) : (m < 4 ? m : m < 16 ? m - 8 : m - 12);
    const int kgrp = 2 * (lane >> 5) + (g1 ? 1 : 0);
    const int row = 2 * wv + (kgrp >> 1), col = (kgrp & 1) * 16 + pos;
    const float* tb = til + (row * IW + col) * 4;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};

    glds_wait_all();
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        // ---- build the FPN-sum tile of this pass (lat0 on the matrix cores + bilinear x2 of the patch) into the quad planes ----
#pragma unroll
        for (int it = 0; it < NBI; ++it) {
            const int t = wv + 4 * it;
            if (t < NT16) {                                           // wave-uniform
                const int px = t * 16 + j, pxc = px < NPX ? px : NPX - 1;
                const int ly = (pxc * 241) >> 13, lx = pxc - ly * IW;
                const float4 rt = *reinterpret_cast<const float4*>(tabs + ly * 4);
                const float4 ct = *reinterpret_cast<const float4*>(tabs + (IH + lx) * 4);
                const int ro0 = __float_as_int(rt.x), ro1 = __float_as_int(rt.y), co0 = __float_as_int(ct.x), co1 = __float_as_int(ct.y);
                const bool inside = (ro0 | co0) >= 0;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_lat[cb].x, cvr[it].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_lat[cb].y, cvr[it].y, acc, 0, 0, 0);
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inside) {
                    Lerp1 vy, vx;
                    vy.l0 = rt.z; vy.l1 = rt.w; vx.l0 = ct.z; vx.l1 = ct.w; vy.i0 = vy.i1 = vx.i0 = vx.i1 = 0;
                    const float* pb = pat + g * 4;
                    const float4 u00 = *reinterpret_cast<const float4*>(pb + ro0 + co0), u01 = *reinterpret_cast<const float4*>(pb + ro0 + co1);
                    const float4 u10 = *reinterpret_cast<const float4*>(pb + ro1 + co0), u11 = *reinterpret_cast<const float4*>(pb + ro1 + co1);
                    o.x = ac_blend(vy, vx, u00.x, u01.x, u10.x, u11.x) + (acc[0] + bias4[cb].x);
                    o.y = ac_blend(vy, vx, u00.y, u01.y, u10.y, u11.y) + (acc[1] + bias4[cb].y);
                    o.z = ac_blend(vy, vx, u00.z, u01.z, u10.z, u11.z) + (acc[2] + bias4[cb].z);
                    o.w = ac_blend(vy, vx, u00.w, u01.w, u10.w, u11.w) + (acc[3] + bias4[cb].w);
                }
                if (px < NPX) *reinterpret_cast<float4*>(til + (g * NPX + px) * 4) = o;
            }
        }
        __syncthreads();                                              // tile complete; nobody reads the patch any more
        if (cb == 0) {                                                // pass 1's patch + weights travel during pass 0's MFMAs
            issue_pat(1);
#pragma unroll
            for (int r = 0; r < 18; ++r) wr[1][r] = wcb[(18 + r) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- 3x3 conv over the pass's 16 channels: per (tap, quad) ONE ds_read_b128 and 8 broadcast-A MFMAs ----
        float4 bq[2];
        bq[0] = *reinterpret_cast<const float4*>(tb);
#pragma unroll
        for (int tq = 0; tq < 36; ++tq) {
            if (tq + 1 < 36) {
                const int t1 = (tq + 1) >> 2, q1 = (tq + 1) & 3;
                bq[(tq + 1) & 1] = *reinterpret_cast<const float4*>(tb + q1 * NPX * 4 + ((t1 / 3) * IW + (t1 % 3)) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float4 bb = bq[tq & 1];
            const float a = wr[cb][tq >> 1];
            const int kb = (tq & 1) * 8;
            acc0 = mfma4_bc(a, bb.x, acc0, kb + 0); acc1 = mfma4_bc(a, bb.x, acc1, kb + 1);
            acc0 = mfma4_bc(a, bb.y, acc0, kb + 2); acc1 = mfma4_bc(a, bb.y, acc1, kb + 3);
            acc0 = mfma4_bc(a, bb.z, acc0, kb + 4); acc1 = mfma4_bc(a, bb.z, acc1, kb + 5);
            acc0 = mfma4_bc(a, bb.w, acc0, kb + 6); acc1 = mfma4_bc(a, bb.w, acc1, kb + 7);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cb == 0) {
            glds_wait_all();                                          // this wave's share of pass 1's patch has landed
            __syncthreads();                                          // everyone's has; everyone is done reading the tile
        }
    }
    // ---- epilogue: bias, channels-last / texel store (cout = 8): a pixel's 8 features (+ rgb texel) from its own lane ----
    const int oy = oy0 + row, ox = ox0 + col;
    if (oy >= H || ox >= W) return;
    const long long o = ((long long)n * H + oy) * W + ox;
    *reinterpret_cast<float4*>(out + o * out_stride) =
        make_float4(acc0[0] * scale[0] + shift[0], acc0[1] * scale[1] + shift[1], acc0[2] * scale[2] + shift[2], acc0[3] * scale[3] + shift[3]);
    *reinterpret_cast<float4*>(out + o * out_stride + 4) =
        make_float4(acc1[0] * scale[4] + shift[4], acc1[1] * scale[5] + shift[5], acc1[2] * scale[6] + shift[6], acc1[3] * scale[7] + shift[7]);
    if (rgb_src != nullptr) {
        const float* sp = rgb_src + (long long)n * 3 * H * W + (long long)oy * W + ox;
        *reinterpret_cast<float4*>(out + o * out_stride + 8) =
            make_float4(sp[0] * 0.5f + 0.5f, sp[(long long)H * W] * 0.5f + 0.5f, sp[2LL * H * W] * 0.5f + 0.5f, 0.f);
    }
}


// =====================================================================================================================
// smooth1( up2(f2) + lat1(c1) )  — feature_net.py:33-34 — in ONE kernel (round 5, VERDICT r04 #1a).
// Unfused, the lateral kernel writes the 32-channel half-resolution FPN sum f1pre (31.5 MB at dtu, 134 MB at zju) and smooth1
// reads it back with halos; on zju the top-down half IS the frame's critical path (FeatureNet 0.88 of 1.85 ms).  Here a block
// rebuilds its haloed 10 x 34 tile of f1pre in LDS — lat1 (1x1, 16 -> 32) on the matrix cores from the c1 values the lanes hold
// in registers, plus the align-corners x2 blend of the f2 patch that arrives by LDS-DMA — 16 channels per pass, runs the 3x3
// 32 -> 16 convolution from there on 16x16x4 MFMAs, and stores the tile's INTERIOR of f1pre on the way (the fused smooth0
// kernel consumes it).  One launch and the halo'd read-back of f1pre go away.  Structure = k_smooth0_cb's.
// lat_w / sm_w: the layers' ordinary operand images (k_conv2d_pack): lat1 [ks 0..3][rt 0..1][64], smooth1 [tap][ks 0..7][64].
// =====================================================================================================================
// (k_smooth0_cbp — k_smooth0_cb as a persistent, tile-pipelined kernel — measured NOT faster alone (round 5, profiles/r05_smooth0_ablation.txt) and, with a
// capped grid as a side-lane kernel, slower in the frame (round 6, profiles/r06_ab_smooth0_persistent_capped.txt): tools/patches/r06_pruned_conv2d_variants.diff)
#ifndef ENERF_S1F_BLOCKS
#define ENERF_S1F_BLOCKS 4
#endif
__global__ __launch_bounds__(256, ENERF_S1F_BLOCKS) void k_smooth1_fused(const float* __restrict__ sm_w, const float* __restrict__ sm_scale,
                                                          const float* __restrict__ sm_shift, const float* __restrict__ lat_w,
                                                          const float* __restrict__ lat_shift, const float* __restrict__ c1,
                                                          const float* __restrict__ f2, float* __restrict__ f1pre,
                                                          float* __restrict__ out, int N, int H, int W, int tiles_y, int tiles_x) {
    constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, NPX = IH * IW;             // 10 x 34 halo tile (half resolution)
    constexpr int PH = 7, PW = 20, NPP = PH * PW;                                       // f2 patch (quarter resolution)
    constexpr int NPCH = (NPP * 4 + 63) / 64;
    constexpr int NT16 = (NPX + 15) / 16, NBI = (NT16 + 3) / 4;                         // build items (16 px x 16 ch): 22, <= 6 per wave
    constexpr int CTW = 4;
    ENERF_DYN_SMEM(float, lds);
    float* pat = lds;                       // [NPCH * 64] float4: [patch pixel][16 channels of the pass], DMA destination
    float* til = pat + NPCH * 256;          // [4 quads][NPX] float4 planes (16 channels of the current pass)
    float* tabs = til + 4 * NPX * 4;        // bilinear tables of the x2 upsample (see k_smooth0_b4)

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int H2 = H / 2, W2 = W / 2;                                                   // (H, W: the half-resolution extent)
    const float sy = ac_scale(H2, H), sx = ac_scale(W2, W);
    const int py0 = (int)(sy * (float)max(iy0, 0)), px0 = (int)(sx * (float)max(ix0, 0));

    // ---- requested before the first wait: pass 0's operands (lat1 + smooth1 weights), the lanes' c1 values, the f2 patch ----
    // One register set for the per-pass operands (41 loads); pass 1's are requested behind pass 0's last MFMA and land during
    // pass 1's build.  (A second set — and the ring / index registers given up below — cost the kernel its FOURTH wave per SIMD:
    // dtu's 960 tiles are then co-resident in one round, 4 x 256 slots, instead of 768 + a quarter-filled second round.)
    float a_lat[4];                                                   // lat1 A operands of the pass (row tile = pass): [k-step]
    float4 bias4;
    float aq[36];                                                     // smooth1 A operands of the pass: [tap][r]
    auto load_pass_operands = [&](int cb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a_lat[ks] = lat_w[(ks * 2 + cb) * 64 + lane];
        bias4 = *reinterpret_cast<const float4*>(lat_shift + cb * 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) aq[t * 4 + r] = sm_w[(t * 8 + cb * 4 + r) * 64 + lane];
    };
    load_pass_operands(0);
    f32x4 cvr[NBI];                                                   // c1 channels 4g..4g+3 of this lane's build pixels (both passes)
#pragma unroll
    for (int it = 0; it < NBI; ++it) {
        const int t = wv + 4 * it;
        cvr[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < NT16) {                                               // wave-uniform
            const int px = t * 16 + j, pxc = px < NPX ? px : NPX - 1;
            const int ly = (pxc * 241) >> 13, lx = pxc - ly * IW, gy = iy0 + ly, gx = ix0 + lx;
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const long long off = ok ? ((long long)n * H + gy) * W + gx : 0;
            const f32x4 v = *reinterpret_cast<const f32x4*>(c1 + off * 16 + 4 * g);
            if (ok) cvr[it] = v;
        }
    }
    unsigned psrc[(NPCH + 3) / 4];
#pragma unroll
    for (int k = 0; k < (NPCH + 3) / 4; ++k) {
        const int i = (wv + 4 * k) * 64 + lane, ic = i < NPP * 4 ? i : NPP * 4 - 1;
        const int pp = ic >> 2, q = ic & 3, pr = pp / PW, pc = pp - pr * PW;
        const int gy = min(py0 + pr, H2 - 1), gx = min(px0 + pc, W2 - 1);
        psrc[k] = (unsigned)((((long long)n * H2 + gy) * W2 + gx) * 32 + q * 4);
    }
    auto issue_pat = [&](int cb) {
#pragma unroll
        for (int k = 0; k < (NPCH + 3) / 4; ++k)
            if (wv + 4 * k < NPCH) glds16(f2 + psrc[k] + cb * 16, pat + (wv + 4 * k) * 256, lane);
    };
    issue_pat(0);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < IH + IW) {                                              // upsample tables, once per block (see k_smooth0_b4)
        const bool isrow = tid < IH;
        const int k = isrow ? tid : tid - IH, gq = (isrow ? iy0 : ix0) + k, lim = isrow ? H : W;
        const Lerp1 v = ac_lerp(min(max(gq, 0), lim - 1), isrow ? sy : sx, isrow ? H2 : W2);
        const int unit = isrow ? PW * 16 : 16, org = isrow ? py0 : px0;
        const bool ok = gq >= 0 && gq < lim;
        float4 e;
        e.x = __int_as_float(ok ? (v.i0 - org) * unit : -1);
        e.y = __int_as_float(ok ? (v.i1 - org) * unit : -1);
        e.z = v.l0; e.w = v.l1;
        *reinterpret_cast<float4*>(tabs + tid * 4) = e;
    }
    f32x4 acc[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    glds_wait_all();
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        // ---- build f1pre's tile for this pass: lat1 on the matrix cores + bilinear x2 of the f2 patch; interior -> global ----
#pragma unroll
        for (int it = 0; it < NBI; ++it) {
            const int t = wv + 4 * it;
            if (t < NT16) {                                           // wave-uniform
                const int px = t * 16 + j, pxc = px < NPX ? px : NPX - 1;
                const int ly = (pxc * 241) >> 13, lx = pxc - ly * IW;
                const float4 rt = *reinterpret_cast<const float4*>(tabs + ly * 4);
                const float4 ct = *reinterpret_cast<const float4*>(tabs + (IH + lx) * 4);
                const int ro0 = __float_as_int(rt.x), ro1 = __float_as_int(rt.y), co0 = __float_as_int(ct.x), co1 = __float_as_int(ct.y);
                const bool inside = (ro0 | co0) >= 0;
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x4f32(a_lat[ks], cvr[it][ks], a, 0, 0, 0);
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inside) {
                    Lerp1 vy, vx;
                    vy.l0 = rt.z; vy.l1 = rt.w; vx.l0 = ct.z; vx.l1 = ct.w; vy.i0 = vy.i1 = vx.i0 = vx.i1 = 0;
                    const float* pb = pat + g * 4;
                    const float4 u00 = *reinterpret_cast<const float4*>(pb + ro0 + co0), u01 = *reinterpret_cast<const float4*>(pb + ro0 + co1);
                    const float4 u10 = *reinterpret_cast<const float4*>(pb + ro1 + co0), u11 = *reinterpret_cast<const float4*>(pb + ro1 + co1);
                    // (the unfused kernel's order: lat1's epilogue computes y = acc * 1 + bias, then blend + y: conv2d.hip k_conv2d)
                    o.x = ac_blend(vy, vx, u00.x, u01.x, u10.x, u11.x) + (a[0] * 1.f + bias4.x);
                    o.y = ac_blend(vy, vx, u00.y, u01.y, u10.y, u11.y) + (a[1] * 1.f + bias4.y);
                    o.z = ac_blend(vy, vx, u00.z, u01.z, u10.z, u11.z) + (a[2] * 1.f + bias4.z);
                    o.w = ac_blend(vy, vx, u00.w, u01.w, u10.w, u11.w) + (a[3] * 1.f + bias4.w);
                }
                if (px < NPX) *reinterpret_cast<float4*>(til + (g * NPX + px) * 4) = o;
                if (inside && px < NPX && ly >= 1 && ly <= TH && lx >= 1 && lx <= TW)      // the tile's interior -> f1pre (smooth0 reads it)
                    *reinterpret_cast<float4*>(f1pre + (((long long)n * H + iy0 + ly) * W + ix0 + lx) * 32 + cb * 16 + 4 * g) = o;
            }
        }
        __syncthreads();                                              // tile complete; nobody reads the patch any more
        if (cb == 0) {                                                // pass 1's patch travels during pass 0's MFMAs
            issue_pat(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- 3x3 conv over the pass's 16 channels on 16x16x4 MFMAs (no operand ring: four waves per SIMD cover the LDS latency) ----
        auto read_b = [&](int tap, f32x4 (&bv)[CTW]) {
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int ct = wv * CTW + c, tr = ct >> 1, tc = ct & 1;
                bv[c] = *reinterpret_cast<const f32x4*>(til + (g * NPX + (tr + kh) * IW + tc * 16 + j + kw) * 4);
            }
        };
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            f32x4 bq[CTW];
            read_b(tap, bq);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < CTW; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[tap * 4 + r], bq[c][r], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cb == 0) {
            // pass 1's operands into the SAME registers: requested behind pass 0's last MFMA, they land during pass 1's build phase
            load_pass_operands(1);
            __builtin_amdgcn_sched_barrier(0);
            vmem_wait_pending<41>();                                  // the patch copy (issued before these 41 loads) has landed
            block_barrier_raw();                                      // bare s_barrier: the 36 loads stay in flight across it
        }
    }
    // ---- epilogue: bias (scale = 1), channels-last store: lane (g, j) owns channels 4g..4g+3 of its pixel ----
    const float4 sc = *reinterpret_cast<const float4*>(sm_scale + 4 * g), sh = *reinterpret_cast<const float4*>(sm_shift + 4 * g);
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int ct = wv * CTW + c, tr = ct >> 1, tc = ct & 1;
        const int oy = oy0 + tr, ox = ox0 + tc * 16 + j;
        if (oy >= H || ox >= W) continue;
        const long long o = ((long long)n * H + oy) * W + ox;
        *reinterpret_cast<float4*>(out + o * 16 + 4 * g) =
            make_float4(acc[c][0] * sc.x + sh.x, acc[c][1] * sc.y + sh.y, acc[c][2] * sc.z + sh.z, acc[c][3] * sc.w + sh.w);
    }
}

#ifndef ENERF_SMOOTH1_FUSED
#define ENERF_SMOOTH1_FUSED 1        // 1: k_smooth1_fused (lat1 + up2 + smooth1 in one launch); 0: two k_conv2d launches
#endif
bool launch_smooth1_fused(const Conv2dDesc& Llat, const Conv2dDesc& Lsm, const float* c1, const float* f2, float* f1pre, float* out,
                          int N, int H1, int W1, hipStream_t st) {
    if (!ENERF_SMOOTH1_FUSED || Llat.cin != 16 || Llat.cout != 32 || Lsm.cin != 32 || Lsm.cout != 16 || (H1 & 1) || (W1 & 1)) return false;
    const int tiles_y = cdiv(H1, 8), tiles_x = cdiv(W1, 32);
    const size_t shmem = (size_t)(9 * 256 + 4 * 340 * 4 + 44 * 4) * sizeof(float);
    ENERF_LAUNCH(k_smooth1_fused, (unsigned)(N * tiles_y * tiles_x), 256, shmem, st, Lsm.w, Lsm.scale, Lsm.shift, Llat.w, Llat.shift, c1,
                 f2, f1pre, out, N, H1, W1, tiles_y, tiles_x);
    return true;
}

void launch_smooth0_fused(const Conv2dDesc& L, const float* c0, const float* f1pre, const float* lat_w, const float* lat_b,
                          const float* w_cb, float* out, int N, int H, int W, hipStream_t st) {
    const int out_stride = L.out_stride > 0 ? L.out_stride : 8;
    const int tiles_y = cdiv(H, 8), tiles_x = cdiv(W, 32);
    const size_t shmem = (size_t)(9 * 256 + 4 * 340 * 4 + 44 * 4) * sizeof(float);
    ENERF_LAUNCH(k_smooth0_cb, (unsigned)(N * tiles_y * tiles_x), 256, shmem, st, w_cb, L.scale, L.shift, c0, f1pre, lat_w,
                 lat_b, out, L.rgb_src, out_stride, N, H, W, tiles_y, tiles_x);
}

// The eleven FeatureNet layers use exactly these shapes (feature_net.py:7-22).
int launch_conv2d(const Conv2dDesc& L, const float* in, float* out, const float* up, int N, int Hi, int Wi, int Hc,
                  int Wc, hipStream_t st) {
    const int key = L.cin * 10000 + L.cout * 100 + L.k * 10 + L.stride;
    // 4-row tiles for the 3x3 layers whose 8-row tiling gives fewer than ~4 blocks per CU (quarter/half-resolution maps;
    // measured: smooth1 33.5 -> 30.0 us, conv2.1 18.0 -> 17.1, conv1.1 17.9 -> 17.2).
    const bool th4 = (long long)N * cdiv(Hi, 8) * cdiv(Wi, 32) < 1024;
    // (2-row tiles for conv2.0 / conv2.1 at dtu — 480 four-row tiles for 256 CUs — measured SLOWER: 24.5 -> 29.1 / 16.7 -> 19.6 us,
    // profiles/r05_ab_conv2d_planar.txt)
    switch (key) {
        case 3 * 10000 + 8 * 100 + 31: launch_c2<4, 1, 3, 1, 8, true>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0;     // conv0.0
        case 8 * 10000 + 8 * 100 + 31: launch_c2<8, 1, 3, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0;    // conv0.1
        case 8 * 10000 + 16 * 100 + 52: launch_c2<8, 1, 5, 2, 4, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0;   // conv1.0
        case 16 * 10000 + 16 * 100 + 31:                                                                                   // conv1.1
            if (th4) launch_c2<16, 1, 3, 1, 4, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st);
            else launch_c2<16, 1, 3, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st);
            return 0;
        case 16 * 10000 + 32 * 100 + 52:                                                                                   // conv2.0
            launch_c2<16, 2, 5, 2, 4, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st);
            return 0;
        case 32 * 10000 + 32 * 100 + 31:                                                                                   // conv2.1 (+ toplayer)
            if (L.chain_w != nullptr && th4) launch_c2<32, 2, 3, 1, 4, false, true>(L, in, out, up, N, Hi, Wi, Hc, Wc, st);
            else if (L.chain_w != nullptr) launch_c2<32, 2, 3, 1, 8, false, true>(L, in, out, up, N, Hi, Wi, Hc, Wc, st);
            else launch_c2<32, 2, 3, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st);
            return 0;
        case 32 * 10000 + 32 * 100 + 11: launch_c2<32, 2, 1, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0; // toplayer
        case 16 * 10000 + 32 * 100 + 11: launch_c2<16, 2, 1, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0; // lat1 (4-row tiles: 20.1 -> 21.5 us)
        case 8 * 10000 + 32 * 100 + 11: launch_c2<8, 2, 1, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0;   // lat0
        case 32 * 10000 + 16 * 100 + 31:                                                                                   // smooth1
            if (th4) launch_c2<32, 1, 3, 1, 4, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st);
            else launch_c2<32, 1, 3, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st);
            return 0;
        case 32 * 10000 + 8 * 100 + 31: launch_c2<32, 1, 3, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0;  // smooth0
        // input gradients of the training path (enerf_conv2d_layer): dgrad(stride-1 conv cin -> cout) is the stride-1 conv
        // cout -> cin on the flipped, channel-transposed weights; the shapes above cover conv0.1 / conv1.1 / conv2.1 / toplayer
        case 16 * 10000 + 32 * 100 + 31: launch_c2<16, 2, 3, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0; // d smooth1
        case 8 * 10000 + 32 * 100 + 31: launch_c2<8, 2, 3, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0;   // d smooth0
        case 32 * 10000 + 16 * 100 + 11: launch_c2<32, 1, 1, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0; // d lat1
        case 32 * 10000 + 8 * 100 + 11: launch_c2<32, 1, 1, 1, 8, false>(L, in, out, up, N, Hi, Wi, Hc, Wc, st); return 0;  // d lat0
        default: return -1;
    }
}

}  // namespace enerf
