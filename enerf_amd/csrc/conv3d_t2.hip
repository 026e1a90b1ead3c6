// conv3d_t2.hip — ConvTranspose3d(k3, s2, p1, op1) 16 -> 8 channels (+BN, skip add) with the input box in LDS.
//
// conv11 of both cost-regularisation nets (the last up-convolution: 655,360 outputs at level 1) ran on the
// global-load kernel: 30 us against a 10 us MFMA / 8 us HBM floor, each of the 8 output-parity classes a separate
// wave re-gathering the same inputs and writing half of every 64-B output line.  Here a block owns a 2 x 4 x 16 box
// of input ("q") positions = a 4 x 8 x 32 box of outputs: the 3 x 5 x 17 haloed input box (16 KB) and the weight
// image (27.6 KB) are staged once,
// a wave evaluates ALL eight parity classes of its two q-tiles — the 8 neighbour offsets (0/1 per axis) are read
// from LDS once each and feed the 27 (class, tap) products — and writes both x-parities of a voxel pair, i.e. whole
// 64-B lines.  Same packed weight image (class-major tap order) and epilogue conventions as conv3d.hip.
#include "kernels.h"

namespace enerf {

__host__ __device__ constexpr int t2_ntaps(int cls) { return (1 + ((cls >> 2) & 1)) * (1 + ((cls >> 1) & 1)) * (1 + (cls & 1)); }
__host__ __device__ constexpr int t2_tap0(int cls) {
    int o = 0;
    for (int c = 0; c < cls; ++c) o += t2_ntaps(c);
    return o;
}
// tap index inside class `cls` of the tap that reads the neighbour offset bit `a` (0/1) along an axis of parity p:
// parity 0 has one tap (offset 0); parity 1 has idx 0 -> offset 1, idx 1 -> offset 0 (conv3d.hip convt_axis)
__host__ __device__ constexpr int t2_axis_idx(int p, int a) { return p == 0 ? 0 : (a == 1 ? 0 : 1); }

template <int CIN>
__global__ __launch_bounds__(256, 3) void k_conv3d_t2_lds(const float* __restrict__ wpk, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ in,
                                                          const float* __restrict__ residual, float* __restrict__ out,
                                                          int cout, int relu, int B, int Di, int Hi, int Wi, int nbd, int nbh,
                                                          int nbw) {
    static_assert(CIN == 16, "only the 16 -> 8 up-convolution is routed here");
    constexpr int QD = 2, QH = 4, QW = 16;                        // q-box
    constexpr int HX = QW + 1, HY = QH + 1, HZ = QD + 1, NVOX = HZ * HY * HX;
    constexpr int CB = 16, KS = CIN / 4, QV = CB / 4;
    constexpr int NIT = (NVOX * QV + 255) / 256;
    constexpr int CTW = QD * QH / 4;                              // q-tiles per wave (2)
    ENERF_DYN_SMEM(float, lds);

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    int t = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh; t /= nbh;
    const int bd = t % nbd;
    const int b = t / nbd;
    const int x0 = bw * QW, y0 = bh * QH, z0 = bd * QD;
    const float* inb = in + (long long)b * Di * Hi * Wi * CIN;
    // the 27-tap weight image (27.6 KB) is staged into LDS with the box: read straight from global each (class, tap)
    // group waited a full L2 round trip in front of its MFMAs (27 exposed latencies per block)
    float* wlds = lds + NVOX * CB;
    constexpr int NW4 = 27 * KS * 16;                               // float4s of the weight image
    for (int i = threadIdx.x; i < NW4; i += 256)
        *reinterpret_cast<float4*>(wlds + i * 4) = *reinterpret_cast<const float4*>(wpk + i * 4);
    const float* wl = wlds + lane;

    {   // stage the input box (+1 halo on the high side); positions past the volume are the op1/p1 zero border
        float4 sv[NIT];
        bool sk[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * 256;
            const int ic = i < NVOX * QV ? i : NVOX * QV - 1;
            const int v = ic / QV, q = ic - v * QV;
            const int dx = v % HX, dy = (v / HX) % HY, dz = v / (HX * HY);
            const int gx = x0 + dx, gy = y0 + dy, gz = z0 + dz;
            sk[it] = gx < Wi && gy < Hi && gz < Di;
            const long long off = sk[it] ? (((long long)gz * Hi + gy) * Wi + gx) : 0;
            sv[it] = *reinterpret_cast<const float4*>(inb + off * CIN + q * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * 256;
            if (i < NVOX * QV)
                *reinterpret_cast<float4*>(lds + i * 4) = sk[it] ? sv[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    f32x4 acc[8][CTW];
#pragma unroll
    for (int cls = 0; cls < 8; ++cls)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[cls][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* lbase[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int tile = wv * CTW + c, td = tile / QH, th = tile - td * QH;
        lbase[c] = lds + ((td * HY + th) * HX + j) * CB + g * 4;
    }
    // neighbour offset a = (ad, ah, aw) in {0,1}^3; it feeds class (pd,ph,pw) iff a_axis <= p_axis on every axis
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int ad = (a >> 2) & 1, ah = (a >> 1) & 1, aw = a & 1;
        float bv[CTW][4];
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const float4 tq = *reinterpret_cast<const float4*>(lbase[c] + ((ad * HY + ah) * HX + aw) * CB);
            bv[c][0] = tq.x; bv[c][1] = tq.y; bv[c][2] = tq.z; bv[c][3] = tq.w;
        }
#pragma unroll
        for (int cls = 0; cls < 8; ++cls) {
            const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
            if (ad > pd || ah > ph || aw > pw) continue;           // compile-time
            const int nth = 1 + ph, ntw = 1 + pw;
            const int tap = t2_tap0(cls) + (t2_axis_idx(pd, ad) * nth + t2_axis_idx(ph, ah)) * ntw + t2_axis_idx(pw, aw);
            float aq[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) aq[ks] = wl[(tap * KS + ks) * 64];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int c = 0; c < CTW; ++c)
                    acc[cls][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[ks], bv[c][ks], acc[cls][c], 0, 0, 0);
        }
    }

    // ---- epilogue: BN scale/shift, skip add, (ReLU), float4 stores; lane group g < cout/4 owns channels 4g..4g+3 ----
    const int c0 = 4 * g;
    if (c0 >= cout) return;
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[r] = scale[c0 + r]; sh[r] = shift[c0 + r]; }
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int tile = wv * CTW + c, td = tile / QH, th = tile - td * QH;
        const int qz = z0 + td, qy = y0 + th, qx = x0 + j;
        if (qz >= Di || qy >= Hi || qx >= Wi) continue;
#pragma unroll
        for (int cls = 0; cls < 8; ++cls) {
            const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
            const long long o = (((long long)b * Do + 2 * qz + pd) * Ho + 2 * qy + ph) * Wo + 2 * qx + pw;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = acc[cls][c][r] * sc[r] + sh[r];
            if (residual != nullptr) {
                const float4 rr = *reinterpret_cast<const float4*>(residual + o * cout + c0);
                y[0] += rr.x; y[1] += rr.y; y[2] += rr.z; y[3] += rr.w;
            }
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = relu1(y[r]);
            }
            *reinterpret_cast<float4*>(out + o * cout + c0) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}


// =====================================================================================================================
// k_conv3d_t2_all<CIN, COUT, QD, QH> — the same "every parity class from one wave" structure, generalised and tightened:
//   * COUT = 8 (conv11, 16 -> 8): the 16 MFMA rows carry TWO classes that differ in the x parity — rows 0-7 = class
//     (pd, ph, 0), rows 8-15 = class (pd, ph, 1) — for the same neighbour offset, so a k-step issues 18 instead of 27
//     MFMAs (the x-parity-0 rows of an offset with aw = 1 are zero), every lane owns output (lane groups 0,1 = the even
//     voxel's channels 0-3 / 4-7, groups 2,3 = the odd voxel's), and a store instruction writes the 64 contiguous bytes
//     of a voxel PAIR per q position (the unpaired kernel above used 32 of 64 lanes in its epilogue and wrote 16-byte
//     pieces 64 bytes apart: the epilogue, not the MFMAs, was its time).  The paired A operands are assembled from the
//     ordinary packed image while it is staged into LDS.
//   * COUT = 16 (conv9, 32 -> 16): one class per MFMA (all 16 rows used), the whole 27-tap image (55 KB) in LDS.
// A block owns a QD x QH x 16 box of q positions; wave w evaluates q-tiles w*CTW .. w*CTW+CTW-1.
// =====================================================================================================================
__host__ __device__ constexpr bool t2_pair_valid(int a, int pp) {            // pp = pd*2 + ph; offset a = (ad, ah, aw)
    return ((a >> 2) & 1) <= ((pp >> 1) & 1) && ((a >> 1) & 1) <= (pp & 1);
}
__host__ __device__ constexpr int t2_pair_group(int a, int pp) {              // a-major enumeration of the 18 valid (a, pp)
    int n = 0;
    for (int a2 = 0; a2 < 8; ++a2)
        for (int p2 = 0; p2 < 4; ++p2) {
            if (a2 == a && p2 == pp) return n;
            if (t2_pair_valid(a2, p2)) ++n;
        }
    return n;
}
// tap (index into the class-major packed image) of class cls reading neighbour offset a; -1 if the class does not use it
__host__ __device__ constexpr int t2_tap_of(int cls, int a) {
    const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1, ad = (a >> 2) & 1, ah = (a >> 1) & 1, aw = a & 1;
    if (ad > pd || ah > ph || aw > pw) return -1;
    return t2_tap0(cls) + (t2_axis_idx(pd, ad) * (1 + ph) + t2_axis_idx(ph, ah)) * (1 + pw) + t2_axis_idx(pw, aw);
}

template <int CIN, int COUT, int QD, int QH>
__global__ __launch_bounds__(256) void k_conv3d_t2_all(const float* __restrict__ wimg, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ in,
                                                       const float* __restrict__ residual, float* __restrict__ out, int relu,
                                                       int B, int Di, int Hi, int Wi, int nbd, int nbh, int nbw, int out_planar) {
    static_assert((CIN == 16 && COUT == 8) || (CIN == 32 && COUT == 16), "conv11 (16 -> 8) and conv9 (32 -> 16)");
    constexpr bool PAIR = COUT == 8;
    constexpr int QW = 16, HX = QW + 1, HY = QH + 1, HZ = QD + 1, NVOX = HZ * HY * HX;
    constexpr int NCB = CIN / 16, KS = CIN / 4;
    constexpr int CTW = QD * QH / 4;                              // q-tiles per wave
    constexpr int NGRP = PAIR ? 18 : 27, NACC = PAIR ? 4 : 8;
    static_assert(QD * QH % 4 == 0, "whole q-tiles per wave");
    ENERF_DYN_SMEM(float, lds);
    float* wlds = lds + NCB * NVOX * 16;                          // [group][ks][64]

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    int t = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh; t /= nbh;
    const int bd = t % nbd;
    const int b = t / nbd;
    const int x0 = bw * QW, y0 = bh * QH, z0 = bd * QD;
    const float* inb = in + (long long)b * Di * Hi * Wi * CIN;

    // ---- the skip connection is requested FIRST (it depends on nothing computed here), so its latency hides behind the
    // staging and the MFMAs instead of sitting between them and the stores (round 2: load -> wait -> store per class) ----
    const int c0 = PAIR ? 4 * (g & 1) : 4 * g, lpw = g >> 1;
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    unsigned oo[NACC][CTW];                                        // output offsets (launcher: tensors < 2^32 floats)
    bool live[CTW];
    float4 rres[NACC][CTW];
    // out_planar (PAIR only): the output as two channel-quad planes (B, 2, Do, Ho, Wo, 4) for the asynchronously staged heads
    // kernel; the skip input stays channels-last.  plane offset of this lane's quad: (g & 1) * Do*Ho*Wo*4
    const unsigned nvo = (unsigned)(Do * Ho * Wo);
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int tile = wv * CTW + c, td = tile / QH, th = tile - td * QH;
        const int qz = z0 + td, qy = y0 + th, qx = x0 + j;
        live[c] = qz < Di && qy < Hi && qx < Wi;
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            const int pd = PAIR ? (k >> 1) & 1 : (k >> 2) & 1, ph = PAIR ? k & 1 : (k >> 1) & 1, pw = PAIR ? lpw : k & 1;
            oo[k][c] = live[c] ? (unsigned)((((b * Do + 2 * qz + pd) * Ho + 2 * qy + ph) * Wo + 2 * qx + pw) * COUT + c0) : 0u;
            rres[k][c] = residual != nullptr ? *reinterpret_cast<const float4*>(residual + oo[k][c]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    {   // ---- stage the haloed input box [cb][voxel][16] and the weight image; all loads first, then the LDS stores ----
        constexpr int NIN = NCB * NVOX * 4, NIT = (NIN + 255) / 256;            // float4s of the box
        constexpr int NW4 = NGRP * KS * 16, NWIT = (NW4 + 255) / 256;            // float4s of the weight image
        float4 sv[NIT], wq[NWIT];
        bool sk[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * 256, ic = i < NIN ? i : NIN - 1;
            const int q = ic & 3, vv = ic >> 2, cb = vv / NVOX, v = vv - cb * NVOX;
            const int dx = v % HX, dy = (v / HX) % HY, dz = v / (HX * HY);
            const int gx = x0 + dx, gy = y0 + dy, gz = z0 + dz;
            sk[it] = gx < Wi && gy < Hi && gz < Di;                              // past the volume: the op1/p1 zero border
            const long long off = sk[it] ? (((long long)gz * Hi + gy) * Wi + gx) : 0;
            sv[it] = *reinterpret_cast<const float4*>(inb + off * CIN + cb * 16 + q * 4);
        }
        // [group][ks][64] floats: the ordinary class-major image (27 groups) or, for PAIR, the x-parity-paired image built once
        // at pack time (k_conv3d_t2_pair_pack, 18 groups)
#pragma unroll
        for (int it = 0; it < NWIT; ++it) {
            const int i = threadIdx.x + it * 256;
            wq[it] = *reinterpret_cast<const float4*>(wimg + (i < NW4 ? i : NW4 - 1) * 4);
        }
#pragma unroll
        for (int it = 0; it < NWIT; ++it) {
            const int i = threadIdx.x + it * 256;
            if (i < NW4) *reinterpret_cast<float4*>(wlds + i * 4) = wq[it];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * 256;
            if (i < NIN) *reinterpret_cast<float4*>(lds + i * 4) = sk[it] ? sv[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CTW; ++c)                                  // the skip values are registers from here on (hipcc would
#pragma unroll
        for (int k = 0; k < NACC; ++k) {                           // otherwise sink their loads down to the epilogue)
            ENERF_PIN_VGPR(rres[k][c].x); ENERF_PIN_VGPR(rres[k][c].y); ENERF_PIN_VGPR(rres[k][c].z); ENERF_PIN_VGPR(rres[k][c].w);
        }

    f32x4 acc[NACC][CTW];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[k][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* lbase[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int tile = wv * CTW + c, td = tile / QH, th = tile - td * QH;
        lbase[c] = lds + ((td * HY + th) * HX + j) * 16 + g * 4;
    }
    const float* wl = wlds + lane;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int ad = (a >> 2) & 1, ah = (a >> 1) & 1, aw = a & 1;
        float bv[CTW][NCB][4];
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float4 tq = *reinterpret_cast<const float4*>(lbase[c] + cb * NVOX * 16 + ((ad * HY + ah) * HX + aw) * 16);
                bv[c][cb][0] = tq.x; bv[c][cb][1] = tq.y; bv[c][cb][2] = tq.z; bv[c][cb][3] = tq.w;
            }
#pragma unroll
        for (int k = 0; k < NACC; ++k) {                          // PAIR: k = pd*2 + ph;  else: k = class
            const int grp = PAIR ? (t2_pair_valid(a, k) ? t2_pair_group(a, k) : -1) : t2_tap_of(k, a);   // compile-time
            if (grp < 0) continue;
            float aq[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) aq[ks] = wl[(grp * KS + ks) * 64];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int c = 0; c < CTW; ++c)
                    acc[k][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[ks], bv[c][ks >> 2][ks & 3], acc[k][c], 0, 0, 0);
        }
    }

    // ---- epilogue: BN scale/shift, skip add, (ReLU); PAIR: lane group g -> x parity g >> 1, channels 4(g & 1).. ----
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[r] = scale[c0 + r]; sh[r] = shift[c0 + r]; }
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        if (!live[c]) continue;
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = acc[k][c][r] * sc[r] + sh[r];
            y[0] += rres[k][c].x; y[1] += rres[k][c].y; y[2] += rres[k][c].z; y[3] += rres[k][c].w;
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = relu1(y[r]);
            }
            unsigned od = oo[k][c];
            if (PAIR && out_planar) {                              // channels-last offset -> (batch, quad plane, voxel) offset
                const unsigned vox = (od - (unsigned)c0) / COUT, vb = vox - (unsigned)b * nvo;
                od = (((unsigned)b * 2u + (unsigned)(g & 1)) * nvo + vb) * 4u;
            }
            *reinterpret_cast<float4*>(out + od) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

// x-parity-paired A operands of a 16 -> 8 transposed layer, from its class-major packed image (conv3d.hip k_conv3d_pack,
// one 16-row tile whose rows 8-15 are zero): paired[(grp*4 + ks)*64 + lane], lane = (g, row): row < 8 -> class (pd, ph, 0)'s
// tap for offset a (zero when aw = 1: that class does not read it), row >= 8 -> class (pd, ph, 1)'s.
long long conv3d_t2_pair_floats() { return 18LL * 4 * 64; }
__global__ __launch_bounds__(256) void k_conv3d_t2_pair_pack(const float* __restrict__ packed, float* __restrict__ paired) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 18 * 4 * 64) return;
    const int ln = i & 63, ks = (i >> 6) & 3, grp = i >> 8;
    int a = 0, pp = 0, n = 0;                                      // invert the a-major enumeration of the valid (a, pp)
    for (int a2 = 0; a2 < 8; ++a2)
        for (int p2 = 0; p2 < 4; ++p2)
            if (t2_pair_valid(a2, p2)) { if (n == grp) { a = a2; pp = p2; } ++n; }
    const int row = ln & 15, gg = ln >> 4, pw = row >> 3;
    const int tap = t2_tap_of(pp * 2 + pw, a);
    paired[i] = tap >= 0 ? packed[(tap * 4 + ks) * 64 + gg * 16 + (row & 7)] : 0.f;
}
void launch_conv3d_t2_pair_pack(const float* packed, float* paired, hipStream_t st) {
    ENERF_LAUNCH_SIMPLE(k_conv3d_t2_pair_pack, cdiv(18 * 4 * 64, 256), 256, 0, st, packed, paired);
}

template <int CIN, int COUT, int QD, int QH>
static void launch_t2_all(const Conv3dDesc& L, const float* in, const float* residual, float* out, int B, int Di, int Hi, int Wi,
                          hipStream_t st) {
    constexpr int NVOX = (QD + 1) * (QH + 1) * 17, NGRP = COUT == 8 ? 18 : 27;
    const int nbd = cdiv(Di, QD), nbh = cdiv(Hi, QH), nbw = cdiv(Wi, 16);
    const size_t shmem = ((size_t)(CIN / 16) * NVOX * 16 + (size_t)NGRP * (CIN / 4) * 64) * sizeof(float);
    const unsigned grid = (unsigned)((long long)B * nbd * nbh * nbw);
    ENERF_LAUNCH((k_conv3d_t2_all<CIN, COUT, QD, QH>), grid, 256, shmem, st, COUT == 8 ? L.w_t2pair : L.w, L.scale, L.shift, in,
                 residual, out, L.relu, B, Di, Hi, Wi, nbd, nbh, nbw, L.out_planar);
}
// conv11 (16 -> 8, class-paired) and conv9 (32 -> 16) of both nets.  Returns false if the shape is not handled.
bool launch_conv3d_t2_all(const Conv3dDesc& L, const float* in, const float* residual, float* out, int B, int Di, int Hi,
                          int Wi, hipStream_t st) {
    if (L.kind != kConvT2) return false;
    if (L.out_planar && !(L.cin == 16 && L.cout == 8 && L.w_t2pair != nullptr)) return false;
    if ((long long)B * 8 * Di * Hi * Wi * L.cout >= (1LL << 32)) return false;                        // 32-bit output offsets
    // q-boxes (measured, tools/bench_conv3d_layers.py): conv11 1 x 4 x 16 (15.8 / 9.7 us at level 1 / 0; 2 x 4 x 16: 16.5 / 10.5;
    // 2 x 8 x 16: 21.0 / 12.5), conv9 1 x 4 x 16 (10.9 us; 1 x 8 x 16: 15.9)
    if (L.cin == 16 && L.cout == 8 && L.w_t2pair != nullptr) { launch_t2_all<16, 8, 1, 4>(L, in, residual, out, B, Di, Hi, Wi, st); return true; }
    if (L.cin == 32 && L.cout == 16) { launch_t2_all<32, 16, 1, 4>(L, in, residual, out, B, Di, Hi, Wi, st); return true; }
    return false;
}

// Cin = 16, Cout = 8 transposed layers.  Returns false if the shape is not handled.
bool launch_conv3d_t2_lds(const Conv3dDesc& L, const float* in, const float* residual, float* out, int B, int Di, int Hi,
                          int Wi, hipStream_t st) {
    if (L.kind != kConvT2 || L.cin != 16 || L.cout != 8) return false;
    const int nbd = cdiv(Di, 2), nbh = cdiv(Hi, 4), nbw = cdiv(Wi, 16);
    const size_t shmem = ((size_t)3 * 5 * 17 * 16 + 27 * 4 * 64) * sizeof(float);
    const unsigned grid = (unsigned)((long long)B * nbd * nbh * nbw);
    ENERF_LAUNCH((k_conv3d_t2_lds<16>), grid, 256, shmem, st, L.w, L.scale, L.shift, in, residual, out, L.cout, L.relu, B, Di,
                 Hi, Wi, nbd, nbh, nbw);
    return true;
}

}  // namespace enerf
