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
