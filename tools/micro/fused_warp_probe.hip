// Microbenchmark (dev tool): what would the STAGING HALF of a "warp + variance inside conv0" kernel cost?
// The asynchronously staged conv0 (k_conv3d_s1_b4g) gives a block a 4 x 8 x 16 box of output voxels and reads the haloed
// 6 x 10 x 18 = 1080-voxel input box.  Fused, the block would have to PRODUCE that box instead: project every haloed voxel into
// the S source views, gather 4 taps per view, form the variance — the arithmetic of k_feature_volume (volume.hip), per box,
// halo included — and park all C channels of the 1080 voxels in LDS (the gathers deliver a voxel's channels together, while
// the conv consumes one channel quad per pass).
//   mode 0: the fused kernel's staging phase: warp the haloed box into LDS (C*1080*4 B: 138 KB at C = 32 -> ONE block per CU)
//   mode 1: today's staging traffic for comparison: copy the same haloed box of a PRECOMPUTED channels-last volume into LDS
//   mode 2: mode 0's gathers and arithmetic with the LDS footprint taken out of the picture (a 16 KB ring that the box wraps
//           around: full occupancy) — the halo redundancy alone
// All end with a token reduction of the box so that nothing is dead code.  Compare mode 0 with (k_feature_volume + mode 1).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I enerf_amd/csrc -I include -shared -fPIC
//       tools/micro/fused_warp_probe.hip -o tools/micro/fwp.so ; driven by tools/micro/fused_warp_probe.py
#include "kernels.h"

namespace enerf {

template <int CQ, int MODE>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ feat, const float* __restrict__ proj,
                                               const float* __restrict__ dv, const float* __restrict__ vol_in, int B, int S,
                                               int Hs, int Ws, int D, int h, int w, float* __restrict__ sink) {
    constexpr int C = CQ * 4, BD = 4, BH = 8, BW = 16, HD = BD + 2, HH = BH + 2, HW = BW + 2, NV = HD * HH * HW;
    extern __shared__ float box[];                        // [NV][C]
    const int nbw = (w + BW - 1) / BW, nbh = (h + BH - 1) / BH, nbd = (D + BD - 1) / BD;
    int blk = blockIdx.x;
    const int bx = blk % nbw; blk /= nbw;
    const int by = blk % nbh; blk /= nbh;
    const int bz = blk % nbd;
    const int b = blk / nbd;
    const int cq = threadIdx.x & (CQ - 1);
    const int lane = threadIdx.x & 63, lead = lane & ~(CQ - 1);
    const float inv_half_w = 1.f / (float)((Ws - 1) / 2.0), inv_half_h = 1.f / (float)((Hs - 1) / 2.0);
    const unsigned img = (unsigned)(Hs * Ws * C);
    const float inv_s = 1.f / (float)S;
    for (int v0 = 0; v0 < NV; v0 += 256 / CQ) {
        const int v = min(v0 + (int)(threadIdx.x / CQ), NV - 1);      // tail lanes shadow the last voxel (they take part in the broadcasts)
        const int dx = v % HW, dy = (v / HW) % HH, dz = v / (HW * HH);
        const int x = bx * BW - 1 + dx, y = by * BH - 1 + dy, z = bz * BD - 1 + dz;
        const bool inside = (unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h && (unsigned)z < (unsigned)D;
        const int xc = min(max(x, 0), w - 1), yc = min(max(y, 0), h - 1), zc = min(max(z, 0), D - 1);
        const unsigned vox = (((unsigned)b * D + zc) * h + yc) * w + xc;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1) {
            if (inside) o = *reinterpret_cast<const float4*>(vol_in + (long long)vox * C + cq * 4);
        } else {
            const float depth = dv[vox];
            const float fx = (float)xc, fy = (float)yc;
            float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s0 = 0; s0 < S; s0 += CQ) {
                const int sv = min(s0 + cq, S - 1);
                const float* P = proj + (b * S + sv) * 12;
                const float px = P[0] * fx + P[1] * fy + P[2] + P[3] / depth;
                const float py = P[4] * fx + P[5] * fy + P[6] + P[7] / depth;
                const float pz = P[8] * fx + P[9] * fy + P[10] + P[11] / depth;
                const float zz = clamp_min(pz, 1e-6f);
                const float gx = (px / zz) * inv_half_w - 1.f, gy = (py / zz) * inv_half_h - 1.f;
                const Taps2 t = gs_taps2<false>(gs_unnorm(gx, Ws), gs_unnorm(gy, Hs), Ws, Hs);
                const unsigned vb = (unsigned)(b * S + sv) * img;
                const int r0 = mul24(t.y0, Ws), r1 = mul24(t.y1, Ws);
                const int my_o[4] = {(int)(vb + (unsigned)mul24(r0 + t.x0, C)), (int)(vb + (unsigned)mul24(r0 + t.x1, C)),
                                     (int)(vb + (unsigned)mul24(r1 + t.x0, C)), (int)(vb + (unsigned)mul24(r1 + t.x1, C))};
                const float my_w[4] = {t.w00, t.w01, t.w10, t.w11};
#pragma unroll
                for (int k = 0; k < CQ; ++k) {
                    if (s0 + k >= S) break;
                    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned of = (unsigned)__shfl(my_o[c], lead + k) + (unsigned)(cq * 4);
                        const float wgt = __shfl(my_w[c], lead + k);
                        const float4 tv = *reinterpret_cast<const float4*>(feat + of);
                        r.x += tv.x * wgt; r.y += tv.y * wgt; r.z += tv.z * wgt; r.w += tv.w * wgt;
                    }
                    s1.x += r.x; s1.y += r.y; s1.z += r.z; s1.w += r.w;
                    s2.x += r.x * r.x; s2.y += r.y * r.y; s2.z += r.z * r.z; s2.w += r.w * r.w;
                }
            }
            float m;
            m = s1.x * inv_s; o.x = s2.x * inv_s - m * m;
            m = s1.y * inv_s; o.y = s2.y * inv_s - m * m;
            m = s1.z * inv_s; o.z = s2.z * inv_s - m * m;
            m = s1.w * inv_s; o.w = s2.w * inv_s - m * m;
            if (!inside) o = make_float4(0.f, 0.f, 0.f, 0.f);      // conv0's zero padding
        }
        if (v0 + (int)(threadIdx.x / CQ) < NV) *reinterpret_cast<float4*>(box + ((v * C + cq * 4) & (MODE == 2 ? 4095 : 0x7fffffff))) = o;
    }
    __syncthreads();
    float acc = 0.f;                                      // token consumer: one pass over the box
    for (int i = threadIdx.x; i < (MODE == 2 ? 4096 : NV * C); i += 256) acc += box[i];
    sink[(long long)blockIdx.x * 256 + threadIdx.x] = acc;
}

}  // namespace enerf

extern "C" int probe_launch(const float* feat, const float* proj, const float* dv, const float* vol_in, int B, int S, int C,
                            int Hs, int Ws, int D, int h, int w, int mode, float* sink, void* stream) {
    using namespace enerf;
    const int nb = B * ((D + 3) / 4) * ((h + 7) / 8) * ((w + 15) / 16);
    const size_t shmem = mode == 2 ? 16384 : (size_t)6 * 10 * 18 * C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define GO(CQ, MODE)                                                                                                        \
    do {                                                                                                                    \
        hipFuncSetAttribute((const void*)k_probe<CQ, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);         \
        hipLaunchKernelGGL((k_probe<CQ, MODE>), dim3(nb), dim3(256), shmem, st, feat, proj, dv, vol_in, B, S, Hs, Ws, D, h, w, sink); \
    } while (0)
    if (C == 32 && mode == 0) GO(8, 0);
    else if (C == 32 && mode == 1) GO(8, 1);
    else if (C == 32) GO(8, 2);
    else if (C == 16 && mode == 0) GO(4, 0);
    else if (C == 16 && mode == 1) GO(4, 1);
    else if (C == 16) GO(4, 2);
    else return -1;
    return nb;
}
