// nerf_layout.h — the packed weight image of one NeRF/Agg MLP (nerf.py:6-89) and the MFMA helpers shared by the render
// kernel (render.hip) and the training-mode MLP kernels (mlp_train.hip).
#pragma once
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
// ---- device helpers ----------------------------------------------------------------------------------
__device__ __forceinline__ float group_sum(float v) { return group_sum4(v); }   // sum over the 4 lane groups (lanes j, j+16, j+32, j+48)
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

// NV accumulators x NK k-steps of MFMAs whose A operands (one float per lane and tile) come from LDS: the A values of
// k-step ks+1 are read while the MFMAs of k-step ks run (a two-deep register ring), pinned with sched_barrier — left
// alone, hipcc emits `ds_read; s_waitcnt lgkmcnt(0); mfma; mfma` and exposes one LDS latency per MFMA pair.
// A(e) = this lane's element of tile e = ks*NV + v; B(ks) = the k-step's B operand.
#ifndef ENERF_CHAIN_DEPTH
#define ENERF_CHAIN_DEPTH 2          // register ring depth: A operands are read DEPTH-1 k-steps ahead of their MFMAs
#endif
template <int NK, int NV, class AF, class BF>
__device__ __forceinline__ void mfma_chain(f32x4 (&acc)[NV], AF A, BF B) {
    constexpr int DP = ENERF_CHAIN_DEPTH;
    float ring[DP][NV];
#pragma unroll
    for (int k = 0; k < DP - 1 && k < NK; ++k)
#pragma unroll
        for (int v = 0; v < NV; ++v) ring[k][v] = A(k * NV + v);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        if (ks + DP - 1 < NK) {
#pragma unroll
            for (int v = 0; v < NV; ++v) ring[(ks + DP - 1) % DP][v] = A((ks + DP - 1) * NV + v);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float b = B(ks);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = ENERF_MFMA(ring[ks % DP][v], b, acc[v]);
        __builtin_amdgcn_sched_barrier(0);
    }
}


}  // namespace enerf
