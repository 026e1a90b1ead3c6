// train.hip — per-layer building blocks of the TRAINING-mode cost-regularisation networks:
//   * one 3x3x3 convolution layer on the inference path's MFMA kernels with an identity epilogue (enerf_conv3d_layer) —
//     in training the BatchNorm cannot be folded (batch statistics), and the same kernels serve the INPUT gradients:
//     the dgrad of a stride-2 conv is the transposed-conv kernel on the same weight tensor, the dgrad of a transposed conv
//     is the stride-2 conv kernel, the dgrad of a stride-1 conv is the stride-1 kernel on the flipped/transposed weights;
//   * BatchNorm3d in training mode (ConvBnReLU3D utils.py:22-33; cost_reg_net.py:25-31 without ReLU): per-channel sums
//     (enerf_channel_sums) and a per-channel affine map with optional ReLU mask / residual (enerf_channel_affine) — the two
//     kernels cover the forward (statistics, normalise+ReLU+skip add) and the backward (d gamma / d beta sums, d input).
// Tensors are channels-last (n positions, C channels), C a multiple of 4.  HBM-bound elementwise / reduction work.
#include "kernels.h"

namespace enerf {

// sums[c] = sum_p a[p][c]*m ;  sums[C + c] = sum_p a[p][c]*m*b[p][c] ;  m = (zm[p][c]*ms[c] + mh[c] > 0) or 1
template <bool SAME, bool MASK>          // a == b (the forward statistics: read once); a ReLU mask from zm
__global__ __launch_bounds__(256) void k_channel_sums(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ zm, const float* __restrict__ ms,
                                                      const float* __restrict__ mh, long long n, int C,
                                                      double* __restrict__ sums, double* __restrict__ partials) {
    __shared__ double red[2][256][4];
    const int CQ = C >> 2;                                 // float4 channel groups
    const int cq = threadIdx.x % CQ, lane_p = threadIdx.x / CQ, ppb = 256 / CQ;   // positions per block step
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {1.f, 1.f, 1.f, 1.f};
    if (MASK)
        for (int k = 0; k < 4; ++k) { sc[k] = ms[cq * 4 + k]; sh[k] = mh[cq * 4 + k]; }
    // four positions per thread and iteration, all their loads requested before the first is used (one 16-byte load in flight per
    // thread streamed at 0.6 TB/s: two blocks per CU cannot hide a memory round trip per iteration — and with the two optional
    // tensors as RUNTIME conditions hipcc put a branch and an s_waitcnt vmcnt(0) between any two loads: compile-time variants)
    constexpr int U = 4;
    for (long long p0 = (long long)blockIdx.x * ppb * U + lane_p; p0 < n; p0 += (long long)gridDim.x * ppb * U) {
        float4 av[U], bv[U], zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = p0 + (long long)u * ppb, pc = p < n ? p : n - 1;     // clamped: always loads, masked when used
            av[u] = *reinterpret_cast<const float4*>(a + pc * C + cq * 4);
            if (!SAME) bv[u] = *reinterpret_cast<const float4*>(b + pc * C + cq * 4);
            if (MASK) zv[u] = *reinterpret_cast<const float4*>(zm + pc * C + cq * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p0 + (long long)u * ppb >= n) continue;
            float am[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
            const float4 bq = SAME ? av[u] : bv[u];
            const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
            if (MASK) {
                const float zz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
                for (int k = 0; k < 4; ++k) am[k] = (zz[k] * sc[k] + sh[k] > 0.f) ? am[k] : 0.f;
            }
            for (int k = 0; k < 4; ++k) { s1[k] += (double)am[k]; s2[k] += (double)am[k] * (double)bb[k]; }
        }
    }
    // block sums: butterfly over the lanes of a wave that share a channel quad (lane bits >= log2 CQ), then the four waves through
    // LDS.  (Before: thread cq walked all 256 / CQ position lanes in LDS serially — 1024 dependent fp64 reads for C = 8, ~40 us
    // at the end of EVERY block: the kernel streamed its 31 MB layers at 0.5 TB/s.)
    for (int m = 32; m >= CQ; m >>= 1)
        for (int k = 0; k < 4; ++k) { s1[k] += __shfl_xor(s1[k], m); s2[k] += __shfl_xor(s2[k], m); }
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln < CQ)
        for (int k = 0; k < 4; ++k) { red[0][wv * 64 + ln][k] = s1[k]; red[1][wv * 64 + ln][k] = s2[k]; }
    __syncthreads();
    if (threadIdx.x < CQ) {
        // (CQ <= 64 divides 64: lane ln of a wave holds quad ln % CQ = ln for ln < CQ)
        for (int k = 0; k < 4; ++k) {
            double t1 = 0, t2 = 0;
            for (int q = 0; q < 4; ++q) { t1 += red[0][q * 64 + cq][k]; t2 += red[1][q * 64 + cq][k]; }
            if (partials != nullptr) {                     // one row of 2C sums per block, summed by k_channel_sums_finish
                partials[(long long)blockIdx.x * 2 * C + cq * 4 + k] = t1;
                partials[(long long)blockIdx.x * 2 * C + C + cq * 4 + k] = t2;
            } else {
                atomicAdd(sums + cq * 4 + k, t1);
                atomicAdd(sums + C + cq * 4 + k, t2);
            }
        }
    }
}

// sums[c] = sum over the nb block rows of partials[blk][c], c < 2C <= 128: thread t adds the rows t / 2C, t / 2C + 1024 / 2C, ...
// (fixed order: deterministic), the row groups are combined through LDS.
__global__ __launch_bounds__(1024) void k_channel_sums_finish(const double* __restrict__ partials, int nb, int C2,
                                                              double* __restrict__ sums) {
    __shared__ double red[1024];
    const int c = threadIdx.x % C2, r0 = threadIdx.x / C2, nr = 1024 / C2;
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;                  // four rows in flight per thread (the sum order stays fixed)
    int blk = r0;
    for (; blk + 3 * nr < nb; blk += 4 * nr) {
        const double v0 = partials[(long long)blk * C2 + c], v1 = partials[(long long)(blk + nr) * C2 + c];
        const double v2 = partials[(long long)(blk + 2 * nr) * C2 + c], v3 = partials[(long long)(blk + 3 * nr) * C2 + c];
        t0 += v0; t1 += v1; t2 += v2; t3 += v3;
    }
    for (; blk < nb; blk += nr) t0 += partials[(long long)blk * C2 + c];
    red[threadIdx.x] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (threadIdx.x < C2) {
        double tot = 0;
        for (int q = 0; q < nr; ++q) tot += red[q * C2 + c];
        sums[c] = tot;
    }
}

// out[p][c] = f( a[p][c]*m*pp[c] + (b ? b[p][c]*qq[c] : 0) + rr[c] ) (+ residual[p][c]);  f = ReLU if relu;  m as above
__global__ __launch_bounds__(256) void k_channel_affine(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ pp, const float* __restrict__ qq,
                                                        const float* __restrict__ rr, const float* __restrict__ zm,
                                                        const float* __restrict__ ms, const float* __restrict__ mh,
                                                        const float* __restrict__ residual, int relu, long long n4, int CQ,
                                                        float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of channels
    if (i >= n4) return;
    const int c0 = (int)(i % CQ) * 4;
    const float4 av = *reinterpret_cast<const float4*>(a + i * 4);
    float x[4] = {av.x, av.y, av.z, av.w};
    if (zm != nullptr) {
        const float4 zv = *reinterpret_cast<const float4*>(zm + i * 4);
        const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
        for (int k = 0; k < 4; ++k) x[k] = (zz[k] * ms[c0 + k] + mh[c0 + k] > 0.f) ? x[k] : 0.f;
    }
    float y[4];
    for (int k = 0; k < 4; ++k) y[k] = x[k] * pp[c0 + k] + rr[c0 + k];
    if (b != nullptr) {
        const float4 bv = *reinterpret_cast<const float4*>(b + i * 4);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
        for (int k = 0; k < 4; ++k) y[k] += bb[k] * qq[c0 + k];
    }
    if (relu) for (int k = 0; k < 4; ++k) y[k] = fmaxf(y[k], 0.f);
    if (residual != nullptr) {
        const float4 rv = *reinterpret_cast<const float4*>(residual + i * 4);
        y[0] += rv.x; y[1] += rv.y; y[2] += rv.z; y[3] += rv.w;
    }
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(y[0], y[1], y[2], y[3]);
}

// ---- BatchNorm-train per-channel coefficients: the C-sized arithmetic between the statistics kernel (and its all-reduce
// under SyncBatchNorm) and the affine kernel, in ONE launch instead of ~20 C-element torch kernels per layer and direction
// (40 BatchNorm layers x 2 directions per training step).  fp64 like the torch expressions it replaces. ----
// forward: sums = [sum z, sum z^2] -> mean, invstd (fp64), scale = gamma*invstd, shift = beta - mean*gamma*invstd (fp32),
// running statistics updated in place (unbiased variance; momentum < 0: cumulative average 1/num_batches_tracked)
// the forward coefficients of one channel from its two sums (shared by k_bn_train_coeffs and the fused k_bn_finish_coeffs)
__device__ __forceinline__ void bn_fwd_channel(int c, int C, double s1, double s2, double n, long long tracked, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, double eps, double momentum, float* __restrict__ running_mean,
                                               float* __restrict__ running_var, double* __restrict__ mean_invstd, float* __restrict__ scale_shift) {
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;                     // biased (normalisation)
    var = var > 0.0 ? var : 0.0;
    const double invstd = 1.0 / sqrt(var + eps);
    mean_invstd[c] = mean;
    mean_invstd[C + c] = invstd;
    const double g = (double)gamma[c];
    scale_shift[c] = (float)(g * invstd);
    scale_shift[C + c] = (float)((double)beta[c] - mean * g * invstd);
    if (running_mean != nullptr) {
        const float mom = (float)(momentum >= 0.0 ? momentum : 1.0 / (double)tracked);
        const double nm1 = n - 1.0 > 1.0 ? n - 1.0 : 1.0;
        running_mean[c] = running_mean[c] * (1.f - mom) + (float)mean * mom;
        running_var[c] = running_var[c] * (1.f - mom) + (float)(var * (n / nm1)) * mom;
    }
}
__device__ __forceinline__ void bn_bwd_channel(int c, int C, double l1, double l2, double sg, double sgz, double n,
                                               const double* __restrict__ mean_invstd, const float* __restrict__ scale,
                                               float* __restrict__ dgamma_dbeta, float* __restrict__ k2k3) {
    const double mean = mean_invstd[c], invstd = mean_invstd[C + c];
    dgamma_dbeta[c] = (float)(invstd * (l2 - mean * l1));
    dgamma_dbeta[C + c] = (float)l1;
    const double sc = (double)scale[c];
    const double k2 = -sc * invstd * (invstd * (sgz - mean * sg)) / n;
    k2k3[c] = (float)k2;
    k2k3[C + c] = (float)(-sc * sg / n - k2 * mean);
}
// the row reduction of k_channel_sums_finish as a device function: every thread of a 1024-thread block takes part; on return
// tot[0 .. C2) (shared) holds the sums (same order of additions as k_channel_sums_finish: bit-identical)
__device__ __forceinline__ void finish_rows(const double* __restrict__ partials, int nb, int C2, double* red, double* tot) {
    const int c = threadIdx.x % C2, r0 = threadIdx.x / C2, nr = 1024 / C2;
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    int blk = r0;
    for (; blk + 3 * nr < nb; blk += 4 * nr) {
        const double v0 = partials[(long long)blk * C2 + c], v1 = partials[(long long)(blk + nr) * C2 + c];
        const double v2 = partials[(long long)(blk + 2 * nr) * C2 + c], v3 = partials[(long long)(blk + 3 * nr) * C2 + c];
        t0 += v0; t1 += v1; t2 += v2; t3 += v3;
    }
    for (; blk < nb; blk += nr) t0 += partials[(long long)blk * C2 + c];
    red[threadIdx.x] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (threadIdx.x < C2) {
        double t = 0;
        for (int q = 0; q < nr; ++q) t += red[q * C2 + c];
        tot[c] = t;
    }
    __syncthreads();
}
// Round 5: the statistics' final reduction AND the C-sized coefficient arithmetic in ONE launch (without SyncBatchNorm nothing sits
// between them): a training step had 46 BatchNorm directions x (partials, finish, coefficients) = 138 launches, 92 of them
// single-block kernels at the ~4.6 us floor of a graph node.  Same additions and the same coefficient code: bit-identical.
__global__ __launch_bounds__(1024) void k_bn_finish_coeffs(const double* __restrict__ partials, int nb, double n, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, double eps, double momentum,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           long long* __restrict__ nbt, int nbt_increment, int C,
                                                           double* __restrict__ sums_out, double* __restrict__ mean_invstd,
                                                           float* __restrict__ scale_shift) {
    __shared__ double red[1024], tot[128];
    const long long tracked = nbt != nullptr ? *nbt + nbt_increment : 1;
    finish_rows(partials, nb, 2 * C, red, tot);                        // (its barriers order every thread's read of *nbt before the store)
    const int c = threadIdx.x;
    if (c == 0 && nbt != nullptr && nbt_increment) *nbt = tracked;
    if (c < 2 * C && sums_out != nullptr) sums_out[c] = tot[c];
    if (c < C) bn_fwd_channel(c, C, tot[c], tot[C + c], n, tracked, gamma, beta, eps, momentum, running_mean, running_var, mean_invstd, scale_shift);
}
__global__ __launch_bounds__(1024) void k_bn_finish_bwd_coeffs(const double* __restrict__ partials, int nb, double n,
                                                               const double* __restrict__ mean_invstd, const float* __restrict__ scale, int C,
                                                               float* __restrict__ dgamma_dbeta, float* __restrict__ k2k3) {
    __shared__ double red[1024], tot[128];
    finish_rows(partials, nb, 2 * C, red, tot);
    const int c = threadIdx.x;
    if (c < C) bn_bwd_channel(c, C, tot[c], tot[C + c], tot[c], tot[C + c], n, mean_invstd, scale, dgamma_dbeta, k2k3);
}

// ---- round 6: the coefficient launch folded into the AFFINE kernel's prologue (small / mid layers) ----
// Forward and backward of a BatchNorm were three launches each (partial sums; their reduction + the C-sized coefficients in one
// 1024-thread block; the affine pass), and for the 13 of 23 layers below ~8 MB all three sit at the 5 - 8 us floor of a graph node.
// Here every block of the affine kernel first redoes the reduction of the nb partial rows and the coefficient arithmetic itself —
// nb x 2C doubles from L2, a few KB — with 256 threads standing in for finish_rows' 1024 (same additions in the same order: the
// coefficients are bit-identical to k_bn_finish_*'s), keeps them in LDS, and block 0 alone stores what later kernels read (mean / invstd,
// scale / shift, the running statistics, d gamma / d beta).  The launcher uses it when nb * C <= 2048 (<= 32 KB of rows) and caps the
// grid at 512 blocks, so the redone reductions move <= 16 MB through L2 per layer; larger layers keep the three launches.
__device__ __forceinline__ void finish_rows256(const double* __restrict__ partials, int nb, int C2, double* red /* [1024] */, double* tot /* [128] */) {
    const int nr = 1024 / C2;
    for (int vt = threadIdx.x; vt < 1024; vt += 256) {       // virtual thread vt of finish_rows
        const int c = vt % C2, r0 = vt / C2;
        double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        int blk = r0;
        for (; blk + 3 * nr < nb; blk += 4 * nr) {
            const double v0 = partials[(long long)blk * C2 + c], v1 = partials[(long long)(blk + nr) * C2 + c];
            const double v2 = partials[(long long)(blk + 2 * nr) * C2 + c], v3 = partials[(long long)(blk + 3 * nr) * C2 + c];
            t0 += v0; t1 += v1; t2 += v2; t3 += v3;
        }
        for (; blk < nb; blk += nr) t0 += partials[(long long)blk * C2 + c];
        red[vt] = (t0 + t1) + (t2 + t3);
    }
    __syncthreads();
    if ((int)threadIdx.x < C2) {
        const int c = threadIdx.x;
        double t = 0;
        for (int q = 0; q < nr; ++q) t += red[q * C2 + c];
        tot[c] = t;
    }
    __syncthreads();
}
// the affine pass of k_channel_affine with the coefficients in LDS, grid-stride (same expression per element: same bits)
__device__ __forceinline__ void affine_pass(const float* __restrict__ a, const float* __restrict__ b, const float* pp, const float* qq, const float* rr,
                                            const float* __restrict__ zm, const float* ms, const float* mh, const float* __restrict__ residual,
                                            int relu, long long n4, int CQ, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int c0 = (int)(i % CQ) * 4;
        const float4 av = *reinterpret_cast<const float4*>(a + i * 4);
        float x[4] = {av.x, av.y, av.z, av.w};
        if (zm != nullptr) {
            const float4 zv = *reinterpret_cast<const float4*>(zm + i * 4);
            const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
            for (int k = 0; k < 4; ++k) x[k] = (zz[k] * ms[c0 + k] + mh[c0 + k] > 0.f) ? x[k] : 0.f;
        }
        float y[4];
        for (int k = 0; k < 4; ++k) y[k] = x[k] * pp[c0 + k] + rr[c0 + k];
        if (b != nullptr) {
            const float4 bv = *reinterpret_cast<const float4*>(b + i * 4);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
            for (int k = 0; k < 4; ++k) y[k] += bb[k] * qq[c0 + k];
        }
        if (relu) for (int k = 0; k < 4; ++k) y[k] = fmaxf(y[k], 0.f);
        if (residual != nullptr) {
            const float4 rv = *reinterpret_cast<const float4*>(residual + i * 4);
            y[0] += rv.x; y[1] += rv.y; y[2] += rv.z; y[3] += rv.w;
        }
        *reinterpret_cast<float4*>(out + i * 4) = make_float4(y[0], y[1], y[2], y[3]);
    }
}
__global__ __launch_bounds__(256) void k_bn_affine_fwd(const float* __restrict__ z, const double* __restrict__ partials, int nb, double n,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, double eps, double momentum,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
                                                       int nbt_increment, int C, double* __restrict__ sums_out, double* __restrict__ mean_invstd,
                                                       float* __restrict__ scale_shift, const float* __restrict__ residual, int relu, long long n4,
                                                       float* __restrict__ out) {
    __shared__ double red[1024], tot[128];
    __shared__ double mi[128];
    __shared__ float ss[128];
    const long long tracked = (blockIdx.x == 0 && nbt != nullptr) ? *nbt + nbt_increment : 1;       // (only block 0 touches the running statistics)
    finish_rows256(partials, nb, 2 * C, red, tot);
    const int c = threadIdx.x;
    const bool first = blockIdx.x == 0;
    if (first && c == 0 && nbt != nullptr && nbt_increment) *nbt = tracked;
    if (first && c < 2 * C && sums_out != nullptr) sums_out[c] = tot[c];
    if (c < C) {       // bn_fwd_channel's arithmetic into LDS; block 0 also stores it (and updates the running statistics)
        if (first) {
            bn_fwd_channel(c, C, tot[c], tot[C + c], n, tracked, gamma, beta, eps, momentum, running_mean, running_var, mean_invstd, scale_shift);
            ss[c] = scale_shift[c]; ss[C + c] = scale_shift[C + c];
        } else {
            bn_fwd_channel(c, C, tot[c], tot[C + c], n, 1, gamma, beta, eps, momentum, nullptr, nullptr, mi, ss);
        }
    }
    __syncthreads();
    affine_pass(z, nullptr, ss, nullptr, ss + C, nullptr, nullptr, nullptr, residual, relu, n4, C / 4, out);
}
__global__ __launch_bounds__(256) void k_bn_affine_bwd(const float* __restrict__ g, const float* __restrict__ z, const double* __restrict__ partials,
                                                       int nb, double n, const double* __restrict__ mean_invstd, const float* __restrict__ scale_shift,
                                                       int relu_mask, int C, float* __restrict__ dgamma_dbeta, long long n4, float* __restrict__ out) {
    __shared__ double red[1024], tot[128];
    __shared__ float dgb[128], k23[128], ss[128];
    finish_rows256(partials, nb, 2 * C, red, tot);
    const int c = threadIdx.x;
    if (c < C) {
        bn_bwd_channel(c, C, tot[c], tot[C + c], tot[c], tot[C + c], n, mean_invstd, scale_shift, blockIdx.x == 0 ? dgamma_dbeta : dgb, k23);
        ss[c] = scale_shift[c]; ss[C + c] = scale_shift[C + c];
    }
    __syncthreads();
    // d z = g m scale + z k2 + k3,  m = (z scale + shift > 0) under a ReLU
    affine_pass(g, z, ss, k23, k23 + C, relu_mask ? z : nullptr, ss, ss + C, nullptr, 0, n4, C / 4, out);
}

__global__ void k_bn_train_coeffs(const double* __restrict__ sums, const double* __restrict__ count_dev, double count_host,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, double eps, double momentum,
                                  float* __restrict__ running_mean, float* __restrict__ running_var,
                                  long long* __restrict__ nbt, int nbt_increment, int C, double* __restrict__ mean_invstd,
                                  float* __restrict__ scale_shift) {
    const int c = threadIdx.x;
    const double n = count_dev != nullptr ? *count_dev : count_host;
    // num_batches_tracked counts this batch: either the caller incremented it already (nbt_increment = 0) or this launch
    // does (one block: every thread reads the old value, then thread 0 stores old + 1)
    const long long tracked = nbt != nullptr ? *nbt + nbt_increment : 1;
    __syncthreads();
    if (c == 0 && nbt != nullptr && nbt_increment) *nbt = tracked;
    if (c >= C) return;
    const double mean = sums[c] / n;
    double var = sums[C + c] / n - mean * mean;            // biased (normalisation)
    var = var > 0.0 ? var : 0.0;
    const double invstd = 1.0 / sqrt(var + eps);
    mean_invstd[c] = mean;
    mean_invstd[C + c] = invstd;
    const double g = (double)gamma[c];
    scale_shift[c] = (float)(g * invstd);
    scale_shift[C + c] = (float)((double)beta[c] - mean * g * invstd);
    if (running_mean != nullptr) {
        const float mom = (float)(momentum >= 0.0 ? momentum : 1.0 / (double)tracked);
        const double nm1 = n - 1.0 > 1.0 ? n - 1.0 : 1.0;
        running_mean[c] = running_mean[c] * (1.f - mom) + (float)mean * mom;
        running_var[c] = running_var[c] * (1.f - mom) + (float)(var * (n / nm1)) * mom;
    }
}
// backward: local sums [sum g*m, sum g*m*z] -> d gamma, d beta (parameter gradients stay per-rank, like torch's
// SyncBatchNorm); global sums -> the input-gradient coefficients d z = g*m*scale + z*k2 + k3
__global__ void k_bn_train_bwd_coeffs(const double* __restrict__ local, const double* __restrict__ global_,
                                      const double* __restrict__ count_dev, double count_host,
                                      const double* __restrict__ mean_invstd, const float* __restrict__ scale, int C,
                                      float* __restrict__ dgamma_dbeta, float* __restrict__ k2k3) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const double n = count_dev != nullptr ? *count_dev : count_host;
    const double mean = mean_invstd[c], invstd = mean_invstd[C + c];
    dgamma_dbeta[c] = (float)(invstd * (local[C + c] - mean * local[c]));
    dgamma_dbeta[C + c] = (float)local[c];
    const double sc = (double)scale[c], sg = global_[c], sgz = global_[C + c];
    const double k2 = -sc * invstd * (invstd * (sgz - mean * sg)) / n;
    k2k3[c] = (float)k2;
    k2k3[C + c] = (float)(-sc * sg / n - k2 * mean);
}

// ---- adjoint of the FeatureNet's 2x bilinear upsampling (feature_net.py:24-25: F.interpolate(scale_factor=2, bilinear,
// align_corners=True)) on channels-last maps, in GATHER form: one thread per (coarse pixel, channel quad) sums w_y*w_x*g over the
// <= 5 x 5 fine pixels whose taps touch it, with the forward's own ac_lerp weights (conv2d.hip epilogue).  No atomics, every
// fine gradient is read by <= 4 coarse pixels (L2 hits); optional `add`: a second gradient of the coarse map summed in. ----
__global__ __launch_bounds__(256) void k_up2_adjoint(const float* __restrict__ g, const float* __restrict__ add, int N, int Hc, int Wc,
                                                     int CQ, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * Hc * Wc * CQ;
    if (i >= total) return;
    const int cq = (int)(i % CQ);
    long long r = i / CQ;
    const int xc = (int)(r % Wc); r /= Wc;
    const int yc = (int)(r % Hc);
    const int n = (int)(r / Hc);
    const int Hf = 2 * Hc, Wf = 2 * Wc, C = CQ * 4;
    const float sy = ac_scale(Hc, Hf), sx = ac_scale(Wc, Wf);
    // fine rows whose source position falls in (yc - 1, yc + 1): a superset bounded by the reciprocal scale, then filtered
    // with the exact forward tap indices
    const int ylo = max(0, (int)floorf((float)(yc - 1) / sy)), yhi = min(Hf - 1, (int)ceilf((float)(yc + 1) / sy));
    const int xlo = max(0, (int)floorf((float)(xc - 1) / sx)), xhi = min(Wf - 1, (int)ceilf((float)(xc + 1) / sx));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* gb = g + (long long)n * Hf * Wf * C + cq * 4;
    for (int y = ylo; y <= yhi; ++y) {
        const Lerp1 ly = ac_lerp(y, sy, Hc);
        const float wy = (ly.i0 == yc ? ly.l0 : 0.f) + (ly.i1 == yc ? ly.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int x = xlo; x <= xhi; ++x) {
            const Lerp1 lx = ac_lerp(x, sx, Wc);
            const float wx = (lx.i0 == xc ? lx.l0 : 0.f) + (lx.i1 == xc ? lx.l1 : 0.f);
            if (wx == 0.f) continue;
            const float4 v = *reinterpret_cast<const float4*>(gb + ((long long)y * Wf + x) * C);
            const float wgt = wy * wx;
            acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
        }
    }
    const long long o = (((long long)n * Hc + yc) * Wc + xc) * C + cq * 4;
    if (add != nullptr) {
        const float4 a = *reinterpret_cast<const float4*>(add + o);
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    *reinterpret_cast<float4*>(out + o) = acc;
}

}  // namespace enerf

using namespace enerf;
extern "C" {

static bool layer_has_t2pair(int cin, int cout, int kind) { return kind == kConvT2 && cin == 16 && cout == 8; }
long long enerf_conv3d_layer_packed_floats(int cin, int cout, int kind) {
    return conv3d_packed_floats(cin, cout, kind) + 2 * cdiv(cout, 16) * 16 + (layer_has_t2pair(cin, cout, kind) ? conv3d_t2_pair_floats() : 0);
}
int enerf_conv3d_layer_pack(const float* w, int cin, int cout, int kind, float* packed, enerf_stream_t stream) {
    REQUIRE(w && packed, "conv3d_layer_pack: null pointer");
    REQUIRE((cin == 8 || cin == 16 || cin == 32 || cin == 64) && cout >= 1 && cout <= 64 && kind >= kConvS1 && kind <= kConvT2,
            "conv3d_layer_pack: unsupported layer %d -> %d kind %d", cin, cout, kind);
    const long long wf = conv3d_packed_floats(cin, cout, kind);
    const int cp = cdiv(cout, 16) * 16;
    launch_conv3d_pack(w, nullptr, cout, nullptr, nullptr, nullptr, nullptr, 1e-5f, cin, cout, kind, packed, packed + wf,
                       packed + wf + cp, (hipStream_t)stream);
    if (layer_has_t2pair(cin, cout, kind)) launch_conv3d_t2_pair_pack(packed, packed + wf + 2 * cp, (hipStream_t)stream);
    return check_launch("conv3d_layer_pack");
}
int enerf_conv3d_layer(const float* packed, int cin, int cout, int kind, const float* in, const float* residual, float* out, int B,
                       int Di, int Hi, int Wi, const enerf_options_t* options, enerf_stream_t stream) {
    REQUIRE(packed && in && out && B > 0 && Di > 0 && Hi > 0 && Wi > 0, "conv3d_layer: bad arguments");
    const long long wf = conv3d_packed_floats(cin, cout, kind);
    const int cp = cdiv(cout, 16) * 16;
    Conv3dDesc d = {packed, packed + wf, packed + wf + cp, cin, cout, kind, 0, nullptr, nullptr,
                    layer_has_t2pair(cin, cout, kind) ? packed + wf + 2 * cp : nullptr};
    if (!launch_conv3d(d, in, residual, out, nullptr, B, Di, Hi, Wi, resolve_options(options), (hipStream_t)stream))
        return fail(ENERF_EINVAL, "conv3d_layer: no kernel for %d -> %d kind %d", cin, cout, kind);
    return check_launch("conv3d_layer");
}
// ---- one FeatureNet convolution (feature_net.py:7-22) with an identity epilogue (+ bias), channels-last in and out ----
long long enerf_conv2d_layer_packed_floats(int cin, int cout, int k) { return conv2d_packed_floats(cin, cout, k) + 2 * cdiv(cout, 16) * 16; }
int enerf_conv2d_layer_pack(const float* w, const float* bias, int cin, int cout, int k, float* packed, enerf_stream_t stream) {
    REQUIRE(w && packed, "conv2d_layer_pack: null pointer");
    REQUIRE((cin == 3 || cin == 8 || cin == 16 || cin == 32) && (cout == 8 || cout == 16 || cout == 32) && (k == 1 || k == 3 || k == 5),
            "conv2d_layer_pack: unsupported layer %d -> %d k %d", cin, cout, k);
    const long long wf = conv2d_packed_floats(cin, cout, k);
    const int cp = cdiv(cout, 16) * 16;
    launch_conv2d_pack(w, bias, nullptr, nullptr, nullptr, nullptr, 1e-5f, cin, cout, k, packed, packed + wf, packed + wf + cp,
                       (hipStream_t)stream);
    return check_launch("conv2d_layer_pack");
}
int enerf_conv2d_layer(const float* packed, int cin, int cout, int k, int stride, const float* in, const float* up, float* out, int N,
                       int Hi, int Wi, enerf_stream_t stream) {
    REQUIRE(packed && in && out && N > 0 && Hi > 0 && Wi > 0 && (stride == 1 || stride == 2), "conv2d_layer: bad arguments");
    const long long wf = conv2d_packed_floats(cin, cout, k);
    const int cp = cdiv(cout, 16) * 16, P = (k - 1) / 2;
    const int Ho = (Hi + 2 * P - k) / stride + 1, Wo = (Wi + 2 * P - k) / stride + 1;
    if (up) REQUIRE(Ho % 2 == 0 && Wo % 2 == 0, "conv2d_layer: the upsampled map needs even output sizes");
    Conv2dDesc d = {packed, packed + wf, packed + wf + cp, cin, cout, k, stride, 0, 0, nullptr, nullptr, nullptr};
    if (launch_conv2d(d, in, out, up, N, Hi, Wi, Ho / 2, Wo / 2, (hipStream_t)stream) != 0)
        return fail(ENERF_EINVAL, "conv2d_layer: no kernel for %d -> %d k %d stride %d", cin, cout, k, stride);
    return check_launch("conv2d_layer");
}
int enerf_up2_adjoint(const float* grad_fine, const float* add, int N, int Hc, int Wc, int C, float* grad_coarse, enerf_stream_t stream) {
    REQUIRE(grad_fine && grad_coarse && N > 0 && Hc > 1 && Wc > 1 && C >= 4 && C % 4 == 0, "up2_adjoint: bad arguments");
    const long long total = (long long)N * Hc * Wc * (C / 4);
    ENERF_LAUNCH_SIMPLE(k_up2_adjoint, (unsigned)cdivl(total, 256), 256, 0, (hipStream_t)stream, grad_fine, add, N, Hc, Wc, C / 4, grad_coarse);
    return check_launch("up2_adjoint");
}
static long long channel_sums_blocks(long long n, int C) {
    const int ppb = 256 / (C / 4);
    long long blocks = cdivl(n, (long long)ppb * 16);
    const long long cap = (long long)device_cu_count() * 2;        // two blocks per CU stream at full rate
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}
size_t enerf_channel_sums_workspace_bytes(long long n, int C) {
    if (n <= 0 || C < 4 || C > 64 || C % 4 != 0 || (256 % (C / 4)) != 0) return 0;
    return (size_t)channel_sums_blocks(n, C) * 2 * C * sizeof(double);
}
// Without a workspace every block ends with 2C fp64 atomics onto the same 2C addresses of the zeroed `sums`: ~1.3 ns each,
// SERIALISED across blocks — 21 us for 256 blocks at C = 32, more than streaming a 31 MB layer (tools/bench_channel_sums.py,
// profiles/r04_channel_sums.txt).  With one (enerf_channel_sums_workspace_bytes) the blocks store a row of partial sums each and
// a second small launch adds the rows: no atomics, no zeroing launch, deterministic.
int enerf_channel_sums_ws(const float* a, const float* b, const float* z_mask, const float* mask_scale, const float* mask_shift,
                          long long n, int C, double* sums, void* workspace, size_t workspace_bytes, enerf_stream_t stream) {
    REQUIRE(a && b && sums && n > 0 && C >= 4 && C <= 64 && C % 4 == 0 && (256 % (C / 4)) == 0, "channel_sums: bad arguments (C in 4..64, power-of-two quads)");
    if (z_mask) REQUIRE(mask_scale && mask_shift, "channel_sums: mask needs its scale/shift");
    const long long blocks = channel_sums_blocks(n, C);
    double* partials = nullptr;
    if (workspace != nullptr) {
        REQUIRE(workspace_bytes >= (size_t)blocks * 2 * C * sizeof(double), "channel_sums: workspace of %zu bytes, need %zu", workspace_bytes,
                (size_t)blocks * 2 * C * sizeof(double));
        partials = (double*)workspace;
    } else {
        zero_async(sums, (size_t)2 * C * sizeof(double), (hipStream_t)stream);
    }
#define ENERF_CS(SAME, MASK) ENERF_LAUNCH((k_channel_sums<SAME, MASK>), (unsigned)blocks, 256, 0, (hipStream_t)stream, a, b, z_mask, mask_scale, mask_shift, n, C, sums, partials)
    if (a == b) { if (z_mask) ENERF_CS(true, true); else ENERF_CS(true, false); }
    else { if (z_mask) ENERF_CS(false, true); else ENERF_CS(false, false); }
#undef ENERF_CS
    if (partials != nullptr) ENERF_LAUNCH(k_channel_sums_finish, 1, 1024, 0, (hipStream_t)stream, partials, (int)blocks, 2 * C, sums);
    return check_launch("channel_sums");
}
int enerf_channel_sums(const float* a, const float* b, const float* z_mask, const float* mask_scale, const float* mask_shift,
                       long long n, int C, double* sums, enerf_stream_t stream) {
    return enerf_channel_sums_ws(a, b, z_mask, mask_scale, mask_shift, n, C, sums, nullptr, 0, stream);
}
int enerf_bn_train_coeffs(const double* sums, const double* count_dev, double count_host, const float* gamma, const float* beta,
                          double eps, double momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                          int increment_num_batches_tracked, int C, double* mean_invstd, float* scale_shift, enerf_stream_t stream) {
    REQUIRE(sums && gamma && beta && mean_invstd && scale_shift && C >= 1 && C <= 256, "bn_train_coeffs: bad arguments (C in 1..256)");
    REQUIRE(count_dev || count_host > 0.0, "bn_train_coeffs: no position count");
    REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_coeffs: running_mean and running_var come together");
    ENERF_LAUNCH(k_bn_train_coeffs, 1, 256, 0, (hipStream_t)stream, sums, count_dev, count_host, gamma, beta, eps, momentum,
                        running_mean, running_var, num_batches_tracked, increment_num_batches_tracked, C, mean_invstd, scale_shift);
    return check_launch("bn_train_coeffs");
}
int enerf_bn_train_bwd_coeffs(const double* sums_local, const double* sums_global, const double* count_dev, double count_host,
                              const double* mean_invstd, const float* scale, int C, float* dgamma_dbeta, float* k2k3,
                              enerf_stream_t stream) {
    REQUIRE(sums_local && sums_global && mean_invstd && scale && dgamma_dbeta && k2k3 && C >= 1 && C <= 256,
            "bn_train_bwd_coeffs: bad arguments (C in 1..256)");
    REQUIRE(count_dev || count_host > 0.0, "bn_train_bwd_coeffs: no position count");
    ENERF_LAUNCH_SIMPLE(k_bn_train_bwd_coeffs, 1, 256, 0, (hipStream_t)stream, sums_local, sums_global, count_dev, count_host,
                        mean_invstd, scale, C, dgamma_dbeta, k2k3);
    return check_launch("bn_train_bwd_coeffs");
}
int enerf_bn_train_stats(const float* z, long long n, int C, void* workspace, size_t workspace_bytes, const float* gamma,
                         const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                         long long* num_batches_tracked, int increment_num_batches_tracked, double* sums_out, double* mean_invstd,
                         float* scale_shift, enerf_stream_t stream) {
    REQUIRE(z && gamma && beta && mean_invstd && scale_shift && workspace && n > 0 && C >= 4 && C <= 64 && C % 4 == 0 && (256 % (C / 4)) == 0,
            "bn_train_stats: bad arguments (C in 4..64, power-of-two quads)");
    REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_stats: running_mean and running_var come together");
    const long long blocks = channel_sums_blocks(n, C);
    REQUIRE(workspace_bytes >= (size_t)blocks * 2 * C * sizeof(double), "bn_train_stats: workspace of %zu bytes, need %zu", workspace_bytes,
            (size_t)blocks * 2 * C * sizeof(double));
    double* partials = (double*)workspace;
    ENERF_LAUNCH((k_channel_sums<true, false>), (unsigned)blocks, 256, 0, (hipStream_t)stream, z, z, nullptr, nullptr, nullptr, n, C,
                 (double*)nullptr, partials);
    ENERF_LAUNCH(k_bn_finish_coeffs, 1, 1024, 0, (hipStream_t)stream, partials, (int)blocks, (double)n, gamma, beta, eps, momentum,
                 running_mean, running_var, num_batches_tracked, increment_num_batches_tracked, C, sums_out, mean_invstd, scale_shift);
    return check_launch("bn_train_stats");
}
int enerf_bn_train_bwd_stats(const float* g, const float* z, const float* z_mask, const float* mask_scale, const float* mask_shift,
                             long long n, int C, void* workspace, size_t workspace_bytes, const double* mean_invstd,
                             const float* scale, float* dgamma_dbeta, float* k2k3, enerf_stream_t stream) {
    REQUIRE(g && z && mean_invstd && scale && dgamma_dbeta && k2k3 && workspace && n > 0 && C >= 4 && C <= 64 && C % 4 == 0 &&
                (256 % (C / 4)) == 0, "bn_train_bwd_stats: bad arguments (C in 4..64, power-of-two quads)");
    if (z_mask) REQUIRE(mask_scale && mask_shift, "bn_train_bwd_stats: mask needs its scale/shift");
    const long long blocks = channel_sums_blocks(n, C);
    REQUIRE(workspace_bytes >= (size_t)blocks * 2 * C * sizeof(double), "bn_train_bwd_stats: workspace of %zu bytes, need %zu",
            workspace_bytes, (size_t)blocks * 2 * C * sizeof(double));
    double* partials = (double*)workspace;
    if (z_mask) ENERF_LAUNCH((k_channel_sums<false, true>), (unsigned)blocks, 256, 0, (hipStream_t)stream, g, z, z_mask, mask_scale, mask_shift,
                             n, C, (double*)nullptr, partials);
    else ENERF_LAUNCH((k_channel_sums<false, false>), (unsigned)blocks, 256, 0, (hipStream_t)stream, g, z, z_mask, mask_scale, mask_shift, n, C,
                      (double*)nullptr, partials);
    ENERF_LAUNCH(k_bn_finish_bwd_coeffs, 1, 1024, 0, (hipStream_t)stream, partials, (int)blocks, (double)n, mean_invstd, scale, C, dgamma_dbeta,
                 k2k3);
    return check_launch("bn_train_bwd_stats");
}
// the affine launch of the fused forms: at most 512 blocks (each redoes the row reduction)
static unsigned bn_affine_blocks(long long n4) {
    const long long b = cdivl(n4, 256);
    return (unsigned)(b > 512 ? 512 : b);
}
static bool bn_fold_fits(long long blocks, int C) { return blocks * C <= 2048; }
int enerf_bn_train_apply(const float* z, long long n, int C, void* workspace, size_t workspace_bytes, const float* gamma, const float* beta,
                         double eps, double momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                         int increment_num_batches_tracked, double* mean_invstd, float* scale_shift, const float* residual, int relu,
                         float* out, enerf_stream_t stream) {
    REQUIRE(z && gamma && beta && mean_invstd && scale_shift && out && workspace && n > 0 && C >= 4 && C <= 64 && C % 4 == 0 && (256 % (C / 4)) == 0,
            "bn_train_apply: bad arguments (C in 4..64, power-of-two quads)");
    REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_apply: running_mean and running_var come together");
    const long long blocks = channel_sums_blocks(n, C);
    REQUIRE(workspace_bytes >= (size_t)blocks * 2 * C * sizeof(double), "bn_train_apply: workspace of %zu bytes, need %zu", workspace_bytes,
            (size_t)blocks * 2 * C * sizeof(double));
    double* partials = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const long long n4 = n * (C / 4);
    ENERF_LAUNCH((k_channel_sums<true, false>), (unsigned)blocks, 256, 0, st, z, z, nullptr, nullptr, nullptr, n, C, (double*)nullptr, partials);
    if (bn_fold_fits(blocks, C)) {
        ENERF_LAUNCH(k_bn_affine_fwd, bn_affine_blocks(n4), 256, 0, st, z, partials, (int)blocks, (double)n, gamma, beta, eps, momentum, running_mean,
                     running_var, num_batches_tracked, increment_num_batches_tracked, C, (double*)nullptr, mean_invstd, scale_shift, residual, relu, n4, out);
    } else {
        ENERF_LAUNCH(k_bn_finish_coeffs, 1, 1024, 0, st, partials, (int)blocks, (double)n, gamma, beta, eps, momentum, running_mean, running_var,
                     num_batches_tracked, increment_num_batches_tracked, C, (double*)nullptr, mean_invstd, scale_shift);
        ENERF_LAUNCH_SIMPLE(k_channel_affine, (unsigned)cdivl(n4, 256), 256, 0, st, z, nullptr, scale_shift, nullptr, scale_shift + C, nullptr, nullptr,
                            nullptr, residual, relu, n4, C / 4, out);
    }
    return check_launch("bn_train_apply");
}
int enerf_bn_train_bwd_apply(const float* g, const float* z, int relu, long long n, int C, void* workspace, size_t workspace_bytes,
                             const double* mean_invstd, const float* scale_shift, float* dgamma_dbeta, float* grad_z, enerf_stream_t stream) {
    REQUIRE(g && z && mean_invstd && scale_shift && dgamma_dbeta && grad_z && workspace && n > 0 && C >= 4 && C <= 64 && C % 4 == 0 &&
                (256 % (C / 4)) == 0, "bn_train_bwd_apply: bad arguments (C in 4..64, power-of-two quads)");
    const long long blocks = channel_sums_blocks(n, C);
    REQUIRE(workspace_bytes >= (size_t)(blocks * 2 * C) * sizeof(double) + (size_t)2 * C * sizeof(float),
            "bn_train_bwd_apply: workspace of %zu bytes, need %zu", workspace_bytes, (size_t)(blocks * 2 * C) * sizeof(double) + (size_t)2 * C * sizeof(float));
    double* partials = (double*)workspace;
    float* k2k3 = (float*)(partials + blocks * 2 * C);
    hipStream_t st = (hipStream_t)stream;
    const long long n4 = n * (C / 4);
    const float *scale = scale_shift, *shift = scale_shift + C;
    if (relu) ENERF_LAUNCH((k_channel_sums<false, true>), (unsigned)blocks, 256, 0, st, g, z, z, scale, shift, n, C, (double*)nullptr, partials);
    else ENERF_LAUNCH((k_channel_sums<false, false>), (unsigned)blocks, 256, 0, st, g, z, (const float*)nullptr, (const float*)nullptr,
                      (const float*)nullptr, n, C, (double*)nullptr, partials);
    if (bn_fold_fits(blocks, C)) {
        ENERF_LAUNCH(k_bn_affine_bwd, bn_affine_blocks(n4), 256, 0, st, g, z, partials, (int)blocks, (double)n, mean_invstd, scale_shift, relu, C,
                     dgamma_dbeta, n4, grad_z);
    } else {
        ENERF_LAUNCH(k_bn_finish_bwd_coeffs, 1, 1024, 0, st, partials, (int)blocks, (double)n, mean_invstd, scale, C, dgamma_dbeta, k2k3);
        ENERF_LAUNCH_SIMPLE(k_channel_affine, (unsigned)cdivl(n4, 256), 256, 0, st, g, z, scale, k2k3, k2k3 + C, relu ? z : nullptr, scale, shift,
                            nullptr, 0, n4, C / 4, grad_z);
    }
    return check_launch("bn_train_bwd_apply");
}
int enerf_channel_affine(const float* a, const float* b, const float* p, const float* q, const float* r, const float* z_mask,
                         const float* mask_scale, const float* mask_shift, const float* residual, int relu, long long n, int C,
                         float* out, enerf_stream_t stream) {
    REQUIRE(a && p && r && out && n > 0 && C >= 4 && C % 4 == 0, "channel_affine: bad arguments");
    if (b) REQUIRE(q, "channel_affine: b needs q");
    if (z_mask) REQUIRE(mask_scale && mask_shift, "channel_affine: mask needs its scale/shift");
    const long long n4 = n * (C / 4);
    ENERF_LAUNCH_SIMPLE(k_channel_affine, (unsigned)cdivl(n4, 256), 256, 0, (hipStream_t)stream, a, b, p, q, r, z_mask, mask_scale,
                        mask_shift, residual, relu, n4, C / 4, out);
    return check_launch("channel_affine");
}

}  // extern "C"
