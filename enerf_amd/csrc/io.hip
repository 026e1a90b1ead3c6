// io.hip — the steps either side of the rendering path, on device:
//   before: full-image ray generation (lib/datasets/enerf_utils.py:61-71, numpy on the host today; 10.5 MB of
//           rays_1 per 512x640 frame would otherwise cross PCIe every frame)
//   after:  uint8 packing + vertical flip for presentation (gui_human.py:88-91) and the evaluator's masked
//           PSNR / depth statistics (lib/evaluators/enerf.py:67-71, 88-103) without a D2H copy of fp32 images.
// All HBM-bound, one thread per output element.
#include "kernels.h"

namespace enerf {

// -------------------------------------------------------------------------------------------------
// rays[b][y*W+x] = [o(3) | d(3) | x | y],  o = c2w[:3,3],  d = c2w[:3,:3] · inv(K') · [x,y,1]^T,
// K' = K with rows 0,1 scaled by `scale`, c2w = inv(tar_ext).  The reference does this in float64 numpy
// and casts to float32 (enerf_utils.py:61-71); lane 0 of each block builds the 3x3 in fp64 in LDS.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gen_rays(const float* __restrict__ tar_ext, const float* __restrict__ tar_ixt,
                                                  int B, int Hr, int Wr, float scale, float* __restrict__ rays) {
    __shared__ double M[12];       // 3x3 (c2w_R · K'^-1) | origin(3)
    const long long npix = (long long)Hr * Wr;
    const int blocks_per_img = (int)cdivl(npix, 256);
    const int b = blockIdx.x / blocks_per_img;
    const long long p = (long long)(blockIdx.x - b * blocks_per_img) * 256 + threadIdx.x;
    if (threadIdx.x == 0) {
        double e[16], ei[16];
        for (int k = 0; k < 16; ++k) e[k] = (double)tar_ext[b * 16 + k];
        bool ok = inv4x4(e, ei);
        const float* K = tar_ixt + b * 9;
        // inverse of the scaled intrinsics (general 3x3, cofactors)
        double k[9];
        for (int i = 0; i < 9; ++i) k[i] = (double)K[i] * (i < 6 ? (double)scale : 1.0);
        double c00 = k[4] * k[8] - k[5] * k[7], c01 = k[5] * k[6] - k[3] * k[8], c02 = k[3] * k[7] - k[4] * k[6];
        double det = k[0] * c00 + k[1] * c01 + k[2] * c02;
        double ki[9] = {c00, k[2] * k[7] - k[1] * k[8], k[1] * k[5] - k[2] * k[4],
                        c01, k[0] * k[8] - k[2] * k[6], k[2] * k[3] - k[0] * k[5],
                        c02, k[1] * k[6] - k[0] * k[7], k[0] * k[4] - k[1] * k[3]};
        for (int i = 0; i < 9; ++i) ki[i] = (ok && det != 0.0) ? ki[i] / det : NAN;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double a = 0;
                for (int t = 0; t < 3; ++t) a += ei[r * 4 + t] * ki[t * 3 + c];
                M[r * 3 + c] = a;
            }
        M[9] = ei[3]; M[10] = ei[7]; M[11] = ei[11];
    }
    __syncthreads();
    if (p >= npix) return;
    const int y = (int)(p / Wr), x = (int)(p - (long long)y * Wr);
    float* o = rays + ((long long)b * npix + p) * 8;
    const double fx = (double)x, fy = (double)y;
    const float4 a = make_float4((float)M[9], (float)M[10], (float)M[11], (float)(M[0] * fx + M[1] * fy + M[2]));
    const float4 c = make_float4((float)(M[3] * fx + M[4] * fy + M[5]), (float)(M[6] * fx + M[7] * fy + M[8]), (float)x,
                                 (float)y);
    *reinterpret_cast<float4*>(o) = a;
    *reinterpret_cast<float4*>(o + 4) = c;
}
void launch_gen_rays(const float* tar_ext, const float* tar_ixt, int B, int Hr, int Wr, float scale, float* rays,
                     hipStream_t st) {
    const unsigned grid = (unsigned)(cdivl((long long)Hr * Wr, 256) * B);
    ENERF_LAUNCH(k_gen_rays, grid, 256, 0, st, tar_ext, tar_ixt, B, Hr, Wr, scale, rays);
}

// -------------------------------------------------------------------------------------------------
// The training branch of build_rays (enerf_utils.py:33-56): the host RNG picks the pixel list (X, Y) exactly as the
// reference does (np.random permutation / randint / patches); the rays of those pixels are built here:
// rays[b][n] = [o | c2w_R · inv(K') · [X,Y,1]^T | X | Y].  Same fp64 3x3 as k_gen_rays.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ray_matrix(const float* tar_ext, const float* tar_ixt, int b, float scale, double* M) {
    double e[16], ei[16];
    for (int k = 0; k < 16; ++k) e[k] = (double)tar_ext[b * 16 + k];
    bool ok = inv4x4(e, ei);
    const float* K = tar_ixt + b * 9;
    double k[9];
    for (int i = 0; i < 9; ++i) k[i] = (double)K[i] * (i < 6 ? (double)scale : 1.0);
    double c00 = k[4] * k[8] - k[5] * k[7], c01 = k[5] * k[6] - k[3] * k[8], c02 = k[3] * k[7] - k[4] * k[6];
    double det = k[0] * c00 + k[1] * c01 + k[2] * c02;
    double ki[9] = {c00, k[2] * k[7] - k[1] * k[8], k[1] * k[5] - k[2] * k[4],
                    c01, k[0] * k[8] - k[2] * k[6], k[2] * k[3] - k[0] * k[5],
                    c02, k[1] * k[6] - k[0] * k[7], k[0] * k[4] - k[1] * k[3]};
    for (int i = 0; i < 9; ++i) ki[i] = (ok && det != 0.0) ? ki[i] / det : NAN;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double a = 0;
            for (int t = 0; t < 3; ++t) a += ei[r * 4 + t] * ki[t * 3 + c];
            M[r * 3 + c] = a;
        }
    M[9] = ei[3]; M[10] = ei[7]; M[11] = ei[11];
}
__global__ __launch_bounds__(256) void k_gen_rays_at(const float* __restrict__ tar_ext, const float* __restrict__ tar_ixt,
                                                     const int* __restrict__ xy, int B, int N, float scale,
                                                     float* __restrict__ rays) {
    __shared__ double M[12];
    const int blocks_per_b = cdiv(N, 256);
    const int b = blockIdx.x / blocks_per_b;
    const int n = (blockIdx.x - b * blocks_per_b) * 256 + threadIdx.x;
    if (threadIdx.x == 0) ray_matrix(tar_ext, tar_ixt, b, scale, M);
    __syncthreads();
    if (n >= N) return;
    const int x = xy[((long long)b * N + n) * 2], y = xy[((long long)b * N + n) * 2 + 1];
    const double fx = (double)x, fy = (double)y;
    float* o = rays + ((long long)b * N + n) * 8;
    *reinterpret_cast<float4*>(o) = make_float4((float)M[9], (float)M[10], (float)M[11], (float)(M[0] * fx + M[1] * fy + M[2]));
    *reinterpret_cast<float4*>(o + 4) = make_float4((float)(M[3] * fx + M[4] * fy + M[5]), (float)(M[6] * fx + M[7] * fy + M[8]),
                                                    (float)x, (float)y);
}
void launch_gen_rays_at(const float* tar_ext, const float* tar_ixt, const int* xy, int B, int N, float scale, float* rays,
                        hipStream_t st) {
    ENERF_LAUNCH(k_gen_rays_at, (unsigned)(cdiv(N, 256) * B), 256, 0, st, tar_ext, tar_ixt, xy, B, N, scale, rays);
}

// -------------------------------------------------------------------------------------------------
// gen_rays_bbox (lib/utils/net_utils.py:13-28): slab test of every ray against the axis-aligned box `bounds` (2,3), the
// origin taken from the FIRST ray (rays_o[:1], as the reference does).  Bit-exact restatement: fp32, no fma contraction,
// the reference's direction clamps (|v| pushed away from 0 to +-1e-5) and min/max order.  mask[i] = near < far.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mul_rn(float a, float b) {
#ifdef ENERF_EMU
    return a * b;
#else
    return __fmul_rn(a, b);
#endif
}
__device__ __forceinline__ float add_rn(float a, float b) {
#ifdef ENERF_EMU
    return a + b;
#else
    return __fadd_rn(a, b);
#endif
}
__global__ __launch_bounds__(256) void k_rays_bbox_mask(const float* __restrict__ rays, const float* __restrict__ bounds,
                                                        long long n, int* __restrict__ mask) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + i * 8;
    const float dx = r[3], dy = r[4], dz = r[5];
    const float nrm = sqrtf(add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz)));     // torch.norm(dim=-1)
    float v[3] = {dx / nrm, dy / nrm, dz / nrm};
    float near = -INFINITY, far = INFINITY;
    for (int c = 0; c < 3; ++c) {
        if (v[c] < 1e-5f && v[c] > -1e-10f) v[c] = 1e-5f;
        if (v[c] > -1e-5f && v[c] < 1e-10f) v[c] = -1e-5f;
        const float o = rays[c];                                       // rays_o[:1]
        const float t0 = add_rn(bounds[c], -o) / v[c], t1 = add_rn(bounds[3 + c], -o) / v[c];
        near = fmaxf(near, fminf(t0, t1));
        far = fminf(far, fmaxf(t0, t1));
    }
    mask[i] = near < far ? 1 : 0;
}
void launch_rays_bbox_mask(const float* rays, const float* bounds, long long n, int* mask, hipStream_t st) {
    ENERF_LAUNCH_SIMPLE(k_rays_bbox_mask, (unsigned)cdivl(n, 256), 256, 0, st, rays, bounds, n, mask);
}

// -------------------------------------------------------------------------------------------------
// Nearest-view selection of the interactive dataset (zjumocap/enerf_interactive.py:207-210):
//   distances = ||cam_points - c2w[:3,3]||; near_views = argsort(distances)[:k]
// and the gather of the selected source views into the batch (:214-217): inps (V,H,W,3) -> src_inps (k,3,H,W),
// exts (V,4,4) / ixts (V,3,3) -> (k,...).  One block; V <= 1024 cameras, k <= 8.  Ties resolve to the lower index.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_select_views(const float* __restrict__ cam_points, int V,
                                                      const float* __restrict__ c2w, int k, int* __restrict__ idx) {
    __shared__ double dist[1024];
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        double s = 0;
        for (int c = 0; c < 3; ++c) { const double d = (double)cam_points[v * 3 + c] - (double)c2w[c * 4 + 3]; s += d * d; }
        dist[v] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int j = 0; j < k; ++j) {
            int best = 0;
            for (int v = 1; v < V; ++v) if (dist[v] < dist[best]) best = v;
            idx[j] = best;
            dist[best] = INFINITY;
        }
}
__global__ __launch_bounds__(256) void k_gather_views(const float* __restrict__ inps, const float* __restrict__ exts,
                                                      const float* __restrict__ ixts, const int* __restrict__ idx, int k,
                                                      int H, int W, float* __restrict__ src_inps,
                                                      float* __restrict__ src_exts, float* __restrict__ src_ixts) {
    const long long hw = (long long)H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;        // over k*H*W pixels
    if (i < (long long)k * 16) src_exts[i] = exts[(long long)idx[i / 16] * 16 + i % 16];
    if (i < (long long)k * 9) src_ixts[i] = ixts[(long long)idx[i / 9] * 9 + i % 9];
    if (i >= k * hw) return;
    const int j = (int)(i / hw);
    const long long p = i - j * hw;
    const float* s = inps + ((long long)idx[j] * hw + p) * 3;
    float* d = src_inps + (long long)j * 3 * hw + p;
    d[0] = s[0]; d[hw] = s[1]; d[2 * hw] = s[2];                         // permute(0,3,1,2)
}
void launch_select_views(const float* cam_points, int V, const float* c2w, int k, int* idx, hipStream_t st) {
    ENERF_LAUNCH(k_select_views, 1u, 256, 0, st, cam_points, V, c2w, k, idx);
}
void launch_gather_views(const float* inps, const float* exts, const float* ixts, const int* idx, int k, int H, int W,
                         float* src_inps, float* src_exts, float* src_ixts, hipStream_t st) {
    ENERF_LAUNCH_SIMPLE(k_gather_views, (unsigned)cdivl((long long)k * H * W, 256), 256, 0, st, inps, exts, ixts, idx, k, H, W,
                        src_inps, src_exts, src_ixts);
}

// -------------------------------------------------------------------------------------------------
// gui_human.py:88-91:  img *= 255; img.to(uint8); flip(0)   — rgb (H*W,3) float -> (H,W,3) uint8
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_rgb8(const float* __restrict__ rgb, int H, int W, int flip,
                                                   unsigned char* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per pixel
    if (i >= (long long)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (long long)y * W);
    const int yo = flip ? H - 1 - y : y;
    const float* s = rgb + i * 3;
    unsigned char* d = out + ((long long)yo * W + x) * 3;
    for (int c = 0; c < 3; ++c) {
        float v = s[c] * 255.f;
        // bit-exact to torch's `img *= 255; img.to(torch.uint8)` on [0,1] inputs (fp32 multiply, truncation).  Outside
        // [0,1] the reference's C cast is undefined; policy here: saturate (negative / NaN -> 0, > 255 -> 255).
        v = v > 0.f ? (v > 255.f ? 255.f : v) : 0.f;
        d[c] = (unsigned char)(int)v;
    }
}
void launch_pack_rgb8(const float* rgb, int H, int W, int flip, unsigned char* out, hipStream_t st) {
    ENERF_LAUNCH_SIMPLE(k_pack_rgb8, (unsigned)cdivl((long long)H * W, 256), 256, 0, st, rgb, H, W, flip, out);
}

// -------------------------------------------------------------------------------------------------
// Evaluator statistics (evaluators/enerf.py:67-71, 88-103), accumulated into 6 doubles:
//   acc[0] = sum of squared rgb errors over masked pixels (3 channels), acc[1] = their count
//   acc[2] = sum |depth - gt| over gt != 0, acc[3] = count, acc[4] = #(<2), acc[5] = #(<10)
// psnr = 10 log10(acc[1] / acc[0]); abs = acc[2]/acc[3]; acc_2 = acc[4]/acc[3]; acc_10 = acc[5]/acc[3].
// The caller zeroes acc (hipMemsetAsync) before the launch.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    for (int m = 32; m >= 1; m >>= 1) {
        float lo = __int_as_float((int)(__double_as_longlong(v) & 0xffffffffll));
        float hi = __int_as_float((int)(__double_as_longlong(v) >> 32));
        lo = __shfl_xor(lo, m);
        hi = __shfl_xor(hi, m);
        v += __longlong_as_double(((long long)__float_as_int(hi) << 32) | (unsigned int)__float_as_int(lo));
    }
    return v;
}
__global__ __launch_bounds__(256) void k_eval_stats(const float* __restrict__ pred_rgb, const float* __restrict__ gt_rgb,
                                                    const unsigned char* __restrict__ mask, int mask_bytes, long long n_rgb,
                                                    int img_w, int img_h, int crop_h, int crop_w,
                                                    const float* __restrict__ pred_depth,
                                                    const float* __restrict__ gt_depth, long long n_depth,
                                                    double* __restrict__ acc) {
    double a[6] = {0, 0, 0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rgb; i += stride) {
        bool on = true;
        if (mask != nullptr)                                            // masks = msk >= 1 (evaluators/enerf.py:48)
            on = mask_bytes == 4 ? reinterpret_cast<const int*>(mask)[i] >= 1 : mask[i] >= 1;
        if (img_w > 0) {                                                // eval_center: [crop:-crop] on both axes (:50-54)
            const long long p = i % ((long long)img_w * img_h);
            const int y = (int)(p / img_w), x = (int)(p - (long long)y * img_w);
            on = on && y >= crop_h && y < img_h - crop_h && x >= crop_w && x < img_w - crop_w;
        }
        if (on) {
            for (int c = 0; c < 3; ++c) {                               // skimage psnr: float64 mse
                const double d = (double)pred_rgb[i * 3 + c] - (double)gt_rgb[i * 3 + c];
                a[0] += d * d;
            }
            a[1] += 3.0;
        }
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_depth; i += stride) {
        const float g = gt_depth[i];
        if (g != 0.f) {
            const float e = fabsf(pred_depth[i] - g);                   // np.abs(float32 - float32): float32 (:96-98)
            a[2] += (double)e; a[3] += 1.0; a[4] += e < 2.f ? 1.0 : 0.0; a[5] += e < 10.f ? 1.0 : 0.0;
        }
    }
    for (int k = 0; k < 6; ++k) {
        const double s = wave_sum(a[k]);
        if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(acc + k, s);
    }
}
void launch_eval_stats(const float* pred_rgb, const float* gt_rgb, const void* mask, int mask_bytes, long long n_rgb,
                       int img_w, int img_h, int crop_h, int crop_w, const float* pred_depth, const float* gt_depth,
                       long long n_depth, double* acc, hipStream_t st) {
    long long n = n_rgb > n_depth ? n_rgb : n_depth;
    long long blocks = cdivl(n, 256);
    unsigned grid = (unsigned)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024);
    ENERF_LAUNCH(k_eval_stats, grid, 256, 0, st, pred_rgb, gt_rgb, (const unsigned char*)mask, mask_bytes, n_rgb, img_w, img_h,
                 crop_h, crop_w, pred_depth, gt_depth, n_depth, acc);
}

}  // namespace enerf
