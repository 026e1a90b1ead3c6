// Microbenchmark (dev tool): can the VALU issue in the shadow of v_mfma_f32_16x16x4_f32 on gfx950?
//   mode 0: MFMA only (4 independent accumulators, back to back)         -> matrix-pipe rate
//   mode 1: VALU only (NV independent v_fma chains per "MFMA slot")
//   mode 2: same wave interleaves 1 MFMA + NV VALU
//   mode 3: two waves per SIMD: even waves run mode 0, odd waves run mode 1
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed * 0.5f, b = seed * 0.25f;
    const int role = (blockIdx.x >> 8) & 1;          // blocks b and b+256 land on the same CU (round-robin dispatch)
    (void)wave;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && role == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && role == 1);
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[u & 3] = MFMA(a, b, acc[u & 3]);
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] = MFMA(a, b, acc[u & 3]);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NV>
float run(int blocks, int threads, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
    const int iters = 20000;      // x16 MFMA slots
    const int CU = 256;
    // one 256-thread block per CU = 1 wave per SIMD; two blocks per CU = 2 waves per SIMD
    printf("1 wave/SIMD:  mfma %.3f ms | valu4 %.3f | valu7 %.3f | mfma+valu4 same wave %.3f | mfma+valu7 same wave %.3f\n",
           run<0, 4>(CU, 256, iters, d), run<1, 4>(CU, 256, iters, d), run<1, 7>(CU, 256, iters, d), run<2, 4>(CU, 256, iters, d), run<2, 7>(CU, 256, iters, d));
    printf("2 waves/SIMD (2 blocks per CU): mfma %.3f ms | valu4 %.3f | valu7 %.3f | mfma+valu4 same wave %.3f | one block mfma + one block valu4 %.3f, valu7 %.3f\n",
           run<0, 4>(2 * CU, 256, iters, d), run<1, 4>(2 * CU, 256, iters, d), run<1, 7>(2 * CU, 256, iters, d), run<2, 4>(2 * CU, 256, iters, d), run<3, 4>(2 * CU, 256, iters, d), run<3, 7>(2 * CU, 256, iters, d));
    printf("per MFMA slot at 1 wave/SIMD: %.1f ns (32 cycles at 2.4 GHz = 13.3 ns)\n", run<0, 4>(CU, 256, iters, d) * 1e6 / (iters * 16.0));
    return 0;
}
