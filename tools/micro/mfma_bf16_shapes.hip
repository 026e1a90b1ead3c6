// Microbenchmark: issue rate of the bf16 MFMA shapes on gfx950 — does the CDNA3-era v_mfma_f32_16x16x16_bf16 (K = 16) run at the
// chip's bf16 peak, or only the new K = 32 form (v_mfma_f32_16x16x32_bf16)?  Decides the operand width of the bf16x3/x6 layers.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_bf16_shapes.hip -o /tmp/bsh && /tmp/bsh
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{(float)i, 1, 2, 3};
    const float a = seed * 0.5f + threadIdx.x, b = seed * 0.25f;
    bf16x4 a4, b4; bf16x8 a8, b8;
    for (int q = 0; q < 4; ++q) { a4[q] = (__bf16)(a + q); b4[q] = (__bf16)(b + q); }
    for (int q = 0; q < 8; ++q) { a8[q] = (__bf16)(a + q); b8[q] = (__bf16)(b + q); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (SHAPE == 0) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NACC], 0, 0, 0);
            else if (SHAPE == 1) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[u % NACC], 0, 0, 0);
            else acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[u % NACC], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE, int NACC>
float run(int blocks, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
    const int iters = 20000, CU = 256;
    const double n = iters * 16.0;
    for (int waves = 1; waves <= 3; ++waves) {
        const int blocks = CU * waves;
        printf("%d wave(s)/SIMD, ns per instruction (4 acc | 2 acc | 1 acc): f32 16x16x4 %.1f | %.1f | %.1f    bf16 16x16x16 %.1f | %.1f | %.1f    bf16 16x16x32 %.1f | %.1f | %.1f\n", waves,
               run<0, 4>(blocks, iters, d) * 1e6 / n / waves, run<0, 2>(blocks, iters, d) * 1e6 / n / waves, run<0, 1>(blocks, iters, d) * 1e6 / n / waves,
               run<1, 4>(blocks, iters, d) * 1e6 / n / waves, run<1, 2>(blocks, iters, d) * 1e6 / n / waves, run<1, 1>(blocks, iters, d) * 1e6 / n / waves,
               run<2, 4>(blocks, iters, d) * 1e6 / n / waves, run<2, 2>(blocks, iters, d) * 1e6 / n / waves, run<2, 1>(blocks, iters, d) * 1e6 / n / waves);
    }
    return 0;
}
