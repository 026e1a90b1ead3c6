// Microbenchmark (dev tool): issue rate of the fp32 MFMA shapes on gfx950 — is v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4
// blocks, k = 1) as fast per FLOP as v_mfma_f32_16x16x4_f32?  A Cout = 8 convolution fills only 8 of the 16 rows of the
// 16x16x4 tile; two 4-row blocks of the batched 4x4x1 shape would waste nothing.
// hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma_shapes.hip -shared -fPIC -o tools/micro/mshape.so ; run via main()
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x4 acc[NACC];
    f32x16 big[2];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{(float)i, 1, 2, 3};
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 16; ++q) big[i][q] = (float)(i + q);
    const float a = seed * 0.5f + threadIdx.x, b = seed * 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (SHAPE == 0) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NACC], 0, 0, 0);
            else if (SHAPE == 1) acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u % NACC], 0, 0, 0);
            else big[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[u & 1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 16; ++q) s += big[i][q];
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
    for (int waves = 1; waves <= 2; ++waves) {
        const int blocks = CU * waves;
        const float t16 = run<0, 4>(blocks, iters, d), t4a = run<1, 4>(blocks, iters, d), t4b = run<1, 8>(blocks, iters, d),
                    t4c = run<1, 2>(blocks, iters, d), t32 = run<2, 4>(blocks, iters, d);
        printf("%d wave(s)/SIMD: 16x16x4 %.1f ns/instr (2048 flop) | 4x4x1 (4 acc) %.1f ns (512 flop) | 4x4x1 (8 acc) %.1f | 4x4x1 (2 acc) %.1f | 32x32x2 %.1f ns (4096 flop)\n",
               waves, t16 * 1e6 / n / waves, t4a * 1e6 / n / waves, t4b * 1e6 / n / waves, t4c * 1e6 / n / waves, t32 * 1e6 / n / waves);
    }
    return 0;
}
