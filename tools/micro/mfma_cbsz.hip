// Microbenchmark (dev tool, round 5): the A-operand broadcast controls of the batched 4x4x1 matrix instruction on gfx950.
// v_mfma_f32_4x4x1_16b_f32 ... cbsz:4 abid:K takes the 4x1 A column of BLOCK K (lanes 4K..4K+3) for all 16 blocks, so ONE VGPR
// holds 16 different weight columns and a Cout = 8 convolution needs no LDS (or global) read per A operand at all.
// Checks (1) the semantics for every K, (2) the issue time against the plain form.
// hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma_cbsz.hip -o tools/micro/mfma_cbsz.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ f32x4 one(float a, float b) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, f32x4{0.f, 0.f, 0.f, 0.f}, 4, K, 0);
}
__global__ void k_sem(float* out) {          // out[K][lane][4]
    const int lane = threadIdx.x;
    const float a = (float)(1000 + lane), b = (float)(lane + 1) * 0.5f;
    f32x4 r[16] = {one<0>(a, b), one<1>(a, b), one<2>(a, b), one<3>(a, b), one<4>(a, b), one<5>(a, b), one<6>(a, b), one<7>(a, b),
                   one<8>(a, b), one<9>(a, b), one<10>(a, b), one<11>(a, b), one<12>(a, b), one<13>(a, b), one<14>(a, b), one<15>(a, b)};
    for (int k = 0; k < 16; ++k)
        for (int i = 0; i < 4; ++i) out[(k * 64 + lane) * 4 + i] = r[k][i];
}
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float seed) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{(float)i, 1, 2, 3};
    const float a = seed * 0.5f + threadIdx.x, b = seed * 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u & 3], 0, 0, 0);
            else if (u == 0) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 4, 0, 0);
            else if (u == 1) acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[1], 4, 1, 0);
            else if (u == 2) acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[2], 4, 2, 0);
            else if (u == 3) acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[3], 4, 3, 0);
            else if (u == 4) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 4, 4, 0);
            else if (u == 5) acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[1], 4, 5, 0);
            else if (u == 6) acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[2], 4, 6, 0);
            else if (u == 7) acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[3], 4, 7, 0);
            else if (u == 8) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 4, 8, 0);
            else if (u == 9) acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[1], 4, 9, 0);
            else if (u == 10) acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[2], 4, 10, 0);
            else if (u == 11) acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[3], 4, 11, 0);
            else if (u == 12) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 4, 12, 0);
            else if (u == 13) acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[1], 4, 13, 0);
            else if (u == 14) acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[2], 4, 14, 0);
            else acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[3], 4, 15, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
float run(int blocks, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<MODE>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d);
    static float h[16 * 64 * 4];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k = 0; k < 16; ++k)
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 4; ++i) {
                const float want = (float)(1000 + 4 * k + i) * ((float)(lane + 1) * 0.5f), got = h[(k * 64 + lane) * 4 + i];
                if (want != got) { if (bad < 8) printf("  K=%d lane=%d row=%d: got %g (= A-lane %g), want %g\n", k, lane, i, got, got / ((lane + 1) * 0.5f) - 1000, want); ++bad; }
            }
    printf("cbsz:4 abid:K semantics (D[r] of every lane = A[lane 4K + r] * B[own lane]): %s (%d mismatches of 4096)\n", bad ? "DIFFERENT" : "as expected", bad);
    const int iters = 20000, CU = 256;
    const double n = iters * 16.0;
    for (int waves = 1; waves <= 3; ++waves) {
        const int blocks = CU * waves;
        const float t0 = run<0>(blocks, iters, d), t1 = run<1>(blocks, iters, d);
        printf("%d wave(s)/SIMD: 4x4x1 plain %.2f ns/instr | cbsz:4 abid:0..15 %.2f ns/instr\n", waves, t0 * 1e6 / n / waves, t1 * 1e6 / n / waves);
    }
    return 0;
}
