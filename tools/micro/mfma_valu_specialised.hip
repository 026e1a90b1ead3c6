// Microbenchmark (dev tool, round 5): what is the CEILING of wave specialisation for the render kernel on gfx950?
// The exact-fp32 kernel issues, per 16 samples, 201 v_mfma_f32_16x16x4_f32 and ~1078 VALU instructions from every wave (three waves
// per SIMD).  Specialised, the same SIMD would run MFMA-only waves fed by VALU-only "gather" waves through an LDS record ring.  This
// measures only the issue side of that idea — no LDS ring, no hand-off stalls, i.e. its upper bound — at the kernel's own ratio:
//   mixed:        3 waves per SIMD, each: (1 MFMA + NV VALU) x n
//   specialised:  3 waves per SIMD doing the SAME total work: 2 MFMA-only waves (1.5 n MFMAs each) + 1 VALU-only wave (3 n NV VALU)
// NV = 5 and 6 bracket the kernel's 1078 / 201 = 5.4 VALU per MFMA.
// hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma_valu_specialised.hip -o tools/micro/mfma_valu_specialised.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// 768 threads = 12 waves = 3 per SIMD; wave w sits on SIMD w % 4 (round-robin), so waves {w, w+4, w+8} share a SIMD
template <int MODE, int NV>
__global__ __launch_bounds__(768) void k(float* out, int n, float seed) {
    const int slot = (threadIdx.x >> 6) >> 2;             // 0, 1, 2: which of the SIMD's three waves
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed * 0.5f, b = seed * 0.25f;
    if (MODE == 0) {                                       // mixed
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[u & 3] = MFMA(a, b, acc[u & 3]);
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
            }
        }
    } else if (slot < 2) {                                 // specialised: MFMA-only wave, 1.5 x the MFMAs
        for (int it = 0; it < n + n / 2; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] = MFMA(a, b, acc[u & 3]);
        }
    } else {                                               // specialised: VALU-only wave, 3 x the VALU work
        for (int it = 0; it < 3 * n; ++it) {
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
float run(int n, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(768), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(768), 0, 0, d, n, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 768 * 4);
    const int n = 10000;
    for (int rep = 0; rep < 2; ++rep) {
        const float m5 = run<0, 5>(n, d), s5 = run<1, 5>(n, d), m6 = run<0, 6>(n, d), s6 = run<1, 6>(n, d);
        printf("3 waves/SIMD, same total work: NV=5 mixed %.3f ms, specialised %.3f ms (%.1f %%) | NV=6 mixed %.3f ms, specialised %.3f ms (%.1f %%)\n",
               m5, s5, 100.0 * (s5 - m5) / m5, m6, s6, 100.0 * (s6 - m6) / m6);
    }
    return 0;
}
