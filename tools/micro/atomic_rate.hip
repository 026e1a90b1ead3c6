// Micro-benchmark: what bounds fp32 global atomics on MI355X — dwords or cache-line requests?  The training path's scatter
// kernels (k_feature_volume_bwd, k_gather_bwd) are atomic-bound (profiles/r04_ab_mlp_bwd_lds_noatomic.txt): the same number of
// atomic adds is issued with different lane -> address patterns over a 42 MB buffer (texels of C floats at pseudo-random
// positions, as the bilinear taps of a warped volume land).
//   pattern 0: 64 lanes = 16 texels x 4 lanes, lane q adds channel 4q + c (c = instruction index): 16-B stride inside a 64-B
//              texel — what k_feature_volume_bwd<4> issues (4 instructions per tap)
//   pattern 1: 64 lanes = 4 texels x 16 contiguous floats (one 64-B line each), 4 instructions cover the same 16 texels
//   pattern 2: 64 lanes = 64 different texels, one dword each (worst case)
//   pattern 3: 64 lanes = 1 run of 64 contiguous floats (4 adjacent lines)
//   pattern 4: as 1, but the 4 texels of an instruction are x-neighbours (adjacent lines)
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/micro/atomic_rate.hip -o /tmp/ar && /tmp/ar
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int PAT>
__global__ __launch_bounds__(256) void k_atomics(float* buf, unsigned ntex, int iters, int local) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        // 16 texels per (wave, iteration): pseudo-random, or (local) consecutive texels as neighbouring voxels produce
        const unsigned base = local ? (hash(wave * 977u + it) % (ntex - 64)) : 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned tex, ch;
            if (PAT == 0) { const unsigned v = lane >> 2; tex = local ? base + v : hash((wave * 131071u + it) * 16u + v) % ntex; ch = 4 * (lane & 3) + c; }
            else if (PAT == 1) { const unsigned v = 4 * c + (lane >> 4); tex = local ? base + v : hash((wave * 131071u + it) * 16u + v) % ntex; ch = lane & 15; }
            else if (PAT == 2) { tex = hash(((wave * 131071u + it) * 4u + c) * 64u + lane) % ntex; ch = (lane * 5 + c) & 15; }
            else if (PAT == 3) { tex = (hash((wave * 131071u + it) * 4u + c) % (ntex - 4)) + (lane >> 4); ch = lane & 15; }
            else { tex = (hash((wave * 131071u + it) * 4u + c) % (ntex - 4)) + (lane >> 4); ch = lane & 15; }
            unsafeAtomicAdd(buf + (size_t)tex * 16 + ch, 1.0f);
        }
    }
}

template <int PAT>
static void run(float* buf, unsigned ntex, const char* what, int local) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 256;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_atomics<PAT>, dim3(blocks), dim3(256), 0, 0, buf, ntex, iters, local);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double adds = (double)blocks * 256 * iters * 4, instr = adds / 64;
    printf("pattern %d local=%d (%s): %.3f ms  %.1f G adds/s  %.2f G wave-instructions/s\n", PAT, local, what, best, adds / best / 1e6, instr / best / 1e6);
}

int main() {
    const unsigned ntex = 655360;                      // 256 x 320 x 8 texels of 16 floats = 42 MB
    float* buf; hipMalloc(&buf, (size_t)ntex * 16 * 4); hipMemset(buf, 0, (size_t)ntex * 16 * 4);
    for (int local = 0; local < 2; ++local) {
        run<0>(buf, ntex, "16 texels x 4 lanes, 16-B stride (k_feature_volume_bwd<4> today)", local);
        run<1>(buf, ntex, "4 texels x 16 contiguous floats", local);
    }
    run<2>(buf, ntex, "64 texels x 1 dword", 0);
    run<3>(buf, ntex, "64 contiguous floats (4 adjacent lines)", 0);
    return 0;
}
