// Micro-benchmark: what does a device-wide barrier inside a persistent kernel cost on MI355X, against the ~5 us gap between two
// dependent kernel launches?  ("fuse the small cost-reg layers into one launch" only pays if the barrier is well below
// the launch gap.)  Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_barriers(unsigned* ctr, float* data, int iters, int touch) {
    const unsigned nblk = gridDim.x;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        // a little "layer" work: every block writes and later reads other blocks' data (so the fences have something to order)
        if (touch) data[(blockIdx.x * 256 + threadIdx.x)] = acc + (float)i;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                   // release: make this block's writes visible device-wide (all XCDs)
            atomicAdd(ctr, 1u);
            const unsigned target = (unsigned)(i + 1) * nblk;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();                                   // acquire
        }
        __syncthreads();
        if (touch) acc += data[(((blockIdx.x + 7) % nblk) * 256 + threadIdx.x)];
    }
    if (acc == 12345.f) data[0] = acc;
}
__global__ void k_tiny(float* d, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) d[0] += (float)i; }

int main() {
    unsigned* ctr; float* data;
    hipMalloc(&ctr, 4); hipMalloc(&data, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    for (int touch = 0; touch < 2; ++touch)
        for (int g : {64, 256, 512, 1024}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(ctr, 0, 4);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_barriers, dim3(g), dim3(256), 0, 0, ctr, data, iters, touch);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("grid barrier: %4d blocks x 256, touch=%d: %.2f us per barrier\n", g, touch, 1e3f * best / iters);
        }
    // dependent tiny launches on one stream: the launch-to-launch floor
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, 0, data, i);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("dependent launches: %.2f us per (tiny) kernel on one stream\n", 1e3f * best / iters);
    return 0;
}
