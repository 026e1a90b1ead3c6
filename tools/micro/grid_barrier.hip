// Micro-benchmark: what a grid-wide barrier costs on MI355X (8 XCDs, one L2 each: an agent-scope release is an L2 write-back, an
// acquire an L2 invalidate) against the kernel boundary it would replace in a chain of small dependent layers (measured: 2.4 us of
// dependent-launch gap + the ramp of a fresh grid; the small cost-regularisation layers sit at 6 - 8 us each in the frame).
// Each of G co-resident blocks writes `bytes` of its own slice, passes the barrier, then reads the slices of blocks b+1 .. b+R
// (written by other blocks, mostly on other XCDs) and checks them; `iters` rounds.
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/micro/grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: release RMW + acquire load in the poll loop (the textbook form: an L2 invalidate per poll)
// MODE 1: release fence (L2 write-back) by thread 0, relaxed polls, ONE acquire fence (L2 invalidate) per block after the loop
// MODE 2: the data is stored write-through (agent-scope relaxed atomic stores, sc1) so no write-back is needed: workgroup-scope
//         fence (s_waitcnt) + relaxed RMW, relaxed polls, one acquire fence per block
template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        } else {
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}

template <int MODE>          // -1: no barrier (no ordering)
__global__ __launch_bounds__(256) void k_rounds(float* buf, int floats_per_block, int iters, int R, unsigned* ctr, unsigned* bad) {
    const int G = gridDim.x;
    unsigned wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float* mine = buf + ((size_t)(it & 1) * G + blockIdx.x) * floats_per_block;
        for (int i = threadIdx.x; i < floats_per_block; i += 256) {
            if (MODE == 2) __hip_atomic_store(mine + i, (float)(it * 131 + blockIdx.x + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[i] = (float)(it * 131 + blockIdx.x + i);
        }
        if (MODE >= 0) grid_barrier<MODE < 0 ? 0 : MODE>(ctr, (unsigned)(it + 1) * G);
        for (int r = 1; r <= R; ++r) {
            const int ob = (blockIdx.x + r * 37) % G;
            const float* theirs = buf + ((size_t)(it & 1) * G + ob) * floats_per_block;
            for (int i = threadIdx.x; i < floats_per_block; i += 256)
                if (MODE >= 0 && theirs[i] != (float)(it * 131 + ob + i)) ++wrong;
        }
    }
    if (wrong) atomicAdd(bad, wrong);
}

static float time_it(int mode, int G, int fpb, int iters, int R, float* buf, unsigned* ctr, unsigned* bad) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(ctr, 0, 4);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_rounds<0>, dim3(G), dim3(256), 0, 0, buf, fpb, iters, R, ctr, bad);
        else if (mode == 1) hipLaunchKernelGGL(k_rounds<1>, dim3(G), dim3(256), 0, 0, buf, fpb, iters, R, ctr, bad);
        else if (mode == 2) hipLaunchKernelGGL(k_rounds<2>, dim3(G), dim3(256), 0, 0, buf, fpb, iters, R, ctr, bad);
        else hipLaunchKernelGGL(k_rounds<-1>, dim3(G), dim3(256), 0, 0, buf, fpb, iters, R, ctr, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

__global__ void k_layer(float* buf, int floats_per_block, int it, int R) {      // the same round as ONE kernel of a dependent chain
    const int G = gridDim.x;
    float* mine = buf + ((size_t)(it & 1) * G + blockIdx.x) * floats_per_block;
    float s = 0.f;
    for (int r = 1; r <= R; ++r) {
        const int ob = (blockIdx.x + r * 37) % G;
        const float* theirs = buf + ((size_t)((it + 1) & 1) * G + ob) * floats_per_block;
        for (int i = threadIdx.x; i < floats_per_block; i += 256) s += theirs[i];
    }
    for (int i = threadIdx.x; i < floats_per_block; i += 256) mine[i] = s + (float)i;
}

int main() {
    float* buf; hipMalloc(&buf, 2u * 1024 * 65536 * 4); hipMemset(buf, 0, 2u * 1024 * 65536 * 4);
    unsigned *ctr, *bad; hipMalloc(&ctr, 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    const int iters = 200;
    for (int G : {64, 128, 256, 512}) {
        for (int fpb : {256, 4096}) {
            const int R = 4;
            const float t0 = time_it(0, G, fpb, iters, R, buf, ctr, bad), t1 = time_it(1, G, fpb, iters, R, buf, ctr, bad);
            const float t2 = time_it(2, G, fpb, iters, R, buf, ctr, bad), tn = time_it(-1, G, fpb, iters, R, buf, ctr, bad);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_layer, dim3(G), dim3(256), 0, 0, buf, fpb, it, R);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("G=%3d blocks, %6d B written / block / round, reads 4 other blocks' slices: us / round = %6.2f (acquire polls) %6.2f (one write-back + "
                   "one invalidate per block) %6.2f (write-through stores + one invalidate per block) %5.2f (no barrier, no ordering) %5.2f (a chain of kernels)\n",
                   G, fpb * 4, t0 * 1e3 / iters, t1 * 1e3 / iters, t2 * 1e3 / iters, tn * 1e3 / iters, best * 1e3 / iters);
        }
    }
    unsigned hbad; hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    printf("stale reads behind the barrier: %u\n", hbad);
    return 0;
}
