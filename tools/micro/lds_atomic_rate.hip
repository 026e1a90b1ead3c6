// Micro-benchmark: ds_add_f32 (fp32 add onto LDS) on MI355X under the address patterns of the tiled gather backward:
//   0: 64 lanes, 64 distinct consecutive words            1: 4 groups of 11 lanes (44 active), 4 distinct texel runs
//   2: as 1, but groups 0/1 and 2/3 hit the SAME run (the two samples of a ray)     3: all four groups the same run
//   4: plain ds_write_b32 of pattern 0 (no atomic) for reference          5: pattern 1 as read-add-write without atomics
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/micro/lds_atomic_rate.hip -o /tmp/la && /tmp/la
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) float lds_float;
template <int PAT>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, ch = lane & 15;
    unsigned h = 12345u + wave * 977u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const int base = (h >> 8) % 600;                               // a texel index in a 28 x 14 patch + margin
        int idx; bool on = true;
        if (PAT == 0 || PAT == 4) idx = (base * 11 + lane) & 8191;
        else {
            const int grp = PAT == 1 || PAT == 5 ? g : (PAT == 2 ? (g >> 1) : 0);
            idx = ((base + grp * 3) * 11 + ch) & 8191; on = ch < 11;
        }
        if (on) {
            if (PAT == 4) buf[idx] = (float)it;
            else if (PAT == 5) buf[idx] += 1.0f;
            else __builtin_amdgcn_ds_faddf((lds_float*)(buf + idx), 1.0f, 0, 0, false);
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 8192; i += 256) s += buf[i];
    if (s == -1.f) out[0] = s;
}
template <int PAT> static void run(float* out, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4, iters = 4096;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double instr_per_cu = (double)blocks * 4 * iters / 256;      // wave instructions per CU
    printf("pattern %d (%s): %.3f ms, %.1f cycles at 2.4 GHz per wave instruction per CU\n", PAT, what, best, best * 2.4e6 / instr_per_cu);
}
int main() {
    float* out; hipMalloc(&out, 16);
    run<0>(out, "ds_add_f32, 64 distinct consecutive words");
    run<1>(out, "ds_add_f32, 4 groups x 11 lanes, 4 texel runs");
    run<2>(out, "ds_add_f32, groups pairwise on the same run");
    run<3>(out, "ds_add_f32, all four groups on one run");
    run<4>(out, "ds_write_b32, 64 consecutive words");
    run<5>(out, "non-atomic read-add-write, 4 runs");
    return 0;
}
