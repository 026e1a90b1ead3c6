// GPU micro-test: the DPP / permlane primitives of enerf_amd/csrc/common.h against the plain __shfl forms, for every
// group width and source lane.  The CPU lane emulator compiles the ENERF_EMU branches of these helpers, so only a GPU run exercises
// the real instruction sequences.  Lanes of a group must be convergent (bound_ctrl = true reads 0 from an inactive source lane).
// hipcc --offload-arch=gfx950 -O3 -w -I enerf_amd/csrc tools/micro/dpp_primitives.hip -o tools/micro/dpp_primitives.bin
#include "common.h"
#include <stdio.h>
using namespace enerf;

__global__ void k(int* bad) {
    const int lane = threadIdx.x & 63;
    const int v = 1000 + 7 * lane + (int)blockIdx.x;
    const float f = 0.5f + 0.25f * lane;
    int nb = 0;
#define CHK(cond) nb += (cond) ? 0 : 1
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) CHK(group_bcast_i<2>(v, k2) == __shfl(v, (lane & ~1) + k2));
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) CHK(group_bcast_i<4>(v, k4) == __shfl(v, (lane & ~3) + k4));
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) CHK(group_bcast_i<8>(v, k8) == __shfl(v, (lane & ~7) + k8));
    float rs = f;
    rs += __shfl_xor(rs, 8); rs += __shfl_xor(rs, 4); rs += __shfl_xor(rs, 2); rs += __shfl_xor(rs, 1);
    CHK(fabsf(row_sum16(f) - rs) <= 1e-4f * fabsf(rs));                 // (the rotation order differs from the butterfly's)
    CHK(add_xor8(f) == f + __shfl_xor(f, 8));
    CHK(xor16(f) == __shfl_xor(f, 16));
    CHK(xor32(f) == __shfl_xor(f, 32));
    float gs = f; gs += __shfl_xor(gs, 16); gs += __shfl_xor(gs, 32);
    CHK(fabsf(group_sum4(f) - gs) <= 1e-5f * fabsf(gs));
    float gm = fmaxf(f, __shfl_xor(f, 16)); gm = fmaxf(gm, __shfl_xor(gm, 32));
    CHK(group_max4(f) == gm);
    atomicAdd(bad, nb);
}
int main() {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, d);
    int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("DPP / permlane primitives of common.h vs __shfl (group_bcast_i<2|4|8, K>, row_sum16, add_xor8, xor16, xor32, group_sum4, group_max4; "
           "8 blocks x 4 waves, 22 checks per lane): %d mismatches -> %s\n", h, h == 0 ? "OK" : "FAIL");
    return h != 0;
}
