// selftest.hip — the hardware-only primitives of common.h against a specification that uses nothing but memory.
//
// Every helper in common.h that is an intrinsic on gfx950 (LDS-DMA copies, raw buffer loads, DPP / permlane lane exchanges, the
// matrix instructions and their operand layouts, med3 ReLU, 24-bit multiplies, LDS / global fp32 atomics, the hardware
// transcendentals) has a second implementation under ENERF_EMU for the CPU lane emulator the `-m "not gpu"` tests run on.  A kernel
// test passing on the emulator therefore says nothing about the intrinsic branch unless the two branches are known to mean the same
// thing.  This file pins both to ONE specification: each check computes the primitive, then the same quantity from values staged
// through LDS / global memory with plain indexed reads (no cross-lane instruction, no matrix instruction), and counts the lanes
// that disagree.  tests/test_primitives_selftest.py runs it through the C ABI on the emulator build (the ENERF_EMU twins) and, with
// `-m gpu`, on the gfx950 build (the intrinsics): both must report zero mismatches in every check.
// Values are small integers stored as floats wherever a sum is compared, so every association order gives the same bits.
#include "kernels.h"

namespace enerf {

constexpr int kSelftestChecks = 20;

// the value lane `src` of THIS wave holds, through LDS (the specification's only cross-lane device)
__device__ __forceinline__ float staged(float* stage, float v, int src) {
    __syncthreads();
    stage[threadIdx.x] = v;
    __syncthreads();
    return stage[(threadIdx.x & ~63) + src];
}

__global__ __launch_bounds__(256) void k_selftest_primitives(const float* __restrict__ table, int table_floats, float* __restrict__ scratch,
                                                             unsigned* __restrict__ bad) {
    __shared__ float stage[256];
    __shared__ __attribute__((aligned(16))) float dma[4 * 256 + 64];
    __shared__ float lacc[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, j = lane & 15;
    const float f = (float)((lane * 37 + wv * 11 + (int)blockIdx.x * 5) % 201 - 100);     // integers in [-100, 100]
    const int iv = 1000 + 7 * lane + 3 * wv + (int)blockIdx.x;
    unsigned nb[kSelftestChecks];
#pragma unroll
    for (int c = 0; c < kSelftestChecks; ++c) nb[c] = 0;
#define CHK(c, cond) nb[c] += (cond) ? 0u : 1u

    // 0, 1: xor16 / xor32 (v_permlane16_swap / v_permlane32_swap)
    CHK(0, xor16(f) == staged(stage, f, lane ^ 16));
    CHK(1, xor32(f) == staged(stage, f, lane ^ 32));
    // 2, 3: group_sum4 / group_max4 over lanes j, j+16, j+32, j+48
    {
        const float a0 = staged(stage, f, j), a1 = staged(stage, f, j + 16), a2 = staged(stage, f, j + 32), a3 = staged(stage, f, j + 48);
        CHK(2, group_sum4(f) == ((a0 + a1) + (a2 + a3)));
        CHK(3, group_max4(f) == fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    }
    // 4, 5: row_sum16 (four DPP row rotations) and add_xor8
    {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += staged(stage, f, (lane & ~15) + k);
        CHK(4, row_sum16(f) == s);
        CHK(5, add_xor8(f) == f + staged(stage, f, lane ^ 8));
    }
    // 6, 7, 8: group_bcast_i<2 | 4 | 8, K> (DPP quad_perm / row shifts)
#pragma unroll
    for (int k = 0; k < 2; ++k) CHK(6, group_bcast_i<2>(iv, k) == __float_as_int(staged(stage, __int_as_float(iv), (lane & ~1) + k)));
#pragma unroll
    for (int k = 0; k < 4; ++k) CHK(7, group_bcast_i<4>(iv, k) == __float_as_int(staged(stage, __int_as_float(iv), (lane & ~3) + k)));
#pragma unroll
    for (int k = 0; k < 8; ++k) CHK(8, group_bcast_i<8>(iv, k) == __float_as_int(staged(stage, __int_as_float(iv), (lane & ~7) + k)));

    // 9: glds16 — every lane names its own 16-byte source, the destination is wave base + lane * 16; issued before a RAW barrier,
    //    completed by vmem_wait_pending<0> + the barrier, read by ANOTHER wave (the pipelining idiom of the convolution kernels)
    {
        const int ntex = table_floats / 4, src = (int)(((unsigned)(tid * 97 + (int)blockIdx.x * 13 + 5)) % (unsigned)ntex);
        __syncthreads();
        glds16(table + src * 4, dma + wv * 256, lane);
        block_barrier_raw();                                  // the copy stays in flight across this barrier
        vmem_wait_pending<0>();
        block_barrier_raw();
        const int ot = (tid + 64) & 255, osrc = (int)(((unsigned)(ot * 97 + (int)blockIdx.x * 13 + 5)) % (unsigned)ntex);
        for (int k = 0; k < 4; ++k) CHK(9, dma[ot * 4 + k] == table[osrc * 4 + k]);
        // and the plain completion form
        __syncthreads();
        glds16(table + ((src + 1) % ntex) * 4, dma + wv * 256, lane);
        glds_wait_all();
        __syncthreads();
        for (int k = 0; k < 4; ++k) CHK(9, dma[ot * 4 + k] == table[((osrc + 1) % ntex) * 4 + k]);
    }
    // 10: raw buffer loads: in range = the plain load; at / past the size, and the all-ones offset = 0
    {
        const unsigned bytes = (unsigned)table_floats * 4u - 32u;                  // a resource 8 floats shorter than the table
        const BufRsrc r = buf_rsrc(table, bytes);
        const unsigned off = (unsigned)((tid * 29 + (int)blockIdx.x) % (table_floats - 8)) * 4u;
        CHK(10, buf_load_f32(r, off) == table[off / 4]);
        CHK(10, buf_load_f32(r, bytes + (unsigned)(lane & 7) * 4u) == 0.f);
        CHK(10, buf_load_f32(r, 0xffffffffu) == 0.f);
        const unsigned off16 = (unsigned)((tid * 13 + (int)blockIdx.x) % ((table_floats - 8) / 4 - 1)) * 16u;
        const float4 v = buf_load_f32x4(r, off16);
        CHK(10, v.x == table[off16 / 4] && v.y == table[off16 / 4 + 1] && v.z == table[off16 / 4 + 2] && v.w == table[off16 / 4 + 3]);
        const float4 z = buf_load_f32x4(r, (lane & 1) ? bytes : 0xffffffffu);
        CHK(10, z.x == 0.f && z.y == 0.f && z.z == 0.f && z.w == 0.f);
    }
    // 11: v_mfma_f32_16x16x4_f32 operand layout: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
    //     D[row 4 (lane >> 4) + r][col lane & 15] += sum_k A[row][k] B[k][col]
    {
        const float a = (float)((lane * 5 + wv) % 13 - 6), b = (float)((lane * 3 + 2 * wv) % 11 - 5);
        f32x4 c = f32x4{1.f, 2.f, 3.f, 4.f};
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) {
            float s = c[r];
            for (int k = 0; k < 4; ++k) s += staged(stage, a, 16 * k + 4 * g + r) * staged(stage, b, 16 * k + j);
            CHK(11, d[r] == s);
        }
    }
    // 12: v_mfma_f32_4x4x1_16b_f32 with cbsz:4 abid:K (mfma4_bc): D[r] += A[lane 4K + r] * B[own lane], for every K
    {
        const float a = (float)((lane * 7 + wv) % 17 - 8), b = (float)((lane + 3 * wv) % 9 - 4);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f32x4 d = mfma4_bc(a, b, f32x4{0.5f, -1.f, 2.f, 0.f}, k);
            const float c0[4] = {0.5f, -1.f, 2.f, 0.f};
            for (int r = 0; r < 4; ++r) CHK(12, d[r] == c0[r] + staged(stage, a, 4 * k + r) * b);
        }
    }
    // 13: the plain 4x4x1 form (no broadcast): 16 independent 4x4 blocks, lane 4b + i supplies A_b[i] and B_b[i]: D_b[r][i] = A_b[r] B_b[i]
    {
        const float a = (float)(lane % 7 - 3), b = (float)(lane % 5 - 2);
        const f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        for (int r = 0; r < 4; ++r) CHK(13, d[r] == staged(stage, a, (lane & ~3) + r) * b);
    }
    // 14: relu1 (v_med3_f32(x, 0, FLT_MAX)): max(x, 0) for finite x, 0 for NaN
    {
        const float xs[6] = {f, -f, 0.f, -0.f, 1e-30f * (float)(lane + 1), __int_as_float(0x7fc00000)};
        for (int k = 0; k < 5; ++k) CHK(14, relu1(xs[k]) == (xs[k] > 0.f ? xs[k] : 0.f));
        CHK(14, relu1(xs[5]) == 0.f);
    }
    // 15: mul24 on coordinates / extents below 2^24
    {
        const int a = (lane * 523 + wv * 77) & 0xfff, b = (lane * 131 + 9) & 0xfff;
        CHK(15, mul24(a, b) == a * b);
        CHK(15, mul24(-a, b) == -a * b);
    }
    // 16: lds_add_f32 (ds_add_f32) and atomic_add_f32 (global_atomic_add_f32): every lane adds, the slot holds the sum
    {
        __syncthreads();
        if (tid < 64) lacc[tid] = 0.f;
        __syncthreads();
        lds_add_f32(lacc + j, f);
        __syncthreads();
        float s = 0.f;
        for (int t = j; t < 256; t += 16) {                                  // lanes with (t & 15) == j added f(t)
            const int tl = t & 63, tw = t >> 6;
            s += (float)((tl * 37 + tw * 11 + (int)blockIdx.x * 5) % 201 - 100);
        }
        CHK(16, lacc[j] == s);
        float* slot = scratch + (size_t)blockIdx.x * 16 + j;                  // zeroed by the launcher
        atomic_add_f32(slot, f);
        __threadfence();
        __syncthreads();
        CHK(16, *(volatile float*)slot == s);
    }
    // 17: the hardware transcendentals (v_rcp_f32, v_sqrt_f32, v_exp_f32) within 2e-6 of the IEEE forms
    {
        const float x = 0.25f + 0.37f * (float)(lane + 64 * wv);
        CHK(17, fabsf(fast_rcp(x) - 1.f / x) <= 2e-6f * (1.f / x));
        CHK(17, fabsf(fast_sqrt(x) - sqrtf(x)) <= 2e-6f * sqrtf(x));
        const float e = -0.05f * (float)lane;
        CHK(17, fabsf(fast_exp(e) - expf(e)) <= 2e-6f * expf(e));
    }
    // 18: wave_sync: LDS written by some lanes of a wave is visible to the others behind it (no block barrier)
    {
        __syncthreads();
        stage[tid] = f;
        wave_sync();
        CHK(18, stage[(tid & ~63) + ((lane + 17) & 63)] == staged(stage, f, (lane + 17) & 63));
    }
    // 19: xcd_contiguous is a bijection of the block ids (checked by block 0 for this grid)
    if (blockIdx.x == 0) {
        const unsigned n = gridDim.x;
        for (unsigned b = tid; b < n; b += 256) {
            const unsigned m = xcd_contiguous(b, n);
            unsigned hits = 0;
            for (unsigned o = 0; o < n; ++o) hits += xcd_contiguous(o, n) == m ? 1u : 0u;
            CHK(19, m < n && hits == 1u);
        }
    }
#undef CHK
#pragma unroll
    for (int c = 0; c < kSelftestChecks; ++c)
        if (nb[c]) atomicAdd(bad + c, nb[c]);
}

}  // namespace enerf

using namespace enerf;
extern "C" int enerf_selftest_checks(void) { return kSelftestChecks; }
extern "C" int enerf_selftest_primitives(const float* table, int table_floats, float* scratch, int blocks, unsigned* mismatches,
                                         enerf_stream_t stream) {
    REQUIRE(table && scratch && mismatches, "selftest_primitives: null pointer");
    REQUIRE(table_floats >= 1024 && table_floats % 4 == 0 && blocks >= 1 && blocks <= 4096, "selftest_primitives: table of >= 1024 floats (a multiple of 4), 1..4096 blocks");
    zero_async(mismatches, kSelftestChecks * sizeof(unsigned), (hipStream_t)stream);
    zero_async(scratch, (size_t)blocks * 16 * sizeof(float), (hipStream_t)stream);
    ENERF_LAUNCH(k_selftest_primitives, (unsigned)blocks, 256, 0, (hipStream_t)stream, table, table_floats, scratch, mismatches);
    return check_launch("selftest_primitives");
}
