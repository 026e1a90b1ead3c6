// conv3d_ws.hip — stride-1 3x3x3 convolution, cout <= 16, as a PERSISTENT, WARP-SPECIALISED kernel.
//
// Why: phase ablation of the plain LDS-staged kernel (conv3d.hip V2) on MI355X shows its MFMA phase runs at the
// matrix-pipe rate (59 us for conv0 of level 1) but the rest of each block — index math, global->LDS staging,
// barriers, epilogue, block start-up: 36 us — never overlaps it, because the two co-resident blocks of a CU
// stay in lockstep.  Occupancy, box-shape, priority and stagger experiments did not change that.  Here the
// overlap is structural instead of statistical:
//   * one 512-thread block per CU, resident for the whole launch, walking a contiguous run of boxes;
//   * waves 4-7 are PRODUCERS: they stage box n+1 (global -> registers -> LDS buffer (n+1)&1);
//   * waves 0-3 are CONSUMERS: one per SIMD, they run the 27-tap MFMA stream of box n from buffer n&1
//     (weight image resident in LDS for the whole launch, operands read one/two taps ahead) and store its outputs;
//   * one __syncthreads per box hands the buffers over.
// The matrix pipe therefore only idles during the consumers' short epilogue.
#include <stdlib.h>

#include "kernels.h"

namespace enerf {

// CIN in {8,16,32}; BD: box depth (2 or 4); box = BD x 8 x 16 outputs; cout <= 16 (one row tile).
template <int CIN, int BD>
__global__ __launch_bounds__(512) void k_conv3d_s1_ws(const float* __restrict__ wpk, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ in,
                                                         float* __restrict__ out, float* __restrict__ out2, int cout,
                                                         int relu, int B, int D, int H, int W, int nbd, int nbh, int nbw,
                                                         int boxes_per_block, int dbg) {
    constexpr int BH = 8, BW = 16;
    constexpr int CB = CIN >= 16 ? 16 : CIN, CPL = CB / 4, NCB = CIN / CB, KS = CIN / 4;
    constexpr int CTW = BD * BH / 4;
    constexpr int HX = BW + 2, HY = BH + 2, HZ = BD + 2, NVOX = HZ * HY * HX, QV = CB / 4;
    constexpr int NIT = (NVOX * QV + 255) / 256;
    constexpr int BUF = NVOX * CB;                       // floats per LDS buffer
    ENERF_DYN_SMEM(float, lds);

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const bool producer = wv >= 4;
    const int ptid = tid - 256;                          // producer thread id 0..255
    const int total_boxes = B * nbd * nbh * nbw;
    const int bid = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int box_begin = bid * boxes_per_block;
    const int box_end = min(box_begin + boxes_per_block, total_boxes);
    if (box_begin >= box_end) return;
    const int nstages = (box_end - box_begin) * NCB;     // stage = (box, 16-channel pass)

    auto box_origin = [&](int box, int& b, int& z0, int& y0, int& x0) {
        int t = box;
        x0 = (t % nbw) * BW; t /= nbw;
        y0 = (t % nbh) * BH; t /= nbh;
        z0 = (t % nbd) * BD;
        b = t / nbd;
    };
    // producer side: fill LDS buffer (st & 1) with the haloed box of stage st
    auto produce = [&](int st) {
        int b, z0, y0, x0;
        box_origin(box_begin + st / NCB, b, z0, y0, x0);
        const int cb = st % NCB;
        const float* inb = in + (long long)b * D * H * W * CIN + cb * CB;
        float* buf = lds + (st & 1) * BUF;
        float4 sv[NIT];
        bool sk[NIT];
        int pt = ptid;
        ENERF_OPAQUE_V(pt);                               // keep the per-element index math inside the stage loop
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = pt + it * 256;
            const int ic = i < NVOX * QV ? i : NVOX * QV - 1;
            const int v = ic / QV, q = ic - v * QV;
            const int dx = v % HX, dy = (v / HX) % HY, dz = v / (HX * HY);
            const int gx = x0 + dx - 1, gy = y0 + dy - 1, gz = z0 + dz - 1;
            // branch-free: unsigned range tests, clamped (always valid) address, zero-select at the LDS write
            sk[it] = ((unsigned)gx < (unsigned)W) & ((unsigned)gy < (unsigned)H) & ((unsigned)gz < (unsigned)D);
            const int cx = min(max(gx, 0), W - 1), cy = min(max(gy, 0), H - 1), cz = min(max(gz, 0), D - 1);
            const int vox = (cz * H + cy) * W + cx;
            sv[it] = *reinterpret_cast<const float4*>(inb + (long long)vox * CIN + q * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = ptid + it * 256;
            if (i < NVOX * QV) *reinterpret_cast<float4*>(buf + i * 4) = sk[it] ? sv[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    // the whole weight image lives in LDS for the lifetime of the block (27*KS*64 floats: 14/28/55 KB)
    float* wlds = lds + 2 * BUF;
    if (!producer) {
        for (int i = tid; i < 27 * KS * 16; i += 256)
            *reinterpret_cast<float4*>(wlds + i * 4) = *reinterpret_cast<const float4*>(wpk + i * 4);
    } else {
        produce(0);
    }
    __syncthreads();

    f32x4 acc[CTW];
    const float* wl = wlds + lane;
    constexpr int NAT = CPL;                             // A operands per tap (one row tile)
#pragma unroll 1
    for (int st = 0; st < nstages; ++st) {
        asm volatile("" ::: "memory");                    // weights are re-read from LDS per stage, not hoisted
        if (producer) {
            if (st + 1 < nstages && !(dbg & 1)) produce(st + 1);   // dbg&1: profiling aid, stage only the first box
        } else {
            const int cb = st % NCB;
            const float* buf = lds + (st & 1) * BUF;
            if (cb == 0) {
#pragma unroll
                for (int c = 0; c < CTW; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            auto issue_a = [&](int tap, float (&aq)[NAT]) {
                const float* wt = wl + (tap * KS + cb * CPL) * 64;
#pragma unroll
                for (int r = 0; r < CPL; ++r) aq[r] = wt[r * 64];
            };
            float aq[3][NAT];
            issue_a(0, aq[0]);
            issue_a(1, aq[1]);
            const float* lbase[CTW];
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int tile = wv * CTW + c, td = tile / BH, th = tile - td * BH;
                lbase[c] = buf + ((td * HY + th) * HX + j) * CB + g * CPL;
            }
            auto read_b = [&](int tap, float (&bv)[CTW][4]) {
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const int off = ((kd * HY + kh) * HX + kw) * CB;
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    if (CPL == 4) {
                        const float4 tq = *reinterpret_cast<const float4*>(lbase[c] + off);
                        bv[c][0] = tq.x; bv[c][1] = tq.y; bv[c][2] = tq.z; bv[c][3] = tq.w;
                    } else {
                        const float2 tq = *reinterpret_cast<const float2*>(lbase[c] + off);
                        bv[c][0] = tq.x; bv[c][1] = tq.y; bv[c][2] = 0.f; bv[c][3] = 0.f;
                    }
                }
            };
            float bq[2][CTW][4];
            read_b(0, bq[0]);
            if (!(dbg & 2))                               // dbg&2: profiling aid, no MFMA phase
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                if (tap + 2 < 27) issue_a(tap + 2, aq[(tap + 2) % 3]);
                if (tap + 1 < 27) read_b(tap + 1, bq[(tap + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < CPL; ++r)
#pragma unroll
                    for (int c = 0; c < CTW; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[tap % 3][r], bq[tap & 1][c][r], acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (cb == NCB - 1 && !(dbg & 4)) {            // epilogue: BN scale/shift, ReLU, float4 store
                int b, z0, y0, x0;
                box_origin(box_begin + st / NCB, b, z0, y0, x0);
                const int c0 = 4 * g;
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    const int tile = wv * CTW + c, td = tile / BH, th = tile - td * BH;
                    const int z = z0 + td, y = y0 + th, x = x0 + j;
                    if (z >= D || y >= H || x >= W || c0 >= cout) continue;
                    const long long o = (((long long)b * D + z) * H + y) * W + x;
                    float yv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) yv[r] = acc[c][r] * scale[c0 + r] + shift[c0 + r];
                    if (out2 != nullptr) {               // fused heads: channels 0..7 -> out, channel 8 -> out2
                        if (c0 < 8) *reinterpret_cast<float4*>(out + o * 8 + c0) = make_float4(yv[0], yv[1], yv[2], yv[3]);
                        else if (c0 == 8) out2[o] = yv[0];
                        continue;
                    }
                    if (relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) yv[r] = fmaxf(yv[r], 0.f);
                    }
                    *reinterpret_cast<float4*>(out + o * cout + c0) = make_float4(yv[0], yv[1], yv[2], yv[3]);
                }
            }
        }
        __syncthreads();                                  // buffer (st+1)&1 is full, buffer st&1 is free
    }
}

template <int CIN, int BD>
static void launch_ws(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W,
                      hipStream_t st) {
    constexpr int CB = CIN >= 16 ? 16 : CIN;
    const int nbd = cdiv(D, BD), nbh = cdiv(H, 8), nbw = cdiv(W, 16);
    const size_t shmem = ((size_t)2 * (BD + 2) * 10 * 18 * CB + 27 * (CIN / 4) * 64) * sizeof(float);
    const int total = B * nbd * nbh * nbw;
    const char* e = getenv("ENERF_CONV_WS_BLOCKS");                    // persistent blocks (default: one per CU)
    const int nblocks = e ? atoi(e) : 256;
    const int bpb = cdiv(total, nblocks > 0 ? nblocks : 256);
    const unsigned grid = (unsigned)cdiv(total, bpb);
    const char* ed = getenv("ENERF_CONV_DBG");
    const int dbg = ed ? atoi(ed) : 0;
#ifndef ENERF_EMU
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_s1_ws<CIN, BD>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)attr;
#endif
    ENERF_LAUNCH((k_conv3d_s1_ws<CIN, BD>), grid, 512, shmem, st, L.w, L.scale, L.shift, in, out, out2, L.cout, L.relu, B, D,
                 H, W, nbd, nbh, nbw, bpb, dbg);
}
// cout <= 16 stride-1 layers.  Returns false if the shape is not handled.
bool launch_conv3d_ws(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W,
                      hipStream_t st) {
    if (L.kind != kConvS1 || L.cout > 16) return false;
    const char* e = getenv("ENERF_CONV_WS_BD");
    const int bd = e ? atoi(e) : 2;
    const bool bd4 = (bd == 4) && (D % 4 == 0);
    switch (L.cin) {
        case 8: bd4 ? launch_ws<8, 4>(L, in, out, out2, B, D, H, W, st) : launch_ws<8, 2>(L, in, out, out2, B, D, H, W, st); return true;
        // two BD=4 buffers + the weight image exceed the 160 KB of LDS for Cin >= 16
        case 16: launch_ws<16, 2>(L, in, out, out2, B, D, H, W, st); return true;
        case 32: launch_ws<32, 2>(L, in, out, out2, B, D, H, W, st); return true;
        default: return false;
    }
}

}  // namespace enerf
