// conv3d_s2.hip — stride-2 3x3x3 convolution (Cin = 8, Cout <= 16) with the haloed input box staged in LDS.
//
// conv1 of both cost-regularisation nets (8 -> 16, stride 2) ran on the global-load kernel (conv3d.hip V1): 27
// dependent gather rounds per wave, 21 us at level 1 against a 5 us MFMA / 4 us HBM floor (PMC: 46 % of wave
// cycles parked in s_waitcnt).  Here a block owns a 2 x 4 x 16 box of outputs; the 5 x 9 x 33 input box (47.5 KB)
// is copied once into LDS — every input voxel is fetched ~1.2 times instead of ~3.4 — and a tap's B operand is
// one ds_read_b64 at a constant offset from the lane's voxel (2j along x).  Same operand mapping, packed weight
// image and epilogue conventions as k_conv3d_s1_lds.
#include "kernels.h"

namespace enerf {

template <int CIN>
__global__ __launch_bounds__(256, 3) void k_conv3d_s2_lds(const float* __restrict__ wpk, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ in,
                                                          float* __restrict__ out, int cout, int relu, int B, int Di, int Hi,
                                                          int Wi, int Do, int Ho, int Wo, int nbd, int nbh, int nbw) {
    static_assert(CIN == 8, "only the Cin = 8 stride-2 layers are routed here");
    constexpr int BD = 2, BH = 4, BW = 16;                 // output box
    constexpr int CB = CIN, CPL = CB / 4, KS = CIN / 4, QV = CB / 4;
    constexpr int CTW = BD * BH / 4;                        // column tiles per wave (2)
    constexpr int HX = 2 * BW + 1, HY = 2 * BH + 1, HZ = 2 * BD + 1, NVOX = HZ * HY * HX;
    constexpr int NIT = (NVOX * QV + 255) / 256;
    ENERF_DYN_SMEM(float, lds);

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    int t = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh; t /= nbh;
    const int bd = t % nbd;
    const int b = t / nbd;
    const int ox0 = bw * BW, oy0 = bh * BH, oz0 = bd * BD;               // first output voxel of the box
    const int ix0 = 2 * ox0 - 1, iy0 = 2 * oy0 - 1, iz0 = 2 * oz0 - 1;   // first input voxel of the haloed box
    const float* inb = in + (long long)b * Di * Hi * Wi * CIN;
    const float* wl = wpk + lane;

    // weights of the first taps are requested before the box is staged (latency hides behind the staging traffic)
    constexpr int NAT = CPL;
    auto issue_a = [&](int tap, float (&aq)[NAT]) {
        const float* wt = wl + (long long)tap * KS * 64;
#pragma unroll
        for (int r = 0; r < CPL; ++r) aq[r] = wt[r * 64];
    };
    float aq[3][NAT];
    issue_a(0, aq[0]);
    issue_a(1, aq[1]);
    __builtin_amdgcn_sched_barrier(0);
    {   // stage the haloed box: unconditional clamped loads issued back to back, zero-select at the LDS write
        float4 sv[NIT];
        bool sk[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * 256;
            const int ic = i < NVOX * QV ? i : NVOX * QV - 1;
            const int v = ic / QV, q = ic - v * QV;
            const int dx = v % HX, dy = (v / HX) % HY, dz = v / (HX * HY);
            const int gx = ix0 + dx, gy = iy0 + dy, gz = iz0 + dz;
            sk[it] = gx >= 0 && gx < Wi && gy >= 0 && gy < Hi && gz >= 0 && gz < Di;
            const long long off = sk[it] ? (((long long)gz * Hi + gy) * Wi + gx) : 0;
            sv[it] = *reinterpret_cast<const float4*>(inb + off * CIN + q * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = threadIdx.x + it * 256;
            if (i < NVOX * QV)
                *reinterpret_cast<float4*>(lds + i * 4) = sk[it] ? sv[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    f32x4 acc[CTW];
    const float* lbase[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tile = wv * CTW + c, td = tile / BH, th = tile - td * BH;
        lbase[c] = lds + ((2 * td * HY + 2 * th) * HX + 2 * j) * CB + g * CPL;
    }
    auto read_b = [&](int tap, float (&bv)[CTW][2]) {
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const int off = ((kd * HY + kh) * HX + kw) * CB;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const float2 tq = *reinterpret_cast<const float2*>(lbase[c] + off);
            bv[c][0] = tq.x; bv[c][1] = tq.y;
        }
    };
    float bq[2][CTW][2];
    read_b(0, bq[0]);
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

    // ---- epilogue: BN scale/shift, ReLU, float4 store ----
    const int c0 = 4 * g;
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int tile = wv * CTW + c, td = tile / BH, th = tile - td * BH;
        const int z = oz0 + td, y = oy0 + th, x = ox0 + j;
        if (z >= Do || y >= Ho || x >= Wo || c0 >= cout) continue;
        const long long o = (((long long)b * Do + z) * Ho + y) * Wo + x;
        float yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            yv[r] = acc[c][r] * scale[c0 + r] + shift[c0 + r];
            if (relu) yv[r] = relu1(yv[r]);
        }
        *reinterpret_cast<float4*>(out + o * cout + c0) = make_float4(yv[0], yv[1], yv[2], yv[3]);
    }
}

// Cin = 8, Cout <= 16 stride-2 layers.  Returns false if the shape is not handled.
bool launch_conv3d_s2_lds(const Conv3dDesc& L, const float* in, float* out, int B, int Di, int Hi, int Wi, hipStream_t st) {
    if (L.kind != kConvS2 || L.cin != 8 || L.cout > 16) return false;
    const int Do = (Di - 1) / 2 + 1, Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const int nbd = cdiv(Do, 2), nbh = cdiv(Ho, 4), nbw = cdiv(Wo, 16);
    const size_t shmem = (size_t)5 * 9 * 33 * 8 * sizeof(float);
    const unsigned grid = (unsigned)((long long)B * nbd * nbh * nbw);
    ENERF_LAUNCH((k_conv3d_s2_lds<8>), grid, 256, shmem, st, L.w, L.scale, L.shift, in, out, L.cout, L.relu, B, Di, Hi, Wi, Do,
                 Ho, Wo, nbd, nbh, nbw);
    return true;
}

}  // namespace enerf
