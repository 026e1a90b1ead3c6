// conv3d_wl.hip — the small deep stride-1 / stride-2 layers of the cost-regularisation U-Nets (conv3 .. conv6 of
// cost_reg_net.py:13-24,58-62: 16 -> 32 s2, 32 -> 32, 32 -> 64 s2, 64 -> 64 on 1/16 .. 1/512 of the volume) with the BLOCK's
// weight tile in LDS and every operand of a wave requested before its first MFMA.
//
// These layers have 80 .. 640 column tiles of 16 voxels for 1024 SIMDs, so k_conv3d (conv3d.hip) splits the 27 taps of a tile
// over three waves — and then every wave streams its 9 taps' weights (18 KB at Cin = 32) through L1/L2 behind a two-deep ring:
// nine dependent round trips and ~90 vector-memory instructions for 72 MFMAs (VERDICT r05 #1a).  Here
//   * a block = CTB column tiles x 3 waves (the taps split by kw) of ONE 16-row output-channel tile; the tile's packed weights
//     (conv3d.hip layout: 27 * Cin/4 rows of 256 B) are copied ONCE per block by LDS-DMA (global_load_lds_dwordx4, no VGPRs)
//     and all 3 * CTB waves read their A operands from there (ds_read_b32, conflict-free);
//   * the B operands (one float4 of channels-last activations per lane, tap and 16 channels) of all 9 taps of a wave are
//     requested up front next to the copy: ONE memory round trip per wave instead of nine;
//   * kd taps that are padding for the whole block are skipped — copy, loads and MFMAs.  Level 1's deep layers are 1 – 2 voxels
//     thick (8 depth planes -> 4 -> 2 -> 1): conv6 keeps 9 of its 27 taps, conv4 / conv5 18;
//   * scale / shift are requested in the prologue, so nothing but stores follows the last MFMA.
// Summation order: kw-major (k_conv3d: kd-major), so results differ from that kernel by fp32 re-association only.

#include "kernels.h"

// timing ablations (tools/build_variant.py; outputs are garbage): bit 0 = no weight copy, 1 = no activation loads, 2 = no MFMAs,
// 3 = no stores, 4 = no XCD-contiguous block order

namespace enerf {

// 512 B of zeros: the loads of padding lanes are pointed here (-fno-gpu-rdc: device variables are per translation unit)
__device__ float g_wl_zeros[128];

template <int CIN, int KIND, int CTB>
__global__ __launch_bounds__(CTB * 192) void k_conv3d_wl(const float* __restrict__ wpk, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ in,
                                                        float* __restrict__ out, int cout, int relu, int B, int Di, int Hi, int Wi,
                                                        int Do, int Ho, int Wo, int rt_total, int kdlo, int nkd) {
    static_assert(KIND == kConvS1 || KIND == kConvS2, "stride-1 / stride-2 layers");
    static_assert(CIN == 16 || CIN == 32 || CIN == 64, "Cin");
    constexpr int S = KIND == kConvS2 ? 2 : 1;
    constexpr int KS = CIN / 4, NB = CIN / 16, NW = CTB * 3;
    ENERF_DYN_SMEM(float, lds);
    float* wlds = lds;                                   // [nkd * 9 taps][KS][64]
    float* red = lds + nkd * 9 * KS * 64;                // [2][CTB][4][64]: partial sums of the kw = 1, 2 waves

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sp = wv % 3, cl = wv / 3;                  // kw of this wave's taps, column tile inside the block
    const int bid = (0 & 16) ? (int)blockIdx.x : (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int grp = bid / rt_total, rt = bid - grp * rt_total;

    const int n = B * Do * Ho * Wo;                      // output voxels in raster order (n < 2^31: launcher)
    const int tiles = cdiv(n, 16);
    const int tile = grp * CTB + cl;
    // ---- this lane's voxel: scalar decomposition of the tile's first voxel, lanes add j and carry ----
    int vb, vd, vh, vw;
    bool vok;
    {
        const int v0 = (tile < tiles ? tile : tiles - 1) * 16;
        int r = v0 / Wo;
        const int w0 = v0 - r * Wo;
        int q = r / Ho;
        const int h0 = r - q * Ho;
        const int b0 = q / Do, d0 = q - b0 * Do;
        vok = tile < tiles && v0 + j < n;
        vw = w0 + j; vh = h0; vd = d0; vb = b0;
        while (vw >= Wo) {
            vw -= Wo;
            if (++vh == Ho) { vh = 0; if (++vd == Do) { vd = 0; ++vb; } }
        }
    }
    const int id0 = vd * S - 1, ih0 = vh * S - 1, iw0 = vw * S - 1;
    const int vbase = ((vb * Di + id0) * Hi + ih0) * Wi + iw0;
    unsigned vmask = 0;                                   // bit k / 3+k / 6+k: offset k valid along d / h / w
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vmask |= ((unsigned)(id0 + k) < (unsigned)Di ? 1u : 0u) << k;
        vmask |= ((unsigned)(ih0 + k) < (unsigned)Hi ? 1u : 0u) << (3 + k);
        vmask |= ((unsigned)(iw0 + k) < (unsigned)Wi ? 1u : 0u) << (6 + k);
    }
    if (!vok) vmask = 0u;

    // ---- which kd does ANY voxel of this block read inside the volume?  (block-uniform; the launcher's [kdlo, kdlo + nkd) is
    // the same question for the whole layer and sizes the LDS image) ----
    unsigned kdmask = 0;
    {
        const int vfirst = grp * CTB * 16;
        const int vlast = (vfirst + CTB * 16 < n ? vfirst + CTB * 16 : n) - 1;
        const int plane = Ho * Wo;
        const int q0 = vfirst / plane, q1 = vlast / plane;
        const bool one_batch = q0 / Do == q1 / Do;
        const int dmin = one_batch ? q0 % Do : 0, dmax = one_batch ? q1 % Do : Do - 1;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
            if (kd >= kdlo && kd < kdlo + nkd && S * dmax + kd - 1 >= 0 && S * dmin + kd - 1 < Di) kdmask |= 1u << kd;
    }

    // ---- LDS-DMA of the weight rows of the valid kd (a wave instruction copies 4 rows of 256 B; 9 * KS rows per kd) ----
    {
        const int ninstr = nkd * 9 * KS / 4;
        const float* wsrc = wpk + ((long long)(lane >> 4) * rt_total + rt) * 64 + (lane & 15) * 4;
#pragma unroll 1
        for (int i = wv; i < ninstr; i += NW) {
            const int row0 = 4 * i;
            const int kd = kdlo + row0 / (9 * KS);
            if (((kdmask >> kd) & 1u) && !(0 & 1)) glds16(wsrc + (long long)(kdlo * 9 * KS + row0) * rt_total * 64, wlds + row0 * 64, lane);
        }
    }
    // ---- every B operand of this wave's (up to) 9 taps: unconditional loads, padding lanes read zeros ----
    float4 bq[9][NB];
    const float* zeros = g_wl_zeros + g * 4;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
        if (!((kdmask >> kd) & 1u)) continue;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const unsigned sel = (1u << kd) | (8u << kh) | (64u << sp);
            const bool ok = (vmask & sel) == sel;
            const float* p = (ok && !(0 & 2)) ? in + (long long)(vbase + (kd * Hi + kh) * Wi + sp) * CIN + g * 4 : zeros;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) bq[kd * 3 + kh][cb] = *reinterpret_cast<const float4*>(p + cb * 16);
        }
    }
    const int c0 = rt * 16 + 4 * g;
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[r] = scale[c0 + r]; sh[r] = shift[c0 + r]; }      // padded to 16 * row tiles: always valid
    glds_wait_all();
    __syncthreads();

    // ---- MFMAs: A operands from the block's LDS image, the next tap's reads issued ahead of this tap's MFMAs ----
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wl = wlds + lane;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
        if (!((kdmask >> kd) & 1u)) continue;
        float aq[2][KS];
        const float* wk = wl + ((kd - kdlo) * 9 + sp) * KS * 64;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) aq[0][ks] = wk[ks * 64];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            if (kh + 1 < 3) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) aq[(kh + 1) & 1][ks] = wk[((kh + 1) * 3 * KS + ks) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float4 bv = bq[kd * 3 + kh][ks >> 2];
                const float b = (ks & 3) == 0 ? bv.x : ((ks & 3) == 1 ? bv.y : ((ks & 3) == 2 ? bv.z : bv.w));
                if (0 & 4) acc[ks & 3] += aq[kh & 1][ks] * b;
                else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kh & 1][ks], b, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- kw = 1, 2 partial sums into the kw = 0 wave (fixed order: deterministic) ----
    if (sp > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(((sp - 1) * CTB + cl) * 4 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (sp == 0 && vok) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += red[((s2 * CTB + cl) * 4 + r) * 64 + lane];
        if (c0 < cout) {
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[r] = acc[r] * sc[r] + sh[r];
                if (relu) y[r] = relu1(y[r]);
            }
            const long long o = (long long)(tile * 16 + j);
            if (!(0 & 8) || y[0] == 12345.678f) *reinterpret_cast<float4*>(out + o * cout + c0) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

template <int CIN, int KIND, int CTB>
static bool launch_wl(const Conv3dDesc& L, const float* in, float* out, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                      int kdlo, int nkd, hipStream_t st) {
    const int rt_total = cdiv(L.cout, 16);
    const long long n = (long long)B * Do * Ho * Wo;
    const long long groups = cdivl(cdivl(n, 16), CTB);
    const size_t shmem = ((size_t)nkd * 9 * (CIN / 4) * 64 + 2 * CTB * 256) * sizeof(float);
#ifndef ENERF_EMU
    if (shmem > 64 * 1024) {                              // opt in to a large dynamic LDS allocation once per instantiation
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_wl<CIN, KIND, CTB>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr != hipSuccess) { (void)hipGetLastError(); return false; }
    }
#endif
    ENERF_LAUNCH((k_conv3d_wl<CIN, KIND, CTB>), (unsigned)(groups * rt_total), CTB * 192, shmem, st, L.w, L.scale, L.shift, in, out,
                 L.cout, L.relu, B, Di, Hi, Wi, Do, Ho, Wo, rt_total, kdlo, nkd);
    return true;
}

template <int CIN, int KIND>
static bool dispatch_wl(const Conv3dDesc& L, const float* in, float* out, int B, int Di, int Hi, int Wi, int min_blocks, hipStream_t st) {
    constexpr int S = KIND == kConvS2 ? 2 : 1;
    const int Do = S == 2 ? (Di - 1) / 2 + 1 : Di, Ho = S == 2 ? (Hi - 1) / 2 + 1 : Hi, Wo = S == 2 ? (Wi - 1) / 2 + 1 : Wi;
    // the kd taps some output plane reads inside the volume (a contiguous range; all three unless the volume is 1 - 2 planes thick)
    int kdlo = 3, kdhi = -1;
    for (int kd = 0; kd < 3; ++kd)
        for (int d = 0; d < Do; ++d)
            if (S * d + kd - 1 >= 0 && S * d + kd - 1 < Di) { kdlo = kd < kdlo ? kd : kdlo; kdhi = kd > kdhi ? kd : kdhi; break; }
    if (kdhi < kdlo) return false;
    const int nkd = kdhi - kdlo + 1;
    const size_t wbytes = (size_t)nkd * 9 * (CIN / 4) * 256;
    const int rt_total = cdiv(L.cout, 16);
    const long long tiles = cdivl((long long)B * Do * Ho * Wo, 16);
    // the largest block (most sharing of the weight tile) that still leaves min_blocks blocks; the image has to fit beside a second
    // block's (two blocks per CU) unless the layer has no more blocks than CUs
    int ctb = 1;
    for (int c = CIN == 64 ? 2 : 4; c > 1; c >>= 1)      // Cin = 64: 144 B-operand registers, 12 waves would spill
        if (cdivl(tiles, c) * rt_total >= min_blocks) { ctb = c; break; }
    (void)wbytes;
    if constexpr (CIN != 64)                              // (never instantiated for Cin = 64: that kernel spills 56 registers)
        if (ctb == 4) return launch_wl<CIN, KIND, 4>(L, in, out, B, Di, Hi, Wi, Do, Ho, Wo, kdlo, nkd, st);
    switch (ctb) {
        case 2: return launch_wl<CIN, KIND, 2>(L, in, out, B, Di, Hi, Wi, Do, Ho, Wo, kdlo, nkd, st);
        default: return launch_wl<CIN, KIND, 1>(L, in, out, B, Di, Hi, Wi, Do, Ho, Wo, kdlo, nkd, st);
    }
}

#ifndef ENERF_WL_MIN_BLOCKS
#define ENERF_WL_MIN_BLOCKS 400      // measured (level-1 conv4, 640 column tiles x 2 row tiles): 2 tiles per block 14.3 us, 4: 17.6, 1: 16
#endif

// Stride-1 / stride-2 layers with Cin in {16, 32, 64} and Cout a multiple of 16, no skip input.  false: not handled, nothing launched.
bool launch_conv3d_wl(const Conv3dDesc& L, const float* in, float* out, int B, int Di, int Hi, int Wi, hipStream_t st) {
    if (L.kind != kConvS1 && L.kind != kConvS2) return false;
    if (L.cout % 16 != 0 || L.in_planar || L.out_planar) return false;
    if ((long long)B * Di * Hi * Wi * L.cin >= (1LL << 31)) return false;               // 32-bit voxel arithmetic
    const int mb = ENERF_WL_MIN_BLOCKS;
#define ENERF_WL_CASE(C)                                                                                             \
    case C:                                                                                                          \
        return L.kind == kConvS1 ? dispatch_wl<C, kConvS1>(L, in, out, B, Di, Hi, Wi, mb, st)                       \
                                 : dispatch_wl<C, kConvS2>(L, in, out, B, Di, Hi, Wi, mb, st)
    switch (L.cin) {
        ENERF_WL_CASE(16);
        ENERF_WL_CASE(32);
        ENERF_WL_CASE(64);
        default: return false;
    }
#undef ENERF_WL_CASE
}

}  // namespace enerf
