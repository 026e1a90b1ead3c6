// wgrad.hip — weight gradients of the path's convolutions on the matrix cores.
//
// Why first: one training step of dtu_pretrain (512x640, both levels rendered) spent 511 of its 573 ms in the library
// weight-gradient GEMM MIOpen picks for these layers (CK batched_gemm_xdlops_bwd_weight: K = up to 983,040 positions,
// M x N = 8..64 channels — a shape it handles badly).  The layers are tiny in channels and huge in positions, so the
// gradient is a long reduction over positions of 16x16 outer-product tiles: exactly v_mfma_f32_16x16x4_f32 with the
// POSITION as the k index.
//
//   dW[a][b][kd][kh][kw] = sum over positions o of the A grid:  A[a][o] * B[b][o*stride + (kd,kh,kw) - pad]
//
//   Conv{2,3}d           : A = dY (Cout, output grid), B = X  (Cin, input grid)  -> dW laid out (Cout, Cin, k...)
//   ConvTranspose3d (s2) : A = X  (Cin, coarse grid),  B = dY (Cout, fine grid)  -> dW laid out (Cin, Cout, k...)
// (the same index relation: fine = 2*coarse + k - 1), i.e. one kernel for every layer of FeatureNet and both cost-reg nets.
//
// Mapping: both tensors are read channels-last (n, positions, C).  Lane l = (g = l>>4, j = l&15): the A operand element is
// A[position p0+g][channel 16*ta + j] (rows = A channels, k = 4 consecutive positions), the B operand element is
// B[shifted position of p0+g][channel 16*tb + j]; a wave keeps one f32x4 accumulator per kernel tap (<= 27) for its
// (ta, tb) channel-tile pair and walks its share of the positions; the four waves of a block are reduced through LDS and
// the block adds its partial tile set to dW with fp32 atomics.  Out-of-range taps (padding) and channels >= C load zeros.
#include "kernels.h"

namespace enerf {

#define ENERF_MFMA_W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct WgradGeom {
    int n;                       // batch
    int Da, Ha, Wa, Ca;          // A grid (positions enumerated here) and channels
    int Db, Hb, Wb, Cb;          // B grid and channels
    int stride, pad_d, pad_h, pad_w;
    int tiles_a, tiles_b;        // ceil(C/16)
    int chunks;                  // position chunks (blocks per tile pair)
    int lda, ldb;                // floats between consecutive positions of A / B (>= Ca / Cb: column slices of wider rows)
    int bias;                    // 1: B has a virtual column Cb of ones -> dbias[a] = sum_p A[a][p] (the layer's bias gradient)
    long long npos;              // n * Da * Ha * Wa
};

template <int KD, int KH, int KW>
__global__ __launch_bounds__(256) void k_conv_wgrad(const float* __restrict__ A, const float* __restrict__ Bt, WgradGeom q,
                                                    float* __restrict__ dW, float* __restrict__ dbias) {
    constexpr int NT = KD * KH * KW;
    __shared__ float red[NT][256];                      // one wave's partial tiles at a time (27 KB at 27 taps)
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x / q.chunks, chunk = blockIdx.x - pair * q.chunks;
    const int ta = pair / q.tiles_b, tb = pair - ta * q.tiles_b;
    const int ca = ta * 16 + j, cb = tb * 16 + j;
    const bool ca_ok = ca < q.Ca, cb_ok = cb < q.Cb;
    // positions of this wave: groups of 4, interleaved over (chunk, wave) so every block sees the whole volume
    const long long ngroups = cdivl(q.npos, 4);
    const long long stride_g = (long long)q.chunks * 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long grp = (long long)chunk * 4 + wave; grp < ngroups; grp += stride_g) {
        const long long p = grp * 4 + g;
        const bool pv = p < q.npos;
        const unsigned pc = (unsigned)(pv ? p : q.npos - 1);           // npos < 2^31 (checked by the C entry)
        // p -> (b, od, oh, ow), 32-bit
        const unsigned r1 = pc / (unsigned)q.Wa;
        const int ow = (int)(pc - r1 * (unsigned)q.Wa);
        const unsigned r2 = r1 / (unsigned)q.Ha;
        const int oh = (int)(r1 - r2 * (unsigned)q.Ha);
        const int b = (int)(r2 / (unsigned)q.Da), od = (int)(r2 - (unsigned)b * (unsigned)q.Da);
        const float av = (pv && ca_ok) ? A[(long long)pc * q.lda + ca] : 0.f;
        const int id0 = od * q.stride - q.pad_d, ih0 = oh * q.stride - q.pad_h, iw0 = ow * q.stride - q.pad_w;
        const long long bbase = (long long)b * q.Db;
        float bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int kd = t / (KH * KW), kh = (t / KW) % KH, kw = t % KW;
            const int id = id0 + kd, ih = ih0 + kh, iw = iw0 + kw;
            const bool ok = pv && cb_ok && (unsigned)id < (unsigned)q.Db && (unsigned)ih < (unsigned)q.Hb && (unsigned)iw < (unsigned)q.Wb;
            const long long bi = ((bbase + (ok ? id : 0)) * q.Hb + (ok ? ih : 0)) * q.Wb + (ok ? iw : 0);
            bv[t] = ok ? Bt[bi * q.ldb + cb] : 0.f;
            if (NT == 1 && q.bias && cb == q.Cb) bv[t] = pv ? 1.f : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = ENERF_MFMA_W(av, bv[t], acc[t]);
    }
    // block reduction: waves 1..3 hand their tiles to wave 0 through LDS, one wave per round; wave 0 commits
    for (int src = 1; src < 4; ++src) {
        if (wave == src) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[t][r * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] += red[t][r * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[t][r];
                const int a_ch = ta * 16 + 4 * g + r, b_ch = tb * 16 + j;      // D layout: rows 4g+r of column j
                if (a_ch < q.Ca && b_ch < q.Cb && v != 0.f) atomic_add_f32(dW + ((long long)a_ch * q.Cb + b_ch) * NT + t, v);
                if (NT == 1 && q.bias && a_ch < q.Ca && b_ch == q.Cb && v != 0.f) atomic_add_f32(dbias + a_ch, v);
            }
        }
    }
}

template <int KD, int KH, int KW>
static void launch_wgrad_k(const float* A, const float* Bt, const WgradGeom& q, float* dW, float* dbias, hipStream_t st) {
    const unsigned grid = (unsigned)(q.tiles_a * q.tiles_b * q.chunks);
    ENERF_LAUNCH((k_conv_wgrad<KD, KH, KW>), grid, 256, 0, st, A, Bt, q, dW, dbias);
}

bool launch_conv_wgrad(const float* A, const float* Bt, int n, int Da, int Ha, int Wa, int Ca, int Db, int Hb, int Wb, int Cb,
                       int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w, float* dW, hipStream_t st, int lda = 0,
                       int ldb = 0, float* dbias = nullptr) {
    WgradGeom q;
    q.bias = dbias != nullptr;
    q.lda = lda > 0 ? lda : Ca; q.ldb = ldb > 0 ? ldb : Cb;
    q.n = n; q.Da = Da; q.Ha = Ha; q.Wa = Wa; q.Ca = Ca; q.Db = Db; q.Hb = Hb; q.Wb = Wb; q.Cb = Cb;
    q.stride = stride; q.pad_d = pad_d; q.pad_h = pad_h; q.pad_w = pad_w;
    q.tiles_a = cdiv(Ca, 16); q.tiles_b = cdiv(Cb + q.bias, 16);
    q.npos = (long long)n * Da * Ha * Wa;
    // enough blocks to fill the chip a few times over, few enough that the final atomics stay cheap
    const long long groups = cdivl(q.npos, 4);
    long long want = (long long)device_cu_count() * 4 / (q.tiles_a * q.tiles_b);
    if (want < 1) want = 1;
    const long long maxc = cdivl(groups, 4 * 8);                     // >= 8 position groups per wave
    q.chunks = (int)(want < maxc ? want : (maxc < 1 ? 1 : maxc));
    if (kd == 3 && kh == 3 && kw == 3) { launch_wgrad_k<3, 3, 3>(A, Bt, q, dW, dbias, st); return true; }
    if (kd == 1 && kh == 3 && kw == 3) { launch_wgrad_k<1, 3, 3>(A, Bt, q, dW, dbias, st); return true; }
    if (kd == 1 && kh == 5 && kw == 5) { launch_wgrad_k<1, 5, 5>(A, Bt, q, dW, dbias, st); return true; }
    if (kd == 1 && kh == 1 && kw == 1) { launch_wgrad_k<1, 1, 1>(A, Bt, q, dW, dbias, st); return true; }
    return false;
}

// ---------------------------------------------------------------------------------------------------------------------
// Linear-layer weight gradients: grad_w[a][b] = sum_p A[p][a] * B[p][b] with P up to ~2M rows and 1..89 columns per side.
// k_conv_wgrad<1,1,1> gives every (16x16) tile pair its own blocks, so the rows of A are re-read tiles_b times and those
// of B tiles_a times (3 KB per row for the 64 x 88 layer).  Here one wave keeps ALL TA x TB tile accumulators and walks its
// share of the rows once: TA + TB loads and TA*TB MFMAs per 4 rows, every row read exactly once (612 B for 64 x 89).
// ---------------------------------------------------------------------------------------------------------------------
template <int TA, int TB>
__global__ __launch_bounds__(256) void k_gemm_wgrad(const float* __restrict__ A, const float* __restrict__ Bt, int lda, int ldb,
                                                    int Ca, int Cb, int bias, long long P, float* __restrict__ dW,
                                                    float* __restrict__ dbias) {
    constexpr int NT = TA * TB;
    __shared__ float red[NT][256];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long ngroups = cdivl(P, 4);
    for (long long grp = (long long)blockIdx.x * 4 + wave; grp < ngroups; grp += (long long)gridDim.x * 4) {
        const long long p = grp * 4 + g;
        const bool pv = p < P;
        const long long pc = pv ? p : P - 1;
        float av[TA], bv[TB];
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) {
            const int ca = ta * 16 + j;
            av[ta] = (pv && ca < Ca) ? A[pc * lda + ca] : 0.f;
        }
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int cb = tb * 16 + j;
            bv[tb] = (pv && cb < Cb) ? Bt[pc * ldb + cb] : 0.f;
            if (bias && cb == Cb) bv[tb] = pv ? 1.f : 0.f;           // virtual all-ones column -> bias gradient
        }
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) acc[ta * TB + tb] = ENERF_MFMA_W(av[ta], bv[tb], acc[ta * TB + tb]);
    }
    for (int src = 1; src < 4; ++src) {                               // waves 1..3 hand their tiles to wave 0
        if (wave == src) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[t][r * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] += red[t][r * 64 + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[ta * TB + tb][r];
                const int a_ch = ta * 16 + 4 * g + r, b_ch = tb * 16 + j;
                if (a_ch >= Ca || v == 0.f) continue;
                if (b_ch < Cb) atomic_add_f32(dW + (long long)a_ch * Cb + b_ch, v);
                else if (bias && b_ch == Cb) atomic_add_f32(dbias + a_ch, v);
            }
}

template <int TA>
static bool launch_gemm_wgrad_ta(int tb, unsigned grid, hipStream_t st, const float* A, const float* Bt, int lda, int ldb, int Ca,
                                 int Cb, int bias, long long P, float* dW, float* dbias) {
#define ENERF_GW(TBV) case TBV: ENERF_LAUNCH((k_gemm_wgrad<TA, TBV>), grid, 256, 0, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias); return true;
    switch (tb) { ENERF_GW(1) ENERF_GW(2) ENERF_GW(3) ENERF_GW(4) ENERF_GW(5) ENERF_GW(6) default: return false; }
#undef ENERF_GW
}
// all tile pairs in one wave when they fit (<= 4 x 6 tiles); false -> the caller falls back to k_conv_wgrad<1,1,1>
bool launch_gemm_wgrad(const float* A, int lda, int Ca, const float* Bt, int ldb, int Cb, long long P, float* dW, float* dbias,
                       hipStream_t st) {
    const int ta = cdiv(Ca, 16), tb = cdiv(Cb + (dbias != nullptr), 16);
    if (ta > 4 || tb > 6) return false;
    const long long groups = cdivl(P, 4);
    long long blocks = cdivl(groups, 4 * 16);                         // >= 16 row groups per wave
    const long long cap = (long long)device_cu_count() * 2;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const int bias = dbias != nullptr;
    switch (ta) {
        case 1: return launch_gemm_wgrad_ta<1>(tb, (unsigned)blocks, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias);
        case 2: return launch_gemm_wgrad_ta<2>(tb, (unsigned)blocks, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias);
        case 3: return launch_gemm_wgrad_ta<3>(tb, (unsigned)blocks, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias);
        default: return launch_gemm_wgrad_ta<4>(tb, (unsigned)blocks, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias);
    }
}

}  // namespace enerf

using namespace enerf;
extern "C" int enerf_conv_wgrad(const float* a_cl, const float* b_cl, int n, int Da, int Ha, int Wa, int Ca, int Db, int Hb, int Wb,
                                int Cb, int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w, float* grad_w,
                                enerf_stream_t stream) {
    REQUIRE(a_cl && b_cl && grad_w, "conv_wgrad: null pointer");
    REQUIRE(n > 0 && Da > 0 && Ha > 0 && Wa > 0 && Ca > 0 && Db > 0 && Hb > 0 && Wb > 0 && Cb > 0 && stride >= 1,
            "conv_wgrad: bad shape");
    REQUIRE((long long)n * Da * Ha * Wa < (1LL << 31) && (long long)n * Db * Hb * Wb < (1LL << 31), "conv_wgrad: more than 2^31 positions");
    zero_async(grad_w, (size_t)Ca * Cb * kd * kh * kw * sizeof(float), (hipStream_t)stream);
    if (!launch_conv_wgrad(a_cl, b_cl, n, Da, Ha, Wa, Ca, Db, Hb, Wb, Cb, kd, kh, kw, stride, pad_d, pad_h, pad_w, grad_w,
                           (hipStream_t)stream))
        return fail(ENERF_EINVAL, "conv_wgrad: kernel %dx%dx%d unsupported (3x3x3, 1x3x3, 1x5x5, 1x1x1)", kd, kh, kw);
    return check_launch("conv_wgrad");
}

// Plain position-reduction GEMM: grad_w[a][b] = sum_p A[p][a] * B[p][b] — the weight gradient of a Linear layer from its
// pre-activation gradient A (P rows, Ca used columns of rows lda floats wide) and its input B (P x Cb, rows ldb wide).
// grad_bias (nullable): the layer's bias gradient sum_p A[p][a], from the same pass (a virtual all-ones column of B).
extern "C" int enerf_gemm_wgrad(const float* a, int lda, int Ca, const float* b, int ldb, int Cb, long long P, float* grad_w,
                                float* grad_bias, enerf_stream_t stream) {
    REQUIRE(a && b && grad_w && Ca > 0 && Cb > 0 && lda >= Ca && ldb >= Cb, "gemm_wgrad: bad arguments");
    REQUIRE(P > 0 && P < (1LL << 31), "gemm_wgrad: P out of range");
    zero_async(grad_w, (size_t)Ca * Cb * sizeof(float), (hipStream_t)stream);
    if (grad_bias) zero_async(grad_bias, (size_t)Ca * sizeof(float), (hipStream_t)stream);
    if (!launch_gemm_wgrad(a, lda, Ca, b, ldb, Cb, P, grad_w, grad_bias, (hipStream_t)stream))
        launch_conv_wgrad(a, b, 1, 1, 1, (int)P, Ca, 1, 1, (int)P, Cb, 1, 1, 1, 1, 0, 0, 0, grad_w, (hipStream_t)stream, lda, ldb, grad_bias);
    return check_launch("gemm_wgrad");
}
