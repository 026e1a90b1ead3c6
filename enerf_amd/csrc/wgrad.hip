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

#ifndef ENERF_WGRAD_PREFETCH
#define ENERF_WGRAD_PREFETCH 1            /* tiled 2-D kernel: the next tile's global loads are issued before this tile's MFMAs */
#endif
#ifndef ENERF_WGRAD3D_PREFETCH
#define ENERF_WGRAD3D_PREFETCH 1          /* the same in the 3-D kernel */
#endif
#ifndef ENERF_WGRAD_REDUCE_8
#define ENERF_WGRAD_REDUCE_8 1            /* k_wgrad_reduce: eight (1) or four (0) loads in flight per thread */
#endif

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
    unsigned abytes, bbytes;     // byte sizes of the A / B tensors (< 2^32: raw buffer loads, 32-bit lane offsets)
    long long npos;              // n * Da * Ha * Wa
};

// SPLIT: the taps are split (by their leading kernel dimension) over SPLIT waves of a block, PL position-lanes each: a wave keeps
// NT/SPLIT accumulators (36 instead of 108 registers at 3x3x3) and 1 + NT/SPLIT loads per group in flight, so six to eight
// waves fit a SIMD instead of two and the loads' latency hides (conv0 / heads, 655k positions: 311-326 -> see DESIGN.md).
template <int KD, int KH, int KW, int SPLIT, int PL>
__global__ __launch_bounds__(SPLIT * PL * 64) void k_conv_wgrad(const float* __restrict__ A, const float* __restrict__ Bt, WgradGeom q,
                                                                float* __restrict__ dW, float* __restrict__ dbias,
                                                                float* __restrict__ scratch) {
    constexpr int NT = KD * KH * KW, NTW = NT / SPLIT;
    static_assert(SPLIT == 1 || (KD > 1 && SPLIT == KD) || (KD == 1 && SPLIT == KH), "the tap split is the leading kernel dimension");
    __shared__ float red[PL > 1 ? NT : 1][256];         // partial tiles of the position-lanes >= 1, one lane at a time
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    const int ts = __builtin_amdgcn_readfirstlane(wave % SPLIT), pl = wave / SPLIT;     // tap subset (scalar), position lane
    const int pair = blockIdx.x / q.chunks, chunk = blockIdx.x - pair * q.chunks;
    const int ta = pair / q.tiles_b, tb = pair - ta * q.tiles_b;
    const int ca = ta * 16 + j, cb = tb * 16 + j;
    const bool ca_ok = ca < q.Ca, cb_ok = cb < q.Cb;
    // positions of this wave: groups of 4, interleaved over (chunk, position lane) so every block sees the whole volume
    const long long ngroups = cdivl(q.npos, 4);
    const long long stride_g = (long long)q.chunks * PL;
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // One group = 4 positions x (1 A value + NTW shifted B values per lane) -> NTW MFMAs.  Every load is an UNCONDITIONAL raw
    // buffer load: `scalar base + 32-bit lane offset`, and a padding tap / dead lane sets its offset out of range, which reads 0.
    // Per tap that is one add (uniform tap delta) + one select on the offset; the validity of a tap is the AND of three per-axis
    // lane masks that live in scalar registers.  (History: `ok ? B[i] : 0` made hipcc branch around each load and wait on it — 27
    // serial round trips per group; unconditional loads from 64-bit addresses with a zero page cost ~15 VALU of address math
    // per tap, ~400 per group against 27 MFMAs = 864 cycles, and VALU and MFMA serialise on gfx950.)  The loads of group i+1 are
    // issued before the MFMAs of group i.
    const BufRsrc ra = buf_rsrc(A, q.abytes), rb = buf_rsrc(Bt, q.bbytes);
    const int dw4 = q.ldb * 4, dh4 = q.Wb * dw4, dd4 = q.Hb * dh4;      // byte deltas of one step along w / h / d of the B grid (uniform)
    auto issue = [&](long long grp, float& av, float (&bv)[NTW]) {
        const unsigned p = (unsigned)grp * 4u + (unsigned)g;            // npos < 2^31 (checked by the C entry)
        const bool pv = p < (unsigned)q.npos;
        const unsigned pc = pv ? p : (unsigned)q.npos - 1u;
        // p -> (b, od, oh, ow), 32-bit
        const unsigned r1 = pc / (unsigned)q.Wa;
        const int ow = (int)(pc - r1 * (unsigned)q.Wa);
        const unsigned r2 = r1 / (unsigned)q.Ha;
        const int oh = (int)(r1 - r2 * (unsigned)q.Ha);
        const int b = (int)(r2 / (unsigned)q.Da), od = (int)(r2 - (unsigned)b * (unsigned)q.Da);
        av = buf_load_f32(ra, (pv && ca_ok) ? (pc * (unsigned)q.lda + (unsigned)ca) * 4u : 0xffffffffu);
        const int id0 = od * q.stride - q.pad_d, ih0 = oh * q.stride - q.pad_h, iw0 = ow * q.stride - q.pad_w;
        // byte offset of tap (0,0,0)'s element (may be "negative" at the borders: those taps are masked out below)
        const int off0 = ((((b * q.Db + id0) * q.Hb + ih0) * q.Wb + iw0) * q.ldb + cb) * 4;
        const bool lane_ok = pv && cb_ok;
        bool vd[KD], vh[KH], vw[KW];
#pragma unroll
        for (int k = 0; k < KD; ++k) vd[k] = lane_ok && (unsigned)(id0 + k) < (unsigned)q.Db;
#pragma unroll
        for (int k = 0; k < KH; ++k) vh[k] = (unsigned)(ih0 + k) < (unsigned)q.Hb;
#pragma unroll
        for (int k = 0; k < KW; ++k) vw[k] = (unsigned)(iw0 + k) < (unsigned)q.Wb;
#pragma unroll
        for (int k = 0; k < NTW; ++k) {
            // tap (kd, kh, kw) of this wave's k-th accumulator: the split is by the leading kernel dimension, so only that
            // coordinate depends on the wave (a scalar) and the others are compile-time constants of k
            int kd, kh, kw;
            if (SPLIT == 1) { kd = k / (KH * KW); kh = (k / KW) % KH; kw = k % KW; }
            else if (KD > 1) { kd = ts; kh = k / KW; kw = k % KW; }          // SPLIT == KD
            else { kd = 0; kh = ts; kw = k; }                                // SPLIT == KH
            bool ok;
            if (SPLIT == 1) ok = vd[kd] && vh[kh] && vw[kw];
            else if (KD > 1) ok = lane_ok && (unsigned)(id0 + kd) < (unsigned)q.Db && vh[kh] && vw[kw];
            else ok = lane_ok && (unsigned)(ih0 + kh) < (unsigned)q.Hb && vw[kw];
            const int off = off0 + kd * dd4 + kh * dh4 + kw * dw4;
            bv[k] = buf_load_f32(rb, ok ? (unsigned)off : 0xffffffffu);
        }
        if (NT == 1 && q.bias && cb == q.Cb) bv[0] = pv ? 1.f : 0.f;
    };
    float av0, av1, bv0[NTW], bv1[NTW];
    long long grp = (long long)chunk * PL + pl;
    // The unsplit 9- / 27-tap kernels run WITHOUT the cross-group prefetch: two groups in flight cost 182 registers next to the
    // 108 accumulators (one wave per SIMD); one group at a time fits two to three waves, which hide the loads better
    // (measured, 655k positions: 223 -> 130 us).  The tap-split and 1x1x1 kernels keep the prefetch (8 waves per SIMD anyway).
    if (SPLIT == 1 && NT > 1) {
        for (; grp < ngroups; grp += stride_g) {
            issue(grp, av0, bv0);
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t] = ENERF_MFMA_W(av0, bv0[t], acc[t]);
        }
    }
    if (grp < ngroups) issue(grp, av0, bv0);
    for (; grp < ngroups; grp += 2 * stride_g) {
        const bool more1 = grp + stride_g < ngroups;                   // uniform
        if (more1) issue(grp + stride_g, av1, bv1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = ENERF_MFMA_W(av0, bv0[t], acc[t]);
        if (!more1) break;
        if (grp + 2 * stride_g < ngroups) issue(grp + 2 * stride_g, av0, bv0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = ENERF_MFMA_W(av1, bv1[t], acc[t]);
    }
    // block reduction: position lanes 1.. hand their tiles to lane 0 of the same tap subset through LDS, one lane per round
    for (int src = 1; src < PL; ++src) {
        if (pl == src) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[PL > 1 ? ts * NTW + t : 0][r * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (pl == 0) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] += red[PL > 1 ? ts * NTW + t : 0][r * 64 + lane];
        }
        __syncthreads();
    }
    if (pl != 0) return;
    if (scratch != nullptr) {       // two-stage commit: this block's partial tiles, plain coalesced stores (k_wgrad_reduce sums them)
        float* sp = scratch + ((long long)blockIdx.x * NT + ts * NTW) * 256 + lane;       // blockIdx.x = pair * chunks + chunk
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[t * 256 + r * 64] = acc[t][r];
        return;
    }
#pragma unroll
    for (int k = 0; k < NTW; ++k) {
        const int t = ts * NTW + k;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[k][r];
            const int a_ch = ta * 16 + 4 * g + r, b_ch = tb * 16 + j;      // D layout: rows 4g+r of column j
            if (a_ch < q.Ca && b_ch < q.Cb && v != 0.f) atomic_add_f32(dW + ((long long)a_ch * q.Cb + b_ch) * NT + t, v);
            if (NT == 1 && q.bias && a_ch < q.Ca && b_ch == q.Cb && v != 0.f) atomic_add_f32(dbias + a_ch, v);
        }
    }
}

// Second stage of the weight-gradient commit.  With fp32 atomics every block adds its NT x 256 partial sums to the SAME few KB
// of dW: 640-1024 blocks x 6912 atomics onto 6912 addresses run at ~13 G atomics/s — measured 310-580 us per cost-volume layer
// against 7-58 us of MFMA issue time, 7 ms of a 28 ms training step.  Instead every block stores its partial tile set and this
// kernel sums them: item I = (pair or 0, tile/tap t), one 1024-thread block per (item, register r): wave w sums the chunks
// c = w, w+16, ... of its 64 lanes' element, the 16 waves meet in LDS, wave 0 writes dW (and dbias) with plain stores —
// deterministic, and no pre-zeroing of dW.
//   conv (gemm == 0): scratch[((pair*chunks + c)*nt + t)*256 + e], pair = (ta, tb), t = tap, dW[(a*Cb + b)*nt + t]
//   gemm (gemm == 1): scratch[(c*nt + t)*256 + e], t = ta*tiles_b + tb, dW[a*Cb + b]
// this thread's element summed over the chunks (wave w takes c = w, w + 16, ...; the 16 waves meet in LDS; the value is valid in wave 0)
__device__ __forceinline__ float wgrad_reduce_sum(const float* __restrict__ sp, int chunks, long long cstride, float (*red)[64]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    int c = w;
#if ENERF_WGRAD_REDUCE_8
    for (; c + 112 < chunks; c += 128) {                    // eight independent loads in flight per thread (the kernel is a chain of
        s0 += sp[c * cstride]; s1 += sp[(c + 16) * cstride]; s2 += sp[(c + 32) * cstride]; s3 += sp[(c + 48) * cstride];   // L2 round trips)
        s4 += sp[(c + 64) * cstride]; s5 += sp[(c + 80) * cstride]; s6 += sp[(c + 96) * cstride]; s7 += sp[(c + 112) * cstride];
    }
#endif
    for (; c + 48 < chunks; c += 64) {                      // four
        s0 += sp[c * cstride]; s1 += sp[(c + 16) * cstride]; s2 += sp[(c + 32) * cstride]; s3 += sp[(c + 48) * cstride];
    }
    for (; c < chunks; c += 16) s0 += sp[c * cstride];
    red[w][lane] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    __syncthreads();
    float v = 0.f;
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][lane];
    }
    return v;
}
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ scratch, int chunks, int nt, int tiles_b, int Ca, int Cb, int bias,
                                                  int gemm, float* __restrict__ dW, float* __restrict__ dbias, int bid, float (*red)[64]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const int r = bid & 3, item = bid >> 2;
    const int pair = gemm ? 0 : item / nt, t = gemm ? item : item - pair * nt;
    const float* sp = scratch + (((long long)pair * chunks) * nt + t) * 256 + r * 64 + lane;
    const float v = wgrad_reduce_sum(sp, chunks, (long long)nt * 256, red);
    if (w != 0) return;
    const int ta = gemm ? t / tiles_b : pair / tiles_b, tb = gemm ? t - ta * tiles_b : pair - ta * tiles_b;
    const int a_ch = ta * 16 + 4 * g + r, b_ch = tb * 16 + j;          // D layout: rows 4g+r of column j
    if (a_ch >= Ca) return;
    if (b_ch < Cb) dW[gemm ? (long long)a_ch * Cb + b_ch : ((long long)a_ch * Cb + b_ch) * nt + t] = v;
    else if (bias && b_ch == Cb && (gemm || nt == 1)) dbias[a_ch] = v;
}
__global__ __launch_bounds__(1024) void k_wgrad_reduce(const float* __restrict__ scratch, int chunks, int nt, int tiles_b, int Ca,
                                                       int Cb, int bias, int gemm, float* __restrict__ dW,
                                                       float* __restrict__ dbias) {
    __shared__ float red[16][64];
    wgrad_reduce_body(scratch, chunks, nt, tiles_b, Ca, Cb, bias, gemm, dW, dbias, (int)blockIdx.x, red);
}
// out[.] = sum over the rows c < chunks of part[c * n_out + i], fixed order: 16 waves take rows c = w, w + 16, ... (four loads in
// flight), then wave 0 adds the 16 partial sums.  by = segment: its rows start at part + by * chunks * n_out and element i
// goes to out[(i / inner) * ostride + by * inner + i % inner] (channel halves of a wider layer; one segment, inner = n_out: out[i]).
__device__ __forceinline__ void colsum_body(const float* __restrict__ part, int chunks, int n_out, int inner, int ostride, float* __restrict__ out,
                                            int bx, int by, float (*red)[64]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = bx * 64 + lane;
    const bool live = i < n_out;
    const float* sp = part + (long long)by * chunks * n_out + (live ? i : 0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = w;
    for (; c + 48 < chunks; c += 64) {
        s0 += sp[(long long)c * n_out]; s1 += sp[(long long)(c + 16) * n_out];
        s2 += sp[(long long)(c + 32) * n_out]; s3 += sp[(long long)(c + 48) * n_out];
    }
    for (; c < chunks; c += 16) s0 += sp[(long long)c * n_out];
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w != 0 || !live) return;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][lane];
    const int o = i / inner;
    out[(long long)o * ostride + by * inner + (i - o * inner)] = v;
}
__global__ __launch_bounds__(1024) void k_colsum(const float* __restrict__ part, int chunks, int n_out, int inner, int ostride,
                                                 float* __restrict__ out) {
    __shared__ float red[16][64];
    colsum_body(part, chunks, n_out, inner, ostride, out, (int)blockIdx.x, (int)blockIdx.y, red);
}

// ---- deferred second stages (round 6) ----
// Every weight gradient of a network's backward pass ends with one of the two small kernels above (k_wgrad_reduce / k_colsum: 30 launches
// of ~5 us per training step) and nothing reads a weight gradient before the pass is over.  Between enerf_wgrad_reduce_begin() and
// enerf_wgrad_reduce_flush(stream) on the calling thread the second stages are RECORDED instead of launched (the caller keeps the
// workspaces alive) and the flush runs them all as one kernel: block -> (recorded reduction, its own block id); same code, same bits.
struct ReduceDesc {
    const float* scratch;
    float *dW, *dbias;
    int kind;                          // 0: wgrad_reduce_body, 1: colsum_body
    int a0, a1, a2, a3, a4, a5, a6;    // kind 0: chunks, nt, tiles_b, Ca, Cb, bias, gemm;  kind 1: chunks, n_out, inner, ostride, grid x
    int blocks, block0;
};
constexpr int kReduceBatchMax = 24;
struct ReduceBatch {
    ReduceDesc d[kReduceBatchMax];
    int n;
};
__global__ __launch_bounds__(1024) void k_wgrad_reduce_batch(ReduceBatch G) {
    __shared__ float red[16][64];
    int i = 0;
    while (i + 1 < G.n && (int)blockIdx.x >= G.d[i + 1].block0) ++i;
    const ReduceDesc& d = G.d[i];
    const int lb = (int)blockIdx.x - d.block0;
    if (d.kind == 0) wgrad_reduce_body(d.scratch, d.a0, d.a1, d.a2, d.a3, d.a4, d.a5, d.a6, d.dW, d.dbias, lb, red);
    else colsum_body(d.scratch, d.a0, d.a1, d.a2, d.a3, d.dW, lb % d.a4, lb / d.a4, red);
}
static thread_local bool g_reduce_deferring = false;
static thread_local int g_reduce_pending = 0;
static thread_local ReduceDesc g_reduce_list[4 * kReduceBatchMax];
static bool reduce_defer(const ReduceDesc& d) {
    if (!g_reduce_deferring || g_reduce_pending >= 4 * kReduceBatchMax) return false;
    g_reduce_list[g_reduce_pending++] = d;
    return true;
}
static void launch_wgrad_reduce(const float* scratch, int pairs, int chunks, int nt, int tiles_b, int Ca, int Cb, int bias, int gemm,
                                float* dW, float* dbias, hipStream_t st) {
    const ReduceDesc d = {scratch, dW, dbias, 0, chunks, nt, tiles_b, Ca, Cb, bias, gemm, pairs * nt * 4, 0};
    if (reduce_defer(d)) return;
    ENERF_LAUNCH(k_wgrad_reduce, (unsigned)(pairs * nt * 4), 1024, 0, st, scratch, chunks, nt, tiles_b, Ca, Cb, bias, gemm, dW, dbias);
}
static void launch_colsum(const float* part, int chunks, int n_out, int inner, int ostride, float* out, int gx, int gy, hipStream_t st) {
    const ReduceDesc d = {part, out, nullptr, 1, chunks, n_out, inner, ostride, gx, 0, 0, gx * gy, 0};
    if (reduce_defer(d)) return;
    ENERF_LAUNCH(k_colsum, dim3((unsigned)gx, (unsigned)gy), 1024, 0, st, part, chunks, n_out, inner, ostride, out);
}
static void reduce_flush(hipStream_t st) {
    for (int first = 0; first < g_reduce_pending; first += kReduceBatchMax) {
        ReduceBatch G;
        G.n = g_reduce_pending - first < kReduceBatchMax ? g_reduce_pending - first : kReduceBatchMax;
        int blocks = 0;
        for (int i = 0; i < G.n; ++i) {
            G.d[i] = g_reduce_list[first + i];
            G.d[i].block0 = blocks;
            blocks += G.d[i].blocks;
        }
        ENERF_LAUNCH(k_wgrad_reduce_batch, (unsigned)blocks, 1024, 0, st, G);
    }
    g_reduce_pending = 0;
    g_reduce_deferring = false;
}

template <int KD, int KH, int KW, int SPLIT, int PL>
static void launch_wgrad_k(const float* A, const float* Bt, const WgradGeom& q, float* dW, float* dbias, float* scratch, hipStream_t st) {
    const unsigned grid = (unsigned)(q.tiles_a * q.tiles_b * q.chunks);
    ENERF_LAUNCH((k_conv_wgrad<KD, KH, KW, SPLIT, PL>), grid, SPLIT * PL * 64, 0, st, A, Bt, q, dW, dbias, scratch);
    if (scratch != nullptr)
        launch_wgrad_reduce(scratch, q.tiles_a * q.tiles_b, q.chunks, KD * KH * KW, q.tiles_b, q.Ca, q.Cb, q.bias, 0, dW, dbias, st);
}
// position lanes per block for a kernel shape (the tap split is the leading kernel dimension; 1x1x1: four position lanes)
// measured (MI355X, config-5 shapes, tools/gpu_wgrad_variants.sh): the tap split wins on the layers with few positions
// (<= 10k: 32 -> 16-20 us) and loses from 82k positions on (three waves re-read every A value and row), which keep one
// wave per group
#ifndef ENERF_WGRAD_SPLIT_BELOW
#define ENERF_WGRAD_SPLIT_BELOW 50000           /* A/B builds: 0 = never split 3x3(x3), a huge value = always */
#endif
static bool wgrad_split(int taps, long long npos) { return taps == 25 || ((taps == 27 || taps == 9) && npos < (long long)ENERF_WGRAD_SPLIT_BELOW); }
static int wgrad_pl(int taps, long long npos) { return taps == 25 ? 1 : (wgrad_split(taps, npos) ? 2 : 4); }
// position chunks (= blocks per channel-tile pair): enough blocks to fill the chip, >= 8 position groups per wave
static int wgrad_chunks(long long npos, int pairs, int taps, bool two_stage) {
    const long long groups = cdivl(npos, 4);
    long long want = (long long)device_cu_count() * (two_stage ? 2 : 4) / pairs;
    if (want < 1) want = 1;
    const long long maxc = cdivl(groups, wgrad_pl(taps, npos) * 8);
    return (int)(want < maxc ? want : (maxc < 1 ? 1 : maxc));
}
#ifndef ENERF_WGRAD2D_BPC
#define ENERF_WGRAD2D_BPC 4                      /* persistent blocks per CU (112 VGPRs: four waves per SIMD; 21 KB of LDS each) */
#endif
#ifndef ENERF_WGRAD2D_TILE
#define ENERF_WGRAD2D_TILE 1                     /* 0: these layers stay on k_conv_wgrad (A/B builds) */
#endif
#ifndef ENERF_WGRAD3D_TILE
#define ENERF_WGRAD3D_TILE 1
#endif
#ifndef ENERF_WGRAD_1X1_GEMM
#define ENERF_WGRAD_1X1_GEMM 1                   /* 0: 1x1 layers stay on k_conv_wgrad<1,1,1> (A/B builds) */
#endif
size_t gemm_wgrad_workspace_bytes(long long P, int Ca, int Cb, int bias);
bool launch_gemm_wgrad(const float* A, int lda, int Ca, const float* Bt, int ldb, int Cb, long long P, float* dW, float* dbias,
                       hipStream_t st, void* workspace, size_t workspace_bytes);
size_t conv_wgrad_workspace_bytes(long long npos, int Ca, int Cb, int taps, int bias) {
    const int pairs = cdiv(Ca, 16) * cdiv(Cb + bias, 16);
    const size_t two_stage = (size_t)pairs * wgrad_chunks(npos, pairs, taps, true) * taps * 256 * sizeof(float);
    // k_wgrad2d_3x3_c8 (below): one compact row of Ca*Cb*9 sums per persistent block (the query does not know the grid: the bound)
    const size_t tiled = (ENERF_WGRAD2D_TILE && taps == 9 && Ca <= 32 && Cb <= 32 && !bias)
                             ? (size_t)device_cu_count() * (Ca <= 8 ? ENERF_WGRAD2D_BPC : 3) * Ca * Cb * 9 * sizeof(float) : 0;
    // k_wgrad3d_c8: at most 2 blocks per CU in all, one row of the whole dW each
    const size_t tiled3 = (ENERF_WGRAD3D_TILE && taps == 27 && !bias && ((Ca <= 8 && Cb <= 32) || (Ca == 16 && Cb <= 8)))
                              ? (size_t)device_cu_count() * 2 * Ca * Cb * 27 * sizeof(float) : 0;
    size_t t = two_stage > tiled ? two_stage : tiled;
    t = t > tiled3 ? t : tiled3;
    // a 1x1 stride-1 layer is a plain position-reduction GEMM (k_gemm_wgrad: all tile pairs in one wave, rows read once)
    if (ENERF_WGRAD_1X1_GEMM && taps == 1 && cdiv(Ca, 16) <= 4 && cdiv(Cb + bias, 16) <= 6) {
        const size_t gm = gemm_wgrad_workspace_bytes(npos, Ca, Cb, bias);
        t = t > gm ? t : gm;
    }
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the 3x3 stride-1 2-D layers with <= 8 gradient channels (FeatureNet conv0.0 3 -> 8, conv0.1 8 -> 8, smooth0 32 -> 8:
// full resolution, 983k positions each at dtu_pretrain) from LDS tiles, three kernel rows per MFMA.
//
// k_conv_wgrad spends one buffer load (offset add + select) per MFMA and three integer divisions per group of nine: these
// layers ran at 0.12 - 0.15 of the matrix pipe (200 - 260 us each against a 29 us issue floor), and a 16 x 16 tile with 8 x 8
// live channels wastes three quarters of every MFMA on top.  Here a block stages a (TH + 1) x TW tile of A and the haloed
// (TH + 2) x (TW + 2) tile of B in LDS once (zeros outside the image) and a tap's operand is one ds_read_b32 at an immediate
// offset: no address arithmetic per tap, no masks.  And the idle halves of the tile carry SHIFTED copies:
//     A operand row (a, s):  A[a][(y + s, x)]                 s = 0, 1   (the next image row)
//     B operand col (b, u):  B[b][(y + 2u - 1, x + kw - 1)]   u = 0, 1   (two image rows down)
//     D[(a, s)][(b, u)] = sum_p A[a][p + s W] B[b][p + (2u - 1) W + kw - 1] = dW[a][b][kh = 2u - s][kw]
// so ONE MFMA per kw yields kh = 0, 1, 2 (and one discarded quadrant, kh = -1): 3 MFMAs and 4 LDS reads per group of four
// positions instead of 9 and 10.  Every output position o must be counted once per s: the tiles start one row ABOVE the image
// (y0 = -1 + ty TH), where A is zero, so s = 1 reaches row 0 and s = 0 loses nothing.  Row pitches are chosen so that the 64
// lanes of either operand read hit 64 different banks (A: pitch = 32 mod 64 floats; B: 2 x pitch = 32 mod 64).
// Blocks are persistent over tiles and end with one compact row of Ca*Cb*9 partial sums (dW's own order); k_colsum adds the
// rows in a fixed order: deterministic, no atomics.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kW2TH = 8, kW2TW = 32;
// LDS pitches (floats).  ds_read_b32 / ds_read2_b32 are serviced in the two 32-lane halves of a wave with bank = (a / 4) mod 32
// (MI355X_MICROARCH.md, LDS): the lanes of a half are (g in {0,1} or {2,3}, lo 0..7, hi 0..1) at g * 8 + lo + hi * (shift), so the
// shifted copy must sit 16 banks away: A rows ≡ 16 (mod 32), TWO B rows ≡ 16 (mod 32).  (First version: 32 mod 64 — right for a
// 64-bank model, a 2-way conflict on every operand read with 32: PMC counted SQ_LDS_BANK_CONFLICT at 0.42 - 0.55 of the LDS-active
// cycles, profiles/r05_micro_wgrad_pmc.txt.)  The 8-lane groups of the ds_write_b128 staging stores cover one pixel's quads in
// four channel planes: the plane stride ≡ 8 (mod 32) keeps those apart too.
#ifndef ENERF_WGRAD_PITCH32
#define ENERF_WGRAD_PITCH32 1                    /* 0: the first version's pitches (A/B builds) */
#endif
constexpr int kW2BCols = kW2TW + 2;
constexpr int kW2APitch = kW2TW * 8 + (ENERF_WGRAD_PITCH32 ? 16 : 32), kW2BPitch = kW2BCols * 8 + (ENERF_WGRAD_PITCH32 ? 8 : 0);
constexpr int kW2BPlane = (kW2TH + 2) * kW2BPitch + (ENERF_WGRAD_PITCH32 ? 8 : 0);
static_assert(!ENERF_WGRAD_PITCH32 || (kW2APitch % 32 == 16 && (2 * kW2BPitch) % 32 == 16 && kW2BPlane % 32 == 24), "bank-conflict-free operand reads / staging stores");

// one tile of `src` (n, H, W, ld floats per position; C <= 8 NCB channels used) into LDS: rows y0 .., cols x0 .., zeros outside
// the image; one plane of PLANE floats per block of 8 channels, pixel pitch 8 floats inside a plane (so that the 64 lanes of an
// operand read hit 64 banks whatever C is).  Two halves, so that the loads of the NEXT tile are in flight while this tile's MFMAs
// run (a block spent ~5 us per tile waiting for them in front of 0.6 - 2.6 us of MFMAs): fetch() = the global loads into registers,
// commit() = the LDS stores.  V4: 16-byte loads (C a power of two >= 4, ld % 4 == 0, 16-byte aligned base); otherwise (the
// 3-channel image of conv0.0) a scalar loop at commit time, not pipelined.
template <int ROWS, int COLS, int PITCH, int PLANE, int NCB, bool V4, int CPP = 8>      // CPP: channels per plane and pixel (8 or 16)
struct W2Stage {
    static constexpr int QPP = CPP / 4;
    static constexpr int NIT = V4 ? (ROWS * COLS * QPP * NCB + 255) / 256 : 1;
    float4 v[NIT];
    int key[NIT];                                          // tile-invariant slot of this thread: (r << 20) | (c << 8) | quad, -1 = none
    unsigned koff[NIT];                                    // ... and its byte offset from the tile's first element: ((r W + c) ld + 4 quad) 4
    const float* src; int img, H, W, C, ld, y0, x0;        // (scalar path)
    __device__ __forceinline__ void init(int C_, int W_, int ld_) {
        C = C_;
        if (!V4) return;
        const int nq = C >> 2, sh = nq >= 8 ? 3 : nq >= 4 ? 2 : nq >= 2 ? 1 : 0;       // quads per pixel (1, 2, 4, 8)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = (int)threadIdx.x + it * 256;
            const int pos = i >> sh, qd = i & (nq - 1);
            const int r = pos / COLS, c = pos - r * COLS;
            key[it] = pos < ROWS * COLS ? (r << 20) | (c << 8) | qd : -1;
            koff[it] = (unsigned)(((r * W_ + c) * ld_ + qd * 4) * 4);
        }
    }
    // V4: raw buffer loads — a uniform tile offset + the slot's offset, out-of-image slots get an out-of-range offset (zeros): ~6
    // VALU per 16 bytes.  (First version: 64-bit addresses and a branch per load, ~60 VALU each: 845 VALU per tile next to 192
    // MFMAs in the 32-channel kernel, and VALU and MFMA share the issue port: the tiles ran at 0.47 of the matrix rate.)
    __device__ __forceinline__ void fetch(const float* __restrict__ src_, const BufRsrc& rs, int img_, int H_, int W_, int ld_, int y0_, int x0_) {
        src = src_; img = img_; H = H_; W = W_; ld = ld_; y0 = y0_; x0 = x0_;
        if (!V4) return;
        // (the position index may be below zero at the borders, and the byte offset may pass 2^31: unsigned arithmetic, mod 2^32)
        const unsigned tile_off = (unsigned)((img * H + y0) * W + x0) * (unsigned)(ld * 4);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int y = y0 + (key[it] >> 20), x = x0 + ((key[it] >> 8) & 0xfff);
            const bool in = key[it] >= 0 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            v[it] = buf_load_f32x4(rs, in ? tile_off + koff[it] : 0xffffffffu);
        }
    }
    __device__ __forceinline__ void commit(float* __restrict__ dst) const {
        if (V4) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int r = key[it] >> 20, c = (key[it] >> 8) & 0xfff, qd = key[it] & 0xff;
                if (key[it] >= 0) *reinterpret_cast<float4*>(dst + (qd / QPP) * PLANE + r * PITCH + c * CPP + (qd % QPP) * 4) = v[it];
            }
        } else {
            for (int i = (int)threadIdx.x; i < ROWS * COLS * C; i += 256) {
                const int pos = i / C, ch = i - pos * C;
                const int r = pos / COLS, c = pos - r * COLS;
                const int y = y0 + r, x = x0 + c;
                float val = 0.f;
                if (y >= 0 && y < H && x >= 0 && x < W) val = src[(((long long)img * H + y) * W + x) * ld + ch];
                dst[(ch / CPP) * PLANE + r * PITCH + c * CPP + (ch % CPP)] = val;
            }
        }
    }
};

// NCB: blocks of 8 B channels (Cb <= 8 NCB): one accumulator per (channel block, kw); Ca <= 8.
template <bool A4, bool B4, int NCB>
__global__ __launch_bounds__(256) void k_wgrad2d_3x3_c8(const float* __restrict__ A, const float* __restrict__ Bt, int n, int H, int W,
                                                        int Ca, int Cb, int lda, int ldb, int tiles_y, int tiles_x,
                                                        unsigned abytes, unsigned bbytes, float* __restrict__ scratch) {
    constexpr int TH = kW2TH, TW = kW2TW, AP = kW2APitch, BP = kW2BPitch, BPL = kW2BPlane;
    constexpr int GC = NCB == 1 ? 4 : 2;                   // groups per operand chunk (two chunks in registers: 2 GC (1 + 3 NCB) values)
    __shared__ float la[(TH + 1) * AP];
    __shared__ float lb[NCB * BPL];
    static_assert(BPL >= 3 * 3 * 256, "the cross-wave reduction reuses the B tile");
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    const int lo = j & 7, hi = j >> 3;                     // A operand: (a, s) = (lo, hi); B operand: (b, u) = (lo, hi)
    f32x4 acc[NCB][3];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[cb][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = n * tiles_y * tiles_x;
    W2Stage<TH + 1, TW, AP, 0, 1, A4> sa;
    W2Stage<TH + 2, kW2BCols, BP, BPL, NCB, B4> sb;
    sa.init(Ca, W, lda);
    sb.init(Cb, W, ldb);
    const BufRsrc ra = buf_rsrc(A, abytes), rb = buf_rsrc(Bt, bbytes);
    auto fetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
        const int x0 = tx * TW, y0 = ty * TH - 1;
        sa.fetch(A, ra, img, H, W, lda, y0, x0);
        sb.fetch(Bt, rb, img, H, W, ldb, y0 - 1, x0 - 1);
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        sa.commit(la);
        sb.commit(lb);
        __syncthreads();
        if (ENERF_WGRAD_PREFETCH && t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);       // in flight behind this tile's MFMAs
        // Operand pipeline: chunk c + 1's LDS reads are issued before chunk c's MFMAs (two register sets), so a wave's MFMAs run back
        // to back.  (Left to itself hipcc waited for each ds_read2 pair in front of the two MFMAs that use it; with one set per
        // chunk a wave idled a read round trip per chunk: PMC 0.47 matrix-pipe busy at 0.59 SQ_WAIT_INST_ANY.)
        constexpr int NCH = (TH / 4) * (TW / 4) / GC;          // chunks of GC groups per wave and tile, row-major
        float av[2][GC], bv[2][GC][NCB][3];
        auto load = [&](int c, float (&a_)[GC], float (&b_)[GC][NCB][3]) {
            const int rr = (c * GC) / (TW / 4), c0 = (c * GC) % (TW / 4), tr = wv + 4 * rr;
            const float* pa = la + (tr + hi) * AP + g * 8 + lo;
            const float* pb = lb + (tr + 2 * hi) * BP + g * 8 + lo;
#pragma unroll
            for (int cg = 0; cg < GC; ++cg) {
                a_[cg] = pa[(c0 + cg) * 32];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int k = 0; k < 3; ++k) b_[cg][cb][k] = pb[cb * BPL + (c0 + cg) * 32 + k * 8];
            }
        };
        load(0, av[0], bv[0]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) load(c + 1, av[(c + 1) & 1], bv[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cg = 0; cg < GC; ++cg)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[cb][k] = ENERF_MFMA_W(av[c & 1][cg], bv[c & 1][cg][cb][k], acc[cb][k]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (!ENERF_WGRAD_PREFETCH && t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
    }
    // waves 1..3 hand their tiles to wave 0 through LDS (the B tile's storage: the loop ended with a barrier)
    float* red = lb;
    if (wv != 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[cb * BPL + ((wv - 1) * 3 + k) * 256 + r * 64 + lane] = acc[cb][k][r];
    }
    __syncthreads();
    if (wv != 0) return;
    float* out = scratch + (long long)blockIdx.x * (Ca * Cb * 9);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[cb][k][r];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[cb * BPL + (w * 3 + k) * 256 + r * 64 + lane];
                // D rows 4g + r = (a, s), columns j = (b, u): kh = 2u - s, kw = k
                const int row = 4 * g + r, a = row & 7, s = row >> 3, b = cb * 8 + lo, u = hi, kh = 2 * u - s;
                if (kh >= 0 && a < Ca && b < Cb) out[(a * Cb + b) * 9 + kh * 3 + k] = v;
            }
}
// The 3x3 stride-1 2-D layers with 16 NA x 16 NB channels (conv1.1 16 <- 16, smooth1 16 <- 32, conv2.1 32 <- 32): the same LDS tiles
// without shifted copies (there is no idle half): nine MFMAs per group and tile pair, operands at immediate LDS offsets; pixel pitch
// 16 floats, so lane (g, j) reads bank (16 g + j) mod 32 — conflict-free per 32-lane half.
template <int NA, int NB>
__global__ __launch_bounds__(256) void k_wgrad2d_3x3_p16(const float* __restrict__ A, const float* __restrict__ Bt, int n, int H, int W,
                                                         int lda, int ldb, int tiles_y, int tiles_x, unsigned abytes, unsigned bbytes,
                                                         float* __restrict__ scratch) {
    constexpr int TH = kW2TH, TW = kW2TW, AP = TW * 16, APL = TH * AP + 16, BC = TW + 2, BP = BC * 16, BPL = (TH + 2) * BP + 16;
    static_assert(APL % 32 == 16 && BPL % 32 == 16, "the two planes of a 32-channel pixel 16 banks apart (ds_write_b128 groups)");
    constexpr int Ca = 16 * NA, Cb = 16 * NB;
    __shared__ float lds[NA * APL + NB * BPL];
    static_assert(NA * APL + NB * BPL >= 3 * 9 * 256, "the cross-wave reduction reuses the tiles");
    float* la = lds;
    float* lb = lds + NA * APL;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    f32x4 acc[NA][NB][9];
#pragma unroll
    for (int na = 0; na < NA; ++na)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[na][nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = n * tiles_y * tiles_x;
    W2Stage<TH, TW, AP, APL, NA, true, 16> sa;
    W2Stage<TH + 2, BC, BP, BPL, NB, true, 16> sb;
    sa.init(Ca, W, lda);
    sb.init(Cb, W, ldb);
    const BufRsrc ra = buf_rsrc(A, abytes), rb = buf_rsrc(Bt, bbytes);
    auto fetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
        sa.fetch(A, ra, img, H, W, lda, ty * TH, tx * TW);
        sb.fetch(Bt, rb, img, H, W, ldb, ty * TH - 1, tx * TW - 1);
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        sa.commit(la);
        sb.commit(lb);
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
        constexpr int NCH = (TH / 4) * (TW / 4);               // one group per chunk, two register sets (the c8 kernels' pipeline)
        float av[2][NA], bv[2][NB][9];
        auto load = [&](int c, float (&a_)[NA], float (&b_)[NB][9]) {
            const int rr = c / (TW / 4), cg = c % (TW / 4), tr = wv + 4 * rr;
            const float* pa = la + tr * AP + (4 * cg + g) * 16 + j;
            const float* pb = lb + tr * BP + (4 * cg + g) * 16 + j;
#pragma unroll
            for (int na = 0; na < NA; ++na) a_[na] = pa[na * APL];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) b_[nb][kh * 3 + kw] = pb[nb * BPL + kh * BP + kw * 16];
        };
        load(0, av[0], bv[0]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) load(c + 1, av[(c + 1) & 1], bv[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int na = 0; na < NA; ++na)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int t9 = 0; t9 < 9; ++t9) acc[na][nb][t9] = ENERF_MFMA_W(av[c & 1][na], bv[c & 1][nb][t9], acc[na][nb][t9]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float* out = scratch + (long long)blockIdx.x * (Ca * Cb * 9);
#pragma unroll
    for (int na = 0; na < NA; ++na)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (wv != 0) {
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lds[((wv - 1) * 9 + t9) * 256 + r * 64 + lane] = acc[na][nb][t9][r];
            }
            __syncthreads();
            if (wv == 0) {
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[na][nb][t9][r];
#pragma unroll
                        for (int w = 0; w < 3; ++w) v += lds[(w * 9 + t9) * 256 + r * 64 + lane];
                        out[((na * 16 + 4 * g + r) * Cb + nb * 16 + j) * 9 + t9] = v;      // D rows 4g + r, column j
                    }
            }
            __syncthreads();
        }
}
// ---------------------------------------------------------------------------------------------------------------------
// The same for the 3x3x3 stride-1 layers of the cost-regularisation networks with <= 8 channels on one side: conv0 (32 / 16 -> 8:
// A = d y has 8 channels) and the fused heads (8 -> 16: B = y has 8 channels — run with the roles SWAPPED, A' = y, B' = d heads:
// dW[b'][a'][t] = dW'[a'][b'][26 - t]).  A volume is n D images: a tile of plane d stages A's plane d and B's planes d-1, d, d+1
// (zeros outside the volume) and runs the 2-D scheme once per kd: 9 MFMAs per group of four positions and block of 8 B channels
// instead of 27 per 16, operands from LDS at immediate offsets.  Two blocks of 8 B channels per launch column; a 32-channel B is
// two columns (blockIdx.y) on the two 16-channel halves of its rows.
// ---------------------------------------------------------------------------------------------------------------------
template <bool A4, int NCB>
__global__ __launch_bounds__(256) void k_wgrad3d_c8(const float* __restrict__ A, const float* __restrict__ Bt, int n, int D, int H, int W,
                                                    int Ca, int Cb, int lda, int ldb, int tiles_y, int tiles_x, int swapped,
                                                    unsigned abytes, unsigned bbytes, float* __restrict__ scratch) {
    constexpr int TH = kW2TH, TW = kW2TW, AP = kW2APitch, BP = kW2BPitch, BPL = kW2BPlane;
    constexpr int GC = NCB == 1 ? 2 : 1;                   // groups per operand chunk (two chunks in registers: 2 GC (1 + 9 NCB) values)
    __shared__ float la[(TH + 1) * AP];
    __shared__ float lb[3 * NCB * BPL];                    // [kd][channel block][rows][cols][8]
    static_assert(3 * BPL >= 3 * 9 * 256, "the cross-wave reduction reuses the B planes, one channel block at a time");
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    const int lo = j & 7, hi = j >> 3;
    Bt += (int)blockIdx.y * 16;                            // this column's 16-channel half of B's rows
    f32x4 acc[3][NCB][3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[kd][cb][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = n * D * tiles_y * tiles_x;
    W2Stage<TH + 1, TW, AP, 0, 1, A4> sa;
    W2Stage<TH + 2, kW2BCols, BP, BPL, NCB, true> sb[3];
    sa.init(Ca, W, lda);
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) sb[kd].init(8 * NCB, W, ldb);
    const BufRsrc ra = buf_rsrc(A, abytes), rb = buf_rsrc(Bt, bbytes - (unsigned)blockIdx.y * 64u);
    auto fetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y), d = img % D;   // img = (b, d)
        const int x0 = tx * TW, y0 = ty * TH - 1;
        sa.fetch(A, ra, img, H, W, lda, y0, x0);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            // (uniform) a plane outside the volume: every slot out of range -> zeros.  The tile offset is taken with the real H.
            const bool in = (unsigned)(d + kd - 1) < (unsigned)D;
            sb[kd].fetch(Bt, rb, img + kd - 1, H, in ? W : 0, ldb, y0 - 1, x0 - 1);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        sa.commit(la);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) sb[kd].commit(lb + kd * NCB * BPL);
        __syncthreads();
        if (ENERF_WGRAD3D_PREFETCH && t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
        constexpr int NCH = (TH / 4) * (TW / 4) / GC;          // the 2-D kernel's operand pipeline, 9 NCB B values per group
        float av[2][GC], bv[2][GC][3][NCB][3];
        auto load = [&](int c, float (&a_)[GC], float (&b_)[GC][3][NCB][3]) {
            const int rr = (c * GC) / (TW / 4), c0 = (c * GC) % (TW / 4), tr = wv + 4 * rr;
            const float* pa = la + (tr + hi) * AP + g * 8 + lo;
            const float* pb = lb + (tr + 2 * hi) * BP + g * 8 + lo;
#pragma unroll
            for (int cg = 0; cg < GC; ++cg) {
                a_[cg] = pa[(c0 + cg) * 32];
#pragma unroll
                for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int k = 0; k < 3; ++k) b_[cg][kd][cb][k] = pb[(kd * NCB + cb) * BPL + (c0 + cg) * 32 + k * 8];
            }
        };
        load(0, av[0], bv[0]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) load(c + 1, av[(c + 1) & 1], bv[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cg = 0; cg < GC; ++cg)
#pragma unroll
                for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc[kd][cb][k] = ENERF_MFMA_W(av[c & 1][cg], bv[c & 1][cg][kd][cb][k], acc[kd][cb][k]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (!ENERF_WGRAD3D_PREFETCH && t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
    }
    // waves 1..3 hand their tiles to wave 0 through LDS (the B planes' storage), one channel block at a time
    const int n_row = Ca * Cb * 27;
    float* out = scratch + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * n_row;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (wv != 0) {
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lb[((wv - 1) * 9 + kd * 3 + k) * 256 + r * 64 + lane] = acc[kd][cb][k][r];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[kd][cb][k][r];
#pragma unroll
                        for (int w = 0; w < 3; ++w) v += lb[(w * 9 + kd * 3 + k) * 256 + r * 64 + lane];
                        const int row = 4 * g + r, a = row & 7, s = row >> 3, b = cb * 8 + lo, u = hi, kh = 2 * u - s;
                        const int tap = kd * 9 + kh * 3 + k;
                        if (kh >= 0 && a < Ca && b < Cb) out[swapped ? (b * Ca + a) * 27 + (26 - tap) : (a * Cb + b) * 27 + tap] = v;
                    }
        }
        __syncthreads();
    }
}
static int wgrad2d_bpc(int Ca, int Cb) {               // by LDS (21 / 32 / 54 KB per block; p16: 38 / 60 / 77 KB) and registers
    if (Ca > 8) return Cb == 32 ? 1 : 3;                // (272 / 356 registers with a 32-channel B: one wave per SIMD)
    return Cb <= 8 ? ENERF_WGRAD2D_BPC : Cb <= 16 ? 3 : 2;
}
static int wgrad2d_blocks(int n, int H, int W, int Ca, int Cb) {
    const long long ntiles = (long long)n * cdiv(Ca > 8 ? H : H + 1, kW2TH) * cdiv(W, kW2TW);
    const long long cap = (long long)device_cu_count() * wgrad2d_bpc(Ca, Cb);
    return (int)(ntiles < cap ? ntiles : cap);
}
#ifndef ENERF_WGRAD2D_P16
#define ENERF_WGRAD2D_P16 1                      /* 0: the 16 / 32-channel 3x3 layers stay on k_conv_wgrad (A/B builds) */
#endif
static bool wgrad2d_p16(int Ca, int Cb) { return ENERF_WGRAD2D_P16 && ((Ca == 16 && (Cb == 16 || Cb == 32)) || (Ca == 32 && Cb == 32)); }
static bool wgrad2d_fits(int n, int Da, int Ha, int Wa, int Ca, int Db, int Hb, int Wb, int Cb, int kd, int kh, int kw, int stride,
                         int pad_d, int pad_h, int pad_w, bool bias) {
    const bool shape = (Ca <= 8 && (Cb <= 8 || Cb == 16 || Cb == 32)) || wgrad2d_p16(Ca, Cb);
    return ENERF_WGRAD2D_TILE && kd == 1 && kh == 3 && kw == 3 && stride == 1 && pad_d == 0 && pad_h == 1 && pad_w == 1 && Da == 1 &&
           Db == 1 && Ha == Hb && Wa == Wb && shape && !bias && (long long)n * Ha * Wa >= 2048;
}
static size_t wgrad2d_workspace_bytes(int n, int H, int W, int Ca, int Cb) {
    return (size_t)wgrad2d_blocks(n, H, W, Ca, Cb) * Ca * Cb * 9 * sizeof(float);
}
static bool launch_wgrad2d(const float* A, const float* Bt, int n, int H, int W, int Ca, int Cb, int lda, int ldb, float* dW,
                           float* scratch, hipStream_t st) {
    const int blocks = wgrad2d_blocks(n, H, W, Ca, Cb), tiles_y = cdiv(H + 1, kW2TH), tiles_x = cdiv(W, kW2TW);
    const unsigned abytes = (unsigned)((long long)n * H * W * lda * 4), bbytes = (unsigned)((long long)n * H * W * ldb * 4);   // (< 2^32: checked by the C entries)
    if (Ca > 8) {                                          // 16 / 32 gradient channels: the plain tiles
        if (lda % 4 != 0 || ldb % 4 != 0 || (((uintptr_t)A | (uintptr_t)Bt) & 15) != 0) return false;
        const int ty16 = cdiv(H, kW2TH);
#define ENERF_W2P(NA, NB) ENERF_LAUNCH((k_wgrad2d_3x3_p16<NA, NB>), (unsigned)blocks, 256, 0, st, A, Bt, n, H, W, lda, ldb, ty16, tiles_x, abytes, bbytes, scratch)
        if (Ca == 32) ENERF_W2P(2, 2); else if (Cb == 32) ENERF_W2P(1, 2); else ENERF_W2P(1, 1);
#undef ENERF_W2P
        launch_colsum(scratch, blocks, Ca * Cb * 9, Ca * Cb * 9, 0, dW, cdiv(Ca * Cb * 9, 64), 1, st);
        return true;
    }
    const bool a4 = (Ca == 4 || Ca == 8) && lda % 4 == 0 && ((uintptr_t)A & 15) == 0;
    const bool b4 = (Cb == 4 || Cb == 8 || Cb == 16 || Cb == 32) && ldb % 4 == 0 && ((uintptr_t)Bt & 15) == 0;
    if (Cb > 8 && !b4) return false;
#define ENERF_W2(A4, B4, NCB) ENERF_LAUNCH((k_wgrad2d_3x3_c8<A4, B4, NCB>), (unsigned)blocks, 256, 0, st, A, Bt, n, H, W, Ca, Cb, lda, ldb, tiles_y, tiles_x, abytes, bbytes, scratch)
    if (Cb == 32) { if (a4) ENERF_W2(true, true, 4); else ENERF_W2(false, true, 4); }
    else if (Cb == 16) { if (a4) ENERF_W2(true, true, 2); else ENERF_W2(false, true, 2); }
    else if (a4) { if (b4) ENERF_W2(true, true, 1); else ENERF_W2(true, false, 1); }
    else { if (b4) ENERF_W2(false, true, 1); else ENERF_W2(false, false, 1); }
#undef ENERF_W2
    launch_colsum(scratch, blocks, Ca * Cb * 9, Ca * Cb * 9, 0, dW, cdiv(Ca * Cb * 9, 64), 1, st);
    return true;
}
#ifndef ENERF_WGRAD3D_TILE
#define ENERF_WGRAD3D_TILE 1                     /* 0: these layers stay on k_conv_wgrad (A/B builds) */
#endif
// normal: Ca <= 8 gradient channels, Cb = 8 / 16 / 32 input channels; swapped: Ca = 16, Cb <= 8 (the fused heads)
static bool wgrad3d_fits(int n, int Da, int Ha, int Wa, int Ca, int Db, int Hb, int Wb, int Cb, int kd, int kh, int kw, int stride,
                         int pad_d, int pad_h, int pad_w, bool bias) {
    const bool shape = (Ca <= 8 && (Cb == 8 || Cb == 16 || Cb == 32)) || (Ca == 16 && Cb <= 8);
    return ENERF_WGRAD3D_TILE && kd == 3 && kh == 3 && kw == 3 && stride == 1 && pad_d == 1 && pad_h == 1 && pad_w == 1 && Da == Db &&
           Ha == Hb && Wa == Wb && shape && !bias && (long long)n * Da * Ha * Wa >= 32768;
}
static int wgrad3d_blocks(int n, int D, int H, int W, int cols) {
    const long long ntiles = (long long)n * D * cdiv(H + 1, kW2TH) * cdiv(W, kW2TW);
    const long long cap = (long long)device_cu_count() * 2 / cols;                   // 76 KB of LDS per block: two per CU
    return (int)(ntiles < cap ? ntiles : (cap < 1 ? 1 : cap));
}
static size_t wgrad3d_workspace_bytes(int n, int D, int H, int W, int Ca, int Cb) {
    const int cols = (Ca <= 8 && Cb == 32) ? 2 : 1;
    return (size_t)cols * wgrad3d_blocks(n, D, H, W, cols) * Ca * (Cb / cols) * 27 * sizeof(float);
}
static bool launch_wgrad3d(const float* A, const float* Bt, int n, int D, int H, int W, int Ca, int Cb, int lda, int ldb, float* dW,
                           float* scratch, hipStream_t st) {
    const bool swapped = Ca > 8;
    const float* Ak = swapped ? Bt : A;                    // the kernel's A' (<= 8 channels) and B' (8 or 16 per column)
    const float* Bk = swapped ? A : Bt;
    const int Cak = swapped ? Cb : Ca, ldak = swapped ? ldb : lda, ldbk = swapped ? lda : ldb;
    const int Cbt = swapped ? Ca : Cb, cols = Cbt == 32 ? 2 : 1, Cbk = Cbt / cols;
    if (ldbk % 4 != 0 || ((uintptr_t)Bk & 15) != 0) return false;
    const bool a4 = (Cak == 4 || Cak == 8) && ldak % 4 == 0 && ((uintptr_t)Ak & 15) == 0;
    const int blocks = wgrad3d_blocks(n, D, H, W, cols), tiles_y = cdiv(H + 1, kW2TH), tiles_x = cdiv(W, kW2TW);
    const unsigned abytes = (unsigned)((long long)n * D * H * W * ldak * 4), bbytes = (unsigned)((long long)n * D * H * W * ldbk * 4);
    const dim3 grid((unsigned)blocks, (unsigned)cols);
#define ENERF_W3(A4, NCB) ENERF_LAUNCH((k_wgrad3d_c8<A4, NCB>), grid, 256, 0, st, Ak, Bk, n, D, H, W, Cak, Cbk, ldak, ldbk, tiles_y, tiles_x, swapped ? 1 : 0, abytes, bbytes, scratch)
    if (Cbk == 16) { if (a4) ENERF_W3(true, 2); else ENERF_W3(false, 2); }
    else { if (a4) ENERF_W3(true, 1); else ENERF_W3(false, 1); }
#undef ENERF_W3
    const int n_row = Cak * Cbk * 27;
    launch_colsum(scratch, blocks, n_row, swapped ? n_row : Cbk * 27, Cbt * 27, dW, cdiv(n_row, 64), cols, st);
    return true;
}

bool launch_conv_wgrad(const float* A, const float* Bt, int n, int Da, int Ha, int Wa, int Ca, int Db, int Hb, int Wb, int Cb,
                       int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w, float* dW, hipStream_t st, int lda = 0,
                       int ldb = 0, float* dbias = nullptr, void* workspace = nullptr, size_t workspace_bytes = 0) {
    WgradGeom q;
    q.bias = dbias != nullptr;
    q.lda = lda > 0 ? lda : Ca; q.ldb = ldb > 0 ? ldb : Cb;
    q.n = n; q.Da = Da; q.Ha = Ha; q.Wa = Wa; q.Ca = Ca; q.Db = Db; q.Hb = Hb; q.Wb = Wb; q.Cb = Cb;
    q.stride = stride; q.pad_d = pad_d; q.pad_h = pad_h; q.pad_w = pad_w;
    q.tiles_a = cdiv(Ca, 16); q.tiles_b = cdiv(Cb + q.bias, 16);
    q.npos = (long long)n * Da * Ha * Wa;
    q.abytes = (unsigned)(q.npos * q.lda * 4);                       // (< 2^32: checked by the C entries)
    q.bbytes = (unsigned)((long long)n * Db * Hb * Wb * q.ldb * 4);
    // 1x1(x1) stride-1 layers (FeatureNet toplayer / lat1 / lat0): k_conv_wgrad<1,1,1> decodes every position (three divisions per
    // MFMA) and re-reads the rows once per tile pair (lat0 at full resolution: 108 us); as a GEMM over the positions the rows are
    // read once and there is nothing to decode.  (enerf_gemm_wgrad falls back to this function for > 4 x 6 tiles: no recursion,
    // that case fails the tile test here too.)
    if (ENERF_WGRAD_1X1_GEMM && kd == 1 && kh == 1 && kw == 1 && stride == 1 && pad_d == 0 && pad_h == 0 && pad_w == 0 && Da == Db && Ha == Hb &&
        Wa == Wb && cdiv(Ca, 16) <= 4 && cdiv(Cb + q.bias, 16) <= 6 && q.npos * (long long)(q.lda > q.ldb ? q.lda : q.ldb) < (1LL << 30) &&
        launch_gemm_wgrad(A, q.lda, Ca, Bt, q.ldb, Cb, q.npos, dW, dbias, st, workspace, workspace_bytes))
        return true;
    if (wgrad2d_fits(n, Da, Ha, Wa, Ca, Db, Hb, Wb, Cb, kd, kh, kw, stride, pad_d, pad_h, pad_w, dbias != nullptr) && workspace != nullptr &&
        workspace_bytes >= wgrad2d_workspace_bytes(n, Ha, Wa, Ca, Cb) &&
        launch_wgrad2d(A, Bt, n, Ha, Wa, Ca, Cb, q.lda, q.ldb, dW, (float*)workspace, st))
        return true;
    if (wgrad3d_fits(n, Da, Ha, Wa, Ca, Db, Hb, Wb, Cb, kd, kh, kw, stride, pad_d, pad_h, pad_w, dbias != nullptr) && workspace != nullptr &&
        workspace_bytes >= wgrad3d_workspace_bytes(n, Da, Ha, Wa, Ca, Cb) &&
        launch_wgrad3d(A, Bt, n, Da, Ha, Wa, Ca, Cb, q.lda, q.ldb, dW, (float*)workspace, st))
        return true;
    const int taps = kd * kh * kw, pairs = q.tiles_a * q.tiles_b;
    float* scratch = (workspace != nullptr && workspace_bytes >= conv_wgrad_workspace_bytes(q.npos, Ca, Cb, taps, q.bias)) ? (float*)workspace : nullptr;
    q.chunks = wgrad_chunks(q.npos, pairs, taps, scratch != nullptr);
    const bool split = wgrad_split(taps, q.npos);
    if (kd == 3 && kh == 3 && kw == 3) {
        if (split) launch_wgrad_k<3, 3, 3, 3, 2>(A, Bt, q, dW, dbias, scratch, st);       // wave = (kd, position lane)
        else launch_wgrad_k<3, 3, 3, 1, 4>(A, Bt, q, dW, dbias, scratch, st);
        return true;
    }
    if (kd == 1 && kh == 3 && kw == 3) {
        if (split) launch_wgrad_k<1, 3, 3, 3, 2>(A, Bt, q, dW, dbias, scratch, st);       // wave = (kh, position lane)
        else launch_wgrad_k<1, 3, 3, 1, 4>(A, Bt, q, dW, dbias, scratch, st);
        return true;
    }
    if (kd == 1 && kh == 5 && kw == 5) { launch_wgrad_k<1, 5, 5, 5, 1>(A, Bt, q, dW, dbias, scratch, st); return true; }   // wave = kh
    if (kd == 1 && kh == 1 && kw == 1) { launch_wgrad_k<1, 1, 1, 1, 4>(A, Bt, q, dW, dbias, scratch, st); return true; }
    return false;
}

// ---------------------------------------------------------------------------------------------------------------------
// Linear-layer weight gradients: grad_w[a][b] = sum_p A[p][a] * B[p][b] with P up to ~2M rows and 1..89 columns per side.
// k_conv_wgrad<1,1,1> gives every (16x16) tile pair its own blocks, so the rows of A are re-read tiles_b times and those
// of B tiles_a times (3 KB per row for the 64 x 88 layer).  Here one wave keeps ALL TA x TB tile accumulators and walks its
// share of the rows once: TA + TB loads and TA*TB MFMAs per 4 rows, every row read exactly once (612 B for 64 x 89).
// ---------------------------------------------------------------------------------------------------------------------
// (the body: block `bid` of `nblk` — blockIdx.x / gridDim.x for the single GEMM, a slice of the grid in k_gemm_wgrad_group; `red`: NT x 256
// floats of LDS; `ldw`: row stride of dW, Cb unless the gradient is a column block of a wider matrix)
template <int TA, int TB>
__device__ __forceinline__ void gemm_wgrad_body(const float* __restrict__ A, const float* __restrict__ Bt, int lda, int ldb, int Ca, int Cb,
                                                int bias, long long P, float* __restrict__ dW, int ldw, float* __restrict__ dbias,
                                                float* __restrict__ scratch, int bid, int nblk, float* __restrict__ red) {
    constexpr int NT = TA * TB;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long ngroups = cdivl(P, 4);
    // raw buffer loads (common.h): scalar base + 32-bit lane offset, rows past P / columns past C read 0 through an
    // out-of-range offset — no branch around a load, no 64-bit address per load (P * ld < 2^30: checked by the C entry)
    const BufRsrc ra = buf_rsrc(A, (unsigned)(P * lda * 4)), rb = buf_rsrc(Bt, (unsigned)(P * ldb * 4));
    // U row groups per iteration, ALL their loads requested before the first MFMA (round 6).  A wave moves (TA + TB) x 256 B per row
    // group; with one group per iteration and eight waves per CU the 1 x 1 .. 2 x 2 shapes (view_fc, fc, global_fc: most of the MLP's
    // layers) had 4 - 8 KB in flight per CU and streamed their operands at 1.7 - 2.5 TB/s (the 4 x 6 shape: 5.4).  The groups of a
    // wave are still accumulated in ascending order: same sums, bit for bit.
#ifndef ENERF_GW_UNROLL
#define ENERF_GW_UNROLL 1            // 0 (A/B): one row group per iteration (round 5)
#endif
#ifndef ENERF_GW_INFLIGHT
#define ENERF_GW_INFLIGHT 16         // operand registers requested per iteration (x 256 B per wave in flight)
#endif
    constexpr int UQ = ENERF_GW_INFLIGHT / (TA + TB);
    constexpr int U = !ENERF_GW_UNROLL ? 1 : (UQ < 1 ? 1 : (UQ > 8 ? 8 : UQ));
    const long long gstride = (long long)nblk * 4;
    for (long long grp = (long long)bid * 4 + wave; grp < ngroups; grp += gstride * U) {
        float av[U][TA], bv[U][TB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long gu = grp + u * gstride;
            const unsigned p = (unsigned)gu * 4u + (unsigned)g;
            const bool pv = gu < ngroups && p < (unsigned)P;
            const unsigned arow = p * (unsigned)lda * 4u, brow = p * (unsigned)ldb * 4u;
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) {
                const int ca = ta * 16 + j;
                av[u][ta] = buf_load_f32(ra, (pv && ca < Ca) ? arow + (unsigned)ca * 4u : 0xffffffffu);
            }
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int cb = tb * 16 + j;
                bv[u][tb] = buf_load_f32(rb, (pv && cb < Cb) ? brow + (unsigned)cb * 4u : 0xffffffffu);
                if (bias && cb == Cb) bv[u][tb] = pv ? 1.f : 0.f;        // virtual all-ones column -> bias gradient
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) acc[ta * TB + tb] = ENERF_MFMA_W(av[u][ta], bv[u][tb], acc[ta * TB + tb]);
    }
    for (int src = 1; src < 4; ++src) {                               // waves 1..3 hand their tiles to wave 0
        if (wave == src) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[t * 256 + r * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] += red[t * 256 + r * 64 + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    if (scratch != nullptr) {       // two-stage commit (k_wgrad_reduce, gemm mode)
        float* sp = scratch + ((long long)bid * NT) * 256 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[t * 256 + r * 64] = acc[t][r];
        return;
    }
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[ta * TB + tb][r];
                const int a_ch = ta * 16 + 4 * g + r, b_ch = tb * 16 + j;
                if (a_ch >= Ca || v == 0.f) continue;
                if (b_ch < Cb) atomic_add_f32(dW + (long long)a_ch * ldw + b_ch, v);
                else if (bias && b_ch == Cb) atomic_add_f32(dbias + a_ch, v);
            }
}
template <int TA, int TB>
__global__ __launch_bounds__(256) void k_gemm_wgrad(const float* __restrict__ A, const float* __restrict__ Bt, int lda, int ldb,
                                                    int Ca, int Cb, int bias, long long P, float* __restrict__ dW,
                                                    float* __restrict__ dbias, float* __restrict__ scratch) {
    __shared__ float red[TA * TB * 256];
    gemm_wgrad_body<TA, TB>(A, Bt, lda, ldb, Ca, Cb, bias, P, dW, Cb, dbias, scratch, (int)blockIdx.x, (int)gridDim.x, red);
}

// ---- the weight gradients of ONE MLP level in two to four launches (round 6) ----
// NerfMlpFn's backward is ten to eleven of these GEMMs over the rows k_mlp_bwd saved (2.8 GB at level 1), each with its reduction: 21 + 21
// launches per level whose small members (view_fc 11 x 5, agg_w 1 x 33, fc 16 x 33 ...) run 20 - 50 us each on a ramp and a tail.  Here
// the grid is the concatenation of the GEMMs' grids (per register class) — every GEMM keeps its own block count, block -> row-group map and
// scratch slice, so each gradient is bit-identical to its single launch — and one reduction kernel follows for all of them; a gradient that is a column
// block of a wider matrix (color.0 = [shared | per-view] columns, global_fc = [a | var, mean]) is reduced straight into it (ldw).
struct GemmDesc {
    const float *A, *B;
    float *dW, *dbias, *scratch;
    long long P;
    int lda, ldb, Ca, Cb, bias, ldw, ta, tb, blocks, block0, rblock0, pad;
};
constexpr int kGemmGroupMax = 16;
struct GemmGroup {
    GemmDesc d[kGemmGroupMax];
    int n;
};
// CLS: the members' register class — one kernel for all shapes would give the 1 x 1 .. 2 x 3 shapes (52 - 68 registers, their loads
// in flight are what they live on) the 212 registers and two waves per SIMD of the 4 x 6 shape.  0: <= 6 tiles, 1: <= 12, 2: the rest.
__host__ __device__ constexpr int gemm_group_class(int nt) { return nt <= 6 ? 0 : (nt <= 12 ? 1 : 2); }
template <int CLS>
__global__ __launch_bounds__(256) void k_gemm_wgrad_group(GemmGroup G) {
    __shared__ float red[(CLS == 0 ? 6 : (CLS == 1 ? 12 : 24)) * 256];
    int i = 0;
    while (i + 1 < G.n && (int)blockIdx.x >= G.d[i + 1].block0) ++i;
    const GemmDesc& d = G.d[i];
    const int bid = (int)blockIdx.x - d.block0;
#define ENERF_GG(TA, TB)                                                                                                              \
    case TA * 8 + TB:                                                                                                                 \
        if constexpr (gemm_group_class(TA * TB) == CLS)                                                                               \
            gemm_wgrad_body<TA, TB>(d.A, d.B, d.lda, d.ldb, d.Ca, d.Cb, d.bias, d.P, d.dW, d.ldw, d.dbias, d.scratch, bid, d.blocks, red); \
        break;
    switch (d.ta * 8 + d.tb) {
        ENERF_GG(1, 1) ENERF_GG(1, 2) ENERF_GG(1, 3) ENERF_GG(1, 4) ENERF_GG(1, 5) ENERF_GG(1, 6)
        ENERF_GG(2, 1) ENERF_GG(2, 2) ENERF_GG(2, 3) ENERF_GG(2, 4) ENERF_GG(2, 5) ENERF_GG(2, 6)
        ENERF_GG(3, 1) ENERF_GG(3, 2) ENERF_GG(3, 3) ENERF_GG(3, 4) ENERF_GG(3, 5) ENERF_GG(3, 6)
        ENERF_GG(4, 1) ENERF_GG(4, 2) ENERF_GG(4, 3) ENERF_GG(4, 4) ENERF_GG(4, 5) ENERF_GG(4, 6)
        default: break;
    }
#undef ENERF_GG
}
// k_wgrad_reduce (gemm mode) for every member: block = (member, tile t, register r)
__global__ __launch_bounds__(1024) void k_gemm_wgrad_group_reduce(GemmGroup G) {
    __shared__ float red[16][64];
    int i = 0;
    while (i + 1 < G.n && (int)blockIdx.x >= G.d[i + 1].rblock0) ++i;
    const GemmDesc& d = G.d[i];
    const int lb = (int)blockIdx.x - d.rblock0, lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const int r = lb & 3, t = lb >> 2, nt = d.ta * d.tb;
    const float v = wgrad_reduce_sum(d.scratch + (long long)t * 256 + r * 64 + lane, d.blocks, (long long)nt * 256, red);
    if (w != 0) return;
    const int ta = t / d.tb, tb = t - ta * d.tb;
    const int a_ch = ta * 16 + 4 * g + r, b_ch = tb * 16 + j;
    if (a_ch >= d.Ca) return;
    if (b_ch < d.Cb) d.dW[(long long)a_ch * d.ldw + b_ch] = v;
    else if (d.bias && b_ch == d.Cb) d.dbias[a_ch] = v;
}

template <int TA>
static bool launch_gemm_wgrad_ta(int tb, unsigned grid, hipStream_t st, const float* A, const float* Bt, int lda, int ldb, int Ca,
                                 int Cb, int bias, long long P, float* dW, float* dbias, float* scratch) {
#define ENERF_GW(TBV) case TBV: ENERF_LAUNCH((k_gemm_wgrad<TA, TBV>), grid, 256, 0, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias, scratch); return true;
    switch (tb) { ENERF_GW(1) ENERF_GW(2) ENERF_GW(3) ENERF_GW(4) ENERF_GW(5) ENERF_GW(6) default: return false; }
#undef ENERF_GW
}
static long long gemm_wgrad_blocks(long long P) {
    const long long groups = cdivl(P, 4);
    long long blocks = cdivl(groups, 4 * 16);                         // >= 16 row groups per wave
#ifndef ENERF_GW_BLOCKS_PER_CU
#define ENERF_GW_BLOCKS_PER_CU 4       // measured: 2 -> 4 blocks per CU 9.92 -> 9.82 ms per training step
#endif
    const long long cap = (long long)device_cu_count() * ENERF_GW_BLOCKS_PER_CU;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}
size_t gemm_wgrad_workspace_bytes(long long P, int Ca, int Cb, int bias) {
    const int ta = cdiv(Ca, 16), tb = cdiv(Cb + bias, 16);
    if (ta > 4 || tb > 6) return conv_wgrad_workspace_bytes(P, Ca, Cb, 1, bias);      // the k_conv_wgrad<1,1,1> fallback
    return (size_t)gemm_wgrad_blocks(P) * ta * tb * 256 * sizeof(float);
}
// all tile pairs in one wave when they fit (<= 4 x 6 tiles); false -> the caller falls back to k_conv_wgrad<1,1,1>
bool launch_gemm_wgrad(const float* A, int lda, int Ca, const float* Bt, int ldb, int Cb, long long P, float* dW, float* dbias,
                       hipStream_t st, void* workspace, size_t workspace_bytes) {
    const int bias = dbias != nullptr;
    const int ta = cdiv(Ca, 16), tb = cdiv(Cb + bias, 16);
    if (ta > 4 || tb > 6) return false;
    const long long blocks = gemm_wgrad_blocks(P);
    float* scratch = (workspace != nullptr && workspace_bytes >= gemm_wgrad_workspace_bytes(P, Ca, Cb, bias)) ? (float*)workspace : nullptr;
    bool ok;
    switch (ta) {
        case 1: ok = launch_gemm_wgrad_ta<1>(tb, (unsigned)blocks, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias, scratch); break;
        case 2: ok = launch_gemm_wgrad_ta<2>(tb, (unsigned)blocks, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias, scratch); break;
        case 3: ok = launch_gemm_wgrad_ta<3>(tb, (unsigned)blocks, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias, scratch); break;
        default: ok = launch_gemm_wgrad_ta<4>(tb, (unsigned)blocks, st, A, Bt, lda, ldb, Ca, Cb, bias, P, dW, dbias, scratch); break;
    }
    if (ok && scratch != nullptr) launch_wgrad_reduce(scratch, 1, (int)blocks, ta * tb, tb, Ca, Cb, bias, 1, dW, dbias, st);
    return ok;
}

// host side of the group: per member its own block count and scratch slice (what launch_gemm_wgrad would use), then the two launches
static size_t gemm_group_plan(const enerf_gemm_wgrad_desc_t* u, int n, GemmGroup* G) {
    size_t floats = 0;
    int block0 = 0, rblock0 = 0;
    for (int i = 0; i < n; ++i) {
        const int bias = u[i].grad_bias != nullptr, ta = cdiv(u[i].Ca, 16), tb = cdiv(u[i].Cb + bias, 16);
        const bool pre = u[i].partials != nullptr;               // first stage done elsewhere: reduce only, no scratch of ours
        const int blocks = pre ? u[i].partial_chunks : (int)gemm_wgrad_blocks(u[i].P);
        if (G != nullptr) {
            GemmDesc& d = G->d[i];
            d.A = u[i].a; d.B = u[i].b; d.dW = u[i].grad_w; d.dbias = u[i].grad_bias; d.scratch = nullptr;
            d.P = u[i].P; d.lda = u[i].lda; d.ldb = u[i].ldb; d.Ca = u[i].Ca; d.Cb = u[i].Cb; d.bias = bias;
            d.ldw = u[i].ldw > 0 ? u[i].ldw : u[i].Cb; d.ta = ta; d.tb = tb; d.blocks = blocks; d.block0 = block0; d.rblock0 = rblock0;
            d.pad = pre ? -1 : (int)(floats / 256);               // scratch offset in 256-float tiles (resolved by the caller); -1: partials
        }
        if (!pre) floats += (size_t)blocks * ta * tb * 256;
        block0 += blocks;
        rblock0 += ta * tb * 4;
    }
    if (G != nullptr) G->n = n;
    return floats * sizeof(float);
}

}  // namespace enerf

using namespace enerf;
extern "C" {
int enerf_wgrad_reduce_begin(void) {
    REQUIRE(!g_reduce_deferring, "wgrad_reduce_begin: already deferring on this thread (missing enerf_wgrad_reduce_flush)");
    g_reduce_deferring = true;
    g_reduce_pending = 0;
    return ENERF_OK;
}
int enerf_wgrad_reduce_flush(enerf_stream_t stream) {
    REQUIRE(g_reduce_deferring, "wgrad_reduce_flush: no enerf_wgrad_reduce_begin on this thread");
    reduce_flush((hipStream_t)stream);
    return check_launch("wgrad_reduce_flush");
}
int enerf_colsum(const float* part, int chunks, int n, float* out, enerf_stream_t stream) {
    REQUIRE(part && out && chunks > 0 && n > 0, "colsum: bad arguments");
    launch_colsum(part, chunks, n, n, 0, out, cdiv(n, 64), 1, (hipStream_t)stream);
    return check_launch("colsum");
}
size_t enerf_gemm_wgrad_group_workspace_bytes(const enerf_gemm_wgrad_desc_t* descs, int n) {
    if (descs == nullptr || n < 1 || n > kGemmGroupMax) return 0;
    for (int i = 0; i < n; ++i)
        if ((descs[i].partials == nullptr && descs[i].P <= 0) || descs[i].Ca < 1 || descs[i].Cb < 1) return 0;
    return gemm_group_plan(descs, n, nullptr);
}
int enerf_gemm_wgrad_group(const enerf_gemm_wgrad_desc_t* descs, int n, void* workspace, size_t workspace_bytes, enerf_stream_t stream) {
    REQUIRE(descs && n >= 1 && n <= kGemmGroupMax, "gemm_wgrad_group: 1..%d members", kGemmGroupMax);
    for (int i = 0; i < n; ++i) {
        const enerf_gemm_wgrad_desc_t& u = descs[i];
        if (u.partials != nullptr) {
            REQUIRE(u.grad_w && u.Ca > 0 && u.Cb > 0 && u.partial_chunks > 0 && (u.ldw == 0 || u.ldw >= u.Cb),
                    "gemm_wgrad_group: member %d (partials): bad arguments", i);
        } else {
            REQUIRE(u.a && u.b && u.grad_w && u.Ca > 0 && u.Cb > 0 && u.lda >= u.Ca && u.ldb >= u.Cb && (u.ldw == 0 || u.ldw >= u.Cb),
                    "gemm_wgrad_group: member %d: bad arguments", i);
            REQUIRE(u.P > 0 && u.P < (1LL << 31), "gemm_wgrad_group: member %d: P out of range", i);
            REQUIRE(u.P * u.lda < (1LL << 30) && u.P * u.ldb < (1LL << 30),
                    "gemm_wgrad_group: member %d: a matrix of 4 GiB or more (32-bit byte offsets inside the kernel)", i);
        }
        REQUIRE(cdiv(u.Ca, 16) <= 4 && cdiv(u.Cb + (u.grad_bias != nullptr), 16) <= 6,
                "gemm_wgrad_group: member %d: more than 4 x 6 tiles (%d x %d columns): use enerf_gemm_wgrad", i, u.Ca, u.Cb);
    }
    GemmGroup G;
    const size_t need = gemm_group_plan(descs, n, &G);
    REQUIRE((workspace != nullptr || need == 0) && workspace_bytes >= need, "gemm_wgrad_group: workspace of %zu bytes, need %zu (enerf_gemm_wgrad_group_workspace_bytes)",
            workspace_bytes, need);
    for (int i = 0; i < n; ++i) {       // pad: 1 = no first-stage blocks of ours (the member brought its partial rows)
        const bool pre = G.d[i].pad < 0;
        G.d[i].scratch = pre ? const_cast<float*>(descs[i].partials) : (float*)workspace + (size_t)G.d[i].pad * 256;
        G.d[i].pad = pre ? 1 : 0;
    }
    const GemmDesc& last = G.d[n - 1];
    for (int cls = 0; cls < 3; ++cls) {                     // the members of one register class: one launch
        GemmGroup Gc;
        Gc.n = 0;
        int blocks = 0;
        for (int i = 0; i < n; ++i)
            if (G.d[i].pad == 0 && gemm_group_class(G.d[i].ta * G.d[i].tb) == cls) {
                Gc.d[Gc.n] = G.d[i];
                Gc.d[Gc.n].block0 = blocks;
                blocks += G.d[i].blocks;
                ++Gc.n;
            }
        if (Gc.n == 0) continue;
        if (cls == 0) ENERF_LAUNCH(k_gemm_wgrad_group<0>, (unsigned)blocks, 256, 0, (hipStream_t)stream, Gc);
        else if (cls == 1) ENERF_LAUNCH(k_gemm_wgrad_group<1>, (unsigned)blocks, 256, 0, (hipStream_t)stream, Gc);
        else ENERF_LAUNCH(k_gemm_wgrad_group<2>, (unsigned)blocks, 256, 0, (hipStream_t)stream, Gc);
    }
    ENERF_LAUNCH(k_gemm_wgrad_group_reduce, (unsigned)(last.rblock0 + last.ta * last.tb * 4), 1024, 0, (hipStream_t)stream, G);
    return check_launch("gemm_wgrad_group");
}
// scratch for the two-stage (atomic-free, deterministic) commit of the weight gradients; without it the entries fall back to
// fp32 atomics onto grad_w (correct, several times slower: see k_wgrad_reduce)
size_t enerf_conv_wgrad_workspace_bytes(long long positions_a, int Ca, int Cb, int kd, int kh, int kw) {
    return conv_wgrad_workspace_bytes(positions_a, Ca, Cb, kd * kh * kw, 0);
}
size_t enerf_gemm_wgrad_workspace_bytes(long long P, int Ca, int Cb, int with_bias) { return gemm_wgrad_workspace_bytes(P, Ca, Cb, with_bias != 0); }

int enerf_conv_wgrad(const float* a_cl, const float* b_cl, int n, int Da, int Ha, int Wa, int Ca, int Db, int Hb, int Wb,
                     int Cb, int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w, float* grad_w, void* workspace,
                     size_t workspace_bytes, enerf_stream_t stream) {
    REQUIRE(a_cl && b_cl && grad_w, "conv_wgrad: null pointer");
    REQUIRE(n > 0 && Da > 0 && Ha > 0 && Wa > 0 && Ca > 0 && Db > 0 && Hb > 0 && Wb > 0 && Cb > 0 && stride >= 1,
            "conv_wgrad: bad shape");
    REQUIRE((long long)n * Da * Ha * Wa < (1LL << 31) && (long long)n * Db * Hb * Wb < (1LL << 31), "conv_wgrad: more than 2^31 positions");
    REQUIRE((long long)n * Da * Ha * Wa * Ca < (1LL << 30) && (long long)n * Db * Hb * Wb * Cb < (1LL << 30),
            "conv_wgrad: a tensor of 4 GiB or more (32-bit byte offsets inside the kernel)");
    const bool two_stage = workspace != nullptr &&
                           workspace_bytes >= conv_wgrad_workspace_bytes((long long)n * Da * Ha * Wa, Ca, Cb, kd * kh * kw, 0);
    if (!two_stage) zero_async(grad_w, (size_t)Ca * Cb * kd * kh * kw * sizeof(float), (hipStream_t)stream);
    if (!launch_conv_wgrad(a_cl, b_cl, n, Da, Ha, Wa, Ca, Db, Hb, Wb, Cb, kd, kh, kw, stride, pad_d, pad_h, pad_w, grad_w,
                           (hipStream_t)stream, 0, 0, nullptr, workspace, workspace_bytes))
        return fail(ENERF_EINVAL, "conv_wgrad: kernel %dx%dx%d unsupported (3x3x3, 1x3x3, 1x5x5, 1x1x1)", kd, kh, kw);
    return check_launch("conv_wgrad");
}

// Plain position-reduction GEMM: grad_w[a][b] = sum_p A[p][a] * B[p][b] — the weight gradient of a Linear layer from its
// pre-activation gradient A (P rows, Ca used columns of rows lda floats wide) and its input B (P x Cb, rows ldb wide).
// grad_bias (nullable): the layer's bias gradient sum_p A[p][a], from the same pass (a virtual all-ones column of B).
int enerf_gemm_wgrad(const float* a, int lda, int Ca, const float* b, int ldb, int Cb, long long P, float* grad_w,
                     float* grad_bias, void* workspace, size_t workspace_bytes, enerf_stream_t stream) {
    REQUIRE(a && b && grad_w && Ca > 0 && Cb > 0 && lda >= Ca && ldb >= Cb, "gemm_wgrad: bad arguments");
    REQUIRE(P > 0 && P < (1LL << 31), "gemm_wgrad: P out of range");
    REQUIRE(P * lda < (1LL << 30) && P * ldb < (1LL << 30), "gemm_wgrad: a matrix of 4 GiB or more (32-bit byte offsets inside the kernel)");
    const bool two_stage = workspace != nullptr && workspace_bytes >= gemm_wgrad_workspace_bytes(P, Ca, Cb, grad_bias != nullptr);
    if (!two_stage) {
        zero_async(grad_w, (size_t)Ca * Cb * sizeof(float), (hipStream_t)stream);
        if (grad_bias) zero_async(grad_bias, (size_t)Ca * sizeof(float), (hipStream_t)stream);
        workspace = nullptr;
    }
    if (!launch_gemm_wgrad(a, lda, Ca, b, ldb, Cb, P, grad_w, grad_bias, (hipStream_t)stream, workspace, workspace_bytes))
        launch_conv_wgrad(a, b, 1, 1, 1, (int)P, Ca, 1, 1, (int)P, Cb, 1, 1, 1, 1, 0, 0, 0, grad_w, (hipStream_t)stream, lda, ldb, grad_bias,
                          workspace, workspace_bytes);
    return check_launch("gemm_wgrad");
}

}  // extern "C"
