// nerf_layout.h — the packed weight image of one NeRF/Agg MLP (nerf.py:6-89) and the MFMA helpers shared by the render
// kernel (render.hip) and the training-mode MLP kernels (mlp_train.hip).
#pragma once
#include "kernels.h"

namespace enerf {

// ---- packed weight image ---------------------------------------------------------------------------
struct NerfLayout {
    int F, R, TR;
    int view, viewb, glob, globb, aggw, fc, fcb, lr0, lr0b, sigma, c0p, c0b, c0v, col2, total;  // float offsets
};
// small_only: the layout WITHOUT the fp32 MFMA tiles of the dense layers (the bf16 render variants keep those as bf16 tiles):
// view_fc, the biases and the width-1 heads, same order
__host__ __device__ __forceinline__ NerfLayout nerf_layout(int F, bool small_only = false) {
    NerfLayout L;
    L.F = F; L.R = (F + 3) / 4; L.TR = (L.R + 3) / 4;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 63) / 64 * 64; return r; };
    auto big = [&](int n) { return take(small_only ? 0 : n); };
    L.view = take(L.TR * 64);
    L.viewb = take(L.TR * 16);
    L.glob = big(3 * L.R * 2 * 64);
    L.globb = take(32);
    L.aggw = take(33);
    L.fc = big(8 * 64);
    L.fcb = take(16);
    L.lr0 = big(6 * 4 * 64);
    L.lr0b = take(64);
    L.sigma = take(65);
    L.c0p = big(22 * 4 * 64);
    L.c0b = take(64);
    L.c0v = big((L.R + 1) * 4 * 64);
    L.col2 = take(65);
    L.total = o;
    return L;
}
// ---- device helpers ----------------------------------------------------------------------------------
__device__ __forceinline__ float group_sum(float v) { return group_sum4(v); }   // sum over the 4 lane groups (lanes j, j+16, j+32, j+48)
__device__ __forceinline__ f32x4 lds4(const float* p) {     // 16-byte aligned LDS/global read
    float4 t = *reinterpret_cast<const float4*>(p);
    return f32x4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ f32x4 relu4(f32x4 a) {
    return f32x4{relu1(a[0]), relu1(a[1]), relu1(a[2]), relu1(a[3])};
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b, float acc) {
    acc += a[0] * b[0]; acc += a[1] * b[1]; acc += a[2] * b[2]; acc += a[3] * b[3];
    return acc;
}
#define ENERF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// NV accumulators x NK k-steps of MFMAs whose A operands (one float per lane and tile) come from LDS: the A values of
// k-step ks+1 are read while the MFMAs of k-step ks run (a two-deep register ring), pinned with sched_barrier — left
// alone, hipcc emits `ds_read; s_waitcnt lgkmcnt(0); mfma; mfma` and exposes one LDS latency per MFMA pair.
// A(e) = this lane's element of tile e = ks*NV + v; B(ks) = the k-step's B operand.
#ifndef ENERF_CHAIN_DEPTH
#define ENERF_CHAIN_DEPTH 2          // register ring depth: A operands are read DEPTH-1 k-steps ahead of their MFMAs
#endif
template <int NK, int NV, class AF, class BF>
__device__ __forceinline__ void mfma_chain(f32x4 (&acc)[NV], AF A, BF B) {
    constexpr int DP = ENERF_CHAIN_DEPTH;
    float ring[DP][NV];
#pragma unroll
    for (int k = 0; k < DP - 1 && k < NK; ++k)
#pragma unroll
        for (int v = 0; v < NV; ++v) ring[k][v] = A(k * NV + v);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        if (ks + DP - 1 < NK) {
#pragma unroll
            for (int v = 0; v < NV; ++v) ring[(ks + DP - 1) % DP][v] = A((ks + DP - 1) * NV + v);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float b = B(ks);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = ENERF_MFMA(ring[ks % DP][v], b, acc[v]);
        __builtin_amdgcn_sched_barrier(0);
    }
}



// =====================================================================================================================
// fp32-accurate products on the bf16 matrix cores (round 4; profiles/r04_bf16x3_micro.txt, r04_ab_render_bf16x3.txt,
// r04_mfma_bf16_shapes.txt).  An fp32 value is split into bf16 pieces, x = hi + mid (+ lo): hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid) — two pieces carry 16-17 significant bits, three carry all 24.  A 32-deep k-chunk of a layer costs
//   "bf16x3": 3 v_mfma_f32_16x16x32_bf16 (hi*hi + hi*mid + mid*hi)                    = 48 matrix-pipe cycles, ~1e-5 relative
//   "bf16x6": 6 (the above + mid*mid + hi*lo + lo*hi; dropped terms <= 2^-24)          = 96 cycles, fp32-level
// instead of the 256 cycles of eight fp32 16x16x4 MFMAs, always with fp32 accumulation.  (The K = 32 instruction: measured
// 6.9 ns = 16.6 cycles, the same as the CDNA3-era 16x16x16 form — which therefore runs at half the bf16 peak.)
// Operand layout of v_mfma_f32_16x16x32_bf16: lane (g, j) supplies A[row j][k = 8g..8g+7] and B[k = 8g..8g+7][col j] (eight
// bf16 each) and receives D[rows 4g..4g+3][col j].  A k-chunk here is a PAIR of 16-unit groups: slots q = 0..3 of lane group g
// are units 4g + q of the first group, q = 4..7 of the second — so two D tiles of the previous layer (or one, the second half
// zero) are one B operand, and activations still never move between lanes.
// Weight image: three planes (hi, mid, lo) of [pair-tile][lane] 16 bytes.
// =====================================================================================================================
#ifdef ENERF_EMU
struct bfx8 { uint16_t v[8]; };
inline uint16_t bf16_rne(float f) {                       // round to nearest even, like v_cvt_pk_bf16_f32
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_widen(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
// pieces of the 8 values (lo4 | hi4): p[q] = q-th bf16 piece
template <int NP> inline void bx_split(const f32x4& lo4, const f32x4& hi4, bfx8 (&p)[NP]) {
    for (int r = 0; r < 8; ++r) {
        float rem = r < 4 ? lo4[r] : hi4[r - 4];
        for (int q = 0; q < NP; ++q) { p[q].v[r] = bf16_rne(rem); rem -= bf16_widen(p[q].v[r]); }
    }
}
inline f32x4 mfma_bf16_16x16x32(const bfx8& a, const bfx8& b, f32x4 c) {
    const unsigned lane = emu::ctx()->cur->tid % emu::kWave, col = lane & 15, g = lane >> 4;
    float A[8][64], Bm[8][64];
    for (int r = 0; r < 8; ++r) {
        auto buf = emu::wave_exchange(bf16_widen(a.v[r]), bf16_widen(b.v[r]));
        for (int l = 0; l < 64; ++l) { A[r][l] = buf[0][l]; Bm[r][l] = buf[1][l]; }
    }
    f32x4 d = c;
    for (int rr = 0; rr < 4; ++rr) {
        const unsigned row = 4 * g + rr;
        float acc = c[rr];
        for (int kg = 0; kg < 4; ++kg)
            for (int r = 0; r < 8; ++r) acc = fmaf(A[r][row + 16 * kg], Bm[r][col + 16 * kg], acc);      // k = 8 kg + r
        d[rr] = acc;
    }
    return d;
}
inline bfx8 bx_load_piece(const float* plane, int e, int lane) {
    bfx8 t; memcpy(&t, plane + ((long long)e * 64 + lane) * 4, 16); return t;
}
#else
typedef __bf16 bfx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
template <int NP> __device__ __forceinline__ void bx_split(const f32x4 lo4, const f32x4 hi4, bfx8 (&p)[NP]) {
    f32x4 ra = lo4, rb = hi4;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const bfx4 pa = __builtin_convertvector(ra, bfx4), pb = __builtin_convertvector(rb, bfx4);
        p[q] = __builtin_shufflevector(pa, pb, 0, 1, 2, 3, 4, 5, 6, 7);
        if (q + 1 < NP) { ra = ra - __builtin_convertvector(pa, f32x4); rb = rb - __builtin_convertvector(pb, f32x4); }
    }
}
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(const bfx8 a, const bfx8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bfx8 bx_load_piece(const float* plane, int e, int lane) {
    return __builtin_bit_cast(bfx8, *reinterpret_cast<const float4*>(plane + (e * 64 + lane) * 4));        // one ds_read_b128
}
#endif
// acc[v] += W[tile e0 + v] * B for NV output tiles that share the B operand.  Every A piece is read from LDS right before the
// terms that use it (lo: 1 term, mid: 2, hi: 3), so at most one piece per tile is live; smallest terms first.
// plane(q) = LDS address of piece q's plane.
template <int NP, int NV, class PF>
__device__ __forceinline__ void bx_mma_n(PF plane, int e0, int lane, const bfx8 (&b)[NP], f32x4 (&acc)[NV]) {
    bfx8 a[NV];
    if constexpr (NP == 3) {
#pragma unroll
        for (int v = 0; v < NV; ++v) a[v] = bx_load_piece(plane(2), e0 + v, lane);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = mfma_bf16_16x16x32(a[v], b[0], acc[v]);                     // lo * hi
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = bx_load_piece(plane(1), e0 + v, lane);
    if constexpr (NP == 3) {
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = mfma_bf16_16x16x32(a[v], b[1], acc[v]);                     // mid * mid
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = mfma_bf16_16x16x32(a[v], b[0], acc[v]);                         // mid * hi
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = bx_load_piece(plane(0), e0 + v, lane);
    if constexpr (NP == 3) {
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = mfma_bf16_16x16x32(a[v], b[2], acc[v]);                     // hi * lo
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = mfma_bf16_16x16x32(a[v], b[1], acc[v]);                         // hi * mid
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = mfma_bf16_16x16x32(a[v], b[0], acc[v]);                         // hi * hi
}
// pair-tile numbering of the bf16 image of one NeRF (F = 11: R = 3): 25 pair-tiles per plane
//   gvm  global_fc, columns of [variance | mean]         (2 output tiles)
//   ga   global_fc, columns of a view's a_s slots | 0    (2)
//   fc   agg.fc, columns of [G tile 0 | G tile 1]        (1)
//   lr0  lr0, columns of [voxel pair | agg]              (4)
//   c0p  color.0 shared columns: pairs [h0|h1] [h2|h3] [voxel pair|agg]   (3 x 4)
//   c0v  color.0 per-view columns [x_s slots + direction code | 0]        (4)
struct BxLayout { int gvm, ga, fc, lr0, c0p, c0v, tiles; };
__host__ __device__ __forceinline__ BxLayout bx_layout() {
    BxLayout b; b.gvm = 0; b.ga = 2; b.fc = 4; b.lr0 = 5; b.c0p = 9; b.c0v = 21; b.tiles = 25; return b;
}

}  // namespace enerf
