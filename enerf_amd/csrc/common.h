// common.h — shared device helpers for the ENeRF gfx950 kernels.
//
// Every kernel in this directory is written for CDNA4 (wave64, MFMA f32 16x16x4, 160 KiB LDS).  The
// same sources are also compiled by tests/ with -DENERF_EMU against tests/emu/hip_emu.h (a CPU
// emulation of lanes/waves used only to check kernel logic without a GPU); that build is never
// loaded by the product.
#pragma once

#ifdef ENERF_EMU
#include "hip_emu.h"
typedef emu_f32x4 f32x4;
#else
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ENERF_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (shmem), (stream), __VA_ARGS__)
#define ENERF_LAUNCH_SIMPLE ENERF_LAUNCH
#define ENERF_DYN_SMEM(type, name) \
    extern __shared__ __attribute__((aligned(16))) char enerf_dyn_smem_[]; \
    type* name = reinterpret_cast<type*>(enerf_dyn_smem_)
// make a VGPR value opaque to the optimiser (stops loop-invariant hoisting of everything derived from it)
#define ENERF_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif
#ifdef ENERF_EMU
#define ENERF_OPAQUE_V(x) (void)(x)
#endif

#include <stdint.h>

namespace enerf {

// Ordering point between LDS writes of some lanes of a wave and LDS reads of other lanes of the SAME wave.  The hardware
// keeps a wave's LDS operations in order; this only stops the compiler (and, in the CPU lane emulator, lets every lane of
// the wave reach this point before any continues).
__device__ __forceinline__ void wave_sync() {
#ifdef ENERF_EMU
    emu::wave_exchange(0.f);
#else
    // no fence: a fence would also drain the outstanding GLOBAL loads (s_waitcnt vmcnt(0)); the LDS store -> load
    // dependence through the same array is visible to the compiler, which waits on lgkmcnt only
    __builtin_amdgcn_wave_barrier();
#endif
}

// ReLU as ONE instruction.  fmaxf(x, 0.f) compiles to two (hipcc first canonicalises x with v_max_f32 x, x
// because the kernels run in IEEE mode); on MFMA outputs that doubled the ~110 ReLUs per 16 points of the
// render kernel.  v_med3_f32(x, 0, FLT_MAX) is the same function for finite x (NaN -> 0 like fmaxf; +inf ->
// FLT_MAX; with +inf as the bound hipcc folds it back into the two-instruction max) and, being a compiler
// intrinsic rather than inline asm, keeps the MFMA->VALU hazard wait states hipcc inserts (an inline-asm
// v_max_f32 on MFMA results read stale registers in the C=32 render variant: caught by the GPU parity tests).
__device__ __forceinline__ float relu1(float x) {
#ifdef ENERF_EMU
    return fmaxf(x, 0.f);
#else
    return __builtin_amdgcn_fmed3f(x, 0.f, 3.402823466e+38f);
#endif
}

// ---- asynchronous global -> LDS copies (global_load_lds_dwordx4: the LDS-DMA path of gfx950) -------------------------------
// glds16(src, dst_wave_base, lane): every lane supplies ITS OWN 16-byte global source; the destination is the wave-uniform
// LDS base + lane * 16 (the hardware's rule, not a choice).  No VGPR holds the data, so a block can have the next pass's tile in
// flight during this pass's MFMAs without the register cost of a staged prefetch.  Completion is tracked by vmcnt:
// glds_wait_all() = s_waitcnt vmcnt(0) as inline asm (hipcc's own __syncthreads would drain it too, but also fences), and
// block_barrier_raw() is the bare s_barrier that lets a copy issued BEFORE it stay in flight ACROSS it
// (cdna_hip_programming.md "Pipelining across barriers").  The CPU lane emulator copies immediately.
#ifdef ENERF_EMU
__device__ __forceinline__ void glds16(const float* src, float* dst_wave_base, int lane) {
    for (int k = 0; k < 4; ++k) dst_wave_base[lane * 4 + k] = src[k];
}
__device__ __forceinline__ void glds_wait_all() {}
template <int N> __device__ __forceinline__ void vmem_wait_pending() {}
__device__ __forceinline__ void block_barrier_raw() { __syncthreads(); }
#else
__device__ __forceinline__ void glds16(const float* src, float* dst_wave_base, int lane) {
    (void)lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's vector-memory operations are outstanding: loads (LDS-DMA copies included) retire in issue
// order, so with exactly N younger loads behind a copy this is "the copy has landed" without draining the younger loads
template <int N> __device__ __forceinline__ void vmem_wait_pending() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void block_barrier_raw() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif

// Raw buffer loads: a 128-bit resource (base, byte size) + a 32-bit byte offset per lane; an offset >= the size returns 0 —
// padding taps and dead lanes cost ONE select on the offset instead of address clamps or a zero page, and the address is
// `scalar base + 32-bit lane offset` (no 64-bit VALU address per load).  The emulator twin is the same bounds-checked read.
#ifdef ENERF_EMU
struct BufRsrc { const char* base; unsigned bytes; };
__device__ __forceinline__ BufRsrc buf_rsrc(const void* p, unsigned bytes) { return BufRsrc{(const char*)p, bytes}; }
__device__ __forceinline__ float buf_load_f32(const BufRsrc& r, unsigned byte_off) {
    return byte_off < r.bytes ? *reinterpret_cast<const float*>(r.base + byte_off) : 0.f;
}
__device__ __forceinline__ float4 buf_load_f32x4(const BufRsrc& r, unsigned byte_off) {
    return byte_off < r.bytes && r.bytes - byte_off >= 16u ? *reinterpret_cast<const float4*>(r.base + byte_off) : make_float4(0.f, 0.f, 0.f, 0.f);
}
#else
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc buf_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);   // gfx9 raw buffer, dword data
}
__device__ __forceinline__ float buf_load_f32(const BufRsrc& r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
// 16 bytes per lane; an offset at or past the resource's size reads zeros (no branch, no 64-bit address)
__device__ __forceinline__ float4 buf_load_f32x4(const BufRsrc& r, unsigned byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
#endif

constexpr int kWave = 64;

// v_mfma_f32_4x4x1_16b_f32 with the A-operand BROADCAST controls (cbsz:4 abid:K): the 4x1 A column of block K — lanes 4K..4K+3 of
// the A register — multiplies the B value of EVERY lane (all 16 blocks).  With lane = pixel (conv3d_b4.hip) this turns one VGPR
// into sixteen different weight columns (4 output channels x 1 input channel each), addressed by an immediate: a Cout = 8
// convolution then needs no LDS broadcast read (2 of the 3 ds_read_b128 per 8 MFMAs in round 4's kernels) and no per-block
// weight re-layout at all.  Semantics + issue rate: tools/micro/mfma_cbsz.hip.  K must fold to a constant (unrolled loops).
template <int K>
__device__ __forceinline__ f32x4 mfma4_bc_k(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, K, 0); }
__device__ __forceinline__ f32x4 mfma4_bc(float a, float b, f32x4 c, int k) {
    switch (k) {
        case 0: return mfma4_bc_k<0>(a, b, c);
        case 1: return mfma4_bc_k<1>(a, b, c);
        case 2: return mfma4_bc_k<2>(a, b, c);
        case 3: return mfma4_bc_k<3>(a, b, c);
        case 4: return mfma4_bc_k<4>(a, b, c);
        case 5: return mfma4_bc_k<5>(a, b, c);
        case 6: return mfma4_bc_k<6>(a, b, c);
        case 7: return mfma4_bc_k<7>(a, b, c);
        case 8: return mfma4_bc_k<8>(a, b, c);
        case 9: return mfma4_bc_k<9>(a, b, c);
        case 10: return mfma4_bc_k<10>(a, b, c);
        case 11: return mfma4_bc_k<11>(a, b, c);
        case 12: return mfma4_bc_k<12>(a, b, c);
        case 13: return mfma4_bc_k<13>(a, b, c);
        case 14: return mfma4_bc_k<14>(a, b, c);
        default: return mfma4_bc_k<15>(a, b, c);
    }
}

// Pin a value in a VGPR at this program point: LLVM's Sink pass otherwise moves a pure arithmetic chain that is only consumed
// under a late condition (an epilogue store) down into that branch, keeping every operand of the chain alive until then.
#ifdef ENERF_EMU
#define ENERF_PIN_VGPR(x) ((void)0)
#else
#define ENERF_PIN_VGPR(x) asm volatile("" : "+v"(x))
#endif
__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// v + (v of the lane 16 away) + (32 away) + (48 away): the sum / max over the four lane groups (lanes j, j+16, j+32, j+48)
// that share an MFMA column.  gfx950 swaps 16- and 32-lane halves between two registers in one VALU instruction
// (v_permlane16_swap / v_permlane32_swap): 2 swaps + 2 adds, no LDS round trip (__shfl_xor lowers to ds_bpermute: an
// LDS-latency stall on the critical path, 14 of them per 16 samples in the render kernel).
#ifdef ENERF_EMU
__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32); }
#else
__device__ __forceinline__ float xor32(float v) {      // value held by lane ^ 32
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0] = {lo, lo}, r[1] = {hi, hi} of v's halves: the "other" half is r[1] for lanes < 32 and r[0] for lanes >= 32;
    // for commutative reductions use quad_reduce below — this form is the general one
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor16(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
#endif
__device__ __forceinline__ float group_sum4(float v) {
#ifdef ENERF_EMU
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
#else
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);       // {lo,lo} and {hi,hi}: their sum is v + v^32
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(v);
    r = __builtin_amdgcn_permlane16_swap(u, u, false, false);            // rows {0,0,2,2} and {1,1,3,3}
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}
// sum over the 16 lanes of a row (lanes with the same lane >> 4), left in every lane of the row: four DPP row rotations (VALU;
// __shfl_xor would be four ds_bpermute round trips through the LDS pipe)
__device__ __forceinline__ float row_sum16(float v) {
#ifdef ENERF_EMU
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
#else
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));    // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));    // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));    // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));    // row_ror:1
    return v;
#endif
}
// Broadcast inside aligned groups of CQ (2, 4, 8) consecutive lanes: every lane gets the value of lane K of ITS group.  DPP
// register permutes (quad_perm; for 8-lane groups one row shift by 4 on top) — `__shfl` lowers to ds_bpermute_b32, a round trip
// through the LDS pipe (~100+ cycles of latency each), and the warp kernel does eight of them per source view.
template <int CQ, int K>
__device__ __forceinline__ int group_bcast_i(int v) {
#ifdef ENERF_EMU
    return __shfl(v, (int)((threadIdx.x & 63) & ~(CQ - 1)) + K);
#else
    static_assert(CQ == 2 || CQ == 4 || CQ == 8, "group width");
    if (CQ == 2) return __builtin_amdgcn_update_dpp(0, v, K | (K << 2) | ((2 + K) << 4) | ((2 + K) << 6), 0xf, 0xf, true);
    constexpr int q = K & 3, bc = q | (q << 2) | (q << 4) | (q << 6);                  // quad_perm: [q, q, q, q]
    const int y = __builtin_amdgcn_update_dpp(0, v, bc, 0xf, 0xf, true);
    if (CQ == 4) return y;
    // CQ == 8: the source quad's value has to reach the other quad of the group: a row shift by 4 lanes (0x114 = row_shr:4, lane
    // i <- i - 4; 0x104 = row_shl:4, lane i <- i + 4); the quads that shift in a neighbour GROUP's value keep their own y
    const bool upper = (threadIdx.x & 4) != 0;
    if (K < 4) { const int t = __builtin_amdgcn_update_dpp(0, y, 0x114, 0xf, 0xf, true); return upper ? t : y; }
    const int u = __builtin_amdgcn_update_dpp(0, y, 0x104, 0xf, 0xf, true);
    return upper ? y : u;
#endif
}
template <int CQ>
__device__ __forceinline__ int group_bcast_i(int v, int k) {      // k: loop-unrolled (folds to one case)
    switch (k) {
        case 0: return group_bcast_i<CQ, 0>(v);
        case 1: return group_bcast_i<CQ, 1 % CQ>(v);
        case 2: return group_bcast_i<CQ, 2 % CQ>(v);
        case 3: return group_bcast_i<CQ, 3 % CQ>(v);
        case 4: return group_bcast_i<CQ, 4 % CQ>(v);
        case 5: return group_bcast_i<CQ, 5 % CQ>(v);
        case 6: return group_bcast_i<CQ, 6 % CQ>(v);
        default: return group_bcast_i<CQ, 7 % CQ>(v);
    }
}
template <int CQ>
__device__ __forceinline__ float group_bcast_f(float v, int k) { return __int_as_float(group_bcast_i<CQ>(__float_as_int(v), k)); }

// v + the value of lane ^ 8 (one DPP row rotation by 8: inside a row of 16 lanes, +8 mod 16 is ^ 8)
__device__ __forceinline__ float add_xor8(float v) {
#ifdef ENERF_EMU
    return v + __shfl_xor(v, 8);
#else
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ float group_max4(float v) {
#ifdef ENERF_EMU
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
#else
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    u = __float_as_uint(v);
    r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#endif
}

// XCD-aware block order.  The dispatcher places block b on XCD b % 8 and every XCD has a private 4 MiB L2,
// so with the natural order neighbouring blocks (which share gather footprints / halos) land on eight
// different L2s and each L2 ends up fetching the whole input (measured: k_feature_volume fetched 6x its
// input).  This bijection hands each XCD one contiguous run of logical block ids instead.  Speed only —
// nothing depends on the placement being what we expect.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned bid, unsigned nblk) {
    const unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, k = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
__host__ __device__ __forceinline__ long long cdivl(long long a, long long b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// torch.nn.functional.interpolate(mode='bilinear', align_corners=True) source coordinate.
// ATen computes scale = (in-1)/(out-1) in float, src = scale*dst, i0 = int(src), l1 = src - i0,
// i1 = i0 + (i0 < in-1).  (reference call sites: utils.py:115-117, 394-396, 611; network.py:32)
// ---------------------------------------------------------------------------------------------
struct Lerp1 {
    int i0, i1;
    float l0, l1;
};
__host__ __device__ __forceinline__ float ac_scale(int in, int out) {
    return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
}
__host__ __device__ __forceinline__ Lerp1 ac_lerp(int dst, float scale, int in) {
    Lerp1 r;
    float src = scale * (float)dst;
    r.i0 = (int)src;
    if (r.i0 > in - 1) r.i0 = in - 1;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}
// value = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)   (ATen upsample_bilinear2d order)
__host__ __device__ __forceinline__ float ac_blend(const Lerp1& y, const Lerp1& x, float v00, float v01, float v10,
                                                   float v11) {
    return y.l0 * (x.l0 * v00 + x.l1 * v01) + y.l1 * (x.l0 * v10 + x.l1 * v11);
}

// ---------------------------------------------------------------------------------------------
// F.grid_sample(align_corners=True) pixel coordinate from a normalised coordinate:
// ((g + 1) / 2) * (size - 1)      (ATen grid_sampler_unnormalize)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float gs_unnorm(float g, int size) { return ((g + 1.f) * 0.5f) * (float)(size - 1); }

// Bilinear taps with ZEROS padding (homo_warp, utils.py:87-89) or BORDER padding (get_img_feat,
// utils.py:706).  Returns corner indices (clamped for addressing) and weights (0 for invalid taps).
struct Taps2 {
    int x0, x1, y0, y1;
    float w00, w01, w10, w11;  // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
};
// 24-bit integer multiply (full-rate v_mul_u32_u24; v_mul_lo_u32 is quarter rate).  Only for operands that are
// image/volume coordinates and extents (< 2^24, launchers check) with a product < 2^32.
__device__ __forceinline__ int mul24(int a, int b) {
#ifdef ENERF_EMU
    return a * b;
#else
    return __mul24(a, b);
#endif
}

template <bool BORDER>
__host__ __device__ __forceinline__ Taps2 gs_taps2(float ix, float iy, int W, int H) {
    Taps2 t;
    if (BORDER) {  // clip_coordinates: min(size-1, max(ix, 0)); fmaxf maps NaN to 0 like ATen's ::max
        ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
        iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
        // After clipping every tap is inside the image except x1 == W (y1 == H), which only happens at
        // ix == W-1 where its weight is exactly 0: no validity tests, one clamp per axis.
        const float fx = floorf(ix), fy = floorf(iy);
        t.x0 = (int)fx; t.y0 = (int)fy;
        t.x1 = t.x0 + 1 < W ? t.x0 + 1 : W - 1;
        t.y1 = t.y0 + 1 < H ? t.y0 + 1 : H - 1;
        const float tx1 = ix - fx, ty1 = iy - fy, tx0 = (fx + 1.f) - ix, ty0 = (fy + 1.f) - iy;
        t.w00 = tx0 * ty0; t.w01 = tx1 * ty0; t.w10 = tx0 * ty1; t.w11 = tx1 * ty1;
        return t;
    }
    // guard against NaN / huge values before the float->int conversion (result is then all-zero taps)
    bool finite = (ix > -1e8f) && (ix < 1e8f) && (iy > -1e8f) && (iy < 1e8f);
    if (!finite) { ix = -10.f; iy = -10.f; }
    float fx = floorf(ix), fy = floorf(iy);
    int x0 = (int)fx, y0 = (int)fy;
    int x1 = x0 + 1, y1 = y0 + 1;
    float tx1 = ix - fx, ty1 = iy - fy;      // weight of the +1 corner
    float tx0 = (float)x1 - ix, ty0 = (float)y1 - iy;
    bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
    bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    t.w00 = (vx0 && vy0) ? tx0 * ty0 : 0.f;
    t.w01 = (vx1 && vy0) ? tx1 * ty0 : 0.f;
    t.w10 = (vx0 && vy1) ? tx0 * ty1 : 0.f;
    t.w11 = (vx1 && vy1) ? tx1 * ty1 : 0.f;
    t.x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0);
    t.x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
    t.y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
    t.y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    return t;
}

__device__ __forceinline__ float clamp_min(float v, float lo) { return v < lo ? lo : v; }  // torch.clamp_min

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
#ifdef ENERF_EMU
    atomicAdd(p, v);
#else
    unsafeAtomicAdd(p, v);          // global_atomic_add_f32 (hipMalloc memory is coarse-grained)
#endif
}

// fp32 add onto an LDS word (ds_add_f32).  Through the generic atomic_add_f32 the compiler merges an `LDS or global` pair of
// branches into one FLAT atomic on a selected pointer; this keeps the LDS side a DS instruction.
__device__ __forceinline__ void lds_add_f32(float* p, float v) {
#ifdef ENERF_EMU
    atomicAdd(p, v);
#else
    __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)p, v, 0, 0, false);
#endif
}

// Hardware transcendental forms (v_rcp_f32 / v_sqrt_f32 / v_exp_f32, ~1 ulp) for the render kernel's
// per-sample geometry and softmaxes: an IEEE-correct fp32 divide or sqrt expands to ~10 VALU
// instructions and the sample loop had ~50 of them per 201 MFMAs.  The induced error (<=1e-6 relative)
// is two orders of magnitude inside the parity budget.  The CPU emulator uses libm.
#ifdef ENERF_EMU
__device__ __forceinline__ float fast_rcp(float x) { return 1.f / x; }
__device__ __forceinline__ float fast_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ float fast_exp(float x) { return expf(x); }
#else
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
#endif

// torch.linspace(0,1,D)[k]: step = 1/(D-1); k < D/2 ? step*k : 1 - step*(D-1-k)   (ATen RangeFactories)
__host__ __device__ __forceinline__ float linspace01(int k, int D) {
    if (D == 1) return 0.f;
    float step = 1.f / (float)(D - 1);
    return k < D / 2 ? step * (float)k : 1.f - step * (float)(D - 1 - k);
}

// 4x4 inverse by cofactors in fp64 (camera matrices; a handful of threads per frame)
__host__ __device__ __forceinline__ bool inv4x4(const double* m, double* inv) {
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0) return false;
    double r = 1.0 / det;
    for (int i = 0; i < 16; ++i) inv[i] *= r;
    return true;
}

// ---------------------------------------------------------------------------------------------
// build_rays (utils.py:390-420) for one ray: xN align-corners upsample of {depth, std, near_far} gathered at
// the ray's integer (u, v), per-ray [near, far] clamped into the volume bounds.  Shared by k_build_rays and
// the fused prologue of k_render_rays so both produce the same bits.
// ---------------------------------------------------------------------------------------------
struct RayBounds {
    float rn, rf, vn, vf;
};
__device__ __forceinline__ RayBounds ray_bounds(float ru, float rv, const float* __restrict__ pd,
                                                const float* __restrict__ ps, const float* __restrict__ n0,
                                                int h, int w, int Hr, int Wr, int depth_inv) {
    int u = (int)ru, v = (int)rv;                            // .long(): truncation
    u = u < 0 ? u + Wr : u;                                  // python negative indexing
    v = v < 0 ? v + Hr : v;
    u = u < 0 ? 0 : (u > Wr - 1 ? Wr - 1 : u);
    v = v < 0 ? 0 : (v > Hr - 1 ? Hr - 1 : v);
    const Lerp1 ly = ac_lerp(v, ac_scale(h, Hr), h), lx = ac_lerp(u, ac_scale(w, Wr), w);
    const int o00 = ly.i0 * w + lx.i0, o01 = ly.i0 * w + lx.i1, o10 = ly.i1 * w + lx.i0, o11 = ly.i1 * w + lx.i1;
    const float* n1 = n0 + h * w;
    float d, s, a0, a1;
    if ((h == Hr) && (w == Wr)) {
        d = pd[o00]; s = ps[o00]; a0 = n0[o00]; a1 = n1[o00];
    } else {
        d = ac_blend(ly, lx, pd[o00], pd[o01], pd[o10], pd[o11]);
        s = ac_blend(ly, lx, ps[o00], ps[o01], ps[o10], ps[o11]);
        a0 = ac_blend(ly, lx, n0[o00], n0[o01], n0[o10], n0[o11]);
        a1 = ac_blend(ly, lx, n1[o00], n1[o01], n1[o10], n1[o11]);
    }
    RayBounds r;
    if (depth_inv) {              // utils.py:402-407
        r.rn = d + s; if (r.rn > a0) r.rn = a0;
        r.rf = d - s; if (r.rf < a1) r.rf = a1;
    } else {                      // utils.py:409-413
        r.rn = d - s; if (r.rn < a0) r.rn = a0;
        r.rf = d + s; if (r.rf > a1) r.rf = a1;
    }
    r.vn = a0; r.vf = a1;
    return r;
}

}  // namespace enerf
