// conv3d_b4.hip — stride-1 3x3x3 convolution for Cout = 8 (+ an optional 9th "depth" output) on the BATCHED 4x4 matrix
// instruction v_mfma_f32_4x4x1_16B_f32.
//
// Why: the 16x16x4 tile needs 16 output channels to be full.  A Cout = 8 layer (conv0 and the fused heads of both
// cost-regularisation nets: a third of the frame) fills 8 of its 16 rows (plain kernels: half of every MFMA multiplies
// zeros) or 12 of 16 with tap packing at 14 useful columns of 16 (conv3d_pk8.hip: 0.66).  The batched shape computes 16
// independent 4x4 outer products (k = 1) per instruction at 0.86 of the 16x16x4 FLOP rate (tools/micro/mfma_shapes.hip:
// 3.9 ns per 512 FLOP against 13.4 ns per 2048), and with the SAME 4x1 weight column in every block it is exactly
//     4 output channels x 64 voxels x 1 input channel   per instruction,
// so two instructions (channels 0-3, 4-7) cover Cout = 8 with nothing wasted: 1.3x fewer matrix cycles than tap packing,
// 1.7x fewer than the plain kernels.
//
// Operand layout of v_mfma_f32_4x4x1_16B_f32: lane l = 4b + i supplies A_b[i] (row i of block b) and B_b[i] (column i);
// lane 4b + j receives D_b[0..3][j] in its four accumulator registers.  With A_b[i] = W[4*half + i][c][tap] for every b,
// LANE = VOXEL: each lane feeds its own voxel's input value and receives its own voxel's four output channels — no
// cross-lane layout at all.  The depth head (one more output channel, heads only) would waste 3 of 4 rows of a third
// instruction; it runs as 27*Cin lane-local v_fmac instead.
//
// Block = 256 threads = a box of BD x 8 x 16 output voxels, BD/2 voxels per lane, 8 input channels per LDS pass.  The
// haloed input box is staged as [channel quad][voxel] float4 planes and the lanes of each ds_read_b128 service group own 16
// consecutive voxels, so a read is bank-conflict free (the [voxel][16 channel] layout of the other kernels costs 2 LDS
// cycles per 16-lane group, MI355X_MICROARCH.md §LDS).  The pass's weights are staged in LDS too (10 KB): an A operand is
// a broadcast ds_read_b128 (4 addresses per wave).  45 KB of LDS and <= 168 VGPRs: three blocks per CU.
#include "kernels.h"

namespace enerf {

#define ENERF_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

// image: [tap][channel quad cq][slot][row i][r], 48 floats per (tap, cq):
//   slot 0/1: W[4*slot + i][4*cq + r][tap];  slot 2: the depth head W_d[4*cq + r][tap] repeated for every i (zeros when the
//   layer has none) — repeated so that every lane loads it at the same lane-dependent offset as its weight row (a uniform
//   address becomes 27*Cin scalar loads that hipcc hoists and spills).
// Round 5: behind it the BROADCAST-A image of the same layer (common.h mfma4_bc) for k_conv3d_s1_b4c:
//   wcb[(cq*14 + r)*64 + l]: column c = 16r + (l >> 2) = (tap*4 + ch)*2 + half  ->  W[4*half + (l & 3)][4*cq + ch][tap]  (216 of 224)
constexpr int kB4cRegs = 14;
long long conv3d_b4_packed_floats(int cin) { return 27LL * (cin / 4) * 48 + (long long)(cin / 4) * kB4cRegs * 64; }

__global__ __launch_bounds__(256) void k_conv3d_b4_pack(const float* __restrict__ w, const float* __restrict__ wd, int cin,
                                                        float* __restrict__ packed) {
    const long long total = 27LL * (cin / 4) * 48;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) {
        const long long k = idx - total;
        if (k >= (long long)(cin / 4) * kB4cRegs * 64) return;
        const int l = (int)(k & 63), rr = (int)((k >> 6) % kB4cRegs), cq = (int)((k >> 6) / kB4cRegs), c = rr * 16 + (l >> 2);
        const int half = c & 1, ch = (c >> 1) & 3, tap = c >> 3;
        packed[idx] = tap < 27 ? w[((long long)(4 * half + (l & 3)) * cin + 4 * cq + ch) * 27 + tap] : 0.f;
        return;
    }
    const int e = (int)(idx % 48), r = e & 3, i = (e >> 2) & 3, slot = e >> 4;
    const long long q = idx / 48;
    const int nq = cin / 4, cq = (int)(q % nq), tap = (int)(q / nq);
    if (slot < 2) packed[idx] = w[((long long)(4 * slot + i) * cin + 4 * cq + r) * 27 + tap];
    else packed[idx] = wd != nullptr ? wd[(long long)(4 * cq + r) * 27 + tap] : 0.f;
}
void launch_conv3d_b4_pack(const float* w, const float* wd, int cin, float* packed, hipStream_t st) {
    const long long total = conv3d_b4_packed_floats(cin);
    ENERF_LAUNCH_SIMPLE(k_conv3d_b4_pack, (unsigned)cdivl(total, 256), 256, 0, st, w, wd, cin, packed);
}

// plane pitch (in float4 voxels): the staging store is a ds_write_b128 (8-lane groups = 4 voxels x 2 quads, 8 slots of
// 16 B per LDS cycle) -> conflict free when the two planes sit 4 slots apart (mod 8)
constexpr int b4_plane_voxels(int nvox) { return nvox + ((4 - nvox % 8) + 8) % 8; }

template <int CIN, int BD, bool HEADS>
__global__ __launch_bounds__(256, 3) void k_conv3d_s1_b4(const float* __restrict__ wb4, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ in,
                                                         float* __restrict__ out, float* __restrict__ out2, int relu, int B,
                                                         int D, int H, int W, int nbd, int nbh, int nbw) {
    constexpr int BH = 8, BW = 16, V = BD / 2;                     // voxels per lane
    constexpr int CB = 8, QV = 2, NCB = CIN / CB, NQ = CIN / 4;    // 8 input channels (2 quads) per LDS pass
    constexpr int NS = HEADS ? 3 : 2;                              // weight slots per (tap, quad): rows 0-3, rows 4-7(, depth)
    constexpr int HX = BW + 2, HY = BH + 2, HZ = BD + 2, NVOX = HZ * HY * HX;
    constexpr int PLANE = b4_plane_voxels(NVOX) * 4;               // floats per channel-quad plane
    constexpr int NWF4 = 27 * QV * 12;                             // float4s of one pass's weights (48 floats per (tap, quad))
    constexpr int NWIT = (NWF4 + 255) / 256;
    ENERF_DYN_SMEM(float, lds);
    float* wlds = lds + QV * PLANE;                                // [tap][quad][slot][row i][r]

    const int tid = threadIdx.x, li = tid & 3, lane = tid & 63, wv = tid >> 6;
    // lane -> voxel.  ds_read_b128 is serviced in four fixed 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same
    // +32; a group is conflict free when its lanes read 16 distinct 16-byte slots.  Any lane may own any voxel here (the
    // MFMA blocks are independent), so group k of a wave takes the 16 consecutive voxels of box row 4*(wave&1) + k.
    const int m = lane & 31;
    const bool g1 = (m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28;
    const int xl = g1 ? (m < 12 ? m - 4 : m < 20 ? m - 8 : m - 16) : (m < 4 ? m : m < 16 ? m - 8 : m - 12);
    const int yl = 4 * (wv & 1) + 2 * (lane >> 5) + (g1 ? 1 : 0), zl = wv >> 1;   // group v adds 2v to z
    int t = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh; t /= nbh;
    const int bd = t % nbd;
    const int b = t / nbd;
    const int x0 = bw * BW, y0 = bh * BH, z0 = bd * BD;

    f32x4 acc[V][2];
    float dacc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { acc[v][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[v][1] = f32x4{0.f, 0.f, 0.f, 0.f}; dacc[v] = 0.f; }
    const float* inb = in + (long long)b * D * H * W * CIN;
    const float* lbase = lds + ((zl * HY + yl) * HX + xl) * 4;
    const float* wbase = wlds + li * 4;                            // row i = lane & 3 of every block: 4 addresses per wave

#pragma unroll 1
    for (int cb = 0; cb < NCB; ++cb) {
        if (cb > 0) __syncthreads();
        {   // stage the haloed box as channel-quad planes (unconditional clamped loads, zero-select afterwards) and this
            // pass's weights: from LDS an A operand costs a broadcast ds_read_b128; from global it was 16 texture-unit
            // cycles per load for 64 unique bytes — 85 % of the address pipe at two blocks per CU.
            // Row-based mapping: a thread keeps its (column, quad) of a box row and walks rows slot, slot+7, ... — the only
            // index arithmetic per load is one division of the row number (the flat index -> (dx,dy,dz,q) form cost ~45 VALU
            // instructions per load, ~15 % of the kernel's issue slots: the kernel is issue bound at three blocks per CU).
            constexpr int RW = HX * QV, SLOTS = 256 / RW, NR = HZ * HY, NITR = (NR + SLOTS - 1) / SLOTS;
            const int sub = tid % RW, slot = tid / RW, sdx = sub / QV, sq = sub - sdx * QV;
            const int gx = x0 + sdx - 1;
            const bool colok = slot < SLOTS && gx >= 0 && gx < W;
            float4 sv[NITR], wq[NWIT];
            bool sk[NITR];
            const float* src = inb + cb * CB + sq * 4;
#pragma unroll
            for (int it = 0; it < NITR; ++it) {
                const int r = slot + it * SLOTS, dz = r / HY, dy = r - dz * HY;
                const int gy = y0 + dy - 1, gz = z0 + dz - 1;
                sk[it] = colok && r < NR && gy >= 0 && gy < H && gz >= 0 && gz < D;
                const long long off = sk[it] ? (((long long)gz * H + gy) * W + gx) : 0;
                sv[it] = *reinterpret_cast<const float4*>(src + off * CIN);
            }
#pragma unroll
            for (int it = 0; it < NWIT; ++it) {                    // the pass's two quads of a tap are 96 contiguous floats
                const int i = tid + it * 256, ic = i < NWF4 ? i : NWF4 - 1;
                const int tap = ic / 24, e = ic - tap * 24;
                wq[it] = *reinterpret_cast<const float4*>(wb4 + ((long long)tap * NQ + cb * QV) * 48 + e * 4);
            }
            float* dst = lds + sq * PLANE + (slot * HX + sdx) * 4;
#pragma unroll
            for (int it = 0; it < NITR; ++it) {
                const int r = slot + it * SLOTS;
                if (slot < SLOTS && r < NR)
                    *reinterpret_cast<float4*>(dst + it * SLOTS * HX * 4) = sk[it] ? sv[it] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int it = 0; it < NWIT; ++it) {
                const int i = tid + it * 256;
                if (i < NWF4) *reinterpret_cast<float4*>(wlds + i * 4) = wq[it];
            }
        }
        __syncthreads();

        auto read_a = [&](int tap, float4 (&aq)[QV][NS]) {
#pragma unroll
            for (int q = 0; q < QV; ++q)
#pragma unroll
                for (int sl = 0; sl < NS; ++sl)
                    aq[q][sl] = *reinterpret_cast<const float4*>(wbase + (tap * QV + q) * 48 + sl * 16);
        };
        auto read_b = [&](int tap, float4 (&bv)[V][QV]) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int q = 0; q < QV; ++q)
                    bv[v][q] = *reinterpret_cast<const float4*>(lbase + q * PLANE + (((2 * v + kd) * HY + kh) * HX + kw) * 4);
        };
        float4 aq[2][QV][NS], bq[2][V][QV];
        read_a(0, aq[0]);
        read_b(0, bq[0]);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 1 < 27) { read_a(tap + 1, aq[(tap + 1) & 1]); read_b(tap + 1, bq[(tap + 1) & 1]); }
            __builtin_amdgcn_sched_barrier(0);
            if (HEADS) {                                           // depth_conv row: lane-local FMAs
#pragma unroll
                for (int q = 0; q < QV; ++q)
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float4 bb = bq[tap & 1][v][q], wd = aq[tap & 1][q][NS - 1];
                        dacc[v] = __builtin_fmaf(wd.x, bb.x, dacc[v]);
                        dacc[v] = __builtin_fmaf(wd.y, bb.y, dacc[v]);
                        dacc[v] = __builtin_fmaf(wd.z, bb.z, dacc[v]);
                        dacc[v] = __builtin_fmaf(wd.w, bb.w, dacc[v]);
                        ENERF_PIN_VGPR(dacc[v]);                   // keep the chain here (it is only stored under out2 != nullptr)
                    }
            }
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                const float4 A0 = aq[tap & 1][q][0], A1 = aq[tap & 1][q][1];
                const float a0[4] = {A0.x, A0.y, A0.z, A0.w}, a1[4] = {A1.x, A1.y, A1.z, A1.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float4 bb = bq[tap & 1][v][q];
                        const float bx = r == 0 ? bb.x : r == 1 ? bb.y : r == 2 ? bb.z : bb.w;
                        acc[v][0] = ENERF_MFMA4(a0[r], bx, acc[v][0]);
                        acc[v][1] = ENERF_MFMA4(a1[r], bx, acc[v][1]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: BN scale/shift (+ReLU), two float4 stores per voxel; the depth head goes to out2 ----
    const int x = x0 + xl, y = y0 + yl;
    if (x >= W || y >= H) return;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int z = z0 + zl + 2 * v;
        if (z >= D) continue;
        const long long o = (((long long)b * D + z) * H + y) * W + x;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                yv[r] = acc[v][half][r] * scale[4 * half + r] + shift[4 * half + r];
                if (relu) yv[r] = fmaxf(yv[r], 0.f);
            }
            *reinterpret_cast<float4*>(out + o * 8 + 4 * half) = make_float4(yv[0], yv[1], yv[2], yv[3]);
        }
        if (HEADS && out2 != nullptr) out2[o] = dacc[v] * scale[8] + shift[8];
    }
}

// =====================================================================================================================
// k_conv3d_s1_b4g — the same convolution with ASYNCHRONOUS staging (global_load_lds) and two LDS buffers.
// Why: the kernel above is a strict stage -> barrier -> 27 taps -> barrier chain per 8-channel pass; the co-resident blocks of
// a CU run in lockstep, so nothing covers the staging.  Measured (r03): level-1 conv0 53.6 us against 33.8 us of batched-4x4
// issue time; level-0 conv0 (480 blocks for 256 CUs: <= 2 waves per SIMD) 47.1 us against 25.4.  Here a pass is ONE channel
// quad (4 input channels): its haloed box is a single [voxel] float4 plane (17 KB) + 3.4-5 KB of weights, which every wave
// requests with <= 7 global_load_lds instructions and no VGPRs; the copy of pass q+1 is issued right after the barrier that
// opens pass q and lands during pass q's 27 x 16 MFMAs.  One bare s_barrier per pass (the copy stays in flight across it),
// two 21.5-23.5 KB buffers: three blocks per CU as before.  Per-lane source offsets (the halo geometry) are computed once.
// =====================================================================================================================
__device__ float g_b4_zeros[64];                                  // source of the zero padding (a lane's 16 bytes)

template <int CIN, int BD, bool HEADS>
__global__ __launch_bounds__(256, 3) void k_conv3d_s1_b4g(const float* __restrict__ wb4, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ in,
                                                          float* __restrict__ out, float* __restrict__ out2, int relu, int B,
                                                          int D, int H, int W, int nbd, int nbh, int nbw, int in_planar) {
    constexpr int BH = 8, BW = 16, V = BD / 2;
    constexpr int NQ = CIN / 4, NS = HEADS ? 3 : 2;
    constexpr int HX = BW + 2, HY = BH + 2, HZ = BD + 2, NVOX = HZ * HY * HX;
    constexpr int NCH = (NVOX + 63) / 64, PLANE = NCH * 64 * 4;   // plane chunks of 64 voxels (one glds instruction each)
    constexpr int NW4 = 27 * NS * 4, NWCH = (NW4 + 63) / 64;      // float4s / chunks of one pass's weights
    constexpr int BUF = PLANE + NWCH * 64 * 4;                    // floats per buffer
    constexpr int MYCH = (NCH + 3) / 4, MYW = (NWCH + 3) / 4;     // chunks per wave
    ENERF_DYN_SMEM(float, lds);

    const int tid = threadIdx.x, li = tid & 3, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31;
    const bool g1 = (m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28;
    const int xl = g1 ? (m < 12 ? m - 4 : m < 20 ? m - 8 : m - 16) : (m < 4 ? m : m < 16 ? m - 8 : m - 12);
    const int yl = 4 * (wv & 1) + 2 * (lane >> 5) + (g1 ? 1 : 0), zl = wv >> 1;
    int t = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh; t /= nbh;
    const int bd = t % nbd;
    const int b = t / nbd;
    const int x0 = bw * BW, y0 = bh * BH, z0 = bd * BD;
    const float* inb = in + (long long)b * D * H * W * CIN;
    // input addressing: channels-last (voxel stride CIN, quad stride 4) or channel-quad planes (voxel stride 4, quad stride
    // D*H*W*4): with planes a pass reads 16 CONSECUTIVE bytes per voxel and whole cache lines per box row
    const long long vstride = in_planar ? 4 : CIN, qstride = in_planar ? (long long)D * H * W * 4 : 4;

    // ---- per-lane sources, computed once: chunk c = wv + 4k of the plane holds voxels 64c .. 64c+63 of the haloed box ----
    long long src_off[MYCH];                                       // element offset of the voxel (channel 0), or -1: zeros
#pragma unroll
    for (int k = 0; k < MYCH; ++k) {
        const int v = (wv + 4 * k) * 64 + lane;
        const int dz = v / (HY * HX), r2 = v - dz * (HY * HX), dy = r2 / HX, dx = r2 - dy * HX;
        const int gx = x0 + dx - 1, gy = y0 + dy - 1, gz = z0 + dz - 1;
        const bool ok = v < NVOX && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        src_off[k] = ok ? (((long long)gz * H + gy) * W + gx) * vstride : -1;
    }
    int w_off[MYW];                                                // float offset inside (tap, quad)'s 48 floats + tap stride
#pragma unroll
    for (int k = 0; k < MYW; ++k) {
        const int i = (wv + 4 * k) * 64 + lane, ic = i < NW4 ? i : NW4 - 1;
        const int tap = ic / (NS * 4), e = ic - tap * (NS * 4);
        w_off[k] = tap * NQ * 48 + e * 4;
    }
    auto issue = [&](int cq, float* buf) {
#pragma unroll
        for (int k = 0; k < MYCH; ++k)
            if (wv + 4 * k < NCH)                                  // wave-uniform
                glds16(src_off[k] >= 0 ? inb + src_off[k] + cq * qstride : g_b4_zeros, buf + (wv + 4 * k) * 256, lane);
#pragma unroll
        for (int k = 0; k < MYW; ++k)
            if (wv + 4 * k < NWCH)
                glds16(wb4 + w_off[k] + cq * 48, buf + PLANE + (wv + 4 * k) * 256, lane);
    };

    f32x4 acc[V][2];
    float dacc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { acc[v][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[v][1] = f32x4{0.f, 0.f, 0.f, 0.f}; dacc[v] = 0.f; }

    issue(0, lds);
#pragma unroll 1
    for (int cq = 0; cq < NQ; ++cq) {
        float* buf = lds + (cq & 1) * BUF;
        glds_wait_all();                                           // this wave's copies of pass cq have landed
        block_barrier_raw();                                       // everyone's have; everyone is done reading the other buffer
        if (cq + 1 < NQ) issue(cq + 1, lds + ((cq + 1) & 1) * BUF);
        const float* lbase = buf + ((zl * HY + yl) * HX + xl) * 4;
        const float* wbase = buf + PLANE + li * 4;
        auto read_a = [&](int tap, float4 (&aq)[NS]) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) aq[sl] = *reinterpret_cast<const float4*>(wbase + tap * (NS * 16) + sl * 16);
        };
        auto read_b = [&](int tap, float4 (&bv)[V]) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
            for (int v = 0; v < V; ++v) bv[v] = *reinterpret_cast<const float4*>(lbase + (((2 * v + kd) * HY + kh) * HX + kw) * 4);
        };
        float4 aq[2][NS], bq[2][V];
        read_a(0, aq[0]);
        read_b(0, bq[0]);
        constexpr int NTAP = 27;
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            if (tap + 1 < NTAP) { read_a(tap + 1, aq[(tap + 1) & 1]); read_b(tap + 1, bq[(tap + 1) & 1]); }
            __builtin_amdgcn_sched_barrier(0);
            if (HEADS) {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float4 bb = bq[tap & 1][v], wd = aq[tap & 1][NS - 1];
                    dacc[v] = __builtin_fmaf(wd.x, bb.x, dacc[v]);
                    dacc[v] = __builtin_fmaf(wd.y, bb.y, dacc[v]);
                    dacc[v] = __builtin_fmaf(wd.z, bb.z, dacc[v]);
                    dacc[v] = __builtin_fmaf(wd.w, bb.w, dacc[v]);
                    ENERF_PIN_VGPR(dacc[v]);
                }
            }
            const float4 A0 = aq[tap & 1][0], A1 = aq[tap & 1][1];
            const float a0[4] = {A0.x, A0.y, A0.z, A0.w}, a1[4] = {A1.x, A1.y, A1.z, A1.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float4 bb = bq[tap & 1][v];
                    const float bx = r == 0 ? bb.x : r == 1 ? bb.y : r == 2 ? bb.z : bb.w;
                    acc[v][0] = ENERF_MFMA4(a0[r], bx, acc[v][0]);
                    acc[v][1] = ENERF_MFMA4(a1[r], bx, acc[v][1]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    const int x = x0 + xl, y = y0 + yl;
    if (x >= W || y >= H) return;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int z = z0 + zl + 2 * v;
        if (z >= D) continue;
        const long long o = (((long long)b * D + z) * H + y) * W + x;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                yv[r] = acc[v][half][r] * scale[4 * half + r] + shift[4 * half + r];
                if (relu) yv[r] = fmaxf(yv[r], 0.f);
            }
            *reinterpret_cast<float4*>(out + o * 8 + 4 * half) = make_float4(yv[0], yv[1], yv[2], yv[3]);
        }
        if (HEADS && out2 != nullptr) out2[o] = dacc[v] * scale[8] + shift[8];
    }
}

// =====================================================================================================================
// k_conv3d_s1_b4c (round 5) — k_conv3d_s1_b4g with the pass's weights in REGISTERS: 216 weight columns of a channel quad are 14
// VGPRs (A-operand broadcast, common.h mfma4_bc), loaded with 14 coalesced 256-byte loads one pass ahead.  What leaves the
// kernel: the weight half of every buffer (3.4-5 KB of LDS-DMA per pass and wave set), two of the four (heads: three of the
// five) ds_read_b128 per tap.  The depth head's 27 x 4 weights (heads only) still travel as one DMA chunk and are read as a
// uniform-address float4.  Same arithmetic and order as b4g: bit-identical outputs.
// =====================================================================================================================
template <int CIN, int BD, bool HEADS>
__global__ __launch_bounds__(256, 3) void k_conv3d_s1_b4c(const float* __restrict__ wb4, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ in,
                                                          float* __restrict__ out, float* __restrict__ out2, int relu, int B,
                                                          int D, int H, int W, int nbd, int nbh, int nbw, int in_planar) {
    constexpr int BH = 8, BW = 16, V = BD / 2;
    constexpr int NQ = CIN / 4;
    constexpr int HX = BW + 2, HY = BH + 2, HZ = BD + 2, NVOX = HZ * HY * HX;
    constexpr int NCH = (NVOX + 63) / 64, PLANE = NCH * 64 * 4;   // plane chunks of 64 voxels (one glds instruction each)
    constexpr int BUF = PLANE + (HEADS ? 256 : 0);                // + one chunk: the depth head's [tap][4] weights of the pass
    constexpr int MYCH = (NCH + 3) / 4;
    static_assert(NQ % 2 == 0, "passes are processed in pairs (two register sets)");
    ENERF_DYN_SMEM(float, lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31;
    const bool g1 = (m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28;
    const int xl = g1 ? (m < 12 ? m - 4 : m < 20 ? m - 8 : m - 16) : (m < 4 ? m : m < 16 ? m - 8 : m - 12);
    const int yl = 4 * (wv & 1) + 2 * (lane >> 5) + (g1 ? 1 : 0), zl = wv >> 1;
    int t = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh; t /= nbh;
    const int bd = t % nbd;
    const int b = t / nbd;
    const int x0 = bw * BW, y0 = bh * BH, z0 = bd * BD;
    const float* inb = in + (long long)b * D * H * W * CIN;
    const long long vstride = in_planar ? 4 : CIN, qstride = in_planar ? (long long)D * H * W * 4 : 4;
    const float* wcb = wb4 + 27 * NQ * 48;                        // the broadcast-A image follows the b4 image

    long long src_off[MYCH];                                       // element offset of the voxel (channel 0), or -1: zeros
#pragma unroll
    for (int k = 0; k < MYCH; ++k) {
        const int v = (wv + 4 * k) * 64 + lane;
        const int dz = v / (HY * HX), r2 = v - dz * (HY * HX), dy = r2 / HX, dx = r2 - dy * HX;
        const int gx = x0 + dx - 1, gy = y0 + dy - 1, gz = z0 + dz - 1;
        const bool ok = v < NVOX && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        src_off[k] = ok ? (((long long)gz * H + gy) * W + gx) * vstride : -1;
    }
    const int wd_off = ((lane < 27 ? lane : 26) * NQ * 3 + 2) * 16;   // b4 image: [tap][cq][slot 2][row 0][r]
    auto issue = [&](int cq, float* buf) {
#pragma unroll
        for (int k = 0; k < MYCH; ++k)
            if (wv + 4 * k < NCH)                                  // wave-uniform
                glds16(src_off[k] >= 0 ? inb + src_off[k] + cq * qstride : g_b4_zeros, buf + (wv + 4 * k) * 256, lane);
        if (HEADS && wv == (NCH & 3)) glds16(wb4 + wd_off + cq * 48, buf + PLANE, lane);
    };
    auto load_w = [&](int cq, float (&wr)[kB4cRegs]) {
#pragma unroll
        for (int r = 0; r < kB4cRegs; ++r) wr[r] = wcb[(cq * kB4cRegs + r) * 64 + lane];
    };

    f32x4 acc[V][2];
    float dacc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { acc[v][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[v][1] = f32x4{0.f, 0.f, 0.f, 0.f}; dacc[v] = 0.f; }

    // one pass (channel quad cq) from buffer cq & 1 with the weights in `wr`; the next pass's box + weights are requested first
    auto pass = [&](int cq, const float (&wr)[kB4cRegs], float (&wnext)[kB4cRegs]) {
        float* buf = lds + (cq & 1) * BUF;
        glds_wait_all();                                           // this wave's copies of pass cq have landed
        block_barrier_raw();                                       // everyone's have; everyone is done reading the other buffer
        if (cq + 1 < NQ) {
            issue(cq + 1, lds + ((cq + 1) & 1) * BUF);
            load_w(cq + 1, wnext);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float* lbase = buf + ((zl * HY + yl) * HX + xl) * 4;
        const float* wdl = buf + PLANE;
        auto read_b = [&](int tap, float4 (&bv)[V], float4& wd) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
            for (int v = 0; v < V; ++v) bv[v] = *reinterpret_cast<const float4*>(lbase + (((2 * v + kd) * HY + kh) * HX + kw) * 4);
            if (HEADS) wd = *reinterpret_cast<const float4*>(wdl + tap * 4);
        };
        float4 bq[2][V], wq[2];
        read_b(0, bq[0], wq[0]);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 1 < 27) read_b(tap + 1, bq[(tap + 1) & 1], wq[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (HEADS) {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float4 bb = bq[tap & 1][v], wd = wq[tap & 1];
                    dacc[v] = __builtin_fmaf(wd.x, bb.x, dacc[v]);
                    dacc[v] = __builtin_fmaf(wd.y, bb.y, dacc[v]);
                    dacc[v] = __builtin_fmaf(wd.z, bb.z, dacc[v]);
                    dacc[v] = __builtin_fmaf(wd.w, bb.w, dacc[v]);
                    ENERF_PIN_VGPR(dacc[v]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float4 bb = bq[tap & 1][v];
                    const float bx = r == 0 ? bb.x : r == 1 ? bb.y : r == 2 ? bb.z : bb.w;
                    const int c = tap * 8 + r * 2;                 // column (tap*4 + ch)*2 + half
                    acc[v][0] = mfma4_bc(wr[c >> 4], bx, acc[v][0], c & 15);
                    acc[v][1] = mfma4_bc(wr[(c + 1) >> 4], bx, acc[v][1], (c + 1) & 15);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    float w0[kB4cRegs], w1[kB4cRegs];
    issue(0, lds);
    load_w(0, w0);
#pragma unroll 1
    for (int cq = 0; cq < NQ; cq += 2) {
        pass(cq, w0, w1);
        pass(cq + 1, w1, w0);
    }

    const int x = x0 + xl, y = y0 + yl;
    if (x >= W || y >= H) return;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int z = z0 + zl + 2 * v;
        if (z >= D) continue;
        const long long o = (((long long)b * D + z) * H + y) * W + x;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                yv[r] = acc[v][half][r] * scale[4 * half + r] + shift[4 * half + r];
                if (relu) yv[r] = fmaxf(yv[r], 0.f);
            }
            *reinterpret_cast<float4*>(out + o * 8 + 4 * half) = make_float4(yv[0], yv[1], yv[2], yv[3]);
        }
        if (HEADS && out2 != nullptr) out2[o] = dacc[v] * scale[8] + shift[8];
    }
}

// Measured (profiles/r05_ab_b4c_conv3d_upfront.txt): 34.8 KB of LDS per block lets FOUR blocks share a CU where b4g has three.  With >= 8 boxes
// per CU that pays (zju level-0 conv0, 4096 boxes: 79.4 -> 65.3 us; zju level 1: 133.8 -> 131.0); at ~5 boxes per CU the
// occupancy quantisation loses (dtu level-1 conv0, 1280 boxes = 4 + 1 instead of 3 + 2 per CU: 46.1 -> 51.4 us), and the
// heads / dtu level 0 (1920 boxes) are unchanged: the tap loops were already at the instruction's rate.
// Routing (round 5, second pass): what decides is the occupancy QUANTISATION, not the box count — with k co-resident blocks per CU a
// layer of n boxes needs ceil(n / (CUs k)) rounds of k slots: zju level 0 (1024 boxes = exactly one round of four: 79.4 -> 65.3 us) and
// zju level 1 (4096 = four rounds of four instead of 5.33 of three) want b4c, dtu level 1 (1280 = 3 + 2 instead of 4 + 1) wants b4g.
// b4c is taken when its slot-rounds ceil(n / (4 CUs)) * 4 do not exceed b4g's ceil(n / (3 CUs)) * 3.
#ifndef ENERF_B4_CB
#define ENERF_B4_CB 1                // 1: route by slot-rounds (above); 2: b4c always; 0: never
#endif
template <int CIN, int BD, bool HEADS>
static void launch_b4g(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W, hipStream_t st) {
    constexpr int NVOX = (BD + 2) * 10 * 18, NCH = (NVOX + 63) / 64, NS = HEADS ? 3 : 2, NWCH = (27 * NS * 4 + 63) / 64;
    const int nbd = cdiv(D, BD), nbh = cdiv(H, 8), nbw = cdiv(W, 16);
    const unsigned grid = (unsigned)((long long)B * nbd * nbh * nbw);
    const long long cus_ = device_cu_count();
    // (half-depth boxes — 2 x 8 x 16 on the register-weight kernel, five co-resident blocks per CU — measured in round 5 and not kept:
    // tools/patches/r06_pruned_knobs.diff)
    const long long sr4 = cdivl(grid, 4 * cus_) * 4, sr3 = cdivl(grid, 3 * cus_) * 3;
    if (ENERF_B4_CB == 2 || (ENERF_B4_CB == 1 && sr4 <= sr3)) {
        const size_t shmem_c = (size_t)2 * (NCH + (HEADS ? 1 : 0)) * 64 * 4 * sizeof(float);
        ENERF_LAUNCH((k_conv3d_s1_b4c<CIN, BD, HEADS>), grid, 256, shmem_c, st, L.w_b4, L.scale, L.shift, in, out, out2, L.relu, B,
                     D, H, W, nbd, nbh, nbw, L.in_planar);
        return;
    }
    const size_t shmem = (size_t)2 * (NCH + NWCH) * 64 * 4 * sizeof(float);
    ENERF_LAUNCH((k_conv3d_s1_b4g<CIN, BD, HEADS>), grid, 256, shmem, st, L.w_b4, L.scale, L.shift, in, out, out2, L.relu, B, D,
                 H, W, nbd, nbh, nbw, L.in_planar);
}

template <int CIN, int BD, bool HEADS>
static void launch_b4(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W, hipStream_t st) {
    constexpr int QV = 2, NVOX = (BD + 2) * 10 * 18;
    const int nbd = cdiv(D, BD), nbh = cdiv(H, 8), nbw = cdiv(W, 16);
    const size_t shmem = ((size_t)QV * b4_plane_voxels(NVOX) * 4 + 27 * QV * 48) * sizeof(float);   // box planes + one pass of weights
    const unsigned grid = (unsigned)((long long)B * nbd * nbh * nbw);
    ENERF_LAUNCH((k_conv3d_s1_b4<CIN, BD, HEADS>), grid, 256, shmem, st, L.w_b4, L.scale, L.shift, in, out, out2, L.relu, B, D,
                 H, W, nbd, nbh, nbw);
}
bool launch_conv3d_b4(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W, bool glds,
                      hipStream_t st) {
    if (L.w_b4 == nullptr || L.kind != kConvS1) return false;
    if (L.in_planar && !(glds && D % 4 == 0)) return false;       // only the glds kernel reads quad planes
    const bool heads = L.cout == 9 && out2 != nullptr;
    if (!(L.cout == 8 && out2 == nullptr) && !heads) return false;
    const bool bd4 = D % 4 == 0;
    if (heads) {                                                   // the fused heads of both nets have Cin = 8
        if (L.cin != 8) return false;
        if (glds && bd4) launch_b4g<8, 4, true>(L, in, out, out2, B, D, H, W, st);
        else if (bd4) launch_b4<8, 4, true>(L, in, out, out2, B, D, H, W, st);
        else launch_b4<8, 2, true>(L, in, out, out2, B, D, H, W, st);
        return true;
    }
#define ENERF_B4(CINV) \
    if (glds && bd4) launch_b4g<CINV, 4, false>(L, in, out, out2, B, D, H, W, st); \
    else if (bd4) launch_b4<CINV, 4, false>(L, in, out, out2, B, D, H, W, st); \
    else launch_b4<CINV, 2, false>(L, in, out, out2, B, D, H, W, st); \
    return true;
    switch (L.cin) {
        case 8: ENERF_B4(8)
        case 16: ENERF_B4(16)
        case 32: ENERF_B4(32)
        default: return false;
    }
#undef ENERF_B4
}

}  // namespace enerf
