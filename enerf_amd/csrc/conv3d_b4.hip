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
// Block = 256 threads = a box of BD x 8 x 16 output voxels, BD/2 voxels per lane.  The haloed input box is staged in LDS
// as [channel quad][voxel] float4 planes: one ds_read_b128 then reads 4 channels of 64 consecutive voxels, bank-conflict
// free (the [voxel][16 channel] layout of the other kernels costs 2 LDS cycles per 16-lane group, MI355X_MICROARCH.md
// §LDS); the plane pitch is chosen so that the staging writes (8-lane groups: 2 voxels x 4 quads) are conflict free too.
// Weights (A) come from a packed global image, 4 distinct 16-byte addresses per wave, through a register ring.
#include "kernels.h"

namespace enerf {

#define ENERF_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

// image: [tap][channel quad cq][slot][row i][r], 48 floats per (tap, cq):
//   slot 0/1: W[4*slot + i][4*cq + r][tap];  slot 2: the depth head W_d[4*cq + r][tap] repeated for every i (zeros when the
//   layer has none) — repeated so that every lane loads it at the same lane-dependent offset as its weight row (a uniform
//   address becomes 27*Cin scalar loads that hipcc hoists and spills).
long long conv3d_b4_packed_floats(int cin) { return 27LL * (cin / 4) * 48; }

__global__ __launch_bounds__(256) void k_conv3d_b4_pack(const float* __restrict__ w, const float* __restrict__ wd, int cin,
                                                        float* __restrict__ packed) {
    const long long total = 27LL * (cin / 4) * 48;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
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

constexpr int b4_plane_voxels(int nvox) { return nvox + ((2 - nvox % 8) + 8) % 8; }   // pitch (in float4) = 2 mod 8

template <int CIN, int BD, bool HEADS>
__global__ __launch_bounds__(256, 2) void k_conv3d_s1_b4(const float* __restrict__ wb4, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ in,
                                                         float* __restrict__ out, float* __restrict__ out2, int relu, int B,
                                                         int D, int H, int W, int nbd, int nbh, int nbw) {
    constexpr int BH = 8, BW = 16, V = BD / 2;                     // voxels per lane
    constexpr int CB = CIN >= 16 ? 16 : CIN, QV = CB / 4, NCB = CIN / CB, NQ = CIN / 4;
    constexpr int HX = BW + 2, HY = BH + 2, HZ = BD + 2, NVOX = HZ * HY * HX;
    constexpr int PLANE = b4_plane_voxels(NVOX) * 4;               // floats per channel-quad plane
    constexpr int NIT = (NVOX * QV + 255) / 256;
    ENERF_DYN_SMEM(float, lds);

    const int tid = threadIdx.x, li = tid & 3;
    const int xl = tid & 15, yl = (tid >> 4) & 7, zl = tid >> 7;   // this lane's voxel (group v adds 2v to z)
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
    const float* wl = wb4 + li * 4;                                // row i = lane & 3 of every block
    const float* lbase = lds + ((zl * HY + yl) * HX + xl) * 4;

#pragma unroll 1
    for (int cb = 0; cb < NCB; ++cb) {
        // the weights of one tap: QV quads x {rows 0-3, rows 4-7 (, depth row)}
        auto issue_a = [&](int tap, float4 (&aq)[QV][HEADS ? 3 : 2]) {
            const float* wt = wl + ((long long)tap * NQ + cb * QV) * 48;
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                aq[q][0] = *reinterpret_cast<const float4*>(wt + q * 48);
                aq[q][1] = *reinterpret_cast<const float4*>(wt + q * 48 + 16);
                if (HEADS) aq[q][2] = *reinterpret_cast<const float4*>(wt + q * 48 + 32);
            }
        };
        float4 aq[3][QV][HEADS ? 3 : 2];
        issue_a(0, aq[0]);
        issue_a(1, aq[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (cb > 0) __syncthreads();
        {   // stage the haloed box as channel-quad planes: unconditional clamped loads, zero-select afterwards
            float4 sv[NIT];
            bool sk[NIT];
            int so[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * 256;
                const int ic = i < NVOX * QV ? i : NVOX * QV - 1;
                const int vx = ic / QV, q = ic - vx * QV;
                const int dx = vx % HX, dy = (vx / HX) % HY, dz = vx / (HX * HY);
                const int gx = x0 + dx - 1, gy = y0 + dy - 1, gz = z0 + dz - 1;
                sk[it] = gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
                so[it] = q * PLANE + vx * 4;
                const long long off = sk[it] ? (((long long)gz * H + gy) * W + gx) : 0;
                sv[it] = *reinterpret_cast<const float4*>(inb + off * CIN + cb * CB + q * 4);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + it * 256;
                if (i < NVOX * QV)
                    *reinterpret_cast<float4*>(lds + so[it]) = sk[it] ? sv[it] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();

        auto read_b = [&](int tap, float4 (&bv)[V][QV]) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int q = 0; q < QV; ++q)
                    bv[v][q] = *reinterpret_cast<const float4*>(lbase + q * PLANE + (((2 * v + kd) * HY + kh) * HX + kw) * 4);
        };
        float4 bq[2][V][QV];
        read_b(0, bq[0]);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 2 < 27) issue_a(tap + 2, aq[(tap + 2) % 3]);
            if (tap + 1 < 27) read_b(tap + 1, bq[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (HEADS) {                                           // depth_conv row: lane-local FMAs
#pragma unroll
                for (int q = 0; q < QV; ++q)
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float4 bb = bq[tap & 1][v][q], wd = aq[tap % 3][q][HEADS ? 2 : 0];
                        dacc[v] = __builtin_fmaf(wd.x, bb.x, dacc[v]);
                        dacc[v] = __builtin_fmaf(wd.y, bb.y, dacc[v]);
                        dacc[v] = __builtin_fmaf(wd.z, bb.z, dacc[v]);
                        dacc[v] = __builtin_fmaf(wd.w, bb.w, dacc[v]);
                        ENERF_PIN_VGPR(dacc[v]);                   // keep the chain here (it is only stored under out2 != nullptr)
                    }
            }
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                const float a0[4] = {aq[tap % 3][q][0].x, aq[tap % 3][q][0].y, aq[tap % 3][q][0].z, aq[tap % 3][q][0].w};
                const float a1[4] = {aq[tap % 3][q][1].x, aq[tap % 3][q][1].y, aq[tap % 3][q][1].z, aq[tap % 3][q][1].w};
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

template <int CIN, int BD, bool HEADS>
static void launch_b4(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W, hipStream_t st) {
    constexpr int CB = CIN >= 16 ? 16 : CIN, QV = CB / 4, NVOX = (BD + 2) * 10 * 18;
    const int nbd = cdiv(D, BD), nbh = cdiv(H, 8), nbw = cdiv(W, 16);
    const size_t shmem = (size_t)QV * b4_plane_voxels(NVOX) * 4 * sizeof(float);
    const unsigned grid = (unsigned)((long long)B * nbd * nbh * nbw);
    ENERF_LAUNCH((k_conv3d_s1_b4<CIN, BD, HEADS>), grid, 256, shmem, st, L.w_b4, L.scale, L.shift, in, out, out2, L.relu, B, D,
                 H, W, nbd, nbh, nbw);
}
bool launch_conv3d_b4(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W, hipStream_t st) {
    if (L.w_b4 == nullptr || L.kind != kConvS1) return false;
    const bool heads = L.cout == 9 && out2 != nullptr;
    if (!(L.cout == 8 && out2 == nullptr) && !heads) return false;
    const bool bd4 = D % 4 == 0;
    if (heads) {                                                   // the fused heads of both nets have Cin = 8
        if (L.cin != 8) return false;
        if (bd4) launch_b4<8, 4, true>(L, in, out, out2, B, D, H, W, st); else launch_b4<8, 2, true>(L, in, out, out2, B, D, H, W, st);
        return true;
    }
#define ENERF_B4(CINV) \
    if (bd4) launch_b4<CINV, 4, false>(L, in, out, out2, B, D, H, W, st); else launch_b4<CINV, 2, false>(L, in, out, out2, B, D, H, W, st); \
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
