// conv3d_pk8.hip — stride-1 3x3x3 convolution for Cout = 8 (+ an optional 9th "depth" output) with TAP PACKING.
//
// The plain kernels put one tap's 8 output channels in the 16 rows of v_mfma_f32_16x16x4_f32, so half of every
// MFMA multiplies zeros, and phase ablation shows these layers' MFMA phase already runs at the pipe rate — the
// only way to make conv0 (both levels) and the fused heads faster is to issue fewer MFMAs.  Here the 16 rows
// carry TWO taps: for every (kd,kh) and every 4-channel k-step
//     MFMA "P": rows 0-7 = W[kw=0], rows 8-15 = W[kw=2]
//     MFMA "Q": rows 0-7 = W[kw=1], rows 8,9,10 = depth_conv taps kw=0,1,2 (heads only)
// both with the SAME B operand, the input voxel at column position p (so one LDS read serves three taps).
// A product in column p belongs to output x = p+1 (kw=0), p (kw=1) or p-1 (kw=2); the shifts are applied once,
// after the whole K loop, with two lane permutes per register.  A 16-column tile therefore yields 14 outputs
// (columns 1..14); boxes advance 14 voxels in x.  72 instead of 108 MFMAs per row tile at Cin=16, and
// 9 instead of 27 LDS reads.  Same LDS-staged box, operand rings and epilogue conventions as conv3d.hip V2.

#include "kernels.h"

namespace enerf {

long long conv3d_pk8_packed_floats(int cin) { return 9LL * (cin / 4) * 2 * 64; }

// packed[((kdkh*KS + ks)*2 + pq)*64 + lane], lane=(g,i).  w: feat/conv weights (8,cin,3,3,3); wd: depth_conv
// (1,cin,3,3,3) or nullptr.
__global__ __launch_bounds__(256) void k_conv3d_pk8_pack(const float* __restrict__ w, const float* __restrict__ wd,
                                                         int cin, float* __restrict__ packed) {
    const int KS = cin / 4, CPL = cin >= 16 ? 4 : cin / 4;
    const long long total = 9LL * KS * 2 * 64;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    long long q = i >> 6;
    const int pq = (int)(q & 1); q >>= 1;
    const int ks = (int)(q % KS);
    const int kdkh = (int)(q / KS);
    const int g = lane >> 4, row = lane & 15;
    const int cb = ks / CPL, r = ks - cb * CPL;
    const int ci = cb * 4 * CPL + g * CPL + r;
    float v = 0.f;
    if (pq == 0) {                                   // P: rows 0-7 kw=0, rows 8-15 kw=2
        const int co = row & 7, kw = row < 8 ? 0 : 2;
        v = w[((long long)co * cin + ci) * 27 + kdkh * 3 + kw];
    } else if (row < 8) {                            // Q: rows 0-7 kw=1
        v = w[((long long)row * cin + ci) * 27 + kdkh * 3 + 1];
    } else if (row < 11 && wd != nullptr) {          // Q: rows 8,9,10 = depth_conv kw=0,1,2
        v = wd[(long long)ci * 27 + kdkh * 3 + (row - 8)];
    }
    packed[i] = v;
}
void launch_conv3d_pk8_pack(const float* w, const float* wd, int cin, float* packed, hipStream_t st) {
    long long total = conv3d_pk8_packed_floats(cin);
    ENERF_LAUNCH_SIMPLE(k_conv3d_pk8_pack, (unsigned)cdivl(total, 256), 256, 0, st, w, wd, cin, packed);
}

template <int CIN, int BD>
__global__ __launch_bounds__(256, 2) void k_conv3d_s1_pk8(const float* __restrict__ wpk, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ in,
                                                          float* __restrict__ out, float* __restrict__ out2, int relu,
                                                          int B, int D, int H, int W, int nbd, int nbh, int nbw) {
    constexpr int BH = 8, OW = 14;                      // outputs per tile row: 14 of the 16 columns
    constexpr int CB = CIN >= 16 ? 16 : CIN;
    constexpr int CPL = CB / 4, NCB = CIN / CB, KS = CIN / 4;
    constexpr int CTW = BD * BH / 4;
    constexpr int HX = 16, HY = BH + 2, HZ = BD + 2, NVOX = HZ * HY * HX, QV = CB / 4;
    constexpr int NIT = (NVOX * QV + 255) / 256;
    constexpr int NAT = CPL * 2;                        // A operands per (kd,kh): CPL k-steps x {P,Q}
    ENERF_DYN_SMEM(float, lds);

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    int t = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh; t /= nbh;
    const int bd = t % nbd;
    const int b = t / nbd;
    const int x0 = bw * OW, y0 = bh * BH, z0 = bd * BD;  // first output voxel of the box; input column p = x0-1+j

    f32x4 accP[CTW], accQ[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) { accP[c] = f32x4{0.f, 0.f, 0.f, 0.f}; accQ[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float* inb = in + (long long)b * D * H * W * CIN;
    const float* wl = wpk + lane;

#pragma unroll 1
    for (int cb = 0; cb < NCB; ++cb) {
        auto issue_a = [&](int kdkh, float (&aq)[NAT]) {
            const float* wt = wl + ((long long)kdkh * KS + cb * CPL) * 2 * 64;
#pragma unroll
            for (int r = 0; r < CPL; ++r) { aq[r * 2] = wt[(r * 2) * 64]; aq[r * 2 + 1] = wt[(r * 2 + 1) * 64]; }
        };
        float aq[3][NAT];
        issue_a(0, aq[0]);
        issue_a(1, aq[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (cb > 0) __syncthreads();
        {   // stage the haloed box (16 columns wide): unconditional clamped loads, zero-select afterwards
            float4 sv[NIT];
            bool sk[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * 256;
                const int ic = i < NVOX * QV ? i : NVOX * QV - 1;
                const int v = ic / QV, q = ic - v * QV;
                const int dx = v % HX, dy = (v / HX) % HY, dz = v / (HX * HY);
                const int gx = x0 + dx - 1, gy = y0 + dy - 1, gz = z0 + dz - 1;
                sk[it] = gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
                const long long off = sk[it] ? (((long long)gz * H + gy) * W + gx) : 0;
                sv[it] = *reinterpret_cast<const float4*>(inb + off * CIN + cb * CB + q * 4);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + it * 256;
                if (i < NVOX * QV)
                    *reinterpret_cast<float4*>(lds + i * 4) = sk[it] ? sv[it] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();

        const float* lbase[CTW];
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const int tile = wv * CTW + c, td = tile / BH, th = tile - td * BH;
            lbase[c] = lds + ((td * HY + th) * HX + j) * CB + g * CPL;
        }
        auto read_b = [&](int kdkh, float (&bv)[CTW][4]) {
            const int off = ((kdkh / 3) * HY + (kdkh % 3)) * HX * CB;
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
#pragma unroll
        for (int s = 0; s < 9; ++s) {                   // s = kd*3 + kh
            if (s + 2 < 9) issue_a(s + 2, aq[(s + 2) % 3]);
            if (s + 1 < 9) read_b(s + 1, bq[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < CPL; ++r)
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    accP[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s % 3][r * 2], bq[s & 1][c][r], accP[c], 0, 0, 0);
                    accQ[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s % 3][r * 2 + 1], bq[s & 1][c][r], accQ[c], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- combine the three kw-shifted partial sums, then BN scale/shift (+ReLU) and store ----
    // lane (g<2, j) finalises channels 4g..4g+3 of the output at column j:  Q[g][j] + P[g][j-1] + P[g+2][j+1]
    // lane (g=2, j) finalises the depth output:  Q[2][j][1] + Q[2][j-1][0] + Q[2][j+1][2]
    const int src_lo = lane - 1, src_hi = lane + 33;    // (g, j-1) and (g+2, j+1)
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lo = __shfl(accP[c][r], src_lo), hi = __shfl(accP[c][r], src_hi);
            y[r] = accQ[c][r] + lo + hi;
        }
        const float d_lo = __shfl(accQ[c][0], lane - 1), d_hi = __shfl(accQ[c][2], lane + 1);
        const float yd = accQ[c][1] + d_lo + d_hi;
        const int tile = wv * CTW + c, td = tile / BH, th = tile - td * BH;
        const int z = z0 + td, yy = y0 + th, x = x0 - 1 + j;
        if (j < 1 || j > OW || z >= D || yy >= H || x >= W) continue;
        const long long o = (((long long)b * D + z) * H + yy) * W + x;
        if (g < 2) {
            const int c0 = 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[r] = y[r] * scale[c0 + r] + shift[c0 + r];
                if (relu) y[r] = fmaxf(y[r], 0.f);
            }
            *reinterpret_cast<float4*>(out + o * 8 + c0) = make_float4(y[0], y[1], y[2], y[3]);
        } else if (g == 2 && out2 != nullptr) {
            out2[o] = yd * scale[8] + shift[8];
        }
    }
}

template <int CIN, int BD>
static void launch_pk8(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W,
                       hipStream_t st) {
    constexpr int CB = CIN >= 16 ? 16 : CIN;
    const int nbd = cdiv(D, BD), nbh = cdiv(H, 8), nbw = cdiv(W, 14);
    const size_t shmem = (size_t)(BD + 2) * 10 * 16 * CB * sizeof(float);
    const unsigned grid = (unsigned)((long long)B * nbd * nbh * nbw);
    ENERF_LAUNCH((k_conv3d_s1_pk8<CIN, BD>), grid, 256, shmem, st, L.w_pk8, L.scale, L.shift, in, out, out2, L.relu, B, D, H,
                 W, nbd, nbh, nbw);
}
bool launch_conv3d_pk8(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W,
                       bool all_layers, hipStream_t st) {
    if (L.w_pk8 == nullptr || L.kind != kConvS1 || !(L.cout == 8 || (L.cout == 9 && out2 != nullptr))) return false;
    // Measured on MI355X (same box, rocprofv3): Cin=16 conv0 80-88 us vs 94 us for the plain LDS kernel; Cin=32
    // conv0 and the Cin=8 heads are no faster (fewer MFMAs, but 14/16 column efficiency, 15 % more blocks and a
    // heavier epilogue eat the gain), so by default only Cin=16 takes this path; `all_layers`
    // (enerf_options_t.conv3d_pk8 == 2) routes every Cout=8(+1) layer here.
    // Re-measured after the VALU work: the Cin=8 fused heads gain at level 1 only (655,360 voxels: 54.3 -> 50.6 us; level 0:
    // 20.9 -> 24.2 us), so they take this path above 512 K voxels.
    const bool big = (long long)B * D * H * W >= (1LL << 19);
    if (!all_layers && !(L.cin == 16 || (L.cin == 8 && big))) return false;
    const bool bd4 = (D % 4 == 0);
    switch (L.cin) {
        case 8: bd4 ? launch_pk8<8, 4>(L, in, out, out2, B, D, H, W, st) : launch_pk8<8, 2>(L, in, out, out2, B, D, H, W, st); return true;
        case 16: bd4 ? launch_pk8<16, 4>(L, in, out, out2, B, D, H, W, st) : launch_pk8<16, 2>(L, in, out, out2, B, D, H, W, st); return true;
        case 32: bd4 ? launch_pk8<32, 4>(L, in, out, out2, B, D, H, W, st) : launch_pk8<32, 2>(L, in, out, out2, B, D, H, W, st); return true;
        default: return false;
    }
}

}  // namespace enerf
