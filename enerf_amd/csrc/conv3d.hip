// conv3d.hip — 3x3x3 convolutions of the cost-regularisation U-Nets (cost_reg_net.py:4-86,
// ConvBnReLU3D utils.py:22-33) as implicit GEMMs on the fp32 matrix cores.
//
//   Conv3d(k3,p1,s1|s2,no bias) [+BN eval] [+ReLU]          kinds kConvS1 / kConvS2
//   ConvTranspose3d(k3,s2,p1,op1,no bias) + BN + skip add     kind  kConvT2 (8 output-parity classes, 1..8 taps)
//
// GEMM view: D[cout][voxel] += W[cout][k] * X[k][voxel], k = (tap, cin).  v_mfma_f32_16x16x4_f32:
//   A (weights)     lane l holds W[row = l&15][k = l>>4]
//   B (activations) lane l holds X[k = l>>4][col = l&15]
//   D               lane l holds rows 4*(l>>4)+r (r=0..3) of column l&15
// so a lane owns ONE voxel (column) and 4 consecutive output channels: the epilogue (BN scale/shift,
// skip add, ReLU) is lane-local and the store is one float4 per lane (voxel-major, channels-last).
// Activations are channels-last (B,D,H,W,C): the B operand of a tap is a single float4 (float2 for
// C=8) per lane — lane group g = l>>4 reads channels [4g,4g+4) of its voxel — and register r of that
// load is k-step r.  Because the K order of a dot product is free, the packed weight image is simply
// permuted to match (conv3d_pack), so there is no cross-lane shuffle anywhere.
// fp32 MFMA is exact fp32 (a k-ordered fmaf chain), which keeps the 1e-3 PSNR parity bar.
//
// Roofline: MFMA-bound (fp32 157.3 TF peak).  Algorithmic FLOPs = 2*27*cin*cout per output voxel (s1).

#include "kernels.h"

namespace enerf {

// ---- tap enumeration ---------------------------------------------------------------------------
// conv: tap t = (kd*3+kh)*3+kw, input offset (kd-1, kh-1, kw-1) from stride*o.
// convT: output o = 2q + par. par 0: (k=1, dq=0).  par 1: idx 0 -> (k=0, dq=+1), idx 1 -> (k=2, dq=0).
__host__ __device__ __forceinline__ void convt_axis(int par, int idx, int& k, int& dq) {
    if (par == 0) { k = 1; dq = 0; }
    else if (idx == 0) { k = 0; dq = 1; }
    else { k = 2; dq = 0; }
}
__host__ __device__ __forceinline__ int convt_class_ntaps(int cls) {
    return (1 + ((cls >> 2) & 1)) * (1 + ((cls >> 1) & 1)) * (1 + (cls & 1));
}
__host__ __device__ __forceinline__ int convt_class_offset(int cls) {
    int o = 0;
    for (int c = 0; c < cls; ++c) o += convt_class_ntaps(c);
    return o;
}

__host__ __device__ __forceinline__ int conv_cpl(int cin) { return cin >= 16 ? 4 : cin / 4; }  // channels/lane/load

long long conv3d_packed_floats(int cin, int cout, int kind) {
    (void)kind;
    int rt = cdiv(cout, 16);
    return 27LL * (cin / 4) * rt * 64;
}

// packed[((tap*KS + ks)*RT + rt)*64 + lane],  lane=(g,i): W[cout=rt*16+i][cin=chan(ks,g)][tap]
__global__ __launch_bounds__(256) void k_conv3d_pack(const float* __restrict__ w, const float* __restrict__ w2, int cout1,
                                                     const float* __restrict__ bn_w,
                                                     const float* __restrict__ bn_b, const float* __restrict__ bn_mean,
                                                     const float* __restrict__ bn_var, float eps, int cin, int cout,
                                                     int kind, float* __restrict__ packed, float* __restrict__ scale,
                                                     float* __restrict__ shift) {
    int RT = cdiv(cout, 16), KS = cin / 4, CPL = conv_cpl(cin);
    long long total = 27LL * KS * RT * 64;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < RT * 16) {
        int co = (int)i;
        float sc = 1.f, sh = 0.f;
        if (co < cout && bn_w != nullptr) {
            sc = bn_w[co] / sqrtf(bn_var[co] + eps);
            sh = bn_b[co] - bn_mean[co] * sc;
        }
        scale[co] = sc;
        shift[co] = sh;
    }
    if (i >= total) return;
    int lane = (int)(i & 63);
    long long q = i >> 6;
    int rt = (int)(q % RT); q /= RT;
    int ks = (int)(q % KS);
    int tap = (int)(q / KS);
    int g = lane >> 4, co = rt * 16 + (lane & 15);
    int cb = ks / CPL, r = ks - cb * CPL;
    int ci = cb * 4 * CPL + g * CPL + r;
    int kd, kh, kw;
    if (kind == kConvT2) {
        int cls = 0, off = 0;
        while (cls < 7 && tap >= off + convt_class_ntaps(cls)) { off += convt_class_ntaps(cls); ++cls; }
        int idx = tap - off;
        int pw = cls & 1, ph = (cls >> 1) & 1, pd = (cls >> 2) & 1;
        int nw = 1 + pw, nh = 1 + ph;
        int tw = idx % nw, th = (idx / nw) % nh, td = idx / (nw * nh);
        int dq;
        convt_axis(pd, td, kd, dq);
        convt_axis(ph, th, kh, dq);
        convt_axis(pw, tw, kw, dq);
        (void)dq;
    } else {
        kw = tap % 3; kh = (tap / 3) % 3; kd = tap / 9;
    }
    int t = (kd * 3 + kh) * 3 + kw;
    float v = 0.f;
    // rows [0,cout1) come from w, rows [cout1,cout) from w2 (fused heads: feat_conv ++ depth_conv)
    if (co < cout) {
        if (kind == kConvT2) v = w[((long long)ci * cout + co) * 27 + t];
        else if (co < cout1) v = w[((long long)co * cin + ci) * 27 + t];
        else v = w2[((long long)(co - cout1) * cin + ci) * 27 + t];
    }
    packed[i] = v;
}
void launch_conv3d_pack(const float* w, const float* w2, int cout1, const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var,
                        float eps, int cin, int cout, int kind, float* packed, float* scale, float* shift,
                        hipStream_t st) {
    long long total = conv3d_packed_floats(cin, cout, kind);
    ENERF_LAUNCH_SIMPLE(k_conv3d_pack, (unsigned)cdivl(total, 256), 256, 0, st, w, w2, cout1, bn_w, bn_b, bn_mean, bn_var, eps, cin,
                        cout, kind, packed, scale, shift);
}

// 256 B of zeros: loads of padding taps are pointed here (any CIN <= 64 floats from a lane's channel offset)
__device__ float g_conv_zeros[128];

// Input-voxel offset (per axis, relative to the tile's tap-(0,0,0) voxel) of tap t.
//   conv (stride 1/2): t = (kd*3+kh)*3+kw, offsets = (kd,kh,kw).
//   convT stride 2: parity class (pd,ph,pw) has (1+pd)(1+ph)(1+pw) taps, offset dq in {0,1} per axis.
template <int KIND>
__device__ __forceinline__ void tap_offsets(int t, int ntw, int nth, int pd, int ph, int pw, int& od, int& oh, int& ow) {
    if (KIND == kConvT2) {
        const int tw = (ntw == 2) ? (t & 1) : 0;
        const int t2 = (ntw == 2) ? (t >> 1) : t;
        const int th = (nth == 2) ? (t2 & 1) : 0;
        const int td = (nth == 2) ? (t2 >> 1) : t2;
        int k;
        convt_axis(pd, td, k, od);
        convt_axis(ph, th, k, oh);
        convt_axis(pw, tw, k, ow);
    } else {
        ow = t % 3; oh = (t / 3) % 3; od = t / 9;
    }
}

// ---- the implicit-GEMM kernel ---------------------------------------------------------------------
// CIN: input channels (8,16,32,64); RT: cout row tiles of 16; KIND; CT: 16-voxel column tiles per wave.
// SPLIT (1, 3 or 9): the 27 taps of a stride-1/2 layer are split (by kd, or by (kd,kh)) over SPLIT waves of one block (one tile group per
// block) and the partial sums are reduced through LDS.  The small deep layers (80-640 tiles) are a pure latency
// chain of one L2 round trip per tap: a third of the chain is worth more than the idle lanes it costs.
template <int CIN, int RT, int KIND, int CT, int SPLIT = 1>
__global__ __launch_bounds__(KIND == kConvT2 ? 512 : (SPLIT * 64 > 256 ? SPLIT * 64 : 256)) void k_conv3d(const float* __restrict__ wpk, const float* __restrict__ scale,
                                                const float* __restrict__ shift, const float* __restrict__ in,
                                                const float* __restrict__ residual, float* __restrict__ out,
                                                float* __restrict__ out2, int cout, int relu, int B, int Di, int Hi,
                                                int Wi, int Do, int Ho, int Wo, int rt_total) {
    constexpr int CPL = (CIN >= 16) ? 4 : CIN / 4;
    constexpr int NB = CIN / (4 * CPL);
    constexpr int KS = CIN / 4;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    // a wave owns RT of the layer's rt_total row tiles (output-channel tiles of 16): tiny deep layers are
    // split over more waves this way (conv6 has only 80 voxel tiles but 4 row tiles)
    const int rsplit = rt_total / RT;
    // wave-uniform by construction; readfirstlane tells the compiler, so the tile decomposition below runs on
    // the scalar unit (it was ~1200 VALU of 64-bit divisions per wave, a third of a conv1 wave's time)
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sp = SPLIT > 1 ? wave_in_block : 0;                   // which third of the taps (SPLIT > 1: one tile group per block)
    const int wave_all = SPLIT > 1 ? (int)blockIdx.x : (int)blockIdx.x * (int)(blockDim.x >> 6) + wave_in_block;
    const int wave = wave_all / rsplit;
    const int rt_base = (wave_all - wave * rsplit) * RT;

    // ---- which voxels does this wave own? ----
    // conv: n = B*Do*Ho*Wo output voxels in raster order.  convT: per parity class, n = B*Di*Hi*Wi
    // "q" positions; classes are laid out back to back in units of CT-tile groups.  (n < 2^31: launcher)
    const int n = (KIND == kConvT2) ? B * Di * Hi * Wi : B * Do * Ho * Wo;
    const int tiles = cdiv(n, 16);
    const int groups = cdiv(tiles, CT);
    int cls = 0;
    int grp = wave;
    if (KIND == kConvT2) {
        // the eight parity classes of one group of q-tiles are eight consecutive waves = one 512-thread block:
        // they read the same inputs and write the two halves of the same 64-B output lines from one CU at about
        // the same time (class-major order spread them over different XCDs and moments: half-line writes)
        grp = wave >> 3;
        cls = wave & 7;
        if (grp >= groups) return;
    } else if (wave >= groups) {
        return;
    }
    const int pw = cls & 1, ph = (cls >> 1) & 1, pd = (cls >> 2) & 1;

    int vb[CT], vd[CT], vh[CT], vw[CT];
    bool vok[CT];
    {
        const int dw = (KIND == kConvT2) ? Wi : Wo, dh = (KIND == kConvT2) ? Hi : Ho, dd = (KIND == kConvT2) ? Di : Do;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int v0 = (grp * CT + ct) * 16;             // uniform: scalar div/mod, lanes add j and carry
            int r = v0 / dw;
            const int w0 = v0 - r * dw;
            int q = r / dh;
            const int h0 = r - q * dh;
            const int b0 = q / dd, d0 = q - b0 * dd;
            vok[ct] = v0 + j < n;
            vw[ct] = w0 + j; vh[ct] = h0; vd[ct] = d0; vb[ct] = b0;
            while (vw[ct] >= dw) {
                vw[ct] -= dw;
                if (++vh[ct] == dh) { vh[ct] = 0; if (++vd[ct] == dd) { vd[ct] = 0; ++vb[ct]; } }
            }
        }
    }

    f32x4 acc[CT][RT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[ct][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntd = (KIND == kConvT2) ? 1 + pd : 3, nth = (KIND == kConvT2) ? 1 + ph : 3,
              ntw = (KIND == kConvT2) ? 1 + pw : 3;
    const int ntaps = ntd * nth * ntw;
    const int tap0 = (KIND == kConvT2) ? convt_class_offset(cls) : 0;
    const float* wl = wpk + lane + rt_base * 64;
    constexpr int NA = KS * RT;                       // A operands (weights) per tap

    // Per column tile: the input voxel index of tap (0,0,0) and a 9-bit mask of the per-axis offsets that fall
    // inside the volume.  A tap's address is then base + (uniform tap offset) and its validity one and+compare; loads of
    // padding taps are redirected to a block of zeros, so no per-value select is needed afterwards.
    // (Before: ~35 VALU of 64-bit index math, six compares and eight selects per tile and tap against 8-16
    // MFMAs — these layers were VALU-bound at 5-7x their MFMA time.)
    int vbase[CT];
    unsigned vmask[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        constexpr int s = (KIND == kConvS2) ? 2 : 1;
        const int id0 = (KIND == kConvT2) ? vd[ct] : vd[ct] * s - 1, ih0 = (KIND == kConvT2) ? vh[ct] : vh[ct] * s - 1,
                  iw0 = (KIND == kConvT2) ? vw[ct] : vw[ct] * s - 1;
        vbase[ct] = ((vb[ct] * Di + id0) * Hi + ih0) * Wi + iw0;
        unsigned m = 0;                                  // bit k / 3+k / 6+k: offset k valid along d / h / w
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            m |= ((unsigned)(id0 + k) < (unsigned)Di ? 1u : 0u) << k;
            m |= ((unsigned)(ih0 + k) < (unsigned)Hi ? 1u : 0u) << (3 + k);
            m |= ((unsigned)(iw0 + k) < (unsigned)Wi ? 1u : 0u) << (6 + k);
        }
        vmask[ct] = vok[ct] ? m : 0u;
    }
    const float* zeros = g_conv_zeros + g * CPL;

    // One tap's operands: all loads are UNCONDITIONAL and issued back to back, and taps are double-buffered,
    // so a tap's ~10 L2 round trips overlap the previous tap's MFMAs instead of serialising behind exec-mask
    // branches and per-load s_waitcnt vmcnt(0) (which is what hipcc emits for `ok ? *p : 0`).
    auto issue = [&](int t, float4 (&bq)[CT][NB], float (&aq)[NA]) {
        int od, oh, ow;
        tap_offsets<KIND>(t, ntw, nth, pd, ph, pw, od, oh, ow);
        const int toff = (od * Hi + oh) * Wi + ow;         // uniform
        const unsigned sel = (1u << od) | (8u << oh) | (64u << ow);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const bool ok = (vmask[ct] & sel) == sel;
            const float* p = ok ? in + (long long)(vbase[ct] + toff) * CIN + g * CPL : zeros;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                if (CPL == 4) {
                    bq[ct][cb] = *reinterpret_cast<const float4*>(p + cb * 16);
                } else {
                    const float2 t2 = *reinterpret_cast<const float2*>(p + cb * 8);
                    bq[ct][cb] = make_float4(t2.x, t2.y, 0.f, 0.f);
                }
            }
        }
        const float* wt = wl + (long long)(tap0 + t) * KS * rt_total * 64;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) aq[ks * RT + rt] = wt[(ks * rt_total + rt) * 64];
    };
    auto compute = [&](const float4 (&bq)[CT][NB], const float (&aq)[NA]) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            float bv[CT][4];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                bv[ct][0] = bq[ct][cb].x; bv[ct][1] = bq[ct][cb].y; bv[ct][2] = bq[ct][cb].z; bv[ct][3] = bq[ct][cb].w;
            }
#pragma unroll
            for (int r = 0; r < CPL; ++r)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[(cb * CPL + r) * RT + rt], bv[ct][r],
                                                                           acc[ct][rt], 0, 0, 0);
        }
    };
    {
        // PF-deep operand ring: the loads of taps t+1 .. t+PF-1 are in flight while tap t runs on the matrix core.
        // These layers are a latency chain (a tap = 8-16 MFMAs = 0.1-0.3 us against ~0.7 us per L2 round trip).
        // Measured on MI355X: depth 3-4 is 10-18 % faster for the 27-tap stride-2 layers, but 10-20 % SLOWER for
        // the transposed layers (1-8 taps per parity class: the ring only adds redundant clamped loads) and no
        // better for the small stride-1 layers, which keep depth 2.  Depth is capped by registers.
        constexpr int REGS_PER_TAP = CT * NB * 4 + NA;
        // (Round 5, VERDICT r04 #2 "request all tap operands of a wave up front": built as an operand window of 96 / 160 registers, measured, NOT
        // kept — conv4 11.0 -> 13.4 / 11.8 us, conv3 7.9 -> 8.1, conv5 8.2 -> 8.1 / 7.6, conv6 12.0 -> 11.8 / 11.4: profiles/r05_ab_b4c_conv3d_upfront.txt,
        // tools/patches/r06_pruned_knobs.diff.  Round 6 measured why (profiles/r06_ab_conv3d_wl.txt): these layers are bound by the BYTES their taps
        // pull through L2 -> L1 (~8.5 TB/s), not by the round trips of the ring.)
        {
        constexpr int PF = KIND != kConvS2 ? 2 : (REGS_PER_TAP <= 24 ? 4 : (REGS_PER_TAP <= 40 ? 3 : 2));
        float4 bq[PF][CT][NB];
        float aq[PF][NA];
        const int tbeg = SPLIT > 1 ? sp * (27 / SPLIT) : 0, tend = SPLIT > 1 ? tbeg + 27 / SPLIT : ntaps;
#pragma unroll
        for (int k = 0; k < PF - 1; ++k) issue(tbeg + k < tend ? tbeg + k : tend - 1, bq[k], aq[k]);
#pragma unroll 1
        for (int t = tbeg; t < tend; t += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int tn = t + k + PF - 1;                       // clamped: always loads, never branches
                issue(tn < tend ? tn : tend - 1, bq[(k + PF - 1) % PF], aq[(k + PF - 1) % PF]);
                __builtin_amdgcn_sched_barrier(0);                   // keep the prefetch ahead of this tap's MFMAs
                if (t + k < tend) compute(bq[k], aq[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
    }
    if (SPLIT > 1) {      // reduce the per-kd partial sums into wave 0 (fixed order: deterministic)
        __shared__ float red[(SPLIT > 1 ? SPLIT - 1 : 1) * CT * RT * 4 * 64];
        if (sp > 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(((sp - 1) * CT + ct) * RT + rt) * 256 + r * 64 + lane] = acc[ct][rt][r];
        }
        __syncthreads();
        if (sp > 0) return;
#pragma unroll
        for (int s2 = 0; s2 < SPLIT - 1; ++s2)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ct][rt][r] += red[((s2 * CT + ct) * RT + rt) * 256 + r * 64 + lane];
    }

    // ---- epilogue: BN scale/shift, skip add, ReLU; lane owns channels rt*16+4g..+3 of voxel j ----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        if (!vok[ct]) continue;
        long long o;
        if (KIND == kConvT2)
            o = (((long long)vb[ct] * Do + 2 * vd[ct] + pd) * Ho + 2 * vh[ct] + ph) * Wo + 2 * vw[ct] + pw;
        else
            o = (((long long)vb[ct] * Do + vd[ct]) * Ho + vh[ct]) * Wo + vw[ct];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            int c0 = (rt_base + rt) * 16 + 4 * g;
            if (c0 >= cout) continue;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int c = c0 + r;
                float sc = scale[c], sh = shift[c];      // always valid (pack writes 1/0 without BN)
                y[r] = acc[ct][rt][r] * sc + sh;
            }
            if (out2 != nullptr) {          // fused heads: channels 0..7 -> out (8 ch), channel 8 -> out2
                if (c0 < 8) *reinterpret_cast<float4*>(out + o * 8 + c0) = make_float4(y[0], y[1], y[2], y[3]);
                else if (c0 == 8) out2[o] = y[0];
                continue;
            }
            if (residual != nullptr) {
                float4 rr = *reinterpret_cast<const float4*>(residual + o * cout + c0);
                y[0] += rr.x; y[1] += rr.y; y[2] += rr.z; y[3] += rr.w;
            }
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = relu1(y[r]);
            }
            *reinterpret_cast<float4*>(out + o * cout + c0) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

template <int CIN, int RT, int KIND, int CT, int SPLIT = 1>
static void launch_one(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B,
                       int Di, int Hi, int Wi, hipStream_t st) {
    int Do, Ho, Wo;
    if (KIND == kConvS1) { Do = Di; Ho = Hi; Wo = Wi; }
    else if (KIND == kConvS2) { Do = (Di - 1) / 2 + 1; Ho = (Hi - 1) / 2 + 1; Wo = (Wi - 1) / 2 + 1; }
    else { Do = 2 * Di; Ho = 2 * Hi; Wo = 2 * Wi; }
    const int rt_total = cdiv(L.cout, 16);
    long long n = (KIND == kConvT2) ? (long long)B * Di * Hi * Wi : (long long)B * Do * Ho * Wo;
    long long groups = cdivl(cdivl(n, 16), CT);
    long long waves = ((KIND == kConvT2) ? groups * 8 : groups) * (rt_total / RT);
    constexpr int WPB = SPLIT > 1 ? 1 : ((KIND == kConvT2) ? 8 : 4);    // tile groups ("logical waves") per block
    unsigned grid = (unsigned)cdivl(waves, WPB);
    ENERF_LAUNCH((k_conv3d<CIN, RT, KIND, CT, SPLIT>), grid, WPB * SPLIT * 64, 0, st, L.w, L.scale, L.shift, in, residual, out, out2, L.cout,
                 L.relu, B, Di, Hi, Wi, Do, Ho, Wo, rt_total);
}

// Pick (row tiles per wave, column tiles per wave): favour operand reuse when the layer has plenty of
// voxel tiles, favour wave count when it does not (the deep levels have 80..2000 tiles for 1024 SIMDs).
template <int CIN, int KIND>
static bool dispatch_rt(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B,
                        int Di, int Hi, int Wi, int small_variant, hipStream_t st) {
    const int rt_total = cdiv(L.cout, 16);
    long long n = (KIND == kConvS1) ? (long long)B * Di * Hi * Wi
                  : (KIND == kConvS2) ? (long long)B * ((Di - 1) / 2 + 1) * ((Hi - 1) / 2 + 1) * ((Wi - 1) / 2 + 1)
                                      : 8LL * B * Di * Hi * Wi;
    const long long tiles = cdivl(n, 16);
    const int ct_default = rt_total == 1 ? 4 : (rt_total == 2 ? 2 : 1);
    const bool small = cdivl(tiles, ct_default) < 1024;      // fewer waves than SIMDs: split finer
    if (small && KIND != kConvT2 && CIN >= 16 && (small_variant == 2 || small_variant == 3)) {
        // A/B (enerf_options_t.conv3d_small_variant): operand reuse instead of wave count for the deep layers
        if (small_variant == 2 && rt_total >= 2) { launch_one<CIN, 2, KIND, 2>(L, in, residual, out, out2, B, Di, Hi, Wi, st); return true; }
        launch_one<CIN, 1, KIND, 4>(L, in, residual, out, out2, B, Di, Hi, Wi, st);
        return true;
    }
    if (small) {
        // taps split by kd over three waves + LDS reduction (a 9-way split and no split both measured slower)
        if (KIND != kConvT2 && CIN >= 16) launch_one<CIN, 1, KIND, 1, (KIND != kConvT2 && CIN >= 16 ? 3 : 1)>(L, in, residual, out, out2, B, Di, Hi, Wi, st);
        else launch_one<CIN, 1, KIND, 1>(L, in, residual, out, out2, B, Di, Hi, Wi, st);
        return true;
    }
    switch (rt_total) {
        case 1: launch_one<CIN, 1, KIND, 4>(L, in, residual, out, out2, B, Di, Hi, Wi, st); return true;
        case 2: launch_one<CIN, 2, KIND, 2>(L, in, residual, out, out2, B, Di, Hi, Wi, st); return true;
        case 4: launch_one<CIN, 4, KIND, 1>(L, in, residual, out, out2, B, Di, Hi, Wi, st); return true;
        default: return false;
    }
}
template <int KIND>
static bool dispatch_cin(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B,
                         int Di, int Hi, int Wi, int sv, hipStream_t st) {
    switch (L.cin) {
        case 8: return dispatch_rt<8, KIND>(L, in, residual, out, out2, B, Di, Hi, Wi, sv, st);
        case 16: return dispatch_rt<16, KIND>(L, in, residual, out, out2, B, Di, Hi, Wi, sv, st);
        case 32: return dispatch_rt<32, KIND>(L, in, residual, out, out2, B, Di, Hi, Wi, sv, st);
        case 64: return dispatch_rt<64, KIND>(L, in, residual, out, out2, B, Di, Hi, Wi, sv, st);
        default: return false;
    }
}

// =====================================================================================================
// V2: stride-1 convolution with the haloed input box staged in LDS.
// A block (4 waves) owns a BD x 8 x 16 box of output voxels = BD*8 MFMA column tiles (16 consecutive x
// each); the (BD+2) x 10 x 18 input box of one 16-channel block (8 for Cin=8) is copied once into LDS
// (zero-filled outside the volume = the conv's zero padding) and every tap's B operand is a single
// ds_read_b128 (ds_read_b64 for Cin=8) at a constant offset from the lane's voxel.  That removes the 27x
// re-read of activations through the TA/L1 path that bounds V1; A operands (weights) still stream from
// L1/L2 as 256-B coalesced loads, one per CTW MFMAs.
// =====================================================================================================
template <int CIN, int RT, int BD, int BH = 8>
__global__ __launch_bounds__(256, (BD == 4 ? 2 : 3)) void k_conv3d_s1_lds(   // = co-resident blocks/CU the LDS box allows
const float* __restrict__ wpk, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ in,
                                                       float* __restrict__ out, float* __restrict__ out2, int cout,
                                                       int relu, int B, int D, int H, int W, int nbd, int nbh, int nbw) {
    constexpr int BW = 16;                              // box = BD x BH x 16 outputs (BH = 8, or 4 for mid-size layers)
    // (Two identical blocks share a CU and stay in lockstep — both stage, both compute, both store at the same time;
    // a phase ablation gave 70 us MFMA + 23 us rest = 93 us.  Staggering the second block of each CU by one MFMA
    // phase, and s_setprio tickets, were measured and changed nothing: not kept.)
    constexpr int CB = CIN >= 16 ? 16 : CIN;          // channels staged per pass
    constexpr int CPL = CB / 4;                         // channels per lane per LDS read
    constexpr int NCB = CIN / CB;
    constexpr int KS = CIN / 4;
    constexpr int CTW = BD * BH / 4;                    // column tiles per wave
    constexpr int HX = BW + 2, HY = BH + 2, HZ = BD + 2;
    constexpr int NVOX = HZ * HY * HX;
    constexpr int QV = CB / 4;
    ENERF_DYN_SMEM(float, lds);

    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    // XCD-aware block order: consecutive block ids land on different XCDs (private L2s), so give each
    // XCD a contiguous run of boxes; neighbouring boxes (shared halos) then hit the same L2.
    const int bid = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    int t = bid;
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh; t /= nbh;
    const int bd = t % nbd;
    const int b = t / nbd;
    const int x0 = bw * BW, y0 = bh * BH, z0 = bd * BD;

    f32x4 acc[CTW][RT];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* inb = in + (long long)b * D * H * W * CIN;
    const float* wl = wpk + lane;

    constexpr int NIT = (NVOX * QV + 255) / 256;     // float4 staging loads per thread
    constexpr int NAT = CPL * RT;                     // A operands (weights) of one tap
#pragma unroll 1
    for (int cb = 0; cb < NCB; ++cb) {
        // Operand pipeline (all 27 taps unrolled, order pinned with sched_barrier):
        //   weights (A): 3-deep register ring, tap t+2 requested from L1/L2 while tap t runs on the matrix core
        //                (one tap = CTW*CPL*RT MFMAs >= 512 cycles, two taps cover the L2 latency);
        //   activations (B): 2-deep ring of LDS reads, tap t+1 read while tap t runs.
        // Left to itself hipcc sinks every weight load to just before its MFMA (s_waitcnt vmcnt(0) per load)
        // or, fully unrolled without fences, hoists ~250 VGPRs of LDS reads and spills.
        auto issue_a = [&](int tap, float (&aq)[NAT]) {
            const float* wt = wl + ((long long)tap * KS + cb * CPL) * RT * 64;
#pragma unroll
            for (int r = 0; r < CPL; ++r)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) aq[r * RT + rt] = wt[(r * RT + rt) * 64];
        };
        float aq[3][NAT];
        issue_a(0, aq[0]);
        issue_a(1, aq[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (cb > 0) __syncthreads();                   // previous pass finished reading LDS
        {   // stage the haloed box: loads issued back to back and unconditionally (clamped address,
            // zero-select afterwards) so they overlap instead of serialising behind exec-mask branches
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

        const float* lbase[CTW];                        // this lane's voxel (tap 0,0,0) in each of its column tiles
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const int tile = wv * CTW + c, td = tile / BH, th = tile - td * BH;
            lbase[c] = lds + ((td * HY + th) * HX + j) * CB + g * CPL;
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
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 2 < 27) issue_a(tap + 2, aq[(tap + 2) % 3]);
            if (tap + 1 < 27) read_b(tap + 1, bq[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < CPL; ++r)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int c = 0; c < CTW; ++c)
                        acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[tap % 3][r * RT + rt], bq[tap & 1][c][r],
                                                                          acc[c][rt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue (same as V1): BN scale/shift, ReLU, float4 store; fused heads go to out/out2 ----
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int tile = wv * CTW + c, td = tile / BH, th = tile - td * BH;
        const int z = z0 + td, y = y0 + th, x = x0 + j;
        if (z >= D || y >= H || x >= W) continue;
        const long long o = (((long long)b * D + z) * H + y) * W + x;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int c0 = rt * 16 + 4 * g;
            if (c0 >= cout) continue;
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = c0 + r;
                const float sc = scale[ch], sh = shift[ch];
                yv[r] = acc[c][rt][r] * sc + sh;
            }
            if (out2 != nullptr) {
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

template <int CIN, int RT, int BD, int BH = 8>
static void launch_s1_lds(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W,
                          hipStream_t st) {
    constexpr int CB = CIN >= 16 ? 16 : CIN;
    const int nbd = cdiv(D, BD), nbh = cdiv(H, BH), nbw = cdiv(W, 16);
    const size_t shmem = (size_t)(BD + 2) * (BH + 2) * 18 * CB * sizeof(float);
    const unsigned grid = (unsigned)((long long)B * nbd * nbh * nbw);
    ENERF_LAUNCH((k_conv3d_s1_lds<CIN, RT, BD, BH>), grid, 256, shmem, st, L.w, L.scale, L.shift, in, out, out2, L.cout,
                 L.relu, B, D, H, W, nbd, nbh, nbw);
}
template <int CIN>
static bool dispatch_s1_lds(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W,
                            hipStream_t st) {
    const int rt_total = cdiv(L.cout, 16);
    if (rt_total == 1) {
        // box depth 4 (2 blocks/CU, fewer halo reads) for layers that fill the chip, depth 2 for the mid-size ones
        // (level-1 conv2: 160 boxes of depth 4 leave 96 CUs idle).
        const long long boxes4 = (long long)B * cdiv(D, 4) * cdiv(H, 8) * cdiv(W, 16);
        const int bd = boxes4 >= 256 ? 4 : 2;   // measured: L0 conv2 20 -> 12.5 us; 480-box layers stay at 4
        const int bh = boxes4 >= 256 ? 8 : 4;   // mid-size layers: 2 x 4 x 16 boxes, 4x the blocks
        if (D % 4 == 0 && bd == 4) launch_s1_lds<CIN, 1, 4>(L, in, out, out2, B, D, H, W, st);
        else if (bh == 4) launch_s1_lds<CIN, 1, 2, 4>(L, in, out, out2, B, D, H, W, st);
        else launch_s1_lds<CIN, 1, 2>(L, in, out, out2, B, D, H, W, st);
        return true;
    }
    if (rt_total == 2) { launch_s1_lds<CIN, 2, 2>(L, in, out, out2, B, D, H, W, st); return true; }
    return false;
}
// Kernel selection.  `o` carries the caller's explicit choices (enerf_options_t; nothing is read from the environment).
// Returns false for a layer shape no kernel handles (the C entry reports ENERF_EINVAL).
bool launch_conv3d(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B, int Di,
                   int Hi, int Wi, const Options& o, hipStream_t st) {
    const bool lds_ok = !o.conv3d_global_only;
    const long long min_vox = o.conv3d_lds_min_voxels > 0 ? o.conv3d_lds_min_voxels : 16384;
    const long long vox_in = (long long)B * Di * Hi * Wi;
    // every-class transposed kernels (conv3d_t2.hip): conv11 (16 -> 8, x-parity-paired MFMA rows) and conv9 (32 -> 16)
    if (L.kind == kConvT2 && out2 == nullptr && lds_ok && o.conv3d_t2_variant != 1) {
        // measured (profiles/r03_conv3d_layers.txt): conv11 16.3 vs 24.0 us (level 1), 10.3 vs 15.2 (level 0); conv9 10.7 vs
        // 17.2 (level 1) but 10.1 vs 9.1 at level 0's 3840 positions (240 q-tiles: too few blocks)
        const bool big = L.cout == 8 ? vox_in >= min_vox : vox_in >= min_vox / 2;
        if ((big || o.conv3d_t2_variant >= 2) && launch_conv3d_t2_all(L, in, residual, out, B, Di, Hi, Wi, st)) return true;
    }
    if (L.out_planar) return false;                    // only the kernel above writes channel-quad planes
    // round-2 LDS-staged transposed path (conv11, 16 -> 8, one class per MFMA): kept for A/B (conv3d_t2_variant = 1)
    if (L.kind == kConvT2 && out2 == nullptr && lds_ok && 8 * vox_in >= 32 * min_vox &&
        launch_conv3d_t2_lds(L, in, residual, out, B, Di, Hi, Wi, st))
        return true;
    // LDS-staged stride-2 path (conv1 of both nets).  Level-1 conv1: 19.7 -> 14.1 us; level 0 is no faster
    if (L.kind == kConvS2 && residual == nullptr && out2 == nullptr && lds_ok && vox_in >= 32 * min_vox &&
        launch_conv3d_s2_lds(L, in, out, B, Di, Hi, Wi, st))
        return true;
    // batched-4x4 path for the Cout=8(+1) stride-1 layers (conv0 of both levels, fused heads): no wasted MFMA rows
    if (o.conv3d_b4 != 1 && residual == nullptr && lds_ok && vox_in >= min_vox && launch_conv3d_b4(L, in, out, out2, B, Di, Hi, Wi, o.conv3d_b4 != 3, st))
        return true;
    if (L.in_planar) return false;                     // only the asynchronously staged b4 kernel reads channel-quad planes
    // tap-packed path for the Cout=8 stride-1 layers (conv0 of both levels, fused heads): 2/3 of the MFMAs
    if (o.conv3d_pk8 != 1 && residual == nullptr && lds_ok && vox_in >= min_vox &&
        launch_conv3d_pk8(L, in, out, out2, B, Di, Hi, Wi, o.conv3d_pk8 == 2, st))
        return true;
    // LDS-staged path: stride-1 layers with enough voxels to fill the chip and cout <= 32
    if (L.kind == kConvS1 && residual == nullptr && lds_ok && L.cout <= 32 && vox_in >= min_vox) {
        bool ok = false;
        switch (L.cin) {
            case 8: ok = dispatch_s1_lds<8>(L, in, out, out2, B, Di, Hi, Wi, st); break;
            case 16: ok = dispatch_s1_lds<16>(L, in, out, out2, B, Di, Hi, Wi, st); break;
            case 32: ok = dispatch_s1_lds<32>(L, in, out, out2, B, Di, Hi, Wi, st); break;
            default: break;
        }
        if (ok) return true;
    }
    // small deep stride-1 / stride-2 layers whose input is one or two planes thick (level 1's conv4 .. conv6 at 8 depth planes): whole kd
    // taps are padding there, and conv3d_wl.hip skips them per block — copy, loads and MFMAs (conv6 12.0 -> 7.5 us, conv5 8.3 -> 7.3,
    // conv4 15.1 -> 14.3; on thicker volumes the tap-split kernel below is as fast or faster: profiles/r06_ab_conv3d_wl.txt).
    // conv3d_small_variant: 0 = this routing, 4 = conv3d_wl for every small layer (A/B), 1 .. 3 = the round-2 .. 5 forms only
    if ((L.kind == kConvS1 || L.kind == kConvS2) && residual == nullptr && out2 == nullptr && lds_ok && L.cin >= 16 &&
        L.cout % 16 == 0 && (o.conv3d_small_variant == 4 || (o.conv3d_small_variant == 0 && Di <= 2))) {
        const int rt_total = cdiv(L.cout, 16), ct_default = rt_total == 1 ? 4 : (rt_total == 2 ? 2 : 1);
        const long long n = L.kind == kConvS1 ? vox_in : (long long)B * ((Di - 1) / 2 + 1) * ((Hi - 1) / 2 + 1) * ((Wi - 1) / 2 + 1);
        if (cdivl(cdivl(n, 16), ct_default) < 1024 && launch_conv3d_wl(L, in, out, B, Di, Hi, Wi, st)) return true;
    }
    switch (L.kind) {
        case kConvS1: return dispatch_cin<kConvS1>(L, in, residual, out, out2, B, Di, Hi, Wi, o.conv3d_small_variant, st);
        case kConvS2: return dispatch_cin<kConvS2>(L, in, residual, out, out2, B, Di, Hi, Wi, o.conv3d_small_variant, st);
        case kConvT2: return dispatch_cin<kConvT2>(L, in, residual, out, out2, B, Di, Hi, Wi, o.conv3d_small_variant, st);
        default: return false;
    }
}

// The layout decisions enerf_cost_reg / enerf_forward make BEFORE the launches must mirror the routing above.
bool conv3d_routes_b4_glds(const Options& o, long long vox, int D) {
    const long long min_vox = o.conv3d_lds_min_voxels > 0 ? o.conv3d_lds_min_voxels : 16384;
    return o.conv3d_b4 != 1 && o.conv3d_b4 != 3 && !o.conv3d_global_only && vox >= min_vox && D % 4 == 0;
}
bool conv3d_routes_t2_pair(const Options& o, long long vox_in) {
    const long long min_vox = o.conv3d_lds_min_voxels > 0 ? o.conv3d_lds_min_voxels : 16384;
    return !o.conv3d_global_only && o.conv3d_t2_variant != 1 && (vox_in >= min_vox || o.conv3d_t2_variant >= 2);
}
bool cost_reg_wants_planar_volume(const enerf_options_t& o, int in_channels, int B, int D, int h, int w) {
    (void)in_channels;
    return conv3d_routes_b4_glds(o, (long long)B * D * h * w, D);
}

}  // namespace enerf
