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

// ---- the implicit-GEMM kernel ---------------------------------------------------------------------
template <int CPL>
__device__ __forceinline__ void load_b(const float* __restrict__ p, bool ok, float (&v)[4]) {
    if (CPL == 4) {
        float4 t = ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        float2 t = ok ? *reinterpret_cast<const float2*>(p) : make_float2(0.f, 0.f);
        v[0] = t.x; v[1] = t.y; v[2] = 0.f; v[3] = 0.f;
    }
}

// CIN: input channels (8,16,32,64); RT: cout row tiles of 16; KIND; CT: 16-voxel column tiles per wave.
template <int CIN, int RT, int KIND, int CT>
__global__ __launch_bounds__(256) void k_conv3d(const float* __restrict__ wpk, const float* __restrict__ scale,
                                                const float* __restrict__ shift, const float* __restrict__ in,
                                                const float* __restrict__ residual, float* __restrict__ out,
                                                float* __restrict__ out2, int cout, int relu, int B, int Di, int Hi,
                                                int Wi, int Do, int Ho, int Wo) {
    constexpr int CPL = (CIN >= 16) ? 4 : CIN / 4;
    constexpr int NB = CIN / (4 * CPL);
    constexpr int KS = CIN / 4;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;

    // ---- which voxels does this wave own? ----
    // conv: n = B*Do*Ho*Wo output voxels in raster order.  convT: per parity class, n = B*Di*Hi*Wi
    // "q" positions; classes are laid out back to back in units of CT-tile groups.
    const long long n = (KIND == kConvT2) ? (long long)B * Di * Hi * Wi : (long long)B * Do * Ho * Wo;
    const long long tiles = cdivl(n, 16);
    const long long groups = cdivl(tiles, CT);
    int cls = 0;
    long long grp = wave;
    if (KIND == kConvT2) {
        cls = (int)(wave / groups);
        grp = wave - (long long)cls * groups;
        if (cls >= 8) return;
    } else if (wave >= groups) {
        return;
    }
    const int pw = cls & 1, ph = (cls >> 1) & 1, pd = (cls >> 2) & 1;

    int vb[CT], vd[CT], vh[CT], vw[CT];
    bool vok[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        long long v = (grp * CT + ct) * 16 + j;
        vok[ct] = v < n;
        long long vv = vok[ct] ? v : 0;
        int dw = (KIND == kConvT2) ? Wi : Wo, dh = (KIND == kConvT2) ? Hi : Ho, dd = (KIND == kConvT2) ? Di : Do;
        vw[ct] = (int)(vv % dw); vv /= dw;
        vh[ct] = (int)(vv % dh); vv /= dh;
        vd[ct] = (int)(vv % dd);
        vb[ct] = (int)(vv / dd);
    }

    f32x4 acc[CT][RT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[ct][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntd = (KIND == kConvT2) ? 1 + pd : 3, nth = (KIND == kConvT2) ? 1 + ph : 3,
              ntw = (KIND == kConvT2) ? 1 + pw : 3;
    int tap = (KIND == kConvT2) ? convt_class_offset(cls) : 0;
    const float* wl = wpk + lane;
    for (int td = 0; td < ntd; ++td)
        for (int th = 0; th < nth; ++th)
            for (int tw = 0; tw < ntw; ++tw, ++tap) {
                const float* pin[CT];
                bool ok[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    int id, ih, iw;
                    if (KIND == kConvT2) {
                        int k, dq;
                        convt_axis(pd, td, k, dq); id = vd[ct] + dq;
                        convt_axis(ph, th, k, dq); ih = vh[ct] + dq;
                        convt_axis(pw, tw, k, dq); iw = vw[ct] + dq;
                    } else {
                        constexpr int s = (KIND == kConvS2) ? 2 : 1;
                        id = vd[ct] * s - 1 + td; ih = vh[ct] * s - 1 + th; iw = vw[ct] * s - 1 + tw;
                    }
                    ok[ct] = vok[ct] && id >= 0 && id < Di && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi;
                    long long off = ok[ct] ? ((((long long)vb[ct] * Di + id) * Hi + ih) * Wi + iw) : 0;
                    pin[ct] = in + off * CIN + g * CPL;
                }
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    float bv[CT][4];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) load_b<CPL>(pin[ct] + cb * 4 * CPL, ok[ct], bv[ct]);
#pragma unroll
                    for (int r = 0; r < CPL; ++r) {
                        const int ks = cb * CPL + r;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            float a = wl[(((long long)tap * KS + ks) * RT + rt) * 64];
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct)
                                acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[ct][r], acc[ct][rt], 0, 0, 0);
                        }
                    }
                }
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
            int c0 = rt * 16 + 4 * g;
            if (c0 >= cout) continue;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int c = c0 + r;
                float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
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
                for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
            }
            *reinterpret_cast<float4*>(out + o * cout + c0) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
}

template <int CIN, int RT, int KIND, int CT>
static void launch_one(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B,
                       int Di, int Hi, int Wi, hipStream_t st) {
    int Do, Ho, Wo;
    if (KIND == kConvS1) { Do = Di; Ho = Hi; Wo = Wi; }
    else if (KIND == kConvS2) { Do = (Di - 1) / 2 + 1; Ho = (Hi - 1) / 2 + 1; Wo = (Wi - 1) / 2 + 1; }
    else { Do = 2 * Di; Ho = 2 * Hi; Wo = 2 * Wi; }
    long long n = (KIND == kConvT2) ? (long long)B * Di * Hi * Wi : (long long)B * Do * Ho * Wo;
    long long groups = cdivl(cdivl(n, 16), CT);
    long long waves = (KIND == kConvT2) ? groups * 8 : groups;
    unsigned grid = (unsigned)cdivl(waves, 4);
    ENERF_LAUNCH((k_conv3d<CIN, RT, KIND, CT>), grid, 256, 0, st, L.w, L.scale, L.shift, in, residual, out, out2, L.cout,
                 L.relu, B, Di, Hi, Wi, Do, Ho, Wo);
}

template <int CIN, int KIND>
static bool dispatch_rt(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B,
                        int Di, int Hi, int Wi, hipStream_t st) {
    switch (cdiv(L.cout, 16)) {
        case 1: launch_one<CIN, 1, KIND, 4>(L, in, residual, out, out2, B, Di, Hi, Wi, st); return true;
        case 2: launch_one<CIN, 2, KIND, 2>(L, in, residual, out, out2, B, Di, Hi, Wi, st); return true;
        case 4: launch_one<CIN, 4, KIND, 1>(L, in, residual, out, out2, B, Di, Hi, Wi, st); return true;
        default: return false;
    }
}
template <int KIND>
static bool dispatch_cin(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B,
                         int Di, int Hi, int Wi, hipStream_t st) {
    switch (L.cin) {
        case 8: return dispatch_rt<8, KIND>(L, in, residual, out, out2, B, Di, Hi, Wi, st);
        case 16: return dispatch_rt<16, KIND>(L, in, residual, out, out2, B, Di, Hi, Wi, st);
        case 32: return dispatch_rt<32, KIND>(L, in, residual, out, out2, B, Di, Hi, Wi, st);
        case 64: return dispatch_rt<64, KIND>(L, in, residual, out, out2, B, Di, Hi, Wi, st);
        default: return false;
    }
}
void launch_conv3d(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B, int Di,
                   int Hi, int Wi, hipStream_t st) {
    switch (L.kind) {
        case kConvS1: dispatch_cin<kConvS1>(L, in, residual, out, out2, B, Di, Hi, Wi, st); break;
        case kConvS2: dispatch_cin<kConvS2>(L, in, residual, out, out2, B, Di, Hi, Wi, st); break;
        case kConvT2: dispatch_cin<kConvT2>(L, in, residual, out, out2, B, Di, Hi, Wi, st); break;
        default: break;
    }
}

}  // namespace enerf
