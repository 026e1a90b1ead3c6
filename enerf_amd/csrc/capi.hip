// capi.hip — the extern "C" boundary (include/enerf_hip.h): argument validation, the cost-regularisation
// network driver, error reporting.  No torch types; raw device pointers + sizes + a hipStream_t.
#include <stdarg.h>
#include <stdio.h>

#include "kernels.h"

using namespace enerf;

namespace enerf {
__global__ __launch_bounds__(256) void k_zero(unsigned* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0u;
}
void zero_async(void* p, size_t bytes, hipStream_t st) {
    const long long n = (long long)(bytes / 4);
    if (n <= 0) return;
    long long blocks = cdivl(n, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    ENERF_LAUNCH_SIMPLE(k_zero, (unsigned)blocks, 256, 0, st, (unsigned*)p, n);
}
// two accumulators in one launch when the caller allocated them back to back (the Python binding does)
void zero_async2(void* p, size_t bytes_p, void* q, size_t bytes_q, hipStream_t st) {
    if ((char*)p + bytes_p == (char*)q && bytes_p % 4 == 0) { zero_async(p, bytes_p + bytes_q, st); return; }
    zero_async(p, bytes_p, st);
    zero_async(q, bytes_q, st);
}
int device_cu_count() {
#ifdef ENERF_EMU
    return 256;
#else
    static int cached[64];                         // per device ordinal; 0 = not queried yet (benign race: same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
#endif
}
}  // namespace enerf

namespace enerf {
static thread_local char g_err[512] = "";
const char* last_error() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ENERF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ENERF_OK;
}
}  // namespace enerf
namespace {

// ---- cost-reg layer table ------------------------------------------------------------------------
struct LayerSpec { int idx, cin, cout, kind, relu, bn; };
// returns number of layers; the fused heads (8 -> 8+1) are appended last with idx = -1
int costreg_layers(int cin0, int full, LayerSpec* L) {
    int n = 0;
    L[n++] = {0, cin0, 8, kConvS1, 1, 1};
    L[n++] = {1, 8, 16, kConvS2, 1, 1};
    L[n++] = {2, 16, 16, kConvS1, 1, 1};
    L[n++] = {3, 16, 32, kConvS2, 1, 1};
    L[n++] = {4, 32, 32, kConvS1, 1, 1};
    if (full) {
        L[n++] = {5, 32, 64, kConvS2, 1, 1};
        L[n++] = {6, 64, 64, kConvS1, 1, 1};
        L[n++] = {7, 64, 32, kConvT2, 0, 1};
    }
    L[n++] = {9, 32, 16, kConvT2, 0, 1};
    L[n++] = {11, 16, 8, kConvT2, 0, 1};
    L[n++] = {-1, 8, 9, kConvS1, 0, 0};
    return n;
}
bool layer_pk8(const LayerSpec& s) { return s.kind == kConvS1 && (s.cout == 8 || s.idx < 0); }   // conv0, fused heads
bool layer_t2pair(const LayerSpec& s) { return s.kind == kConvT2 && s.cin == 16 && s.cout == 8; }   // conv11
long long layer_floats(const LayerSpec& s) {
    return conv3d_packed_floats(s.cin, s.cout, s.kind) + 2 * cdiv(s.cout, 16) * 16 +
           (layer_pk8(s) ? conv3d_pk8_packed_floats(s.cin) + conv3d_b4_packed_floats(s.cin) : 0) +
           (layer_t2pair(s) ? conv3d_t2_pair_floats() : 0);
}
}  // namespace

extern "C" {

int enerf_abi_version(void) { return ENERF_ABI_VERSION; }
const char* enerf_last_error(void) { return last_error(); }

int enerf_channels_last(const float* src, float* dst, int n, int C, long long P, int Cpad, enerf_stream_t stream) {
    REQUIRE(src && dst && n > 0 && C > 0 && P > 0 && Cpad >= C, "channels_last: bad arguments");
    launch_channels_last(src, dst, n, C, P, Cpad, (hipStream_t)stream);
    return check_launch("channels_last");
}
int enerf_channels_first(const float* src, float* dst, int n, int C, long long P, int Cpad, enerf_stream_t stream) {
    REQUIRE(src && dst && n > 0 && C > 0 && P > 0 && Cpad >= C, "channels_first: bad arguments");
    launch_channels_first(src, dst, n, C, P, Cpad, (hipStream_t)stream);
    return check_launch("channels_first");
}
int enerf_pack_img_feat_rgb(const float* im_feat, int C, int Hf, int Wf, const float* src_inps, int H, int W, int Hr,
                            int Wr, int tex, int n_img, float* out, enerf_stream_t stream) {
    REQUIRE(im_feat && src_inps && out, "pack_img_feat_rgb: null pointer");
    REQUIRE(C > 0 && tex >= C + 3 && tex % 4 == 0 && n_img > 0 && Hr > 0 && Wr > 0, "pack_img_feat_rgb: bad shape");
    launch_pack_img_feat_rgb(im_feat, C, Hf, Wf, src_inps, H, W, Hr, Wr, tex, n_img, out, (hipStream_t)stream);
    return check_launch("pack_img_feat_rgb");
}
int enerf_get_proj_mats(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int B,
                        int S, float src_scale, float tar_scale, float* proj, enerf_stream_t stream) {
    REQUIRE(src_ixts && src_exts && tar_ixt && tar_ext && proj && B > 0 && S > 0, "get_proj_mats: bad arguments");
    launch_proj_mats(src_ixts, src_exts, tar_ixt, tar_ext, B, S, src_scale, tar_scale, proj, (hipStream_t)stream);
    return check_launch("get_proj_mats");
}
int enerf_get_depth_values(const float* near_far, const float* prev_depth, const float* prev_std,
                           const float* prev_near_far, int B, int D, int h, int w, int hp, int wp, int depth_inv,
                           float* depth_values, float* near_far_out, enerf_stream_t stream) {
    REQUIRE(depth_values && near_far_out && B > 0 && D > 0 && h > 0 && w > 0, "get_depth_values: bad arguments");
    if (prev_depth) REQUIRE(prev_std && prev_near_far && hp > 0 && wp > 0, "get_depth_values: incomplete previous level");
    else REQUIRE(near_far, "get_depth_values: near_far required at level 0");
    launch_depth_values(near_far, prev_depth, prev_std, prev_near_far, B, D, h, w, hp, wp, depth_inv, depth_values,
                        near_far_out, (hipStream_t)stream);
    return check_launch("get_depth_values");
}
int enerf_level_prep(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int B,
                     int S, float src_scale, float tar_scale, float* proj, const float* near_far,
                     const float* prev_depth, const float* prev_std, const float* prev_near_far, int D, int h, int w,
                     int hp, int wp, int depth_inv, float* depth_values, float* near_far_out, enerf_stream_t stream) {
    REQUIRE(src_ixts && src_exts && tar_ixt && tar_ext && proj && B > 0 && S > 0, "level_prep: bad projection arguments");
    REQUIRE(depth_values && near_far_out && D > 0 && h > 0 && w > 0, "level_prep: bad depth arguments");
    if (prev_depth) REQUIRE(prev_std && prev_near_far && hp > 0 && wp > 0, "level_prep: incomplete previous level");
    else REQUIRE(near_far, "level_prep: near_far required at level 0");
    launch_level_prep(src_ixts, src_exts, tar_ixt, tar_ext, S, src_scale, tar_scale, proj, near_far, prev_depth, prev_std,
                      prev_near_far, B, D, h, w, hp, wp, depth_inv, depth_values, near_far_out, (hipStream_t)stream);
    return check_launch("level_prep");
}
int enerf_build_feature_volume(const float* feat, const float* proj, const float* depth_values, int B, int S, int C,
                               int Hs, int Ws, int D, int h, int w, float* vol, enerf_stream_t stream) {
    REQUIRE(feat && proj && depth_values && vol, "build_feature_volume: null pointer");
    REQUIRE(C == 8 || C == 16 || C == 32, "build_feature_volume: C=%d unsupported (8/16/32)", C);
    REQUIRE(B > 0 && S > 0 && Hs > 1 && Ws > 1 && D > 0 && h > 0 && w > 0, "build_feature_volume: bad shape");
    REQUIRE((long long)B * S * Hs * Ws * C < (1LL << 32) && (long long)Hs * Ws < (1LL << 23),
            "build_feature_volume: source features too large for 32-bit gather offsets");
    REQUIRE((long long)B * D * h * w * (C / 4) < (1LL << 31) && (long long)h * w < (1LL << 23),
            "build_feature_volume: volume too large for 32-bit voxel indices");
    // the launch carries (b, d) in gridDim.z and forms the voxel index with 24-bit multiplies (volume.hip, round 4)
    REQUIRE((long long)B * D <= 65535 && (long long)B * D * h < (1LL << 23) && w < (1 << 23),
            "build_feature_volume: B*D=%lld planes / B*D*h=%lld rows beyond the grid-carried voxel decomposition (65535 / 2^23)",
            (long long)B * D, (long long)B * D * h);
    launch_feature_volume(feat, proj, depth_values, B, S, C, Hs, Ws, D, h, w, vol, (hipStream_t)stream);
    return check_launch("build_feature_volume");
}

long long enerf_cost_reg_packed_floats(int in_channels, int full) {
    LayerSpec L[12];
    int n = costreg_layers(in_channels, full, L);
    long long t = 0;
    for (int i = 0; i < n; ++i) t += layer_floats(L[i]);
    return t;
}
int enerf_cost_reg_pack(const enerf_costreg_raw_t* raw, float* packed, enerf_stream_t stream) {
    REQUIRE(raw && packed, "cost_reg_pack: null pointer");
    REQUIRE(raw->in_channels == 8 || raw->in_channels == 16 || raw->in_channels == 32, "cost_reg_pack: in_channels");
    LayerSpec L[12];
    int n = costreg_layers(raw->in_channels, raw->full, L);
    float* p = packed;
    for (int i = 0; i < n; ++i) {
        const LayerSpec& s = L[i];
        long long wf = conv3d_packed_floats(s.cin, s.cout, s.kind);
        int cp = cdiv(s.cout, 16) * 16;
        if (s.idx >= 0) {
            const enerf_conv_bn_t& c = raw->conv[s.idx];
            REQUIRE(c.w && c.bn_weight && c.bn_bias && c.bn_mean && c.bn_var, "cost_reg_pack: conv%d missing", s.idx);
            launch_conv3d_pack(c.w, nullptr, s.cout, c.bn_weight, c.bn_bias, c.bn_mean, c.bn_var, 1e-5f, s.cin, s.cout,
                               s.kind, p, p + wf, p + wf + cp, (hipStream_t)stream);
            if (layer_t2pair(s)) launch_conv3d_t2_pair_pack(p, p + wf + 2 * cp, (hipStream_t)stream);   // from the image just packed
            if (layer_pk8(s)) {
                launch_conv3d_pk8_pack(c.w, nullptr, s.cin, p + wf + 2 * cp, (hipStream_t)stream);
                launch_conv3d_b4_pack(c.w, nullptr, s.cin, p + wf + 2 * cp + conv3d_pk8_packed_floats(s.cin), (hipStream_t)stream);
            }
        } else {
            REQUIRE(raw->feat_conv_w && raw->depth_conv_w, "cost_reg_pack: heads missing");
            launch_conv3d_pack(raw->feat_conv_w, raw->depth_conv_w, 8, nullptr, nullptr, nullptr, nullptr, 1e-5f, s.cin,
                               s.cout, s.kind, p, p + wf, p + wf + cp, (hipStream_t)stream);
            launch_conv3d_pk8_pack(raw->feat_conv_w, raw->depth_conv_w, s.cin, p + wf + 2 * cp, (hipStream_t)stream);
            launch_conv3d_b4_pack(raw->feat_conv_w, raw->depth_conv_w, s.cin, p + wf + 2 * cp + conv3d_pk8_packed_floats(s.cin),
                                  (hipStream_t)stream);
        }
        p += layer_floats(s);
    }
    return check_launch("cost_reg_pack");
}
size_t enerf_cost_reg_workspace_bytes(int full, int B, int D, int h, int w) {
    long long n0 = (long long)B * D * h * w, n1 = n0 / 8, n2 = n1 / 8, n3 = n2 / 8;
    long long f = n0 * 8 /*c0*/ + n1 * 16 * 2 /*c1,c2*/ + n2 * 32 * 2 /*c3,c4*/ + n1 * 16 /*y9*/ + n0 * 8 /*y11*/;
    if (full) f += n3 * 64 * 2 + n2 * 32;
    return (size_t)f * sizeof(float);
}
int enerf_cost_reg(const float* packed, int in_channels, int full, const float* vol, int B, int D, int h, int w,
                   float* feat, float* prob, void* workspace, size_t workspace_bytes, const enerf_options_t* options,
                   enerf_stream_t stream) {
    return cost_reg_run(packed, in_channels, full, vol, 0, B, D, h, w, feat, prob, workspace, workspace_bytes, options,
                        (hipStream_t)stream);
}
}  // extern "C"
namespace enerf {
int cost_reg_run(const float* packed, int in_channels, int full, const float* vol, int vol_planar, int B, int D, int h, int w,
                 float* feat, float* prob, void* workspace, size_t workspace_bytes, const enerf_options_t* options,
                 hipStream_t stream, const CostRegHook* hook) {
    REQUIRE(packed && vol && feat && prob && workspace, "cost_reg: null pointer");
    REQUIRE(in_channels == 8 || in_channels == 16 || in_channels == 32, "cost_reg: in_channels=%d unsupported (8/16/32)",
            in_channels);
    REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "cost_reg: bad shape");
    const Options opt = resolve_options(options);
    int div = full ? 8 : 4;
    REQUIRE(D % div == 0 && h % div == 0 && w % div == 0, "cost_reg: D,h,w (%d,%d,%d) must be divisible by %d", D, h, w, div);
    if (workspace_bytes < enerf_cost_reg_workspace_bytes(full, B, D, h, w))
        return fail(ENERF_EWORKSPACE, "cost_reg: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    LayerSpec L[12];
    int n = costreg_layers(in_channels, full, L);
    Conv3dDesc desc[12];
    const float* p = packed;
    for (int i = 0; i < n; ++i) {
        long long wf = conv3d_packed_floats(L[i].cin, L[i].cout, L[i].kind);
        int cp = cdiv(L[i].cout, 16) * 16;
        desc[i] = {p, p + wf, p + wf + cp, L[i].cin, L[i].cout, L[i].kind, L[i].relu,
                   layer_pk8(L[i]) ? p + wf + 2 * cp : nullptr,                           // scale/shift = 1/0 without BN
                   layer_pk8(L[i]) ? p + wf + 2 * cp + conv3d_pk8_packed_floats(L[i].cin) : nullptr,
                   layer_t2pair(L[i]) ? p + wf + 2 * cp : nullptr, 0, 0};
        p += layer_floats(L[i]);
    }
    long long n0 = (long long)B * D * h * w, n1 = n0 / 8, n2 = n1 / 8, n3 = n2 / 8;
    float* ws = (float*)workspace;
    auto take = [&](long long nf) { float* r = ws; ws += nf; return r; };
    float *c0 = take(n0 * 8), *c1 = take(n1 * 16), *c2 = take(n1 * 16), *c3 = take(n2 * 32), *c4 = take(n2 * 32);
    float *y9 = take(n1 * 16), *y11 = take(n0 * 8);
    // layouts between producer/consumer pairs that are BOTH on the asynchronous kernels: the cost volume (warp -> conv0) and
    // conv11's output (-> fused heads) travel as channel-quad planes (conv3d_b4.hip k_conv3d_s1_b4g)
    if (vol_planar && !conv3d_routes_b4_glds(opt, n0, D))
        return fail(ENERF_EINVAL, "cost_reg: a quad-planar volume needs the asynchronously staged conv0 kernel (conv3d_b4 0/2, D % 4 == 0, >= conv3d_lds_min_voxels)");
    desc[0].in_planar = vol_planar;
    if (conv3d_routes_b4_glds(opt, n0, D) && conv3d_routes_t2_pair(opt, n1)) desc[n - 2].out_planar = desc[n - 1].in_planar = 1;
    int i = 0;
    bool ok = true;
    auto hooked = [&](int layer) { if (hook != nullptr && hook->after_layer == layer) hook->fn(hook->ctx); };
    ok &= launch_conv3d(desc[i++], vol, nullptr, c0, nullptr, B, D, h, w, opt, st);                    // conv0
    hooked(0);
    ok &= launch_conv3d(desc[i++], c0, nullptr, c1, nullptr, B, D, h, w, opt, st);                     // conv1 (s2)
    hooked(1);
    ok &= launch_conv3d(desc[i++], c1, nullptr, c2, nullptr, B, D / 2, h / 2, w / 2, opt, st);         // conv2
    hooked(2);
    ok &= launch_conv3d(desc[i++], c2, nullptr, c3, nullptr, B, D / 2, h / 2, w / 2, opt, st);         // conv3 (s2)
    ok &= launch_conv3d(desc[i++], c3, nullptr, c4, nullptr, B, D / 4, h / 4, w / 4, opt, st);         // conv4
    const float* x = c4;
    if (full) {
        float *c5 = take(n3 * 64), *c6 = take(n3 * 64), *y7 = take(n2 * 32);
        ok &= launch_conv3d(desc[i++], c4, nullptr, c5, nullptr, B, D / 4, h / 4, w / 4, opt, st);     // conv5 (s2)
        ok &= launch_conv3d(desc[i++], c5, nullptr, c6, nullptr, B, D / 8, h / 8, w / 8, opt, st);     // conv6
        ok &= launch_conv3d(desc[i++], c6, c4, y7, nullptr, B, D / 8, h / 8, w / 8, opt, st);          // conv4 + conv7
        x = y7;
    }
    ok &= launch_conv3d(desc[i++], x, c2, y9, nullptr, B, D / 4, h / 4, w / 4, opt, st);               // conv2 + conv9
    ok &= launch_conv3d(desc[i++], y9, c0, y11, nullptr, B, D / 2, h / 2, w / 2, opt, st);             // conv0 + conv11
    ok &= launch_conv3d(desc[i++], y11, nullptr, feat, prob, B, D, h, w, opt, st);                     // feat_conv ++ depth_conv
    if (!ok) return fail(ENERF_EINVAL, "cost_reg: a layer shape has no kernel (in_channels=%d full=%d)", in_channels, full);
    return check_launch("cost_reg");
}
}  // namespace enerf
extern "C" {

int enerf_depth_regression(const float* prob, const float* depth_values, int B, int D, int h, int w, int depth_inv,
                           float* depth, float* std, enerf_stream_t stream) {
    REQUIRE(prob && depth_values && depth && std && B > 0 && D > 0 && h > 0 && w > 0, "depth_regression: bad arguments");
    launch_depth_regression(prob, depth_values, B, D, h, w, depth_inv, depth, std, nullptr, (hipStream_t)stream);
    return check_launch("depth_regression");
}
int enerf_build_rays(const float* rays8, const float* depth, const float* std, const float* near_far, int B, int N,
                     int h, int w, int Hr, int Wr, int depth_inv, float* rays12, enerf_stream_t stream) {
    if (N == 0 && B > 0) return ENERF_OK;
    REQUIRE(rays8 && depth && std && near_far && rays12, "build_rays: null pointer");
    REQUIRE(B > 0 && N >= 0 && h > 0 && w > 0 && Hr >= h && Wr >= w, "build_rays: bad shape");
    launch_build_rays(rays8, depth, std, near_far, B, N, h, w, Hr, Wr, depth_inv, rays12, (hipStream_t)stream);
    return check_launch("build_rays");
}

long long enerf_nerf_packed_floats(int F) { return nerf_packed_floats(F); }
int enerf_nerf_pack(const enerf_nerf_raw_t* raw, int F, int viewdir_agg, float* packed, enerf_stream_t stream) {
    REQUIRE(raw && packed, "nerf_pack: null pointer");
    REQUIRE(F == 11 || F == 35, "nerf_pack: F=%d unsupported (feat_ch+3 must be 11 or 35)", F);
    REQUIRE(raw->glob_w && raw->glob_b && raw->aggw_w && raw->aggw_b && raw->fc_w && raw->fc_b && raw->lr0_w &&
                raw->lr0_b && raw->sigma_w && raw->sigma_b && raw->col0_w && raw->col0_b && raw->col2_w && raw->col2_b,
            "nerf_pack: missing parameter");
    if (viewdir_agg) REQUIRE(raw->view_w && raw->view_b, "nerf_pack: view_fc missing");
    launch_nerf_pack(*raw, F, viewdir_agg, packed, (hipStream_t)stream);
    return check_launch("nerf_pack");
}
int enerf_render_rays(const enerf_render_args_t* a, enerf_stream_t stream) {
    REQUIRE(a, "render_rays: null args");
    if (a->N == 0 && a->B > 0) return ENERF_OK;              // empty ray list (e.g. an all-false mask_at_box)
    REQUIRE((a->rays12 || a->rays8) && a->tex && a->vol && a->src_exts && a->src_ixts && a->tar_ext && a->packed &&
                a->rgb && a->depth && a->weights, "render_rays: null pointer");
    if (a->rays8)
        REQUIRE(a->depth_map && a->std_map && a->nf_map && a->map_h > 0 && a->map_w > 0,
                "render_rays: fused build_rays needs depth/std/near_far maps and their size");
    REQUIRE(a->B > 0 && a->N >= 0 && a->Hr > 1 && a->Wr > 1 && a->D > 0 && a->h > 0 && a->w > 0, "render_rays: bad shape");
    if (a->ray_index) REQUIRE(a->ray_count && a->B == 1, "render_rays: ray_index needs ray_count and B == 1");
    if (a->N == 0) return ENERF_OK;
    int rc = launch_render_rays(*a, (hipStream_t)stream);
    if (rc != 0)
        return fail(ENERF_EINVAL, "render_rays: unsupported configuration (code %d: F=%d S=%d n_samples=%d B=%d)", rc,
                    a->F, a->S, a->n_samples, a->B);
    return check_launch("render_rays");
}

}  // extern "C"

// ---- FeatureNet driver ------------------------------------------------------------------------------
namespace {
struct FLayer { int cin, cout, k, stride, relu; };
// order: conv0.0 conv0.1 conv1.0 conv1.1 conv2.0 conv2.1 toplayer lat1 lat0 smooth1 smooth0
const FLayer kFeat[11] = {{3, 8, 3, 1, 1},  {8, 8, 3, 1, 1},   {8, 16, 5, 2, 1},  {16, 16, 3, 1, 1}, {16, 32, 5, 2, 1},
                          {32, 32, 3, 1, 1}, {32, 32, 1, 1, 0}, {16, 32, 1, 1, 0}, {8, 32, 1, 1, 0},  {32, 16, 3, 1, 0},
                          {32, 8, 3, 1, 0}};
long long flayer_floats(const FLayer& f) { return conv2d_packed_floats(f.cin, f.cout, f.k) + 2 * cdiv(f.cout, 16) * 16; }
}  // namespace

extern "C" {
long long enerf_feature_net_packed_floats(void) {
    long long t = 0;
    for (int i = 0; i < 11; ++i) t += flayer_floats(kFeat[i]);
    // tail: raw lat0 weight (32x8) + bias (32) for the fused smooth0 kernel, smooth0's P/Q image (3072), smooth0's broadcast-A image
    // (round 5: 2 passes x 18 registers x 64 lanes), conv0.0's (4 x 64) and conv0.1's (9 x 64)
    return t + 320 + 3072 + 2304 + 256 + 576;
}
int enerf_feature_net_pack(const enerf_featnet_raw_t* raw, float* packed, enerf_stream_t stream) {
    REQUIRE(raw && packed, "feature_net_pack: null pointer");
    const float* pw[5] = {raw->toplayer_w, raw->lat1_w, raw->lat0_w, raw->smooth1_w, raw->smooth0_w};
    const float* pb[5] = {raw->toplayer_b, raw->lat1_b, raw->lat0_b, raw->smooth1_b, raw->smooth0_b};
    float* p = packed;
    for (int i = 0; i < 11; ++i) {
        const FLayer& f = kFeat[i];
        long long wf = conv2d_packed_floats(f.cin, f.cout, f.k);
        int cp = cdiv(f.cout, 16) * 16;
        if (i < 6) {
            const enerf_conv_bn_t& c = raw->conv[i];
            REQUIRE(c.w && c.bn_weight && c.bn_bias && c.bn_mean && c.bn_var, "feature_net_pack: conv %d missing", i);
            launch_conv2d_pack(c.w, nullptr, c.bn_weight, c.bn_bias, c.bn_mean, c.bn_var, 1e-5f, f.cin, f.cout, f.k, p,
                               p + wf, p + wf + cp, (hipStream_t)stream);
        } else {
            REQUIRE(pw[i - 6] && pb[i - 6], "feature_net_pack: FPN conv %d missing", i);
            launch_conv2d_pack(pw[i - 6], pb[i - 6], nullptr, nullptr, nullptr, nullptr, 1e-5f, f.cin, f.cout, f.k, p,
                               p + wf, p + wf + cp, (hipStream_t)stream);
        }
        p += flayer_floats(f);
    }
    hipMemcpyAsync(p, raw->lat0_w, 256 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    hipMemcpyAsync(p + 256, raw->lat0_b, 32 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    zero_async(p + 320, 3072 * sizeof(float), (hipStream_t)stream);                      // reserved (rounds 2-5: smooth0's tap-packed P/Q image)
    launch_conv2d_cb_pack(raw->smooth0_w, 32, 0, 16, p + 320 + 3072, (hipStream_t)stream);          // broadcast-A images (conv2d.hip
    launch_conv2d_cb_pack(raw->smooth0_w, 32, 16, 16, p + 320 + 3072 + 1152, (hipStream_t)stream);  //  k_smooth0_cb, k_conv0_fused_cb)
    launch_conv2d_cb_pack(raw->conv[0].w, 3, 0, 3, p + 320 + 3072 + 2304, (hipStream_t)stream);
    launch_conv2d_cb_pack(raw->conv[1].w, 8, 0, 8, p + 320 + 3072 + 2304 + 256, (hipStream_t)stream);
    return check_launch("feature_net_pack");
}
size_t enerf_feature_net_workspace_bytes(int n_img, int H, int W) {
    long long p0 = (long long)n_img * H * W, p1 = p0 / 4, p2 = p1 / 4;
    // c0a(8) c0(8) f0pre(32) @full ; c1a(16) c1(16) f1pre(32) @half ; c2a(32) c2(32) @quarter
    return (size_t)(p0 * (8 + 8 + 32) + p1 * (16 + 16 + 32) + p2 * (32 + 32)) * sizeof(float);
}
int enerf_feature_net(const float* packed, const float* src_inps, int n_img, int H, int W, float* feat_l0,
                      float* feat_l1, float* feat_l2, int l2_stride, void* workspace, size_t workspace_bytes,
                      const enerf_options_t* options, enerf_stream_t stream) {
    return enerf_feature_net_stage(packed, src_inps, n_img, H, W, feat_l0, feat_l1, feat_l2, l2_stride, workspace,
                                   workspace_bytes, ENERF_FEAT_ALL, options, stream);
}
int enerf_feature_net_stage(const float* packed, const float* src_inps, int n_img, int H, int W, float* feat_l0,
                            float* feat_l1, float* feat_l2, int l2_stride, void* workspace, size_t workspace_bytes,
                            int stage, const enerf_options_t* options, enerf_stream_t stream) {
    return enerf::feature_net_stage_job(packed, src_inps, n_img, H, W, feat_l0, feat_l1, feat_l2, l2_stride, workspace, workspace_bytes,
                                        stage, options, (hipStream_t)stream, nullptr, nullptr);
}
}  // extern "C"
namespace enerf {
int feature_net_stage_job(const float* packed, const float* src_inps, int n_img, int H, int W, float* feat_l0, float* feat_l1,
                          float* feat_l2, int l2_stride, void* workspace, size_t workspace_bytes, int stage,
                          const enerf_options_t* options, hipStream_t stream, const PrepJob* job, int* job_done) {
    if (job_done != nullptr) *job_done = 0;
    const Options opt = resolve_options(options);
    REQUIRE(stage >= ENERF_FEAT_ALL && stage <= ENERF_FEAT_LEVEL2, "feature_net: unknown stage %d", stage);
    const bool trunk = stage == ENERF_FEAT_ALL || stage == ENERF_FEAT_TRUNK;
    const bool lvl1 = stage == ENERF_FEAT_ALL || stage == ENERF_FEAT_LEVEL1;
    const bool lvl2 = stage == ENERF_FEAT_ALL || stage == ENERF_FEAT_LEVEL2;
    REQUIRE(packed && src_inps && feat_l0 && feat_l1 && feat_l2 && workspace, "feature_net: null pointer");
    REQUIRE(n_img > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "feature_net: H,W (%d,%d) must be divisible by 4", H, W);
    REQUIRE(l2_stride == 8 || l2_stride == 12, "feature_net: l2_stride must be 8 (features) or 12 (texels)");
    if (workspace_bytes < enerf_feature_net_workspace_bytes(n_img, H, W))
        return fail(ENERF_EWORKSPACE, "feature_net: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Conv2dDesc d[11];
    const float* p = packed;
    for (int i = 0; i < 11; ++i) {
        const FLayer& f = kFeat[i];
        long long wf = conv2d_packed_floats(f.cin, f.cout, f.k);
        int cp = cdiv(f.cout, 16) * 16;
        d[i] = {p, p + wf, p + wf + cp, f.cin, f.cout, f.k, f.stride, f.relu, 0, nullptr, nullptr, nullptr};
        p += flayer_floats(f);
    }
    const long long p0 = (long long)n_img * H * W, p1 = p0 / 4, p2 = p1 / 4;
    const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4;
    float* ws = (float*)workspace;
    auto take = [&](long long nf) { float* r = ws; ws += nf; return r; };
    float *c0a = take(p0 * 8), *c0 = take(p0 * 8), *f0pre = take(p0 * 32);
    float *c1a = take(p1 * 16), *c1 = take(p1 * 16), *f1pre = take(p1 * 32);
    float *c2a = take(p2 * 32), *c2 = take(p2 * 32);
    int rc = 0;
    if (trunk) {
        if (!opt.featnet_unfused) {
            const bool took = launch_conv0_fused(d[0], d[1], p + 320 + 3072 + 2304, p + 320 + 3072 + 2304 + 256, src_inps, c0, n_img, H, W, st, job);   // conv0.1(conv0.0(image))
            if (took && job_done != nullptr) *job_done = 1;
        } else {
            rc |= launch_conv2d(d[0], src_inps, c0a, nullptr, n_img, H, W, 0, 0, st);  // conv0.0 (NCHW image in)
            rc |= launch_conv2d(d[1], c0a, c0, nullptr, n_img, H, W, 0, 0, st);        // conv0.1
        }
        rc |= launch_conv2d(d[2], c0, c1a, nullptr, n_img, H, W, 0, 0, st);            // conv1.0 (s2)
        rc |= launch_conv2d(d[3], c1a, c1, nullptr, n_img, H1, W1, 0, 0, st);          // conv1.1
        rc |= launch_conv2d(d[4], c1, c2a, nullptr, n_img, H1, W1, 0, 0, st);          // conv2.0 (s2)
        if (!opt.featnet_unfused) {
            d[5].chain_w = d[6].w;                                                     // toplayer (1x1, bias) runs in
            d[5].chain_shift = d[6].shift;                                             // conv2.1's epilogue -> level_0
            rc |= launch_conv2d(d[5], c2a, feat_l0, nullptr, n_img, H2, W2, 0, 0, st);
        } else {
            rc |= launch_conv2d(d[5], c2a, c2, nullptr, n_img, H2, W2, 0, 0, st);      // conv2.1
            rc |= launch_conv2d(d[6], c2, feat_l0, nullptr, n_img, H2, W2, 0, 0, st);  // toplayer  -> level_0
        }
    }
    if (lvl1) {
        if (opt.featnet_unfused || !launch_smooth1_fused(d[7], d[9], c1, feat_l0, f1pre, feat_l1, n_img, H1, W1, st)) {
            rc |= launch_conv2d(d[7], c1, f1pre, feat_l0, n_img, H1, W1, H2, W2, st);      // up2(feat2) + lat1(conv1)
            rc |= launch_conv2d(d[9], f1pre, feat_l1, nullptr, n_img, H1, W1, 0, 0, st);   // smooth1   -> level_1
        }
    }
    if (lvl2) {
        d[10].out_stride = l2_stride;
        d[10].rgb_src = (l2_stride == 12) ? src_inps : nullptr;
        if (!opt.featnet_unfused && !opt.featnet_smooth0_plain) {
            // smooth0(up2(feat1) + lat0(conv0)) in one kernel: the 32-channel full-res sum never touches HBM
            launch_smooth0_fused(d[10], c0, f1pre, p, p + 256, p + 320 + 3072, feat_l2, n_img, H, W, st);
        } else {                                                                       // the two plain 16x16x4-MFMA launches
            rc |= launch_conv2d(d[8], c0, f0pre, f1pre, n_img, H, W, H1, W1, st);      // up2(feat1) + lat0(conv0)
            rc |= launch_conv2d(d[10], f0pre, feat_l2, nullptr, n_img, H, W, 0, 0, st);  // smooth0 -> level_2 / texels
        }
    }
    if (rc != 0) return fail(ENERF_EINVAL, "feature_net: unsupported layer shape");
    return check_launch("feature_net");
}
}  // namespace enerf
extern "C" {
int enerf_pack_texels_cl(const float* feat_cl, int C, const float* src_inps, int H, int W, int Hr, int Wr, int tex,
                         int n_img, float* out, enerf_stream_t stream) {
    REQUIRE(feat_cl && src_inps && out, "pack_texels_cl: null pointer");
    REQUIRE(C > 0 && C % 4 == 0 && tex >= C + 3 && tex % 4 == 0 && n_img > 0 && Hr > 0 && Wr > 0, "pack_texels_cl: bad shape");
    launch_pack_texels_cl(feat_cl, C, src_inps, H, W, Hr, Wr, tex, n_img, out, (hipStream_t)stream);
    return check_launch("pack_texels_cl");
}
}  // extern "C"

extern "C" {
int enerf_gen_rays(const float* tar_ext, const float* tar_ixt, int B, int Hr, int Wr, float scale, float* rays,
                   enerf_stream_t stream) {
    REQUIRE(tar_ext && tar_ixt && rays && B > 0 && Hr > 0 && Wr > 0 && scale > 0.f, "gen_rays: bad arguments");
    launch_gen_rays(tar_ext, tar_ixt, B, Hr, Wr, scale, rays, (hipStream_t)stream);
    return check_launch("gen_rays");
}
int enerf_pack_rgb8(const float* rgb, int H, int W, int flip, unsigned char* out, enerf_stream_t stream) {
    REQUIRE(rgb && out && H > 0 && W > 0, "pack_rgb8: bad arguments");
    launch_pack_rgb8(rgb, H, W, flip, out, (hipStream_t)stream);
    return check_launch("pack_rgb8");
}
int enerf_eval_stats(const float* pred_rgb, const float* gt_rgb, const void* mask, int mask_elem_bytes, long long n_rgb,
                     int img_w, int img_h, int crop_h, int crop_w, const float* pred_depth, const float* gt_depth,
                     long long n_depth, double* acc, enerf_stream_t stream) {
    REQUIRE(acc && n_rgb >= 0 && n_depth >= 0, "eval_stats: bad arguments");
    if (n_rgb > 0) REQUIRE(pred_rgb && gt_rgb, "eval_stats: rgb pointers missing");
    if (n_depth > 0) REQUIRE(pred_depth && gt_depth, "eval_stats: depth pointers missing");
    if (mask) REQUIRE(mask_elem_bytes == 1 || mask_elem_bytes == 4, "eval_stats: mask must be uint8/bool or int32");
    if (img_w > 0) REQUIRE(img_h > 0 && crop_h >= 0 && crop_w >= 0 && n_rgb % ((long long)img_w * img_h) == 0,
                           "eval_stats: crop needs the image extent (n_rgb a multiple of img_w*img_h)");
    zero_async(acc, 6 * sizeof(double), (hipStream_t)stream);
    if (n_rgb + n_depth == 0) return ENERF_OK;
    launch_eval_stats(pred_rgb, gt_rgb, mask, mask_elem_bytes, n_rgb, img_w, img_h, crop_h, crop_w, pred_depth, gt_depth,
                      n_depth, acc, (hipStream_t)stream);
    return check_launch("eval_stats");
}
int enerf_gen_rays_at(const float* tar_ext, const float* tar_ixt, const int* xy, int B, int N, float scale, float* rays,
                      enerf_stream_t stream) {
    REQUIRE(tar_ext && tar_ixt && xy && rays && B > 0 && N >= 0 && scale > 0.f, "gen_rays_at: bad arguments");
    if (N == 0) return ENERF_OK;
    launch_gen_rays_at(tar_ext, tar_ixt, xy, B, N, scale, rays, (hipStream_t)stream);
    return check_launch("gen_rays_at");
}
int enerf_rays_bbox_mask(const float* rays, const float* bounds, long long n, int* mask, enerf_stream_t stream) {
    REQUIRE(rays && bounds && mask && n > 0, "rays_bbox_mask: bad arguments");
    launch_rays_bbox_mask(rays, bounds, n, mask, (hipStream_t)stream);
    return check_launch("rays_bbox_mask");
}
int enerf_select_views(const float* cam_points, int V, const float* c2w, int k, int* idx, enerf_stream_t stream) {
    REQUIRE(cam_points && c2w && idx && V > 0 && V <= 1024 && k > 0 && k <= V, "select_views: bad arguments (V <= 1024, k <= V)");
    launch_select_views(cam_points, V, c2w, k, idx, (hipStream_t)stream);
    return check_launch("select_views");
}
int enerf_gather_views(const float* inps, const float* exts, const float* ixts, const int* idx, int k, int H, int W,
                       float* src_inps, float* src_exts, float* src_ixts, enerf_stream_t stream) {
    REQUIRE(inps && exts && ixts && idx && src_inps && src_exts && src_ixts && k > 0 && H > 0 && W > 0 &&
                (long long)k * H * W >= 16LL * k, "gather_views: bad arguments");
    launch_gather_views(inps, exts, ixts, idx, k, H, W, src_inps, src_exts, src_ixts, (hipStream_t)stream);
    return check_launch("gather_views");
}
}  // extern "C"
