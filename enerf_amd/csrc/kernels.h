// kernels.h — host-side launchers implemented by the .hip files in this directory.
// All pointers are device pointers; nothing here allocates or synchronises.
#pragma once
#include "common.h"
#include "../../include/enerf_hip.h"

namespace enerf {

// Explicit kernel-variant choices (include/enerf_hip.h: enerf_options_t); a NULL pointer at the C ABI means all zero.
using Options = enerf_options_t;
inline Options resolve_options(const enerf_options_t* o) { return o ? *o : Options{}; }
// error reporting shared by the C-ABI translation units (capi.hip, frame.hip): thread-local message + code
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
const char* last_error();
#define REQUIRE(cond, ...) \
    do { if (!(cond)) return ::enerf::fail(ENERF_EINVAL, __VA_ARGS__); } while (0)
// number of compute units of the current device (queried once per device; 256 on MI355X)
int device_cu_count();
// Zero `bytes` (a multiple of 4) on the stream with a fill KERNEL: a hipMemsetAsync becomes a memset node when the stream is
// being captured, and those did not replay reliably inside a whole-training-step hipGraph on this stack.
void zero_async(void* p, size_t bytes, hipStream_t st);
void zero_async2(void* p, size_t bytes_p, void* q, size_t bytes_q, hipStream_t st);

// ---- geometry.hip -------------------------------------------------------------------------------
void launch_channels_last(const float* src, float* dst, int n, int C, long long P, int Cpad, hipStream_t st);
void launch_channels_first(const float* src, float* dst, int n, int C, long long P, int Cpad, hipStream_t st);
void launch_pack_img_feat_rgb(const float* im_feat, int C, int Hf, int Wf, const float* src_inps, int H, int W,
                              int Hr, int Wr, int tex, int n_img, float* out, hipStream_t st);
void launch_proj_mats(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int B,
                      int S, float src_scale, float tar_scale, float* proj, hipStream_t st);
void launch_depth_values(const float* near_far, const float* pdepth, const float* pstd, const float* pnf, int B, int D,
                         int h, int w, int hp, int wp, int depth_inv, float* dv, float* nf_out, hipStream_t st);
void launch_level_prep(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int S,
                       float src_scale, float tar_scale, float* proj, const float* near_far, const float* pdepth,
                       const float* pstd, const float* pnf, int B, int D, int h, int w, int hp, int wp, int depth_inv,
                       float* dv, float* nf_out, hipStream_t st);     // proj_mats + depth_values in one launch
// depth_mvs (optional): 1/depth for disparity-space levels, depth otherwise (network.py:105-108)
void launch_depth_regression(const float* prob, const float* dv, int B, int D, int h, int w, int depth_inv,
                             float* depth, float* std, float* depth_mvs, hipStream_t st);
// depth_regression of the previous (not rendered) level + proj_mats + depth_values of this level in one launch; false = shape
// not handled, nothing launched
bool launch_regress_and_values(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int S,
                               float src_scale, float tar_scale, float* proj, const float* prob_p, const float* dv_p,
                               const float* nf_p, int Dp, int hp, int wp, int depth_inv_p, float* depth_p, float* std_p, int B,
                               int D, int h, int w, int depth_inv, float* dv, float* nf_out, hipStream_t st);
void launch_build_rays(const float* rays8, const float* depth, const float* std, const float* nf, int B, int N, int h,
                       int w, int Hr, int Wr, int depth_inv, float* rays12, hipStream_t st);

// ---- volume.hip ---------------------------------------------------------------------------------
// planar = 1: vol as channel-quad planes (B, C/4, D, h, w, 4) instead of channels-last (B, D, h, w, C)
void launch_feature_volume(const float* feat_nhwc, const float* proj, const float* dv, int B, int S, int C, int Hs,
                           int Ws, int D, int h, int w, float* vol, hipStream_t st, int planar = 0);
// does enerf_cost_reg read a quad-planar volume for this shape under these options? (enerf_forward asks before the warp)
bool cost_reg_wants_planar_volume(const enerf_options_t& o, int in_channels, int B, int D, int h, int w);
bool conv3d_routes_b4_glds(const enerf_options_t& o, long long vox, int D);     // mirrors launch_conv3d's routing
bool conv3d_routes_t2_pair(const enerf_options_t& o, long long vox_in);
// enerf_cost_reg with the volume layout made explicit (vol_planar = 1: channel-quad planes, see launch_feature_volume)
// hook: called on the host right after layer `after_layer` (0 = conv0) has been enqueued (enerf_forward uses it to start a
// side-lane stage at that point of the chain); nullptr = none
struct CostRegHook { void (*fn)(void* ctx); void* ctx; int after_layer; };
int cost_reg_run(const float* packed, int in_channels, int full, const float* vol, int vol_planar, int B, int D, int h, int w,
                 float* feat, float* prob, void* workspace, size_t workspace_bytes, const enerf_options_t* options, hipStream_t st,
                 const CostRegHook* hook = nullptr);

// ---- conv3d.hip ---------------------------------------------------------------------------------
enum ConvKind { kConvS1 = 0, kConvS2 = 1, kConvT2 = 2 };
struct Conv3dDesc {
    const float* w;         // packed A operands (see conv3d.hip)
    const float* scale;     // per-cout epilogue scale (BN folded; 1 without BN), padded to 16*row tiles
    const float* shift;     // per-cout epilogue shift (0 without BN)
    int cin, cout, kind, relu;
    const float* w_pk8;     // tap-packed image for stride-1 cout=8(+1) layers (conv3d_pk8.hip) or nullptr
    const float* w_b4;      // batched-4x4 image for the same layers (conv3d_b4.hip) or nullptr
    const float* w_t2pair;  // x-parity-paired image of a transposed 16 -> 8 layer (conv3d_t2.hip) or nullptr
    int in_planar;          // input is channel-quad planes (B, cin/4, D, H, W, 4): only the glds b4 kernel reads that
    int out_planar;         // output as channel-quad planes: only the class-paired transposed kernel writes that
};
long long conv3d_t2_pair_floats();
void launch_conv3d_t2_pair_pack(const float* packed, float* paired, hipStream_t st);   // from the class-major packed image
// batched 4x4x1 variant for cout = 8 (+ optional depth row on the VALU): see conv3d_b4.hip
long long conv3d_b4_packed_floats(int cin);
void launch_conv3d_b4_pack(const float* w, const float* wd, int cin, float* packed, hipStream_t st);
bool launch_conv3d_b4(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W, bool glds,
                      hipStream_t st);   // glds: the asynchronously staged variant (global_load_lds, two LDS buffers)
// tap-packed variant for cout = 8 (+ optional depth row): see conv3d_pk8.hip
long long conv3d_pk8_packed_floats(int cin);
void launch_conv3d_pk8_pack(const float* w, const float* wd, int cin, float* packed, hipStream_t st);
bool launch_conv3d_s2_lds(const Conv3dDesc& L, const float* in, float* out, int B, int Di, int Hi, int Wi,
                          hipStream_t st);   // LDS-staged stride-2 variant, Cin = 8, Cout <= 16 (conv3d_s2.hip)
bool launch_conv3d_t2_lds(const Conv3dDesc& L, const float* in, const float* residual, float* out, int B, int Di, int Hi,
                          int Wi, hipStream_t st);   // LDS-staged transposed variant, 16 -> 8 (conv3d_t2.hip)
bool launch_conv3d_t2_all(const Conv3dDesc& L, const float* in, const float* residual, float* out, int B, int Di, int Hi,
                          int Wi, hipStream_t st);   // every-class transposed kernel: 16 -> 8 (class-paired), 32 -> 16 (conv3d_t2.hip)
// small deep stride-1 / stride-2 layers (Cin 16 / 32 / 64, Cout % 16 == 0): block-shared weight tile in LDS, operands up front (conv3d_wl.hip)
bool launch_conv3d_wl(const Conv3dDesc& L, const float* in, float* out, int B, int Di, int Hi, int Wi, hipStream_t st);
bool launch_conv3d_pk8(const Conv3dDesc& L, const float* in, float* out, float* out2, int B, int D, int H, int W,
                       bool all_layers, hipStream_t st);
// number of floats of the packed weight image for a layer
long long conv3d_packed_floats(int cin, int cout, int kind);
// pack torch-layout weights (Conv3d: (cout,cin,3,3,3); ConvTranspose3d: (cin,cout,3,3,3)) + BN into
// {packed A operands, scale[coutpad], shift[coutpad]}
// rows [0,cout1) of the GEMM come from w, rows [cout1,cout) from w2 (used to fuse feat_conv ++ depth_conv)
void launch_conv3d_pack(const float* w, const float* w2, int cout1, const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var,
                        float eps, int cin, int cout, int kind, float* packed, float* scale, float* shift,
                        hipStream_t st);
// in: (B, Di, Hi, Wi, cin) channels-last; out: (B, Do, Ho, Wo, cout_store); residual (same shape as out) optional.
// cout_store lets the fused heads write feat (8 ch) and prob (1 ch) to two tensors: if out2 != nullptr,
// channels [0,8) go to out (stride 8) and channel 8 goes to out2 (stride 1).
// Returns false when no kernel handles the layer shape (nothing launched).
bool launch_conv3d(const Conv3dDesc& L, const float* in, const float* residual, float* out, float* out2, int B, int Di,
                   int Hi, int Wi, const Options& o, hipStream_t st);

// ---- conv2d.hip (FeatureNet) ----------------------------------------------------------------------
struct Conv2dDesc {
    const float* w;        // packed A operands
    const float* scale;    // per-cout scale (BN folded; 1 for plain convs)
    const float* shift;    // per-cout shift (BN folded, or the conv bias)
    int cin, cout, k, stride, relu;
    int out_stride;        // floats between consecutive output pixels (0 = cout)
    const float* rgb_src;  // texel mode: (n,3,Ho,Wo) images appended as [rgb*0.5+0.5 | 0] behind the features
    const float* chain_w;      // packed weights / shift (bias) of a following 1x1 conv (cout -> 32) applied in the
    const float* chain_shift;  // epilogue instead of storing this layer's output (conv2.1 -> toplayer), or nullptr
};
long long conv2d_packed_floats(int cin, int cout, int k);
void launch_conv2d_pack(const float* w, const float* bias, const float* bn_w, const float* bn_b, const float* bn_mean,
                        const float* bn_var, float eps, int cin, int cout, int k, float* packed, float* scale,
                        float* shift, hipStream_t st);
// in: channels-last (N,Hi,Wi,cin) — or the NCHW image batch for the 3-channel first layer; out: channels-last.
// up (optional): coarser channels-last map (N,Hc,Wc,cout) added after a x2 align-corners bilinear upsample.
int launch_conv2d(const Conv2dDesc& L, const float* in, float* out, const float* up, int N, int Hi, int Wi, int Hc,
                  int Wc, hipStream_t st);
// conv0.1(conv0.0(image)) fused (feature_net.py:7-9): L0/L1 = the two layers' descriptors, img (N,3,H,W) -> out (N,H,W,8)
// job (prep_job.h, optional): the frame's camera-only preparation carried by extra blocks of this launch; returns whether it was
struct PrepJob;
bool launch_conv0_fused(const Conv2dDesc& L0, const Conv2dDesc& L1, const float* w_cb0, const float* w_cb1, const float* img,
                        float* out, int N, int H, int W, hipStream_t st, const PrepJob* job = nullptr);
// enerf_feature_net_stage with a preparation job for the trunk's first launch; *job_done = 1 when a kernel carried it
int feature_net_stage_job(const float* packed, const float* src_inps, int n_img, int H, int W, float* feat_l0, float* feat_l1,
                          float* feat_l2, int l2_stride, void* workspace, size_t workspace_bytes, int stage,
                          const enerf_options_t* options, hipStream_t stream, const PrepJob* job, int* job_done);
// smooth0(up2(f1pre) + lat0(c0)) fused (feature_net.py:32-35); L = smooth0's descriptor, lat_w/lat_b raw (32,8)/(32)
// smooth1(up2(f2) + lat1(c1)) fused (feature_net.py:33-34, round 5): writes f1pre (N,H1,W1,32) and out (N,H1,W1,16); false: not applicable
bool launch_smooth1_fused(const Conv2dDesc& Llat, const Conv2dDesc& Lsm, const float* c1, const float* f2, float* f1pre, float* out,
                          int N, int H1, int W1, hipStream_t st);
// w_cb: the layer's broadcast-A image (launch_conv2d_cb_pack)
void launch_smooth0_fused(const Conv2dDesc& L, const float* c0, const float* f1pre, const float* lat_w, const float* lat_b,
                          const float* w_cb, float* out, int N, int H, int W, hipStream_t st);
// broadcast-A image (common.h mfma4_bc) of input channels ci0 .. ci0+cinp-1 of a 3x3, Cout = 8 layer: ceil(18*cinp/16)*64 floats
void launch_conv2d_cb_pack(const float* w, int cin, int ci0, int cinp, float* packed, hipStream_t st);
// texels from channels-last features at the render resolution + resized colours (general case)
void launch_pack_texels_cl(const float* feat_cl, int C, const float* src_inps, int H, int W, int Hr, int Wr, int tex,
                           int n_img, float* out, hipStream_t st);

// ---- io.hip (the steps before/after the path: SURVEY.md 8f rows 3,4) --------------------------------
void launch_gen_rays(const float* tar_ext, const float* tar_ixt, int B, int Hr, int Wr, float scale, float* rays,
                     hipStream_t st);
void launch_pack_rgb8(const float* rgb, int H, int W, int flip, unsigned char* out, hipStream_t st);
void launch_eval_stats(const float* pred_rgb, const float* gt_rgb, const void* mask, int mask_bytes, long long n_rgb,
                       int img_w, int img_h, int crop_h, int crop_w, const float* pred_depth, const float* gt_depth,
                       long long n_depth, double* acc, hipStream_t st);
void launch_gen_rays_at(const float* tar_ext, const float* tar_ixt, const int* xy, int B, int N, float scale, float* rays,
                        hipStream_t st);
void launch_rays_bbox_mask(const float* rays, const float* bounds, long long n, int* mask, hipStream_t st);
void launch_select_views(const float* cam_points, int V, const float* c2w, int k, int* idx, hipStream_t st);
void launch_gather_views(const float* inps, const float* exts, const float* ixts, const int* idx, int k, int H, int W,
                         float* src_inps, float* src_exts, float* src_ixts, hipStream_t st);

// ---- frame.hip (mask_at_box compaction; the whole-frame driver enerf_forward lives there too) ---------
size_t mask_compact_workspace_bytes(long long n);
void launch_mask_compact(const void* mask, int elem_bytes, long long n, int* index, int* count, void* workspace,
                         hipStream_t st);

// ---- render.hip ---------------------------------------------------------------------------------
using NerfRaw = enerf_nerf_raw_t;     // torch-layout parameter pointers of one NeRF (nerf.py:6-89)
long long nerf_packed_floats(int feat_ch_plus3);
void launch_nerf_pack(const NerfRaw& raw, int F, int viewdir_agg, float* packed, hipStream_t st);
using RenderArgs = enerf_render_args_t;
int launch_render_rays(const RenderArgs& a, hipStream_t st);  // returns 0, or <0 for unsupported shapes

}  // namespace enerf
