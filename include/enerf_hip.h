/* enerf_hip.h — C ABI of the MI355X-native ENeRF rendering path (libenerf_hip.so).
 *
 * The reference (zju3dv/ENeRF) has no FFI: its hot path is a chain of torch ops inside
 * lib/networks/enerf/{network,utils,nerf,cost_reg_net}.py.  Each entry point below replaces the
 * reference function cited next to it, fused where the reference materialises intermediates.
 * Conventions:
 *   - every pointer is a DEVICE pointer to fp32 data unless stated; buffers are caller-owned
 *     (PyTorch tensors in the Python binding); nothing is allocated or freed here;
 *   - `stream` is a hipStream_t; calls only enqueue work (no host sync);
 *   - return 0 on success, a negative ENERF_E* code otherwise; enerf_last_error() gives the text
 *     (thread-local).
 * Layouts (HBM):  2-D features channels-last (n_img,H,W,C); 3-D volumes channels-last (B,D,h,w,C);
 *   depth hypotheses / probabilities (B,D,h,w); depth/std maps (B,h,w); near_far maps (B,2,h,w);
 *   rays (B,N,8) -> (B,N,12); texels (B,S,Hr,Wr,TEX) with TEX = 4*ceil((C+3)/4) = [feat C | rgb 3 | 0].
 */
#ifndef ENERF_HIP_H_
#define ENERF_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ENERF_ABI_VERSION 11
#define ENERF_OK 0
#define ENERF_EINVAL (-1)   /* bad argument / unsupported shape */
#define ENERF_ELAUNCH (-2)  /* HIP launch error */
#define ENERF_EWORKSPACE (-3)

typedef void* enerf_stream_t; /* hipStream_t */

int enerf_abi_version(void);
const char* enerf_last_error(void);

/* ABI v11: self-test of the gfx950-only primitives the kernels are written on (LDS-DMA copies, raw buffer loads, DPP / permlane lane
 * exchanges, the two matrix instructions' operand layouts incl. the broadcast-A form, med3 ReLU, 24-bit multiplies, LDS / global
 * fp32 atomics, hardware rcp / sqrt / exp, wave-level LDS ordering, the XCD block map): each check computes the primitive and the
 * same quantity from values staged through memory with plain indexed reads, and counts disagreeing lanes into
 * mismatches[enerf_selftest_checks()] (device, zeroed here).  No reference function: it ties the intrinsic build to the lane
 * emulator's twins of the same helpers (csrc/common.h), on which the CPU test suite runs the kernel sources.
 * table: >= 1024 arbitrary device floats (a multiple of 4); scratch: blocks * 16 device floats. */
int enerf_selftest_checks(void);
int enerf_selftest_primitives(const float* table, int table_floats, float* scratch, int blocks, unsigned* mismatches, enerf_stream_t stream);

/* ---- kernel-variant choices, passed EXPLICITLY per call (ABI >= 3; no environment variables are read).
 * Every entry point that has more than one kernel variant takes a `const enerf_options_t*`; NULL (or a
 * zero-filled struct) selects the defaults, which are the single-frame-latency choices measured on MI355X.
 * Results of different variants agree to fp32 re-association (<= 2e-5 relative), not bit for bit. */
typedef struct {
    int conv3d_global_only;          /* 1: every 3-D conv through the global-load kernel (A/B, tests); 0: auto */
    long long conv3d_lds_min_voxels; /* smallest layer (output voxels) routed to the LDS-staged kernels; 0 = 16384 */
    int conv3d_pk8;                  /* tap-packed Cout=8 kernel: 0 = where it is faster alone (Cin=16, big heads),
                                        1 = never, 2 = every Cout=8(+1) layer (fewest MFMAs: throughput mode) */
    int featnet_unfused;             /* 1: one launch per FeatureNet layer (no conv0/toplayer/lat0 fusions) */
    int featnet_smooth0_plain;       /* 1: lat0 + smooth0 as the two plain 16x16x4-MFMA launches instead of the fused batched-4x4 kernel
                                        (ABI >= 10; rounds 2-5: a fused 16x16x4 kernel) */
    int conv3d_b4;                   /* batched 4x4x1-MFMA kernel for the Cout=8(+1) stride-1 3-D layers (conv0, fused heads):
                                        0 = on (default; since ABI 5 the asynchronously staged kernel: global_load_lds, two LDS
                                        buffers, one channel quad per pass, cost volume / conv11 output handed over as
                                        channel-quad planes), 1 = never (falls back to the tap-packed / plain kernels),
                                        2 = same as 0, 3 = on, the round-2 register-staged kernel (A/B) */
    int single_stream;               /* enerf_forward: 1 = every kernel of the frame on the caller's stream, in order.
                                        0 (default) = the FeatureNet's top-down half (lat1/smooth1, lat0/smooth0), which only
                                        level 1 and the final render consume, is forked onto a library-owned side stream and
                                        overlaps level 0's warp + cost regularisation; joined with events before its first
                                        consumer — still ONE frame, no frames in flight (ABI >= 4) */
    int side_gate;                   /* enerf_forward, ABI >= 5: WHEN the side lane starts the FeatureNet's last stage (lat0 + smooth0,
                                        the render texels: needed only at the end of the frame, and the largest side kernel).
                                        0 / 1 = ungated (default: right behind lat1/smooth1, overlapping level 0);
                                        2 = gated behind the LAST cascade level's conv0 (beside that level's small deep layers;
                                        measured 1-2 % slower on dtu / lego / zju: profiles/r03_ab_side_gate.txt);
                                        3 = at the start of the last level (after the previous depth regression);
                                        4 = behind the last level's conv2 */
    int fuse_depth_prep;             /* enerf_forward, ABI >= 5: 0 = the depth regression of a NOT rendered level runs inside the
                                        next level's prep launch (k_regress_and_values: one launch instead of two between
                                        cascade levels); 1 = separate launches (round 2) */
    int conv3d_t2_variant;           /* transposed 3-D layers (conv9 32->16, conv11 16->8), ABI >= 5: 0 = the every-class LDS kernel
                                        with x-parity-paired MFMA rows where it measured faster (default); 1 = the round-2
                                        kernels only (A/B); 2 = the every-class kernel for every layer it handles */
    int conv3d_small_variant;        /* deep (< 1024 wave-tile) stride-1/2 layers, ABI >= 5: 0 = default choice per layer (ABI >= 10: layers whose
                                        input is 1 - 2 planes thick on the block-shared-weight kernel that skips padding kd taps,
                                        conv3d_wl.hip; the others as 1);
                                        1 = taps split over 3 waves + LDS reduce, one tile per wave (the round-2 default);
                                        2 = no tap split, 2 row tiles x 2 column tiles per wave (operand reuse over wave count);
                                        3 = no tap split, 1 row tile x 4 column tiles;
                                        4 = every small layer on the block-shared-weight kernel (A/B) */
    int render_precision;            /* enerf_render_rays / enerf_forward, ABI >= 7, the cascade's last level (F = 11, <= 2 samples per
                                        ray): how the dense layers of the Agg/NeRF MLP are multiplied.
                                        0 / 1 (default) = exact fp32 16x16x4 MFMAs;
                                        2 = "bf16x3": every fp32 operand split into two bf16 pieces, three v_mfma_f32_16x16x32_bf16
                                            per 32-deep k-chunk, fp32 accumulation: kernel 192 -> 128 us, dtu +7 % frames/s, outputs
                                            within 7e-6 of max|ref| at 512x640 (profiles/r04_ab_render_precision_k32.txt) — but a
                                            head with a large gain amplifies the ~1e-5 operand error (tests/test_adversarial.py
                                            adv_sigma: 1.3e-3), so it is an opt-in, not the default;
                                        3 = "bf16x6": three pieces, six MFMAs: fp32-level accuracy, measured SLOWER than the exact
                                            kernel (203 us: the operand splits are VALU work, which does not overlap the MFMAs) */
} enerf_options_t;

/* ---- layout adapters at the PyTorch boundary (FeatureNet output is NCHW, network.py:58-67) ---- */
/* (n, C, P) -> (n, P, Cpad), pad channels zero-filled;  and back (drops the padding). */
int enerf_channels_last(const float* src, float* dst, int n, int C, long long P, int Cpad, enerf_stream_t stream);
int enerf_channels_first(const float* src, float* dst, int n, int C, long long P, int Cpad, enerf_stream_t stream);

/* unpreprocess (utils.py:605-612) + im_feat resize (network.py:29-32) + cat (network.py:34), written
 * as texels.  im_feat (n_img,C,Hf,Wf) NCHW, src_inps (n_img,3,H,W) in [-1,1]; out (n_img,Hr,Wr,tex). */
int enerf_pack_img_feat_rgb(const float* im_feat, int C, int Hf, int Wf, const float* src_inps, int H, int W,
                            int Hr, int Wr, int tex, int n_img, float* out, enerf_stream_t stream);

/* A bias-free convolution followed by BatchNorm in eval mode (ConvBnReLU utils.py:10-33). */
typedef struct {
    const float* w;         /* Conv2d (cout,cin,k,k) / Conv3d (cout,cin,3,3,3) / ConvTranspose3d (cin,cout,3,3,3) */
    const float* bn_weight; /* BatchNorm affine + running statistics (eps 1e-5) */
    const float* bn_bias;
    const float* bn_mean;
    const float* bn_var;
} enerf_conv_bn_t;

/* ---- FeatureNet (feature_net.py:4-36) + forward_feat's reshapes (network.py:58-67), in HIP.
 * BASELINE's north_star keeps the 2-D FPN in PyTorch-ROCm; it measured 49 % of the
 * frame there, so this optional entry point computes the same three feature maps on the matrix cores and
 * writes them CHANNELS-LAST (what the warp and render kernels read), removing the layout adapters.
 * src_inps (n_img,3,H,W) in [-1,1], H and W divisible by 4.  Outputs:
 *   feat_l0 (n_img,H/4,W/4,32) = reference feats['level_0'];  feat_l1 (n_img,H/2,W/2,16) = 'level_1';
 *   feat_l2 (n_img,H,W,l2_stride): l2_stride 8 -> plain 'level_2'; l2_stride 12 -> render texels
 *   [level_2 (8) | src*0.5+0.5 (3) | 0] (unpreprocess utils.py:605-612 at render_scale 1 + cat network.py:34). */
typedef struct {
    enerf_conv_bn_t conv[6];                      /* conv0.0, conv0.1, conv1.0, conv1.1, conv2.0, conv2.1 (Conv2d + BN2d) */
    const float *toplayer_w, *toplayer_b;         /* (32,32,1,1),(32) */
    const float *lat1_w, *lat1_b;                 /* (32,16,1,1) */
    const float *lat0_w, *lat0_b;                 /* (32,8,1,1) */
    const float *smooth1_w, *smooth1_b;           /* (16,32,3,3) */
    const float *smooth0_w, *smooth0_b;           /* (8,32,3,3) */
} enerf_featnet_raw_t;
long long enerf_feature_net_packed_floats(void);
int enerf_feature_net_pack(const enerf_featnet_raw_t* raw, float* packed, enerf_stream_t stream);
size_t enerf_feature_net_workspace_bytes(int n_img, int H, int W);
int enerf_feature_net(const float* packed, const float* src_inps, int n_img, int H, int W, float* feat_l0,
                      float* feat_l1, float* feat_l2, int l2_stride, void* workspace, size_t workspace_bytes,
                      const enerf_options_t* options, enerf_stream_t stream);
/* The same network in three independently enqueueable stages, so a host can overlap the level-0 cost volume
 * (which needs feat_l0 only) with the rest of the FPN on a second stream:
 *   ENERF_FEAT_TRUNK  conv0.0 .. conv2.1, toplayer  -> feat_l0           (feature_net.py:27-31)
 *   ENERF_FEAT_LEVEL1 lat1 + up2 + smooth1          -> feat_l1           (feature_net.py:32-33,36; needs TRUNK)
 *   ENERF_FEAT_LEVEL2 lat0 + up2 + smooth0          -> feat_l2 / texels  (feature_net.py:34-35;    needs LEVEL1)
 * Same arguments as enerf_feature_net (same workspace across the three calls); ordering between stages issued on
 * different streams is the caller's job (events).  ENERF_FEAT_ALL == enerf_feature_net. */
enum { ENERF_FEAT_ALL = 0, ENERF_FEAT_TRUNK = 1, ENERF_FEAT_LEVEL1 = 2, ENERF_FEAT_LEVEL2 = 3 };
int enerf_feature_net_stage(const float* packed, const float* src_inps, int n_img, int H, int W, float* feat_l0,
                            float* feat_l1, float* feat_l2, int l2_stride, void* workspace, size_t workspace_bytes,
                            int stage, const enerf_options_t* options, enerf_stream_t stream);
/* texels from channels-last features already at the render resolution (level-0 rendering with the HIP
 * FeatureNet): out (n_img,Hr,Wr,tex) = [feat (C) | bilinear_ac(src*0.5+0.5) (3) | 0]. */
int enerf_pack_texels_cl(const float* feat_cl, int C, const float* src_inps, int H, int W, int Hr, int Wr, int tex,
                         int n_img, float* out, enerf_stream_t stream);

/* ---- get_proj_mats (utils.py:35-55): proj (B,S,3,4) ---- */
int enerf_get_proj_mats(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext,
                        int B, int S, float src_scale, float tar_scale, float* proj, enerf_stream_t stream);

/* ---- get_depth_values (utils.py:98-151).  prev_* = NULL at level 0 (then near_far is (B,2));
 * otherwise prev_depth/prev_std (B,hp,wp) and prev_near_far (B,2,hp,wp) from the previous level
 * (disparity units).  Outputs depth_values (B,D,h,w), near_far_out (B,2,h,w). ---- */
int enerf_get_depth_values(const float* near_far, const float* prev_depth, const float* prev_std,
                           const float* prev_near_far, int B, int D, int h, int w, int hp, int wp, int depth_inv,
                           float* depth_values, float* near_far_out, enerf_stream_t stream);
/* get_proj_mats + get_depth_values of one cascade level in ONE launch (both only depend on the batch and the
 * previous level): same arguments and outputs as the two calls above. */
int enerf_level_prep(const float* src_ixts, const float* src_exts, const float* tar_ixt, const float* tar_ext, int B,
                     int S, float src_scale, float tar_scale, float* proj, const float* near_far,
                     const float* prev_depth, const float* prev_std, const float* prev_near_far, int D, int h, int w,
                     int hp, int wp, int depth_inv, float* depth_values, float* near_far_out, enerf_stream_t stream);

/* ---- homo_warp + build_feature_volume (utils.py:57-95, 322-349), fused.
 * feat (B,S,Hs,Ws,C) channels-last, C in {8,16,32}; vol (B,D,h,w,C). ---- */
int enerf_build_feature_volume(const float* feat, const float* proj, const float* depth_values, int B, int S, int C,
                               int Hs, int Ws, int D, int h, int w, float* vol, enerf_stream_t stream);

/* ---- CostRegNet / MinCostRegNet (cost_reg_net.py:4-86) ---- */
typedef struct {
    enerf_conv_bn_t conv[12]; /* conv0..conv6, conv7, conv9, conv11 at their own index; others unused */
    const float* feat_conv_w; /* (8,8,3,3,3) */
    const float* depth_conv_w; /* (1,8,3,3,3) */
    int in_channels;          /* 32 (level 0) or 16 (level 1) */
    int full;                 /* 1 = CostRegNet (3 down/3 up), 0 = MinCostRegNet (2/2) */
} enerf_costreg_raw_t;
long long enerf_cost_reg_packed_floats(int in_channels, int full);
int enerf_cost_reg_pack(const enerf_costreg_raw_t* raw, float* packed, enerf_stream_t stream);
size_t enerf_cost_reg_workspace_bytes(int full, int B, int D, int h, int w);
/* vol (B,D,h,w,in_channels) -> feat (B,D,h,w,8), prob (B,D,h,w).  D,h,w divisible by 4 (8 if full). */
int enerf_cost_reg(const float* packed, int in_channels, int full, const float* vol, int B, int D, int h, int w,
                   float* feat, float* prob, void* workspace, size_t workspace_bytes, const enerf_options_t* options,
                   enerf_stream_t stream);

/* ---- depth_regression (utils.py:658-667) ---- */
int enerf_depth_regression(const float* prob, const float* depth_values, int B, int D, int h, int w, int depth_inv,
                           float* depth, float* std, enerf_stream_t stream);

/* ---- build_rays (utils.py:390-420): rays8 (B,N,8) + maps at (h,w) -> rays12 (B,N,12); (Hr,Wr) is the
 * render resolution the maps are upsampled to. ---- */
int enerf_build_rays(const float* rays8, const float* depth, const float* std, const float* near_far, int B, int N,
                     int h, int w, int Hr, int Wr, int depth_inv, float* rays12, enerf_stream_t stream);

/* ---- NeRF + Agg MLP weights (nerf.py:6-89), torch nn.Linear layouts (out,in) ---- */
typedef struct {
    const float *view_w, *view_b;   /* agg.view_fc.0    (F,4),(F)   — ignored when viewdir_agg == 0 */
    const float *glob_w, *glob_b;   /* agg.global_fc.0  (32,3F),(32) */
    const float *aggw_w, *aggw_b;   /* agg.agg_w_fc.0   (1,32),(1) */
    const float *fc_w, *fc_b;       /* agg.fc.0         (16,32),(16) */
    const float *lr0_w, *lr0_b;     /* lr0.0            (64,24),(64) */
    const float *sigma_w, *sigma_b; /* sigma.0          (1,64),(1) */
    const float *col0_w, *col0_b;   /* color.0          (64,88+F+4),(64) */
    const float *col2_w, *col2_b;   /* color.2          (1,64),(1) */
} enerf_nerf_raw_t;
long long enerf_nerf_packed_floats(int F); /* F = feat_ch + 3 */
int enerf_nerf_pack(const enerf_nerf_raw_t* raw, int F, int viewdir_agg, float* packed, enerf_stream_t stream);

/* ---- render_rays (network.py:24-43) = sample_along_depth + get_vox_feat + get_img_feat + NeRF +
 * raw2outputs, fused.  F in {11, 35}; S in {2,3,4}; n_samples in [1,8]. ---- */
typedef struct {
    const float* rays12;   /* (B,N,12) */
    const float* tex;      /* (B,S,Hr,Wr,TEX) from enerf_pack_img_feat_rgb */
    const float* vol;      /* (B,D,h,w,8) feature volume from enerf_cost_reg */
    const float* src_exts; /* (B,S,4,4) */
    const float* src_ixts; /* (B,S,3,3) */
    const float* tar_ext;  /* (B,4,4) */
    const float* packed;   /* enerf_nerf_pack output */
    float* rgb;            /* (B,N,3) */
    float* depth;          /* (B,N) */
    float* weights;        /* (B,N,n_samples) */
    int B, N, S, n_samples, depth_inv, Hr, Wr, F, D, h, w, white_bkgd;
    float render_scale;
    /* Optional fused build_rays (utils.py:390-420): when rays8 != NULL, rays12 is ignored and every ray's
     * [near, far | volume near, far] is derived inside the kernel from the 8-float rays and the level's
     * depth / std (B,map_h,map_w) and near_far (B,2,map_h,map_w) maps, exactly as enerf_build_rays does
     * (same device function) — one launch and a 48 B/ray round trip less.  ABI >= 2. */
    const float* rays8;
    const float *depth_map, *std_map, *nf_map;
    int map_h, map_w;
    const enerf_options_t* options; /* NULL = defaults (ABI >= 3) */
    /* Optional device-side ray selection (network_human.py:90-93,102-107; ABI >= 3, B must be 1): when ray_index != NULL
     * the kernel renders rays ray_index[0 .. *ray_count) of the (N) ray list — *ray_count is read ON THE DEVICE, so a
     * data-dependent mask never forces a host sync — writes depth/weights compacted (row r <- ray ray_index[r], rows
     * >= *ray_count untouched) and, with scatter_rgb != 0, writes rgb to row ray_index[r] of the (N,3) buffer the caller
     * zeroed (only if *ray_count > 1: the reference's `mask.sum() > 1` quirk), else compacted like depth. */
    const int* ray_index;
    const int* ray_count;
    int scatter_rgb;
    int max_blocks;        /* ABI >= 10: > 0 caps the launch at this many persistent blocks (a render that overlaps other work of the
                              frame — enerf_forward's forked non-final level — leaves the remaining compute units to that work);
                              0 = as many blocks as the device holds */
} enerf_render_args_t;
int enerf_render_rays(const enerf_render_args_t* args, enerf_stream_t stream);

/* ---- mask_at_box -> ray index list (network_human.py:90-93: rays[mask_at_box]), stable order, no host sync.
 * mask: n elements of elem_bytes in {1,2,4,8} (bool/uint8, int16, int32/float32, int64); an element selects its ray when
 * any of its bytes is non-zero (.bool()).  index (n) int32: the first *count entries are the selected positions in
 * ascending order; count (1) int32 on the device.  workspace: enerf_mask_compact_workspace_bytes(n). ---- */
size_t enerf_mask_compact_workspace_bytes(long long n);
int enerf_mask_compact(const void* mask, int elem_bytes, long long n, int* index, int* count, void* workspace,
                       size_t workspace_bytes, enerf_stream_t stream);

/* ---- Network.forward (network.py:76-113 / network_human.py:69-119): the WHOLE cascade in one call. ----
 * FeatureNet -> for every level {get_proj_mats + get_depth_values, homo_warp + variance, cost regularisation,
 * depth_regression, [build_rays + render_rays]} enqueued on `stream` from one C call (a non-Python host does not have to
 * re-implement the cascade loop; the Python binding's forward() is this call).  All buffers are caller-owned; scratch
 * comes from `workspace` (enerf_forward_workspace_bytes).  Nothing synchronises. */
#define ENERF_MAX_LEVELS 3
typedef struct {                               /* cfg.enerf.cas_config (configs/enerf/dtu_pretrain.yaml:27-43) */
    int num;
    int depth_inv[ENERF_MAX_LEVELS];
    double volume_scale[ENERF_MAX_LEVELS];
    int volume_planes[ENERF_MAX_LEVELS];
    double im_feat_scale[ENERF_MAX_LEVELS];
    double im_ibr_scale[ENERF_MAX_LEVELS];
    double render_scale[ENERF_MAX_LEVELS];
    int render_im_feat_level[ENERF_MAX_LEVELS];
    int nerf_model_feat_ch[ENERF_MAX_LEVELS];
    int render_if[ENERF_MAX_LEVELS];
    int num_samples[ENERF_MAX_LEVELS];
    int white_bkgd;                            /* cfg.enerf.white_bkgd (network.py:42) */
} enerf_cascade_t;
/* stage_events slots: after each stage of the frame the driver records the caller's hipEvent_t (if non-NULL) on `stream`
 * — per-stage timings without leaving the single call.  Slot = ENERF_STAGE_FEATURE_NET, or
 * ENERF_STAGE_LEVEL(level, ENERF_STAGE_{PREP,VOLUME,COST_REG,DEPTH_REG,TEXELS,RENDER}). */
enum { ENERF_STAGE_BEGIN = 0, ENERF_STAGE_FEATURE_NET = 1, ENERF_STAGE_PREP = 0, ENERF_STAGE_VOLUME = 1,
       ENERF_STAGE_COST_REG = 2, ENERF_STAGE_DEPTH_REG = 3, ENERF_STAGE_TEXELS = 4, ENERF_STAGE_RENDER = 5,
       ENERF_STAGES_PER_LEVEL = 6, ENERF_STAGE_COUNT = 2 + 6 * ENERF_MAX_LEVELS };
#define ENERF_STAGE_LEVEL(level, k) (2 + (level) * ENERF_STAGES_PER_LEVEL + (k))
typedef struct {
    /* the batch dict (lib/datasets/dtu/enerf.py:100-119) */
    const float* src_inps;                     /* (B,S,3,H,W) in [-1,1] */
    const float* src_exts;                     /* (B,S,4,4) world->camera */
    const float* src_ixts;                     /* (B,S,3,3) */
    const float* tar_ext;                      /* (B,4,4) */
    const float* tar_ixt;                      /* (B,3,3) */
    const float* near_far;                     /* (B,2) */
    const float* rays[ENERF_MAX_LEVELS];       /* rays_{i} (B,n_rays[i],8); NULL for a rendered level = the full image at
                                                  render_scale[i], generated on the device (enerf_utils.py:61-71) */
    int n_rays[ENERF_MAX_LEVELS];
    const void* mask_at_box;                   /* network_human.py:90-107 (last level, B == 1), (H*W) elements, or NULL */
    int mask_elem_bytes;
    int B, S, H, W;
    enerf_cascade_t cas;
    /* weight images (enerf_*_pack) */
    const float* feature_net_packed;           /* may be NULL when feats_nchw is given */
    const float* cost_reg_packed[ENERF_MAX_LEVELS];
    const float* nerf_packed[ENERF_MAX_LEVELS];/* rendered levels only */
    /* north_star's split: a FeatureNet run elsewhere (PyTorch-ROCm) hands over its NCHW maps
     * feats_nchw[l] = feats['level_l'] (B*S,C_l,H_l,W_l), all three or none */
    const float* feats_nchw[3];
    /* outputs of every rendered level i (network.py:105-112); pointers of non-rendered levels are ignored */
    float* rgb[ENERF_MAX_LEVELS];              /* (B,N_i,3); masked level: (1,H*W,3), zero outside the mask */
    float* depth[ENERF_MAX_LEVELS];            /* (B,N_i);   masked level: first *ray_count rows */
    float* weights[ENERF_MAX_LEVELS];          /* (B,N_i,num_samples[i]); masked level: first *ray_count rows */
    float* depth_mvs[ENERF_MAX_LEVELS];        /* (B,h_i,w_i) */
    float* std[ENERF_MAX_LEVELS];              /* (B,h_i,w_i) */
    /* masked level: selected ray positions / their number, on the device.  If ray_index is NULL they are computed into
     * the workspace; a caller that wants the count on the host passes its own buffers (see enerf_mask_compact). */
    int* ray_index;                            /* (H*W) int32 or NULL */
    int* ray_count;                            /* (1) int32 or NULL */
    int ray_index_ready;                       /* 1: the caller already ran enerf_mask_compact into ray_index/ray_count */
    void* workspace;
    size_t workspace_bytes;
    const enerf_options_t* options;
    void* const* stage_events;                 /* ENERF_STAGE_COUNT hipEvent_t slots or NULL */
} enerf_frame_args_t;
size_t enerf_forward_workspace_bytes(const enerf_frame_args_t* args);   /* 0 + enerf_last_error() on invalid arguments */
int enerf_forward(const enerf_frame_args_t* args, enerf_stream_t stream);

/* ---- backward kernels of the training path (enerf_amd/autograd.py wraps them as
 * torch.autograd.Functions).  First batch: the stages around the dense layers.
 *   enerf_build_feature_volume_bwd  homo_warp + variance (utils.py:57-95,322-349): grad_vol (B,D,h,w,C) ->
 *       grad_feat (B,S,Hs,Ws,C) (zeroed here, then scatter-added) and grad_depth_values (B,D,h,w) (through the warp grid).
 *   enerf_depth_regression_bwd      utils.py:658-667: grad_depth, grad_std (B,h,w) -> grad_prob, grad_depth_values (B,D,h,w).
 *   enerf_composite / _bwd          raw2outputs (utils.py:571-603): raw (n,Ns,4) = [rgb, sigma], z (n,Ns) ->
 *       rgb (n,3), depth (n), weights (n,Ns); backward -> grad_raw (n,Ns,4), grad_z (n,Ns). ---- */
/*   enerf_conv_wgrad  weight gradient of every convolution of the path on the matrix cores:
 *       grad_w[a][b][kd][kh][kw] = sum over positions o of the A grid of A[a][o] * B[b][o*stride + k - pad]
 *       a_cl (n, Da*Ha*Wa, Ca) and b_cl (n, Db*Hb*Wb, Cb) channels-last.  Conv{2,3}d: A = grad_output, B = input ->
 *       (Cout,Cin,k..); ConvTranspose3d(k3,s2,p1,op1): A = input, B = grad_output -> (Cin,Cout,k..).  2-D layers pass
 *       Da = Db = kd = 1.  Kernels 3x3x3, 1x3x3, 1x5x5, 1x1x1.
 *       workspace (ABI v6): enerf_conv_wgrad_workspace_bytes(positions of the A grid incl. n, Ca, Cb, kd, kh, kw) bytes of
 *       scratch for the two-stage commit (every block stores its partial tiles, a second kernel sums them: deterministic,
 *       no atomics).  NULL / too small: grad_w is zeroed and the blocks add into it with fp32 atomics (correct, several
 *       times slower: ~1000 blocks x 6912 atomics onto 6912 addresses). */
size_t enerf_conv_wgrad_workspace_bytes(long long positions_a, int Ca, int Cb, int kd, int kh, int kw);
int enerf_conv_wgrad(const float* a_cl, const float* b_cl, int n, int Da, int Ha, int Wa, int Ca, int Db, int Hb, int Wb, int Cb,
                     int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w, float* grad_w, void* workspace,
                     size_t workspace_bytes, enerf_stream_t stream);
/*   Training-mode cost-regularisation layers (ConvBnReLU3D utils.py:22-33, cost_reg_net.py) — BatchNorm uses batch
 *   statistics, so it cannot be folded into the convolution:
 *   enerf_conv3d_layer[_pack]  one bias-free 3x3x3 layer on the inference path's MFMA kernels, identity epilogue (+ optional
 *       residual).  kind 0 = stride 1, 1 = stride 2 (weights (cout,cin,3,3,3)), 2 = transposed stride 2 (weights
 *       (cin,cout,3,3,3), output 2x).  The same entry computes INPUT gradients: dgrad(stride-2 conv, w) = kind 2 on w,
 *       dgrad(transposed, w) = kind 1 on w, dgrad(stride-1, w) = kind 0 on w flipped and channel-transposed.
 *   enerf_channel_sums    sums[c] = sum_p a*m, sums[C+c] = sum_p a*m*b  (fp64), m = (z_mask*mask_scale[c]+mask_shift[c] > 0)
 *       or 1: BN batch statistics (a = b = z) and the d beta / d gamma reductions of its backward (a = grad, b = z).
 *   enerf_channel_affine  out = f(a*m*p[c] + b*q[c] + r[c]) (+ residual), f = ReLU if relu: BN normalise + ReLU + skip add,
 *       and the input gradient of BN.  All tensors channels-last (n, C). */
/*   enerf_conv2d_layer[_pack]  (ABI v6) one FeatureNet convolution (feature_net.py:7-22: k 1/3/5, stride 1/2, padding (k-1)/2,
 *       weights (cout,cin,k,k), optional bias) on the inference path's MFMA kernel with an identity epilogue, for the
 *       training-mode FeatureNet.  in: channels-last (N,Hi,Wi,cin) — for cin = 3 the NCHW image batch (N,3,Hi,Wi) as the
 *       batch holds it; out: channels-last (N,Ho,Wo,cout); up (optional): a channels-last (N,Ho/2,Wo/2,cout) map that is
 *       upsampled 2x (bilinear, align_corners) and added (feature_net.py:24-25).  Supported (cin,cout,k,stride): the eleven
 *       FeatureNet layers and the input gradients of its stride-1 layers (dgrad = the stride-1 conv cout -> cin on the
 *       flipped, channel-transposed weights): 16->32 k3, 8->32 k3, 32->16 k1, 32->8 k1. */
/*   enerf_up2_adjoint  (ABI v6) adjoint of the top-down 2x upsampling (feature_net.py:24-25: bilinear, align_corners=True):
 *       grad_fine (N,2Hc,2Wc,C) channels-last -> grad_coarse (N,Hc,Wc,C) (+ add, an optional second gradient of the coarse
 *       map), gather form with the forward's weights — no atomics. */
int enerf_up2_adjoint(const float* grad_fine, const float* add, int N, int Hc, int Wc, int C, float* grad_coarse, enerf_stream_t stream);
long long enerf_conv2d_layer_packed_floats(int cin, int cout, int k);
int enerf_conv2d_layer_pack(const float* w, const float* bias, int cin, int cout, int k, float* packed, enerf_stream_t stream);
int enerf_conv2d_layer(const float* packed, int cin, int cout, int k, int stride, const float* in, const float* up, float* out, int N,
                       int Hi, int Wi, enerf_stream_t stream);
long long enerf_conv3d_layer_packed_floats(int cin, int cout, int kind);
int enerf_conv3d_layer_pack(const float* w, int cin, int cout, int kind, float* packed, enerf_stream_t stream);
int enerf_conv3d_layer(const float* packed, int cin, int cout, int kind, const float* in, const float* residual, float* out, int B,
                       int Di, int Hi, int Wi, const enerf_options_t* options, enerf_stream_t stream);
int enerf_channel_sums(const float* a, const float* b, const float* z_mask, const float* mask_scale, const float* mask_shift,
                       long long n, int C, double* sums, enerf_stream_t stream);
/*   the same with caller-provided scratch (ABI v7): the blocks store partial rows and a second launch adds them — no atomics (the
 *   2C fp64 atomics per block of the form above serialise on their 2C addresses: 21 us for 256 blocks at C = 32), no zeroing
 *   launch, deterministic.  workspace == NULL is the form above. */
size_t enerf_channel_sums_workspace_bytes(long long n, int C);
int enerf_channel_sums_ws(const float* a, const float* b, const float* z_mask, const float* mask_scale, const float* mask_shift,
                          long long n, int C, double* sums, void* workspace, size_t workspace_bytes, enerf_stream_t stream);
/*   enerf_bn_train_coeffs / enerf_bn_train_bwd_coeffs  (ABI v6) the C-sized arithmetic of a training-mode BatchNorm between the
 *       statistics kernel (and, under SyncBatchNorm, its all-reduce) and the affine kernel, one launch each, fp64:
 *       forward  sums = [sum z, sum z^2] (2C), position count (device scalar count_dev, or count_host when NULL) ->
 *                mean_invstd (2C fp64), scale_shift (2C: gamma*invstd, beta - mean*gamma*invstd); running_mean/var (optional)
 *                updated in place with the unbiased variance, momentum < 0 = cumulative average 1/num_batches_tracked
 *                (int64 device scalar; increment_num_batches_tracked = 1 (ABI v7): this launch adds the batch to it first,
 *                0: the caller already did);
 *       backward sums_local / sums_global = [sum g*m, sum g*m*z] of this rank / of all ranks (the same pointer without
 *                SyncBatchNorm) -> dgamma_dbeta (2C, from the local sums: DDP averages parameter gradients) and k2k3 (2C):
 *                d z = g*m*scale + z*k2 + k3 (enerf_channel_affine). */
int enerf_bn_train_coeffs(const double* sums, const double* count_dev, double count_host, const float* gamma, const float* beta,
                          double eps, double momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                          int increment_num_batches_tracked, int C, double* mean_invstd, float* scale_shift, enerf_stream_t stream);
int enerf_bn_train_bwd_coeffs(const double* sums_local, const double* sums_global, const double* count_dev, double count_host,
                              const double* mean_invstd, const float* scale, int C, float* dgamma_dbeta, float* k2k3,
                              enerf_stream_t stream);
/*   enerf_bn_train_stats / enerf_bn_train_bwd_stats  (ABI v8) the statistics AND the coefficients of one training-mode BatchNorm
 *       direction in TWO launches (block partial sums; their reduction + the coefficient arithmetic) instead of three — the
 *       path without SyncBatchNorm, where nothing sits between the sums and the coefficients.  Arguments as
 *       enerf_channel_sums_ws (workspace: enerf_channel_sums_workspace_bytes) + enerf_bn_train_coeffs / _bwd_coeffs with the
 *       position count n as a host number; forward: sums over z of [z, z^2] (sums_out optional, 2C fp64); backward: sums of
 *       [g*m, g*m*z].  Bit-identical to the three-launch sequence.  (utils.py:10-33: BatchNorm2d/3d in .train()) */
int enerf_bn_train_stats(const float* z, long long n, int C, void* workspace, size_t workspace_bytes, const float* gamma,
                         const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                         long long* num_batches_tracked, int increment_num_batches_tracked, double* sums_out, double* mean_invstd,
                         float* scale_shift, enerf_stream_t stream);
int enerf_bn_train_bwd_stats(const float* g, const float* z, const float* z_mask, const float* mask_scale, const float* mask_shift,
                             long long n, int C, void* workspace, size_t workspace_bytes, const double* mean_invstd,
                             const float* scale, float* dgamma_dbeta, float* k2k3, enerf_stream_t stream);
/* ABI v11: a whole BatchNorm direction in TWO launches where the layer is small enough (partial rows x channels <= 2048: the layers below
 * ~8 MB, 13 of the 23 of a dtu_pretrain step), three otherwise: the affine kernel's blocks redo the row reduction and the coefficient
 * arithmetic of enerf_bn_train[_bwd]_stats in their prologue (same additions, same order: identical coefficients) instead of a
 * one-block launch in between.  forward: out = [relu](z * scale + shift) [+ residual], plus everything enerf_bn_train_stats leaves
 * (mean_invstd, scale_shift, running statistics).  backward: grad_z = g m scale + z k2 + k3 with m the ReLU mask (relu != 0), plus
 * dgamma_dbeta.  workspace: enerf_channel_sums_workspace_bytes(n, C) (+ 2 C floats for the backward). */
int enerf_bn_train_apply(const float* z, long long n, int C, void* workspace, size_t workspace_bytes, const float* gamma, const float* beta,
                         double eps, double momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                         int increment_num_batches_tracked, double* mean_invstd, float* scale_shift, const float* residual /* nullable */,
                         int relu, float* out, enerf_stream_t stream);
int enerf_bn_train_bwd_apply(const float* g, const float* z, int relu, long long n, int C, void* workspace, size_t workspace_bytes,
                             const double* mean_invstd, const float* scale_shift, float* dgamma_dbeta, float* grad_z, enerf_stream_t stream);
int enerf_channel_affine(const float* a, const float* b, const float* p, const float* q, const float* r, const float* z_mask,
                         const float* mask_scale, const float* mask_shift, const float* residual, int relu, long long n, int C,
                         float* out, enerf_stream_t stream);
/*   Agg + NeRF MLP backward (nerf.py:29-89), fused per point: recomputes the forward in registers (same weight image and MFMA
 *   operand chaining as enerf_render_rays) and back-propagates with the transposed weights as MFMA A operands.  Writes the
 *   input gradients and, per layer, {pre-activation gradient, layer input} as channels-last rows; enerf_gemm_wgrad
 *   (grad_w[a][b] = sum_p A[p][a] B[p][b] on the matrix cores) turns those into the weight gradients.
 *   vox (P,8), x (P,S,F+4) = [features F | direction code 4], g_raw (P,4) = d loss / d [rgb, sigma]; packed = enerf_nerf_pack
 *   image; bimg + image_offsets = the transposed-weight images (enerf_amd/autograd.py: mlp_backward_images);
 *   save[16] = hv (P,88) G (P,32) q (P,S,64) g (P,S,32) a (P,S,F) varmean (P,2F) | d_cpre (P,S) d_qpre (P,S,64) d_p2 (P,64)
 *   d_spre (P) d_hpre (P,64) d_aggpre (P,16) d_upre (P,S) d_gpre (P,S,32) d_gsum (P,32) d_vpre (P,S,F). */
typedef struct {
    const float *vox, *x, *g_raw, *packed, *bimg;
    float *g_vox, *g_x;
    float* save[16];
    long long P;
    int F, S;
    int image_offsets[8];
} enerf_mlp_bwd_args_t;
int enerf_nerf_mlp_bwd(const enerf_mlp_bwd_args_t* args, enerf_stream_t stream);
/* ABI v11, F = 11: the same backward with the weight gradients of the per-(point, view) layers accumulated INSIDE the kernel, in registers
 * for its whole run, instead of through saved rows:
 *   level 1: the colour branch — color.0's per-view columns (nerf.py:64-66: 64 x (F + 4), a 16-point product on the matrix cores per view) and
 *            color.2 (+ bias); q, d_qpre, d_cpre (save[2], save[7], save[6]: 40 % of the saved bytes) are never written and may be NULL;
 *   level 2 (S <= 3): also the aggregation branch — global_fc's `a` columns (32 x F), agg_w_fc (+ bias), view_fc (+ bias) (nerf.py:29-52);
 *            g, a, d_upre, d_gpre, d_vpre (save[3], save[4], save[12], save[13], save[15]: another 27 %) may be NULL too.
 * Every wave writes one row of partial sums: tiles wg_q[chunks][4][256], wg_g[chunks][2][256], wg_v[chunks][256] (what a member with
 * `partials` of enerf_gemm_wgrad_group reduces: 64 x (F + 4), 32 x F, F x 4) and wg_rows[chunks][128] = [color.2 weight 64 | bias | 0 x 15 |
 * agg_w_fc weight 32 | bias | view_fc bias F | 0 ...] (enerf_colsum); chunks = enerf_nerf_mlp_bwd_chunks(P).  wg_g / wg_v: level 2 only. */
long long enerf_nerf_mlp_bwd_chunks(long long P);
int enerf_nerf_mlp_bwd_partials(const enerf_mlp_bwd_args_t* args, int level, float* wg_q, float* wg_g, float* wg_v, float* wg_rows,
                                enerf_stream_t stream);
/* out[i] = sum over the chunks c of part[c * n + i] (fixed order): the second stage of a per-wave partial-sum layout. */
int enerf_colsum(const float* part, int chunks, int n, float* out, enerf_stream_t stream);
/* forward of the same MLP on materialised inputs (training): raw (P,4) = [rgb, sigma]; nothing else is written. */
int enerf_nerf_mlp_fwd(const float* vox, const float* x, const float* packed, long long P, int S, int F, float* raw,
                       enerf_stream_t stream);
/* Render-side feature fetches of the TRAINING path, forward and backward (utils.py:689-722 get_img_feat + utils.py:456-458
 * get_vox_feat; inference does them inside enerf_render_rays).  Per batch element b and point p (P points each):
 *   xyz (B,P,3) world position, dn (B,P) normalised depth coordinate, uv (B,P,2) ray pixel coordinates at the level's
 *   resolution; tex (B,S,Hr,Wr,F) channels-last [features | rgb]; vol (B,D,h,w,8) channels-last feature volume;
 *   cam (B,S,16) = K'E[:3,:3] (9, row-major) | K't (3) | source camera centre (3) | 0, with K' = K scaled to the level;
 *   tcen (B,4) target camera centre.
 * fwd writes x (B,P,S,F+4) = [bilinear(border) texel | direction code] and vox (B,P,8) = trilinear(zeros) volume sample.
 * bwd reads g_x, g_vox and writes g_tex, g_vol (zeroed here, scatter-added), g_xyz (B,P,3) and g_dn (B,P). */
typedef struct {
    const float *xyz, *dn, *uv, *tex, *vol, *cam, *tcen;
    float *x, *vox;
    const float *g_x, *g_vox;
    float *g_tex, *g_vol, *g_xyz, *g_dn;
    long long P;
    int B, S, F, Hr, Wr, D, h, w;
    /* optional hints for enerf_gather_bwd (0 = none): the points are n_samples consecutive samples of each ray and the rays are
     * a row-major raster of ray_w rays per row (config 5 trains on full images, dtu_pretrain.yaml:41): p = (y * ray_w + x) *
     * n_samples + k.  With both set, a block owns a 2-D ray tile and accumulates its texel / volume scatter in LDS patches that
     * are flushed once (fp32 atomics cost one L2 request per 64-byte line and instruction, tools/micro/atomic_rate.hip); taps
     * outside a patch fall back to global atomics, so a wrong hint costs time, never correctness. */
    int ray_w, n_samples;
} enerf_gather_args_t;
int enerf_gather_fwd(const enerf_gather_args_t* args, enerf_stream_t stream);
int enerf_gather_bwd(const enerf_gather_args_t* args, enerf_stream_t stream);
size_t enerf_gemm_wgrad_workspace_bytes(long long P, int Ca, int Cb, int with_bias);   /* scratch of the two-stage commit, as above */
int enerf_gemm_wgrad(const float* a, int lda, int Ca, const float* b, int ldb, int Cb, long long P, float* grad_w,
                     float* grad_bias /* nullable: (Ca) = sum_p a[p][:] from the same pass */, void* workspace,
                     size_t workspace_bytes, enerf_stream_t stream);
/* ABI v11: several of those GEMMs in two to four launches (the members' grids concatenated per register class, then one reduction kernel
 * for all): the weight
 * gradients of one Agg + NeRF MLP (nerf.py:29-89) are ten to eleven position reductions over the rows enerf_nerf_mlp_bwd saved.  Every
 * member keeps the block count, row map and summation order of its single enerf_gemm_wgrad call: bit-identical gradients.  ldw: row
 * stride of grad_w in floats (0 = Cb) — a gradient that is a column block of a wider weight matrix (color.0 = [shared | per-view]
 * columns, nerf.py:64-66) is written in place.  At most 16 members of at most 4 x 6 tiles (64 x 95 columns); the workspace
 * (enerf_gemm_wgrad_group_workspace_bytes) is required. */
/* ABI v11: deferred second stages.  Every enerf_conv_wgrad / enerf_gemm_wgrad with a workspace is two launches: the blocks' partial sums,
 * then a small reduction into grad_w (30 of those per dtu_pretrain step).  Between enerf_wgrad_reduce_begin() and
 * enerf_wgrad_reduce_flush(stream) ON THE CALLING THREAD the reductions are recorded instead of launched and the flush runs them as one
 * kernel (same code per gradient: same bits); the caller keeps every workspace and grad_w alive and unread until the flush.  At most 96
 * recorded reductions (further ones are launched at once). */
int enerf_wgrad_reduce_begin(void);
int enerf_wgrad_reduce_flush(enerf_stream_t stream);
typedef struct {
    const float* a; int lda, Ca;
    const float* b; int ldb, Cb;
    long long P;
    float* grad_w; int ldw;
    float* grad_bias;                       /* nullable */
    const float* partials;                  /* nullable.  non-NULL: the member's first stage already ran elsewhere (enerf_nerf_mlp_bwd_partials): */
    int partial_chunks;                     /* partials[partial_chunks][ceil(Ca/16) * ceil(Cb/16)][256] are only reduced; a, b, P are ignored */
} enerf_gemm_wgrad_desc_t;
size_t enerf_gemm_wgrad_group_workspace_bytes(const enerf_gemm_wgrad_desc_t* descs, int n);
int enerf_gemm_wgrad_group(const enerf_gemm_wgrad_desc_t* descs, int n, void* workspace, size_t workspace_bytes, enerf_stream_t stream);
int enerf_build_feature_volume_bwd(const float* feat, const float* proj, const float* depth_values, const float* grad_vol, int B,
                                   int S, int C, int Hs, int Ws, int D, int h, int w, float* grad_feat, float* grad_depth_values,
                                   enerf_stream_t stream);
int enerf_depth_regression_bwd(const float* prob, const float* depth_values, const float* grad_depth, const float* grad_std, int B,
                               int D, int h, int w, int depth_inv, float* grad_prob, float* grad_depth_values,
                               enerf_stream_t stream);
int enerf_composite(const float* raw, const float* z, long long n, int n_samples, int white_bkgd, float* rgb, float* depth,
                    float* weights, enerf_stream_t stream);
int enerf_composite_bwd(const float* raw, const float* z, const float* grad_rgb, const float* grad_depth, const float* grad_weights,
                        long long n, int n_samples, float* grad_raw, float* grad_z, enerf_stream_t stream);

/* ---- ABI v7: the rest of the training step: what was still eager PyTorch after ABI v6 ----
 *   enerf_conv2d_s2k5_dgrad   input gradient of Conv2d(cin->cout, k5, s2, p2) (feature_net.py:11,14; 8->16 and 16->32):
 *       w (cout,cin,5,5) torch layout, dz (N,Ho,Wo,cout) channels-last -> gx (N,2Ho,2Wo,cin) (+ add, optional, same shape).
 *       Four output-parity classes = one stride-1 3x3 launch of the inference MFMA kernel with 4*cin output channels + a
 *       depth-to-space pass; scratch from the caller (enerf_conv2d_s2k5_dgrad_workspace_bytes).
 *   enerf_resize_ac_adjoint   adjoint of F.interpolate(bilinear, align_corners=True) on planar maps (n_maps,Hf,Wf) ->
 *       (n_maps,Hc,Wc) (+ add), gather form, any scale >= 1 (utils.py:115-117, 394-396).
 *   enerf_get_depth_values_bwd  get_depth_values (utils.py:98-151), level > 0: grad_dv (B,D,h,w) -> grad_depth / grad_std
 *       (the two halves of ONE (2,B,hp,wp) buffer) of the previous level's maps; clamped entries (utils.py:122-127) carry no
 *       gradient, near_far is detached (utils.py:148).  scratch: 2*B*h*w floats.
 *   enerf_ray_samples_fwd/bwd   build_rays + sample_along_depth (utils.py:390-441): rays8 (B,N,8), depth/std (B,h,w),
 *       near_far (B,2,h,w) -> z (B,N,Ns), xyz (B,N,Ns,3), dn (B,N,Ns), uv (B,N,Ns,2), rays12 (B,N,12; optional).
 *       bwd: grad_xyz, grad_dn -> grad_depth, grad_std (B,h,w) (zeroed here, scatter-added with the rays' bilinear taps).
 *   enerf_camera_tables        the per-view constants of enerf_gather_* (utils.py:697-704, 712-715): cam (B,S,16) =
 *       K'E[:3,:3] | K't | source camera centre | 0, tcen (B,4); fp64 products / 4x4 inverses on the device.
 *   enerf_weights_flip_transpose  w (cout,cin,taps) -> (cin,cout,taps) with the taps reversed (dgrad weights, stride 1).
 *   enerf_concat2_pad          out (n) = [a (na) | b (nb) | 0]  (feat_conv ++ depth_conv ++ zero rows).
 *   enerf_pack_texels_train    tex (n,Hr,Wr,C+3) = [feat_cl (n,Hr,Wr,C) | bilinear_ac(src*0.5+0.5) (3)] (network.py:28-33).
 *   enerf_slice_channels       dst (n,C) = src (n,F)[:, c0:c0+C].
 *   enerf_concat_channels      out (n,C) = [a (n,Ca) | b (n,Cb) | 0]  (the fused heads' gradient from d feat and d prob).
 *   enerf_gather_images        out[i] = idx[i] >= 0 ? srcs[which[i]][idx[i]] : 0 over <= 64 source tensors (the transposed-
 *       weight images of enerf_nerf_mlp_bwd, and every packed convolution-weight image of a network's training step, in one launch each).
 *   enerf_add                  out = a + b.
 *   enerf_cast_f64_f32         out (n) fp32 = in (n) fp64 (bias gradients come out of enerf_channel_sums in fp64).
 *   enerf_reciprocal           out = 1 / x  (depth_mvs of a disparity-space level, network.py:105-108).
 *   enerf_composite_bwd        (changed) grad_rgb / grad_depth / grad_weights may be NULL = zeros (outputs the loss ignores). */
size_t enerf_conv2d_s2k5_dgrad_workspace_bytes(int cin, int cout, int N, int Ho, int Wo);
/* (ABI v9) the same in two halves, for a caller that prepares the step's weight images itself: _pack writes the parts' packed
 * sub-kernel images (enerf_conv2d_s2k5_dgrad_packed_floats floats; w3_scratch: 4*cin*cout*9 floats), _packed runs on them. */
long long enerf_conv2d_s2k5_dgrad_packed_floats(int cin, int cout);
int enerf_conv2d_s2k5_dgrad_pack(const float* w, int cin, int cout, float* w3_scratch, float* packed, enerf_stream_t stream);
int enerf_conv2d_s2k5_dgrad_packed(const float* packed, int cin, int cout, const float* dz, const float* add, float* gx, int N, int Ho,
                                   int Wo, void* workspace, size_t workspace_bytes, enerf_stream_t stream);
int enerf_conv2d_s2k5_dgrad(const float* w, int cin, int cout, const float* dz, const float* add, float* gx, int N, int Ho, int Wo,
                            void* workspace, size_t workspace_bytes, enerf_stream_t stream);
int enerf_resize_ac_adjoint(const float* grad_fine, const float* add, int n_maps, int Hf, int Wf, int Hc, int Wc, float* grad_coarse,
                            enerf_stream_t stream);
int enerf_get_depth_values_bwd(const float* prev_depth, const float* prev_std, const float* prev_near_far, const float* grad_dv, int B,
                               int D, int h, int w, int hp, int wp, int depth_inv, float* grad_depth, float* grad_std, float* scratch,
                               enerf_stream_t stream);
int enerf_ray_samples_fwd(const float* rays8, const float* depth, const float* std, const float* near_far, int B, int N, int n_samples,
                          int h, int w, int Hr, int Wr, int depth_inv, float* z, float* xyz, float* dn, float* uv, float* rays12,
                          enerf_stream_t stream);
int enerf_ray_samples_bwd(const float* rays8, const float* depth, const float* std, const float* near_far, const float* grad_xyz,
                          const float* grad_dn, int B, int N, int n_samples, int h, int w, int Hr, int Wr, int depth_inv,
                          float* grad_depth, float* grad_std, enerf_stream_t stream);
int enerf_camera_tables(const float* src_ixts, const float* src_exts, const float* tar_ext, int B, int S, float render_scale, float* cam,
                        float* tcen, enerf_stream_t stream);
int enerf_weights_flip_transpose(const float* w, int cout, int cin, int taps, float* out, enerf_stream_t stream);
int enerf_concat2_pad(const float* a, long long na, const float* b, long long nb, long long n, float* out, enerf_stream_t stream);
int enerf_pack_texels_train(const float* feat_cl, int C, const float* src_inps, int H, int W, int Hr, int Wr, int n_img, float* tex,
                            enerf_stream_t stream);
int enerf_slice_channels(const float* src, long long n, int F, int c0, int C, float* dst, enerf_stream_t stream);
int enerf_concat_channels(const float* a, int Ca, const float* b, int Cb, long long n, int C, float* out, enerf_stream_t stream);
int enerf_gather_images(const float* const* srcs, int n_srcs, const int* which, const int* idx, long long n, float* out,
                        enerf_stream_t stream);
int enerf_add(const float* a, const float* b, long long n, float* out, enerf_stream_t stream);
int enerf_cast_f64_f32(const double* in, long long n, float* out, enerf_stream_t stream);
int enerf_reciprocal(const float* x, long long n, float* out, enerf_stream_t stream);

/* ---- the steps before / after the path ----
 * Before (ray generation, view selection):
 *   enerf_gen_rays        full-image rays of lib/datasets/enerf_utils.py:61-71: rays (B,Hr*Wr,8) = [o, d, x, y] at the
 *                         intrinsics scaled by `scale`; tar_ext (B,4,4), tar_ixt (B,3,3).
 *   enerf_gen_rays_at     the training branch (enerf_utils.py:33-56): rays of the pixel list xy (B,N,2) int32 = (X,Y) the
 *                         host RNG picked (mask / patch sampling stays numpy, bit-compatible with the reference's seeds).
 *   enerf_rays_bbox_mask  gen_rays_bbox (lib/utils/net_utils.py:13-28): mask (n) int32 = ray hits the box bounds (2,3);
 *                         origin of the first ray, like the reference.  Bit-exact fp32 restatement.
 *   enerf_select_views    zjumocap/enerf_interactive.py:207-210: idx (k) int32 = the k cameras of cam_points (V,3) nearest
 *                         to the target camera centre c2w[:3,3] (c2w (4,4)), nearest first.
 *   enerf_gather_views    :214-217: inps (V,H,W,3) -> src_inps (k,3,H,W); exts (V,4,4) / ixts (V,3,3) -> (k,...).
 * After (presentation, evaluation):
 *   enerf_pack_rgb8       gui_human.py:88-91 (x255, uint8, optional vertical flip); rgb (H*W,3) -> out (H,W,3) bytes;
 *                         bit-exact on [0,1]; outside it saturates (the reference's cast is undefined there).
 *   enerf_eval_stats      evaluators/enerf.py:45-71,88-103.  acc (6 doubles, zeroed by this call):
 *     {sum sq rgb err over selected pixels (x3 ch, float64 like skimage), its count, sum |depth-gt| over gt!=0 (float32
 *     differences like numpy), count, #(<2), #(<10)}; mask (n_rgb, uint8/bool or int32; selected = value >= 1) may be
 *     NULL; img_w > 0 enables the eval_center crop [crop_h:-crop_h, crop_w:-crop_w] of (img_h,img_w) images
 *     (n_rgb = B*img_h*img_w); pass n_depth = 0 to skip the depth part. */
int enerf_gen_rays(const float* tar_ext, const float* tar_ixt, int B, int Hr, int Wr, float scale, float* rays,
                   enerf_stream_t stream);
int enerf_gen_rays_at(const float* tar_ext, const float* tar_ixt, const int* xy, int B, int N, float scale, float* rays,
                      enerf_stream_t stream);
int enerf_rays_bbox_mask(const float* rays, const float* bounds, long long n, int* mask, enerf_stream_t stream);
int enerf_select_views(const float* cam_points, int V, const float* c2w, int k, int* idx, enerf_stream_t stream);
int enerf_gather_views(const float* inps, const float* exts, const float* ixts, const int* idx, int k, int H, int W,
                       float* src_inps, float* src_exts, float* src_ixts, enerf_stream_t stream);
int enerf_pack_rgb8(const float* rgb, int H, int W, int flip, unsigned char* out, enerf_stream_t stream);
int enerf_eval_stats(const float* pred_rgb, const float* gt_rgb, const void* mask, int mask_elem_bytes, long long n_rgb,
                     int img_w, int img_h, int crop_h, int crop_w, const float* pred_depth, const float* gt_depth,
                     long long n_depth, double* acc, enerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ENERF_HIP_H_ */
