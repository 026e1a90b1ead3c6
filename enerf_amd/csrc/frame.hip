// frame.hip — the whole-frame driver enerf_forward (Network.forward, network.py:76-113 / network_human.py:69-119) and
// the device-side mask_at_box compaction (network_human.py:90-93).  The cascade loop the reference runs in Python is a
// plan (buffer carving) + ~25 kernel enqueues here, in one C call, on one stream, with no host synchronisation.
#include <string.h>

#include <map>
#include <mutex>

#include "kernels.h"
#include "prep_job.h"

using namespace enerf;

namespace enerf {

// =====================================================================================================================
// mask_at_box -> ascending list of selected ray positions (rays[mask_at_box], network_human.py:93).  Three tiny launches:
// per-block counts (1024 elements a block), an exclusive scan of the block counts by one block (also the total),
// and the scatter, which redoes the in-block scan in LDS.  Stable order, so the compacted depth/weights rows match the
// reference's boolean-mask indexing.  HBM-bound byte work: n bytes in, 4*count bytes out.
// =====================================================================================================================
constexpr int kMaskPerThread = 4;
constexpr int kMaskPerBlock = 256 * kMaskPerThread;

__device__ __forceinline__ bool mask_set(const unsigned char* m, int eb, long long i, long long n) {
    if (i >= n) return false;
    if (eb == 1) return m[i] != 0;
    const unsigned char* p = m + i * eb;
    unsigned v = 0;
    for (int k = 0; k < eb; ++k) v |= p[k];
    return v != 0;
}
// in-block exclusive scan of one int per thread (256 threads); returns the block total through `total`
__device__ __forceinline__ int block_exclusive_scan(int v, int* sh, int& total) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    total = sh[255];
    const int excl = sh[t] - v;
    __syncthreads();
    return excl;
}
__global__ __launch_bounds__(256) void k_mask_count(const unsigned char* __restrict__ mask, int eb, long long n,
                                                    int* __restrict__ block_counts) {
    __shared__ int sh[256];
    const long long base = (long long)blockIdx.x * kMaskPerBlock + (long long)threadIdx.x * kMaskPerThread;
    int c = 0;
    for (int k = 0; k < kMaskPerThread; ++k) c += mask_set(mask, eb, base + k, n) ? 1 : 0;
    int total;
    block_exclusive_scan(c, sh, total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void k_mask_scan(const int* __restrict__ block_counts, int nblocks,
                                                   int* __restrict__ block_offsets, int* __restrict__ count) {
    __shared__ int sh[256];
    int carry = 0;
    for (int base = 0; base < nblocks; base += 256) {
        const int i = base + (int)threadIdx.x;
        const int v = i < nblocks ? block_counts[i] : 0;
        int total;
        const int excl = block_exclusive_scan(v, sh, total);
        if (i < nblocks) block_offsets[i] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) count[0] = carry;
}
__global__ __launch_bounds__(256) void k_mask_scatter(const unsigned char* __restrict__ mask, int eb, long long n,
                                                      const int* __restrict__ block_offsets, int* __restrict__ index) {
    __shared__ int sh[256];
    const long long base = (long long)blockIdx.x * kMaskPerBlock + (long long)threadIdx.x * kMaskPerThread;
    bool f[kMaskPerThread];
    int c = 0;
    for (int k = 0; k < kMaskPerThread; ++k) { f[k] = mask_set(mask, eb, base + k, n); c += f[k] ? 1 : 0; }
    int total;
    int o = block_offsets[blockIdx.x] + block_exclusive_scan(c, sh, total);
    for (int k = 0; k < kMaskPerThread; ++k)
        if (f[k]) index[o++] = (int)(base + k);
}
size_t mask_compact_workspace_bytes(long long n) { return (size_t)(2 * cdivl(n > 0 ? n : 1, kMaskPerBlock)) * sizeof(int); }
void launch_mask_compact(const void* mask, int elem_bytes, long long n, int* index, int* count, void* workspace,
                         hipStream_t st) {
    const int nb = (int)cdivl(n, kMaskPerBlock);
    int* counts = (int*)workspace;
    int* offsets = counts + nb;
    const unsigned char* m = (const unsigned char*)mask;
    ENERF_LAUNCH(k_mask_count, (unsigned)nb, 256, 0, st, m, elem_bytes, n, counts);
    ENERF_LAUNCH(k_mask_scan, 1u, 256, 0, st, counts, nb, offsets, count);
    ENERF_LAUNCH(k_mask_scatter, (unsigned)nb, 256, 0, st, m, elem_bytes, n, offsets, index);
}

// =====================================================================================================================
// Frame plan: shapes of every level and the carving of the caller's workspace.
// =====================================================================================================================
namespace {
struct LevelPlan {
    int D, h, w, C, Hs, Ws;            // volume extent; cost-volume feature channels and source-map size
    int Hr, Wr, render, masked, F, Ns; // render extent, flags, nerf feature width (C_f + 3), samples per ray
    long long n_rays;
    // workspace offsets (floats)
    size_t proj, dv, nf, vol, feat3d, prob, depth, std, dmvs, tex, rays;
};
struct FramePlan {
    int tex2, hip_feats;
    size_t f[3], featnet_ws, featnet_ws_bytes, costreg_ws, costreg_ws_bytes;
    size_t ray_index, ray_count, mask_ws;          // float-sized slots
    LevelPlan L[ENERF_MAX_LEVELS];
    size_t total_floats;
};
inline int scaled(int n, double s) { return (int)((double)n * s); }      // python: int(H * scale)

int make_plan(const enerf_frame_args_t* a, FramePlan* P) {
    REQUIRE(a, "forward: null args");
    const enerf_cascade_t& c = a->cas;
    REQUIRE(c.num >= 1 && c.num <= ENERF_MAX_LEVELS, "forward: cas_config.num=%d unsupported (1..%d)", c.num, ENERF_MAX_LEVELS);
    REQUIRE(a->B > 0 && a->S >= 2 && a->S <= 4 && a->H > 0 && a->W > 0 && a->H % 4 == 0 && a->W % 4 == 0,
            "forward: bad batch shape B=%d S=%d H=%d W=%d (S in 2..4, H and W divisible by 4)", a->B, a->S, a->H, a->W);
    REQUIRE(a->src_inps && a->src_exts && a->src_ixts && a->tar_ext && a->tar_ixt && a->near_far, "forward: null batch tensor");
    const int nf = (a->feats_nchw[0] != nullptr) + (a->feats_nchw[1] != nullptr) + (a->feats_nchw[2] != nullptr);
    REQUIRE(nf == 0 || nf == 3, "forward: feats_nchw needs all three levels or none");
    P->hip_feats = nf == 0;
    if (P->hip_feats) REQUIRE(a->feature_net_packed, "forward: feature_net_packed missing");
    // level_2 is only ever the im_feat of a full-resolution render: then the FeatureNet emits it as render texels
    int uses = 0, all_full = 1;
    for (int i = 0; i < c.num; ++i)
        if (c.render_if[i] && c.render_im_feat_level[i] == 2) {
            ++uses;
            all_full &= (c.render_scale[i] == 1.0 && c.im_ibr_scale[i] == 1.0 && c.nerf_model_feat_ch[i] == 8);
        }
    P->tex2 = P->hip_feats && uses > 0 && all_full && c.num <= 2;
    size_t off = 0;
    auto take = [&](size_t nfloats) { size_t r = off; off += (nfloats + 63) / 64 * 64; return r; };   // 256-B aligned
    const long long n_img = (long long)a->B * a->S;
    const int fh[3] = {a->H / 4, a->H / 2, a->H}, fw[3] = {a->W / 4, a->W / 2, a->W}, fc[3] = {32, 16, 8};
    for (int l = 0; l < 3; ++l) P->f[l] = take((size_t)n_img * fh[l] * fw[l] * (l == 2 && P->tex2 ? 12 : fc[l]));
    P->featnet_ws_bytes = P->hip_feats ? enerf_feature_net_workspace_bytes((int)n_img, a->H, a->W) : 0;
    P->featnet_ws = take(P->featnet_ws_bytes / sizeof(float));
    P->costreg_ws_bytes = 0;
    for (int i = 0; i < c.num; ++i) {
        LevelPlan& L = P->L[i];
        memset(&L, 0, sizeof(L));
        L.D = c.volume_planes[i];
        L.h = scaled(a->H, c.volume_scale[i]);
        L.w = scaled(a->W, c.volume_scale[i]);
        REQUIRE(i < 3, "forward: level %d has no feature map (FeatureNet has three scales)", i);
        L.C = fc[i]; L.Hs = fh[i]; L.Ws = fw[i];
        REQUIRE(!(i == 2 && P->tex2), "forward: level_2 texels cannot feed a cost volume");
        REQUIRE(L.D > 0 && L.h > 0 && L.w > 0, "forward: level %d volume is empty", i);
        if (i > 0) REQUIRE(c.depth_inv[i - 1], "forward: cascade levels after a depth-space level are undefined in the "
                                               "reference (utils.py:130)");
        REQUIRE(a->cost_reg_packed[i], "forward: cost_reg_packed[%d] missing", i);
        const long long nv = (long long)a->B * L.D * L.h * L.w;
        L.proj = take((size_t)a->B * a->S * 12);
        L.dv = take((size_t)nv);
        L.nf = take((size_t)a->B * 2 * L.h * L.w);
        L.vol = take((size_t)nv * L.C);
        L.feat3d = take((size_t)nv * 8);
        L.prob = take((size_t)nv);
        L.depth = take((size_t)a->B * L.h * L.w);
        L.std = take((size_t)a->B * L.h * L.w);
        L.dmvs = take((size_t)a->B * L.h * L.w);
        const size_t cw = enerf_cost_reg_workspace_bytes(i != 0, a->B, L.D, L.h, L.w);
        if (cw > P->costreg_ws_bytes) P->costreg_ws_bytes = cw;
        L.render = c.render_if[i] != 0;
        if (!L.render) continue;
        L.Hr = scaled(a->H, c.render_scale[i]);
        L.Wr = scaled(a->W, c.render_scale[i]);
        L.Ns = c.num_samples[i];
        L.F = c.nerf_model_feat_ch[i] + 3;
        REQUIRE(L.Hr > 1 && L.Wr > 1, "forward: level %d render extent too small", i);
        REQUIRE(a->nerf_packed[i], "forward: nerf_packed[%d] missing", i);
        REQUIRE(a->rgb[i] && a->depth[i] && a->weights[i] && a->depth_mvs[i] && a->std[i], "forward: level %d output missing", i);
        const int fl = c.render_im_feat_level[i];
        REQUIRE(fl >= 0 && fl <= 2 && fc[fl] == c.nerf_model_feat_ch[i],
                "forward: render_im_feat_level[%d]=%d does not have nerf_model_feat_ch=%d channels", i, fl, c.nerf_model_feat_ch[i]);
        if (P->hip_feats)
            REQUIRE(fh[fl] == L.Hr && fw[fl] == L.Wr, "forward: level %d renders at %dx%d but feature level_%d is %dx%d "
                    "(the HIP FeatureNet path needs render_scale == im_ibr_scale)", i, L.Hr, L.Wr, fl, fh[fl], fw[fl]);
        else {
            const double up = c.render_scale[i] / c.im_ibr_scale[i];
            REQUIRE(scaled(fh[fl], up) == L.Hr && scaled(fw[fl], up) == L.Wr,
                    "forward: im_feat resolution inconsistent with render_scale / im_ibr_scale at level %d", i);
        }
        if (!(fl == 2 && P->tex2)) L.tex = take((size_t)n_img * L.Hr * L.Wr * 4 * ((L.F + 3) / 4));
        if (a->rays[i] != nullptr) {
            REQUIRE(a->n_rays[i] >= 0, "forward: n_rays[%d] negative", i);
            L.n_rays = a->n_rays[i];
        } else {
            L.n_rays = (long long)L.Hr * L.Wr;
            L.rays = take((size_t)a->B * L.n_rays * 8);
        }
        L.masked = a->mask_at_box != nullptr && i == c.num - 1;
        if (L.masked) {
            REQUIRE(a->B == 1, "forward: mask_at_box needs B == 1 (network_human.py:91 reshapes the mask to (1,-1))");
            REQUIRE(L.n_rays == (long long)a->H * a->W, "forward: mask_at_box has H*W=%lld elements but level %d has %lld rays",
                    (long long)a->H * a->W, i, L.n_rays);
            REQUIRE(a->mask_elem_bytes == 1 || a->mask_elem_bytes == 2 || a->mask_elem_bytes == 4 || a->mask_elem_bytes == 8,
                    "forward: mask_elem_bytes=%d unsupported", a->mask_elem_bytes);
            REQUIRE((a->ray_index != nullptr) == (a->ray_count != nullptr), "forward: pass both ray_index and ray_count or neither");
            REQUIRE(!a->ray_index_ready || a->ray_index, "forward: ray_index_ready without ray_index");
            if (!a->ray_index) { P->ray_index = take((size_t)L.n_rays); P->ray_count = take(64); }
            if (!a->ray_index_ready) P->mask_ws = take(mask_compact_workspace_bytes(L.n_rays) / sizeof(int) + 1);
        }
    }
    P->costreg_ws = take(P->costreg_ws_bytes / sizeof(float));
    P->total_floats = off;
    return ENERF_OK;
}
}  // namespace
}  // namespace enerf

// =====================================================================================================================
// Side lane of a frame.  Level 0 (warp + CostRegNet + depth regression) consumes only the FeatureNet's coarsest map, and
// its deep small layers leave most of the chip idle (mfma busy 0.05-0.13 on 80-640 tiles); the FeatureNet's top-down half
// — up2+lat1, smooth1 (level 1's source maps) and the fused up2+lat0+smooth0 (the render texels, the second largest
// kernel of the frame) — is needed later.  So enerf_forward forks that half onto a library-owned stream right after the
// trunk and joins it with events before its first consumer: the two chains overlap inside ONE frame.
// The render of a non-final cascade level (render_if True,True: lego, training-style eval) is a leaf as well — nothing in the
// next level reads its rgb/depth/weights — so it is forked onto the lane's second stream after the level's depth regression
// and joined at the end of the frame.
// One lane (two streams + events) per caller stream, created on first use, never destroyed (process lifetime).
// =====================================================================================================================
#ifndef ENERF_EMU
namespace {
struct SideLane { hipStream_t stream, rstream; hipEvent_t trunk, l1, l2, fork, done; std::mutex busy; };
SideLane* side_lane(hipStream_t main) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, SideLane*> lanes;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(dev, main);
    auto it = lanes.find(key);
    if (it != lanes.end()) return it->second;
    SideLane* L = new SideLane();
    // lowest priority: the lanes carry leaves of the frame, the caller's stream carries its critical path — when both have
    // workgroups waiting, the chain everything else depends on should get the compute units first (a GPU-filling smooth0
    // on the lane stretched a small level-0 layer on the caller's stream 10x at 1024x1024: zju 430 -> 437 frames/s)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    // (a CU-masked lane stream — hipExtStreamCreateWithCUMask keeping 4 / 6 / 7 of every 8 CUs, so that the chain's small layers always
    // find free CUs — measured round 6: dtu 1320 -> 962 frames/s, zju 555 -> 479 whatever the mask: profiles/r06_ab_lane_cu_mask.txt)
    bool ok = hipStreamCreateWithPriority(&L->stream, hipStreamNonBlocking, least) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&L->trunk, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&L->l1, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&L->l2, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&L->rstream, hipStreamNonBlocking, least) == hipSuccess;   // renders of non-final levels
    ok = ok && hipEventCreateWithFlags(&L->fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&L->done, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete L; L = nullptr; (void)hipGetLastError(); }
    lanes[key] = L;                                  // a failed creation is remembered: the frame then runs on one stream
    return L;
}
}  // namespace
#endif

extern "C" {

size_t enerf_mask_compact_workspace_bytes(long long n) { return mask_compact_workspace_bytes(n); }
int enerf_mask_compact(const void* mask, int elem_bytes, long long n, int* index, int* count, void* workspace,
                       size_t workspace_bytes, enerf_stream_t stream) {
    REQUIRE(mask && index && count && workspace && n > 0, "mask_compact: bad arguments");
    REQUIRE(n < (1LL << 31), "mask_compact: more than 2^31 elements");
    REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, "mask_compact: elem_bytes=%d unsupported", elem_bytes);
    if (workspace_bytes < mask_compact_workspace_bytes(n)) return fail(ENERF_EWORKSPACE, "mask_compact: workspace too small");
    launch_mask_compact(mask, elem_bytes, n, index, count, workspace, (hipStream_t)stream);
    return check_launch("mask_compact");
}

size_t enerf_forward_workspace_bytes(const enerf_frame_args_t* a) {
    FramePlan P;
    if (make_plan(a, &P) != ENERF_OK) return 0;
    return P.total_floats * sizeof(float);
}

int enerf_forward(const enerf_frame_args_t* a, enerf_stream_t stream) {
    FramePlan P;
    int rc = make_plan(a, &P);
    if (rc != ENERF_OK) return rc;
    REQUIRE(a->workspace, "forward: null workspace");
    if (a->workspace_bytes < P.total_floats * sizeof(float))
        return fail(ENERF_EWORKSPACE, "forward: workspace too small (%zu < %zu bytes)", a->workspace_bytes,
                    P.total_floats * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)a->workspace;
    const enerf_cascade_t& c = a->cas;
    const int n_img = a->B * a->S;
    hipStream_t cur = st;                // the stream the current stage is enqueued on (the lane's for a forked render)
    auto mark = [&](int slot) {
#ifndef ENERF_EMU
        if (a->stage_events != nullptr && a->stage_events[slot] != nullptr) hipEventRecord((hipEvent_t)a->stage_events[slot], cur);
#else
        (void)slot;
#endif
    };
    mark(ENERF_STAGE_BEGIN);
    // a caller-independent early start for the mask compaction: it only depends on the batch
    int *ray_index = a->ray_index, *ray_count = a->ray_count;
    const LevelPlan& last = P.L[c.num - 1];
    if (last.render && last.masked) {
        if (!ray_index) { ray_index = (int*)(ws + P.ray_index); ray_count = (int*)(ws + P.ray_count); }
        if (!a->ray_index_ready)
            launch_mask_compact(a->mask_at_box, a->mask_elem_bytes, last.n_rays, ray_index, ray_count, ws + P.mask_ws, st);
    }

    // ---- FeatureNet (feature_net.py:27-36) -> channels-last maps; level_2 straight to render texels when it can ----
    float* f[3] = {ws + P.f[0], ws + P.f[1], ws + P.f[2]};
    const int fh[3] = {a->H / 4, a->H / 2, a->H}, fw[3] = {a->W / 4, a->W / 2, a->W}, fc[3] = {32, 16, 8};
    bool forked = false;                 // the FeatureNet's top-down half runs on the side lane
    int gate_mode = 1;                   // see enerf_options_t.side_gate (resolved below)
    bool stage2_enqueued = true;         // the FeatureNet's last stage (smooth0) has been enqueued (false while gated)
    int stage2_rc = ENERF_OK;
    int render_forks = 0;                // renders of non-final levels enqueued on the lane's second stream
    int joined[3] = {1, 1, 1};           // feature level l is visible to the caller's stream
#ifndef ENERF_EMU
    SideLane* lane = nullptr;
    // the lane's events are shared by every frame enqueued on this caller stream: two host threads calling enerf_forward on
    // the SAME stream would interleave hipEventRecord / hipStreamWaitEvent pairs (a wait could bind to the other call's
    // record).  The enqueue is serialised per lane; the lock is held until this call has enqueued its last join.
    std::unique_lock<std::mutex> lane_busy;
#endif
    auto need_level = [&](int l) {       // call before the first consumer of f[l] on the caller's stream
#ifndef ENERF_EMU
        if (forked && !joined[l]) { hipStreamWaitEvent(st, l == 1 ? lane->l1 : lane->l2, 0); joined[l] = 1; }
#else
        (void)l;
#endif
    };
    auto bail = [&](int code) {          // error exit after a fork: never leave a lane stream un-joined
#ifndef ENERF_EMU
        if (render_forks > 0) { hipEventRecord(lane->done, lane->rstream); hipStreamWaitEvent(st, lane->done, 0); }
#endif
        need_level(1); need_level(2);
        return code;
    };
    // The camera-only preparation — level 0's depth planes and EVERY level's projection matrices (utils.py:35-55, 98-111) — rides
    // in the frame's first launch (prep_job.h): its blocks run beside conv0's instead of as a launch of their own between the trunk
    // and the warp.  prep_carried: a kernel took the job (the fused conv0 pair; otherwise the level loop launches the prep kernels).
    int prep_carried = 0;
    PrepJob job;
    memset(&job, 0, sizeof(job));
#ifndef ENERF_PREP_JOB
#define ENERF_PREP_JOB 1             // 0 (A/B): the preparation as launches of its own inside the level loop (rounds 1 - 5)
#endif
    if (P.hip_feats && ENERF_PREP_JOB) {
        const LevelPlan& L0 = P.L[0];
        job.near_far = a->near_far; job.dv = ws + L0.dv; job.nf = ws + L0.nf;
        job.B = a->B; job.D = L0.D; job.h = L0.h; job.w = L0.w; job.depth_inv = c.depth_inv[0];
        for (int i = 0; i < c.num && i < 3; ++i)
            job.pj[i] = ProjJob{a->src_ixts, a->src_exts, a->tar_ixt, a->tar_ext, ws + P.L[i].proj, a->S, (float)c.im_feat_scale[i],
                                (float)c.volume_scale[i]};
        job.nblocks = prep_job_blocks(a->B, L0.D, L0.h, L0.w, 256);
    }
    if (P.hip_feats) {
        const int l2s = P.tex2 ? 12 : 8;
        auto fstage = [&](int stage, enerf_stream_t s) {
            const bool first = stage == ENERF_FEAT_ALL || stage == ENERF_FEAT_TRUNK;
            return feature_net_stage_job(a->feature_net_packed, a->src_inps, n_img, a->H, a->W, f[0], f[1], f[2], l2s,
                                         ws + P.featnet_ws, P.featnet_ws_bytes, stage, a->options, (hipStream_t)s,
                                         first && job.nblocks > 0 ? &job : nullptr, first ? &prep_carried : nullptr);
        };
#ifndef ENERF_EMU
        if (!(a->options && a->options->single_stream)) lane = side_lane(st);
        if (lane != nullptr) {
            lane_busy = std::unique_lock<std::mutex>(lane->busy);
            rc = fstage(ENERF_FEAT_TRUNK, stream);
            if (rc != ENERF_OK) return rc;
            hipEventRecord(lane->trunk, st);
            hipStreamWaitEvent(lane->stream, lane->trunk, 0);
            rc = fstage(ENERF_FEAT_LEVEL1, (enerf_stream_t)lane->stream);
            hipEventRecord(lane->l1, lane->stream);
            // The last stage (lat0 + smooth0 -> render texels) is needed only by the final render.  Enqueued here (default) it
            // shares the chip with level 0; GATED (enerf_options_t.side_gate >= 2) it is enqueued later, from inside the last
            // level's cost regularisation, behind an event, to run beside that level's small deep layers instead.  Measured in
            // round 3 (profiles/r03_ab_side_gate.txt): gating LOSES 1-2 % on all three workloads — the early start wins.
            gate_mode = a->options ? a->options->side_gate : 0;
            if (gate_mode == 0) gate_mode = 1;
            const int lastl = c.num - 1;
            const bool gateable = c.num >= 2 && P.tex2 && c.render_if[lastl] && c.render_im_feat_level[lastl] == 2;
            for (int l = 0; l < lastl && gateable; ++l)                       // nobody before the last level may need level_2
                if (c.render_if[l] && c.render_im_feat_level[l] == 2) gate_mode = 1;
            if (!gateable) gate_mode = 1;
            if (gate_mode == 1) {
                if (rc == ENERF_OK) rc = fstage(ENERF_FEAT_LEVEL2, (enerf_stream_t)lane->stream);
                hipEventRecord(lane->l2, lane->stream);
                stage2_enqueued = true;
            }
            forked = true; joined[1] = joined[2] = 0;
            if (rc != ENERF_OK) return bail(rc);      // never leave the lane un-joined
        } else
#endif
        {
            rc = fstage(ENERF_FEAT_ALL, stream);
            if (rc != ENERF_OK) return rc;
        }
    } else {
        for (int l = 0; l < c.num; ++l)   // the levels that feed a cost volume (texels are packed from NCHW below)
            launch_channels_last(a->feats_nchw[l], f[l], n_img, fc[l], (long long)fh[l] * fw[l], fc[l], st);
    }
    mark(ENERF_STAGE_FEATURE_NET);

#ifndef ENERF_EMU
    // deferred enqueue of the FeatureNet's last stage on the side lane, behind an event recorded on the caller's stream NOW
    struct GateCtx { void* self; };
    auto enqueue_stage2 = [&]() {
        if (stage2_enqueued || lane == nullptr) return;
        hipEventRecord(lane->fork, st);                               // "the chain has reached this point"
        hipStreamWaitEvent(lane->stream, lane->fork, 0);
        const int l2s = P.tex2 ? 12 : 8;
        stage2_rc = enerf_feature_net_stage(a->feature_net_packed, a->src_inps, n_img, a->H, a->W, f[0], f[1], f[2], l2s,
                                            ws + P.featnet_ws, P.featnet_ws_bytes, ENERF_FEAT_LEVEL2, a->options,
                                            (enerf_stream_t)lane->stream);
        hipEventRecord(lane->l2, lane->stream);
        stage2_enqueued = true;
    };
    auto* enq_ptr = &enqueue_stage2;
    CostRegHook hook = {[](void* ctx) { (*static_cast<decltype(enq_ptr)>(ctx))(); }, enq_ptr, gate_mode == 4 ? 2 : 0};
    if (forked && gate_mode != 1) stage2_enqueued = false;
#endif
    const float *pdepth = nullptr, *pstd = nullptr, *pnf = nullptr;
    const float *pending_prob = nullptr, *pending_dv = nullptr;       // a depth regression deferred into the next level's prep
    int pending_D = 0, pending_inv = 0;
    int hp = 0, wp = 0;
    for (int i = 0; i < c.num; ++i) {
        const LevelPlan& L = P.L[i];
        float *proj = ws + L.proj, *dv = ws + L.dv, *nf = ws + L.nf, *vol = ws + L.vol, *feat3d = ws + L.feat3d;
        float *prob = ws + L.prob, *depth = ws + L.depth;
        float* std = L.render ? a->std[i] : ws + L.std;
        float* dmvs = L.render ? a->depth_mvs[i] : nullptr;
        // the previous level's depth regression rides in this level's prep launch when that level is not rendered (its depth /
        // std are then only this level's inputs): one launch instead of two on the critical chain between the levels
        bool prep_done = false;
        if (pending_prob != nullptr) {
            // (the level's projection matrices are already there when the frame's first launch carried the preparation job)
            prep_done = launch_regress_and_values(a->src_ixts, a->src_exts, a->tar_ixt, a->tar_ext, a->S, (float)c.im_feat_scale[i],
                                                  (float)c.volume_scale[i], prep_carried && i < 3 ? nullptr : proj, pending_prob, pending_dv, pnf, pending_D, hp, wp,
                                                  pending_inv, const_cast<float*>(pdepth), const_cast<float*>(pstd), a->B, L.D,
                                                  L.h, L.w, c.depth_inv[i], dv, nf, st);
            if (!prep_done)         // shape outside the fused kernel's limits: the two separate launches
                launch_depth_regression(pending_prob, pending_dv, a->B, pending_D, hp, wp, pending_inv,
                                        const_cast<float*>(pdepth), const_cast<float*>(pstd), nullptr, st);
            pending_prob = nullptr;
        }
        if (i == 0 && prep_carried) prep_done = true;                  // level 0: planes + matrices came with the first launch
        if (!prep_done)
            rc = enerf_level_prep(a->src_ixts, a->src_exts, a->tar_ixt, a->tar_ext, a->B, a->S, (float)c.im_feat_scale[i],
                                  (float)c.volume_scale[i], proj, a->near_far, pdepth, pstd, pnf, L.D, L.h, L.w, hp, wp,
                                  c.depth_inv[i], dv, nf, stream);
        else
            rc = check_launch("level_prep");
        if (rc != ENERF_OK) return bail(rc);
        mark(ENERF_STAGE_LEVEL(i, ENERF_STAGE_PREP));
        need_level(i);                                                 // level i's source maps (side lane for i >= 1)
        // the volume goes to conv0 as channel-quad planes when conv0 runs on the asynchronously staged kernel (one quad per pass)
        const int vol_planar = cost_reg_wants_planar_volume(resolve_options(a->options), L.C, a->B, L.D, L.h, L.w) ? 1 : 0;
        if (!vol_planar)
            rc = enerf_build_feature_volume(f[i], proj, dv, a->B, a->S, L.C, L.Hs, L.Ws, L.D, L.h, L.w, vol, stream);
        else {      // same argument checks as the C entry (shapes come from the validated plan; the 32-bit limits are re-checked)
            REQUIRE((long long)a->B * a->S * L.Hs * L.Ws * L.C < (1LL << 32) && (long long)L.Hs * L.Ws < (1LL << 23) &&
                    (long long)a->B * L.D * L.h * L.w * (L.C / 4) < (1LL << 31) && (long long)L.h * L.w < (1LL << 23) &&
                    (long long)a->B * L.D <= 65535 && (long long)a->B * L.D * L.h < (1LL << 23) && L.w < (1 << 23),
                    "forward: level %d volume too large for 32-bit indices / the grid-carried voxel decomposition", i);
            launch_feature_volume(f[i], proj, dv, a->B, a->S, L.C, L.Hs, L.Ws, L.D, L.h, L.w, vol, st, 1);
            rc = check_launch("build_feature_volume");
        }
        if (rc != ENERF_OK) return bail(rc);
        mark(ENERF_STAGE_LEVEL(i, ENERF_STAGE_VOLUME));
        const CostRegHook* hk = nullptr;
#ifndef ENERF_EMU
        if (!stage2_enqueued && i == c.num - 1) {
            if (gate_mode == 3) enqueue_stage2(); else hk = &hook;
        }
#endif
        rc = cost_reg_run(a->cost_reg_packed[i], L.C, i != 0, vol, vol_planar, a->B, L.D, L.h, L.w, feat3d, prob, ws + P.costreg_ws,
                          P.costreg_ws_bytes, a->options, st, hk);
#ifndef ENERF_EMU
        if (!stage2_enqueued && i == c.num - 1) enqueue_stage2();     // (an error path inside cost_reg skipped the hook)
        if (rc == ENERF_OK && stage2_rc != ENERF_OK) rc = stage2_rc;
#endif
        if (rc != ENERF_OK) return bail(rc);
        mark(ENERF_STAGE_LEVEL(i, ENERF_STAGE_COST_REG));
        const bool defer_regression = !L.render && i + 1 < c.num && !(a->options && a->options->fuse_depth_prep == 1);
        if (defer_regression) { pending_prob = prob; pending_dv = dv; pending_D = L.D; pending_inv = c.depth_inv[i]; }
        else launch_depth_regression(prob, dv, a->B, L.D, L.h, L.w, c.depth_inv[i], depth, std, dmvs, st);
        mark(ENERF_STAGE_LEVEL(i, ENERF_STAGE_DEPTH_REG));
        pdepth = depth; pstd = std; pnf = nf; hp = L.h; wp = L.w;
        if (!L.render) continue;

        // ---- texels: unpreprocess + cat (network.py:28-34) as the channels-last gather source ----
        const int fl = c.render_im_feat_level[i];
        const int TEX = 4 * ((L.F + 3) / 4);
        const float* tex;
        need_level(fl);                                                // the texel source (joined here, inside the texel stage)
        // a non-final level's render is a leaf of the frame: fork it (its inputs are complete on the caller's stream here)
        bool render_forked = false;
        enerf_stream_t rs = stream;
#ifndef ENERF_EMU
        if (forked && i + 1 < c.num && !L.masked) {
            if (render_forks > 0) hipStreamWaitEvent(lane->rstream, lane->done, 0);   // (ordering only: same stream anyway)
            hipEventRecord(lane->fork, st);
            hipStreamWaitEvent(lane->rstream, lane->fork, 0);
            rs = (enerf_stream_t)lane->rstream; cur = lane->rstream; render_forked = true; ++render_forks;
        }
#endif
        if (fl == 2 && P.tex2) tex = f[2];
        else {
            float* t = ws + L.tex;
            if (P.hip_feats)
                rc = enerf_pack_texels_cl(f[fl], fc[fl], a->src_inps, a->H, a->W, L.Hr, L.Wr, TEX, n_img, t, rs);
            else
                rc = enerf_pack_img_feat_rgb(a->feats_nchw[fl], fc[fl], fh[fl], fw[fl], a->src_inps, a->H, a->W, L.Hr, L.Wr,
                                             TEX, n_img, t, rs);
            if (rc != ENERF_OK) return bail(rc);
            tex = t;
        }
        mark(ENERF_STAGE_LEVEL(i, ENERF_STAGE_TEXELS));

        // ---- rays: the batch's, or the full image generated here (enerf_utils.py:61-71) ----
        const float* rays8 = a->rays[i];
        if (rays8 == nullptr) {
            float* r = ws + L.rays;
            rc = enerf_gen_rays(a->tar_ext, a->tar_ixt, a->B, L.Hr, L.Wr, (float)c.render_scale[i], r, rs);
            if (rc != ENERF_OK) return bail(rc);
            rays8 = r;
        }
        // ---- build_rays + render_rays (utils.py:390-420, network.py:24-43), one launch ----
        enerf_render_args_t ra;
        memset(&ra, 0, sizeof(ra));
        ra.tex = tex; ra.vol = feat3d; ra.src_exts = a->src_exts; ra.src_ixts = a->src_ixts; ra.tar_ext = a->tar_ext;
        ra.packed = a->nerf_packed[i];
        ra.rgb = a->rgb[i]; ra.depth = a->depth[i]; ra.weights = a->weights[i];
        ra.B = a->B; ra.N = (int)L.n_rays; ra.S = a->S; ra.n_samples = L.Ns; ra.depth_inv = c.depth_inv[i];
        ra.Hr = L.Hr; ra.Wr = L.Wr; ra.F = L.F; ra.D = L.D; ra.h = L.h; ra.w = L.w; ra.white_bkgd = c.white_bkgd;
        ra.render_scale = (float)c.render_scale[i];
        ra.rays8 = rays8; ra.depth_map = depth; ra.std_map = std; ra.nf_map = nf; ra.map_h = L.h; ra.map_w = L.w;
        ra.options = a->options;
        // A forked render is a leaf that shares the device with the next level.  As one persistent block per compute unit (130 KB of
        // LDS each at C = 32) it kept the next level's LDS-staged kernels off every CU until its blocks exited; on HALF of the CUs,
        // with the balanced tile deal, the next level starts at once and the leaf ends under its small layers: lego 543 -> 558
        // frames/s (64 blocks 509, 96: 556, 128: 558, 160: 546, all 256: 535; profiles/r06_ab_bg_render_blocks.txt)
#ifndef ENERF_BG_RENDER_DIV
#define ENERF_BG_RENDER_DIV 2        // a forked render's persistent blocks = compute units / this; 0 = one per compute unit
#endif
        if (render_forked && ENERF_BG_RENDER_DIV > 0) ra.max_blocks = device_cu_count() / ENERF_BG_RENDER_DIV;
        if (L.masked) {
            ra.ray_index = ray_index; ra.ray_count = ray_count; ra.scatter_rgb = 1;
            zero_async(a->rgb[i], (size_t)L.n_rays * 3 * sizeof(float), (hipStream_t)rs);      // torch.zeros_like(...), network_human.py:103
        }
        rc = enerf_render_rays(&ra, rs);
        if (rc != ENERF_OK) return bail(rc);
        mark(ENERF_STAGE_LEVEL(i, ENERF_STAGE_RENDER));
#ifndef ENERF_EMU
        if (render_forked) { hipEventRecord(lane->done, lane->rstream); cur = st; }
#endif
        (void)render_forked;
    }
#ifndef ENERF_EMU
    if (render_forks > 0) hipStreamWaitEvent(st, lane->done, 0);       // join the forked renders
#endif
    need_level(1); need_level(2);        // the caller's stream never returns ahead of the side lane
    return check_launch("forward");
}

}  // extern "C"
