// Micro-benchmark: the vector-memory request rate of MI355X for the warp kernel's access pattern — what a "gather roof" is.
// k_feature_volume_mp (enerf_amd/csrc/volume.hip) issues, per voxel and source view, four bilinear taps of C channels: CQ = C/4
// consecutive lanes read one contiguous C*4-byte texel (one float4 per lane), 64/CQ texels per wave instruction at data-dependent
// positions that are close together for neighbouring voxels.  Here the same instruction shape is issued back to back:
//   * lanes: 64/CQ texels x CQ lanes x 16 B;  texel = hash(wave, iteration, texel slot) inside a footprint of `fp` bytes per BLOCK
//     (8 KB: L1-resident; 512 KB: per-XCD L2; 32 MB shared: MALL / HBM side), or "near": consecutive texels +- a few rows, as the warp's
//   * `batch` independent loads in flight per wave before their values are consumed (8 = the kernel's two planes x four taps)
// Prints GB/s of REQUESTED bytes and wave-instructions per ns; the warp kernel's achieved request rate is quoted against these in
// bench.py's stage_roofline.volume_* (VERDICT r05 #7).
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/micro/gather_rate.hip -o /tmp/gr && /tmp/gr
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int CQ, int BATCH>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ buf, unsigned texels_per_block, unsigned total_texels, int iters,
                                                int shared_fp, float* out) {
    const unsigned lane = threadIdx.x & 63, cq = lane & (CQ - 1), slot = lane / CQ;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned base = shared_fp ? 0u : (blockIdx.x * texels_per_block) % (total_texels - texels_per_block);
    const unsigned span = shared_fp ? total_texels : texels_per_block;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; it += BATCH) {
        float4 v[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const unsigned t = base + hash((wave * 8191u + (unsigned)(it + b)) * 64u + slot) % span;
            v[b] = buf[(size_t)t * CQ + cq];
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) { acc.x += v[b].x; acc.y += v[b].y; acc.z += v[b].z; acc.w += v[b].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int CQ, int BATCH>
static void run(const float4* buf, unsigned total_texels, size_t fp_bytes, int shared_fp, int waves_per_simd, float* out, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd, iters = 2048;          // one 256-thread block = one wave per SIMD of a CU
    const unsigned tpb = (unsigned)(fp_bytes / (CQ * 16));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_gather<CQ, BATCH>), dim3(blocks), dim3(256), 0, 0, buf, tpb, total_texels, iters, shared_fp, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double instr = (double)blocks * 4 * iters, bytes = instr * 1024.0;
    printf("CQ=%d batch=%d waves/SIMD=%d %-34s: %8.3f ms  %7.1f GB/s requested  %6.2f wave-instr/ns  (%5.1f B/clk/CU at 2.4 GHz)\n", CQ, BATCH,
           waves_per_simd, what, best, bytes / best / 1e6, instr / best / 1e6, bytes / best / 1e6 / 256 / 2.4);
}

int main() {
    const size_t total = 64u << 20;                                   // 64 MB of texels
    float4* buf; hipMalloc(&buf, total); hipMemset(buf, 0, total);
    float* out; hipMalloc(&out, 16);
    for (int w : {2, 5, 8}) {
        run<8, 8>(buf, (unsigned)(total / 128), 8 << 10, 0, w, out, "8 KB per block (L1-resident)");
        run<8, 8>(buf, (unsigned)(total / 128), 512 << 10, 0, w, out, "512 KB per block (L2)");
        run<8, 8>(buf, (unsigned)(total / 128), 0, 1, w, out, "64 MB shared (MALL / HBM side)");
        run<4, 8>(buf, (unsigned)(total / 64), 8 << 10, 0, w, out, "8 KB per block (L1-resident)");
        run<4, 8>(buf, (unsigned)(total / 64), 512 << 10, 0, w, out, "512 KB per block (L2)");
    }
    run<8, 4>(buf, (unsigned)(total / 128), 512 << 10, 0, 5, out, "512 KB per block (L2), batch 4");
    run<8, 16>(buf, (unsigned)(total / 128), 512 << 10, 0, 5, out, "512 KB per block (L2), batch 16");
    run<8, 24>(buf, (unsigned)(total / 128), 512 << 10, 0, 3, out, "512 KB per block (L2), batch 24");
    return 0;
}
