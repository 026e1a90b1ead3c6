// Micro-benchmark / lead for the next round: fp32-accurate layers on the bf16 matrix cores ("bf16x3": x = hi + lo in bf16,
// W x ~= Whi xhi + Whi xlo + Wlo xhi, fp32 accumulation) against the exact fp32 MFMA the product uses today.
// A chain of L = 4 dense 64 -> 64 layers with ReLU on 16-point tiles (the shape of the render kernel's MLP phase), weights in LDS,
// activations chained D -> B in registers.  Reports time per tile and the error of both paths against float64.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_bf16x3.hip -o /tmp/bx3 && /tmp/bx3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int L = 4, U = 64;

// fp32 image: [layer][out tile t (4)][k-step ks (16)][lane]: A[row 16t + j][unit 16 (ks/4) + 4 g + ks%4]   (lane = 16 g + j)
// bf16 image: [layer][part (hi, lo)][t (4)][k-chunk c (4)][lane] x 4: A[row 16t + j][k = 16c + 4g + r]
__global__ __launch_bounds__(256) void k_fp32(const float* __restrict__ wimg, const float* __restrict__ x, float* __restrict__ y, int ntiles, int reps) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < L * 4 * 16 * 64; i += 256) lds[i] = wimg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    for (int tile = blockIdx.x * 4 + wv; tile < ntiles; tile += gridDim.x * 4) {
        f32x4 act[4], act0[4];
        for (int t = 0; t < 4; ++t) act0[t] = act[t] = *reinterpret_cast<const f32x4*>(x + ((long long)tile * 16 + j) * U + 16 * t + 4 * g);
        for (int rep = 0; rep < reps; ++rep) {
        if (rep > 0) for (int t = 0; t < 4; ++t) act[t] = act0[t] + act[t] * 1e-30f;     // same chain again (timing), not hoistable
        for (int l = 0; l < L; ++l) {
            f32x4 out[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[((l * 4 + t) * 16 + ks) * 64 + lane], act[ks >> 2][ks & 3], acc, 0, 0, 0);
                out[t] = acc;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) act[t][r] = fmaxf(out[t][r], 0.f);
        }
        }
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(y + ((long long)tile * 16 + j) * U + 16 * t + 4 * g) = act[t];
    }
}

__device__ __forceinline__ void split(const f32x4 v, bf16x4& hi, bf16x4& lo) {
    hi = __builtin_convertvector(v, bf16x4);
    const f32x4 back = __builtin_convertvector(hi, f32x4);
    lo = __builtin_convertvector(v - back, bf16x4);
}
template <int TERMS>   // 3: hi*hi + hi*lo + lo*hi ; 4: + lo*lo
__global__ __launch_bounds__(256) void k_bf16x(const bf16x4* __restrict__ wimg, const float* __restrict__ x, float* __restrict__ y, int ntiles, int reps) {
    extern __shared__ bf16x4 ldsb[];
    for (int i = threadIdx.x; i < L * 2 * 4 * 4 * 64; i += 256) ldsb[i] = wimg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, wv = threadIdx.x >> 6;
    for (int tile = blockIdx.x * 4 + wv; tile < ntiles; tile += gridDim.x * 4) {
        f32x4 act[4], act0[4];
        for (int t = 0; t < 4; ++t) act0[t] = act[t] = *reinterpret_cast<const f32x4*>(x + ((long long)tile * 16 + j) * U + 16 * t + 4 * g);
        for (int rep = 0; rep < reps; ++rep) {
        if (rep > 0) for (int t = 0; t < 4; ++t) act[t] = act0[t] + act[t] * 1e-30f;
        for (int l = 0; l < L; ++l) {
            bf16x4 bh[4], bl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) split(act[c], bh[c], bl[c]);       // the D tile c IS the B operand of k-chunk c
            f32x4 out[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bf16x4 ah = ldsb[(((l * 2 + 0) * 4 + t) * 4 + c) * 64 + lane], al = ldsb[(((l * 2 + 1) * 4 + t) * 4 + c) * 64 + lane];
                    if (TERMS >= 4) acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bl[c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh[c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl[c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh[c], acc, 0, 0, 0);
                }
                out[t] = acc;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) act[t][r] = fmaxf(out[t][r], 0.f);
        }
        }
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(y + ((long long)tile * 16 + j) * U + 16 * t + 4 * g) = act[t];
    }
}

static unsigned short to_bf16(float f) {              // round to nearest even
    unsigned u; memcpy(&u, &f, 4);
    const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
static float from_bf16(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int ntiles = 40960, P = ntiles * 16;          // 655,360 points: the level-1 render of a 512x640 frame (2 samples per ray)
    std::vector<float> W(L * U * U), X((size_t)P * U);
    srand(1);
    for (auto& w : W) w = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.22f;            // ~kaiming for fan-in 64
    for (auto& v : X) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    std::vector<float> img32(L * 4 * 16 * 64);
    std::vector<unsigned short> imgb((size_t)L * 2 * 4 * 4 * 64 * 4);
    for (int l = 0; l < L; ++l)
        for (int t = 0; t < 4; ++t) {
            for (int ks = 0; ks < 16; ++ks)
                for (int lane = 0; lane < 64; ++lane) img32[((l * 4 + t) * 16 + ks) * 64 + lane] = W[(l * U + 16 * t + (lane & 15)) * U + 16 * (ks >> 2) + 4 * (lane >> 4) + (ks & 3)];
            for (int c = 0; c < 4; ++c)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const float w = W[(l * U + 16 * t + (lane & 15)) * U + 16 * c + 4 * (lane >> 4) + r];
                        const unsigned short hi = to_bf16(w), lo = to_bf16(w - from_bf16(hi));
                        imgb[((((size_t)(l * 2 + 0) * 4 + t) * 4 + c) * 64 + lane) * 4 + r] = hi;
                        imgb[((((size_t)(l * 2 + 1) * 4 + t) * 4 + c) * 64 + lane) * 4 + r] = lo;
                    }
        }
    float *dW, *dX, *dY; void* dB;
    hipMalloc(&dW, img32.size() * 4); hipMalloc(&dB, imgb.size() * 2); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, X.size() * 4);
    hipMemcpy(dW, img32.data(), img32.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, imgb.data(), imgb.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    // float64 reference of the first 4096 points
    const int NR = 4096;
    std::vector<double> ref((size_t)NR * U);
    for (int p = 0; p < NR; ++p) {
        double a[U], b[U];
        for (int k = 0; k < U; ++k) a[k] = X[(size_t)p * U + k];
        for (int l = 0; l < L; ++l) {
            for (int o = 0; o < U; ++o) { double s = 0; for (int k = 0; k < U; ++k) s += (double)W[(l * U + o) * U + k] * a[k]; b[o] = s > 0 ? s : 0; }
            for (int k = 0; k < U; ++k) a[k] = b[k];
        }
        for (int k = 0; k < U; ++k) ref[(size_t)p * U + k] = a[k];
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> Y((size_t)NR * U);
    const int REPS = 8;
    auto report = [&](const char* name, float ms, double mfma_cycles_per_tile) {
        hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, rmax = 0, e2 = 0, r2 = 0;
        for (size_t i = 0; i < Y.size(); ++i) { const double d = fabs(Y[i] - ref[i]); emax = fmax(emax, d); rmax = fmax(rmax, fabs(ref[i])); e2 += d * d; r2 += ref[i] * ref[i]; }
        // time of ONE 4-layer chain per tile: (t(REPS chains) - t(1 chain)) / (REPS - 1) takes the HBM stream of x / y out
        printf("%-34s %7.1f us per 4-layer chain over %d tiles (matrix-pipe floor %.1f us)   max|err|/max|ref| %.2e  rms rel %.2e\n", name, 1e3f * ms,
               ntiles, mfma_cycles_per_tile * ntiles / 1024.0 / 2400.0, emax / rmax, sqrt(e2 / r2));
    };
    auto time_it = [&](auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        return best;
    };
    const int grid = 256 * 3;
    auto both = [&](auto launch) {     // (REPS chains - 1 chain) / (REPS - 1); the 1-chain run leaves the result that is checked
        const float tn = time_it([&] { launch(REPS); }), t1 = time_it([&] { launch(1); });
        return (tn - t1) / (REPS - 1);
    };
    float ms = both([&](int r) { hipLaunchKernelGGL(k_fp32, dim3(grid), dim3(256), L * 4 * 16 * 64 * 4, 0, dW, dX, dY, ntiles, r); });
    report("fp32 MFMA 16x16x4 (today)", ms, L * 64 * 32.0);
    ms = both([&](int r) { hipLaunchKernelGGL(k_bf16x<3>, dim3(grid), dim3(256), L * 2 * 4 * 4 * 64 * 8, 0, (const bf16x4*)dB, dX, dY, ntiles, r); });
    report("bf16x3 (hi*hi + hi*lo + lo*hi)", ms, L * 48 * 8.0);
    ms = both([&](int r) { hipLaunchKernelGGL(k_bf16x<4>, dim3(grid), dim3(256), L * 2 * 4 * 4 * 64 * 8, 0, (const bf16x4*)dB, dX, dY, ntiles, r); });
    report("bf16x4 (+ lo*lo)", ms, L * 64 * 8.0);
    return 0;
}
