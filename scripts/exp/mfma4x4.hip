// Experiment (GPU): semantics and rates of v_mfma_f32_4x4x1_16b_f32 used as a "per-lane 4-output FMA":
//   D[r](lane) = C[r](lane) + A(lane 4*(lane/4) + r) * B(lane)
// i.e. with the particle on the lane (B = the lane's own activation) and the weights of 4 output rows held by the 4 lanes
// of each block (A), one instruction does 4 FMAs per lane without leaving the particle-per-lane layout.
//   hipcc --offload-arch=gfx950 -O3 -o mfma4x4 mfma4x4.hip && ./mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void sem_kernel(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[lane * 4 + r] = c[r];
}

// mode 0: NACC independent accumulators, ITERS rounds (throughput); mode 1: one dependent chain (latency);
// mode 2: throughput with one v_fma per MFMA interleaved (co-issue from the same wave)
template <int MODE>
__global__ void rate_kernel(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-3f, b = 1.0f - lane * 1e-3f;
    f32x4 c[8];
    for (int k = 0; k < 8; ++k) c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = lane + k;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[0], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[k], 0, 0, 0);
                if (MODE == 2) v[k] = __builtin_fmaf(v[k], a, b);
                if (MODE == 3) { v[k] = __builtin_fmaf(v[k], a, b); v[(k + 4) & 7] = __builtin_fmaf(v[(k + 4) & 7], b, a); }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3] + v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// exact instruction mixes (inline asm, nothing for the compiler to fuse): per round 8 MFMA (or 0) + NF plain v_fma_f32
template <int NM, int NF, int BIG = 0>
__global__ void mix_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-3f, b = 1.0f - lane * 1e-3f;
    f32x4 c[8];
    for (int k = 0; k < 8; ++k) c[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = lane + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < NM && !BIG) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(b));
            if (k < NM && BIG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF / 8; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(k + 3 * f) & 7]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3] + v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NF, int BIG = 0>
int run_mix(int waves_per_simd) {
    const int iters = 4000, blocks = 256, threads = 256 * waves_per_simd;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 1024));
    mix_kernel<NM, NF, BIG><<<blocks, threads>>>(out, 10);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mix_kernel<NM, NF, BIG><<<blocks, threads>>>(out, iters);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("mix %d %s MFMA + %2d v_fma per round, waves/SIMD %d: %.3f ms -> %.2f ns per round per SIMD-wave-slot, %.2f ns per SIMD round of all waves\n",
           NM, BIG ? "16x16x4" : "4x4x1", NF, waves_per_simd, best, best * 1e6 / iters / waves_per_simd, best * 1e6 / iters);
    hipFree(out);
    return 0;
}

template <int MODE>
int run_rate(const char* name, int waves_per_simd) {
    const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 1024));
    CHECK(hipMalloc(&cyc, 8));
    rate_kernel<MODE><<<blocks, threads>>>(out, cyc, 10);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n_mfma = 8.0 * iters;
    printf("%-44s waves/SIMD %d: %.2f memtime ticks per MFMA per wave (wall %.3f ms)\n", name, waves_per_simd, (double)c / n_mfma, ms);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; ++l) { a[l] = 1.0f + l; b[l] = 100.0f + l; }
    float *da, *db, *dd;
    CHECK(hipMalloc(&da, 256)); CHECK(hipMalloc(&db, 256)); CHECK(hipMalloc(&dd, 1024));
    CHECK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    sem_kernel<<<1, 64>>>(da, db, dd);
    CHECK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r)
            if (d[l * 4 + r] != a[4 * (l / 4) + r] * b[l]) ++bad;
    printf("semantics D[r](lane) = A(4*(lane/4)+r) * B(lane): %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, d[4 * l], d[4 * l + 1], d[4 * l + 2], d[4 * l + 3]);
    for (int wps = 1; wps <= 4; wps *= 2) {
        run_rate<0>("independent accumulators", wps);
        run_rate<1>("one dependent chain", wps);
        run_rate<2>("independent + 1 v_fma per MFMA", wps);
        run_rate<3>("independent + 2 v_fma per MFMA", wps);
    }
    for (int wps = 1; wps <= 4; wps += 3) {
        run_mix<8, 0>(wps); run_mix<0, 8>(wps); run_mix<0, 16>(wps); run_mix<8, 8>(wps); run_mix<8, 16>(wps);
        run_mix<8, 24>(wps); run_mix<8, 32>(wps); run_mix<4, 32>(wps); run_mix<0, 32>(wps);
        run_mix<8, 0, 1>(wps); run_mix<8, 32, 1>(wps); run_mix<8, 64, 1>(wps); run_mix<8, 96, 1>(wps); run_mix<0, 64>(wps); run_mix<0, 96>(wps);
    }
    return 0;
}
