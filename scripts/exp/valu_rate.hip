// Experiment (GPU): fp32 VALU issue rate on gfx950 -- cycles per wave64 instruction on one SIMD for
// v_fma_f32, v_pk_fma_f32, v_exp_f32 and v_mfma_f32_16x16x4_f32, with 1, 2, 4 and 8 waves per SIMD, and
// the chip-wide FLOP/s they give.  Pins the "VALU issue" roofline bench.py prices the spline flow kernels
// against.  build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) float f4;
constexpr int UNROLL = 16, ITERS = 4096;

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc) {
    const float s = 1.0f + 1e-9f * threadIdx.x;
    float a[UNROLL];
    f2 p[UNROLL];
    f4 c[4];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { a[u] = 0.5f + u; p[u] = f2{0.25f + u, 0.75f + u}; }
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = f4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) a[u] = __builtin_fmaf(a[u], s, 0.5f);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) p[u] = __builtin_elementwise_fma(p[u], f2{s, s}, f2{0.5f, 0.25f});
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) a[u] = __builtin_amdgcn_exp2f(a[u]);
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], s, c[u & 3], 0, 0, 0);
        } else if (MODE == 4) {   // half MFMA, half FMA in one wave: do the two pipes overlap inside ONE instruction stream?
#pragma unroll
            for (int u = 0; u < UNROLL; u += 4) {
                c[(u >> 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], s, c[(u >> 2) & 3], 0, 0, 0);
                a[u + 1] = __builtin_fmaf(a[u + 1], s, 0.5f);
                a[u + 2] = __builtin_fmaf(a[u + 2], s, 0.5f);
                a[u + 3] = __builtin_fmaf(a[u + 3], s, 0.5f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r += a[u] + p[u].x + p[u].y;
#pragma unroll
    for (int u = 0; u < 4; ++u) r += c[u].x + c[u].y + c[u].z + c[u].w;
    if (r == 123.456f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
int run(const char* name, double flop_per_lane_inst, float* out, unsigned long long* cyc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4, 8}) {                     // waves per SIMD: block = 4 SIMDs x wps waves, one block per CU
        if (wps * 256 > 1024) {                        // 8 waves/SIMD = two 1024-thread blocks per CU
            hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(1024), 0, 0, out, cyc);
        } else {
            hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256 * wps), 0, 0, out, cyc);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (wps * 256 > 1024) hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(1024), 0, 0, out, cyc);
        else hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256 * wps), 0, 0, out, cyc);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        const double insts_per_wave = (double)ITERS * UNROLL;
        const double waves = 256.0 * 4 * wps;
        const double flops = insts_per_wave * waves * 64 * flop_per_lane_inst;
        printf("%-28s %d waves/SIMD: %6.2f cyc per wave-instruction per SIMD (in-kernel clock), %7.1f us, %7.2f TFLOP/s\n", name, wps,
               (double)h / insts_per_wave / wps, ms * 1e3, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    run<0>("v_fma_f32", 2.0, out, cyc);
    run<1>("v_pk_fma_f32", 4.0, out, cyc);
    run<2>("v_exp_f32", 1.0, out, cyc);
    run<3>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4 / 64.0, out, cyc);
    run<4>("1 mfma + 3 v_fma (mixed)", (2.0 * 16 * 16 * 4 / 64.0 + 3 * 2.0) / 4.0, out, cyc);
    return 0;
}
