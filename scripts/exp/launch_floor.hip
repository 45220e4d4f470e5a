// Experiment (GPU): floor of two dependent kernels per iteration inside a hipGraph on MI355X, for the grid shapes of the
// single-clique training iteration (945 one-wave blocks + 210 x 256-thread blocks), with empty bodies and with bodies that
// do one dependent global-memory round trip each.   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)
__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 0; }
__global__ void k_rt(float* p, int n) {            // read one line written by the previous kernel, write one
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
    p[i] = p[(i * 97 + 13) % n] + 1.0f;
}
static int run(const char* name, int gridA, int blkA, int gridB, int blkB, int mode, float* buf, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int it = 0; it < 200; ++it) {
        if (mode == 0) { hipLaunchKernelGGL(k_empty, dim3(gridA), dim3(blkA), 0, s, buf); hipLaunchKernelGGL(k_empty, dim3(gridB), dim3(blkB), 0, s, buf); }
        else { hipLaunchKernelGGL(k_rt, dim3(gridA), dim3(blkA), 0, s, buf, 1 << 16); hipLaunchKernelGGL(k_rt, dim3(gridB), dim3(blkB), 0, s, buf, 1 << 16); }
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %.2f us per pair of kernels\n", name, ms * 1e3 / 1000);
    return 0;
}
int main() {
    float* buf; CK(hipMalloc(&buf, 4 << 16)); CK(hipMemset(buf, 0, 4 << 16));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (run("empty: 945x64 + 210x256", 945, 64, 210, 256, 0, buf, s)) return 1;
    if (run("empty: 1x64 + 1x64", 1, 64, 1, 64, 0, buf, s)) return 1;
    if (run("one round trip each: 945x64 + 210x256", 945, 64, 210, 256, 1, buf, s)) return 1;
    if (run("one round trip each: 237x256 + 64x256", 237, 256, 64, 256, 1, buf, s)) return 1;
    if (run("empty: 128x384 + 256x256 (C2)", 128, 384, 256, 256, 0, buf, s)) return 1;
    return 0;
}
