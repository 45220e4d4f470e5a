// clique_sim.hip — the clique training-batch simulator as ONE kernel.
//
// The reference draws a clique's training batch by ancestral simulation over its factors
// (src/sampler/SimulationBasedSampler.py:14-133 calling the factors' `sample` methods,
// src/factors/Factors.py:725-743,1196-1317,2575-2649,3146-3157,3260-3276): per-sample Python loops over SE2Pose
// objects.  Here the schedule (which factor draws which variable, decided on the host from the graph structure alone)
// is compiled into a short list of ops that every thread interprets for its own sample: one thread = one joint sample,
// columns live in LDS while the list runs, random numbers come from a counter-based generator (Philox4x32-10 keyed by
// the clique's seed, counter = (sample, op)), and the batch leaves the kernel row-major, ready for
// nfisam_normalize_columns.  HBM traffic = the batch itself (4 D bytes per sample) + the children's messages.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/nfisam_hip.h"

namespace {

constexpr int SIM_BLOCK = 256;

struct SimArgs {
    nfisam_sim_op ops[NFISAM_SIM_MAX_OPS];
    int n_ops, n, D_out, D_total;
    unsigned long long seed;
    float* x_out;
};

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// four uniforms in (0,1) for (sample, op) under `seed`
__device__ __forceinline__ void philox4(unsigned long long seed, uint32_t sample, uint32_t op, float (&u)[4]) {
    uint32_t c[4] = {sample, op, 0x9E3779B9u, 0x243F6A88u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = ((float)(c[i] >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

struct Rand {
    float z0, z1, z2, u2, u3;      // three standard normals; u2 / u3 are the uniforms z2 was made from
};
__device__ __forceinline__ Rand draw(unsigned long long seed, uint32_t sample, uint32_t op) {
    float u[4];
    philox4(seed, sample, op, u);
    const float two_pi = 6.283185307179586f;
    const float r0 = sqrtf(-2.0f * logf(u[0])), r1 = sqrtf(-2.0f * logf(u[2]));
    Rand r;
    r.z0 = r0 * cosf(two_pi * u[1]);
    r.z1 = r0 * sinf(two_pi * u[1]);
    r.z2 = r1 * cosf(two_pi * u[3]);
    r.u2 = u[2]; r.u3 = u[3];
    return r;
}

struct Pose { float x, y, t; };
__device__ __forceinline__ float wrap_angle(float t) {
    const float two_pi = 6.283185307179586f, pi = 3.141592653589793f;
    float r = fmodf(t + pi, two_pi);
    if (r < 0.0f) r += two_pi;
    return r - pi;
}
__device__ __forceinline__ Pose se2_exp(float vx, float vy, float w) {
    float a, b;
    if (fabsf(w) < 1e-6f) { a = 1.0f; b = 0.5f * w; } else { a = sinf(w) / w; b = (1.0f - cosf(w)) / w; }
    return Pose{a * vx - b * vy, b * vx + a * vy, wrap_angle(w)};
}
__device__ __forceinline__ Pose compose(const Pose& A, const Pose& B) {
    const float c = cosf(A.t), s = sinf(A.t);
    return Pose{A.x + c * B.x - s * B.y, A.y + s * B.x + c * B.y, wrap_angle(A.t + B.t)};
}
__device__ __forceinline__ Pose inverse(const Pose& A) {
    const float c = cosf(A.t), s = sinf(A.t);
    return Pose{-(c * A.x + s * A.y), -(-s * A.x + c * A.y), wrap_angle(-A.t)};
}
// Exp(L z): tangent-space Gaussian noise, L = lower Cholesky factor stored as l00 l10 l11 l20 l21 l22
__device__ __forceinline__ Pose noise_pose(const float* L, const Rand& r) {
    return se2_exp(L[0] * r.z0, L[1] * r.z0 + L[2] * r.z1, L[3] * r.z0 + L[4] * r.z1 + L[5] * r.z2);
}

__global__ void __launch_bounds__(SIM_BLOCK) nsf_clique_sim_kernel(SimArgs a) {
    extern __shared__ float xs[];                 // [D_total][SIM_BLOCK]
    const int tid = threadIdx.x;
    const int q = blockIdx.x * SIM_BLOCK + tid;
    const bool valid = q < a.n;
    const uint32_t sample = (uint32_t)q;
    auto col = [&](int c) -> float& { return xs[c * SIM_BLOCK + tid]; };
    for (int o = 0; o < a.n_ops; ++o) {
        const nfisam_sim_op& op = a.ops[o];
        switch (op.code) {
        case NFISAM_SIM_COPY: {                   // k columns of a row-major device array (a child's flow message)
            const float* src = (const float*)op.src;
            for (int j = 0; j < op.k; ++j) col(op.c + j) = valid ? src[(size_t)q * op.a + op.b + j] : 0.0f;
            break;
        }
        case NFISAM_SIM_PRIOR_SE2: {              // x = prior * Exp(eps)
            const Pose x = compose(Pose{op.p[0], op.p[1], op.p[2]}, noise_pose(op.p + 3, draw(a.seed, sample, o)));
            col(op.c) = x.x; col(op.c + 1) = x.y; col(op.c + 2) = x.t;
            break;
        }
        case NFISAM_SIM_REL_FWD: {                // T_j = T_i * (obs * Exp(eps))
            const Pose rel = compose(Pose{op.p[0], op.p[1], op.p[2]}, noise_pose(op.p + 3, draw(a.seed, sample, o)));
            const Pose x = compose(Pose{col(op.a), col(op.a + 1), col(op.a + 2)}, rel);
            col(op.c) = x.x; col(op.c + 1) = x.y; col(op.c + 2) = x.t;
            break;
        }
        case NFISAM_SIM_REL_BWD: {                // T_i = T_j * (obs * Exp(eps))^-1
            const Pose rel = compose(Pose{op.p[0], op.p[1], op.p[2]}, noise_pose(op.p + 3, draw(a.seed, sample, o)));
            const Pose x = compose(Pose{col(op.a), col(op.a + 1), col(op.a + 2)}, inverse(rel));
            col(op.c) = x.x; col(op.c + 1) = x.y; col(op.c + 2) = x.t;
            break;
        }
        case NFISAM_SIM_REL_OBS: {                // simulated odometry: (T_i^-1 T_j) * Exp(eps)
            const Pose d = compose(inverse(Pose{col(op.a), col(op.a + 1), col(op.a + 2)}),
                                   Pose{col(op.b), col(op.b + 1), col(op.b + 2)});
            const Pose x = compose(d, noise_pose(op.p + 3, draw(a.seed, sample, o)));
            col(op.c) = x.x; col(op.c + 1) = x.y; col(op.c + 2) = x.t;
            break;
        }
        case NFISAM_SIM_RING: {                   // the other end on a ring of radius obs + N(0, sigma^2), uniform bearing
            const Rand r = draw(a.seed, sample, o);
            const float rad = op.p[0] + op.p[1] * r.z0;
            const float phi = (2.0f * r.u2 - 1.0f) * 3.141592653589793f;
            col(op.c) = col(op.a) + rad * cosf(phi);
            col(op.c + 1) = col(op.a + 1) + rad * sinf(phi);
            break;
        }
        case NFISAM_SIM_RANGE_OBS: {              // simulated range |t_2 - t_1| + N(0, sigma^2)
            const float dx = col(op.b) - col(op.a), dy = col(op.b + 1) - col(op.a + 1);
            col(op.c) = sqrtf(dx * dx + dy * dy) + op.p[0] * draw(a.seed, sample, o).z0;
            break;
        }
        case NFISAM_SIM_ADA_OBS: {                // one simulated range to ONE of k candidate landmarks (weights cumulated in p)
            const Rand r = draw(a.seed, sample, o);
            int pick = op.k - 1;
            for (int j = op.k - 2; j >= 0; --j) if (r.u2 < op.p[j]) pick = j;
            const int cc = op.cand[pick];
            const float dx = col(cc) - col(op.a), dy = col(cc + 1) - col(op.a + 1);
            col(op.c) = sqrtf(dx * dx + dy * dy) + op.p[4] * r.z0;
            break;
        }
        case NFISAM_SIM_NH_RING: {                // ring whose radius noise is the regular or the inflated ("null") sigma
            const Rand r = draw(a.seed, sample, o);
            const float sig = (r.u3 < op.p[3]) ? op.p[1] : op.p[2];
            const float rad = op.p[0] + sig * r.z0;
            const float phi = (2.0f * r.u2 - 1.0f) * 3.141592653589793f;
            col(op.c) = col(op.a) + rad * cosf(phi);
            col(op.c + 1) = col(op.a + 1) + rad * sinf(phi);
            break;
        }
        case NFISAM_SIM_NH_OBS: {                 // simulated range with the regular or the inflated sigma
            const Rand r = draw(a.seed, sample, o);
            const float dx = col(op.b) - col(op.a), dy = col(op.b + 1) - col(op.a + 1);
            col(op.c) = sqrtf(dx * dx + dy * dy) + ((r.u2 < op.p[2]) ? op.p[0] : op.p[1]) * r.z0;
            break;
        }
        case NFISAM_SIM_PRIOR_R2: {               // point prior on the plane: mu + L z
            const Rand r = draw(a.seed, sample, o);
            col(op.c) = op.p[0] + op.p[2] * r.z0;
            col(op.c + 1) = op.p[1] + op.p[3] * r.z0 + op.p[4] * r.z1;
            break;
        }
        case NFISAM_SIM_PRIOR_R2_RING: {          // ring prior around a fixed centre
            const Rand r = draw(a.seed, sample, o);
            const float rad = op.p[2] + op.p[3] * r.z0;
            const float phi = (2.0f * r.u2 - 1.0f) * 3.141592653589793f;
            col(op.c) = op.p[0] + rad * cosf(phi);
            col(op.c + 1) = op.p[1] + rad * sinf(phi);
            break;
        }
        case NFISAM_SIM_REL_R2_FWD: case NFISAM_SIM_REL_R2_BWD: {   // displacement factor, either direction
            const Rand r = draw(a.seed, sample, o);
            const float nx = op.p[2] * r.z0, ny = op.p[3] * r.z0 + op.p[4] * r.z1;
            const float sg = (op.code == NFISAM_SIM_REL_R2_FWD) ? 1.0f : -1.0f;
            col(op.c) = col(op.a) + sg * (op.p[0] + nx);
            col(op.c + 1) = col(op.a + 1) + sg * (op.p[1] + ny);
            break;
        }
        case NFISAM_SIM_REL_R2_OBS: {             // simulated displacement: x_2 - x_1 + noise
            const Rand r = draw(a.seed, sample, o);
            col(op.c) = col(op.b) - col(op.a) + op.p[2] * r.z0;
            col(op.c + 1) = col(op.b + 1) - col(op.a + 1) + op.p[3] * r.z0 + op.p[4] * r.z1;
            break;
        }
        default: break;
        }
    }
    __syncthreads();
    // row-major store, coalesced: element e of this block's 256 x D_out patch
    const int rows = (a.n - blockIdx.x * SIM_BLOCK) < SIM_BLOCK ? (a.n - blockIdx.x * SIM_BLOCK) : SIM_BLOCK;
    float* dst = a.x_out + (size_t)blockIdx.x * SIM_BLOCK * a.D_out;
    for (int e = tid; e < rows * a.D_out; e += SIM_BLOCK) {
        const int r = e / a.D_out, c = e - r * a.D_out;
        dst[e] = xs[c * SIM_BLOCK + r];
    }
}

}  // namespace

extern "C" int nfisam_simulate_clique(const nfisam_sim_op* ops, int n_ops, int n, int D_out, int D_total,
                                      uint64_t seed, float* x_out, nfisam_stream_t stream) {
    if (ops == nullptr || n_ops < 1 || n_ops > NFISAM_SIM_MAX_OPS || n < 1 || D_out < 1 || D_total < D_out ||
        x_out == nullptr || (size_t)D_total * SIM_BLOCK * sizeof(float) > 150 * 1024)
        return NFISAM_ERR_ARG;
    for (int o = 0; o < n_ops; ++o) {
        const nfisam_sim_op& op = ops[o];
        const int w = (op.code == NFISAM_SIM_COPY) ? op.k : ((op.code == NFISAM_SIM_RING || op.code == NFISAM_SIM_NH_RING ||
                                                              op.code >= NFISAM_SIM_PRIOR_R2) ? 2 :
                      ((op.code == NFISAM_SIM_RANGE_OBS || op.code == NFISAM_SIM_ADA_OBS || op.code == NFISAM_SIM_NH_OBS) ? 1 : 3));
        if (op.code < NFISAM_SIM_COPY || op.code > NFISAM_SIM_REL_R2_OBS || op.c < 0 || op.c + w > D_total) return NFISAM_ERR_ARG;
        if (op.code == NFISAM_SIM_COPY && (op.src == 0 || op.k < 1)) return NFISAM_ERR_ARG;
        if (op.code == NFISAM_SIM_ADA_OBS && (op.k < 1 || op.k > 4)) return NFISAM_ERR_ARG;
    }
    SimArgs a;
    memset(&a, 0, sizeof(a));
    memcpy(a.ops, ops, sizeof(nfisam_sim_op) * (size_t)n_ops);
    a.n_ops = n_ops; a.n = n; a.D_out = D_out; a.D_total = D_total; a.seed = seed; a.x_out = x_out;
    const size_t lds = (size_t)D_total * SIM_BLOCK * sizeof(float);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)nsf_clique_sim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return NFISAM_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(nsf_clique_sim_kernel, dim3((n + SIM_BLOCK - 1) / SIM_BLOCK), dim3(SIM_BLOCK), lds,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NFISAM_OK : NFISAM_ERR_LAUNCH;
}
