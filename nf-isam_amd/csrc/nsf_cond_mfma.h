// The conditioner on the matrix cores WITHOUT leaving the particle-per-lane layout (dim-major training kernel).
//
// v_mfma_f32_4x4x1_16b_f32 runs sixteen independent 4x4 outer products, one per block of four lanes:
//     D[r](lane) = C[r](lane) + A(lane 4*(lane/4) + r) * B(lane)          (checked on gfx950: scripts/exp/mfma4x4.hip)
// With B = the lane's OWN activation and A = the weights of four output rows parked on the four lanes of every block,
// one instruction is four FMAs per lane, exact fp32, and its result registers are already "particle on the lane, four
// consecutive outputs in the four registers" -- the layout the spline needs.  The mat-vec products
//     a1 = W0^T x[:i] + b0,   a2 = W1^T h1 + b1,   theta = W2^T h2 + b2            (src/flows/flows.py:26-41, 82-83)
// and their transposes in the backward pass become chains of these instructions fed by 16-byte LDS reads of a weight
// PANEL (one read = the A operands of four instructions); biases initialise the accumulators.
// Against the scalar-path FMAs this removes every `s_waitcnt lgkmcnt(0)` in front of a weight row (~25 exposed
// scalar-cache round trips per unit), the ~100 weight SGPRs and their spills, and 3/4 of the instruction slots.
// (It does not buy parallel issue: on gfx950 MFMA and VALU instructions of a SIMD do not overlap -- 8 x 16x16x4 + 32
// v_fma per round take the SUM of their separate times, scripts/exp/mfma4x4.hip -- a 4x4x1 costs ~9.5 cycles, the four
// v_fma it replaces ~10.5.)
//
// Panel (floats, per block = one (clique, dim); staged once per launch by the block's waves), ST = H | 4:
//   W2T [PoP][ST]   W2T[o][k] = W2[k][o]            forward, layer 2          b2 [PoP]
//   W1T [H][ST]     W1T[j][k] = W1[k][j]            forward, layer 1          b1 [H]
//   W2N [H][PoP+4]  W2N[k][o] = W2[k][o]            backward, layer 2
//   W1N [H][ST]     W1N[k][j] = W1[k][j]            backward, layer 1
//   W0T [H][s0]     W0T[j][k] = W0[k][j], zero for k >= i; s0 = 8*ceil(i/8) + 4     forward, layer 0      b0 [H]
// Row strides are odd multiples of 4 floats: 16-byte reads, and the four rows one read touches (lane & 3) fall into
// different banks.
#pragma once
#include "nsf_device.h"
#include "nsf_host.h"
#ifndef NSF_COND_SPLIT
#define NSF_COND_SPLIT 1
#endif

typedef float cm_f32x4 __attribute__((ext_vector_type(4)));

// diagnostic build (-DNSF_STAMPS=3, unit 0): cycle sums of the staging's sub-phases, thread 64 of block (1, 0, 0)
#if defined(NSF_STAMPS) && NSF_STAMPS == 3 && defined(NSF_UNIT) && NSF_UNIT == 0
__device__ unsigned long long g_stg[32];
#define STG_T0() unsigned long long stg_prev_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stg_prev_)::"memory")
#define STG_STAMP(id, v)                                                                                   \
    do {                                                                                                   \
        unsigned long long t_;                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(v)::"memory");                \
        if (tid == 64 && blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0) { g_stg[id] += t_ - stg_prev_; g_stg[16 + (id)] += 1ull; } \
        stg_prev_ = t_;                                                                                    \
    } while (0)
#else
#define STG_T0()
#define STG_STAMP(id, v)
#endif

__device__ __forceinline__ cm_f32x4 mfma1(float a, float b, cm_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

template <int K, int H>
struct CondPanel {
    using LY = Layout<K, H>;
    static constexpr int PoP = LY::PoP;
    static constexpr int ST = H | 4;
    static constexpr int GH = H / 4, G2 = PoP / 4;
    static constexpr int NS2 = PoP + 4;
    static constexpr int oW2T = 0;
    static constexpr int ob2 = oW2T + PoP * ST;
    static constexpr int oW1T = ob2 + PoP;
    static constexpr int ob1 = oW1T + H * ST;
    static constexpr int oW2N = ob1 + H;
    static constexpr int oW1N = oW2N + H * NS2;
    static constexpr int ob0 = oW1N + H * ST;
    static constexpr int oW0T = ob0 + H;
    static_assert(H % 4 == 0 && PoP % 8 == 0, "output rows come in groups of four");
    __host__ __device__ static constexpr int s0_of(int i) { return ((i + 7) & ~7) + 4; }
    __host__ __device__ static constexpr int floats(int max_D) { return oW0T + H * s0_of(max_D > 1 ? max_D - 1 : 1); }
};
// theta column o is a spline parameter (not layout padding): nsf_device.h Layout::iw / ih / idv
template <int K, int H>
struct ThetaColUsed {
    using LY = Layout<K, H>;
    __host__ __device__ static constexpr bool at(int o) { return (o % LY::HP) < K + (o < LY::HP ? LY::ND0 : LY::ND1); }
};
struct EveryCol {
    __host__ __device__ static constexpr bool at(int) { return true; }
};

// The previous iteration's Adam update, applied on the way into the panel ("fused Adam").
// A kernel boundary costs ~3 us on this chip (end-of-kernel L2 write-back, dispatch, cache invalidation) and a
// device-wide hand-over inside a kernel costs the same (agent-scope release = L2 write-back across eight XCDs), so a
// separate Adam launch per iteration is a third of a latency-bound iteration.  In the dim-major kernel a (clique, dim)
// block of parameters is read by that dim's blocks only: each of them re-derives the updated block from the previous
// launch's gradient copies (same summation order and arithmetic as nsf_adam_kernel: bit-identical), stages it, and
// the dim's first block also writes the new parameters / moments to the OTHER of two state buffers (nobody may
// overwrite what late-starting blocks of the same launch still read; the gradient copies alternate likewise).
struct FusedAdam {
    const __attribute__((address_space(1))) float* grads;   // previous launch's gradient copies (copy 0), or nullptr: nothing pending
    size_t gstride;                                          // floats between copies
    int copies;                                              // copies of this clique
    const __attribute__((address_space(1))) float *m_src, *v_src;
    __attribute__((address_space(1))) float *t_dst, *m_dst, *v_dst;   // written by the dim's first block only (else nullptr)
    AdamCoef kc;
};

// sum of parameter j's gradient copies, in nsf_adam_kernel's one-thread-per-parameter order (at most 8 copies: the host checks)
// COH = true was round 3's in-launch exchange (agent-scope loads / stores of words another block of the launch wrote); the
// chunk-persistent kernel exchanges tagged pairs since round 4 (stage_cond_panel_persist*), so only COH = false is
// instantiated: the one-launch-per-iteration kernel reads what the PREVIOUS launch wrote.
template <bool COH>
__device__ __forceinline__ float group_load(const __attribute__((address_space(1))) float* p) {
    if constexpr (COH) return __hip_atomic_load((const float*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void group_store(__attribute__((address_space(1))) float* p, float v) {
    if constexpr (COH) __hip_atomic_store((float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH = false>
__device__ __forceinline__ void fused_load_grads(const FusedAdam& fa, int j, float (&gv)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) gv[c] = group_load<COH>(&fa.grads[(size_t)(c < fa.copies ? c : 0) * fa.gstride + j]);
}
__device__ __forceinline__ float fused_sum_grads(const FusedAdam& fa, const float (&gv)[8]) {
    float gs = gv[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) gs += (c < fa.copies) ? gv[c] : 0.0f;
    return gs;
}

// Where parameter jj of dim i's kernel-layout block goes in the panel: (first word) | (second word) << 16, LDS words
// counted from the start of the block's dynamic LDS (the panel starts PANEL_BASE words in; word 0 is a dump for the
// "second store" of parameters that have one destination).  Host-built once per (K, H) for dims 0 .. max_i and read by
// the staging threads next to the parameter itself: the index arithmetic (two divisions and a six-way branch per
// parameter) costs a lone wave ~2k cycles when done in the kernel.
constexpr uint32_t PANEL_SCALED = 0x8000u;          // panel map: the first destination takes the parameter x 2 log2(e)
constexpr float kTanhScale = 2.0f * 1.4426950408889634f;
// tanh(a) from a' = 2 log2(e) a
__device__ __forceinline__ float ftanh_scaled(float as) { return 1.0f - 2.0f * frcp(1.0f + __builtin_amdgcn_exp2f(as)); }

template <int K, int H>
static inline void build_panel_map(uint32_t* map, int max_D) {
    using CP = CondPanel<K, H>;
    using LY = Layout<K, H>;
    constexpr int PoP = CP::PoP, ST = CP::ST;
    for (int j = 0; j < PoP; ++j) map[j] = (uint32_t)(PANEL_BASE + j);           // dim 0: the spline parameters themselves
    for (int i = 1; i < max_D; ++i) {
        const int s0 = CP::s0_of(i);
        uint32_t* m = map + LY::off(i);
        for (int jj = 0; jj < LY::block(i); ++jj) {
            int d0 = 0, d1 = -PANEL_BASE;
            if (jj < LY::ob0(i)) {
                const int k = jj / H, j = jj - k * H;
                d0 = CP::oW0T + j * s0 + k;
            } else if (jj < LY::oW1(i)) {
                d0 = CP::ob0 + (jj - LY::ob0(i));
            } else if (jj < LY::ob1(i)) {
                const int e = jj - LY::oW1(i), k = e / H, j = e - k * H;
                d0 = CP::oW1T + j * ST + k;
                d1 = CP::oW1N + k * ST + j;
            } else if (jj < LY::oW2(i)) {
                d0 = CP::ob1 + (jj - LY::ob1(i));
            } else if (jj < LY::ob2(i)) {
                const int e = jj - LY::oW2(i), k = e / PoP, o = e - k * PoP;
                d0 = CP::oW2T + o * ST + k;
                d1 = CP::oW2N + k * CP::NS2 + o;
            } else {
                d0 = CP::ob2 + (jj - LY::ob2(i));
            }
            // forward copies of layers 0 and 1 (W0T, b0, W1T, b1) are staged pre-multiplied by 2 log2(e): tanh of the
            // pre-activation is then 1 - 2 / (1 + exp2(a')) without a multiplication per hidden unit and particle
            const bool scaled = jj < LY::oW2(i);
            m[jj] = (uint32_t)(PANEL_BASE + d0) | (scaled ? PANEL_SCALED : 0u) | ((uint32_t)(PANEL_BASE + d1) << 16);
        }
    }
}

// The block's threads bring dim i's parameter block (i > 0: conditioner weights -> panel; i == 0: the PoP spline
// parameters -> pan[0..PoP)) into LDS; a workgroup barrier follows at the call site.  Thread t owns parameters
// t, t + NT, ... of the block: two at a time, every global load of both (parameter, panel word, and with an update
// pending the gradient copies and moments) issued before the first use.  Padding no 16-byte read ever touches is not
// written; the zero weights behind W0's i rows are (layer 0 contracts whole groups of eight inputs).
// `st_step`, `st_stop` are the clique's state words, requested at kernel entry: they are first LOOKED AT after this
// function's own loads have arrived (the empty asm is a use of the loaded values in front of the branch, so the loads are
// not sunk behind it), i.e. the two round trips overlap.  -> false: the clique is finished, the block returns.
template <int K, int H, bool COH = false>
__device__ __forceinline__ bool stage_cond_panel(float* lds0, const float* theta_generic, FusedAdam& fa, const uint32_t* map_generic,
                                                 int i, int tid, int NT, int st_step, int st_stop, const TrainArgs& a, int n, int iter) {
    using CP = CondPanel<K, H>;
    using LY = Layout<K, H>;
    typedef const __attribute__((address_space(1))) float* gp;
    typedef const __attribute__((address_space(1))) uint32_t* gu;
    gp t_src = (gp)theta_generic;                              // the clique's parameter vector (before the pending update)
    constexpr int PoP = CP::PoP;                               // NT = threads of the block
    const int j0 = (i == 0) ? 0 : LY::off(i), nj = (i == 0) ? PoP : LY::block(i);
    gu map = (gu)map_generic + j0;
    for (int base = 0; base < nj; base += 2 * NT) {
        const int ja = base + tid, jb = base + NT + tid;
        const int ca = (ja < nj ? ja : 0), cb = (jb < nj ? jb : 0);
        const int ia = j0 + ca, ib = j0 + cb;
        float ta = group_load<COH>(&t_src[ia]), tb = group_load<COH>(&t_src[ib]);
        uint32_t da = map[ca], db = map[cb];
        float ga[8], gb[8], ma = 0.f, va = 0.f, mb = 0.f, vb = 0.f;
        if (fa.grads != nullptr) {                             // launch-uniform: an update is pending
            fused_load_grads<COH>(fa, ia, ga);
            fused_load_grads<COH>(fa, ib, gb);
            ma = group_load<COH>(&fa.m_src[ia]); va = group_load<COH>(&fa.v_src[ia]);
            mb = group_load<COH>(&fa.m_src[ib]); vb = group_load<COH>(&fa.v_src[ib]);
            asm volatile("" : "+v"(ga[0]), "+v"(ga[7]), "+v"(gb[0]), "+v"(gb[7]), "+v"(ma), "+v"(vb));
        }
        asm volatile("" : "+v"(ta), "+v"(tb), "+v"(da), "+v"(db));
        if (base == 0) {
            if (st_stop != 0 || st_step + iter >= a.max_iters) return false;     // block-uniform
            if (fa.grads != nullptr)
                fa.kc = adam_coef(a.adam.lr, a.adam.beta1, a.adam.beta2, a.adam.eps, a.log_b1, a.log_b2, st_step + iter, n);
        }
        if (fa.grads != nullptr) {
            adam_update(fa.kc, fused_sum_grads(fa, ga), ma, va, ta);
            adam_update(fa.kc, fused_sum_grads(fa, gb), mb, vb, tb);
            if (fa.t_dst != nullptr) {                         // the dim's first block records the new state
                // (COH: read by the group's other blocks in the next iteration of the same launch: agent-scope stores)
                if (ja < nj) { group_store<COH>(&fa.t_dst[ia], ta); group_store<COH>(&fa.m_dst[ia], ma); group_store<COH>(&fa.v_dst[ia], va); }
                if (jb < nj) { group_store<COH>(&fa.t_dst[ib], tb); group_store<COH>(&fa.m_dst[ib], mb); group_store<COH>(&fa.v_dst[ib], vb); }
            }
        }
        if (ja < nj) { lds0[da & 0x7fffu] = (da & PANEL_SCALED) ? ta * kTanhScale : ta; lds0[da >> 16] = ta; }
        if (jb < nj) { lds0[db & 0x7fffu] = (db & PANEL_SCALED) ? tb * kTanhScale : tb; lds0[db >> 16] = tb; }
    }
    if (i > 0) {
        const int s0 = CP::s0_of(i);
        const int npad = (((i + 7) & ~7) - i) * H;             // zero weights behind W0's rows
        for (int e = tid; e < npad; e += NT) {
            const int k = i + e / H, j = e % H;
            lds0[PANEL_BASE + CP::oW0T + j * s0 + k] = 0.0f;
        }
    }
    return true;
}

// ---- the chunk-persistent form's staging (round 4): flag-in-data exchange instead of a barrier ------------------------------
// Between two iterations of ONE launch the blocks of a (clique, dim) group have to see each other's gradient copies.  Round
// 3 did that with store -> wait for the acknowledgement -> arrive at a counter -> poll the counter -> load the copies: four
// dependent memory round trips, 8.5 k of a lone Plaza wave's 20.7 k cycles per iteration (13.2 k of 28.2 k on C3).  Now every
// gradient word travels WITH its flag: a copy is written as 8-byte pairs (value, tag), tag = the iteration's number in the
// run (state->step + it + 1, never 0 and never repeated for a buffer while the workspace lives: a re-used plan zeroes it), by
// agent-scope write-through stores that nobody waits for, and the reader simply loads the pairs of all copies (8-byte
// agent-scope loads: a pair is written and read in one piece) until every tag is the expected one.  On the critical path
// that is the writer's store reaching L2 and the reader's load: no counter, no acknowledgement, no workgroup barrier beyond
// the one the staging needs anyway.  The copies still alternate between two buffers with the iteration's parity: a block
// can produce copy k + 2 only after it has seen everybody's copy k + 1, i.e. after everybody has finished reading copy k.
// theta, m and v of the dim never leave the block between iterations: every block of the group computes the same update from
// the same sums (bit for bit: one function, one order), so each keeps its own set in LDS; the dim's first block still
// records them in the clique's buffers (plain stores: their readers are the NEXT kernels).
struct PersistAdam {
    const __attribute__((address_space(1))) float* tagged;   // previous iteration's tagged copies (copy 0), or nullptr: nothing pending (first iteration of the chunk)
    size_t cstride;                                          // floats between tagged copies (2 per parameter)
    int copies;
    uint32_t tag;                                            // the tag the previous iteration's copies carry
    const __attribute__((address_space(1))) float *m_src, *v_src;     // first iteration of the chunk: the clique's moments
    __attribute__((address_space(1))) float *t_dst, *m_dst, *v_dst;   // written by the dim's first block only (else nullptr)
    unsigned looks;                                          // (diagnostic builds: unsuccessful looks of this thread, summed over the call)
    float* keep;                                             // LDS: theta | m | v of this dim, `kstride` floats each
    int kstride;
    unsigned* ctr;                                           // the dim's control word: bit 31 = the group's abort flag
    int spin_log2;
    int max_iters;                                           // (the launch's arguments the staging needs, by value)
    float lr, beta1, beta2, eps, log_b1, log_b2;
    AdamCoef kc;
};

// One (value, tag) pair of copy c: a 64-bit scalar base per copy + a 32-bit lane offset (the `saddr + voffset` form of
// global_load_dwordx2: no 64-bit vector address arithmetic per load).  `cs_bytes` = bytes between copies (< 2^31 / copies:
// a clique's whole workspace is a few MB); copies beyond the group's size re-read its last one (their values are not used,
// their tags are this iteration's as soon as that copy's are).
__device__ __forceinline__ unsigned long long tagged_pair(const __attribute__((address_space(1))) float* tagged, int c, int copies,
                                                          unsigned cs_bytes, unsigned lane_off) {
    typedef const __attribute__((address_space(1))) char* gc;
    typedef const __attribute__((address_space(1))) unsigned long long* gq;
    const unsigned cc = (unsigned)(c < copies ? c : copies - 1);
    gc base = (gc)tagged + (size_t)(cc * cs_bytes);
    return __hip_atomic_load((const unsigned long long*)(gq)(base + (size_t)lane_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// -> 0: staged; 1: the clique is finished (the block returns); 2: gave up waiting for a copy (the group's abort flag is up)
template <int K, int H>
__device__ __forceinline__ int stage_cond_panel_persist(float* lds0, const float* theta_generic, PersistAdam& fa, const uint32_t* map_generic,
                                                        int i, int tid, int NT, int st_step, int st_stop, int n, int iter) {
    using CP = CondPanel<K, H>;
    using LY = Layout<K, H>;
    typedef const __attribute__((address_space(1))) float* gp;
    typedef const __attribute__((address_space(1))) uint32_t* gu;
    gp t_src = (gp)theta_generic;
    constexpr int PoP = CP::PoP;
    const int j0 = (i == 0) ? 0 : LY::off(i), nj = (i == 0) ? PoP : LY::block(i);
    gu map = (gu)map_generic + j0;
    const bool pending = fa.tagged != nullptr;
    int gave_up = 0;
    // (looked at BEFORE anything is waited for: nobody writes the copies of a finished clique)
    if (st_stop != 0 || st_step + iter >= fa.max_iters) return 1;    // block-uniform
    STG_T0();
    for (int base = 0; base < nj; base += 2 * NT) {
        const int ja = base + tid, jb = base + NT + tid;
        const int ca = (ja < nj ? ja : 0), cb = (jb < nj ? jb : 0);
        const int ia = j0 + ca, ib = j0 + cb;
        uint32_t da = map[ca], db = map[cb];
        float ta, tb, ma, va, mb, vb;
        float ga[8], gb[8];
        if (!pending) {                                        // launch-uniform per iteration: the chunk's first iteration
            ta = t_src[ia]; tb = t_src[ib];
            ma = fa.m_src[ia]; va = fa.v_src[ia];
            mb = fa.m_src[ib]; vb = fa.v_src[ib];
            asm volatile("" : "+v"(ta), "+v"(tb), "+v"(ma), "+v"(va), "+v"(mb), "+v"(vb), "+v"(da), "+v"(db));
        } else {
            ta = fa.keep[ca]; ma = fa.keep[fa.kstride + ca]; va = fa.keep[2 * fa.kstride + ca];
            tb = fa.keep[cb]; mb = fa.keep[fa.kstride + cb]; vb = fa.keep[2 * fa.kstride + cb];
            // the copies' (value, tag) pairs of this thread's two parameters: all sixteen loads in flight, again until every
            // tag is this iteration's (a copy whose block is still computing shows the tag of two iterations ago, or 0)
            unsigned spins = 0;
            for (;;) {
                unsigned long long qa[8], qb[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    qa[c] = tagged_pair(fa.tagged, c, fa.copies, (unsigned)fa.cstride * 4u, 8u * (unsigned)ia);
                    qb[c] = tagged_pair(fa.tagged, c, fa.copies, (unsigned)fa.cstride * 4u, 8u * (unsigned)ib);
                }
                // (the iteration's bias corrections while the loads are under way: ~80 instructions that depend on nothing loaded)
                if (base == 0 && spins == 0u)
                    fa.kc = adam_coef(fa.lr, fa.beta1, fa.beta2, fa.eps, fa.log_b1, fa.log_b2, st_step + iter, n);
                if (spins == 0u) STG_STAMP(0, fa.kc.step_size);
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    ok = ok && (uint32_t)(qa[c] >> 32) == fa.tag && (uint32_t)(qb[c] >> 32) == fa.tag;
                    ga[c] = __uint_as_float((uint32_t)qa[c]);
                    gb[c] = __uint_as_float((uint32_t)qb[c]);
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
#if defined(NSF_STAMPS)
                ++fa.looks;
#endif
                // every 64 looks: has a member of the group given up?  after 2^spin_log2 looks: give up (and say so)
                if ((spins & 63u) == 0u) {
                    const bool timeout = spins > (1u << fa.spin_log2);
                    if (timeout) __hip_atomic_fetch_or(fa.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (timeout || (__hip_atomic_load(fa.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u) != 0u) { gave_up = 1; break; }
                }
            }
        }
        STG_STAMP(1, ga[0]);
        if (pending) {
            FusedAdam sum_order;                               // (the summation order of nsf_adam_kernel / stage_cond_panel)
            sum_order.copies = fa.copies;
            adam_update(fa.kc, fused_sum_grads(sum_order, ga), ma, va, ta);
            adam_update(fa.kc, fused_sum_grads(sum_order, gb), mb, vb, tb);
            if (fa.t_dst != nullptr) {                         // the dim's first block records the new state (read by later KERNELS)
                if (ja < nj) { fa.t_dst[ia] = ta; fa.m_dst[ia] = ma; fa.v_dst[ia] = va; }
                if (jb < nj) { fa.t_dst[ib] = tb; fa.m_dst[ib] = mb; fa.v_dst[ib] = vb; }
            }
        }
        STG_STAMP(2, ta);
        if (ja < nj) {
            fa.keep[ca] = ta; fa.keep[fa.kstride + ca] = ma; fa.keep[2 * fa.kstride + ca] = va;
            lds0[da & 0x7fffu] = (da & PANEL_SCALED) ? ta * kTanhScale : ta; lds0[da >> 16] = ta;
        }
        if (jb < nj) {
            fa.keep[cb] = tb; fa.keep[fa.kstride + cb] = mb; fa.keep[2 * fa.kstride + cb] = vb;
            lds0[db & 0x7fffu] = (db & PANEL_SCALED) ? tb * kTanhScale : tb; lds0[db >> 16] = tb;
        }
    }
    if (i > 0 && !pending) {                                   // (the zero weights behind W0's rows stay: the panel is the block's for the whole chunk)
        const int s0 = CP::s0_of(i);
        const int npad = (((i + 7) & ~7) - i) * H;
        for (int e = tid; e < npad; e += NT) {
            const int k = i + e / H, j = e % H;
            lds0[PANEL_BASE + CP::oW0T + j * s0 + k] = 0.0f;
        }
    }
    { float d_ = lds0[4]; STG_STAMP(3, d_); }
    return gave_up ? 2 : 0;
}

// The same staging, ONE parameter per thread (round 6: blocks launched with four helper waves -- nsf_train1_kernel, two-wave
// builds of a lone clique: the block has a CU to itself, so waves 4 .. 7 cost nothing while waves 0 .. 3 compute and halve every
// thread's share of the exchange, which is bound by its instructions; up to eight copies).  The same loads, the same sums in the
// same order, the same update per parameter: which thread applies it does not show in the bits.
template <int K, int H>
__device__ __forceinline__ int stage_cond_panel_persist_solo(float* lds0, const float* theta_generic, PersistAdam& fa, const uint32_t* map_generic,
                                                             int i, int tid, int NT, int st_step, int st_stop, int n, int iter) {
    using CP = CondPanel<K, H>;
    using LY = Layout<K, H>;
    typedef const __attribute__((address_space(1))) float* gp;
    typedef const __attribute__((address_space(1))) uint32_t* gu;
    gp t_src = (gp)theta_generic;
    constexpr int PoP = CP::PoP;
    const int j0 = (i == 0) ? 0 : LY::off(i), nj = (i == 0) ? PoP : LY::block(i);
    gu map = (gu)map_generic + j0;
    const bool pending = fa.tagged != nullptr;
    int gave_up = 0;
    if (st_stop != 0 || st_step + iter >= fa.max_iters) return 1;    // block-uniform
    STG_T0();
    for (int base = 0; base < nj; base += NT) {
        const int ja = base + tid;
        const int ca = (ja < nj ? ja : 0);
        const int ia = j0 + ca;
        uint32_t da = map[ca];
        float ta, ma, va;
        float ga[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!pending) {
            ta = t_src[ia];
            ma = fa.m_src[ia]; va = fa.v_src[ia];
            asm volatile("" : "+v"(ta), "+v"(ma), "+v"(va), "+v"(da));
        } else {
            ta = fa.keep[ca]; ma = fa.keep[fa.kstride + ca]; va = fa.keep[2 * fa.kstride + ca];
            unsigned spins = 0;
            for (;;) {
                unsigned long long qa[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) qa[c] = tagged_pair(fa.tagged, c, fa.copies, (unsigned)fa.cstride * 4u, 8u * (unsigned)ia);
                if (base == 0 && spins == 0u)
                    fa.kc = adam_coef(fa.lr, fa.beta1, fa.beta2, fa.eps, fa.log_b1, fa.log_b2, st_step + iter, n);
                if (spins == 0u) STG_STAMP(0, fa.kc.step_size);
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    ok = ok && (uint32_t)(qa[c] >> 32) == fa.tag;
                    ga[c] = __uint_as_float((uint32_t)qa[c]);
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
#if defined(NSF_STAMPS)
                ++fa.looks;
#endif
                if ((spins & 63u) == 0u) {
                    const bool timeout = spins > (1u << fa.spin_log2);
                    if (timeout) __hip_atomic_fetch_or(fa.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (timeout || (__hip_atomic_load(fa.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u) != 0u) { gave_up = 1; break; }
                }
            }
        }
        STG_STAMP(1, ga[0]);
        if (pending && !gave_up) {
            FusedAdam sum_order;                               // (the summation order of nsf_adam_kernel / stage_cond_panel)
            sum_order.copies = fa.copies;
            adam_update(fa.kc, fused_sum_grads(sum_order, ga), ma, va, ta);
            if (fa.t_dst != nullptr && ja < nj) { fa.t_dst[ia] = ta; fa.m_dst[ia] = ma; fa.v_dst[ia] = va; }
        }
        STG_STAMP(2, ta);
        if (ja < nj) {
            fa.keep[ca] = ta; fa.keep[fa.kstride + ca] = ma; fa.keep[2 * fa.kstride + ca] = va;
            lds0[da & 0x7fffu] = (da & PANEL_SCALED) ? ta * kTanhScale : ta; lds0[da >> 16] = ta;
        }
    }
    if (i > 0 && !pending) {
        const int s0 = CP::s0_of(i);
        const int npad = (((i + 7) & ~7) - i) * H;
        for (int e = tid; e < npad; e += NT) {
            const int k = i + e / H, j = e % H;
            lds0[PANEL_BASE + CP::oW0T + j * s0 + k] = 0.0f;
        }
    }
    { float d_ = lds0[4]; STG_STAMP(3, d_); }
    return gave_up ? 2 : 0;
}

// ... and its form for groups of nine to sixteen blocks: two passes, nsf_adam_kernel's lane-partial order (see stage_cond_panel_persist_wide)
template <int K, int H>
__device__ __forceinline__ int stage_cond_panel_persist_solo_wide(float* lds0, const float* theta_generic, PersistAdam& fa, const uint32_t* map_generic,
                                                                  int i, int tid, int NT, int st_step, int st_stop, int n, int iter) {
    using CP = CondPanel<K, H>;
    using LY = Layout<K, H>;
    typedef const __attribute__((address_space(1))) float* gp;
    typedef const __attribute__((address_space(1))) uint32_t* gu;
    gp t_src = (gp)theta_generic;
    constexpr int PoP = CP::PoP;
    const int j0 = (i == 0) ? 0 : LY::off(i), nj = (i == 0) ? PoP : LY::block(i);
    gu map = (gu)map_generic + j0;
    const bool pending = fa.tagged != nullptr;
    int gave_up = 0;
    if (st_stop != 0 || st_step + iter >= fa.max_iters) return 1;    // block-uniform
    STG_T0();
    for (int base = 0; base < nj; base += NT) {
        const int ja = base + tid;
        const int ca = (ja < nj ? ja : 0);
        const int ia = j0 + ca;
        uint32_t da = map[ca];
        float ta, ma, va;
        float ga[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!pending) {
            ta = t_src[ia];
            ma = fa.m_src[ia]; va = fa.v_src[ia];
            asm volatile("" : "+v"(ta), "+v"(ma), "+v"(va), "+v"(da));
        } else {
            ta = fa.keep[ca]; ma = fa.keep[fa.kstride + ca]; va = fa.keep[2 * fa.kstride + ca];
            const int passes = fa.copies > 8 ? 2 : 1;          // block-uniform
            for (int pass = 0; pass < passes && !gave_up; ++pass) {
                const int c0 = 8 * pass;
                unsigned spins = 0;
                for (;;) {
                    unsigned long long qa[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) qa[c] = tagged_pair(fa.tagged, c0 + c, fa.copies, (unsigned)fa.cstride * 4u, 8u * (unsigned)ia);
                    if (base == 0 && spins == 0u && pass == 0)
                        fa.kc = adam_coef(fa.lr, fa.beta1, fa.beta2, fa.eps, fa.log_b1, fa.log_b2, st_step + iter, n);
                    if (spins == 0u && pass == 0) STG_STAMP(0, fa.kc.step_size);
                    bool ok = true;
                    float la[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        ok = ok && (uint32_t)(qa[c] >> 32) == fa.tag;
                        la[c] = __uint_as_float((uint32_t)qa[c]);
                    }
                    if (ok) {
                        if (pass == 0) {
#pragma unroll
                            for (int c = 0; c < 8; ++c) ga[c] = la[c];
                        } else {
#pragma unroll
                            for (int c = 0; c < 8; ++c)
                                if (8 + c < fa.copies) ga[c] += la[c];
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
#if defined(NSF_STAMPS)
                    ++fa.looks;
#endif
                    if ((spins & 63u) == 0u) {
                        const bool timeout = spins > (1u << fa.spin_log2);
                        if (timeout) __hip_atomic_fetch_or(fa.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (timeout || (__hip_atomic_load(fa.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u) != 0u) { gave_up = 1; break; }
                    }
                }
            }
        }
        STG_STAMP(1, ga[0]);
        if (pending && !gave_up) {
            FusedAdam sum_order;
            sum_order.copies = fa.copies < 8 ? fa.copies : 8;  // (more than eight copies: the eight lane partials)
            adam_update(fa.kc, fused_sum_grads(sum_order, ga), ma, va, ta);
            if (fa.t_dst != nullptr && ja < nj) { fa.t_dst[ia] = ta; fa.m_dst[ia] = ma; fa.v_dst[ia] = va; }
        }
        STG_STAMP(2, ta);
        if (ja < nj) {
            fa.keep[ca] = ta; fa.keep[fa.kstride + ca] = ma; fa.keep[2 * fa.kstride + ca] = va;
            lds0[da & 0x7fffu] = (da & PANEL_SCALED) ? ta * kTanhScale : ta; lds0[da >> 16] = ta;
        }
    }
    if (i > 0 && !pending) {
        const int s0 = CP::s0_of(i);
        const int npad = (((i + 7) & ~7) - i) * H;
        for (int e = tid; e < npad; e += NT) {
            const int k = i + e / H, j = e % H;
            lds0[PANEL_BASE + CP::oW0T + j * s0 + k] = 0.0f;
        }
    }
    { float d_ = lds0[4]; STG_STAMP(3, d_); }
    return gave_up ? 2 : 0;
}

// The same staging for groups of MORE THAN EIGHT blocks (n > 2048: up to sixteen copies, round 5) -- a second instantiation of the
// kernel (template flag WIDE), so that the code of the common case stays what round 4 measured (the two-pass loop in ONE function
// cost the Plaza clique 3.7 % and C3 2.4 %, scripts/ab.py, one box).
template <int K, int H>
__device__ __forceinline__ int stage_cond_panel_persist_wide(float* lds0, const float* theta_generic, PersistAdam& fa, const uint32_t* map_generic,
                                                        int i, int tid, int NT, int st_step, int st_stop, int n, int iter) {
    using CP = CondPanel<K, H>;
    using LY = Layout<K, H>;
    typedef const __attribute__((address_space(1))) float* gp;
    typedef const __attribute__((address_space(1))) uint32_t* gu;
    gp t_src = (gp)theta_generic;
    constexpr int PoP = CP::PoP;
    const int j0 = (i == 0) ? 0 : LY::off(i), nj = (i == 0) ? PoP : LY::block(i);
    gu map = (gu)map_generic + j0;
    const bool pending = fa.tagged != nullptr;
    int gave_up = 0;
    // (looked at BEFORE anything is waited for: nobody writes the copies of a finished clique)
    if (st_stop != 0 || st_step + iter >= fa.max_iters) return 1;    // block-uniform
    STG_T0();
    for (int base = 0; base < nj; base += 2 * NT) {
        const int ja = base + tid, jb = base + NT + tid;
        const int ca = (ja < nj ? ja : 0), cb = (jb < nj ? jb : 0);
        const int ia = j0 + ca, ib = j0 + cb;
        uint32_t da = map[ca], db = map[cb];
        float ta, tb, ma, va, mb, vb;
        float ga[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // (a pass that gives up leaves them unset)
        if (!pending) {                                        // launch-uniform per iteration: the chunk's first iteration
            ta = t_src[ia]; tb = t_src[ib];
            ma = fa.m_src[ia]; va = fa.v_src[ia];
            mb = fa.m_src[ib]; vb = fa.v_src[ib];
            asm volatile("" : "+v"(ta), "+v"(tb), "+v"(ma), "+v"(va), "+v"(mb), "+v"(vb), "+v"(da), "+v"(db));
        } else {
            ta = fa.keep[ca]; ma = fa.keep[fa.kstride + ca]; va = fa.keep[2 * fa.kstride + ca];
            tb = fa.keep[cb]; mb = fa.keep[fa.kstride + cb]; vb = fa.keep[2 * fa.kstride + cb];
            // the copies' (value, tag) pairs of this thread's two parameters: all sixteen loads in flight, again until every
            // tag is this iteration's (a copy whose block is still computing shows the tag of two iterations ago, or 0).
            // A group of more than eight blocks (n > 2048: up to sixteen copies, round 5) takes two such passes; the sums are
            // formed in nsf_adam_kernel's lane-partial order for that many copies -- p_c = g_c + g_{c+8}, then p_0 + ... + p_7 --
            // which for eight copies or fewer IS the copy order (one pass, the code of round 4).
            const int passes = fa.copies > 8 ? 2 : 1;          // block-uniform
            for (int pass = 0; pass < passes && !gave_up; ++pass) {
            const int c0 = 8 * pass;
            unsigned spins = 0;
            for (;;) {
                unsigned long long qa[8], qb[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    qa[c] = tagged_pair(fa.tagged, c0 + c, fa.copies, (unsigned)fa.cstride * 4u, 8u * (unsigned)ia);
                    qb[c] = tagged_pair(fa.tagged, c0 + c, fa.copies, (unsigned)fa.cstride * 4u, 8u * (unsigned)ib);
                }
                // (the iteration's bias corrections while the loads are under way: ~80 instructions that depend on nothing loaded)
                if (base == 0 && spins == 0u && pass == 0)
                    fa.kc = adam_coef(fa.lr, fa.beta1, fa.beta2, fa.eps, fa.log_b1, fa.log_b2, st_step + iter, n);
                if (spins == 0u && pass == 0) STG_STAMP(0, fa.kc.step_size);
                bool ok = true;
                float la[8], lb[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    ok = ok && (uint32_t)(qa[c] >> 32) == fa.tag && (uint32_t)(qb[c] >> 32) == fa.tag;
                    la[c] = __uint_as_float((uint32_t)qa[c]);
                    lb[c] = __uint_as_float((uint32_t)qb[c]);
                }
                if (ok) {
                    if (pass == 0) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) { ga[c] = la[c]; gb[c] = lb[c]; }
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            if (8 + c < fa.copies) { ga[c] += la[c]; gb[c] += lb[c]; }
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                ++spins;
#if defined(NSF_STAMPS)
                ++fa.looks;
#endif
                // every 64 looks: has a member of the group given up?  after 2^spin_log2 looks: give up (and say so)
                if ((spins & 63u) == 0u) {
                    const bool timeout = spins > (1u << fa.spin_log2);
                    if (timeout) __hip_atomic_fetch_or(fa.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (timeout || (__hip_atomic_load(fa.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u) != 0u) { gave_up = 1; break; }
                }
            }
            }
        }
        STG_STAMP(1, ga[0]);
        if (pending && !gave_up) {                             // (a thread that gave up has no gradient sum -- nothing is applied or recorded)
            FusedAdam sum_order;                               // (the summation order of nsf_adam_kernel / stage_cond_panel)
            sum_order.copies = fa.copies < 8 ? fa.copies : 8;  // (more than eight copies: the eight lane partials)
            adam_update(fa.kc, fused_sum_grads(sum_order, ga), ma, va, ta);
            adam_update(fa.kc, fused_sum_grads(sum_order, gb), mb, vb, tb);
            if (fa.t_dst != nullptr) {                         // the dim's first block records the new state (read by later KERNELS)
                if (ja < nj) { fa.t_dst[ia] = ta; fa.m_dst[ia] = ma; fa.v_dst[ia] = va; }
                if (jb < nj) { fa.t_dst[ib] = tb; fa.m_dst[ib] = mb; fa.v_dst[ib] = vb; }
            }
        }
        STG_STAMP(2, ta);
        if (ja < nj) {
            fa.keep[ca] = ta; fa.keep[fa.kstride + ca] = ma; fa.keep[2 * fa.kstride + ca] = va;
            lds0[da & 0x7fffu] = (da & PANEL_SCALED) ? ta * kTanhScale : ta; lds0[da >> 16] = ta;
        }
        if (jb < nj) {
            fa.keep[cb] = tb; fa.keep[fa.kstride + cb] = mb; fa.keep[2 * fa.kstride + cb] = vb;
            lds0[db & 0x7fffu] = (db & PANEL_SCALED) ? tb * kTanhScale : tb; lds0[db >> 16] = tb;
        }
    }
    if (i > 0 && !pending) {                                   // (the zero weights behind W0's rows stay: the panel is the block's for the whole chunk)
        const int s0 = CP::s0_of(i);
        const int npad = (((i + 7) & ~7) - i) * H;
        for (int e = tid; e < npad; e += NT) {
            const int k = i + e / H, j = e % H;
            lds0[PANEL_BASE + CP::oW0T + j * s0 + k] = 0.0f;
        }
    }
    { float d_ = lds0[4]; STG_STAMP(3, d_); }
    return gave_up ? 2 : 0;
}

// ---- the same exchange with the Adam update DIVIDED among the group's blocks (contended launches) --------------------------
// stage_cond_panel_persist lets every block of a (clique, dim) group derive the whole dim's update: right for a lone
// wave per SIMD (one memory round trip), wasteful when three waves share a SIMD's issue port -- on C3 the 8 blocks x 256
// threads of a group each load 16 tagged pairs and run Adam twice, 25 MB of L2 reads and ~10 % of the launch's VALU
// instructions per iteration, and the staging takes 7.5 k of a wave's 27.6 k cycles.  Here block b owns the slice
// [b S, (b + 1) S) of the dim's parameters (S = ceil(nj / members) <= 256: one thread per parameter, the first waves of the
// block): it alone loads that slice's copies, keeps its m, v, theta in LDS, applies Adam (the same function on the same sums:
// the same bits), records the slice in the clique's buffers and PUBLISHES the new theta as (value, tag) pairs in a
// per-clique exchange buffer; then every thread of every block fetches the two parameters it stages into the panel from
// that buffer, again until the tags are this iteration's.  Two dependent round trips instead of one, an eighth of the
// loads and of the arithmetic.  The exchange buffer needs no second copy: a block can publish theta of iteration k + 1
// only after it has seen everybody's gradient copy of iteration k, which a block writes after it has read theta of iteration k.
template <int K, int H>
__device__ __forceinline__ int stage_cond_panel_persist_split(float* lds0, const float* theta_generic, PersistAdam& fa, const uint32_t* map_generic,
                                                              int i, int tid, int NT, int st_step, int st_stop, int n, int iter, int bx,
                                                              __attribute__((address_space(1))) float* xch) {
    using CP = CondPanel<K, H>;
    using LY = Layout<K, H>;
    typedef const __attribute__((address_space(1))) float* gp;
    typedef const __attribute__((address_space(1))) uint32_t* gu;
    typedef const __attribute__((address_space(1))) unsigned long long* gq;
    gp t_src = (gp)theta_generic;
    constexpr int PoP = CP::PoP;
    const int j0 = (i == 0) ? 0 : LY::off(i), nj = (i == 0) ? PoP : LY::block(i);
    gu map = (gu)map_generic + j0;
    const bool pending = fa.tagged != nullptr;
    if (st_stop != 0 || st_step + iter >= fa.max_iters) return 1;    // block-uniform
    const int S = (nj + fa.copies - 1) / fa.copies;                  // slice of this block: parameters [bx S, min((bx + 1) S, nj))
    const uint32_t tag2 = fa.tag + 1u;                               // the tag of this iteration's theta (= state->step + it + 1)
    int gave_up = 0;
    auto look_again = [&](unsigned& spins) -> bool {                 // -> true: give up
        __builtin_amdgcn_s_sleep(1);
        ++spins;
#if defined(NSF_STAMPS)
        ++fa.looks;
#endif
        if ((spins & 63u) == 0u) {
            const bool timeout = spins > (1u << fa.spin_log2);
            if (timeout) __hip_atomic_fetch_or(fa.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (timeout || (__hip_atomic_load(fa.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u) != 0u) return true;
        }
        return false;
    };
    // one thread per parameter of the slice (S <= 110 for a group of eight blocks: the block's first waves; a clique of few
    // particles has few blocks and long slices: the threads go round)
    for (int q = tid; q < S && bx * S + q < nj; q += NT) {
        const int ia = j0 + bx * S + q;
        if (!pending) {
            // the chunk's first iteration: nothing to apply; the owners pick up their slice's state from the clique's arrays
            fa.keep[q] = t_src[ia];
            fa.keep[fa.kstride + q] = fa.m_src[ia];
            fa.keep[2 * fa.kstride + q] = fa.v_src[ia];
            continue;
        }
        float ta = fa.keep[q], ma = fa.keep[fa.kstride + q], va = fa.keep[2 * fa.kstride + q];
        float ga[8];
        unsigned spins = 0;
        for (;;) {
            unsigned long long qa[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                qa[c] = __hip_atomic_load((const unsigned long long*)(gq)(fa.tagged + (size_t)(c < fa.copies ? c : 0) * fa.cstride + 2 * (size_t)ia),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (spins == 0u && q == tid) fa.kc = adam_coef(fa.lr, fa.beta1, fa.beta2, fa.eps, fa.log_b1, fa.log_b2, st_step + iter, n);
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                ok = ok && (uint32_t)(qa[c] >> 32) == fa.tag;
                ga[c] = __uint_as_float((uint32_t)qa[c]);
            }
            if (ok) break;
            if (gave_up || look_again(spins)) { gave_up = 1; break; }
        }
        FusedAdam sum_order;
        sum_order.copies = fa.copies;
        adam_update(fa.kc, fused_sum_grads(sum_order, ga), ma, va, ta);
        fa.keep[q] = ta; fa.keep[fa.kstride + q] = ma; fa.keep[2 * fa.kstride + q] = va;
        fa.t_dst[ia] = ta; fa.m_dst[ia] = ma; fa.v_dst[ia] = va;      // (every block records ITS slice: read by later kernels)
        const unsigned long long pub = ((unsigned long long)tag2 << 32) | (unsigned long long)__float_as_uint(ta);
        __hip_atomic_store((unsigned long long*)(xch + 2 * (size_t)ia), pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int base = 0; base < nj; base += 2 * NT) {
        const int ja = base + tid, jb = base + NT + tid;
        const int ca = (ja < nj ? ja : 0), cb = (jb < nj ? jb : 0);
        const int ia = j0 + ca, ib = j0 + cb;
        uint32_t da = map[ca], db = map[cb];
        float ta, tb;
        if (!pending) {
            ta = t_src[ia]; tb = t_src[ib];
            asm volatile("" : "+v"(ta), "+v"(tb), "+v"(da), "+v"(db));
        } else {
            unsigned spins = 0;
            for (;;) {
                const unsigned long long qa = __hip_atomic_load((const unsigned long long*)(gq)(xch + 2 * (size_t)ia), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long qb = __hip_atomic_load((const unsigned long long*)(gq)(xch + 2 * (size_t)ib), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ta = __uint_as_float((uint32_t)qa); tb = __uint_as_float((uint32_t)qb);
                if ((uint32_t)(qa >> 32) == tag2 && (uint32_t)(qb >> 32) == tag2) break;
                if (gave_up || look_again(spins)) { gave_up = 1; break; }
            }
        }
        if (ja < nj) { lds0[da & 0x7fffu] = (da & PANEL_SCALED) ? ta * kTanhScale : ta; lds0[da >> 16] = ta; }
        if (jb < nj) { lds0[db & 0x7fffu] = (db & PANEL_SCALED) ? tb * kTanhScale : tb; lds0[db >> 16] = tb; }
    }
    if (i > 0 && !pending) {
        const int s0 = CP::s0_of(i);
        const int npad = (((i + 7) & ~7) - i) * H;
        for (int e = tid; e < npad; e += NT) {
            const int k = i + e / H, j = e % H;
            lds0[PANEL_BASE + CP::oW0T + j * s0 + k] = 0.0f;
        }
    }
    return gave_up ? 2 : 0;
}

template <int K, int H>
__device__ __forceinline__ int stage_cond_panel_persist_split_wide(float* lds0, const float* theta_generic, PersistAdam& fa, const uint32_t* map_generic,
                                                              int i, int tid, int NT, int st_step, int st_stop, int n, int iter, int bx,
                                                              __attribute__((address_space(1))) float* xch) {
    using CP = CondPanel<K, H>;
    using LY = Layout<K, H>;
    typedef const __attribute__((address_space(1))) float* gp;
    typedef const __attribute__((address_space(1))) uint32_t* gu;
    typedef const __attribute__((address_space(1))) unsigned long long* gq;
    gp t_src = (gp)theta_generic;
    constexpr int PoP = CP::PoP;
    const int j0 = (i == 0) ? 0 : LY::off(i), nj = (i == 0) ? PoP : LY::block(i);
    gu map = (gu)map_generic + j0;
    const bool pending = fa.tagged != nullptr;
    if (st_stop != 0 || st_step + iter >= fa.max_iters) return 1;    // block-uniform
    const int S = (nj + fa.copies - 1) / fa.copies;                  // slice of this block: parameters [bx S, min((bx + 1) S, nj))
    const uint32_t tag2 = fa.tag + 1u;                               // the tag of this iteration's theta (= state->step + it + 1)
    int gave_up = 0;
    auto look_again = [&](unsigned& spins) -> bool {                 // -> true: give up
        __builtin_amdgcn_s_sleep(1);
        ++spins;
#if defined(NSF_STAMPS)
        ++fa.looks;
#endif
        if ((spins & 63u) == 0u) {
            const bool timeout = spins > (1u << fa.spin_log2);
            if (timeout) __hip_atomic_fetch_or(fa.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (timeout || (__hip_atomic_load(fa.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u) != 0u) return true;
        }
        return false;
    };
    // one thread per parameter of the slice (S <= 110 for a group of eight blocks: the block's first waves; a clique of few
    // particles has few blocks and long slices: the threads go round)
    for (int q = tid; q < S && bx * S + q < nj; q += NT) {
        const int ia = j0 + bx * S + q;
        if (!pending) {
            // the chunk's first iteration: nothing to apply; the owners pick up their slice's state from the clique's arrays
            fa.keep[q] = t_src[ia];
            fa.keep[fa.kstride + q] = fa.m_src[ia];
            fa.keep[2 * fa.kstride + q] = fa.v_src[ia];
            continue;
        }
        float ta = fa.keep[q], ma = fa.keep[fa.kstride + q], va = fa.keep[2 * fa.kstride + q];
        float ga[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int passes = fa.copies > 8 ? 2 : 1;                    // (more than eight copies: two passes, lane-partial order -- see stage_cond_panel_persist)
        for (int pass = 0; pass < passes && !gave_up; ++pass) {
        const int c0 = 8 * pass;
        unsigned spins = 0;
        for (;;) {
            unsigned long long qa[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                qa[c] = __hip_atomic_load((const unsigned long long*)(gq)(fa.tagged + (size_t)(c0 + c < fa.copies ? c0 + c : 0) * fa.cstride + 2 * (size_t)ia),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (spins == 0u && q == tid && pass == 0) fa.kc = adam_coef(fa.lr, fa.beta1, fa.beta2, fa.eps, fa.log_b1, fa.log_b2, st_step + iter, n);
            bool ok = true;
            float la[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                ok = ok && (uint32_t)(qa[c] >> 32) == fa.tag;
                la[c] = __uint_as_float((uint32_t)qa[c]);
            }
            if (ok) {
                if (pass == 0) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) ga[c] = la[c];
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (8 + c < fa.copies) ga[c] += la[c];
                }
                break;
            }
            if (gave_up || look_again(spins)) { gave_up = 1; break; }
        }
        }
        if (gave_up) break;                                          // (no gradient sum -- nothing is applied, recorded or published)
        FusedAdam sum_order;
        sum_order.copies = fa.copies < 8 ? fa.copies : 8;
        adam_update(fa.kc, fused_sum_grads(sum_order, ga), ma, va, ta);
        fa.keep[q] = ta; fa.keep[fa.kstride + q] = ma; fa.keep[2 * fa.kstride + q] = va;
        fa.t_dst[ia] = ta; fa.m_dst[ia] = ma; fa.v_dst[ia] = va;      // (every block records ITS slice: read by later kernels)
        const unsigned long long pub = ((unsigned long long)tag2 << 32) | (unsigned long long)__float_as_uint(ta);
        __hip_atomic_store((unsigned long long*)(xch + 2 * (size_t)ia), pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int base = 0; base < nj; base += 2 * NT) {
        const int ja = base + tid, jb = base + NT + tid;
        const int ca = (ja < nj ? ja : 0), cb = (jb < nj ? jb : 0);
        const int ia = j0 + ca, ib = j0 + cb;
        uint32_t da = map[ca], db = map[cb];
        float ta, tb;
        if (!pending) {
            ta = t_src[ia]; tb = t_src[ib];
            asm volatile("" : "+v"(ta), "+v"(tb), "+v"(da), "+v"(db));
        } else {
            unsigned spins = 0;
            for (;;) {
                const unsigned long long qa = __hip_atomic_load((const unsigned long long*)(gq)(xch + 2 * (size_t)ia), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long qb = __hip_atomic_load((const unsigned long long*)(gq)(xch + 2 * (size_t)ib), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ta = __uint_as_float((uint32_t)qa); tb = __uint_as_float((uint32_t)qb);
                if ((uint32_t)(qa >> 32) == tag2 && (uint32_t)(qb >> 32) == tag2) break;
                if (gave_up || look_again(spins)) { gave_up = 1; break; }
            }
        }
        if (ja < nj) { lds0[da & 0x7fffu] = (da & PANEL_SCALED) ? ta * kTanhScale : ta; lds0[da >> 16] = ta; }
        if (jb < nj) { lds0[db & 0x7fffu] = (db & PANEL_SCALED) ? tb * kTanhScale : tb; lds0[db >> 16] = tb; }
    }
    if (i > 0 && !pending) {
        const int s0 = CP::s0_of(i);
        const int npad = (((i + 7) & ~7) - i) * H;
        for (int e = tid; e < npad; e += NT) {
            const int k = i + e / H, j = e % H;
            lds0[PANEL_BASE + CP::oW0T + j * s0 + k] = 0.0f;
        }
    }
    return gave_up ? 2 : 0;
}

// acc[g * SPLIT + (q % SPLIT)][u] += sum over the quads q < NK/4 of  A[4g + u][4q + v] * b[4q + v]
// for the NG output groups g; `rows` = the panel matrix at this lane's row (base + (lane & 3) * stride).
// Software pipeline: the reads of slab s+1 (SG groups x one quad, 16 bytes each) are issued before the MFMAs of slab s,
// and inside a slab the instructions of the SG (x SPLIT) independent chains alternate -- left to itself the scheduler
// minimises registers instead: read, wait, four dependent MFMAs with s_nop between them, next read.
template <int NG, int NK, int SG, int SPLIT, class USED>
__device__ __forceinline__ void mfma_rows(const float* rows, int stride, cm_f32x4 (&acc)[NG * SPLIT], const float (&b)[NK]) {
    static_assert(NK % 4 == 0 && NG % SG == 0, "whole quads, whole slabs");
    constexpr int NQ = NK / 4, NGS = NG / SG, NSLAB = NQ * NGS;
    cm_f32x4 buf[2][SG];
#pragma unroll
    for (int gg = 0; gg < SG; ++gg) buf[0][gg] = *(const cm_f32x4*)(rows + 4 * gg * stride);
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
        const int q = s / NGS, gs = s % NGS;
        if (s + 1 < NSLAB) {
            const int qn = (s + 1) / NGS, gn = (s + 1) % NGS;
#pragma unroll
            for (int gg = 0; gg < SG; ++gg)
                buf[(s + 1) & 1][gg] = *(const cm_f32x4*)(rows + 4 * (gn * SG + gg) * stride + 4 * qn);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int gg = 0; gg < SG; ++gg)
                if (USED::at(4 * q + u)) {
                    cm_f32x4& c = acc[(gs * SG + gg) * SPLIT + (q % SPLIT)];
                    c = mfma1(buf[s & 1][gg][u], b[4 * q + u], c);
                }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// theta and the hidden activations of dim i for the lane's particle; xt = the tile [rows][xs], rows 0..i loaded, the
// lane's particle in column `col`; s0 = row stride of W0T (CondPanel::s0_of(i), or the pair kernel's clique-wide one).
// `pan` may differ between lanes as long as the four lanes of an MFMA block agree (nsf_train3_kernel: two dims per wave;
// `i` is then the larger of the two, the shorter dim's extra weights are zero).
template <int K, int H>
__device__ __forceinline__ void cond_forward_mfma(const float* pan, int i, int s0, const float* xt, int xs, int lane, int col,
                                                  float (&h1)[H], float (&h2)[H], float (&th)[Layout<K, H>::PoP]) {
    using CP = CondPanel<K, H>;
    constexpr int ST = CP::ST, GH = CP::GH, G2 = CP::G2, SP = (H >= 8) ? NSF_COND_SPLIT : 1;
    const int r = lane & 3;
    const cm_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {   // layer 0: the only contraction whose length depends on the dim: eight inputs per round (zero weights past i,
        // the clamped row keeps the activation finite), four chains (group x quad parity)
        cm_f32x4 a1[GH][SP];
#pragma unroll
        for (int g = 0; g < GH; ++g) {
            a1[g][0] = *(const cm_f32x4*)(pan + CP::ob0 + 4 * g);
            if (SP == 2) a1[g][SP - 1] = zero;
        }
        const float* w0 = pan + CP::oW0T + r * s0;
        for (int k0 = 0; k0 < i; k0 += 8) {
            float xk[8];
            cm_f32x4 a4[GH][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + u; xk[u] = xt[(k < i ? k : i) * xs + col]; }
#pragma unroll
            for (int g = 0; g < GH; ++g)
#pragma unroll
                for (int q = 0; q < 2; ++q) a4[g][q] = *(const cm_f32x4*)(w0 + 4 * g * s0 + k0 + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SP == 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int g = 0; g < GH; ++g)
#pragma unroll
                        for (int q = 0; q < 2; ++q) a1[g][q] = mfma1(a4[g][q][u], xk[4 * q + u], a1[g][q]);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int g = 0; g < GH; ++g) a1[g][0] = mfma1(a4[g][q][u], xk[4 * q + u], a1[g][0]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) h1[4 * g + u] = ftanh_scaled(SP == 2 ? a1[g][0][u] + a1[g][SP - 1][u] : a1[g][0][u]);
    }
    {
        cm_f32x4 a2[GH * SP];
#pragma unroll
        for (int g = 0; g < GH; ++g) {
            a2[g * SP] = *(const cm_f32x4*)(pan + CP::ob1 + 4 * g);
            if (SP == 2) a2[g * SP + 1] = zero;
        }
        mfma_rows<GH, H, GH, SP, EveryCol>(pan + CP::oW1T + r * ST, ST, a2, h1);
#pragma unroll
        for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) h2[4 * g + u] = ftanh_scaled(SP == 2 ? a2[g * SP][u] + a2[g * SP + 1][u] : a2[g][u]);
    }
    {
        cm_f32x4 t[G2];
#pragma unroll
        for (int g = 0; g < G2; ++g) t[g] = *(const cm_f32x4*)(pan + CP::ob2 + 4 * g);
        mfma_rows<G2, H, (G2 % 4 == 0 ? 4 : 2), 1, EveryCol>(pan + CP::oW2T + r * ST, ST, t, h2);
#pragma unroll
        for (int g = 0; g < G2; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) th[4 * g + u] = t[g][u];
    }
}

// ga2 = dL/da2, ga1 = dL/da1 of the lane's particle from gth = dL/dtheta (the transposed products of the same rows);
// padding columns of the theta layout carry no gradient and are skipped
template <int K, int H>
__device__ __forceinline__ void cond_backward_mfma(const float* pan, int lane, const float (&gth)[Layout<K, H>::PoP],
                                                   const float (&h1)[H], const float (&h2)[H],
                                                   float (&ga2)[H], float (&ga1)[H]) {
    using CP = CondPanel<K, H>;
    constexpr int PoP = CP::PoP, GH = CP::GH, SP = (H >= 8) ? NSF_COND_SPLIT : 1;
    const int r = lane & 3;
    const cm_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {
        cm_f32x4 s[GH * SP];
#pragma unroll
        for (int g = 0; g < GH * SP; ++g) s[g] = zero;
        mfma_rows<GH, PoP, GH, SP, ThetaColUsed<K, H>>(pan + CP::oW2N + r * CP::NS2, CP::NS2, s, gth);
#pragma unroll
        for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float h = h2[4 * g + u];
                ga2[4 * g + u] = (SP == 2 ? s[SP * g][u] + s[SP * g + SP - 1][u] : s[g][u]) * (1.0f - h * h);
            }
    }
    {
        cm_f32x4 t[GH * SP];
#pragma unroll
        for (int g = 0; g < GH * SP; ++g) t[g] = zero;
        mfma_rows<GH, H, GH, SP, EveryCol>(pan + CP::oW1N + r * CP::ST, CP::ST, t, ga2);
#pragma unroll
        for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float h = h1[4 * g + u];
                ga1[4 * g + u] = (SP == 2 ? t[g * SP][u] + t[g * SP + 1][u] : t[g][u]) * (1.0f - h * h);
            }
    }
}

// =====================================================================================================================
// Two dims per wave (nsf_train3_kernel: multi-layer flows and VJP launches in the latency regime).
//
// Lanes 0-31 of a wave run dim 2j, lanes 32-63 dim 2j+1 of the same 32 particles; the 4x4x1 MFMA blocks are four lanes
// wide, so every block sits inside one dim and the only thing that differs between the halves is the panel a lane reads.
// Panels are wave-private ((layer, dim) of the wave's own dims, all layers resident) with ONE row stride of W0T for the
// whole clique (s0 = s0_of(D - 1): the layer-0 loop runs to the longer dim of the pair, the shorter one multiplies zeros)
// and one more matrix for dL/dx of the layer input:
//   W0N [4*ceil((D-1)/4)][ST]   W0N[k][j] = W0[k][j], zero rows for k >= i        backward, layer 0
// Dim 0 has no conditioner (its PoP parameters ARE theta): its panel is all zeros with b2 = the parameters, so
// theta = b2 + 0 * h2 exactly and db2 of the gradient GEMM is its gradient -- no second code path in the wave.
template <int K, int H>
struct PairPanel {
    using CP = CondPanel<K, H>;
    __host__ __device__ static constexpr int s0(int D) { return CP::s0_of(D > 1 ? D - 1 : 1); }
    __host__ __device__ static constexpr int oW0N(int D) { return CP::oW0T + H * s0(D); }
    __host__ __device__ static constexpr int rowsN(int D) { return (D - 1 + 3) & ~3; }
    __host__ __device__ static constexpr int dump(int D) { return oW0N(D) + rowsN(D) * CP::ST; }   // second store of one-destination parameters
    __host__ __device__ static constexpr int floats(int D) { return dump(D) + 4; }
};
constexpr int PAIR_MAX_D = 16;              // one 16-column operand tile for [x | 1] in the dW0 GEMM
constexpr int PAIR_MAP_HEAD = 32;           // the table starts with the word offsets of the maps of D = 0 .. 31
// PairPanel<K, H>::floats(D) for run-time K, H (the Adam kernel of the common unit addresses the panel image with it)
__host__ __device__ constexpr int pair_panel_floats(int K, int H, int D) {
    const int PoP = pop_of(K), ST = H | 4;
    const int oW0T = PoP * ST + PoP + H * ST + H + H * (PoP + 4) + H * ST + H;
    const int s0 = (((D > 1 ? D - 1 : 1) + 7) & ~7) + 4;
    return oW0T + H * s0 + ((D - 1 + 3) & ~3) * ST + 4;
}
static_assert(pair_panel_floats(9, 8, 6) == PairPanel<9, 8>::floats(6) && pair_panel_floats(5, 4, 11) == PairPanel<5, 4>::floats(11) &&
                  pair_panel_floats(16, 8, 16) == PairPanel<16, 8>::floats(16),
              "run-time restatement of PairPanel::floats");

// table[D] = offset of clique width D's map; map[jj] for the kernel-layout index jj of ONE layer (same packing as
// build_panel_map; offsets counted from the first word of the LAYER's panels: dim i's panel starts i * floats(D) in).
// The same layout exists in device memory ("panel image", behind the loss ring of the clique's workspace): the Adam
// kernel writes every updated parameter to its one or two places there, and the training kernel's prologue becomes a
// straight 16-byte copy instead of ~15 instructions per parameter in every one of the clique's blocks.
template <int K, int H>
static inline size_t pair_map_words() {
    size_t n = PAIR_MAP_HEAD;
    for (int D = 1; D <= PAIR_MAX_D; ++D) n += Layout<K, H>::count(D);
    return n;
}
template <int K, int H>
static inline void build_pair_map(uint32_t* table) {
    using CP = CondPanel<K, H>;
    using PP = PairPanel<K, H>;
    using LY = Layout<K, H>;
    constexpr int PoP = CP::PoP, ST = CP::ST;
    size_t at = PAIR_MAP_HEAD;
    for (int D = 0; D < PAIR_MAP_HEAD; ++D) table[D] = 0;
    for (int D = 1; D <= PAIR_MAX_D; ++D) {
        table[D] = (uint32_t)at;
        uint32_t* map = table + at;
        at += LY::count(D);
        const int s0 = PP::s0(D), dump0 = PP::dump(D), PS = PP::floats(D);
        for (int j = 0; j < PoP; ++j) map[j] = (uint32_t)(CP::ob2 + j) | ((uint32_t)dump0 << 16);
        for (int i = 1; i < D; ++i) {
            uint32_t* m = map + LY::off(i);
            const int dump = i * PS + dump0;
            for (int jj = 0; jj < LY::block(i); ++jj) {
                int d0 = 0, d1 = dump0;
                if (jj < LY::ob0(i)) {
                    const int k = jj / H, j = jj - k * H;
                    d0 = CP::oW0T + j * s0 + k;
                    d1 = PP::oW0N(D) + k * ST + j;
                } else if (jj < LY::oW1(i)) {
                    d0 = CP::ob0 + (jj - LY::ob0(i));
                } else if (jj < LY::ob1(i)) {
                    const int e = jj - LY::oW1(i), k = e / H, j = e - k * H;
                    d0 = CP::oW1T + j * ST + k;
                    d1 = CP::oW1N + k * ST + j;
                } else if (jj < LY::oW2(i)) {
                    d0 = CP::ob1 + (jj - LY::ob1(i));
                } else if (jj < LY::ob2(i)) {
                    const int e = jj - LY::oW2(i), k = e / PoP, o = e - k * PoP;
                    d0 = CP::oW2T + o * ST + k;
                    d1 = CP::oW2N + k * CP::NS2 + o;
                } else {
                    d0 = CP::ob2 + (jj - LY::ob2(i));
                }
                const bool scaled = jj < LY::oW2(i);
                (void)dump;
                m[jj] = (uint32_t)(i * PS + d0) | (scaled ? PANEL_SCALED : 0u) | ((uint32_t)(i * PS + d1) << 16);
            }
        }
    }
}

// One wave brings the panels of the dims iA = 2j and (if nd == 2) iA + 1, all L layers, into LDS: zero fill, then the
// parameters through the map.  The map words of a chunk are loaded once and serve every layer; the loads of both dims and
// up to four layers are in flight together (one memory round trip for a pair of D <= 8).
// lay0 = the panels of layer 0 (dim 0's first); PS = floats per panel; pstride = floats between consecutive layers.
template <int K, int H>
__device__ __forceinline__ void stage_pair_panels(float* lay0, int PS, size_t pstride, const float* theta_generic, size_t layer_stride,
                                                  const uint32_t* map_generic, int iA, int nd, int L, int lane) {
    using LY = Layout<K, H>;
    typedef const __attribute__((address_space(1))) float* gp;
    typedef const __attribute__((address_space(1))) uint32_t* gu;
    int j0[2], nj[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int i = iA + (hb < nd ? hb : 0);
        j0[hb] = (i == 0) ? 0 : LY::off(i);
        nj[hb] = (i == 0) ? LY::PoP : LY::block(i);
    }
    const int njmax = nj[nd - 1] > nj[0] ? nj[nd - 1] : nj[0];
    gp t0 = (gp)theta_generic;
    gu map = (gu)map_generic;
    constexpr int U = 8, LB = 4;
    for (int lb = 0; lb < L; lb += LB) {
        for (int base = 0; base < njmax; base += 64 * U) {
            uint32_t d[2][U];
            float t[2][LB][U];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int jj = base + 64 * u + lane;
                    d[hb][u] = map[j0[hb] + (jj < nj[hb] ? jj : 0)];
                }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    const int l = (lb + q < L) ? lb + q : L - 1;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int jj = base + 64 * u + lane;
                        t[hb][q][u] = t0[(size_t)l * layer_stride + j0[hb] + (jj < nj[hb] ? jj : 0)];
                    }
                }
            if (base == 0) {
                // zero fill while the loads are in flight (the wave's LDS operations execute in order: the parameters land
                // on top); the two dims' panels of a layer are adjacent
                const cm_f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                for (int q = 0; q < LB && lb + q < L; ++q) {
                    float* pq = lay0 + (size_t)(lb + q) * pstride + iA * PS;
                    for (int e = 4 * lane; e < nd * PS; e += 256) *(cm_f32x4*)(pq + e) = z4;
                }
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    if (lb + q < L && hb < nd) {
                        float* pq = lay0 + (size_t)(lb + q) * pstride;
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int jj = base + 64 * u + lane;
                            if (jj < nj[hb]) {
                                const float v = t[hb][q][u];
                                pq[d[hb][u] >> 16] = v;
                                pq[d[hb][u] & 0x7fffu] = (d[hb][u] & PANEL_SCALED) ? v * kTanhScale : v;
                            }
                        }
                    }
                }
        }
    }
}

// The same panels out of the clique's panel image (device memory, maintained by the Adam kernel): 16-byte copies, ONE layer
// at a time -- the prologue brings layer 0, every forward stage requests the next layer's panels before its arithmetic and
// drops them into LDS behind it (the wave's panels are its own: no barrier), so only the first layer's round trip is exposed.
constexpr int PAIR_PANEL_WORDS = 10;                          // D <= 16: 2 x 1268 floats = 634 16-byte words = 10 per lane
__device__ __forceinline__ void load_pair_panels(const float* image_generic, size_t pstride, int PS, int iA, int nd, int l, int lane,
                                                 cm_f32x4 (&v)[PAIR_PANEL_WORDS]) {
    typedef const __attribute__((address_space(1))) cm_f32x4* gv4;
    const int n4 = (nd * PS) >> 2;                            // 16-byte words (PS is a multiple of 4)
    gv4 src = (gv4)(image_generic + (size_t)l * pstride + (size_t)iA * PS);
#pragma unroll
    for (int u = 0; u < PAIR_PANEL_WORDS; ++u) {
        const int e = lane + 64 * u;
        v[u] = src[e < n4 ? e : 0];
    }
}
__device__ __forceinline__ void store_pair_panels(float* lay0, size_t pstride, int PS, int iA, int nd, int l, int lane,
                                                  const cm_f32x4 (&v)[PAIR_PANEL_WORDS]) {
    const int n4 = (nd * PS) >> 2;
    cm_f32x4* dst = (cm_f32x4*)(lay0 + (size_t)l * pstride + (size_t)iA * PS);
#pragma unroll
    for (int u = 0; u < PAIR_PANEL_WORDS; ++u) {
        const int e = lane + 64 * u;
        if (e < n4) dst[e] = v[u];
    }
}

// Forward state of a (layer, dim, particle) that the backward pass needs (hidden activations + the spline's selected bin),
// as NF floats: the multi-layer kernel parks it in device memory during the forward pass (one coalesced row per field)
// instead of recomputing conditioner and spline in the backward pass.
__host__ __device__ constexpr int pair_stash_fields(int K, int H) { return (2 * H + 2 * K + 12 + 3) & ~3; }   // whole 16-byte words
template <int K, int H>
__device__ __forceinline__ void stash_pack(const float (&h1)[H], const float (&h2)[H], const nsf::SplineT<K>& S,
                                           float (&sv)[pair_stash_fields(K, H)]) {
#pragma unroll
    for (int q = 0; q < H; ++q) { sv[q] = h1[q]; sv[H + q] = h2[q]; }
#pragma unroll
    for (int q = 0; q < K; ++q) { sv[2 * H + q] = S.ew[q]; sv[2 * H + K + q] = S.eh[q]; }
    float* t = &sv[2 * H + 2 * K];
    t[0] = S.iw; t[1] = S.ih; t[2] = S.Xk; t[3] = S.dx; t[4] = S.Yk; t[5] = S.dy; t[6] = S.d0; t[7] = S.d1;
    t[8] = S.ud0; t[9] = S.ud1; t[10] = S.t;
    int kc = 0;
#pragma unroll
    for (int j = 1; j < K; ++j) kc += S.sel[j] ? 1 : 0;
    t[11] = __int_as_float(kc | (S.inside ? 256 : 0));
#pragma unroll
    for (int q = 2 * H + 2 * K + 12; q < pair_stash_fields(K, H); ++q) sv[q] = 0.0f;
}
template <int K, int H>
__device__ __forceinline__ void stash_unpack(const float (&sv)[pair_stash_fields(K, H)], float (&h1)[H], float (&h2)[H],
                                             nsf::SplineT<K>& S) {
#pragma unroll
    for (int q = 0; q < H; ++q) { h1[q] = sv[q]; h2[q] = sv[H + q]; }
#pragma unroll
    for (int q = 0; q < K; ++q) { S.ew[q] = sv[2 * H + q]; S.eh[q] = sv[2 * H + K + q]; }
    const float* t = &sv[2 * H + 2 * K];
    S.iw = t[0]; S.ih = t[1]; S.Xk = t[2]; S.dx = t[3]; S.Yk = t[4]; S.dy = t[5]; S.d0 = t[6]; S.d1 = t[7];
    S.ud0 = t[8]; S.ud1 = t[9]; S.t = t[10];
    const int f = __float_as_int(t[11]), kc = f & 255;
    S.sel[0] = true;
#pragma unroll
    for (int j = 1; j < K; ++j) S.sel[j] = kc >= j;
    S.inside = (f & 256) != 0;
}

// dL/dx_k (k < 4 * NG4) of the lane's particle through layer 0 of the conditioner: gx[k] = sum_j W0[k][j] ga1[j]
template <int K, int H, int NG4>
__device__ __forceinline__ void cond_input_grad(const float* pan, int oW0N, int lane, const float (&ga1)[H], cm_f32x4 (&gx)[NG4], int ngroups) {
    using CP = CondPanel<K, H>;
    const float* rows = pan + oW0N + (lane & 3) * CP::ST;
#pragma unroll
    for (int g = 0; g < NG4; ++g) {
        cm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (g < ngroups) {                                   // wave-uniform
            cm_f32x4 a4[H / 4];
#pragma unroll
            for (int q = 0; q < H / 4; ++q) a4[q] = *(const cm_f32x4*)(rows + 4 * g * CP::ST + 4 * q);
#pragma unroll
            for (int q = 0; q < H / 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma1(a4[q][u], ga1[4 * q + u], acc);
        }
        gx[g] = acc;
    }
}
