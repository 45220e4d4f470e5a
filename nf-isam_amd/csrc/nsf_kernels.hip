// nsf_kernels.hip -- the COMMON unit of libnfisam_hip.so: the C ABI (include/nfisam_hip.h) of the NF-iSAM per-clique
// normalizing-flow hot path, training plans (hipGraph of one chunk of iterations), launch-shape decisions, and the
// kernels that do not depend on (K, H) at compile time: Adam, per-chunk bookkeeping (loss record + window early stop),
// training-batch normalisation, the elementwise spline of `flows.utils`.  The (K, H)-templated kernels live in the
// kernel units (nsf_unit.hip, nsf_units.h) and are reached through `NsfUnitOps`.  Written for CDNA4 only.
#include <atomic>
#include <mutex>
#include <thread>
#include <sched.h>
#include <time.h>
#include <vector>

#include "nsf_host.h"
#include "nsf_cond_mfma.h"
#include "nsf_bookkeep.h"

thread_local int nfisam_g_last_hip_error = 0;

typedef __attribute__((ext_vector_type(4))) float f32x4;

// =============================================================================================
// Adam (torch.optim.Adam defaults, src/slam/NFiSAM.py:425,475) + loss record + window early stop
// (NFiSAM.py:473,481-491).  One block per clique.
// =============================================================================================
struct AdamArgs {
    const nfisam_clique* cliques;
    nfisam_clique single;
    nfisam_adam_cfg cfg;
    int slab;               // 0: one gradient buffer; else particles per tile: per-tile slabs, summed here in tile order
    float log_b1, log_b2;   // ln(beta), computed on the host in double
    int L, K, H;
    int iter_idx;           // iteration index inside the chunk
    int chunk;              // bookkeeping kernel: iterations to close
    int max_n;              // largest n of the batch (number of slabs in every workspace)
    int few_copies;         // every clique has <= 8 gradient copies: one thread per parameter
    nfisam_train_state* mirror;   // bookkeeping kernel: host-pinned copy of the states (training plans) or nullptr
    const uint32_t* pair_map;     // != nullptr: multi-layer launches of nsf_train3_kernel -- every updated parameter is also written to
    uint32_t pair_off[PAIR_MAP_OFFSETS];   // the clique's PANEL IMAGE behind its loss ring (nsf_cond_mfma.h: build_pair_map)
    int close_chunk;        // > 0: fused-Adam launches (nsf_cond_mfma.h): apply the LAST iteration's pending update of a chunk of
                            // this many iterations; its gradient copies / source state sit in the buffers of that iteration's parity
    int span;               // != 0 (with close_chunk > 0): behind a WINDOW-SPANNING launch (nsf_unit.hip): which update is pending and the
                            // parity of its buffers come from the workspace's control words (SPAN_WORD_LAST_T / _PARITY), 0 = none
    int fused_close;        // != 0 (nsf_adam_kernel with `fused_close`): step / stop come from the words the chunk-persistent launch left (CLOSE_WORD_*): the
                            // bookkeeping block of the same kernel is advancing state->step / stop meanwhile
};

// One block of the bookkeeping (nsf_bookkeep.h) for clique `clique`, from the fields of the KERNEL'S OWN by-value argument `a`: no
// reference to `a` is formed -- a reference to a by-value kernel argument makes the compiler copy the argument block to scratch (248 bytes
// per thread and +5 us per launch of the multi-layer path's Adam kernel, measured when this code briefly lived in device functions)
#define NSF_BOOKKEEP_BLOCK(a, clique, zc)                                                                            \
    do {                                                                                                             \
        const bool batched_ = (a).cliques != nullptr;                                                                \
        const nfisam_clique* cp_ = batched_ ? ((a).cliques + (clique)) : nullptr;                                    \
        float* G_ = batched_ ? cp_->kgrad : (a).single.kgrad;                                                        \
        const int D_ = batched_ ? cp_->D : (a).single.D;                                                             \
        const int PoP_ = pop_of((a).K);                                                                              \
        const int kfixed_ = (a).H + (a).H * (a).H + (a).H + (a).H * PoP_ + PoP_;                                     \
        const size_t P_ = (size_t)(a).L * (size_t)(PoP_ + (D_ - 1) * kfixed_ + (a).H * ((D_ - 1) * D_ / 2));        \
        const size_t copies_ = (a).slab ? (size_t)(((a).max_n + (a).slab - 1) / (a).slab) : (size_t)1;              \
        BookArgs b_;                                                                                                 \
        b_.ring = G_ + copies_ * P_;                                                                                 \
        b_.iter_loss = batched_ ? cp_->iter_loss : (a).single.iter_loss;                                             \
        b_.st = batched_ ? cp_->state : (a).single.state;                                                            \
        b_.mirror = (a).mirror != nullptr ? (a).mirror + (clique) : nullptr;                                         \
        b_.n = batched_ ? cp_->n : (a).single.n;                                                                     \
        b_.D = D_;                                                                                                   \
        b_.chunk = (a).chunk;                                                                                        \
        b_.cfg = (a).cfg;                                                                                            \
        b_.zero_counters = (zc);                                                                                     \
        bookkeep_body(b_);                                                                                           \
    } while (0)

// The end of a chunk-persistent chunk as ONE launch of this kernel (round 6, `fused_close`): grid (Adam blocks + 1, cliques) -- blocks
// 0 .. gridDim.x - 2 of a row apply the chunk's last pending update (close_chunk mode), the row's LAST block closes the chunk (loss record,
// stop rule, step, the host's mirror).  The two need nothing of each other except the clique's step / stop as the chunk FOUND them, which
// the bookkeeping block is about to change: the Adam blocks read the copies the persistent launch left in the workspace (CLOSE_WORD_STEP /
// _STOP, written before the launch's first iteration and not touched again until the plan's next launch).  One kernel boundary and the
// closing Adam's latency less per chunk: ~8 of the ~20 us between two launches.
__global__ void __launch_bounds__(256) nsf_adam_kernel(AdamArgs a) {
    if (a.fused_close && blockIdx.x + 1 == gridDim.x) {
        NSF_BOOKKEEP_BLOCK(a, (int)blockIdx.y, 2);
        return;
    }
    // grid = (ADAM_BLOCKS, n_cliques).  state->step / stop only change in the bookkeeping kernel that closes a
    // chunk, never during this launch, so every block derives the same iteration number t.  (`fused_close`: the bookkeeping
    // block runs NEXT TO these blocks; they take step / stop from the copies the chunk-persistent launch left instead.)
    const bool batched = a.cliques != nullptr;
    const nfisam_clique* cp = batched ? (a.cliques + blockIdx.y) : nullptr;
    float* theta = batched ? cp->kparams : a.single.kparams;
    float* m = batched ? cp->adam_m : a.single.adam_m;
    float* v = batched ? cp->adam_v : a.single.adam_v;
    float* G = batched ? cp->kgrad : a.single.kgrad;
    nfisam_train_state* st = batched ? cp->state : a.single.state;
    const int n = batched ? cp->n : a.single.n;
    const int D = batched ? cp->D : a.single.D;

    __shared__ int s_step, s_stop;
    const int adam_blocks = (int)gridDim.x - (a.fused_close ? 1 : 0);      // (`fused_close`: the row's last block is the bookkeeping block)
    const int PoP = pop_of(a.K);
    const int kfixed = a.H + a.H * a.H + a.H + a.H * PoP + PoP;
    const int P = a.L * (PoP + (D - 1) * kfixed + a.H * ((D - 1) * D / 2));
    // first pass's operands are requested before the state words are consumed (one round trip, not two)
    const int pj0 = threadIdx.x & 31, tl0 = threadIdx.x >> 5, jf = blockIdx.x * 32 + pj0;
    const int n_tiles0 = a.slab ? (n + a.slab - 1) / a.slab : 1;
    float pre_g = 0.0f, pre_m = 0.0f, pre_v = 0.0f, pre_t = 0.0f;
    const bool closing = a.close_chunk > 0;
    if (jf < P && !a.few_copies && !closing) {
#pragma unroll 8
        for (int tt = tl0; tt < n_tiles0; tt += 8) pre_g += G[(size_t)tt * P + jf];
        if (tl0 == 0) { pre_m = m[jf]; pre_v = v[jf]; pre_t = theta[jf]; }
    }
    if (threadIdx.x == 0) {
        if (a.fused_close) {
            const size_t copies_max = (size_t)((a.max_n + a.slab - 1) / a.slab);
            const unsigned* words = (const unsigned*)(G + copies_max * (size_t)P + (size_t)LOSS_RING * LOSS_SLOTS);
            s_step = (int)__hip_atomic_load(&words[CLOSE_WORD_STEP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_stop = (int)__hip_atomic_load(&words[CLOSE_WORD_STOP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            s_step = __hip_atomic_load(&st->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_stop = __hip_atomic_load(&st->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    int iter_idx = a.iter_idx;
    const float *ms = m, *vs = v, *ts = theta;                  // source state (the destination is always the clique's own)
    int span_t = 0;
    if (closing && a.span) {
        // behind a window-spanning launch: the launch's own bookkeeping has advanced step / set stop already; the last block
        // to close a window left the number of the pending update and the parity of its buffers in the workspace
        const size_t copies_max = (size_t)((a.max_n + a.slab - 1) / a.slab);
        const unsigned* words = (const unsigned*)(G + copies_max * (size_t)P + (size_t)LOSS_RING * LOSS_SLOTS);
        span_t = (int)words[SPAN_WORD_LAST_T];
        if (span_t <= 0) return;                                 // (the launch found the clique finished: nothing pending)
        iter_idx = (int)(words[SPAN_WORD_LAST_PARITY] & 1u);     // (only its parity matters below)
        if (iter_idx & 1) {
            G += copies_max * (size_t)P + (size_t)LOSS_RING * LOSS_SLOTS + FUSED_COUNTERS;
            ts = G + copies_max * (size_t)P;
            ms = ts + P;
            vs = ms + P;
        }
    } else if (closing) {
        // the chunk's last valid iteration: its gradient copies and the state it started from are in the buffers of its
        // parity (odd: the second set behind the loss ring, see nsf_train1_kernel)
        const int left = a.cfg.max_iters - s_step;
        const int cnt = a.close_chunk < left ? a.close_chunk : left;
        if (cnt <= 0) return;
        iter_idx = cnt - 1;
        if (iter_idx & 1) {
            const size_t copies_max = (size_t)((a.max_n + a.slab - 1) / a.slab);
            G += copies_max * (size_t)P + (size_t)LOSS_RING * LOSS_SLOTS + FUSED_COUNTERS;
            ts = G + copies_max * (size_t)P;
            ms = ts + P;
            vs = ms + P;
        }
    }
    const int t = span_t > 0 ? span_t : s_step + iter_idx + 1;
    if (span_t <= 0 && (s_stop != 0 || t > a.cfg.max_iters)) return;

    const AdamCoef kc = adam_coef(a.cfg.lr, a.cfg.beta1, a.cfg.beta2, a.cfg.eps, a.log_b1, a.log_b2, t, n);
    // panel image of the two-dims-per-wave training kernel: [L][D][pair_panel_floats] behind the loss ring
    float* img = nullptr;
    const int Pk1 = P / a.L, img_layer = D * pair_panel_floats(a.K, a.H, D);
    if (a.pair_map != nullptr && D <= PAIR_MAX_D) {
        const size_t copies = a.slab ? (size_t)((a.max_n + a.slab - 1) / a.slab) : (size_t)1;
        img = G + copies * (size_t)P + (size_t)LOSS_RING * LOSS_SLOTS + FUSED_COUNTERS;
    }
    if (a.slab && n_tiles0 <= 8 && a.few_copies) {
        // few gradient copies (throughput launches: one copy per T tiles): one thread per parameter, the copies summed
        // in copy order (bitwise-reproducible), every load independent of the others
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < P; j += adam_blocks * blockDim.x) {
            float gv[8];
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) gv[tt] = (tt < n_tiles0) ? G[(size_t)tt * P + j] : 0.0f;
            float mj = ms[j], vj = vs[j], tj = ts[j];
            float gs = gv[0];
#pragma unroll
            for (int tt = 1; tt < 8; ++tt) gs += gv[tt];
            adam_update(kc, gs, mj, vj, tj);
            m[j] = mj;
            v[j] = vj;
            theta[j] = tj;
            if (img != nullptr) {
                const int l = j / Pk1, jj = j - l * Pk1;
                const uint32_t d = a.pair_map[a.pair_off[D] + jj];
                float* il = img + (size_t)l * img_layer;
                il[d >> 16] = tj;
                il[d & 0x7fffu] = (d & PANEL_SCALED) ? tj * kTanhScale : tj;
            }
        }
        return;
    }
    // 256 threads = 32 parameters x 8 tile-lanes: with per-tile gradient slabs every tile-lane sums the
    // tiles tt = tl, tl+8, ... of its parameter (independent loads, issued together), the 8 partial sums
    // are combined through LDS in a fixed order (bitwise-reproducible), lane 0 applies Adam.
    __shared__ float s_part[8][33];
    const int n_tiles = n_tiles0;
    const int pj = threadIdx.x & 31, tl = threadIdx.x >> 5;
    bool first = true;
    for (int j0 = blockIdx.x * 32; j0 < P; j0 += adam_blocks * 32) {
        const int j = j0 + pj;
        float part = 0.0f;
        if (first && !closing) {
            part = pre_g;
        } else if (j < P) {
#pragma unroll 8
            for (int tt = tl; tt < n_tiles; tt += 8) part += G[(size_t)tt * P + j];
        }
        s_part[tl][pj] = part;
        __syncthreads();
        if (tl == 0 && j < P) {
            float gs = s_part[0][pj];
#pragma unroll
            for (int q = 1; q < 8; ++q) gs += s_part[q][pj];
            const bool pre = first && !closing;
            float mj = pre ? pre_m : ms[j], vj = pre ? pre_v : vs[j], tj = pre ? pre_t : ts[j];
            adam_update(kc, gs, mj, vj, tj);
            m[j] = mj;
            v[j] = vj;
            theta[j] = tj;
            if (!a.slab) G[j] = 0.0f;
            if (img != nullptr) {
                const int l = j / Pk1, jj = j - l * Pk1;
                const uint32_t d = a.pair_map[a.pair_off[D] + jj];
                float* il = img + (size_t)l * img_layer;
                il[d >> 16] = tj;
                il[d & 0x7fffu] = (d & PANEL_SCALED) ? tj * kTanhScale : tj;
            }
        }
        __syncthreads();
        first = false;
    }
}

// Closes a chunk of iterations (one block per clique): turns the ring's loss sums into iter_loss entries
// (NFiSAM.py:473), evaluates the window early-stop rule (NFiSAM.py:481-491) and advances state->step.
// It is the only writer of step / stop, and it runs alone between chunks: the training and Adam kernels of a
// chunk all see the same state.  One memory round trip in front of the arithmetic: the state words and ALL 128 rows of
// the loss ring are requested together (which rows belong to this chunk is decided afterwards), the 32 row sums of a
// wave come out of one reduce-scatter (31 cross-lane exchanges instead of 32 x 6).
// `mirror` (training plans): a host-pinned copy of the state, written last with a chunk sequence number that the host
// polls for -- no device-to-host copy, no stream synchronisation per chunk (nfisam_nsf_train_plan_run).
// Behind a window-spanning launch and its closing Adam kernel (one block of 64 threads per clique): the launch's control words go
// back to zero for the next launch; a group that gave up waiting (abort bit of a dim's control word) becomes the STALLED state the
// host looks for -- the launch's own bookkeeping cannot have seen it (the window was never closed).
__global__ void __launch_bounds__(64) nsf_span_close_kernel(AdamArgs a) {
    const bool batched = a.cliques != nullptr;
    const nfisam_clique* cp = batched ? (a.cliques + blockIdx.x) : nullptr;
    float* G = batched ? cp->kgrad : a.single.kgrad;
    nfisam_train_state* st = batched ? cp->state : a.single.state;
    const int D = batched ? cp->D : a.single.D;
    const int PoP = pop_of(a.K);
    const int kfixed = a.H + a.H * a.H + a.H + a.H * PoP + PoP;
    const size_t P = (size_t)a.L * (size_t)(PoP + (D - 1) * kfixed + a.H * ((D - 1) * D / 2));
    const size_t copies = (size_t)((a.max_n + a.slab - 1) / a.slab);
    unsigned* words = (unsigned*)(G + copies * P + (size_t)LOSS_RING * LOSS_SLOTS);
    const unsigned cv = words[threadIdx.x];
    const int stalled = __any((int)(threadIdx.x < (unsigned)D ? (cv >> 31) : 0u));
    if (threadIdx.x >= (unsigned)SPAN_WORD_LAST_PARITY) words[threadIdx.x] = 0u;
    if (threadIdx.x == 0 && stalled) {
        for (int d = 0; d < D; ++d) words[d] &= 0x7fffffffu;
        st->domain_err |= NFISAM_STATE_STALLED;
        st->stop = 1;
        if (a.mirror != nullptr) {
            nfisam_train_state* m = a.mirror + blockIdx.x;
            const int seq = __hip_atomic_load(&m->reserved[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + (1 << 20);
            m->step = st->step; m->stop = 1; m->have_avg = st->have_avg; m->loss_avg = st->loss_avg; m->domain_err = st->domain_err;
            __hip_atomic_store(&m->reserved[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void __launch_bounds__(256) nsf_bookkeep_kernel(AdamArgs a) { NSF_BOOKKEEP_BLOCK(a, (int)blockIdx.x, 1); }

// Hold-out validation of a fit (reference: src/slam/NFiSAM.py:452-468, `training_set_frac < 1`): every
// `validation_interval` iterations the NLL of the held-out batch is evaluated with the CURRENT parameters (before the
// training step of that iteration); on the first increase the run is scheduled to end at slower_stop_rate x (i + 1), and no
// further evaluation takes place.  `lp` = per-sample log-density of the held-out batch (nsf_forward_kernel, enqueued in front
// of this kernel); one block, a fixed summation order.  State words: have_avg / loss_avg hold the last validation loss (the
// window rule is off in this mode, as in the reference), reserved[2] the scheduled end (0: none).
__global__ void __launch_bounds__(256) nsf_validate_kernel(const float* __restrict__ lp, int n_val, nfisam_train_state* st, float rate,
                                                           int max_iters, int interval, float* __restrict__ record) {
    __shared__ float s_w[4];
    float sm = 0.0f;
    for (int j = threadIdx.x; j < n_val; j += blockDim.x) sm += lp[j];
    sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sm;
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (st->stop != 0 || st->step >= max_iters || st->reserved[2] != 0) return;
    const float new_loss = -((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])) / (float)n_val;
    const int iter1 = st->step + 1;                                   // the reference's i + 1
    if (record != nullptr && interval > 0) record[iter1 / interval - 1] = new_loss;
    if (st->have_avg != 0 && new_loss > st->loss_avg) {
        st->reserved[2] = (int)((double)rate * (double)iter1);         // slower_stop_iter = int(slower_stop_rate * (i + 1))
    } else {
        st->loss_avg = new_loss;
        st->have_avg = 1;
    }
}

// =============================================================================================
// training-batch normalisation (reference: NFiSAM.normalize_training_samples,
// src/slam/NFiSAM.py:515-548): per column, Euclidean -> (x - mean) / std; circular -> mu = direction of the
// mean resultant (scipy.stats.circmean(., high=pi, low=-pi)), wrap(x - mu) / std(wrapped); population std
// (ddof = 0), clipped at 1e-5.  One block per column; sums in double (the reference works in float64).
// =============================================================================================
__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();                  // sh may still be read from the previous reduction
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}

__global__ void __launch_bounds__(256) nsf_normalize_kernel(const float* __restrict__ x, int n, int D,
                                                            const uint8_t* __restrict__ circ, float* __restrict__ xn,
                                                            float* __restrict__ mean, float* __restrict__ stdv) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    const bool circular = circ != nullptr && circ[c] != 0;
    const double inv_n = 1.0 / (double)n;
    double mu;
    if (circular) {
        double ss = 0.0, cs = 0.0;
        for (int r = threadIdx.x; r < n; r += blockDim.x) {
            const double v = (double)x[(size_t)r * D + c];
            ss += sin(v); cs += cos(v);
        }
        ss = block_sum(ss, sh);
        cs = block_sum(cs, sh);
        mu = atan2(ss, cs);
        if (mu >= 3.141592653589793238463) mu -= 6.283185307179586476925;      // [-pi, pi) as theta_to_pipi
    } else {
        double s = 0.0;
        for (int r = threadIdx.x; r < n; r += blockDim.x) s += (double)x[(size_t)r * D + c];
        mu = block_sum(s, sh) * inv_n;
    }
    const double two_pi = 6.283185307179586476925, pi = 3.141592653589793238463;
    // second pass: (wrapped) deviations, their mean and variance (the wrapped deviations of an angle column are
    // not centred exactly, and numpy's std subtracts their mean)
    double s1 = 0.0, s2 = 0.0;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        double d = (double)x[(size_t)r * D + c] - mu;
        if (circular) { d = fmod(d + pi, two_pi); if (d < 0.0) d += two_pi; d -= pi; }
        s1 += d; s2 += d * d;
    }
    s1 = block_sum(s1, sh) * inv_n;
    s2 = block_sum(s2, sh) * inv_n;
    double var = s2 - s1 * s1;
    if (var < 0.0) var = 0.0;
    double sd = sqrt(var);
    if (sd < 1e-5) sd = 1e-5;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        double d = (double)x[(size_t)r * D + c] - mu;
        if (circular) { d = fmod(d + pi, two_pi); if (d < 0.0) d += two_pi; d -= pi; }
        xn[(size_t)r * D + c] = (float)(d / sd);
    }
    if (threadIdx.x == 0) { mean[c] = (float)mu; stdv[c] = (float)sd; }
}

// =============================================================================================
// elementwise spline with per-element logits in memory (flows.utils.unconstrained_RQS / RQS,
// src/flows/utils.py:25-164).  Runtime K: the logits are streamed twice (softmax statistics, then
// cumulative knots + bin selection) so no per-lane arrays are needed.  HBM/latency-bound helper of
// the module surface, not part of the fused clique path.
// =============================================================================================
template <bool INV>
__global__ void __launch_bounds__(256) nsf_rqs_kernel(const float* __restrict__ inp, const float* __restrict__ Wl,
                                                      const float* __restrict__ Hl, const float* __restrict__ Dl,
                                                      int M, int K, float xl, float xr, float yb, float yt,
                                                      int padded, float* __restrict__ out, float* __restrict__ lad_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M) return;
    const float v = inp[e];
    const bool inside = INV ? (v >= yb && v <= yt) : (v >= xl && v <= xr);
    const float vs = inside ? v : (INV ? yb : xl);
    const float* w = Wl + (size_t)e * K;
    const float* h = Hl + (size_t)e * K;
    const float* d = Dl + (size_t)e * (padded ? (K - 1) : (K + 1));
    float mw = w[0], mh = h[0];
    for (int j = 1; j < K; ++j) { mw = fmaxf(mw, w[j]); mh = fmaxf(mh, h[j]); }
    float sw = 0.0f, sh = 0.0f;
    for (int j = 0; j < K; ++j) { sw += fexp(w[j] - mw); sh += fexp(h[j] - mh); }
    const float iw = frcp(sw), ih = frcp(sh);
    const float mix = 1.0f - kMinBin * (float)K;
    float cx = 0.0f, cy = 0.0f, Xl = xl, Yl = yb;
    int k = 0;
    float Xk = xl, Yk = yb, dx = 1.0f, dy = 1.0f;
    for (int j = 0; j < K; ++j) {
        cx += kMinBin + mix * fexp(w[j] - mw) * iw;
        cy += kMinBin + mix * fexp(h[j] - mh) * ih;
        const float Xr = (j == K - 1) ? xr : (xr - xl) * cx + xl;
        const float Yr = (j == K - 1) ? yt : (yt - yb) * cy + yb;
        const bool sel = INV ? (vs >= Yl) : (vs >= Xl);
        if (sel) { k = j; Xk = Xl; dx = Xr - Xl; Yk = Yl; dy = Yr - Yl; }
        Xl = Xr; Yl = Yr;
    }
    float ud0, ud1;
    if (padded) {
        ud0 = (k == 0) ? kBoundLogit : d[k - 1];
        ud1 = (k == K - 1) ? kBoundLogit : d[k];
    } else {
        ud0 = d[k];
        ud1 = d[k + 1];
    }
    const float d0 = kMinDeriv + fsoftplus(ud0), d1 = kMinDeriv + fsoftplus(ud1);
    float t, o, l;
    rq_math<INV>(vs, Xk, dx, Yk, dy, d0, d1, t, o, l);
    if (!inside) { o = v; l = 0.0f; }
    out[e] = o;
    lad_out[e] = l;
}

// =============================================================================================
// host side
// =============================================================================================
extern "C" int nfisam_abi_version(void) { return 1600; }   // 1600: the loss ring of a clique workspace has 128 x 128 words (round 6)
extern "C" int nfisam_last_hip_error(void) { return nfisam_g_last_hip_error; }

// (K, H) -> launchers of the kernel unit that instantiates the pair (nsf_units.h), nullptr if none does
static const NsfUnitOps* find_ops(int K, int H) {
    if (K < 1 || H < 1) return nullptr;
    const NsfUnitOps* o = nullptr;
#define NSF_TRY_UNIT(u) if (o == nullptr) o = nsf_unit_ops_u##u(K, H);
    NSF_UNITS(NSF_TRY_UNIT)
#undef NSF_TRY_UNIT
    return o;
}
// hidden_dim: the reference takes any width (src/flows/flows.py:26-41; its own grid lists 4, 6, 8, 10, 12:
// example/slam/manhattan_world_with_range/lawnmower_4x4/run_nfisam.py:5-6), the kernels are instantiated for 4, 8 and 16.  A
// width-H conditioner IS the next compiled width with zero rows / columns: a padded hidden unit has zero in-weights and bias
// (h = tanh(0) = 0 exactly) and zero out-weights, so every gradient that touches it is an exact zero (its activation or the
// back-propagated signal is a product with 0) and Adam leaves m = v = theta = 0 there -- the padding stays zero for the whole
// fit (asserted in tests/test_hip_parity.py).  Every entry point maps H to the compiled width first; the kernel layout of a
// width-H model is the compiled width's, `nfisam_nsf_param_count` / `nfisam_nsf_layout_map` keep the reference's counts and order.
static inline int compiled_H(int H) { return H < 1 ? H : (H <= 4 ? 4 : (H <= 8 ? 8 : (H <= 16 ? 16 : H))); }
extern "C" int nfisam_nsf_supported(int K, int H) { return find_ops(K, compiled_H(H)) != nullptr ? 1 : 0; }

static size_t torch_block(int i, int K, int H) {
    const size_t Po = 3 * (size_t)K - 1;
    return (size_t)i * H + H + (size_t)H * H + H + (size_t)H * Po + Po;
}
extern "C" size_t nfisam_nsf_param_count(int D, int K, int H) {
    if (D < 1 || K < 1 || H < 1) return 0;
    size_t c = 3 * (size_t)K - 1;
    for (int i = 1; i < D; ++i) c += torch_block(i, K, H);
    return c;
}
extern "C" size_t nfisam_nsf_kparam_count(int D, int K, int H) {
    if (D < 1 || K < 1 || H < 1) return 0;
    return kcount(D, K, compiled_H(H));
}
extern "C" int nfisam_nsf_layout_map(int D, int K, int H, int32_t* map) {
    if (D < 1 || K < 1 || H < 1 || map == nullptr) return NFISAM_ERR_ARG;
    const int Po = 3 * K - 1, PoP = pop_of(K);
    const int Hc = compiled_H(H);                              // row / column stride of the kernel layout; units H..Hc-1 are padding
    const size_t Pk = kcount(D, K, Hc);
    for (size_t j = 0; j < Pk; ++j) map[j] = -1;
    for (int o = 0; o < Po; ++o) map[out_col(K, o)] = o;
    size_t toff = Po, koff = PoP;
    for (int i = 1; i < D; ++i) {
        const size_t tW0 = toff, tb0 = tW0 + (size_t)H * i, tW1 = tb0 + H, tb1 = tW1 + (size_t)H * H,
                     tW2 = tb1 + H, tb2 = tW2 + (size_t)Po * H;
        const size_t kW0 = koff, kb0 = kW0 + (size_t)i * Hc, kW1 = kb0 + Hc, kb1 = kW1 + (size_t)Hc * Hc,
                     kW2 = kb1 + Hc, kb2 = kW2 + (size_t)Hc * PoP;
        for (int k = 0; k < i; ++k) for (int j = 0; j < H; ++j) map[kW0 + (size_t)k * Hc + j] = (int32_t)(tW0 + (size_t)j * i + k);
        for (int j = 0; j < H; ++j) map[kb0 + j] = (int32_t)(tb0 + j);
        for (int k = 0; k < H; ++k) for (int j = 0; j < H; ++j) map[kW1 + (size_t)k * Hc + j] = (int32_t)(tW1 + (size_t)j * H + k);
        for (int j = 0; j < H; ++j) map[kb1 + j] = (int32_t)(tb1 + j);
        for (int k = 0; k < H; ++k) for (int o = 0; o < Po; ++o) map[kW2 + (size_t)k * PoP + out_col(K, o)] = (int32_t)(tW2 + (size_t)o * H + k);
        for (int o = 0; o < Po; ++o) map[kb2 + out_col(K, o)] = (int32_t)(tb2 + o);
        toff += torch_block(i, K, H);
        koff = kb2 + PoP;
    }
    return (koff == Pk) ? NFISAM_OK : NFISAM_ERR_ARG;
}

extern "C" int nfisam_nsf_forward(const float* x, const float* kparams, int n, int D, int K, int H, float B,
                                  int L, size_t layer_stride, float* z, float* logdet, float* logprob,
                                  nfisam_stream_t stream) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    // an empty batch may come with null data pointers
    if ((n != 0 && x == nullptr) || kparams == nullptr || n < 0 || D < 1 || L < 1 || !(B > 0)) return NFISAM_ERR_ARG;
    if (layer_stride != 0 && layer_stride < kcount(D, K, H)) return NFISAM_ERR_ARG;
    if (n == 0) return NFISAM_OK;
    const NsfUnitOps* ops = find_ops(K, H);
    if (ops == nullptr) return NFISAM_ERR_ARG;
    return ops->forward(x, kparams, n, D, B, L, (int)layer_stride, z, logdet, logprob, (hipStream_t)stream);
}

extern "C" int nfisam_nsf_inverse(const float* z, const float* x_sep, const float* kparams, int n, int D, int Ds,
                                  int K, int H, float B, int L, size_t layer_stride, const float* mean,
                                  const float* stdv, const uint8_t* circular, float* x_out, float* logdet,
                                  nfisam_stream_t stream) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    if (D >= 1 && layer_stride != 0 && layer_stride < kcount(D, K, H)) return NFISAM_ERR_ARG;
    if ((n != 0 && (z == nullptr || x_out == nullptr)) || kparams == nullptr || n < 0 || D < 1 || Ds < 0 || Ds >= D || L < 1 ||
        !(B > 0) || (Ds > 0 && x_sep == nullptr) || (mean != nullptr && stdv == nullptr))
        return NFISAM_ERR_ARG;
    if (n == 0) return NFISAM_OK;
    const NsfUnitOps* ops = find_ops(K, H);
    if (ops == nullptr) return NFISAM_ERR_ARG;
    return ops->inverse(z, x_sep, kparams, n, D, Ds, B, L, (int)layer_stride, mean, stdv, circular, x_out, logdet,
                        (hipStream_t)stream);
}

extern "C" int nfisam_nsf_posterior_walk(const nfisam_post_clique* table, int n_cliques, const int32_t* cols,
                                         const float* obs, int max_D, int K, int H, float B, int L, int n,
                                         const float* Zt, float* St, nfisam_stream_t stream) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    if (table == nullptr || cols == nullptr || Zt == nullptr || St == nullptr || n_cliques < 0 || n < 0 || max_D < 1 ||
        L < 1 || !(B > 0))
        return NFISAM_ERR_ARG;
    if (n == 0 || n_cliques == 0) return NFISAM_OK;
    const NsfUnitOps* ops = find_ops(K, H);
    if (ops == nullptr) return NFISAM_ERR_ARG;
    return ops->walk(table, n_cliques, cols, obs, max_D, B, L, n, Zt, St, (hipStream_t)stream);
}

extern "C" int nfisam_normalize_columns(const float* x, int n, int D, const uint8_t* circular, float* x_out,
                                        float* mean, float* stdv, nfisam_stream_t stream) {
    if (n < 1 || D < 1 || x == nullptr || x_out == nullptr || mean == nullptr || stdv == nullptr) return NFISAM_ERR_ARG;
    hipLaunchKernelGGL(nsf_normalize_kernel, dim3(D), dim3(256), 0, (hipStream_t)stream, x, n, D, circular, x_out, mean,
                       stdv);
    HIP_TRY(hipGetLastError());
    return NFISAM_OK;
}

extern "C" int nfisam_rqs(const float* inputs, const float* widths, const float* heights, const float* derivs,
                          int M, int K, int inverse, float left, float right, float bottom, float top,
                          int padded_derivatives, float* out, float* logabsdet, nfisam_stream_t stream) {
    if (inputs == nullptr || widths == nullptr || heights == nullptr || derivs == nullptr || out == nullptr ||
        logabsdet == nullptr || M < 0 || K < 1 || !(right > left) || !(top > bottom) || (padded_derivatives && K < 1) ||
        kMinBin * (float)K > 1.0f)
        return NFISAM_ERR_ARG;
    if (M == 0) return NFISAM_OK;
    const dim3 grid((M + 255) / 256), block(256);
    if (inverse)
        hipLaunchKernelGGL((nsf_rqs_kernel<true>), grid, block, 0, (hipStream_t)stream, inputs, widths, heights, derivs,
                           M, K, left, right, bottom, top, padded_derivatives ? 1 : 0, out, logabsdet);
    else
        hipLaunchKernelGGL((nsf_rqs_kernel<false>), grid, block, 0, (hipStream_t)stream, inputs, widths, heights, derivs,
                           M, K, left, right, bottom, top, padded_derivatives ? 1 : 0, out, logabsdet);
    HIP_TRY(hipGetLastError());
    return NFISAM_OK;
}

// hidden_dim 4: can a multi-layer / dL/dx launch of this shape take the two-dims-per-wave kernel?  Asked of the LARGEST
// num_knots of that width (the panels grow with K), so that the answer -- and with it the tile size, the number of
// gradient copies and the workspace layout -- does not depend on K.
static bool pair_h4_fits(int L, int max_D, int H = 4) {
    const NsfUnitOps* o = find_ops(16, H);                  // (H = 4 or 16: the hidden widths without a two-lanes-per-particle kernel)
    return o != nullptr && o->pair_lds(L, max_D) > 0;
}

extern "C" int nfisam_nsf_backward(const float* x, const float* kparams, int n, int D, int K, int H, float B, int L,
                                   size_t layer_stride, const float* gz, const float* gl, int nll_mode, float* kgrad,
                                   float* gx, float* loss_sum, nfisam_stream_t stream) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    if (D >= 1 && layer_stride != 0 && (layer_stride < kcount(D, K, H) || (layer_stride & 3) != 0)) return NFISAM_ERR_ARG;
    if ((n != 0 && x == nullptr) || kparams == nullptr || kgrad == nullptr || n < 0 || D < 1 || L < 1 || !(B > 0) ||
        (!nll_mode && gz == nullptr))
        return NFISAM_ERR_ARG;
    if (n == 0) return NFISAM_OK;
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.single.x = x;
    a.single.kparams = const_cast<float*>(kparams);
    a.single.kgrad = kgrad;
    a.single.n = n;
    a.single.D = D;
    a.gz = gz; a.gl = gl; a.gx = gx; a.loss_sum = loss_sum;
    a.B = B; a.L = L; a.max_iters = 0x7fffffff; a.nll_mode = nll_mode ? 1 : 0; a.layer_stride = (int)layer_stride;
    const NsfUnitOps* ops = find_ops(K, H);
    if (ops == nullptr) return NFISAM_ERR_ARG;
    a.tile = train_tile(1, n, D, H, false, (H == 4 || H == 16) && (L > 1 || gx != nullptr) && pair_h4_fits(L, D, H));
    return ops->train(a, 1, n, D, (hipStream_t)stream);
}

// Small launches use per-tile gradient slabs (kgrad must then hold n_tiles copies, see
// nfisam_nsf_grad_workspace_count); large ones accumulate with atomics into a single copy.
static int slab_max_tiles() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("NFISAM_SLAB_MAX_TILES");
        v = (e != nullptr) ? atoi(e) : 128;
        if (v < 0) v = 0;
    }
    return v;
}
static bool use_slabs(int max_n, int tile) { return (max_n + tile - 1) / tile <= slab_max_tiles(); }

// Upper bound over both kernel families (the launch shape is picked per launch; several tiles per block only
// lower the number of copies).
extern "C" size_t nfisam_nsf_grad_workspace_count(int n, int D, int K, int H, int L) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    if (n < 1 || D < 1 || K < 1 || H < 1 || L < 1) return 0;
    const size_t tiles = use_slabs(n, TILE2) ? (size_t)((n + TILE2 - 1) / TILE2)
                                             : (use_slabs(n, TILE) ? (size_t)((n + TILE - 1) / TILE) : 1);
    // + the fused-Adam launches' second set of gradient copies (64-particle tiles) and second state buffer (theta | m | v)
    const size_t tiles64 = (size_t)((n + TILE - 1) / TILE);
    // + the chunk-persistent form's two sets of TAGGED copies (8 blocks of 4 waves at most, 2 floats per parameter)
    // (round 5: groups of up to sixteen blocks -- n <= 4096 -- in the chunk-persistent form: 2 sets x copies x 2 floats)
    // (round 6: two lanes per particle, nsf_half.h -- blocks of 128 particles while a group keeps to sixteen of them)
    const size_t copies128 = (tiles64 + 1) / 2;
    const size_t c128 = copies128 < (size_t)PERSIST_MAX_COPIES ? copies128 : (size_t)PERSIST_MAX_COPIES;
    const size_t copies16 = c128 > (tiles64 + 3) / 4 ? c128 : (tiles64 + 3) / 4;                // (monotone in n)
    const size_t fused = (L == 1 && tiles64 <= (size_t)FUSED_MAX_COPIES) ? (tiles64 + 3 + 4 * (copies16 > 8 ? copies16 : 8) + 2) * kcount(D, K, H) : 0;   // (+ the theta exchange of the divided update: 2 floats per parameter)
    // + the panel image of multi-layer cliques (nsf_train3_kernel; maintained by the Adam kernel, nsf_cond_mfma.h)
    const size_t image = (L > 1 && (H == 8 || H == 4 || H == 16) && D <= PAIR_MAX_D) ? (size_t)L * D * pair_panel_floats(K, H, D) : 0;
    // + the forward state that kernel parks between its forward and backward passes (latency-bound launches only)
    const size_t stash = (image > 0 && pair_stash_fits(n, D))
                             ? (size_t)((n + TILE2 - 1) / TILE2) * (size_t)(L - 1) * (size_t)((D + 1) / 2) * pair_stash_fields(K, H) * 64
                             : 0;
    return tiles * (size_t)L * kcount(D, K, H) + (size_t)LOSS_RING * LOSS_SLOTS + (size_t)FUSED_COUNTERS + fused + image + stash;
}

// Launch shape of one training iteration: kernel family, tiles per block, particles per gradient copy (0 = one
// shared copy accumulated with float atomics).
struct TrainShape { int tile, T, slab, W, half; };
static TrainShape train_shape(int n_cliques, int max_n, int max_D, int L, int H, int K) {
    TrainShape sh;
    sh.tile = train_tile(n_cliques, max_n, max_D, H, L == 1, (H == 4 || H == 16) && L > 1 && pair_h4_fits(L, max_D, H));
    sh.T = tiles_per_block(n_cliques, max_n, max_D, L, sh.tile, H);
    sh.slab = use_slabs(max_n, sh.tile) ? sh.tile * sh.T : 0;      // the workspace holds ceil(n / tile) copies at most
    sh.W = 0;
    sh.half = 0;
    if (is_dim_major(n_cliques, max_n, max_D, L, sh.tile, H)) {
        sh.W = dim_major_waves(n_cliques, max_n, max_D, sh.T);
        sh.slab *= sh.W;                                            // one copy per block
        // two lanes per particle (nsf_half.h): 32 particles per wave; a function of the launch shape only, like everything
        // here, so that the gradient, Adam and bookkeeping launches of an iteration agree on the number of gradient copies
        if (sh.slab != 0 && half_shape(n_cliques, max_n, max_D, K, H, L, sh.T)) {
            sh.half = 1;
            sh.W = half_waves();
            sh.slab = 32 * sh.T * sh.W;
        }
    }
    return sh;
}

// The Adam update of iteration j rides at the START of iteration j+1's gradient kernel (nsf_train1_kernel with the
// MFMA conditioner, nsf_cond_mfma.h) when the launch is dim-major with few gradient copies; the chunk's last update is
// applied by nsf_adam_kernel in `close_chunk` mode.  Saves one kernel boundary (~3-4 us) per iteration.
static bool fused_adam_shape(int n_cliques, int max_n, int max_D, int L, int H, const TrainShape& sh) {
    const char* e = getenv("NFISAM_FUSED_ADAM");
    if (e != nullptr && e[0] == '0') return false;
    if (!is_dim_major(n_cliques, max_n, max_D, L, sh.tile, H) || sh.slab == 0) return false;
    return (max_n + sh.slab - 1) / sh.slab <= 8;              // nsf_adam_kernel's one-thread-per-parameter summation order
}

// Multi-layer training launches that go to nsf_train3_kernel keep a PANEL IMAGE per clique (behind the loss ring): the Adam
// kernel of iteration j writes it, the gradient kernel of iteration j + 1 copies it (iteration 0 of a chunk stages from
// the parameters themselves: whoever set them -- the caller, an earlier run -- did not go through the Adam kernel).
static bool pair_image_shape(int max_D, int K, int H, int L, const TrainShape& sh) {
    if (L < 2 || sh.tile != TILE2 || max_D > PAIR_MAX_D) return false;
    const char* e = getenv("NFISAM_PAIR_IMAGE");
    if (e != nullptr && e[0] == '0') return false;
    const NsfUnitOps* ops = find_ops(K, H);
    return ops != nullptr && ops->pair_lds(L, max_D) > 0;
}

static int enqueue_grad(const nfisam_clique* dev_cliques, const nfisam_clique* single, int n_cliques, int max_n,
                        int max_D, int K, int H, float B, int L, int max_iters, int iter_idx, hipStream_t s,
                        const nfisam_adam_cfg* fused_cfg = nullptr, const nfisam_clique* host_cliques = nullptr,
                        int chain = 0, int n_chains = 1, bool pair_image = false, int persist_iters = 0, int span_window = 0,
                        nfisam_train_state* span_mirror = nullptr) {
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.span_window = span_window;
    a.span_mirror = span_mirror;
    const TrainShape sh = train_shape(n_cliques, max_n, max_D, L, H, K);
    a.tile = sh.tile; a.tiles_per_block = sh.T; a.slab = sh.slab; a.waves = sh.W; a.half = sh.half;
    if (fused_cfg != nullptr) {
        a.fused_adam = 1;
        a.adam = *fused_cfg;
        a.log_b1 = (float)log((double)fused_cfg->beta1);
        a.log_b2 = (float)log((double)fused_cfg->beta2);
    }
    a.cliques = dev_cliques;
    a.host_cliques = host_cliques;
    a.chain = chain; a.n_chains = n_chains;
    if (single != nullptr) a.single = *single;
    a.B = B; a.L = L; a.max_iters = max_iters; a.nll_mode = 1; a.iter_idx = iter_idx;
    a.persist_iters = persist_iters;                           // > 0: iterations 0 .. persist_iters - 1 of the chunk in this one launch
    a.pair_image = (pair_image && iter_idx > 0) ? 1 : 0;
    a.pair_ws = 1;                                             // clique descriptors: kgrad is a workspace by contract
    const NsfUnitOps* ops = find_ops(K, H);
    if (ops == nullptr) return NFISAM_ERR_ARG;
    return ops->train(a, n_cliques, max_n, max_D, s);
}

static void fill_adam_args(AdamArgs& ad, const nfisam_clique* dev_cliques, const nfisam_clique* single, int n_cliques,
                           int max_n, int max_D, int K, int H, int L, const nfisam_adam_cfg* cfg) {
    memset(&ad, 0, sizeof(ad));
    ad.cliques = dev_cliques;
    if (single != nullptr) ad.single = *single;
    ad.cfg = *cfg;
    ad.slab = train_shape(n_cliques, max_n, max_D, L, H, K).slab;
    ad.log_b1 = (float)log((double)cfg->beta1);
    ad.log_b2 = (float)log((double)cfg->beta2);
    ad.L = L; ad.K = K; ad.H = H; ad.max_n = max_n;
}

// The chunk-persistent form of the dim-major kernel (nsf_unit.hip: nsf_train1_kernel<K, H, true>) needs every block of the
// launch resident at once: blocks spin at their group's barrier, a member that waits for a CU held by spinning blocks
// would never arrive (the kernel gives up after 2^15 looks -- tens of milliseconds, NFISAM_PERSIST_SPINS -- and the run ends with NFISAM_ERR_STALL: loud, but a lost fit).
// One 4-wave block per (clique, dim, 256 particles); one barrier counter per dim: D <= 64.  How many blocks the device holds
// is ASKED, not assumed: hipOccupancyMaxActiveBlocksPerMultiprocessor for the launch's LDS size x the device's compute units
// (MI355X: H <= 8 compiles to three waves per SIMD = three blocks per CU while the four wave tiles + panel stay within
// 53 KB, i.e. D <= ~20, two beyond; H = 16: 256 VGPRs, two), of which a launch may take 7/8 -- the dispatcher is not asked to
// pack perfectly, and a block of another kernel may sit on a CU for a while.  Other PROCESSES on the device are not seen by
// this count; what protects against them is the barrier's timeout.  NFISAM_PERSIST=0: never.
static inline double mono_seconds() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static std::atomic<bool> g_persist_broken{false};     // a persistent launch of this process stalled once: never again (nfisam_nsf_train_plan_run)
static bool persist_shape(const nfisam_clique* host, int n_cliques, int max_n, int max_D, int K, int H, int L,
                          long* blocks_out = nullptr, long* places_out = nullptr) {
    static const bool on = !(getenv("NFISAM_PERSIST") != nullptr && getenv("NFISAM_PERSIST")[0] == '0');
    if (!on || g_persist_broken.load() || host == nullptr || L != 1 || (H != 16 && H != 8 && H != 4) || max_D > FUSED_COUNTERS) return false;
    const TrainShape sh = train_shape(n_cliques, max_n, max_D, L, H, K);
    // (the one-launch-per-iteration fused form sums at most eight gradient copies -- fused_adam_shape --; the persistent
    //  exchange takes up to sixteen, round 5: single cliques of up to 4096 particles, whose plain graph is then gradient
    //  kernel + Adam kernel per iteration; nsf_adam_kernel's lane-partial order for that many copies is the order the
    //  persistent staging sums in, so the two graphs of such a plan still leave the same bits)
    {
        const char* fe = getenv("NFISAM_FUSED_ADAM");
        if (fe != nullptr && fe[0] == '0') return false;
    }
    if (!is_dim_major(n_cliques, max_n, max_D, L, sh.tile, H) || sh.T != 1 || sh.slab == 0 || (sh.W != 4 && !sh.half) ||
        (max_n + sh.slab - 1) / sh.slab > PERSIST_MAX_COPIES)
        return false;
    long blocks = 0;
    for (int c = 0; c < n_cliques; ++c) blocks += (long)host[c].D * ((host[c].n + sh.slab - 1) / sh.slab);
    // (asked once per (K, H, clique width, device): the query costs a device-properties call, and replica schedulers
    //  create plans by the hundred)
    static std::mutex mu;
    static std::vector<std::pair<long, long>> cache;             // key -> places
    int devn = 0;
    (void)hipGetDevice(&devn);
    const long key = ((long)devn << 40) | ((long)K << 32) | ((long)H << 24) | (long)max_D;
    long places = -1;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const auto& kv : cache) if (kv.first == key) places = kv.second;
    }
    if (places < 0) {
        const NsfUnitOps* ops = find_ops(K, H);
        places = ops != nullptr ? ops->persist_places(max_D) : 0;
        std::lock_guard<std::mutex> lk(mu);
        cache.emplace_back(key, places);
    }
    long limit = places - places / 8;
    if (const char* e = getenv("NFISAM_PERSIST_BLOCKS")) limit = atol(e);    // (measurement aid)
    if (blocks_out != nullptr) *blocks_out = blocks;
    if (places_out != nullptr) *places_out = places;
    return blocks <= limit;
}

// ---- "probe before persisting" -----------------------------------------------------------------------------------------------
// The occupancy API answers for THIS process.  A second process on the device (two ranks sharing a GPU, somebody else's job)
// is invisible to it, and a persistent launch whose late blocks queue behind a foreign kernel spins until its members give
// up (NFISAM_ERR_STALL: a lost fit).  So before a plan takes the chunk-persistent form the device is ASKED, once: a launch of
// as many trivial blocks as the plan's launch has -- 256 threads and the LDS footprint that limits a CU to the same number
// of blocks -- in which every block arrives at a counter and waits until all have arrived or ~200 us of wall clock
// (s_memrealtime, 100 MHz) have passed.  All there in time: the device holds that many blocks of ours at once right now.
// A block that times out says so; the plan then keeps to one launch per iteration (same bits).  The answer is cached per
// device for half a second (replica schedulers create plans by the hundred); NFISAM_PERSIST_PROBE=0 skips the probe.
// A point-in-time answer by construction: what protects a run against a neighbour that arrives LATER is the timeout above.
__global__ void __launch_bounds__(256) nsf_coresidency_probe_kernel(unsigned* ctr, unsigned n_blocks, unsigned budget_ticks) {
    extern __shared__ unsigned probe_lds[];
    // every wave of the block stays until its first thread has the answer (a block of the real kernel holds four waves too)
    if (threadIdx.x == 0) {
        probe_lds[0] = 0u;
        __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (threadIdx.x == 0) {
            unsigned done = 0u;
            if (__hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_blocks) done = 1u;
            else if (__hip_atomic_load(&ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) done = 1u;       // somebody gave up: everybody leaves
            else if (wall_clock64() - t0 > (unsigned long long)budget_ticks) {
                __hip_atomic_fetch_add(&ctr[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = 1u;
            }
            if (done) __hip_atomic_store(&probe_lds[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (__hip_atomic_load(&probe_lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) break;
        __builtin_amdgcn_s_sleep(8);
    }
}
// -> true: `blocks` blocks of `per_cu` per compute unit are resident at once right now (or the probe is switched off / failed to run:
// the occupancy answer stands)
// compute units of a device, asked once per device (hipGetDeviceProperties is not cheap and plans are created by the hundred)
static int cu_count(int dev) {
    static std::mutex mu;
    static int cus[64] = {};
    if (dev < 0 || dev >= 64) return 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (cus[dev] != 0) return cus[dev] > 0 ? cus[dev] : 0;
    }
    hipDeviceProp_t prop;
    const int n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : -1;
    std::lock_guard<std::mutex> lk(mu);
    cus[dev] = n;
    return n > 0 ? n : 0;
}
static bool device_is_quiet(long blocks, long places) {
    static const bool on = !(getenv("NFISAM_PERSIST_PROBE") != nullptr && getenv("NFISAM_PERSIST_PROBE")[0] == '0');
    if (!on || blocks < 1 || places < 1) return true;
    static std::mutex mu;
    struct Seen { int dev; long blocks; long per_cu; double at; bool quiet; };   // (an answer is reused for the SAME footprint only)
    static std::vector<Seen> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    const double now = mono_seconds();
    const int cus = cu_count(dev);
    if (cus < 1) return true;
    const long per_cu = places / cus;
    if (per_cu < 1) return true;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Seen& q : seen)
            if (q.dev == dev && q.per_cu == per_cu && now - q.at < 0.5 && (q.quiet ? q.blocks >= blocks : q.blocks <= blocks)) return q.quiet;
    }
    // LDS per block such that per_cu blocks fit a CU's 160 KB and per_cu + 1 do not
    size_t lds = (size_t)(160 * 1024) / (size_t)(per_cu + 1) + 1024;
    if (lds > (size_t)(160 * 1024) / (size_t)per_cu) lds = (size_t)(160 * 1024) / (size_t)per_cu;
    lds &= ~(size_t)255;
    bool quiet = true;
    unsigned* ctr = nullptr;
    hipStream_t st = nullptr;
    if (hipFuncSetAttribute((const void*)nsf_coresidency_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
        hipMalloc((void**)&ctr, 2 * sizeof(unsigned)) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
        hipMemsetAsync(ctr, 0, 2 * sizeof(unsigned), st) == hipSuccess) {
        hipLaunchKernelGGL(nsf_coresidency_probe_kernel, dim3((unsigned)blocks), dim3(256), lds, st, ctr, (unsigned)blocks, 20000u);
        unsigned out[2] = {0u, 0u};
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(out, ctr, sizeof(out), hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess)
            quiet = out[1] == 0u;
        if (getenv("NFISAM_PROBE_DEBUG") != nullptr)
            fprintf(stderr, "nfisam probe: %ld blocks (%ld places, %ld per CU, %zu B of LDS each): %u arrived, %u gave up, %.3f ms\n", blocks, places,
                    per_cu, lds, out[0], out[1], 1e3 * (mono_seconds() - now));
    }
    if (st) (void)hipStreamDestroy(st);
    if (ctr) (void)hipFree(ctr);
    if (!quiet) {
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "nfisam: the device did not hold %ld blocks of this process at once (another process is using it): "
                            "training plans keep to one launch per iteration while that lasts\n", blocks);
    }
    std::lock_guard<std::mutex> lk(mu);
    for (size_t q = 0; q < seen.size();)
        if (seen[q].dev == dev && now - seen[q].at >= 0.5) seen.erase(seen.begin() + (long)q); else ++q;
    seen.push_back(Seen{dev, blocks, per_cu, now, quiet});
    return quiet;
}

// iteration `iter_idx` of the current chunk: gradient kernel + Adam kernel
static int enqueue_step(const nfisam_clique* dev_cliques, const nfisam_clique* single, int n_cliques, int max_n,
                        int max_D, int K, int H, float B, int L, const nfisam_adam_cfg* cfg, int iter_idx,
                        hipStream_t s, const nfisam_clique* host_cliques = nullptr, int chain = 0, int n_chains = 1,
                        int persist_iters = 0) {
    const TrainShape sh = train_shape(n_cliques, max_n, max_D, L, H, K);
    const bool fused = fused_adam_shape(n_cliques, max_n, max_D, L, H, sh) || persist_iters > 0;   // (the persistent form applies its own updates: up to 16 copies)
    if (!fused && n_chains > 1) return NFISAM_ERR_ARG;
    const bool image = !fused && pair_image_shape(max_D, K, H, L, sh);
    int rc = enqueue_grad(dev_cliques, single, n_cliques, max_n, max_D, K, H, B, L, cfg->max_iters, iter_idx, s,
                          fused ? cfg : nullptr, host_cliques, chain, n_chains, image, persist_iters);
    if (rc || fused) return rc;
    AdamArgs ad;
    fill_adam_args(ad, dev_cliques, single, n_cliques, max_n, max_D, K, H, L, cfg);
    ad.iter_idx = iter_idx;
    if (image) {
        rc = find_ops(K, H)->pair_map(&ad.pair_map, ad.pair_off);
        if (rc) return rc;
    }
    // 32 parameters per block (x 8 tile-lanes); a few hundred small blocks spread the latency-bound work
    const size_t Pmax = (size_t)L * kcount(max_D, K, H);
    int ablocks = (int)((Pmax + 31) / 32);
    if (ablocks < 1) ablocks = 1;
    if (ablocks > 1024) ablocks = 1024;   // one pass for up to 32 k parameters (C2: 275 blocks; a second pass is a second memory round trip)
    ad.few_copies = (ad.slab != 0 && (max_n + ad.slab - 1) / ad.slab <= 8) ? 1 : 0;
    if (ad.few_copies) ablocks = (int)((Pmax + 255) / 256);
    hipLaunchKernelGGL(nsf_adam_kernel, dim3(ablocks, n_cliques), dim3(256), 0, s, ad);
    HIP_TRY(hipGetLastError());
    return NFISAM_OK;
}

// closes a chunk of `chunk` iterations: loss record, early-stop rule, step counter
static int enqueue_bookkeeping(const nfisam_clique* dev_cliques, const nfisam_clique* single, int n_cliques, int max_n,
                               int max_D, int K, int H, int L, const nfisam_adam_cfg* cfg, int chunk, hipStream_t s,
                               nfisam_train_state* mirror = nullptr) {
    AdamArgs ad;
    fill_adam_args(ad, dev_cliques, single, n_cliques, max_n, max_D, K, H, L, cfg);
    ad.chunk = chunk;
    ad.mirror = mirror;
    hipLaunchKernelGGL(nsf_bookkeep_kernel, dim3(n_cliques), dim3(256), 0, s, ad);
    HIP_TRY(hipGetLastError());
    return NFISAM_OK;
}

// end of a chunk: (fused-Adam launches) the last iteration's pending update, then the bookkeeping
static int enqueue_chunk_end(const nfisam_clique* dev_cliques, const nfisam_clique* single, int n_cliques, int max_n,
                             int max_D, int K, int H, int L, const nfisam_adam_cfg* cfg, int chunk, hipStream_t s,
                             nfisam_train_state* mirror = nullptr, bool persistent_chunk = false) {
    // (`persistent_chunk`: the chunk ran as a chunk-persistent launch, whose last update is always pending -- also for
    //  groups of 9 .. 16 copies, which the one-launch-per-iteration form does not fuse)
    if (persistent_chunk || fused_adam_shape(n_cliques, max_n, max_D, L, H, train_shape(n_cliques, max_n, max_D, L, H, K))) {
        AdamArgs ad;
        fill_adam_args(ad, dev_cliques, single, n_cliques, max_n, max_D, K, H, L, cfg);
        ad.close_chunk = chunk;
        const size_t Pmax = (size_t)L * kcount(max_D, K, H);
        ad.few_copies = ((max_n + ad.slab - 1) / ad.slab <= 8) ? 1 : 0;
        int ablocks = ad.few_copies ? (int)((Pmax + 255) / 256) : (int)((Pmax + 31) / 32);
        if (ablocks < 1) ablocks = 1;
        if (ablocks > 1024) ablocks = 1024;
        // behind a chunk-persistent launch of cliques of up to SPAN_MAX_D dims: ONE kernel, the bookkeeping block next to the Adam
        // blocks (nsf_adam_kernel with `fused_close`; NFISAM_FUSED_CLOSE=0: two kernels, the same bits)
        static const bool fuse_on = !(getenv("NFISAM_FUSED_CLOSE") != nullptr && getenv("NFISAM_FUSED_CLOSE")[0] == '0');
        if (persistent_chunk && fuse_on && max_D <= SPAN_MAX_D) {
            ad.fused_close = 1;
            ad.chunk = chunk;
            ad.mirror = mirror;
            hipLaunchKernelGGL(nsf_adam_kernel, dim3(ablocks + 1, n_cliques), dim3(256), 0, s, ad);
            HIP_TRY(hipGetLastError());
            return NFISAM_OK;
        }
        hipLaunchKernelGGL(nsf_adam_kernel, dim3(ablocks, n_cliques), dim3(256), 0, s, ad);
        HIP_TRY(hipGetLastError());
    }
    return enqueue_bookkeeping(dev_cliques, single, n_cliques, max_n, max_D, K, H, L, cfg, chunk, s, mirror);
}

// Iterations per chunk: the early-stop rule is evaluated when a chunk is closed, so the chunk length has to
// divide the window (the rule then only ever fires on a chunk's last iteration, as in the reference loop).
static int chunk_length(const nfisam_adam_cfg* cfg) {
    const int wnd = cfg->average_window;
    if (wnd <= 0) return 50;
    int c = wnd < LOSS_RING ? wnd : LOSS_RING;
    while (wnd % c != 0) --c;
    return c;
}

static int check_cfg(const nfisam_adam_cfg* cfg, int K, int H, int L, float B) {
    if (cfg == nullptr || L < 1 || !(B > 0) || !nfisam_nsf_supported(K, H)) return NFISAM_ERR_ARG;
    if (!(cfg->lr > 0) || !(cfg->beta1 >= 0 && cfg->beta1 < 1) || !(cfg->beta2 >= 0 && cfg->beta2 < 1) ||
        cfg->max_iters < 0)
        return NFISAM_ERR_ARG;
    return NFISAM_OK;
}

// Parallel launches per iteration of a training plan (see nfisam_nsf_train_plan_create): NFISAM_CHAINS=n, default by size.
static int plan_chains(int n_cliques, int max_n, int max_D, int K, int H, int L) {
    (void)K;
    const TrainShape sh = train_shape(n_cliques, max_n, max_D, L, H, K);
    if (!fused_adam_shape(n_cliques, max_n, max_D, L, H, sh)) return 1;
    const long waves = (long)n_cliques * max_D * ((max_n + 64 * sh.T - 1) / (64 * sh.T));
    // measured (MI355X): C3 (3072 waves) 15.0 -> 14.5 us per iteration, 64 cliques (7680 waves) 89.5 -> 82.5;
    // one Plaza clique (480 waves) 11.3 -> 11.7: stays one launch
    int chains = waves >= 1536 ? 2 : 1;
    if (const char* ce = getenv("NFISAM_CHAINS")) chains = atoi(ce);
    const int octets = (n_cliques * max_D + 7) / 8;
    if (chains > octets) chains = octets;
    if (chains < 1) chains = 1;
    if (chains > 8) chains = 8;
    return chains;
}

extern "C" int nfisam_nsf_train_chains(int n_cliques, int max_n, int max_D, int K, int H, int L) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    if (n_cliques < 1 || max_n < 1 || max_D < 1 || L < 1 || !nfisam_nsf_supported(K, H)) return 1;
    return plan_chains(n_cliques, max_n, max_D, K, H, L);
}

extern "C" int nfisam_nsf_train_gradient_part(const nfisam_clique* cliques, int n_cliques, int cliques_on_host, int max_n,
                                              int max_D, int K, int H, float B, int L, int chain, int n_chains,
                                              nfisam_stream_t stream) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    if (cliques == nullptr || n_cliques < 1 || max_n < 1 || max_D < 1 || L < 1 || !(B > 0) ||
        !nfisam_nsf_supported(K, H) || n_chains < 1 || chain < 0 || chain >= n_chains)
        return NFISAM_ERR_ARG;
    if (n_chains > 1 && !fused_adam_shape(n_cliques, max_n, max_D, L, H, train_shape(n_cliques, max_n, max_D, L, H, K)))
        return NFISAM_ERR_ARG;
    if (cliques_on_host) {
        if (n_cliques != 1) return NFISAM_ERR_ARG;
        return enqueue_grad(nullptr, cliques, 1, max_n, max_D, K, H, B, L, 0x7fffffff, 0, (hipStream_t)stream, nullptr,
                            nullptr, chain, n_chains);
    }
    return enqueue_grad(cliques, nullptr, n_cliques, max_n, max_D, K, H, B, L, 0x7fffffff, 0, (hipStream_t)stream, nullptr,
                        nullptr, chain, n_chains);
}

extern "C" int nfisam_nsf_train_gradient(const nfisam_clique* cliques, int n_cliques, int cliques_on_host, int max_n,
                                         int max_D, int K, int H, float B, int L, nfisam_stream_t stream) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    if (cliques == nullptr || n_cliques < 1 || max_n < 1 || max_D < 1 || L < 1 || !(B > 0) ||
        !nfisam_nsf_supported(K, H))
        return NFISAM_ERR_ARG;
    if (cliques_on_host) {
        if (n_cliques != 1) return NFISAM_ERR_ARG;
        return enqueue_grad(nullptr, cliques, 1, max_n, max_D, K, H, B, L, 0x7fffffff, 0, (hipStream_t)stream);
    }
    return enqueue_grad(cliques, nullptr, n_cliques, max_n, max_D, K, H, B, L, 0x7fffffff, 0, (hipStream_t)stream);
}

extern "C" int nfisam_nsf_train_step(const nfisam_clique* cliques, int n_cliques, int cliques_on_host, int max_n,
                                     int max_D, int K, int H, float B, int L, const nfisam_adam_cfg* cfg,
                                     nfisam_stream_t stream) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    int rc = check_cfg(cfg, K, H, L, B);
    if (rc) return rc;
    if (cliques == nullptr || n_cliques < 1 || max_n < 1 || max_D < 1) return NFISAM_ERR_ARG;
    const nfisam_clique* dev = cliques_on_host ? nullptr : cliques;
    const nfisam_clique* single = cliques_on_host ? cliques : nullptr;
    if (cliques_on_host && n_cliques != 1) return NFISAM_ERR_ARG;
    rc = enqueue_step(dev, single, n_cliques, max_n, max_D, K, H, B, L, cfg, 0, (hipStream_t)stream);
    if (rc) return rc;
    return enqueue_chunk_end(dev, single, n_cliques, max_n, max_D, K, H, L, cfg, 1, (hipStream_t)stream);
}

// ---- training plan: descriptors + (optionally) a hipGraph of `chunk` iterations, built once ----
static std::atomic<int> g_hand_stepped{0};            // hand-stepped runs between `begin` and `end` (conveyors of chunks: they fill the machine)
static std::atomic<bool> g_persist_busy{false};      // a run of a chunk-persistent graph is in flight in this process (nfisam_nsf_train_plan_run)

struct nfisam_train_plan {
    std::vector<nfisam_clique> host;
    const nfisam_clique* dev = nullptr;
    int n_cliques = 0, K = 0, H = 0, L = 0, max_n = 0, max_D = 0, chunk = 0;
    float B = 0;
    nfisam_adam_cfg cfg;
    hipStream_t cap = nullptr;
    hipEvent_t ev = nullptr;
    std::vector<hipStream_t> side;         // capture-time streams of the extra chains (parallel branches of the graph)
    std::vector<hipEvent_t> side_ev;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph_p = nullptr;          // the same chunk as ONE chunk-persistent launch per chain (persist_shape), or null
    hipGraphExec_t exec_p = nullptr;
    // round 6: the WHOLE run as one window-spanning persistent launch (single-clique plans whose launch takes a two-wave
    // build: every fit of a real NF-iSAM run) + its closing Adam kernel + the control words' reset, or null
    hipGraph_t graph_s = nullptr;
    hipGraphExec_t exec_s = nullptr;
    bool span_run = false;                // nfisam_nsf_train_plan_run takes the window-spanning graph (NFISAM_SPAN=1); else it serves nfisam_nsf_train_plan_launch_async only
    nfisam_train_state* hst = nullptr;     // pinned, device-mapped host copy of the cliques' states: the bookkeeping kernel
                                           // writes it (last word written: reserved[0] = chunks closed in this run)
    nfisam_train_state* hst_dev = nullptr; // the same memory as the device addresses it
    std::atomic<bool> ahead{false};        // the previous run left a chunk enqueued behind its early stop (the feeder thread sets it too)
    // hand-stepped runs (nfisam_nsf_train_plan_begin ..): a feeder thread keeps `feed_depth` chunks enqueued ahead of the
    // last one closed, so that the caller's thread (Python, in the replica scheduler) never sits in hipGraphLaunch
    int device = 0;
    std::thread feeder;
    std::atomic<int> feed_depth{0};        // 0: paused
    std::atomic<int> feed_busy{0};         // the feeder is inside a launch
    std::atomic<int> feed_quit{0};
    std::atomic<int> feed_error{0};
    std::atomic<long> enqueued{0};         // chunks enqueued since `begin`
    std::atomic<int> refills{0};           // slots refilled since the feeder's last launch (their mirror entries still say "stopped")
    std::vector<hipEvent_t> slot_ev;       // nfisam_nsf_train_plan_refill: orders a slot's copies behind the caller's stream
    hipEvent_t ev_end = nullptr;           // `end` records this one: the caller's stream may still hold the wait on it when the next
                                           // `begin` records p->ev -- re-recording an event a stream still waits for ties that wait to
                                           // the NEW record on this runtime (the stream then waits for itself)
    // hold-out validation (nfisam_nsf_train_plan_create_validated): a chunk is one validation period
    hipStream_t last_work = nullptr;             // the stream the last run's chunks went to (a chunk may still drain there: `ahead`)
    bool ran = false;
    hipEvent_t evk0 = nullptr, evk1 = nullptr;   // (use_graph & 2) timing events recorded around a persistent chunk's training launches
    hipGraph_t graph_end = nullptr;              // ... whose chunk end (closing Adam + bookkeeping) is then a graph of its own
    hipGraphExec_t exec_end = nullptr;
    std::vector<nfisam_validation> val;    // per clique, or empty
    float val_rate = 0.0f;
    bool stepping = false;                 // between `begin` and `end` (counted in g_hand_stepped)
    std::mutex enqueue_mu;                 // a chunk's graph launch and a slot's refill must not interleave on the stream: a graph
                                           // launch is not one atomic enqueue for a second thread (a state reset landed mid-chunk)
};

extern "C" int nfisam_nsf_train_plan_destroy(nfisam_train_plan* p) {
    if (p == nullptr) return NFISAM_OK;
    if (p->feeder.joinable()) {
        p->feed_depth.store(0);
        p->feed_quit.store(1);
        p->feeder.join();
    }
    if (p->stepping) { p->stepping = false; g_hand_stepped.fetch_sub(1); }
    if (p->cap) (void)hipStreamSynchronize(p->cap);       // a chunk enqueued ahead of an early stop may still be draining
    if (p->ahead && p->ran && p->last_work != p->cap) (void)hipStreamSynchronize(p->last_work);   // ... on the caller's stream (null = the legacy stream)
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    if (p->exec_p) (void)hipGraphExecDestroy(p->exec_p);
    if (p->graph_p) (void)hipGraphDestroy(p->graph_p);
    if (p->exec_s) (void)hipGraphExecDestroy(p->exec_s);
    if (p->graph_s) (void)hipGraphDestroy(p->graph_s);
    if (p->exec_end) (void)hipGraphExecDestroy(p->exec_end);
    if (p->graph_end) (void)hipGraphDestroy(p->graph_end);
    if (p->ev) (void)hipEventDestroy(p->ev);
    if (p->evk0) (void)hipEventDestroy(p->evk0);
    if (p->evk1) (void)hipEventDestroy(p->evk1);
    for (hipEvent_t e : p->side_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->slot_ev) if (e) (void)hipEventDestroy(e);
    if (p->ev_end) (void)hipEventDestroy(p->ev_end);
    for (hipStream_t st : p->side) (void)hipStreamDestroy(st);
    if (p->cap) (void)hipStreamDestroy(p->cap);
    if (p->hst) (void)hipHostFree(p->hst);
    delete p;
    return NFISAM_OK;
}

// One validation period of a validated plan on stream `s`: interval - 1 iterations (one chunk: its own closing Adam update
// and bookkeeping, nothing published), the held-out NLL with the parameters as they stand and the reference's rule
// (nsf_validate_kernel), then the period's last iteration as a chunk of one, whose bookkeeping publishes the state.
// `persist`: the first part as ONE chunk-persistent launch.
static int enqueue_validated_period(const nfisam_train_plan* p, hipStream_t s, bool persist);

static int plan_create_impl(const nfisam_clique* host_cliques, const nfisam_clique* dev_cliques, int n_cliques, int K, int H, float B,
                            int L, const nfisam_adam_cfg* cfg, const nfisam_validation* val, int validation_interval,
                            float slower_stop_rate, int use_graph, nfisam_train_plan** out);

extern "C" int nfisam_nsf_train_plan_create(const nfisam_clique* host_cliques, const nfisam_clique* dev_cliques,
                                            int n_cliques, int K, int H, float B, int L,
                                            const nfisam_adam_cfg* cfg, int use_graph, nfisam_train_plan** out) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    return plan_create_impl(host_cliques, dev_cliques, n_cliques, K, H, B, L, cfg, nullptr, 0, 0.0f, use_graph, out);
}

extern "C" int nfisam_nsf_train_plan_create_validated(const nfisam_clique* host_cliques, const nfisam_clique* dev_cliques,
                                                      int n_cliques, int K, int H, float B, int L, const nfisam_adam_cfg* cfg,
                                                      const nfisam_validation* val, int validation_interval,
                                                      float slower_stop_rate, int use_graph, nfisam_train_plan** out) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    // the scheduled end int(rate x (i + 1)) must fall on a period boundary: whole-number rates (the reference's default is 2.0)
    if (val == nullptr || validation_interval < 1 || validation_interval > LOSS_RING + 1 || !(slower_stop_rate >= 1.0f) ||
        slower_stop_rate != (float)(int)slower_stop_rate)
        return NFISAM_ERR_ARG;
    for (int c = 0; c < n_cliques; ++c)
        if (val[c].x_val == nullptr || val[c].logprob == nullptr || val[c].n_val < 1) return NFISAM_ERR_ARG;
    return plan_create_impl(host_cliques, dev_cliques, n_cliques, K, H, B, L, cfg, val, validation_interval, slower_stop_rate, use_graph, out);
}

static int plan_create_impl(const nfisam_clique* host_cliques, const nfisam_clique* dev_cliques, int n_cliques, int K, int H, float B,
                            int L, const nfisam_adam_cfg* cfg, const nfisam_validation* val, int validation_interval,
                            float slower_stop_rate, int use_graph, nfisam_train_plan** out) {
    int rc = check_cfg(cfg, K, H, L, B);
    if (rc) return rc;
    if (out == nullptr || host_cliques == nullptr || n_cliques < 1 || (n_cliques > 1 && dev_cliques == nullptr))
        return NFISAM_ERR_ARG;
    nfisam_train_plan* p = new nfisam_train_plan();
    p->host.assign(host_cliques, host_cliques + n_cliques);
    p->dev = dev_cliques;
    p->n_cliques = n_cliques; p->K = K; p->H = H; p->L = L; p->B = B; p->cfg = *cfg;
    (void)hipGetDevice(&p->device);
    p->chunk = chunk_length(cfg);
    if (val != nullptr) {                  // hold-out validation: one chunk = one validation period, the window rule is off (NFiSAM.py:481: `if testing_data is None`)
        p->val.assign(val, val + n_cliques);
        p->val_rate = slower_stop_rate;
        p->chunk = validation_interval;
        p->cfg.average_window = 0;
    }
    if (hipHostMalloc((void**)&p->hst, sizeof(nfisam_train_state) * (size_t)n_cliques,
                      hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&p->hst_dev, p->hst, 0) != hipSuccess) {
        nfisam_nsf_train_plan_destroy(p);
        return NFISAM_ERR_LAUNCH;
    }
    memset(p->hst, 0, sizeof(nfisam_train_state) * (size_t)n_cliques);
    for (int c = 0; c < n_cliques; ++c) {
        if (host_cliques[c].n < 1 || host_cliques[c].D < 1) { nfisam_nsf_train_plan_destroy(p); return NFISAM_ERR_ARG; }
        p->max_n = host_cliques[c].n > p->max_n ? host_cliques[c].n : p->max_n;
        p->max_D = host_cliques[c].D > p->max_D ? host_cliques[c].D : p->max_D;
    }
    {   // device tables of the training kernels: built here, outside the capture below
        const NsfUnitOps* ops = find_ops(K, H);
        rc = ops != nullptr ? ops->prepare(p->max_D) : NFISAM_ERR_ARG;
        if (rc) { nfisam_nsf_train_plan_destroy(p); return rc; }
    }
    if (use_graph) {
        // capture on a private stream: the caller's stream may be the (un-capturable) null stream
        int status = NFISAM_OK;
        hipError_t e = hipStreamCreateWithFlags(&p->cap, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev, hipEventDisableTiming);
        if (e == hipSuccess && (use_graph & 2) != 0) {          // measurement: the chunk's training launches between two timing events
            e = hipEventCreate(&p->evk0);
            if (e == hipSuccess) e = hipEventCreate(&p->evk1);
        }
        // Chains: with one layer the (clique, dim) groups are independent optimisation problems for a whole chunk (every
        // group's blocks read and write that group's parameters, moments and gradient copies only), so an iteration may
        // be split into n launches that form n PARALLEL branches of the graph: the branches drift apart, and one
        // branch's kernel prologue / tail (memory round trips, a barrier, nothing to issue) runs under another's
        // arithmetic.  Pays in the latency regime (few waves per SIMD); NFISAM_CHAINS=n, default by launch size.
        // (a plan that is ONE short chunk keeps one launch per iteration: the two-branch graph costs ~12 us more to launch,
        //  which 20 iterations do not earn back -- C3, 20 iterations: 18.2 vs 18.8 us per iteration; with several chunks the
        //  launch of chunk k + 1 hides behind chunk k and two branches win from 20 iterations per chunk up: 15.2 -> 14.5)
        const bool one_short_chunk = cfg->max_iters <= p->chunk && p->chunk < 40 && getenv("NFISAM_CHAINS") == nullptr;
        const int chains = (one_short_chunk || !p->val.empty()) ? 1 : plan_chains(n_cliques, p->max_n, p->max_D, K, H, L);
        for (int g = 1; g < chains && e == hipSuccess; ++g) {
            hipStream_t st = nullptr;
            hipEvent_t ev2 = nullptr;
            e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e == hipSuccess) { p->side.push_back(st); e = hipEventCreateWithFlags(&ev2, hipEventDisableTiming); }
            if (e == hipSuccess) p->side_ev.push_back(ev2);
        }
        long p_blocks = 0, p_places = 0;
        bool can_persist = persist_shape(p->host.data(), n_cliques, p->max_n, p->max_D, K, H, L, &p_blocks, &p_places) && p->chunk > (p->val.empty() ? 1 : 2);
        // probe before persisting (device_is_quiet): is the device ours right now?  (outside the capture below)
        // (not while a hand-stepped conveyor of this process fills the machine -- slam.ReplicaNFiSAM creates plans by the hundred
        //  next to one: the probe's blocks would queue behind its chunks, and no run takes the persistent graph then anyway)
        // (a launch of few blocks may take a two-wave build with helper waves -- nsf_unit.hip: unit_train1 -- and then a whole CU per
        //  block: asked for at that footprint)
        if (can_persist && device_cus() > 0 && p_blocks <= device_cus() - device_cus() / 16 && p_places > device_cus()) p_places = device_cus();
        if (can_persist && g_hand_stepped.load() == 0 && !device_is_quiet(p_blocks, p_places)) can_persist = false;
        for (int pass = 0; pass < (can_persist ? 2 : 1) && e == hipSuccess && status == NFISAM_OK; ++pass) {
        const bool persist = pass == 1;
        hipGraph_t* graph_out = persist ? &p->graph_p : &p->graph;
        // (the persistent form is ONE launch over all groups: its blocks are resident for the whole chunk, there is no kernel
        //  boundary for a second branch to hide -- two launches side by side measured 13.5 against 13.0 us per C3 iteration, and
        //  304 against 283 us for a 20-iteration launch)
        const int ch = persist ? 1 : chains;
        if (e == hipSuccess) e = hipStreamBeginCapture(p->cap, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            const nfisam_clique* single = (p->dev == nullptr) ? p->host.data() : nullptr;
            for (int g = 1; g < ch && e == hipSuccess; ++g) {      // fork
                e = hipEventRecord(p->ev, p->cap);
                if (e == hipSuccess) e = hipStreamWaitEvent(p->side[g - 1], p->ev, 0);
            }
            if (!p->val.empty()) status = enqueue_validated_period(p, p->cap, persist);
            for (int it = 0; p->val.empty() && it < (persist ? 1 : p->chunk) && status == NFISAM_OK && e == hipSuccess; ++it)
                for (int g = 0; g < ch && status == NFISAM_OK; ++g)
                    status = enqueue_step(p->dev, single, n_cliques, p->max_n, p->max_D, K, H, B, L, &p->cfg, it,
                                          g == 0 ? p->cap : p->side[g - 1], p->host.data(), g, ch, persist ? p->chunk : 0);
            for (int g = 1; g < ch && e == hipSuccess; ++g) {      // join
                e = hipEventRecord(p->side_ev[g - 1], p->side[g - 1]);
                if (e == hipSuccess) e = hipStreamWaitEvent(p->cap, p->side_ev[g - 1], 0);
            }
            // (measurement plans: the persistent chunk's graph ends behind the training launches; its closing Adam update and
            //  bookkeeping are a graph of their own, so that two timing events can be recorded on the stream in between --
            //  events recorded INSIDE a captured graph cannot be asked for their elapsed time)
            const bool split_end = persist && p->evk0 != nullptr && p->val.empty();
            if (status == NFISAM_OK && e == hipSuccess && p->val.empty() && !split_end)
                status = enqueue_chunk_end(p->dev, single, n_cliques, p->max_n, p->max_D, K, H, L, &p->cfg, p->chunk,
                                           p->cap, p->hst_dev, persist);
            e = hipStreamEndCapture(p->cap, graph_out);
            if (split_end && e == hipSuccess && status == NFISAM_OK) {
                e = hipStreamBeginCapture(p->cap, hipStreamCaptureModeThreadLocal);
                if (e == hipSuccess) {
                    status = enqueue_chunk_end(p->dev, single, n_cliques, p->max_n, p->max_D, K, H, L, &p->cfg, p->chunk, p->cap, p->hst_dev, true);
                    e = hipStreamEndCapture(p->cap, &p->graph_end);
                }
                if (e == hipSuccess && status == NFISAM_OK) e = hipGraphInstantiate(&p->exec_end, p->graph_end, nullptr, nullptr, 0);
            }
        }
        if (e == hipSuccess && status == NFISAM_OK)
            e = hipGraphInstantiate(persist ? &p->exec_p : &p->exec, *graph_out, nullptr, nullptr, 0);
        }
        if (e != hipSuccess || status != NFISAM_OK) {
            if (e != hipSuccess) nfisam_g_last_hip_error = (int)e;
            nfisam_nsf_train_plan_destroy(p);
            return status != NFISAM_OK ? status : NFISAM_ERR_LAUNCH;
        }
        // ---- the run as ONE window-spanning launch (nsf_unit.hip: the clique's blocks close every window themselves) ----------
        // A 50-iteration chunk of a real fit is ~340 us of launch and ~36 us of fixed cost around and inside it (the launch's cold
        // first iteration and drain, the gap between two graph launches, the closing Adam and bookkeeping kernels): 10 % of a fit.
        // Single-clique plans only (every fit of a real run; a batch's cliques stop at different windows), no hold-out validation,
        // not the measurement plans; the launcher refuses when the launch would not take a two-wave build (the in-kernel
        // bookkeeping exists in those) -- then the plan simply has no such graph.  NFISAM_SPAN=0: never.
        // Measured (scripts/exp/span_windows.py, one Plaza clique, 400 iterations as windows of 25 / 50 / 100): a window's end costs
        // ~10 us inside the launch against ~20 us between two launches (gap 8.9 + closing Adam 4.8 + bookkeeping 6.8), but the run
        // itself ~55 us more (the long launch iterates ~2 % slower than four short ones): Plaza1's fits, ~12 windows each, gain 1.5 %.
        // Not worth being the default of the last round: OFF unless NFISAM_SPAN=1 (read per plan: tests switch it in-process).
        const char* span_env = getenv("NFISAM_SPAN");
        const bool span_on = (span_env != nullptr && span_env[0] == '1') || (use_graph & 4) != 0;      // (bit 2 of use_graph: the caller wants nfisam_nsf_train_plan_launch_async)
        p->span_run = span_env != nullptr && span_env[0] == '1';
        if (span_on && p->exec_p != nullptr && n_cliques == 1 && p->val.empty() && L == 1 && (use_graph & 2) == 0 && p->max_D <= SPAN_MAX_D &&
            cfg->max_iters > p->chunk && cfg->average_window > 0) {
            const nfisam_clique* single = (p->dev == nullptr) ? p->host.data() : nullptr;
            hipError_t es = hipStreamBeginCapture(p->cap, hipStreamCaptureModeThreadLocal);
            int ss = NFISAM_ERR_LAUNCH;
            if (es == hipSuccess) {
                ss = enqueue_grad(p->dev, single, n_cliques, p->max_n, p->max_D, K, H, B, L, p->cfg.max_iters, 0, p->cap, &p->cfg, p->host.data(),
                                  0, 1, false, p->cfg.max_iters, p->chunk, p->hst_dev);
                if (ss == NFISAM_OK) {
                    AdamArgs ad;
                    fill_adam_args(ad, p->dev, single, n_cliques, p->max_n, p->max_D, K, H, L, &p->cfg);
                    ad.close_chunk = p->chunk;
                    ad.span = 1;
                    ad.mirror = p->hst_dev;
                    const size_t Pmax = (size_t)L * kcount(p->max_D, K, H);
                    ad.few_copies = ((p->max_n + ad.slab - 1) / ad.slab <= 8) ? 1 : 0;
                    int ablocks = ad.few_copies ? (int)((Pmax + 255) / 256) : (int)((Pmax + 31) / 32);
                    if (ablocks < 1) ablocks = 1;
                    if (ablocks > 1024) ablocks = 1024;
                    hipLaunchKernelGGL(nsf_adam_kernel, dim3(ablocks, n_cliques), dim3(256), 0, p->cap, ad);
                    hipLaunchKernelGGL(nsf_span_close_kernel, dim3(n_cliques), dim3(64), 0, p->cap, ad);
                    if (hipGetLastError() != hipSuccess) ss = NFISAM_ERR_LAUNCH;
                }
                es = hipStreamEndCapture(p->cap, &p->graph_s);
            }
            if (es == hipSuccess && ss == NFISAM_OK) es = hipGraphInstantiate(&p->exec_s, p->graph_s, nullptr, nullptr, 0);
            if (getenv("NFISAM_SPAN_DEBUG") != nullptr)
                fprintf(stderr, "nfisam: window-spanning graph of a plan (n %d, D %d, chunk %d, max_iters %d): capture %d, launcher %d\n", p->max_n, p->max_D,
                        p->chunk, (int)cfg->max_iters, (int)es, ss);
            if (es != hipSuccess || ss != NFISAM_OK) {             // (not an error of the plan: it keeps to one launch per chunk)
                if (p->exec_s) { (void)hipGraphExecDestroy(p->exec_s); p->exec_s = nullptr; }
                if (p->graph_s) { (void)hipGraphDestroy(p->graph_s); p->graph_s = nullptr; }
                (void)hipGetLastError();
            }
        }
    }
    *out = p;
    return NFISAM_OK;
}

static int enqueue_validated_period(const nfisam_train_plan* p, hipStream_t s, bool persist) {
    const nfisam_clique* single = (p->dev == nullptr) ? p->host.data() : nullptr;
    const int head = p->chunk - 1;                              // iterations in front of the evaluation
    int rc = NFISAM_OK;
    if (head > 0) {
        if (persist && head > 1) {
            rc = enqueue_step(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->B, p->L, &p->cfg, 0, s, p->host.data(), 0, 1, head);
        } else {
            for (int it = 0; it < head && rc == NFISAM_OK; ++it)
                rc = enqueue_step(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->B, p->L, &p->cfg, it, s, p->host.data());
        }
        if (rc == NFISAM_OK)
            rc = enqueue_chunk_end(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->L, &p->cfg, head, s, nullptr, persist && head > 1);
        if (rc) return rc;
    }
    const NsfUnitOps* ops = find_ops(p->K, p->H);
    if (ops == nullptr) return NFISAM_ERR_ARG;
    for (int c = 0; c < p->n_cliques; ++c) {
        const nfisam_clique& q = p->host[c];
        const nfisam_validation& v = p->val[(size_t)c];
        rc = ops->forward(v.x_val, q.kparams, v.n_val, q.D, p->B, p->L, 0, nullptr, nullptr, v.logprob, s);
        if (rc) return rc;
        hipLaunchKernelGGL(nsf_validate_kernel, dim3(1), dim3(256), 0, s, (const float*)v.logprob, (int)v.n_val, q.state, p->val_rate,
                           (int)p->cfg.max_iters, p->chunk, v.val_loss);
        HIP_TRY(hipGetLastError());
    }
    rc = enqueue_step(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->B, p->L, &p->cfg, 0, s, p->host.data());
    if (rc) return rc;
    return enqueue_chunk_end(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->L, &p->cfg, 1, s, p->hst_dev);
}

// Waits until the bookkeeping kernel of chunk number `k` (1-based, this run) has written every clique's mirror.  The
// device publishes the sequence word last (system-scope release); a chunk takes 0.1-1 ms.  All bounds are TIMES
// (CLOCK_MONOTONIC), not look counts: the host pause-spins for the first ~50 us (the common case at the end of a short
// chunk), yields (`sched_yield`: free when nobody else wants the core -- a timed sleep costs ~70 us of timer slack per look,
// 5 % of the driver's 0.3 ms 20-step plan) up to 2 ms, and from then on sleeps 50 us between looks, so that a worker thread or a
// rank of a parallel run does not hold a core for the whole of a slow chunk; every ~20 ms the WORK stream -- the one the
// chunk was enqueued on, which for plans without a graph is the caller's -- is queried for a launch failure, and after 60 s
// without progress the wait gives up (-> `poisoned`: the caller must not synchronise a wedged stream either).
static int wait_chunk(const nfisam_train_plan* p, int k, hipStream_t work, bool* poisoned) {
    const volatile nfisam_train_state* m = p->hst;
    const struct timespec nap = {0, 50000};
    static const double give_up_s = getenv("NFISAM_WAIT_SECONDS") != nullptr ? atof(getenv("NFISAM_WAIT_SECONDS")) : 60.0;   // (test knob)
    double t0 = 0.0, last_query = 0.0;
    for (long looks = 0;; ++looks) {
        bool all = true;
        for (int c = 0; c < p->n_cliques; ++c)
            if (m[c].reserved[0] < k) { all = false; break; }
        if (all) break;
        if (looks < 256) { __builtin_ia32_pause(); continue; }         // (~10 us before the first clock read)
        const double now = mono_seconds();
        if (t0 == 0.0) { t0 = now; last_query = now; }
        const double waited = now - t0;
        if (waited > give_up_s) { *poisoned = true; return NFISAM_ERR_LAUNCH; }
        if (now - last_query > 0.02) {
            last_query = now;
            if (hipStreamQuery(work) == hipErrorLaunchFailure) return NFISAM_ERR_LAUNCH;
        }
        if (waited < 50e-6) __builtin_ia32_pause();
        else if (waited < 2e-3) sched_yield();
        else nanosleep(&nap, nullptr);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return NFISAM_OK;
}

// The WHOLE run of a single-clique plan enqueued as one window-spanning launch (+ the closing Adam kernel and the reset of its control
// words) -- and back to the caller at once: the launch closes its windows itself (nsf_unit.hip), so nothing of the run needs the host.
// The outcome is in the clique's `state` (step = iterations run, stop, domain_err incl. NFISAM_STATE_STALLED) and `iter_loss` when the
// stream has drained; the caller looks when it wants to (slam.NFiSAM with `async_fits`: once per update).  -> NFISAM_ERR_ARG when the
// plan has no such graph (created without bit 2 of `use_graph`, or its shape does not take the window-spanning form) or the
// chunk-persistent form is not available to this process right now: the caller falls back to nfisam_nsf_train_plan_run.
extern "C" int nfisam_nsf_train_plan_launch_async(nfisam_train_plan* p, nfisam_stream_t stream) {
    if (p == nullptr || p->exec_s == nullptr || p->feeder.joinable() || p->stepping) return NFISAM_ERR_ARG;
    if (g_persist_broken.load() || g_hand_stepped.load() != 0 || g_persist_busy.load()) return NFISAM_ERR_ARG;
    hipStream_t work = (hipStream_t)stream;
    if (p->ahead && p->ran && p->last_work != work) HIP_TRY(hipStreamSynchronize(p->last_work));   // (an earlier run left work on ANOTHER stream)
    p->last_work = work;
    p->ran = true;
    p->ahead = true;                                   // a later synchronous run, and plan_destroy, drain this stream first
    const hipError_t e = hipGraphLaunch(p->exec_s, work);
    if (e != hipSuccess) { nfisam_g_last_hip_error = (int)e; return NFISAM_ERR_LAUNCH; }
    return NFISAM_OK;
}

extern "C" int nfisam_nsf_train_plan_run(nfisam_train_plan* p, int32_t* iters_run, nfisam_stream_t stream) {
    if (p == nullptr) return NFISAM_ERR_ARG;
    if (p->feeder.joinable()) {                          // a hand-stepped run's feeder must not launch into this one
        p->feed_depth.store(0);
        while (p->feed_busy.load() != 0) __builtin_ia32_pause();
    }
    hipStream_t user = (hipStream_t)stream;
    hipStream_t work = user;
    const nfisam_clique* single = (p->dev == nullptr) ? p->host.data() : nullptr;
    // The chunks' graphs were captured on the plan's private stream; they are REPLAYED on the caller's own stream (round 4:
    // two event record / wait pairs per run, ~12 us of a 20-iteration plan's 330, bought nothing -- a graph may be launched into
    // any stream, the legacy null stream included; NFISAM_PLAN_STREAM=private restores the hand-over).
    static const bool own_stream = getenv("NFISAM_PLAN_STREAM") != nullptr && strcmp(getenv("NFISAM_PLAN_STREAM"), "private") == 0;
    if (p->exec && (own_stream || p->stepping)) {
        HIP_TRY(hipEventRecord(p->ev, user));          // order after prior work on the caller's stream
        HIP_TRY(hipStreamWaitEvent(p->cap, p->ev, 0));
        work = p->cap;
    }
    if (p->ahead && p->ran && p->last_work != work) {              // (the previous run left a chunk draining on ANOTHER stream)
        HIP_TRY(hipStreamSynchronize(p->last_work));
        p->ahead = false;
    }
    p->last_work = work;
    p->ran = true;
    // An error return must not leave graph work running on buffers the caller is about to reset or free: every
    // failure path drains the work stream first.
    // The chunk-persistent graph needs its blocks resident at once (persist_shape): ONE run per process uses it at a time
    // (two of them could each hold the places the other's late blocks wait for); the others, and hand-stepped runs, take
    // the plain graph -- same results bit for bit.
    // (not next to a conveyor either: its launches take every place as soon as one is free, the persistent blocks would
    //  spin at their barriers for members that queue behind them)
    const bool persist = p->exec_p != nullptr && !g_persist_broken.load() && g_hand_stepped.load() == 0 && !g_persist_busy.exchange(true);
    hipGraphExec_t const exec = persist ? p->exec_p : p->exec;
    struct Release {
        bool on; nfisam_train_plan* p; hipStream_t* w;
        ~Release() {
            if (!on) return;
            if (p->ahead) { (void)hipStreamSynchronize(*w); p->ahead = false; }   // (behind a stop: one launch that returns at once)
            g_persist_busy.store(false);
        }
    } release{persist, p, &work};
    bool poisoned = false;            // the wait gave up on a wedged stream: return without draining it (the plan is unusable)
    auto fail = [&](int rc) { if (!poisoned) (void)hipStreamSynchronize(work); else p->ahead = false; return rc; };
    // Nothing of an earlier run writes the mirror any more: its last closed chunk was waited for, and a chunk enqueued
    // ahead of an early stop only republishes the final state, so restarting the sequence needs that chunk drained.
    if (p->ahead) { HIP_TRY(hipStreamSynchronize(work)); p->ahead = false; }
    for (int c = 0; c < p->n_cliques; ++c) p->hst[c].reserved[0] = 0;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    const int total_chunks = (p->cfg.max_iters + p->chunk - 1) / p->chunk;
    int launched = 0, closed = 0, status = NFISAM_OK;
    // the whole run as ONE window-spanning launch (plan_create_impl): every window's bookkeeping inside it publishes the mirror as a
    // chunk's bookkeeping kernel would, so the loop below simply has all its chunks "launched"
    const bool span = persist && p->exec_s != nullptr && p->span_run && total_chunks > 1;
    if (span) {
        const hipError_t e = hipGraphLaunch(p->exec_s, work);
        if (e != hipSuccess) { nfisam_g_last_hip_error = (int)e; return fail(NFISAM_ERR_LAUNCH); }
        launched = total_chunks;
    }
    auto launch_chunk = [&]() -> int {
        const int left = p->cfg.max_iters - launched * p->chunk;
        const int todo = left < p->chunk ? left : p->chunk;       // a final partial chunk is enqueued eagerly
        if (exec && todo == p->chunk && persist && p->exec_end != nullptr) {      // (measurement plan, see plan_create_impl)
            hipError_t e = hipEventRecord(p->evk0, work);
            if (e == hipSuccess) e = hipGraphLaunch(exec, work);
            if (e == hipSuccess) e = hipEventRecord(p->evk1, work);
            if (e == hipSuccess) e = hipGraphLaunch(p->exec_end, work);
            if (e != hipSuccess) { nfisam_g_last_hip_error = (int)e; return NFISAM_ERR_LAUNCH; }
        } else if (exec && todo == p->chunk) {
            // A chunk-persistent chunk is TWO kernels (the launch, the kernel that closes it): launched as such, not as a graph (round 6:
            // hipGraphLaunch of a two-node graph costs more host time than the two launches -- Plaza1's fits 3.45 -> 3.41 s, the 20-step
            // bench line -0.1 us per step, same bits; NFISAM_PERSIST_DIRECT=0 replays the captured graph as before).  The plain form's
            // chunk (up to 2 x 128 kernels) stays a graph.
            static const bool direct = !(getenv("NFISAM_PERSIST_DIRECT") != nullptr && getenv("NFISAM_PERSIST_DIRECT")[0] == '0');
            if (direct && persist && p->val.empty()) {
                int rc = enqueue_step(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->B, p->L, &p->cfg, 0, work, p->host.data(), 0, 1, p->chunk);
                if (rc == NFISAM_OK)
                    rc = enqueue_chunk_end(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->L, &p->cfg, p->chunk, work, p->hst_dev, true);
                if (rc) return rc;
            } else {
                const hipError_t e = hipGraphLaunch(exec, work);
                if (e != hipSuccess) { nfisam_g_last_hip_error = (int)e; return NFISAM_ERR_LAUNCH; }
            }
        } else if (!p->val.empty() && todo == p->chunk) {       // a validated plan without a graph: the same period, launch by launch
            int rcv = enqueue_validated_period(p, work, false);
            if (rcv) return rcv;
        } else {
            for (int it = 0; it < todo; ++it) {
                int rc = enqueue_step(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->B, p->L,
                                      &p->cfg, it, work, p->host.data());
                if (rc) return rc;
            }
            int rcb = enqueue_chunk_end(p->dev, single, p->n_cliques, p->max_n, p->max_D, p->K, p->H, p->L, &p->cfg,
                                        todo, work, p->hst_dev);
            if (rcb) return rcb;
        }
        ++launched;
        return NFISAM_OK;
    };
    // Chunk k + 1 is enqueued BEFORE the host looks at the outcome of chunk k: the stop flags are checked on the device
    // (a chunk behind an early stop is ~chunk empty launches), so the GPU never waits for the host between chunks.
    // NFISAM_RUN_AHEAD=0 restores launch - wait - launch.
    static const bool run_ahead = !(getenv("NFISAM_RUN_AHEAD") != nullptr && getenv("NFISAM_RUN_AHEAD")[0] == '0');
    if (total_chunks > 0 && !span) {
        int rc = launch_chunk();
        if (rc) return fail(rc);
    }
    while (closed < launched) {
        if (run_ahead && launched < total_chunks) {
            int rc = launch_chunk();
            if (rc) return fail(rc);
        }
        int rc = wait_chunk(p, closed + 1, work, &poisoned);
        if (rc) return fail(rc);
        ++closed;
        bool all_stopped = true;
        for (int c = 0; c < p->n_cliques; ++c) {
            if ((p->hst[c].domain_err & NFISAM_STATE_STALLED) != 0) {          // a group barrier of a persistent chunk timed out
                status = NFISAM_ERR_STALL;
                if (!g_persist_broken.exchange(true))
                    fprintf(stderr, "nfisam: a chunk-persistent training launch stalled (a block of it never became resident); "
                                    "this process keeps to one launch per iteration from now on\n");
            } else if (p->hst[c].domain_err != 0 && status != NFISAM_ERR_STALL) {
                status = NFISAM_ERR_DOMAIN;
            }
            if (!p->hst[c].stop && p->hst[c].step < p->cfg.max_iters) all_stopped = false;
        }
        if (all_stopped) break;        // (a clique with a domain error has stop set: the others run to their own end)
        if (!run_ahead && launched < total_chunks) {
            rc = launch_chunk();
            if (rc) return fail(rc);
        }
    }
    p->ahead = closed < launched || span;      // an enqueued chunk behind the stop: empty launches still draining on `work` (span: its closing kernels)
    if (total_chunks == 0) {           // max_iters = 0: report the state as it is
        HIP_TRY(hipMemcpyAsync(p->hst, p->host[0].state, sizeof(nfisam_train_state), hipMemcpyDeviceToHost, work));
        for (int c = 1; c < p->n_cliques; ++c)
            HIP_TRY(hipMemcpyAsync(&p->hst[c], p->host[c].state, sizeof(nfisam_train_state), hipMemcpyDeviceToHost, work));
        HIP_TRY(hipStreamSynchronize(work));
    }
    if (work != user) {                // the caller's stream continues behind everything enqueued here
        HIP_TRY(hipEventRecord(p->ev, work));
        HIP_TRY(hipStreamWaitEvent(user, p->ev, 0));
    }
    if (iters_run != nullptr) for (int c = 0; c < p->n_cliques; ++c) iters_run[c] = p->hst[c].step;
    return status;
}

// GPU time of the training launches of the plan's most recent PERSISTENT chunk replay (a plan created with `use_graph | 2`:
// the chunk's closing Adam update and bookkeeping are a second graph and two timing events are recorded on the stream
// around the first).  The caller has synchronised with the plan's work.  Measurement only: one more graph launch per chunk.
extern "C" int nfisam_nsf_train_plan_kernel_ms(const nfisam_train_plan* p, float* ms) {
    if (p == nullptr || ms == nullptr || p->evk0 == nullptr || p->evk1 == nullptr) return NFISAM_ERR_ARG;
    HIP_TRY(hipEventElapsedTime(ms, p->evk0, p->evk1));
    return NFISAM_OK;
}

// Most XCDs one (clique, dim) group of the plan's chunk-persistent launches ran on, as of the last closed chunk (0: no
// persistent chunk has run; 1: the placement the grid asks for).  Diagnostic: the exchange inside a group is agent-scope
// coherent, so a larger value costs time (L2 misses), not correctness -- tests use it to prove that a scattered launch
// really straddled XCDs.
extern "C" int nfisam_nsf_train_plan_xcd_span(const nfisam_train_plan* p) {
    if (p == nullptr || p->hst == nullptr) return -1;
    int span = 0;
    const volatile nfisam_train_state* m = p->hst;
    for (int c = 0; c < p->n_cliques; ++c) span = m[c].reserved[1] > span ? m[c].reserved[1] : span;
    return span;
}

// ---- stepping a plan by hand (slam.ReplicaNFiSAM's slot scheduler) ---------------------------------------------------------
// A graph plan as a conveyor of chunks: the caller enqueues chunks ahead, looks at the host mirror without blocking, and
// swaps a finished clique's slot for a new problem BETWEEN chunks by enqueueing the re-initialisation (new batch, fresh
// parameters, zeroed moments / state / loss record) on the plan's own stream.  A stopped clique's launches exit early and
// nothing writes its buffers, so the slot may wait any number of chunks for its refill.
extern "C" int nfisam_nsf_train_plan_begin(nfisam_train_plan* p, nfisam_stream_t stream) {
    if (p == nullptr || p->exec == nullptr) return NFISAM_ERR_ARG;
    if (p->feeder.joinable()) {                          // (a feeder left running by a caller that skipped `end`)
        p->feed_depth.store(0);
        while (p->feed_busy.load() != 0) __builtin_ia32_pause();
    }
    hipStream_t user = (hipStream_t)stream;
    HIP_TRY(hipEventRecord(p->ev, user));              // order after prior work on the caller's stream
    HIP_TRY(hipStreamWaitEvent(p->cap, p->ev, 0));
    if (p->ahead) {
        if (p->ran && p->last_work != p->cap) HIP_TRY(hipStreamSynchronize(p->last_work));     // (left there by nfisam_nsf_train_plan_run)
        HIP_TRY(hipStreamSynchronize(p->cap));
        p->ahead = false;
    }
    for (int c = 0; c < p->n_cliques; ++c) p->hst[c].reserved[0] = 0;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    p->enqueued.store(0);
    p->refills.store(0);
    p->feed_error.store(0);
    if (!p->stepping) { p->stepping = true; g_hand_stepped.fetch_add(1); }
    return NFISAM_OK;
}
extern "C" int nfisam_nsf_train_plan_enqueue(nfisam_train_plan* p) {
    if (p == nullptr || p->exec == nullptr) return NFISAM_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->enqueue_mu);
    const hipError_t e = hipGraphLaunch(p->exec, p->cap);
    if (e != hipSuccess) { nfisam_g_last_hip_error = (int)e; return NFISAM_ERR_LAUNCH; }
    p->ahead = true;                                   // (until `end` has seen the stream drained)
    p->enqueued.fetch_add(1);
    return NFISAM_OK;
}
// Feeder: keeps `depth` chunks enqueued ahead of the last one closed (depth 0 pauses it; the call returns once the feeder
// is outside hipGraphLaunch).  Launching a chunk's graph costs ~0.3 ms of host time per 0.8 ms of device time.
static void feeder_main(nfisam_train_plan* p) {
    (void)hipSetDevice(p->device);
    const struct timespec nap = {0, 20000};
    while (p->feed_quit.load() == 0) {
        const int depth = p->feed_depth.load();
        if (depth > 0 && p->feed_error.load() == 0) {
            const volatile nfisam_train_state* m = p->hst;
            const long closed = m[0].reserved[0];
            // a chunk is worth launching while some slot trains: as of the last closed chunk, or refilled since the last launch
            bool work = p->refills.load() > 0;
            for (int c = 0; c < p->n_cliques && !work; ++c) work = (m[c].stop == 0 && m[c].step < p->cfg.max_iters);
            if (work && p->enqueued.load() - closed < depth) {
                p->feed_busy.store(1);
                if (p->feed_depth.load() > 0) {          // (a pause that arrived meanwhile wins)
                    std::lock_guard<std::mutex> lk(p->enqueue_mu);
                    p->refills.store(0);
                    const hipError_t e = hipGraphLaunch(p->exec, p->cap);
                    if (e != hipSuccess) p->feed_error.store((int)e);
                    else { p->ahead = true; p->enqueued.fetch_add(1); }
                }
                p->feed_busy.store(0);
            }
        }
        nanosleep(&nap, nullptr);                        // (also lets a refill waiting for the lock go first)
    }
}
extern "C" int nfisam_nsf_train_plan_feed(nfisam_train_plan* p, int depth) {
    if (p == nullptr || p->exec == nullptr || depth < 0) return NFISAM_ERR_ARG;
    if (p->feed_error.load() != 0) { nfisam_g_last_hip_error = p->feed_error.load(); return NFISAM_ERR_LAUNCH; }
    p->feed_depth.store(depth);
    if (depth > 0 && !p->feeder.joinable()) p->feeder = std::thread(feeder_main, p);
    if (depth == 0)
        while (p->feed_busy.load() != 0) __builtin_ia32_pause();
    return NFISAM_OK;
}
extern "C" long nfisam_nsf_train_plan_enqueued(const nfisam_train_plan* p) { return p != nullptr ? p->enqueued.load() : 0; }
// Slot c gets a new problem of the same shape: batch and parameters copied from the caller's buffers (ordered behind
// `stream`), moments / workspace / loss record and -- last -- the state zeroed, all on the plan's stream, i.e. between chunks.
extern "C" int nfisam_nsf_train_plan_refill(nfisam_train_plan* p, int c, const float* x, const float* kparams,
                                            nfisam_stream_t stream) {
    if (p == nullptr || p->exec == nullptr || c < 0 || c >= p->n_cliques || x == nullptr || kparams == nullptr) return NFISAM_ERR_ARG;
    const nfisam_clique& q = p->host[(size_t)c];
    std::lock_guard<std::mutex> lk(p->enqueue_mu);
    // (the slot's own event: p->ev belongs to begin / end)
    if (p->slot_ev.size() != (size_t)p->n_cliques) p->slot_ev.assign((size_t)p->n_cliques, nullptr);
    if (p->slot_ev[(size_t)c] == nullptr) HIP_TRY(hipEventCreateWithFlags(&p->slot_ev[(size_t)c], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->slot_ev[(size_t)c], (hipStream_t)stream));
    HIP_TRY(hipStreamWaitEvent(p->cap, p->slot_ev[(size_t)c], 0));
    const size_t pk = (size_t)p->L * kcount(q.D, p->K, p->H) * sizeof(float);
    HIP_TRY(hipMemcpyAsync((void*)q.x, x, (size_t)q.n * q.D * sizeof(float), hipMemcpyDeviceToDevice, p->cap));
    HIP_TRY(hipMemcpyAsync(q.kparams, kparams, pk, hipMemcpyDeviceToDevice, p->cap));
    HIP_TRY(hipMemsetAsync(q.adam_m, 0, pk, p->cap));
    HIP_TRY(hipMemsetAsync(q.adam_v, 0, pk, p->cap));
    HIP_TRY(hipMemsetAsync(q.kgrad, 0, nfisam_nsf_grad_workspace_count(p->max_n, q.D, p->K, p->H, p->L) * sizeof(float), p->cap));
    HIP_TRY(hipMemsetAsync(q.iter_loss, 0, (size_t)(p->cfg.max_iters > 0 ? p->cfg.max_iters : 1) * sizeof(float), p->cap));
    HIP_TRY(hipMemsetAsync(q.state, 0, sizeof(nfisam_train_state), p->cap));
    p->refills.fetch_add(1);
    return NFISAM_OK;
}
extern "C" int nfisam_nsf_train_plan_peek(const nfisam_train_plan* p, nfisam_train_state* out) {
    if (p == nullptr || out == nullptr) return NFISAM_ERR_ARG;
    if (p->feed_error.load() != 0) { nfisam_g_last_hip_error = p->feed_error.load(); return NFISAM_ERR_LAUNCH; }   // the feeder's launch failed
    const volatile nfisam_train_state* m = p->hst;
    for (int c = 0; c < p->n_cliques; ++c) {
        // the sequence word is written last (system-scope release): read it first, the state behind an acquire fence, and
        // again -- a chunk closing in between gives a torn copy that the next look repairs, so it is reported as not closed
        const int s0 = m[c].reserved[0];
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        nfisam_train_state t;
        t.step = m[c].step; t.stop = m[c].stop; t.have_avg = m[c].have_avg; t.loss_avg = m[c].loss_avg; t.domain_err = m[c].domain_err;
        memset(t.reserved, 0, sizeof(t.reserved));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        t.reserved[0] = (m[c].reserved[0] == s0) ? s0 : -1;
        out[c] = t;
    }
    return NFISAM_OK;
}
extern "C" nfisam_stream_t nfisam_nsf_train_plan_stream(nfisam_train_plan* p) {
    return p != nullptr ? (nfisam_stream_t)p->cap : nullptr;
}
extern "C" int nfisam_nsf_train_plan_end(nfisam_train_plan* p, nfisam_stream_t stream) {
    if (p == nullptr || p->exec == nullptr) return NFISAM_ERR_ARG;
    if (p->feeder.joinable()) {                          // pause the feeder (it stays for the next `begin`)
        p->feed_depth.store(0);
        while (p->feed_busy.load() != 0) __builtin_ia32_pause();
    }
    hipStream_t user = (hipStream_t)stream;
    if (p->ev_end == nullptr) HIP_TRY(hipEventCreateWithFlags(&p->ev_end, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->ev_end, p->cap));        // the caller's stream continues behind everything enqueued here
    HIP_TRY(hipStreamWaitEvent(user, p->ev_end, 0));
    if (p->stepping) { p->stepping = false; g_hand_stepped.fetch_sub(1); }
    return NFISAM_OK;
}

extern "C" int nfisam_nsf_train_loop(const nfisam_clique* host_cliques, const nfisam_clique* dev_cliques,
                                     int n_cliques, int K, int H, float B, int L, const nfisam_adam_cfg* cfg,
                                     int use_graph, int32_t* iters_run, nfisam_stream_t stream) {
    H = compiled_H(H);                                        // any hidden_dim <= 16: the next compiled width, zero-padded
    nfisam_train_plan* p = nullptr;
    int rc = nfisam_nsf_train_plan_create(host_cliques, dev_cliques, n_cliques, K, H, B, L, cfg, use_graph, &p);
    if (rc) return rc;
    rc = nfisam_nsf_train_plan_run(p, iters_run, stream);
    nfisam_nsf_train_plan_destroy(p);
    return rc;
}
