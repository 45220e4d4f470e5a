// nsf_bookkeep.h -- closing a chunk (or, round 6, a WINDOW inside a window-spanning persistent launch) of training iterations:
// the ring's loss sums -> iter_loss entries (reference: src/slam/NFiSAM.py:473), the window early-stop rule (NFiSAM.py:481-491),
// state->step / stop, the host-pinned mirror.  ONE copy of the arithmetic for `nsf_bookkeep_kernel` (nsf_kernels.hip: one block per
// clique between two chunks) and for the block of `nsf_train1_kernel` that arrives last at a window's end (nsf_unit.hip): the same
// sums in the same order, so the loss record and the stop decision do not depend on which of the two closed the window.
// The block's first 256 threads (four waves) do the work; it contains workgroup barriers, which EVERY thread of the block passes (a
// block of the training kernel may have eight waves: its helper waves call this too and only take part in the barriers -- the sums are
// partitioned over 256 threads whoever calls, so the bits do not depend on the caller's block size).
#pragma once
#include "nsf_host.h"

struct BookArgs {
    float* ring;                    // the clique's loss ring [LOSS_RING][LOSS_SLOTS] (+ FUSED_COUNTERS control words behind it)
    float* iter_loss;
    nfisam_train_state* st;
    nfisam_train_state* mirror;     // this clique's entry of the plan's host-pinned mirror, or nullptr
    int n, D;
    int chunk;                      // iterations to close
    nfisam_adam_cfg cfg;
    int zero_counters;              // != 0: the per-dim control words are zeroed (between chunks: nobody else is running); 2: all but
                                    // CLOSE_WORD_STEP / _STOP, which the Adam blocks of the same kernel are reading (nsf_adam_kernel with `fused_close`)
};

__device__ __forceinline__ void bookkeep_body(const BookArgs& a) {
    float* ring = a.ring;
    float* iter_loss = a.iter_loss;
    nfisam_train_state* st = a.st;
    const int n = a.n, D = a.D;
    const int lane = threadIdx.x & 63;
    const bool worker = threadIdx.x < 256;                           // (block-size independent: see the header)
    const int w = worker ? (int)(threadIdx.x >> 6) : 0;
    constexpr int NTW = 256;                                         // threads that share the sums
    constexpr int PER_WAVE = LOSS_RING / 4;
    float part[PER_WAVE];
    static_assert(LOSS_SLOTS == 128, "a lane takes slots l and l + 64 of a row");
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k)                                                       // ring row w + 4k
        part[k] = worker ? ring[(w + 4 * k) * LOSS_SLOTS + lane] + ring[(w + 4 * k) * LOSS_SLOTS + 64 + lane] : 0.0f;   // (launches of up to 128 blocks use the lower half only: x + 0)
    // the group-barrier counters of the chunk-persistent training kernel (one per dim; nsf_unit.hip: bits 0-22 arrivals,
    // 23-30 the XCC ids the group's blocks ran on, 31 the group's abort flag): looked at, then zeroed for the next chunk
    int stalled = 0, xcd_span = 0;
    if (threadIdx.x < FUSED_COUNTERS) {                              // (= the block's first wave)
        unsigned* ctr = (unsigned*)(ring + (size_t)LOSS_RING * LOSS_SLOTS) + threadIdx.x;
        const bool close_word = a.zero_counters == 2 && (threadIdx.x == CLOSE_WORD_STEP || threadIdx.x == CLOSE_WORD_STOP);
        const unsigned cv = close_word ? 0u : *ctr;
        if (a.zero_counters && !close_word) *ctr = 0u;
        stalled = __any((int)(cv >> 31));
        int span = __popc((cv >> 23) & 0xffu);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(span, off, 64); span = o > span ? o : span; }
        xcd_span = span;
    }
    const int s0 = st->step, stop0 = st->stop, have_avg = st->have_avg;
    const float loss_avg = st->loss_avg;
    int new_step = s0, new_stop = stop0, new_have = have_avg, new_err = st->domain_err;
    const int slower_stop = st->reserved[2];                         // hold-out validation: the scheduled end (nsf_validate_kernel), 0: none
    float new_avg = loss_avg;
    __shared__ float s_loss[LOSS_RING];
    __shared__ float s_wsum[4];
    __shared__ int s_bad[4];
    const bool active = (stop0 == 0 && s0 < a.cfg.max_iters);        // block-uniform
    if (active) {
        const int cnt = (a.chunk < a.cfg.max_iters - s0) ? a.chunk : (a.cfg.max_iters - s0);
        const float inv_n = 1.0f / (float)n;
        const int wnd = a.cfg.average_window;
        // ring row r holds iteration (r - s0) mod 128 of this chunk (if that is < cnt)
        const float rowsum = butterfly<PER_WAVE>(part, lane);        // lane l: total of row w + 4 (l & 31)
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int it = ((w + 4 * k) - s0) & (LOSS_RING - 1);
            if (it < cnt && worker) { ring[(w + 4 * k) * LOSS_SLOTS + lane] = 0.0f; ring[(w + 4 * k) * LOSS_SLOTS + 64 + lane] = 0.0f; }   // wave-uniform
        }
        if (lane < PER_WAVE && worker) {
            const int it = ((w + 4 * lane) - s0) & (LOSS_RING - 1);
            if (it < cnt) {
                const float loss = rowsum * inv_n + 0.5f * (float)D * 1.8378770664093453f;   // log(2 pi)
                s_loss[it] = loss;
                iter_loss[s0 + it] = loss;
            }
        }
        __syncthreads();
        // a non-finite loss ends the run at its iteration (the rule below can only fire on the chunk's last one)
        int bad_at = cnt;
        for (int it = threadIdx.x; it < cnt && worker; it += NTW) {
            const float l = s_loss[it];
            if (!(l == l) || fabsf(l) > 3.0e38f) bad_at = (it < bad_at) ? it : bad_at;
        }
        bad_at = __reduce_min_sync(~0ull, bad_at);                 // per wave
        if (lane == 0 && worker) s_bad[w] = bad_at;
        __syncthreads();
        bad_at = min(min(s_bad[0], s_bad[1]), min(s_bad[2], s_bad[3]));
        if (bad_at < cnt) {
            new_err |= 1; new_stop = 1; new_step = s0 + bad_at + 1;
        } else {
            const int t_end = s0 + cnt;
            if (wnd > 0 && (t_end % wnd) == 0) {   // window mean over iter_loss[t_end - wnd, t_end): this chunk's part from LDS
                float sm = 0.0f;
                for (int j = t_end - wnd + (int)threadIdx.x; j < t_end && worker; j += NTW)
                    sm += (j >= s0) ? s_loss[j - s0] : iter_loss[j];
                sm = wave_sum(sm);
                if (lane == 0 && worker) s_wsum[w] = sm;
                __syncthreads();
                const float nw = (s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3]) / (float)wnd;
                if (have_avg != 0 && loss_avg != 0.0f) {
                    const float delta = fabsf(1.0f - nw / loss_avg);
                    if (delta < a.cfg.loss_delta_tol) new_stop = 1;
                }
                new_avg = nw;
                new_have = 1;
            }
            new_step = t_end;
        }
        // (reference NFiSAM.py:453-456: the loop breaks in front of iteration i when i + 1 >= slower_stop_iter)
        if (slower_stop != 0 && new_step + 1 >= slower_stop) new_stop = 1;
    }
    if (threadIdx.x == 0) {
        if (stalled) {                                               // a group barrier of the chunk timed out: the run is over, loudly
            new_err |= NFISAM_STATE_STALLED; new_stop = 1;
            st->domain_err = new_err; st->stop = new_stop;
        }
        if (xcd_span > st->reserved[1]) st->reserved[1] = xcd_span;   // most XCDs a (clique, dim) group of a persistent chunk spanned
        if (active) {
            st->loss_avg = new_avg; st->have_avg = new_have; st->domain_err = new_err; st->stop = new_stop; st->step = new_step;
        }
        if (a.mirror != nullptr) {
            nfisam_train_state* m = a.mirror;
            const int seq = __hip_atomic_load(&m->reserved[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
            m->step = new_step; m->stop = new_stop; m->have_avg = new_have; m->loss_avg = new_avg; m->domain_err = new_err;
            m->reserved[1] = st->reserved[1];
            m->reserved[2] = st->reserved[2];
            __hip_atomic_store(&m->reserved[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
