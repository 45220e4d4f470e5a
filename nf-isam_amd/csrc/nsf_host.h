// nsf_host.h -- what the translation units of libnfisam_hip.so share on the host side.
//
// The library is built from one COMMON unit (nsf_kernels.hip: the C ABI of include/nfisam_hip.h, training plans, the
// Adam / bookkeeping / normalisation / elementwise-spline kernels, launch-shape decisions) and several KERNEL units
// (nsf_unit.hip compiled once per -DNSF_UNIT=u): the kernels templated on (K bins, H hidden width) and their launchers
// for the (K, H) pairs of that unit (nsf_units.h).  Splitting the instantiations lets `make -j` compile them in
// parallel, which is what makes K = 2..16 x H in {4, 8, 16} affordable.  The common unit reaches a kernel unit through
// the `NsfUnitOps` table that unit exports.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/nfisam_hip.h"
#include "nsf_device.h"
#include "nsf_split.h"
#include "nsf_units.h"

using namespace nsf;

extern thread_local int nfisam_g_last_hip_error;      // defined in the common unit

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) {                         \
            nfisam_g_last_hip_error = (int)e_;          \
            return NFISAM_ERR_LAUNCH;                   \
        }                                               \
    } while (0)

constexpr int XS = 66;            // LDS row stride (floats) of every [feature][particle] tile
// Per-iteration loss sums live behind the gradient slabs in the kgrad workspace: a ring of LOSS_RING
// iterations x LOSS_SLOTS words (spread so that hundreds of waves do not serialise on one address).  The
// bookkeeping kernel that closes a chunk of iterations consumes and clears them.
constexpr int LOSS_RING = 128, LOSS_SLOTS = 128;   // (round 6: 128 slots -- a lone clique of 240 blocks, two per slot, still leaves an order-free sum)
constexpr int ONES_ROW = 68;       // nsf_train1_kernel: LDS words of 1.0 read as the bias column of the gradient GEMM operands
// fused Adam (nsf_cond_mfma.h): 64 reserved words behind the loss ring, then the second set of gradient copies and the
// second state buffer; the workspace is sized for cliques of up to this many 64-particle tiles
constexpr int FUSED_COUNTERS = 64, FUSED_MAX_COPIES = 64;     // (tiles of 64 particles: up to sixteen four-wave blocks per (clique, dim) group -- n <= 4096)
constexpr int PERSIST_MAX_COPIES = 16;                           // gradient copies (= blocks) per group the chunk-persistent form exchanges (round 5; 8 before)

struct TrainArgs {
    const nfisam_clique* cliques;   // device array (batched) or nullptr
    const nfisam_clique* host_cliques;   // host copy of `cliques` when the caller has one (training plans), else nullptr: launcher-side only
    nfisam_clique single;           // by-value descriptor when cliques == nullptr
    const float* gz;                // VJP mode: upstream dL/dz [n,D]
    const float* gl;                // VJP mode: upstream dL/dlogdet [n] (nullable => 0)
    float* gx;                      // optional dL/dx [n,D]
    float* loss_sum;                // optional accumulator when no train state is attached
    float B;
    int L;
    int max_iters;
    int nll_mode;
    int layer_stride;               // floats between layers in kparams/kgrad (0: kparam_count(D))
    int wl_floats;                  // LDS floats reserved for the parameter copy (WL variants)
    int slab;                       // != 0: kgrad is [n_tiles][L*Pk], plain stores (see gsink)
    int tile;                       // host only: particles per tile (kernel family)
    int iter_idx;                   // iteration index inside the chunk (state->step advances once per chunk)
    int g_tiles;                    // 1: LDS holds the two dL/dx tiles (L > 1 or gx requested)
    int tiles_per_block;            // wide kernels, L == 1: 64-particle tiles summed into one gradient copy (per block / per wave)
    int xrows;                      // nsf_train1_kernel: rows of a wave's particle tile in LDS (largest D of the launch)
    int n_copies;                   // nsf_train1_kernel: gradient copies per clique workspace (the loss ring sits behind them)
    int waves;                      // nsf_train1_kernel: waves per block (1, 2, 4 or 8)
    int grid_cliques;               // nsf_train1_kernel: cliques of the launch
    int groups;                     // nsf_train1_kernel: (clique, dim) groups = cliques x largest D (grid = 8 x blocks per group x groups / 8)
    unsigned magic_cliques;         // nsf_train1_kernel: ceil(2^32 / cliques) (0: one clique): group / cliques without a division
    int t_shift, w_shift;           // nsf_train1_kernel: log2 of tiles per wave / waves per block
    int pair_dims;                  // nsf_train2_kernel: waves that share a SIMD take the cheapest dims (see the kernel)
    int chain, n_chains;            // nsf_train1_kernel: this launch covers the (clique, dim) groups g with (g / 8) % n_chains == chain
    const uint32_t* panel_map;      // nsf_train1_kernel: kernel-layout parameter index -> LDS word(s) of the conditioner panel (nsf_cond_mfma.h)
    int pair_ws;                    // nsf_train3_kernel: kgrad is a clique WORKSPACE (nfisam_nsf_grad_workspace_count): the forward state may be parked behind the panel image
    int pair_stash;                 // launcher-side: park it (the launch is in the workspace's size bound)
    int pair_image;                 // nsf_train3_kernel: the clique's panel image is current (written by the previous iteration's Adam kernel)
    int fused_adam;                 // nsf_train1_kernel: apply the previous iteration's Adam update on the way into LDS (nsf_cond_mfma.h)
    int persist_iters;              // nsf_train1_kernel, chunk-persistent form: iterations 0 .. persist_iters - 1 of the chunk in ONE launch (0: one iteration per launch)
    int persist_split;              // ... != 0: the Adam update of a dim is divided among the group's blocks (contended launches; nsf_cond_mfma.h)
    int persist_spins;              // ... log2 of the polls a block waits at its group barrier before it raises the group's abort flag (15; NFISAM_PERSIST_SPINS)
    nfisam_adam_cfg adam;
    float log_b1, log_b2;
    int half;                       // nsf_train1_kernel: two lanes per particle (nsf_half.h): a wave covers 32 particles, a block 32 x waves
    // round 6, window-spanning persistent launch (nsf_unit.hip, nsf_bookkeep.h): span_window > 0 -- the launch runs up to
    // persist_iters iterations and closes every window of span_window iterations ITSELF (the clique's block that arrives last at
    // the window's end runs the bookkeeping: loss record, stop rule, step, mirror), all of the clique's blocks leave together
    // when the rule fires or the budget ends; the closing Adam kernel behind the launch applies the last iteration's update
    int span_window;
    nfisam_train_state* span_mirror;   // the plan's host-pinned mirror (indexed by clique) or nullptr
};
// control words of a clique's workspace (behind the loss ring: FUSED_COUNTERS words, one per dim) that the window-spanning launch
// uses for itself; cliques of more than SPAN_MAX_D dims keep to one launch per chunk
constexpr int SPAN_WORD_TICKET = 63, SPAN_WORD_DECISION = 62, SPAN_WORD_LAST_T = 61, SPAN_WORD_LAST_PARITY = 60, SPAN_MAX_D = 56;
// ... and two words a chunk-persistent launch leaves for the kernel that closes its chunk (nsf_adam_kernel with `fused_close`, nsf_kernels.hip): the clique's
// step and stop AS THE LAUNCH FOUND THEM -- the closing Adam blocks read these while the bookkeeping block of the same kernel advances
// state->step / stop (cliques of up to SPAN_MAX_D dims; wider ones close a chunk with two kernels)
constexpr int CLOSE_WORD_STEP = 58, CLOSE_WORD_STOP = 57;


// Launchers of one (K, H) pair, exported by the kernel unit that instantiates it.
struct NsfUnitOps {
    int K, H;
    int (*forward)(const float* x, const float* kparams, int n, int D, float B, int L, int layer_stride, float* z, float* logdet,
                   float* logprob, hipStream_t s);
    int (*inverse)(const float* z, const float* x_sep, const float* kparams, int n, int D, int Ds, float B, int L,
                   int layer_stride, const float* mean, const float* stdv, const uint8_t* circular, float* x_out,
                   float* logdet, hipStream_t s);
    int (*walk)(const nfisam_post_clique* table, int n_cliques, const int32_t* cols, const float* obs, int max_D, float B,
                int L, int n, const float* Zt, float* St, hipStream_t s);
    int (*train)(const TrainArgs& a, int n_cliques, int max_n, int max_D, hipStream_t s);   // gradient kernel of an iteration
    int (*prepare)(int max_D);      // device-resident tables of the training kernels (idempotent; call once outside stream capture)
    // nsf_train3_kernel (two dims per wave): LDS bytes of a launch (0: the launch does not fit that kernel) and the
    // device-resident panel map + the offsets of each clique width's map in it (nsf_cond_mfma.h: build_pair_map)
    size_t (*pair_lds)(int L, int max_D);
    int (*pair_map)(const uint32_t** map, uint32_t* offsets /* [PAIR_MAP_OFFSETS] */);
    // blocks of the chunk-persistent dim-major kernel (4 waves, tiles of max_D rows) the CURRENT DEVICE holds at once:
    // hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units (0: no such kernel / the query failed)
    long (*persist_places)(int max_D);
};
constexpr int PAIR_MAP_OFFSETS = 17;
// the clique workspace reserves room for the parked forward state of multi-layer launches up to this size (the
// two-dims-per-wave kernel is chosen for latency-bound launches: train_tile)
static inline bool pair_stash_fits(int n, int D) { return (long)((n + TILE2 - 1) / TILE2) * (long)D <= 1280; }
#define NSF_DECLARE_UNIT(u) extern "C" const NsfUnitOps* nsf_unit_ops_u##u(int K, int H);
NSF_UNITS(NSF_DECLARE_UNIT)

static inline size_t kcount(int D, int K, int H) {
    const size_t PoP = (size_t)pop_of(K);
    const size_t kfixed = (size_t)H + (size_t)H * H + H + (size_t)H * PoP + PoP;
    return PoP + (size_t)(D - 1) * kfixed + (size_t)H * ((size_t)(D - 1) * D / 2);
}

static inline int pick_waves(int D) { return D < 1 ? 1 : (D > 8 ? 8 : D); }

template <typename KernelT>
static int set_lds(KernelT kernel, size_t bytes) {
    if (bytes > 160 * 1024) return NFISAM_ERR_ARG;
    if (bytes > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    }
    return NFISAM_OK;
}

// ---- launch-shape knobs (environment variables are for A/B measurements and tests) -------------------------------
// The weight-gradient GEMMs run on the matrix cores when H == 8 (ga2 | ga1 share one 16-row MFMA operand tile); other
// hidden widths, and NFISAM_GRAD=butterfly, take the cross-lane butterfly reduction of round 1.
static inline bool use_mfma_grad(int H) {
    const char* e = getenv("NFISAM_GRAD");
    return H == 8 && !(e != nullptr && strcmp(e, "butterfly") == 0);
}

static inline int weights_mode() {   // 0: scalar-cache path, 1: LDS copy, -1: automatic
    const char* e = getenv("NFISAM_WEIGHTS");
    return (e == nullptr) ? -1 : (strcmp(e, "lds") == 0 ? 1 : (strcmp(e, "scalar") == 0 ? 0 : -1));
}

// Training kernel family, chosen per launch: "split" (two lanes per particle, 32-particle tiles) while the
// launch leaves SIMDs idle -- it halves the instructions per wave and doubles the waves -- and "wide" (one
// lane per particle, 64-particle tiles: no duplicated scalar spline work, half the gradient copies) once the
// chip is full.  NFISAM_TRAIN=wide|split forces one family (A/B measurements).  The split kernel exists for H == 8.
// `nll_L1`: a training iteration of a one-layer flow (no upstream gradient): the dim-major kernel takes it at every size
// (with the Adam update fused into the next launch it beats the two-lanes-per-particle kernel from D = 2, n = 500 up,
// scripts/exp/regime_grid.sh); VJP launches and L > 1 keep the split kernel while the launch is small.
static inline bool dim_major_enabled();
// `pair_h4`: hidden_dim 4 has ONE small-launch kernel, the two-dims-per-wave one (no two-lanes-per-particle kernel for
// it): the caller says whether this launch can take it (layers or dL/dx couple the dims, the panels fit: pair_lds > 0).
// (round 5: hidden_dim 16 likewise -- its multi-layer / dL/dx launches in the latency regime take the two-dims-per-wave kernel
//  with ONE layer's panels resident; `pair_h4` = "a hidden width other than 8 whose pair kernel takes this launch")
static inline int train_tile(int n_cliques, int max_n, int max_D, int H, bool nll_L1 = false, bool pair_h4 = false) {
    if (H != 8 && !((H == 4 || H == 16) && pair_h4 && !nll_L1)) return TILE;
    const char* e = getenv("NFISAM_TRAIN");            // read per call: tests switch families in-process
    if (e != nullptr) {
        if (strcmp(e, "wide") == 0) return TILE;
        if (strcmp(e, "split") == 0) return TILE2;
    }
    if (nll_L1 && dim_major_enabled() && max_D <= 96) return TILE;
    const long waves = (long)((max_n + TILE2 - 1) / TILE2) * (long)max_D * (long)n_cliques;
    return waves <= 1280 ? TILE2 : TILE;
}

// Throughput launches (wide family, L == 1) go to nsf_train1_kernel (dim-major blocks) unless NFISAM_DIM_MAJOR=0
// (A/B against nsf_train_kernel's tile-major blocks).
static inline bool dim_major_enabled() {
    const char* e = getenv("NFISAM_DIM_MAJOR");
    return !(e != nullptr && e[0] == '0');
}
// LDS rows (of XS floats) of one wave of the dim-major kernel: particle tile [max_D] + staging [16] + h1 [H] + (D > 16) the
// rows 16.. of dW0 summed over the wave's tiles
__host__ __device__ static inline int train1_wave_rows(int max_D, int H) {
    return max_D + 16 + H + (max_D > 16 ? ((max_D - 16) * H + XS - 1) / XS : 0);
}
// LDS floats of one wave (its rows, rounded so that every wave's base stays 16-byte aligned: 16-byte fragment stores)
__host__ __device__ static inline int train1_wave_floats(int max_D, int H) { return (train1_wave_rows(max_D, H) * XS + 3) & ~3; }
#if defined(NSF_STAMPS) && NSF_STAMPS == 3
constexpr int PANEL_BASE = 68;     // + 4 waves x 16 words of per-phase cycle sums (diagnostic build)
#else
constexpr int PANEL_BASE = 4;      // LDS words in front of the conditioner panel: word 0 takes the stores of parameters that have one destination only
#endif
// Waves per block of the dim-major kernel (they share the (clique, dim): one weight panel, one gradient copy): 4.
// NFISAM_BIG_W = 1..8 for experiments.  Eight (half the gradient copies for the fused Adam update to read back, half the
// staging work per thread) measured 13 % SLOWER on a single Plaza clique: two waves per SIMD on 60 CUs instead of one
// wave per SIMD on 120 -- in the latency regime a wave wants its SIMD to itself.
static inline int dim_major_waves(int n_cliques, int max_n, int max_D, int T) {
    (void)n_cliques; (void)max_n; (void)max_D; (void)T;
    const char* e = getenv("NFISAM_BIG_W");
    const int v = e != nullptr ? atoi(e) : 4;
    return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 4;
}
// ---- two lanes per particle on the dim-major kernel (round 6, nsf_half.h) --------------------------------------------------
// The family for launches that leave most SIMDs idle -- ONE clique, i.e. every fit of a real NF-iSAM run: a wave covers 32
// particles with a shorter per-lane program (stamped arithmetic of a unit 9.3 k -> 7.1 k cycles), twice the waves, blocks of
// 128 particles.  Instantiated for hidden_dim 8 and sixteen theta columns per half (num_knots 9 .. 11: every shipped large run
// uses 9), cliques of up to 16 dims (one operand tile of [x | 1] in the dW0 GEMM).  A launch takes it when its blocks get a CU
// each (<= 256) and a (clique, dim) group keeps to EIGHT blocks (n <= 1024): the exchange between a group's blocks is bound
// by the instructions a thread spends per gradient copy, and with sixteen copies it gives back what the shorter unit gained
// (one clique, D = 15, us per iteration, 64-particle family / this one: n = 500 .. 1024: 7.4 / 6.25; n = 1500 .. 2048: 7.40 / 7.34).
// NFISAM_HALF=0: never; =2: groups of up to sixteen blocks too (measurements); NFISAM_HALF_W=4|8: waves per block (8: 256
// particles per block, two waves per SIMD -- measured slower: the younger wave of a SIMD runs at half speed).
static inline int half_waves() {
    const char* e = getenv("NFISAM_HALF_W");
    const int v = e != nullptr ? atoi(e) : 4;
    return (v == 4 || v == 8) ? v : 4;
}
// compute units of the current device, asked once per device and translation unit (first call OUTSIDE a stream capture:
// unit_prepare / plan creation); 0: unknown
static inline int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int n = 0;
        v = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : -1;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v > 0 ? v : 0;
}
static inline bool half_shape(int n_cliques, int max_n, int max_D, int K, int H, int L, int T) {
    const char* e = getenv("NFISAM_HALF");               // read per call: tests switch families in-process
    if (e != nullptr && e[0] == '0') return false;
    if (H != 8 || K < 2 || hp_of(K) != 16 || L != 1 || T != 1 || max_D > 16 || max_D < 1) return false;
    const int per_block = 32 * half_waves();
    const long copies = (max_n + per_block - 1) / per_block;
    const long blocks = (long)n_cliques * max_D * copies;
    // "2": also nine to sixteen copies per group (1025 .. 2048 particles).  Measured with helper waves (nsf_unit.hip: unit_train1 --
    // at most 240 blocks, a CU each): one Plaza clique 6.71 us per iteration against 6.94 for the 64-particle family, +3 % of
    // Plaza1's training rate -- but 240 whole-CU blocks fill seven of the eight XCDs to the last CU, so it is not the default
    // (DESIGN.md 3.1h).  Its loss record is order-free like every other launch's (LOSS_SLOTS = 128: two blocks per slot).
    if (e != nullptr && e[0] == '2') return copies <= PERSIST_MAX_COPIES && blocks <= 256;
    return copies <= 8 && blocks <= 256;
}
// hidden widths the dim-major kernel is instantiated for (H <= 8: [ga2 | ga1] share one 16-row MFMA operand tile;
// H = 16: one tile each and separate bias chains)
static inline bool dim_major_hidden(int H) {
    const char* e = getenv("NFISAM_GRAD");
    if (e != nullptr && strcmp(e, "butterfly") == 0) return false;
    return H == 8 || H == 4 || H == 16;
}
// smallest launch ((tile, dim) units) that goes to the dim-major kernel; NFISAM_DIM_MAJOR_MIN overrides (experiments)
static inline long dim_major_min_units() {
    const char* e = getenv("NFISAM_DIM_MAJOR_MIN");
    return e != nullptr ? atol(e) : 0;
}
static inline bool is_dim_major(int n_cliques, int max_n, int max_D, int L, int tile, int H) {
    const long tiles = (long)((max_n + TILE - 1) / TILE) * n_cliques;
    // four waves' LDS rows + the weight panel must fit next to each other (160 KB per CU): D <= 96; wider cliques take
    // the tile-major kernels
    // (H = 16: D <= 80 -- the panel and the h1 rows are twice as big)
    return tile == TILE && L == 1 && dim_major_hidden(H) && dim_major_enabled() && max_D <= (H == 16 ? 80 : 96) &&
           tiles * max_D > dim_major_min_units();
}

// 64-particle tiles summed into one gradient copy (one wave's sweep in nsf_train1_kernel, one block's in
// nsf_train_kernel).  More tiles per copy = fewer copies for the Adam kernel to read back, fewer prologues and
// better scalar-cache reuse of the weights, as long as the launch still has ~1.5x the waves the chip holds at four
// waves per SIMD (4096).  The decision depends on the launch shape only, so the gradient, Adam and bookkeeping
// launches of an iteration agree on the number of gradient copies.
static inline int tiles_per_block(int n_cliques, int max_n, int max_D, int L, int tile, int H) {
    if (tile != TILE || L != 1 || !(use_mfma_grad(H) || dim_major_hidden(H))) return 1;
    const long tiles_c = (max_n + TILE - 1) / TILE, tiles = tiles_c * n_cliques;
    if (tiles * max_D <= dim_major_min_units()) return 1;
    const char* e = getenv("NFISAM_TILES_PER_BLOCK");
    int T = 1;
    if (is_dim_major(n_cliques, max_n, max_D, L, tile, H)) {
        if (e != nullptr && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4 || atoi(e) == 8)) return atoi(e);
        // One tile per wave while the launch is at most ~2 rounds of resident waves (3 per SIMD x 1024 SIMDs); beyond that
        // waves take two tiles (and share one prologue: tile fetch, weight panel, pending Adam update), then four.
        // Measured on 8..64 cliques of n = 2000, D = 15 with the iteration split over two parallel graph branches
        // (scripts/time_grad.py, NFISAM_TILES_PER_BLOCK=1|2|4, us per iteration):
        //   cliques   8      12     16     24     32     48     64
        //   T = 1    16.8   21.5   28.5   41.9   55.0   82.2  109.4
        //   T = 2    17.3   22.1   25.9   35.5   46.7   69.4   92.5
        //   T = 4    25.9   27.3   29.1   36.7   44.6   61.9   81.6
        // i.e. double T while the launch would exceed ~6k waves (8 tiles per wave only beyond 12k waves at T = 4).
        const auto waves = [&](int t) { return (long)n_cliques * ((tiles_c + t - 1) / t) * max_D; };
        while (T < 8 && waves(T) > (T < 4 ? 6144 : 12288)) T *= 2;
        return T;
    }
    if (!use_mfma_grad(H)) return 1;             // nsf_train_kernel sweeps several tiles per block on its MFMA gradient path only
    if (tiles <= 256) return 1;                  // nsf_train_kernel spreads the dims over grid.z there: one tile per block
    if (e != nullptr && atoi(e) >= 1 && atoi(e) <= 8) return atoi(e);
    const int W = max_D < 4 ? max_D : 4;
    while (T < 4 && tiles * W / (2 * T) >= 4096) T *= 2;
    return T;
}
