/*
 * nfisam_hip.h — C ABI of the MI355X (gfx950) implementation of NF-iSAM's per-clique
 * normalizing-flow hot path (autoregressive rational-quadratic neural spline flow).
 *
 * The reference (MarineRoboticsGroup/NF-iSAM) has no FFI for this path: its boundary is the
 * Python module surface `flows.*` + the solver hooks of `slam.NFiSAM`.  This
 * header is what a binding for that surface links against; `nf-isam_amd/nfisam_hip/` is the
 * ctypes binding and INTEGRATION.md shows the stub a reference maintainer would add.  Each
 * entry point cites the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - All `float*`/`uint8_t*` data arguments are DEVICE pointers (HBM), contiguous, row-major,
 *    borrowed for the duration of the enqueued work.  `int32_t* map` of the layout helper and
 *    everything documented "host" are host pointers.
 *  - `stream` is a `hipStream_t` passed as `void*` (0 = the null stream).  Work is enqueued
 *    on it; no entry point synchronises with the host unless its comment says so.
 *  - Return value: NFISAM_OK or an error code below; on NFISAM_ERR_LAUNCH the HIP error is
 *    left readable through `nfisam_last_hip_error()`.
 *  - No global mutable state except the last-error word; re-entrant across streams/devices.
 *  - There is NO CPU fallback anywhere behind this ABI.
 *
 * Symbols: n particles, D clique dimension (columns = [obs | separator | frontal]),
 * Ds leading columns that are given in conditional sampling, K spline bins (`num_knots`),
 * H hidden width, Po = 3K-1, B tail bound (5.0 in the reference, flows.py:51), L flow layers.
 *
 * Parameter storage ("kernel layout", float32, one block of `nfisam_nsf_kparam_count`
 * floats per flow layer, layers concatenated):
 *     init_param[PoP]
 *     for i = 1..D-1:  W0t[i][H]  b0[H]  W1t[H][H]  b1[H]  W2t[H][PoP]  b2[PoP]
 * where W*t are the TRANSPOSES ([in][out]) of the reference's nn.Linear weights
 * (flows.py:31-37), so that every weight row a wavefront consumes is contiguous (16-byte
 * aligned rows: scalar loads or ds_read_b128).  The PoP = 2*HP output columns are two halves,
 *     [ K width logits  | first floor(K/2) derivative logits | 0-pad to HP ]
 *     [ K height logits | remaining derivative logits        | 0-pad to HP ],  HP = 4*ceil((K + floor(K/2))/4)
 * (the reference's order is widths | heights | derivatives, flows.py:84-88): the two lanes that share
 * a particle in the training kernel each own one half.  Padding entries are zero and stay zero under training.
 * `nfisam_nsf_layout_map` gives the permutation to/from the reference's parameter order
 * (init_param, layers.{i-1}.network.{0,2,4}.{weight,bias}; flows.py:57-59).
 */
#ifndef NFISAM_HIP_H
#define NFISAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFISAM_OK 0
#define NFISAM_ERR_ARG 1          /* bad shape / NULL pointer / unsupported (K,H)  (ValueError) */
#define NFISAM_ERR_LAUNCH 2       /* HIP launch or runtime failure                              */
#define NFISAM_ERR_DOMAIN 3       /* numerical domain error flagged by a kernel (utils.py:74-76,133) */
#define NFISAM_ERR_NO_DEVICE 4    /* no usable gfx950 device                                    */
#define NFISAM_ERR_STALL 5        /* a chunk-persistent training launch gave up waiting for a block of its own that never
                                   * became resident (another process or launch held its place): the plan's parameters,
                                   * moments and loss record are those of an INCOMPLETE chunk -- not a numerical failure.
                                   * The library takes the one-launch-per-iteration graph for every later run of the
                                   * process; the caller re-runs the fit from its initial state (ABI 1400)          */

typedef void* nfisam_stream_t;

/* ABI version of this header (major*1000 + minor). */
int nfisam_abi_version(void);
/* Last HIP error code recorded by a failing call on this thread (0 if none). */
int nfisam_last_hip_error(void);
/* 1 if flows with (num_knots K, hidden_dim H) run on the kernels, else 0: K in 2..16, H in 1..16.  The kernels are
 * instantiated for H in {4, 8, 16}; every other width runs as the next compiled width with zero-padded hidden units (the
 * reference takes any width, src/flows/flows.py:26-41): a padded unit has zero in-weights, bias and out-weights, so its
 * activation is tanh(0) = 0 and every gradient touching it is an exact 0 -- Adam leaves the padding at zero (ABI 1500). */
int nfisam_nsf_supported(int K, int H);

/* ---- layout (host-only helpers, no GPU needed) ---------------------------------------- */
/* Parameters per layer in the reference's order = numel of NSF_AR.parameters() (flows.py:51-60). */
size_t nfisam_nsf_param_count(int D, int K, int H);
/* Floats per layer in kernel layout (>= param_count because of padding; for a hidden_dim that is not a compiled width:
 * the count of the next compiled width). */
size_t nfisam_nsf_kparam_count(int D, int K, int H);
/* host: map[kidx] = index into the reference-order blob, or -1 for padding. map has kparam_count entries. */
int nfisam_nsf_layout_map(int D, int K, int H, int32_t* map);

/* ---- inference ------------------------------------------------------------------------ */
/* NSF_AR.forward chained over L layers + prior log-prob (flows.py:65-93, models.py:11-24),
 * in the CORRECT layout (the reference returns a scrambled one).
 *   x[n,D] -> z[n,D] (nullable), logdet[n] (nullable), logprob[n] (nullable)
 *   logprob = -0.5|z|^2 - D/2 log(2 pi) + logdet
 * layer_stride: floats between consecutive layers' parameter blocks; 0 = kparam_count(D).  Because
 * the flow is autoregressive and dim i's block only depends on i, the first D' < D_model dims of a
 * trained D_model-dimensional flow ARE its D'-dimensional marginal flow: pass D = D' and
 * layer_stride = kparam_count(D_model) to evaluate it in place
 * (NormalizingFlowModelWithSeparator.separator_forward, slam/NFiSAM.py:157-173).  The same argument
 * exists on nfisam_nsf_inverse and nfisam_nsf_backward.                                     */
int nfisam_nsf_forward(const float* x, const float* kparams, int n, int D, int K, int H, float B,
                       int L, size_t layer_stride, float* z, float* logdet, float* logprob,
                       nfisam_stream_t stream);

/* NSF_AR.inverse / inverse_given_separator (flows.py:95-137) fused with the adapter's
 * normalisation of the given columns and un-normalisation + angle wrap of the result
 * (NormalizingFlowModelWithSeparator.{normalize_samples,inverse_given_separator,
 *  unnormalize_samples}, slam/NFiSAM.py:96-118,140-155).
 *   z[n,D-Ds] latent draws; x_sep[n,Ds] RAW (un-normalised) given columns, NULL iff Ds==0
 *   mean[D], std[D], circular[D] (1 = angle column): NULL mean => data already normalised and
 *   the output is left normalised (plain NSF_AR.inverse_given_separator).
 *   x_out[n,D-Ds]; logdet[n] nullable (sum of -log|dz/dx| over the solved columns, as NSF_AR.inverse)
 * L > 1 with given columns: layer l is conditioned on the given columns pushed through the marginal flow of
 * layers 0..l-1, so that forward(cat(x_sep, x_out)) returns z.  (The reference's literal loop feeds every layer
 * the raw columns, slam/NFiSAM.py:151-152, which inverts no composition; with flow_number = 1, its default, the
 * two coincide.)                                                                                              */
int nfisam_nsf_inverse(const float* z, const float* x_sep, const float* kparams, int n, int D, int Ds,
                       int K, int H, float B, int L, size_t layer_stride, const float* mean, const float* std,
                       const uint8_t* circular, float* x_out, float* logdet, nfisam_stream_t stream);

/* Elementwise rational-quadratic spline with per-element logits (flows.utils.unconstrained_RQS and
 * RQS, src/flows/utils.py:25-164).  inputs[M], widths[M,K], heights[M,K];
 * padded_derivatives=1: derivs[M,K-1], end slopes fixed to 1 and identity outside the box
 *                       (unconstrained_RQS with left=bottom=-tail_bound, right=top=tail_bound);
 * padded_derivatives=0: derivs[M,K+1] (bounded RQS; callers validate the domain, utils.py:74-76).
 * out[M], logabsdet[M].  Any K >= 1 with 1e-3*K <= 1 (utils.py:80-83).                      */
int nfisam_rqs(const float* inputs, const float* widths, const float* heights, const float* derivs, int M, int K,
               int inverse, float left, float right, float bottom, float top, int padded_derivatives,
               float* out, float* logabsdet, nfisam_stream_t stream);

/* ---- posterior traversal ------------------------------------------------------------------ */
/* One clique of the tree walk, parents before children.  Column indices refer to the sample matrix. */
typedef struct nfisam_post_clique {
    const float* kparams;        /* the clique's trained flow, kernel layout, D_model dims, L layers   */
    const float* mean;           /* [D_model] normalisation constants (slam/NFiSAM.py:515-548)        */
    const float* std;
    const uint8_t* circular;     /* [D_model] 1 = angle column                                        */
    int32_t D_model;             /* n_obs + all clique variable dims                                  */
    int32_t n_obs;               /* leading true-observation columns                                  */
    int32_t n_sep;               /* separator columns (already sampled by ancestors)                  */
    int32_t n_frontal;           /* columns sampled by this clique                                    */
    int32_t obs_off;             /* offset of the clique's true observations in `obs`                 */
    int32_t sep_off;             /* offset in `cols` of the n_sep source column indices               */
    int32_t front_off;           /* offset in `cols` of the n_frontal destination column indices      */
    int32_t reserved;
} nfisam_post_clique;

/* FactorGraphSolver.sample_posterior (src/slam/FactorGraphSolver.py:497-550) for a whole tree in ONE
 * launch: for every clique in `table` order (root first) sample its frontal columns conditioned on its
 * true observations and on the separator columns sampled earlier, exactly as n_cliques successive
 * conditional_sample_given_observation calls would (slam/NFiSAM.py:120-155), but without leaving
 * the device.  Zt[total_dim][n]: standard-normal draws, consumed in WALK order (the j-th frontal column of
 * clique c takes row sum_{c'<c} n_frontal(c') + j), St[total_dim][n]: output samples (rows = columns of `cols`), both
 * COLUMN-major; cols/obs: device arrays indexed by the table; max_D: largest D_model.        */
int nfisam_nsf_posterior_walk(const nfisam_post_clique* table, int n_cliques, const int32_t* cols, const float* obs,
                              int max_D, int K, int H, float B, int L, int n, const float* Zt, float* St,
                              nfisam_stream_t stream);

/* Training-batch normalisation on the device (NFiSAM.normalize_training_samples, src/slam/NFiSAM.py:515-548):
 * per column c of x[n,D]: Euclidean -> mean / population std; circular[c] != 0 -> mean = direction of the mean
 * resultant (scipy.stats.circmean(., high=pi, low=-pi)), deviations wrapped to [-pi, pi), std of the wrapped
 * deviations; std clipped at 1e-5.  x_out[n,D] = (wrapped) deviation / std, mean[D], std[D].  Sums in double.
 * `circular` may be NULL (all Euclidean); x_out may alias x.                                             */
int nfisam_normalize_columns(const float* x, int n, int D, const uint8_t* circular, float* x_out, float* mean,
                             float* std, nfisam_stream_t stream);

/* ---- clique training-batch simulator (f-2) -------------------------------------------------------
 * One op of the ancestral-simulation schedule of a clique (sampler/SimulationBasedSampler.plan; reference:
 * src/sampler/SimulationBasedSampler.py:14-133).  Columns index the [n, D_total] batch; an SE(2) variable takes 3
 * consecutive columns (x y theta), an R2 variable 2.  p: SE(2) ops = pose / measurement (3) then the lower Cholesky
 * factor of the tangent-space noise covariance (l00 l10 l11 l20 l21 l22).                                     */
#define NFISAM_SIM_MAX_OPS 40
#define NFISAM_SIM_COPY       1   /* c.. <- k columns b.. of the row-major device array `src` with row stride a       */
#define NFISAM_SIM_PRIOR_SE2  2   /* c <- p[0:3] * Exp(eps)                       (Factors.py:725-743)                */
#define NFISAM_SIM_REL_FWD    3   /* c <- col a * (p[0:3] * Exp(eps))             (Factors.py:1196-1317)              */
#define NFISAM_SIM_REL_BWD    4   /* c <- col a * (p[0:3] * Exp(eps))^-1                                              */
#define NFISAM_SIM_REL_OBS    5   /* c <- (col a)^-1 (col b) * Exp(eps)           simulated odometry measurement      */
#define NFISAM_SIM_RING       6   /* c(xy) <- a(xy) + (p[0] + p[1] z)(cos phi, sin phi), phi ~ U(-pi, pi)  (:2575-2649) */
#define NFISAM_SIM_RANGE_OBS  7   /* c <- |b(xy) - a(xy)| + p[0] z                simulated range measurement         */
#define NFISAM_SIM_ADA_OBS    8   /* c <- range from a(xy) to ONE of the k candidates cand[], cumulative weights
                                     p[0:k], noise p[4] z                          (Factors.py:3146-3157)             */
#define NFISAM_SIM_NH_RING    9   /* RING whose noise is p[1] with probability p[3], else p[2]   (BinaryFactorWithNullHypo,   */
#define NFISAM_SIM_NH_OBS    10   /* RANGE_OBS whose noise is p[0] with probability p[2], else p[1]    Factors.py:3300-3462)  */
#define NFISAM_SIM_PRIOR_R2  11   /* c(xy) <- p[0:2] + L z, L = p[2] p[3] p[4] (l00 l10 l11)      (UnaryR2GaussianPriorFactor, :362)   */
#define NFISAM_SIM_PRIOR_R2_RING 12 /* c(xy) <- p[0:2] + (p[2] + p[3] z)(cos phi, sin phi)  (UnaryR2RangeGaussianPriorFactor, :451)      */
#define NFISAM_SIM_REL_R2_FWD 13  /* c(xy) <- a(xy) + p[0:2] + L z                                (R2RelativeGaussianLikelihoodFactor, */
#define NFISAM_SIM_REL_R2_BWD 14  /* c(xy) <- a(xy) - L z - p[0:2]                                 Factors.py:998-1030)                */
#define NFISAM_SIM_REL_R2_OBS 15  /* c(xy) <- b(xy) - a(xy) + L z                simulated displacement measurement                   */
typedef struct nfisam_sim_op {
    int32_t code;
    int32_t a, b, c;
    int32_t cand[4];
    int32_t k;
    float p[9];
    uint64_t src;
} nfisam_sim_op;

/* Simulate the n joint samples of a clique: every thread interprets `ops` (HOST array, copied into the launch) for its
 * own sample with a counter-based generator keyed by `seed`; x_out[n, D_out] row-major receives the first D_out
 * columns (D_total - D_out scratch columns may hold variables that are not part of the batch).                */
int nfisam_simulate_clique(const nfisam_sim_op* ops, int n_ops, int n, int D_out, int D_total, uint64_t seed,
                           float* x_out, nfisam_stream_t stream);

/* ---- training -------------------------------------------------------------------------- */
/* Vector-Jacobian product of the L-layer flow (what torch autograd computes for
 * `loss.backward()` in slam/NFiSAM.py:474): kgrad[L*kparam_count] += d<gz,z>/dtheta + d<gl,logdet>/dtheta,
 * gx[n,D] (nullable) = the same w.r.t. x.  nll_mode=1 ignores gz/gl and uses
 * sum_p(0.5|z_p|^2 - logdet_p) (the un-normalised NLL of NFiSAM.py:470-472; divide by n and add
 * D/2 log 2pi for the reference's loss); loss_sum[1] += that sum (nullable).
 * kgrad / loss_sum are ACCUMULATED into (caller zeroes them).                               */
int nfisam_nsf_backward(const float* x, const float* kparams, int n, int D, int K, int H, float B, int L,
                        size_t layer_stride, const float* gz, const float* gl, int nll_mode, float* kgrad,
                        float* gx, float* loss_sum, nfisam_stream_t stream);

/* Device-resident control block of one clique's training run. */
typedef struct nfisam_train_state {      /* 32 bytes since ABI 1200 (the unused loss_acc / loss_slots[64] of ABI 1100 are gone);
                                          * ABI 1300: reserved[0] of a plan's host mirror counts the chunks closed in the current run */
    int32_t step;        /* iterations completed and recorded so far (advances when a chunk is closed) */
    int32_t stop;        /* set by the device when the early-stop rule fired                 */
    int32_t have_avg;    /* a previous window mean exists                                    */
    float   loss_avg;    /* previous window mean (NFiSAM.py:481-491)                         */
    int32_t domain_err;  /* bit 0: a kernel saw a non-finite loss; bit 1 (NFISAM_STATE_STALLED, ABI 1400): a group
                          * barrier of a chunk-persistent launch timed out (-> NFISAM_ERR_STALL)            */
    int32_t reserved[3]; /* [0]: a plan's host mirror counts the chunks closed in the current run; [1] (ABI 1400): the
                          * most XCDs one (clique, dim) group of a chunk-persistent launch ran on (1 = the placement the
                          * grid asks for; diagnostic, correctness does not depend on it)                   */
} nfisam_train_state;

#define NFISAM_STATE_STALLED 2

typedef struct nfisam_adam_cfg {
    float lr, beta1, beta2, eps;      /* torch.optim.Adam defaults: betas .9/.999, eps 1e-8 (NFiSAM.py:425) */
    int32_t max_iters;                /* flow_iterations                                      */
    int32_t average_window;           /* <=0 disables early stopping                          */
    float loss_delta_tol;
    int32_t reserved;
} nfisam_adam_cfg;

/* One clique of a training batch: everything device-resident. */
typedef struct nfisam_clique {
    const float* x;              /* [n,D] normalised training batch (NFiSAM.py:379)           */
    float* kparams;              /* [L*kparam_count]                                          */
    float* adam_m;               /* [L*kparam_count] zero-initialised                         */
    float* adam_v;               /* [L*kparam_count] zero-initialised                         */
    float* kgrad;                /* [nfisam_nsf_grad_workspace_count(max n of the batch, D,..)] workspace, zero-initialised --
                                  * and ZEROED AGAIN by the caller whenever it resets `state` (step = 0) to re-run a plan: the
                                  * chunk-persistent kernel's blocks exchange gradient words as (value, tag) pairs whose tag is
                                  * the iteration's number in the run (state->step + i + 1), so a workspace that still holds the
                                  * pairs of an earlier run with the same numbering would be read as this run's (ABI 1400)     */
    float* iter_loss;            /* [max_iters] zero-initialised; per-iteration loss (NFiSAM.py:473) */
    nfisam_train_state* state;   /* zero-initialised                                          */
    int32_t n, D;
} nfisam_clique;

/* Floats the `kgrad` workspace of a clique must hold (gradient copies + a ring of 128 x 128 per-iteration
 * loss words behind them -- ABI 1600; 128 x 64 before: a lone clique of up to 256 blocks keeps two blocks per word, an
 * order-free sum -- + 64 counter words: the per-dim group barriers of the chunk-persistent training
 * kernel, which the library keeps zero between chunks -- the caller provides the workspace ZEROED) when the largest clique of its batch has n
 * particles: launches of <= 128 particle tiles write per-tile partial gradients with plain stores and
 * the Adam kernel sums them in tile order (no atomics); larger ones accumulate with float atomics
 * into a single copy.  A tile is 32 particles (two-lanes-per-particle kernel) or 64 (one lane per
 * particle; the dim-major kernel of one-layer flows writes one copy per block of 4 waves); the count is
 * the upper bound of all, plus -- for one-layer flows of <= 32 tiles of 64 -- the second set of copies
 * and the second (theta | m | v) buffer of the launches that apply the previous iteration's Adam update
 * themselves (nfisam_nsf_train_plan_run; no separate Adam launch per iteration); for multi-layer flows
 * of hidden width 8 and D <= 16 (ABI 1310) the clique's PANEL IMAGE (the parameters in the order the
 * multi-layer training kernel reads them from LDS, L x D panels; the Adam kernel keeps it current)
 * and, for latency-bound sizes, the forward state that kernel parks between its two passes.
 * Always allocate what this function returns -- the layout behind the gradient copies is the library's. */
size_t nfisam_nsf_grad_workspace_count(int n, int D, int K, int H, int L);

/* The gradient half of a training iteration on its own (forward + analytic backward + reduction into
 * each clique's `kgrad` workspace and loss slots, exactly the kernel `nfisam_nsf_train_step` launches
 * first; no Adam update, no bookkeeping, state->stop / step are still honoured).  With <= 128 tiles the
 * workspace is overwritten (per-tile slabs), so repeated calls are idempotent: bench.py times this
 * entry to price the dominant kernel; callers with their own optimiser use it as the gradient oracle. */
int nfisam_nsf_train_gradient(const nfisam_clique* cliques, int n_cliques, int cliques_on_host, int max_n,
                              int max_D, int K, int H, float B, int L, nfisam_stream_t stream);

/* Training plans split an iteration of a one-layer batch into `nfisam_nsf_train_chains(...)` launches that run as
 * parallel branches of the chunk's hipGraph (the (clique, dim) groups of a one-layer flow are independent optimisation
 * problems, reference loop src/slam/NFiSAM.py:451-494; a branch's kernel prologue and tail then run under another
 * branch's arithmetic).  `nfisam_nsf_train_gradient_part` is launch `chain` of `n_chains` of the gradient half, for
 * callers that want to time or drive the launches exactly as a plan issues them (bench.py: one stream per chain).
 * n_chains = 1 is nfisam_nsf_train_gradient.  Launch shapes that are not split return 1 from nfisam_nsf_train_chains
 * and NFISAM_ERR_ARG for n_chains > 1. */
int nfisam_nsf_train_chains(int n_cliques, int max_n, int max_D, int K, int H, int L);
int nfisam_nsf_train_gradient_part(const nfisam_clique* cliques, int n_cliques, int cliques_on_host, int max_n,
                                   int max_D, int K, int H, float B, int L, int chain, int n_chains,
                                   nfisam_stream_t stream);

/* One full-batch training iteration of `n_cliques` independent cliques (grid.y = clique):
 * forward + analytic backward + gradient reduction, the Adam update, and a one-wave bookkeeping kernel
 * that records iter_loss[step], evaluates the reference's window early-stop rule on the device and
 * advances state->step.  Cliques whose state->stop is set or whose step reached max_iters are skipped,
 * so the call can be replayed without host intervention.
 * `cliques` is a DEVICE array unless n_cliques==1 and `cliques_on_host` is non-zero.
 * All cliques share K, H, B, L and the Adam configuration; n and D may differ.             */
int nfisam_nsf_train_step(const nfisam_clique* cliques, int n_cliques, int cliques_on_host, int max_n,
                          int max_D, int K, int H, float B, int L, const nfisam_adam_cfg* cfg,
                          nfisam_stream_t stream);

/* Convenience loop == the `for i in range(flow_iterations)` loop of NFiSAM.fit_clique_density_model
 * (slam/NFiSAM.py:451-491) for a batch of independent cliques.  Enqueues iterations in chunks (the
 * largest divisor of `average_window` that is <= 128; 50 without early stopping): 2 kernels per iteration
 * plus one bookkeeping kernel per chunk -- the only writer of state->step / stop -- captured once as a
 * hipGraph and replayed, and SYNCHRONISES WITH THE HOST after each chunk to read the stop flags.  host_cliques: HOST array of descriptors (device pointers
 * inside); dev_cliques: the same array in device memory.  iters_run[n_cliques] (host) receives the
 * iterations each clique ran.                                                                */
int nfisam_nsf_train_loop(const nfisam_clique* host_cliques, const nfisam_clique* dev_cliques, int n_cliques,
                          int K, int H, float B, int L, const nfisam_adam_cfg* cfg, int use_graph,
                          int32_t* iters_run, nfisam_stream_t stream);

/* The same loop with the set-up split off, so that descriptor validation and hipGraph capture /
 * instantiation happen once (outside any timed region) and the plan can be re-run, e.g. after the
 * caller re-initialised parameters / Adam moments / state in place.  `plan_run` synchronises with
 * the host after every chunk of `average_window` (50 if early stopping is off) iterations.
 * Re-running a plan from step 0: re-initialise parameters, moments, `iter_loss`, `state` AND the `kgrad` workspace (zero) --
 * see nfisam_clique.kgrad; continuing a run (state left as it is) needs nothing.                 */
typedef struct nfisam_train_plan nfisam_train_plan;
int nfisam_nsf_train_plan_create(const nfisam_clique* host_cliques, const nfisam_clique* dev_cliques, int n_cliques,
                                 int K, int H, float B, int L, const nfisam_adam_cfg* cfg, int use_graph,
                                 nfisam_train_plan** out);
int nfisam_nsf_train_plan_run(nfisam_train_plan* plan, int32_t* iters_run, nfisam_stream_t stream);
/* ABI 1400, measurement only.  A plan created with `use_graph = 3` (graph | timing) splits a chunk-persistent chunk into two
 * graphs -- the training launch(es), then the closing Adam update + bookkeeping -- and records two timing events on the
 * stream around the first: after a run and a synchronisation, `ms` = GPU time of the persistent launch(es) of the most
 * recent chunk (bench.py's `roofline.kernel_us`).  One more graph launch per chunk: not for production plans.
 * NFISAM_ERR_ARG for plans without the events, NFISAM_ERR_LAUNCH when no persistent chunk has run. */
int nfisam_nsf_train_plan_kernel_ms(const nfisam_train_plan* plan, float* ms);
int nfisam_nsf_train_plan_destroy(nfisam_train_plan* plan);

/* Stepping a graph plan by hand (ABI 1320; the slot scheduler of slam/ReplicaNFiSAM.py: independent runs whose cliques
 * enter and leave ONE batched launch sequence as they finish -- the reference trains the cliques of its eight dataset
 * variants one after the other, example/slam/plaza_dataset/run_nfisam.py:11-21):
 *   begin    orders the plan's stream behind `stream` and restarts the chunk count of the host mirror;
 *   enqueue  appends ONE chunk of `average_window` iterations (non-blocking: enqueue a few ahead);
 *   peek     copies the host mirror, out[n_cliques]: each clique's state as of the last chunk closed and, in
 *            reserved[0], the number of chunks closed since `begin` (-1: a chunk closed during the copy, look again);
 *   stream   the plan's stream: a clique that has stopped (or reached max_iters) is not written by any launch any more,
 *            so the caller may re-initialise its slot for a NEW problem of the same (n, D) -- batch, parameters, zeroed
 *            moments / workspace / loss record and, LAST, the zeroed state -- with copies enqueued on this stream, i.e.
 *            between two chunks;
 *   refill   does exactly that for slot `clique` from the caller's device buffers x [n,D] / kparams (ordered behind
 *            `stream`; they must stay alive until the copies have run);
 *   feed     a thread of the library keeps `depth` chunks enqueued ahead of the last one closed, so that the caller's
 *            thread never sits in the graph launch (~0.3 ms per chunk); depth 0 pauses it (returns once it is outside a
 *            launch); `enqueued` = chunks enqueued since `begin`: a slot refilled when this read E trains from chunk
 *            E + 1 on at the latest, i.e. its mirror entry is its own once more than E chunks have closed;
 *   end      pauses the feeder and orders `stream` behind everything enqueued on the plan.                */
int nfisam_nsf_train_plan_begin(nfisam_train_plan* plan, nfisam_stream_t stream);
int nfisam_nsf_train_plan_enqueue(nfisam_train_plan* plan);
/* ABI 1400.  A training plan with the reference's HOLD-OUT stop rule (src/slam/NFiSAM.py:452-468, `training_set_frac < 1`):
 * every `validation_interval` iterations -- in front of iteration i whenever (i + 1) % validation_interval == 0 -- the
 * negative mean log-density of the held-out batch under the current parameters is evaluated on the device; the first time
 * it exceeds the previous evaluation the run is scheduled to end at slower_stop_iter = int(slower_stop_rate * (i + 1))
 * (the loop breaks in front of the iteration with i + 1 >= slower_stop_iter) and evaluation ceases.  The window rule is
 * off in this mode (cfg->average_window is ignored), as in the reference (`if testing_data is None`, :481).
 * One graph replay = one validation period; whole-number slower_stop_rate >= 1 only (the reference's default 2.0: the
 * scheduled end then falls on a period boundary), 1 <= validation_interval <= 129; anything else: NFISAM_ERR_ARG (callers
 * step such fits themselves).  state->have_avg / loss_avg hold the last validation loss, state->reserved[2] the scheduled
 * end (0: none).  `val[c].logprob` is scratch of n_val floats; `val[c].val_loss` (nullable) receives evaluation number k
 * at index k (max_iters / validation_interval entries).  All device pointers; `val` itself is a host array. */
typedef struct nfisam_validation {
    const float* x_val;       /* [n_val, D] held-out batch, normalised as the reference does (its OWN statistics, NFiSAM.py:381) */
    float* logprob;           /* [n_val] scratch                                                              */
    float* val_loss;          /* [max_iters / validation_interval] or NULL                                    */
    int32_t n_val;
    int32_t reserved;
} nfisam_validation;
int nfisam_nsf_train_plan_create_validated(const nfisam_clique* host_cliques, const nfisam_clique* dev_cliques, int n_cliques,
                                           int K, int H, float B, int L, const nfisam_adam_cfg* cfg,
                                           const nfisam_validation* val, int validation_interval, float slower_stop_rate,
                                           int use_graph, nfisam_train_plan** out);

/* ABI 1400.  Most XCDs one (clique, dim) group of the plan's chunk-persistent launches ran on (0: none has run; 1: what
 * the grid asks for; > 1: slower, equally correct -- the group's exchange uses agent-scope write-through stores).  */
int nfisam_nsf_train_plan_xcd_span(const nfisam_train_plan* plan);

/* ABI 1600 (round 6), not in the reference: the whole run of a SINGLE-CLIQUE plan enqueued on `stream` as one window-spanning launch
 * that evaluates the reference's window rule (src/slam/NFiSAM.py:481-491) itself, without the host -- the call returns at once.  The
 * plan must have been created with bit 2 (value 4) of `use_graph` set.  Outcome, once the stream has drained: `state->step` =
 * iterations run, `state->stop`, `state->domain_err` (bit 0 non-finite loss, NFISAM_STATE_STALLED), `iter_loss`; parameters and moments
 * are the trained ones -- the same bits nfisam_nsf_train_plan_run leaves.  -> NFISAM_ERR_ARG: this plan / this moment does not allow
 * it (no such graph; hidden_dim or width without the two-wave build; another run holds the chunk-persistent form): use
 * nfisam_nsf_train_plan_run. */
int nfisam_nsf_train_plan_launch_async(nfisam_train_plan* plan, nfisam_stream_t stream);
int nfisam_nsf_train_plan_feed(nfisam_train_plan* plan, int depth);
long nfisam_nsf_train_plan_enqueued(const nfisam_train_plan* plan);
int nfisam_nsf_train_plan_peek(const nfisam_train_plan* plan, nfisam_train_state* out);
nfisam_stream_t nfisam_nsf_train_plan_stream(nfisam_train_plan* plan);
int nfisam_nsf_train_plan_refill(nfisam_train_plan* plan, int clique, const float* x, const float* kparams,
                                 nfisam_stream_t stream);
int nfisam_nsf_train_plan_end(nfisam_train_plan* plan, nfisam_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NFISAM_HIP_H */
