"""slam.NFiSAM — density-model adapter of NF-iSAM on MI355X (reference: src/slam/NFiSAM.py).

Same names and call signatures as the reference: `NFiSAMArgs`, `NormalizingFlowModelWithSeparator`,
`FlowsPriorFactor`, `NFiSAM` with the solver hooks `fit_clique_density_model`,
`root_clique_density_model_to_leaf`, `clique_density_to_separator_factor`.

What changes underneath:
  * the `for i in range(flow_iterations)` loop (NFiSAM.py:451-491: forward, loss, autograd
    backward, Adam, window early stop — ~4.4k eager ops per iteration) is ONE device-resident loop:
    a fused forward+backward kernel and a fused Adam/early-stop kernel per iteration, replayed from
    a hipGraph; the host only reads a stop flag once per `average_window` iterations;
  * conditional sampling (NFiSAM.py:120-155) is one kernel that also normalises the given columns
    and un-normalises / angle-wraps the result;
  * trained models stay on the GPU (the reference moves them to the CPU because its sampling
    runs there, NFiSAM.py:507-511).
`cuda_training` / `adaptive_flow_setup` / `data_parallel` are accepted and ignored: there is no CPU
path to choose.
"""
import logging
import time
from collections.abc import Mapping
from typing import List

import numpy as np
import torch

import nfisam_hip as _nh
from flows.flows import NSF_AR
from flows.models import NormalizingFlowModel
from flows.prior_dist import CustomMultivariateNormal, MultivariateNormalVonmises  # noqa: F401
from slam.FactorGraphSolver import CliqueSeparatorFactor, ConditionalSampler, FactorGraphSolver, SolverArgs, \
    run_incrementally
from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
from utils.Functions import theta_to_pipi


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("slam.NFiSAM needs a ROCm device (MI355X): the flow hot path has no CPU fallback")
    return "cuda:%d" % torch.cuda.current_device()


class NFiSAMArgs(SolverArgs):
    """reference: NFiSAM.py:18-66 (same keyword names and defaults)."""

    def __init__(self, elimination_method: str = "pose_first", posterior_sample_num: int = 500,
                 local_sample_num: int = 500, store_clique_samples: bool = False, local_sampling_method="direct",
                 learning_rate: float = 0.015, flow_number: int = 1, flow_type: str = "NSF_AR",
                 flow_iterations: int = 10, num_knots: int = 12, cuda_training: bool = False,
                 adaptive_flow_setup: bool = False, hidden_dim: int = 8, average_window=50, loss_delta_tol=1e-2,
                 training_set_frac=1.0, validation_interval=10, slower_stop_rate=2.0, data_parallel=False,
                 training_loss_dir=None, device_simulation=True, lazy_posterior=False, async_fits=False, *args, **kwargs):
        super().__init__(elimination_method=elimination_method, posterior_sample_num=posterior_sample_num,
                         local_sample_num=local_sample_num, store_clique_samples=store_clique_samples,
                         local_sampling_method=local_sampling_method, *args, **kwargs)
        self.flow_number = flow_number
        self.flow_type = flow_type
        self.flow_iterations = flow_iterations
        self.num_knots = num_knots
        self.cuda_training = cuda_training
        self.learning_rate = learning_rate
        self.adaptive_flow_setup = adaptive_flow_setup
        self.hidden_dim = hidden_dim
        self.average_window = average_window
        self.loss_delta_tol = loss_delta_tol
        self.training_set_frac = training_set_frac
        self.validation_interval = validation_interval
        self.slower_stop_rate = slower_stop_rate
        self.data_parallel = data_parallel
        self.training_loss_dir = training_loss_dir
        # not in the reference: simulate and normalise the clique training batches on the GPU.
        # True: one fused kernel per clique (csrc/clique_sim.hip); False: the factors' host (numpy, float64) samplers.
        # Cliques with a factor type the device simulator does not know fall back to the host.  Plaza1: the fused
        # simulator takes ~1.3 s off the 4-5 s spent outside training.
        self.device_simulation = device_simulation
        # not in the reference: `sample_posterior` / `incremental_inference` return at once with a read-only mapping that waits for
        # the tree walk and its device-to-host copy when it is first LOOKED AT (LazyPosterior below) -- a caller that adds the next
        # nodes and factors and updates the graphs before it reads the samples has that host work run under the walk.
        self.lazy_posterior = lazy_posterior
        # not in the reference: a clique's whole fit is ENQUEUED (one window-spanning launch that evaluates the early-stop rule on the
        # device: nfisam_nsf_train_plan_launch_async) and `fit_clique_density_model` returns without waiting -- the next clique's
        # simulation, normalisation and fit read the trained model on the device, in stream order, so the host runs ahead of the GPU
        # through the whole upward pass and looks at the fits' outcomes ONCE, behind it.  What it gives up: the reference-deviating
        # "retry a non-finite fit once" of the synchronous path (an async update with a failed fit raises instead), and per-fit
        # timers (they time the enqueue; the wait is added to the update's last fit).  Fits whose shape has no window-spanning
        # form (hold-out validation, several layers, hidden_dim / width without the two-wave build) run synchronously as before.
        self.async_fits = async_fits
        self.tl_cnt = 0


class LazyPosterior(Mapping):
    """What `NFiSAM.sample_posterior` returns under `NFiSAMArgs(lazy_posterior=True)`: variable -> samples [n, dim], as the dict
    of the synchronous call, filled in at the first access (waits for the walk's event; the copy into pinned host memory was
    enqueued behind the walk on the same stream, so nothing later on that stream is waited for)."""

    def __init__(self, handle, host, done):
        self._handle, self._host, self._done, self._dict = handle, host, done, None

    def _get(self):
        if self._dict is None:
            self._done.synchronize()
            S = self._host.numpy()
            pcol = self._handle["pcol"]
            self._dict = {v: S[:, pcol[v]:pcol[v] + v.dim] for v in self._handle["order"]}
            self._handle = None                    # (the models the walk read may go now)
        return self._dict

    def ready(self) -> bool:
        return self._dict is not None or self._done.query()

    def __getitem__(self, v): return self._get()[v]
    def __iter__(self): return iter(self._get())
    def __len__(self): return len(self._get())
    def __contains__(self, v): return v in self._get()


_CIRC_FLAGS = {}     # (device, circular flags as bytes) -> uint8 device tensor, immutable (NormalizingFlowModelWithSeparator._norm_dev)


class NormalizingFlowModelWithSeparator(NormalizingFlowModel, ConditionalSampler):
    """Joint density of a clique's (observation | separator | frontal) columns as a normalizing
    flow with input normalisation (reference: NFiSAM.py:68-199)."""

    def __init__(self, flows, prior, separator_prior, circular_dim_list, samples_mean: "torch.Tensor" = None,
                 samples_std: "torch.Tensor" = None):
        super().__init__(prior, flows)
        self.separator_prior = separator_prior
        self.separator_dim = separator_prior.dim if separator_prior is not None else 0
        self.samples_mean = samples_mean
        self.samples_std = samples_std
        self.circular_dim_list = circular_dim_list
        self._norm_cache = None

    @property
    def dim(self):
        return len(self.circular_dim_list)

    # ---- device copies of the normalisation constants --------------------------------------
    def _norm_dev(self, device):
        if self._norm_cache is None or self._norm_cache[0] != str(device):
            circ8 = np.asarray(self.circular_dim_list, dtype=np.uint8)
            if torch.is_tensor(self.samples_mean) and self.samples_mean.is_cuda and torch.is_tensor(self.samples_std) and self.samples_std.is_cuda:
                # (the device normalisation produced them where they are needed)
                mean = self.samples_mean.to(device=device, dtype=torch.float32).contiguous()
                std = self.samples_std.to(device=device, dtype=torch.float32).contiguous()
                # the flags by CONTENT: a run has a handful of patterns (which columns are angles), so after the first models of
                # a run no copy happens at all (the first one is host-synchronous: complete on return, safe on any stream)
                key = (str(device), circ8.tobytes())
                circ = _CIRC_FLAGS.get(key)
                if circ is None:
                    circ, = _nh.upload(circ8, device=device, cached=True)
                    _CIRC_FLAGS[key] = circ
            else:
                # ONE host-synchronous copy for the three (kept for the model's life and used on any stream: `cached`)
                mean, std, circ = _nh.upload(torch.as_tensor(self.samples_mean, dtype=torch.float32).numpy(),
                                             torch.as_tensor(self.samples_std, dtype=torch.float32).numpy(), circ8,
                                             device=device, cached=True)
            self._norm_cache = (str(device), mean, std, circ)
        return self._norm_cache[1:]

    def posterior_static(self):
        """Device pointers and shape constants the tree-walk kernel needs for this clique; computed once
        after training (parameters of a trained clique model never change) and kept alive by the model."""
        st = getattr(self, "_post_static", None)
        if st is None:
            f0, L, device = self._flow_cfg()
            kp = self.kernel_params()
            mean, std, circ = self._norm_dev(device)
            st = dict(cfg=(f0.K, f0.hidden_dim, f0.B, L), device=device, D_model=f0.dim, keep=(kp, mean, std, circ),
                      ptrs=(kp.data_ptr(), mean.data_ptr(), std.data_ptr(), circ.data_ptr()))
            self._post_static = st
        return st

    def _flow_cfg(self):
        f0 = self.flows[0]
        if not self._homogeneous():
            raise NotImplementedError("the fused sampling kernel needs identical NSF_AR layers")
        return f0, len(self.flows), f0.device

    # ---- reference API: host-visible (un)normalisation (NFiSAM.py:96-118) ---------------------
    def normalize_samples(self, samples, init_dim):
        circ = np.asarray(self.circular_dim_list, dtype=bool)[init_dim:init_dim + samples.shape[-1]]
        ci = np.where(circ)[0]
        ei = np.setdiff1d(np.arange(samples.shape[-1]), ci)
        mean = torch.as_tensor(self.samples_mean).to(samples.device)
        std = torch.as_tensor(self.samples_std).to(samples.device)
        samples[:, ci] = theta_to_pipi(samples[:, ci] - mean[ci + init_dim]) / std[ci + init_dim]
        samples[:, ei] = (samples[:, ei] - mean[ei + init_dim]) / std[ei + init_dim]
        return samples

    def unnormalize_samples(self, normalized, init_dim):
        circ = np.asarray(self.circular_dim_list, dtype=bool)[init_dim:init_dim + normalized.shape[-1]]
        ci = np.where(circ)[0]
        ei = np.setdiff1d(np.arange(normalized.shape[-1]), ci)
        mean = torch.as_tensor(self.samples_mean).to(normalized.device)
        std = torch.as_tensor(self.samples_std).to(normalized.device)
        normalized[:, ei] = normalized[:, ei] * std[ei + init_dim] + mean[ei + init_dim]
        normalized[:, ci] = theta_to_pipi(normalized[:, ci] * std[ci + init_dim] + mean[ci + init_dim])
        return normalized

    # ---- sampling -------------------------------------------------------------------------------
    def conditional_sample_given_observation(self, conditional_dim, obs_samples=None,
                                             sample_number=None) -> "np.ndarray":
        """Samples of C in P(C | O = obs_samples) (reference: NFiSAM.py:120-138)."""
        if sample_number is None and obs_samples is not None:
            n_samples, x_s = obs_samples.shape[0], obs_samples
        elif sample_number is not None:
            n_samples, x_s = sample_number, None
        else:
            raise ValueError("must input one of obs_samples or sample_number")
        f0, L, device = self._flow_cfg()
        # the reference draws all D latent columns and slices (NFiSAM.py:136); only the needed ones are drawn here
        z = torch.randn(n_samples, conditional_dim, device=device, dtype=torch.float32)
        return self.inverse_given_separator(z, x_s).cpu().numpy()

    def conditional_sample_on_device(self, conditional_dim, obs_row=None, sample_number=None) -> "torch.Tensor":
        """`conditional_sample_given_observation` with ONE observation row tiled on the device and the samples
        left there (the child -> parent message of the on-device batch simulator)."""
        f0, L, device = self._flow_cfg()
        z = torch.randn(int(sample_number), conditional_dim, device=device, dtype=torch.float32)
        x_s = None
        if obs_row is not None and len(obs_row) > 0:
            x_s = _nh.upload(np.asarray(obs_row, dtype=np.float32).reshape(-1), device=device)[0].reshape(1, -1) \
                .expand(int(sample_number), -1).contiguous()
        return self.inverse_given_separator(z, x_s)

    def inverse_given_separator(self, z, x_s=None):
        """z: latent samples [n, c]; x_s: UN-normalised given columns [n, Ds] (numpy or tensor).
        Returns un-normalised samples of columns Ds .. Ds+c-1 (reference: NFiSAM.py:140-155)."""
        f0, L, device = self._flow_cfg()
        z = torch.as_tensor(z, dtype=torch.float32).to(device).contiguous()
        xs = None
        if x_s is not None:
            xs = torch.as_tensor(np.float32(x_s) if isinstance(x_s, np.ndarray) else x_s,
                                 dtype=torch.float32).to(device).contiguous()
        mean, std, circ = self._norm_dev(device)
        return _nh.inverse(z, xs, self.kernel_params(), f0.K, f0.hidden_dim, f0.B, L, mean=mean, std=std,
                           circular=circ, model_D=f0.dim)

    def separator_forward(self, x):
        """Push UN-normalised samples of the first `separator_dim` columns to the latent space with
        the marginal flow (valid because the flow is autoregressive; reference: NFiSAM.py:157-173).
        Returns (z, separator_prior_logprob, separator_log_det)."""
        m, d = x.shape
        assert d == self.separator_dim
        f0, L, device = self._flow_cfg()
        x = torch.as_tensor(x, dtype=torch.float32).to(device).clone()
        xn = self.normalize_samples(x, init_dim=0).contiguous()
        z, ld, lp = _nh.forward(xn, self.kernel_params(), f0.K, f0.hidden_dim, f0.B, L, want_logprob=True,
                                model_D=f0.dim)
        return z, lp - ld, ld

    def separator_grad_x_log_pdf(self, x):
        """d/dx [prior_logprob + log_det] of `separator_forward` w.r.t. the UN-normalised input."""
        f0, L, device = self._flow_cfg()
        x = torch.as_tensor(x, dtype=torch.float32).to(device).clone()
        xn = self.normalize_samples(x, init_dim=0).contiguous()
        _, gx, _ = _nh.backward(xn, self.kernel_params(), f0.K, f0.hidden_dim, f0.B, L, nll_mode=True, want_gx=True,
                                model_D=f0.dim)
        _, std, _ = self._norm_dev(device)
        return -gx / std[:x.shape[1]]

    @property
    def is_cpu(self):
        return self.prior.is_cpu()

    def to_cpu(self):
        raise RuntimeError("this model runs on the GPU only; use state_dict() to export parameters")

    def to(self, device: str):
        dev_sep = None if self.separator_prior is None else self.separator_prior.to(device)
        return NormalizingFlowModelWithSeparator(flows=self.flows.to(device), prior=self.prior.to(device),
                                                 separator_prior=dev_sep, circular_dim_list=self.circular_dim_list,
                                                 samples_mean=self.samples_mean, samples_std=self.samples_std)


class FlowsPriorFactor(CliqueSeparatorFactor):
    """A trained child clique seen from its parent: prior over the separator variables given the
    clique's true observations (reference: NFiSAM.py:202-315).  `sample` is the child->parent
    message of NF-iSAM."""

    def __init__(self, vars: List, flow_model: NormalizingFlowModelWithSeparator, true_obs: np.ndarray,
                 circular_dim_list: List) -> None:
        super().__init__()
        self._vars = vars
        self._flow_model = flow_model
        self._is_gaussian = False
        self._true_obs = true_obs
        self._obs_dim = len(true_obs)
        self._circular_dim_list = circular_dim_list[:]
        assert self.dim == len(circular_dim_list)

    def append_obs_sample(self, x):
        """Prepend the true observations to x (observation columns come first in the flow)."""
        if self._obs_dim == 0:
            return x
        return np.concatenate((np.tile(self._true_obs, (x.shape[0], 1)), x), axis=1)

    def log_pdf(self, x: np.ndarray, **kwargs) -> np.ndarray:
        """log p(obs, x) up to the (fixed) observation constant (reference: NFiSAM.py:233-251)."""
        aug = self.append_obs_sample(x)
        _, lp, ld = self._flow_model.separator_forward(np.float32(aug))
        return (lp + ld).cpu().numpy()

    def grad_x_log_pdf(self, x: np.ndarray, **kwargs) -> np.ndarray:
        aug = self.append_obs_sample(x)
        g = self._flow_model.separator_grad_x_log_pdf(np.float32(aug)).cpu().numpy()
        return g[:, self._obs_dim:self._obs_dim + x.shape[1]]

    def sample(self, num_samples: int, **kwargs) -> np.ndarray:
        if self._obs_dim == 0:
            return self._flow_model.conditional_sample_given_observation(conditional_dim=self.dim,
                                                                         sample_number=num_samples)
        obs_samples = np.tile(self._true_obs, (num_samples, 1))
        return self._flow_model.conditional_sample_given_observation(conditional_dim=self.dim,
                                                                     obs_samples=obs_samples)

    def sample_on_device(self, num_samples: int) -> "torch.Tensor":
        """`sample` without leaving the GPU (sampler.DeviceSimulation)."""
        return self._flow_model.conditional_sample_on_device(conditional_dim=self.dim,
                                                             obs_row=self._true_obs if self._obs_dim else None,
                                                             sample_number=num_samples)

    def unif_to_sample(self, u) -> np.ndarray:
        """Nested-sampling prior transform: uniform [1,D] -> sample [D] (reference: NFiSAM.py:290-303)."""
        import scipy.stats
        normal_var = np.array([scipy.stats.norm.ppf(u)]).astype(np.float32).reshape(1, -1)
        obs = None if self._obs_dim == 0 else np.tile(self._true_obs, (1, 1))
        x = self._flow_model.inverse_given_separator(torch.tensor(normal_var), x_s=obs)
        return x[0, :].cpu().numpy()

    @property
    def is_gaussian(self) -> bool:
        return self._is_gaussian

    @property
    def vars(self) -> List:
        return self._vars

    @property
    def circular_dim_list(self) -> List[bool]:
        return self._circular_dim_list


class NFiSAM(FactorGraphSolver):
    TRAIN_PLAN_CACHE = 6      # training plans kept (least recently used first out)

    def __init__(self, args: NFiSAMArgs = None):
        super().__init__(args=args if args is not None else NFiSAMArgs())
        a = self._args
        # fail at construction, not in the middle of an incremental update (the kernels are compiled per (K, H))
        if a.flow_type != "NSF_AR":
            raise NotImplementedError("Unknown flow type for the pipeline")
        if not _nh.supported(a.num_knots, a.hidden_dim):
            raise ValueError("num_knots=%r, hidden_dim=%r: the gfx950 kernels are instantiated for num_knots in 2..16 and "
                             "hidden_dim in 1..16 (compiled widths 4, 8, 16; others zero-padded: nf-isam_amd/csrc/nsf_units.h)" % (a.num_knots, a.hidden_dim))
        if int(a.flow_number) < 1:
            raise ValueError("flow_number must be >= 1")
        import threading
        self._train_lock = threading.RLock()     # train_prepared: one caller at a time per solver

    # The loss curves of the update's fits (reference attribute: FactorGraphSolver._temp_training_loss, clique name ->
    # list of per-iteration losses) are fetched from the device when somebody LOOKS at them: one copy for all fits recorded
    # since the last look instead of one blocking copy per fit (a fit's curve is ready on the device long before that).
    @property
    def _temp_training_loss(self) -> dict:
        inflight = self.__dict__.pop("_loss_inflight", None)
        if inflight is not None:                 # (lazy posterior: the copy was enqueued in front of the walk, see _prefetch_loss_record)
            names, host, done = inflight
            done.synchronize()
            flat = host.numpy().astype(np.float64)
            off = 0
            for name, cnt in names:
                self.__dict__["_loss_record"][name] = flat[off:off + cnt].tolist()
                off += cnt
        pending = self.__dict__.get("_loss_pending")
        if pending:
            self.__dict__["_loss_pending"] = []
            host = torch.cat([t for _, t in pending]).cpu().numpy().astype(np.float64)
            off = 0
            for name, t in pending:
                self.__dict__["_loss_record"][name] = host[off:off + t.numel()].tolist()
                off += t.numel()
        return self.__dict__["_loss_record"]

    @_temp_training_loss.setter
    def _temp_training_loss(self, value: dict):
        self.__dict__["_loss_record"] = value
        self.__dict__["_loss_pending"] = []
        self.__dict__.pop("_loss_inflight", None)

    def _prefetch_loss_record(self):
        """Lazy posterior: the device-to-host copy of the update's loss curves is enqueued NOW, in front of the tree walk (pinned
        memory, its own event), so that a look at `_temp_training_loss` after `incremental_inference` waits for the fits only --
        a blocking copy behind the walk would wait for the walk."""
        pending = self.__dict__.get("_loss_pending")
        if not pending or self.__dict__.get("_loss_inflight") is not None:
            return
        self.__dict__["_loss_pending"] = []
        flat = torch.cat([t for _, t in pending])
        host = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
        host.copy_(flat, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream())
        self.__dict__["_loss_inflight"] = ([(name, int(t.numel())) for name, t in pending], host, done)

    def _simulation_backend(self):
        if not getattr(self._args, "device_simulation", False) or not torch.cuda.is_available():
            return None
        be = self.__dict__.get("_sim_backend")
        if be is None:
            from sampler.DeviceSimulation import FusedSimulationBackend
            be = self._sim_backend = FusedSimulationBackend(_device())
        return be

    # ---- the hot loop ---------------------------------------------------------------------------
    def fit_clique_density_model(self, clique, samples: np.ndarray, var_ordering: List, timer: List, *args,
                                 **kwargs) -> NormalizingFlowModelWithSeparator:
        """Train the clique's flow on `samples` [n, D] (columns = simulated observations, separator
        variables, frontal variables).  reference: NFiSAM.py:323-513.
        = `prepare_fit` (normalise, initialise) + `train_prepared` (the device-resident loop) + `finish_fit` (wrap the
        trained parameters); `slam.ReplicaNFiSAM` calls the three parts itself to train the cliques of several
        replicas in ONE batched launch sequence."""
        prep = self.prepare_fit(clique, samples, var_ordering)
        if prep["testing_data"] is not None:
            return self._fit_holdout(prep, timer)
        opt_start = time.time()
        # (only from here, and only when THIS class's upward pass will collect the outcome: ParallelNFiSAM / ReplicaNFiSAM have their own)
        prep["allow_async"] = bool(getattr(self._args, "async_fits", False)) and \
            type(self).fit_tree_density_models is NFiSAM.fit_tree_density_models
        self.train_prepared([prep])
        if prep.get("state_dev") is None:              # (async_fits: the fit is in flight; `fit_tree_density_models` collects it)
            torch.cuda.synchronize()
        if timer is not None:
            timer.append(time.time() - opt_start)
        return self.finish_fit(prep)

    def fit_tree_density_models(self, timer: List[float] = None, clique_dim_timer: List[List[float]] = None, *args, **kwargs):
        """The reference's upward pass (FactorGraphSolver.fit_tree_density_models); with `async_fits` the outcomes of the fits that
        were only enqueued are looked at here, once: ONE device-to-host copy of their state words behind the pass."""
        try:
            super().fit_tree_density_models(timer=timer, clique_dim_timer=clique_dim_timer, *args, **kwargs)
        finally:
            pend = self.__dict__.pop("_async_fits", [])
        if pend:
            t0 = time.time()
            states = torch.stack([p["state_dev"] for p in pend]).cpu().numpy()
            for p, s in zip(pend, states):
                p["iters"] = int(s[0])
                if int(s[4]) & 2:                      # NFISAM_STATE_STALLED
                    raise _nh.PersistentStall("an enqueued fit stalled (a block of its persistent launch never became resident): "
                                              "run the update again with async_fits=False")
                if int(s[4]) != 0:
                    raise _nh.DomainError("non-finite loss in an enqueued fit (async_fits has no retry: the update's later fits "
                                          "were computed from it)")
            self.last_fit_iterations = pend[-1]["iters"]
            if timer is not None and len(timer) > 0:
                timer[-1] += time.time() - t0          # (the update's last fit timer carries the wait for all of them)

    def prepare_fit(self, clique, samples, var_ordering) -> dict:
        """Everything of `fit_clique_density_model` in front of the training loop (NFiSAM.py:323-449): circular flags,
        shuffle + split, normalisation (on the device when the batch was simulated there), fresh parameters."""
        a = self._args
        if a.flow_type != "NSF_AR":
            raise NotImplementedError("Unknown flow type for the pipeline")
        device = _device()
        frontal_dim = clique.frontal_dim
        aug_clique_dim = samples.shape[-1]
        circular_dim_list = []
        for var in var_ordering:
            circular_dim_list += var.circular_dim_list
        if len(circular_dim_list) != aug_clique_dim:
            # observation columns precede the variables and are Euclidean
            circular_dim_list = [False] * (aug_clique_dim - len(circular_dim_list)) + circular_dim_list

        on_device = isinstance(samples, torch.Tensor) and samples.is_cuda
        if on_device and a.training_set_frac < 1.0:
            samples, on_device = samples.cpu().numpy(), False       # the hold-out split is done on the host
        testing_data = None
        if on_device:
            # batch simulated on the GPU (sampler.DeviceSimulation): normalise it there too (f-3).  With
            # training_set_frac = 1 the reference's shuffle only permutes the rows of a full-batch mean.
            training_data, means, stds = _nh.normalize_columns(samples.to(torch.float32).contiguous(),
                                                               circular_dim_list)
        else:
            # train/test split on a shuffled COPY (the reference shuffles the caller's array in place)
            samples = np.array(samples, dtype=np.float64, copy=True)
            train_size = min(int(samples.shape[0] * a.training_set_frac), samples.shape[0])
            np.random.shuffle(samples)
            train_samples, test_samples = samples[:train_size], samples[train_size:]
            training_data, means, stds = self.normalize_training_samples(train_samples, circular_dim_list,
                                                                         a.flow_type)
            if len(test_samples) > 0:
                testing_data, _, _ = self.normalize_training_samples(test_samples, circular_dim_list, a.flow_type)

        # Parameters are initialised directly on the device in the reference's order (one op per layer) and
        # trained as kernel-layout blobs; the nn.Module tree of each NSF_AR is only built on demand.
        from flows.flows import init_reference_blob
        K, H, B, L = a.num_knots, a.hidden_dim, 5.0, a.flow_number
        if not _nh.supported(K, H):
            raise ValueError("no kernel instantiation for num_knots=%d, hidden_dim=%d" % (K, H))
        kp0 = torch.cat([_nh.pack(init_reference_blob(aug_clique_dim, K, H, device), aug_clique_dim, K, H, 1)
                         for _ in range(L)])
        return dict(clique=clique, training_data=training_data, testing_data=testing_data, means=means, stds=stds,
                    circular=circular_dim_list, kp0=kp0, D=aug_clique_dim, sep_dim=aug_clique_dim - frontal_dim,
                    n=int(training_data.shape[0]), cfg=(K, H, B, L), device=device)

    def train_prepared(self, preps: List[dict], retry: bool = True) -> None:
        """The reference's `for i in range(flow_iterations)` loop (NFiSAM.py:451-491) for one or several prepared
        cliques at once (grid.y = clique; every clique has its own Adam state, loss record and early-stop decision).
        Training plans (device buffers + the captured hipGraph of one chunk of iterations) are kept per batch shape
        and re-used: the batches and the fresh parameters are copied into the plan's buffers, Adam moments / state /
        loss record are cleared in place.  Fills prep["trained"], ["iters"], ["iter_loss"]."""
        # One caller at a time per solver: the plan cache below is an LRU whose eviction CLOSES a TrainBatch, and a worker
        # thread of slam.ReplicaNFiSAM may be inside this function (an odd-shaped batch) while the scheduler retries a failed
        # clique through the same solver (re-entrant: the retry below calls back in).
        with self._train_lock:
            return self._train_prepared_locked(preps, retry)

    def _train_prepared_locked(self, preps: List[dict], retry: bool) -> None:
        a = self._args
        K, H, B, L = preps[0]["cfg"]
        device = preps[0]["device"]
        key = (tuple((p["n"], p["D"]) for p in preps), K, H, L, float(a.learning_rate), int(a.flow_iterations),
               int(a.average_window), float(a.loss_delta_tol), str(device))
        # Plans (hipGraphExec + pinned state mirror + device buffers incl. the multi-copy gradient workspace) are kept for
        # the batch shapes used most recently: replicas dropping out of a lock-step batch and cliques of varying width make
        # almost every composition a new key, so the cache is a small LRU and evicted plans are closed.
        import collections
        plans = self.__dict__.setdefault("_train_plans", collections.OrderedDict())
        tb = plans.get(key)
        if tb is not None:
            plans.move_to_end(key)
        if tb is None:
            while len(plans) >= self.TRAIN_PLAN_CACHE:
                _, old = plans.popitem(last=False)
                old.close()
            tb = _nh.TrainBatch([torch.empty(p["n"], p["D"], dtype=torch.float32, device=device) for p in preps],
                                [torch.zeros_like(p["kp0"]) for p in preps], K, H, B, L, lr=a.learning_rate,
                                max_iters=a.flow_iterations, average_window=a.average_window,
                                loss_delta_tol=a.loss_delta_tol, early_stop=True)
            plans[key] = tb
        for x, p in zip(tb.xs, preps):
            x.copy_(p["training_data"])
        tb.reset(kparams=[p["kp0"] for p in preps])
        logger = logging.getLogger("flows on clique")
        if preps[0].get("allow_async") and len(preps) == 1 and retry and tb.launch_async():
            # the whole fit is enqueued; what the caller needs of it is copied out behind it, in stream order (the plan and its
            # buffers serve the next clique of this shape)
            p = preps[0]
            p["trained"], p["iter_loss"], p["state_dev"], p["iters"] = tb.kparams[0].clone(), tb.iter_loss[0].clone(), tb.states[0].clone(), -1
            self.__dict__.setdefault("_async_fits", []).append(p)
            self.async_fits_enqueued = getattr(self, "async_fits_enqueued", 0) + 1       # (diagnostic: fits that took this path)
            return
        try:
            try:
                iters = tb.run(use_graph=True)
            except _nh.PersistentStall:
                # Not a numerical event: a chunk-persistent launch gave up waiting for a block of its own (another process
                # held its place on the device).  The SAME fit -- same batch, same initial parameters -- runs again; the
                # library has switched this process to one launch per iteration, whose results are the same bit for bit.
                logger.warning("a chunk-persistent training launch stalled: re-running the fit with one launch per iteration")
                tb.reset(kparams=[p["kp0"] for p in preps])
                iters = tb.run(use_graph=True)
        except _nh.DomainError:
            # A non-finite loss (the reference raises "Input outside domain" / fails its discriminant assert there,
            # src/flows/utils.py:74-76,133, and the whole update dies).  The other cliques of the batch have run to their
            # own end; a failed clique is retried ONCE from a fresh initialisation, on its own.
            iters = tb.last_iters
            failed = [c for c in range(len(preps)) if tb.state(c)["domain_err"]]
            if retry and failed:
                from flows.flows import init_reference_blob
                for c in failed:
                    logger.warning("non-finite loss while fitting clique %d of the batch: retrying once with fresh parameters", c)
                    p = preps[c]
                    # The fresh draw comes from a generator of its own, seeded by the failed initialisation: it consumes
                    # nobody's random stream (replicas run this outside their RNG turn and must stay bit-identical to
                    # the sequential run) and is reproducible.  `retried` tells callers and tests that this fit deviates
                    # from the reference, which aborts the update here.
                    gen = torch.Generator(device=device)
                    gen.manual_seed(int(p["kp0"].double().abs().sum().item() * 1e6) % (2 ** 31 - 1) + 1)
                    p["kp0"] = torch.cat([_nh.pack(init_reference_blob(p["D"], K, H, device, generator=gen), p["D"], K, H, 1)
                                          for _ in range(L)])
                    p["retried"] = True
                    self.train_prepared([p], retry=False)
                done = set(failed)
                for c, p in enumerate(preps):
                    if c not in done:
                        p["trained"], p["iters"], p["iter_loss"] = tb.kparams[c].clone(), iters[c], tb.iter_loss[c].clone()
                return
            raise
        for c, p in enumerate(preps):
            if iters[c] < a.flow_iterations:
                logger.info(f"Early stopping at iter {iters[c]}")
            p["trained"] = tb.kparams[c].clone()
            p["iters"] = iters[c]
            p["iter_loss"] = tb.iter_loss[c].clone()

    def finish_fit(self, prep: dict) -> NormalizingFlowModelWithSeparator:
        """Behind the loop (NFiSAM.py:493-513): wrap the trained parameters, record the loss curve."""
        K, H, B, L = prep["cfg"]
        D, device = prep["D"], prep["device"]
        Pk = _nh.kparam_count(D, K, H)
        trained = prep["trained"]
        flows = [NSF_AR.from_kernel_params(D, K, B, H, trained[l * Pk:(l + 1) * Pk]) for l in range(L)]
        normal_clique = CustomMultivariateNormal(dim=D, device=device)
        normal_separator = CustomMultivariateNormal(dim=prep["sep_dim"], device=device) if prep["sep_dim"] > 0 else None
        model = NormalizingFlowModelWithSeparator(flows, normal_clique, normal_separator, prep["circular"],
                                                  prep["means"], prep["stds"])
        clique_name = ''.join([str(var.name) for var in prep["clique"].vars])
        self.__dict__["_loss_pending"].append((clique_name, prep["iter_loss"]))
        self.last_fit_iterations = prep["iters"]
        self.last_fit_retried = bool(prep.get("retried", False))
        return model

    def _fit_holdout(self, prep: dict, timer) -> NormalizingFlowModelWithSeparator:
        """training_set_frac < 1: the reference's validation-driven stop (NFiSAM.py:452-468), one iteration per launch."""
        a = self._args
        K, H, B, L = prep["cfg"]
        device = prep["device"]
        logger = logging.getLogger("flows on clique")
        opt_start = time.time()
        x_dev = prep["training_data"].to(device).contiguous()
        x_val = prep["testing_data"].to(device).contiguous()
        # The rule runs on the device as part of the training plan (one graph replay per validation period, no host
        # synchronisation per evaluation: nfisam_nsf_train_plan_create_validated) when the scheduled end
        # int(slower_stop_rate x (i + 1)) falls on a period boundary -- whole-number rates, the reference's default is 2.0;
        # other settings are stepped from the host, iteration by iteration.
        on_device = float(a.slower_stop_rate).is_integer() and a.slower_stop_rate >= 1 and 1 <= int(a.validation_interval) <= 129
        kp_init = prep["kp0"].clone()                  # (the plan trains prep["kp0"] in place)
        tb = None
        try:
            if on_device:
                tb = _nh.TrainBatch([x_dev], [prep["kp0"]], K, H, B, L, lr=a.learning_rate, max_iters=a.flow_iterations,
                                    early_stop=False, x_val=[x_val], validation_interval=a.validation_interval,
                                    slower_stop_rate=a.slower_stop_rate)
                try:
                    prep["iters"] = tb.run(use_graph=True)[0]
                except _nh.PersistentStall:
                    # as in `_train_prepared_locked`: not a numerical event -- the SAME fit runs again from its initial state
                    # (the library keeps to one launch per iteration from now on: same bits)
                    logger.warning("a chunk-persistent training launch stalled: re-running the hold-out fit with one launch per iteration")
                    tb.reset(kparams=[kp_init])
                    prep["iters"] = tb.run(use_graph=True)[0]
                if prep["iters"] < a.flow_iterations:
                    logger.info(f"Slower stop at iter {prep['iters'] + 1}")
                self.last_validation_losses = tb.val_loss[0]
            else:
                tb = _nh.TrainBatch([x_dev], [prep["kp0"]], K, H, B, L, lr=a.learning_rate, max_iters=a.flow_iterations,
                                    average_window=a.average_window, loss_delta_tol=a.loss_delta_tol, early_stop=False)
                f0 = NSF_AR.from_kernel_params(prep["D"], K, B, H, prep["kp0"])
                prep["iters"] = self._fit_with_validation(tb, x_val, f0, logger)
            torch.cuda.synchronize()
            if timer is not None:
                timer.append(time.time() - opt_start)
            prep["trained"] = tb.kparams[0]
            prep["iter_loss"] = tb.iter_loss[0]
            return self.finish_fit(prep)
        finally:
            if tb is not None:
                tb.close()

    def _fit_with_validation(self, tb, testing_data, f0, logger):
        """training_set_frac < 1: hold-out early stopping (reference: NFiSAM.py:452-468)."""
        a = self._args
        last_validation_loss = float('inf')
        slower_stop_iter = None
        iters = 0
        for i in range(a.flow_iterations):
            if slower_stop_iter is not None:
                if (i + 1) >= slower_stop_iter:
                    logger.info(f"Slower stop at iter {i + 1}")
                    break
            elif (i + 1) % a.validation_interval == 0:
                _, _, lp = _nh.forward(testing_data, tb.kparams[0], f0.K, f0.hidden_dim, f0.B, a.flow_number,
                                       want_z=False, want_logdet=False, want_logprob=True)
                new_loss = float(-lp.mean().item())
                logger.info(f"Iter: {i + 1}\t, validation loss: {new_loss}")
                if new_loss > last_validation_loss:
                    logger.info(f"Early stopping at iter {i + 1}")
                    slower_stop_iter = int(a.slower_stop_rate * (i + 1))
                else:
                    last_validation_loss = new_loss
            tb.step()
            iters = i + 1
        return iters

    def normalize_training_samples(self, samples, circular_dim_list, flow_type: str):
        """Per-column standardisation with circular statistics for angle columns
        (reference: NFiSAM.py:515-548).  Host side: n x D doubles, once per clique."""
        if flow_type != "NSF_AR":
            raise NotImplementedError("Unknown flow type for the pipeline")
        samples = np.array(samples, dtype=np.float64, copy=True)
        aug_clique_dim = samples.shape[-1]
        means = np.zeros(aug_clique_dim)
        stds = np.zeros(aug_clique_dim)
        ci = np.where(circular_dim_list)[0]
        ei = np.setdiff1d(np.arange(aug_clique_dim), ci)
        if len(ci) > 0:
            # scipy.stats.circmean(., high=pi, low=-pi) in closed form: direction of the mean resultant
            means[ci] = theta_to_pipi(np.arctan2(np.sin(samples[:, ci]).sum(0), np.cos(samples[:, ci]).sum(0)))
            shifted = theta_to_pipi(samples[:, ci] - means[ci])
            stds[ci] = np.std(shifted, axis=0)
            samples[:, ci] = shifted
        means[ei] = np.mean(samples[:, ei], axis=0)
        stds[ei] = np.std(samples[:, ei], axis=0)
        samples[:, ei] = samples[:, ei] - means[ei]
        stds = np.clip(stds, a_min=1e-5, a_max=None)
        samples = samples / stds
        return torch.Tensor(samples), torch.Tensor(means), torch.Tensor(stds)

    # ---- posterior: the whole tree in one launch ------------------------------
    def sample_posterior(self, timer: List = None, *args, **kwargs):
        """Root -> leaves conditional sampling of every clique (reference:
        FactorGraphSolver.sample_posterior, FactorGraphSolver.py:497-550) as ONE kernel launch: each
        wave of 64 samples walks all cliques on the device; one D2H copy at the end.  The per-clique
        device pointers are cached on the model (they do not change after training); only the column
        indices, which shift as the elimination ordering grows, are rebuilt per update."""
        if getattr(self._args, "lazy_posterior", False):
            self._prefetch_loss_record()
            handle = self.posterior_launch()
            S = handle["S"]
            host = torch.empty(S.shape, dtype=S.dtype, pin_memory=True)
            host.copy_(S, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream())
            if timer is not None:
                timer.append(time.time() - handle["start"])       # (what the caller waited for: table assembly and launch)
            return LazyPosterior(handle, host, done)
        return self.posterior_collect(self.posterior_launch(), timer)

    def posterior_launch(self):
        """First half of `sample_posterior`: assemble the clique table and enqueue the walk on the current stream.
        -> handle for `posterior_collect` (several solvers can have their walks in flight on different streams:
        slam.ReplicaNFiSAM)."""
        start = time.time()
        num_samples = self._args.posterior_sample_num
        order = self._elimination_ordering
        # every variable owns a permanent column range of the sample matrix (assigned when first seen), so a
        # clique's column indices never change and are cached on its model together with the device pointers
        pcol = self.__dict__.setdefault("_post_col", {})
        total_dim = self.__dict__.get("_post_total", 0)
        for v in order:
            if v not in pcol:
                pcol[v] = total_dim
                total_dim += v.dim
        self._post_total = total_dim
        cliques, stack = [], [self._physical_bayes_tree.root]
        while stack:
            c = stack.pop()
            cliques.append(c)
            stack.extend(c.children)
        rows, cols, obs, cfg, device, max_D = [], [], [], None, None, 1
        rmap = self._reverse_ordering_map
        models = []                            # kept alive by the handle: the walk reads their parameters until it is collected
        for clique in cliques:
            model = self._clique_density_model[clique]
            models.append(model)
            e = model.__dict__.get("_post_entry")
            if e is None:
                st = model.posterior_static()
                o = np.asarray(self._clique_true_obs[clique], dtype=np.float32).ravel()
                sep = [pcol[v] + k for v in sorted(clique.separator, key=rmap.__getitem__) for k in range(v.dim)]
                fro = [pcol[v] + k for v in sorted(clique.frontal, key=rmap.__getitem__) for k in range(v.dim)]
                row = np.zeros(1, dtype=_nh.POST_DTYPE)
                row["kparams"], row["mean"], row["std"], row["circular"] = st["ptrs"]
                row["D_model"] = st["D_model"]
                row["n_obs"], row["n_sep"], row["n_frontal"] = o.size, len(sep), len(fro)
                e = dict(row=row.tobytes(), cols=np.asarray(sep + fro, dtype=np.int32), obs=o, cfg=st["cfg"],
                         device=st["device"], D_model=st["D_model"])
                model.__dict__["_post_entry"] = e
            if cfg is None:
                cfg, device = e["cfg"], e["device"]
            elif cfg != e["cfg"]:
                raise NotImplementedError("the tree walk needs one (K, H, B, L) for all cliques")
            rows.append(e["row"]); cols.append(e["cols"]); obs.append(e["obs"])
            if e["D_model"] > max_D:
                max_D = e["D_model"]
        table = np.frombuffer(b"".join(rows), dtype=_nh.POST_DTYPE).copy()
        n_obs, n_sep = table["n_obs"].astype(np.int64), table["n_sep"].astype(np.int64)
        n_col = n_sep + table["n_frontal"]
        table["obs_off"] = np.cumsum(n_obs) - n_obs
        table["sep_off"] = np.cumsum(n_col) - n_col
        table["front_off"] = table["sep_off"] + n_sep
        K, H, B, L = cfg
        S = _nh.posterior_walk_raw(table, np.concatenate(cols), np.concatenate(obs), total_dim, num_samples, max_D,
                                   K, H, B, L, device)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream())
        return dict(S=S, pcol=pcol, order=list(order), start=start, stream=torch.cuda.current_stream(), keep=models, done=done)

    def posterior_collect(self, handle, timer: List = None, copy_stream=None):
        """Second half: wait for the walk, one D2H copy, per-variable views of the sample matrix (in the elimination ordering
        of the launch).  `copy_stream`: copy behind the walk's event on that stream instead of the walk's own (which may
        already hold the caller's NEXT walk: slam.ReplicaNFiSAM)."""
        if copy_stream is not None:
            copy_stream.wait_event(handle["done"])
        with torch.cuda.stream(copy_stream if copy_stream is not None else handle["stream"]):
            S = handle["S"].cpu().numpy()
        pcol = handle["pcol"]
        samples = {v: S[:, pcol[v]:pcol[v] + v.dim] for v in handle["order"]}
        if timer is not None:
            timer.append(time.time() - handle["start"])
        return samples

    # ---- model reuse / message construction --------------------------------------------------
    def root_clique_density_model_to_leaf(self, old_clique, new_clique, device=None):
        """Same variables, different frontal/separator split: re-wrap the trained flows instead of
        re-training (reference: NFiSAM.py:550-577)."""
        old_flow = self._clique_density_model[old_clique]
        obs_dim = old_flow.dim - old_clique.dim
        separator_dim = new_clique.separator_dim + obs_dim
        if not isinstance(old_flow.flows[0], NSF_AR):
            raise NotImplementedError("Unknown flow type for the pipeline")
        dev = old_flow.prior._device
        normal_separator = CustomMultivariateNormal(dim=separator_dim, device=dev) if separator_dim > 0 else None
        return NormalizingFlowModelWithSeparator(flows=old_flow.flows, prior=old_flow.prior,
                                                 separator_prior=normal_separator,
                                                 circular_dim_list=old_flow.circular_dim_list,
                                                 samples_mean=old_flow.samples_mean, samples_std=old_flow.samples_std)

    def clique_density_to_separator_factor(self, separator_var_list, density_model, true_obs):
        """true_obs: 1-D array concatenating all observations of the clique (reference: NFiSAM.py:579-586)."""
        obs_dim = true_obs.shape[-1]
        obs_separator_dim = sum([var.dim for var in separator_var_list]) + obs_dim
        return FlowsPriorFactor(vars=separator_var_list, flow_model=density_model, true_obs=true_obs,
                                circular_dim_list=density_model.circular_dim_list[obs_dim: obs_separator_dim])


def NFiSAM_empirial_study(knots, iters, training_samples, learning_rates, hidden_dims, case_dir, data_file,
                          data_format, incremental_step=1, prior_cov_scale=0.1, traj_plot=False, plot_args=None,
                          check_root_transform=False, **kwargs):
    """Grid driver of the example scripts (reference: NFiSAM.py:589-609): parse the graph, group it
    into incremental updates and run one solver per hyper-parameter combination.  Returns the run
    directories (the reference returns None)."""
    import os
    nodes, truth, factors = graph_file_parser(data_file=os.path.join(case_dir, data_file), data_format=data_format,
                                              prior_cov_scale=prior_cov_scale)
    nodes_factors_by_step = group_nodes_factors_incrementally(nodes=nodes, factors=factors,
                                                              incremental_step=incremental_step)
    run_dirs = []
    for knt in knots:
        for it in iters:
            for training_sample in training_samples:
                for lr in learning_rates:
                    for hidden_dim in hidden_dims:
                        args = NFiSAMArgs(num_knots=knt, flow_iterations=it, local_sample_num=training_sample,
                                          learning_rate=lr, hidden_dim=hidden_dim, **kwargs)
                        solver = NFiSAM(args)
                        run_dirs.append(run_incrementally(case_dir, solver, nodes_factors_by_step, truth, traj_plot,
                                                          plot_args, check_root_transform))
    return run_dirs
