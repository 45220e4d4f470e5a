"""CPU-only tests of the drop-in module surface: importability, parameter naming/ordering identical
to the reference, host-side normalisation against the reference golden, loud failure without GPU."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_modules_import_and_expose_reference_names():
    import flows.flows as ff
    import flows.models as fm
    import flows.prior_dist as fp
    import flows.utils as fu
    import slam.NFiSAM as sn
    import slam.Variables as sv
    for mod, names in ((ff, ["FCNN", "NSF_AR", "unconstrained_RQS"]), (fm, ["NormalizingFlowModel"]),
                       (fp, ["CustomMultivariateNormal", "MultivariateNormalVonmises"]),
                       (fu, ["unconstrained_RQS", "RQS", "searchsorted", "DEFAULT_MIN_BIN_WIDTH"]),
                       (sn, ["NFiSAMArgs", "NFiSAM", "NormalizingFlowModelWithSeparator", "FlowsPriorFactor"]),
                       (sv, ["Variable", "R2Variable", "SE2Variable", "R1Variable", "Bearing2DVariable", "VariableType"])):
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)


def test_nsf_ar_state_dict_matches_reference_layout():
    from flows.flows import NSF_AR
    g = dict(np.load(os.path.join(GOLDEN, "nsf_n128_d6_k9.npz")))
    ref_keys = sorted(k[4:].replace("__", ".") for k in g if k.startswith("p0__"))
    f = NSF_AR(dim=6, K=9, hidden_dim=8)
    sd = f.state_dict()
    assert sorted(sd.keys()) == ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == g["p0__" + k.replace(".", "__")].shape
    # .parameters() order == reference order == oracle blob order
    from oracle import nsf_torch as O
    f.load_state_dict({k: torch.tensor(g["p0__" + k.replace(".", "__")]) for k in ref_keys})
    blob = O.blob_from_state_dict({k: g["p0__" + k.replace(".", "__")] for k in ref_keys}, 6)
    np.testing.assert_array_equal(f.reference_blob().detach().numpy(), blob)
    # init_param ~ U(-1/2, 1/2) (flows.py:62-63)
    assert float(NSF_AR(4, K=12).init_param.abs().max()) <= 0.5


def test_cpu_tensors_fail_loudly():
    from flows.flows import NSF_AR
    f = NSF_AR(dim=3, K=9)
    with pytest.raises(RuntimeError):
        f(torch.zeros(5, 3))
    with pytest.raises(RuntimeError):
        f.inverse(torch.zeros(5, 3))


def test_normalize_training_samples_matches_reference_golden():
    from slam.NFiSAM import NFiSAM, NFiSAMArgs, NormalizingFlowModelWithSeparator
    g = dict(np.load(os.path.join(GOLDEN, "normalize.npz")))
    circ = [bool(c) for c in g["circular"]]
    solver = NFiSAM(NFiSAMArgs())
    src = g["samples"].copy()
    td, mu, sd = solver.normalize_training_samples(src, circ, "NSF_AR")
    np.testing.assert_array_equal(src, g["samples"])            # caller's array is not modified
    np.testing.assert_allclose(mu.numpy(), g["mean"], atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(sd.numpy(), g["std"], atol=1e-7, rtol=1e-6)
    np.testing.assert_allclose(td.numpy(), g["train_norm"], atol=1e-5, rtol=1e-5)
    with pytest.raises(NotImplementedError):
        solver.normalize_training_samples(src, circ, "NSF_AR_CS")
    m = NormalizingFlowModelWithSeparator([], None, None, circ, mu, sd)
    np.testing.assert_allclose(m.normalize_samples(torch.tensor(np.float32(g["q"])), 0).numpy(), g["q_norm_init0"],
                               atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(m.unnormalize_samples(torch.tensor(g["zz"].copy()), 4).numpy(), g["zz_unnorm_init4"],
                               atol=1e-5, rtol=1e-5)


def test_args_defaults_match_reference():
    from slam.NFiSAM import NFiSAMArgs
    a = NFiSAMArgs()
    assert (a.num_knots, a.hidden_dim, a.learning_rate, a.flow_iterations, a.flow_number, a.flow_type) == \
        (12, 8, 0.015, 10, 1, "NSF_AR")
    assert (a.average_window, a.loss_delta_tol, a.training_set_frac, a.elimination_method) == (50, 1e-2, 1.0, "pose_first")
    assert "num_knots" in a.jsonStr()


def test_variables():
    from slam.Variables import SE2Variable, R2Variable, Variable, VariableType
    assert SE2Variable("X0").circular_dim_list == [False, False, True]
    assert R2Variable("L0", VariableType.Landmark).circular_dim_list == [False, False]
    v = Variable.construct_from_text("Variable Pose SE2 X3")
    assert v.name == "X3" and v.dim == 3 and v == SE2Variable("X3") and hash(v) == hash("X3")


def test_searchsorted_matches_reference_golden():
    from flows.utils import searchsorted
    g = dict(np.load(os.path.join(GOLDEN, "rqs_direct.npz")))
    bins = torch.tensor(g["ss_bins"].copy())
    idx = searchsorted(bins, torch.tensor(g["ss_q"]))
    np.testing.assert_array_equal(idx.numpy(), g["ss_idx"])
    np.testing.assert_array_equal(bins.numpy(), g["ss_bins_after"])     # in-place eps bump, like the reference


def test_file2vars_reads_an_ordering_file(tmp_path):
    """`Variable.file2vars` (reference src/slam/Variables.py:142-154): names from a run folder's ordering file; 'L...' are R2
    landmarks, the rest poses of the requested space."""
    from slam.Variables import R2Variable, SE2Variable, Variable, VariableType
    p = tmp_path / "step3_ordering"
    p.write_text("X0 X1 L1 X2 L2")
    vs = Variable.file2vars(str(p))
    assert [v.name for v in vs] == ["X0", "X1", "L1", "X2", "L2"]
    assert [type(v) for v in vs] == [SE2Variable, SE2Variable, R2Variable, SE2Variable, R2Variable]
    assert [v.type for v in vs] == [VariableType.Pose, VariableType.Pose, VariableType.Landmark, VariableType.Pose, VariableType.Landmark]
    assert [v.dim for v in Variable.file2vars(str(p), pose_space="R2")] == [2, 2, 2, 2, 2]
    single = tmp_path / "one"
    single.write_text("X7\n")
    assert [v.name for v in Variable.file2vars(str(single))] == ["X7"]


def test_multivariate_normal_vonmises_prior():
    """`flows.prior_dist.MultivariateNormalVonmises` (reference prior_dist.py:29-70): independent columns, N(0,1) for
    Euclidean and VonMises(0, 1) for circular ones; log_prob = sum of the column log densities; sample shape / support."""
    import torch
    from flows.prior_dist import MultivariateNormalVonmises
    circ = [False, True, False, True]
    d = MultivariateNormalVonmises(circ)
    assert d.dim == 4 and d.is_cpu()
    torch.manual_seed(0)
    x = d.sample((4000,))
    assert tuple(x.shape) == (4000, 4)
    assert float(x[:, 1].abs().max()) <= np.pi + 1e-6 and float(x[:, 3].abs().max()) <= np.pi + 1e-6
    assert abs(float(x[:, 0].std()) - 1.0) < 0.05 and abs(float(x[:, 2].mean())) < 0.06
    # circular variance of VonMises(kappa = 1): 1 - I1(1)/I0(1) = 0.5536
    assert abs(1.0 - float(torch.cos(x[:, 1]).mean()) - 0.5536) < 0.03
    lp = d.log_prob(x[:16])
    ref = torch.zeros(16)
    for c, is_c in enumerate(circ):
        col = x[:16, c]
        ref = ref + (torch.distributions.VonMises(torch.zeros(1), torch.ones(1)).log_prob(col) if is_c
                     else torch.distributions.Normal(0.0, 1.0).log_prob(col))
    np.testing.assert_allclose(lp.numpy(), ref.numpy(), atol=1e-5)
    with pytest.raises(ValueError):
        d.log_prob(x[:, :3])
    assert d.to("cpu").dim == 4 and d.cpu().is_cpu()
