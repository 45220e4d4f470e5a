"""Generator of BASELINE config 4: a Manhattan-world range-only SLAM graph with ambiguous data association.

Own restatement of the recipe of the reference's example
(example/slam/manhattan_world_with_range/manhattan_plaza/factor_graph_generator.py:11-89 with
src/manhattan_world_with_range/Simulator.py:117-190,253-316; the shipped instance has 136 poses):
  * 20 x 20 grid of vertices, cell size 20 m; 4 beacons on distinct vertices of the square (2,2)-(17,17);
  * the robot walks the boundary of its rectangular area counter-clockwise, then lawn-mows the area column by
    column, one vertex per step; pose heading = direction of travel (turn in place, then drive one cell);
  * odometry: relative pose perturbed on the manifold with cov diag((20*[s, s/5, s/10])^2), s = 0.01;
    prior on X0 with cov diag(1e-4, 1e-6, 1e-8);
  * every pose measures the range (sigma 2 m) to ONE random beacon.  With probability p_outlier (0 in the shipped
    cases) the measurement is an outlier: shifted by outlier_scale * sigma and wrapped in a BinaryFactorWithNullHypo
    (weights .5/.5, null sigma scale = outlier_scale = 5).  Else with probability p_ada (0.4), if that beacon is
    already in the graph and at least two beacons are, it becomes an AmbiguousDataAssociationFactor over the true
    beacon and up to max_ada-1 other known ones (uniform weights); otherwise a plain range factor (which introduces
    the beacon if it is new).
With the default robot area (3,3)-(15,14) the walk has 46 - 1 + 156 = 201 poses ("200 poses").

usage: make_manhattan.py out.fg [seed] [x0 y0 x1 y1]      (P_OUTLIER=<prob> in the environment adds outliers)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd"))
from factors.Factors import (AmbiguousDataAssociationFactor, BinaryFactorWithNullHypo,   # noqa: E402
                             SE2R2RangeGaussianLikelihoodFactor,
                             SE2RelativeGaussianLikelihoodFactor, UnarySE2ApproximateGaussianPriorFactor)
from geometry.TwoDimension import se2_compose, se2_exp, se2_inverse   # noqa: E402
from slam.FactorGraphSimulator import factor_graph_to_string   # noqa: E402
from slam.Variables import R2Variable, SE2Variable, VariableType   # noqa: E402

CELL, GRID = 20.0, 20


def boundary_then_lawnmower(x0, y0, x1, y1):
    """Vertices visited: rectangle boundary counter-clockwise from (x0, y0) (closed loop minus its last vertex),
    then all vertices column by column, alternating direction."""
    edge = [(x, y0) for x in range(x0, x1 + 1)] + [(x1, y) for y in range(y0 + 1, y1 + 1)] + \
           [(x, y1) for x in range(x1 - 1, x0 - 1, -1)] + [(x0, y) for y in range(y1 - 1, y0, -1)]
    lawn, flip = [], False
    for y in range(y0, y1 + 1):
        col = [(x, y) for x in range(x0, x1 + 1)]
        lawn += col[::-1] if flip else col
        flip = not flip
    return edge + lawn


def generate(seed=0, area=(3, 3, 15, 14), n_beacons=4, range_std=2.0, odom_scale=0.01, p_ada=0.4, max_ada=3,
             p_outlier=0.0, outlier_scale=5.0):
    rng = np.random.RandomState(seed)
    verts = [(i, j) for i in range(2, 18) for j in range(2, 18)]
    beacons = [verts[k] for k in rng.choice(len(verts), size=n_beacons, replace=False)]
    beacon_xy = {"L%d" % b: CELL * np.array(v, dtype=float) for b, v in enumerate(beacons)}
    odom_cov = np.diag((CELL * np.array([odom_scale, odom_scale / 5, odom_scale / 10])) ** 2)
    path = boundary_then_lawnmower(*area)
    # drop consecutive duplicates (the lawn-mower starts where the boundary walk started)
    path = [p for k, p in enumerate(path) if k == 0 or p != path[k - 1]]
    variables, truth, factors, known = [], {}, [], []

    def measure(pose_var, pose):
        name = "L%d" % rng.randint(n_beacons)
        r = float(np.linalg.norm(beacon_xy[name] - pose[:2])) + range_std * rng.randn()
        var = R2Variable(name, variable_type=VariableType.Landmark)
        odd = rng.rand()
        if odd < p_outlier:
            if var not in known:
                known.append(var)
                truth[var] = beacon_xy[name]
            factors.append(BinaryFactorWithNullHypo(pose_var, var, np.array([.5, .5]), SE2R2RangeGaussianLikelihoodFactor,
                                                    r + outlier_scale * range_std, range_std, null_sigma_scale=outlier_scale))
        elif odd < p_outlier + p_ada and var in known and len(known) > 1:
            others = [v for v in known if v != var]
            rng.shuffle(others)
            observed = [var] + others[:max_ada - 1]
            factors.append(AmbiguousDataAssociationFactor(pose_var, observed, np.ones(len(observed)) / len(observed),
                                                          SE2R2RangeGaussianLikelihoodFactor, r, range_std))
        else:
            if var not in known:
                known.append(var)
                truth[var] = beacon_xy[name]
            factors.append(SE2R2RangeGaussianLikelihoodFactor(pose_var, var, r, range_std))

    pose = np.array([CELL * path[0][0], CELL * path[0][1], 0.0])
    last = SE2Variable("X0")
    variables.append(last); truth[last] = pose.copy()
    factors.append(UnarySE2ApproximateGaussianPriorFactor(last, pose, np.diag([1e-4, 1e-6, 1e-8])))
    measure(last, pose)
    for k, v in enumerate(path[1:], start=1):
        goal = CELL * np.array(v, dtype=float)
        heading = np.arctan2(goal[1] - pose[1], goal[0] - pose[0])
        nxt = np.array([goal[0], goal[1], heading])
        move = se2_compose(se2_inverse(pose)[0], nxt)[0]                      # turn in place, drive one cell
        noise = np.linalg.cholesky(odom_cov) @ rng.randn(3)
        noisy = se2_compose(move, se2_exp(noise[None, :])[0])[0]
        var = SE2Variable("X%d" % k)
        variables.append(var); truth[var] = nxt.copy()
        factors.append(SE2RelativeGaussianLikelihoodFactor(last, var, noisy, odom_cov))
        measure(var, nxt)
        pose, last = nxt, var
    return variables + known, truth, factors


if __name__ == "__main__":
    out = sys.argv[1]
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    area = tuple(int(v) for v in sys.argv[3:7]) if len(sys.argv) >= 7 else (3, 3, 15, 14)
    vs, tr, fs = generate(seed=seed, area=area, p_outlier=float(os.environ.get("P_OUTLIER", "0")))
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as fh:
        fh.write(factor_graph_to_string(vs, fs, tr))
    n_pose = sum(1 for v in vs if str(v.name).startswith("X"))
    print("wrote %s: %d poses, %d landmarks, %d factors (%d ambiguous)" % (
        out, n_pose, len(vs) - n_pose, len(fs), sum(isinstance(f, AmbiguousDataAssociationFactor) for f in fs)))
