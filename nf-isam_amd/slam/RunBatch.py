"""slam.RunBatch — grouping of variables/factors into incremental updates
(reference: src/slam/RunBatch.py:90-336)."""
from typing import List, Tuple, Union

from factors.Factors import AmbiguousDataAssociationFactor, BinaryFactor, Factor, OdomFactor, \
    UnaryFactor
from slam.FactorGraphSimulator import read_factor_graph_from_file
from slam.Variables import Variable, VariableType


def graph_file_parser(data_file: str, data_format: Union['fg', 'g2o', 'toro'], prior_cov_scale=0.1):
    if data_format != 'fg':
        raise NotImplementedError("only the `.fg` format is rebuilt (g2o/TORO readers: SURVEY.md §2 row 14, out of scope)")
    return read_factor_graph_from_file(data_file)


def group_nodes_factors_incrementally(nodes: List[Variable], factors: List[Factor], incremental_step: int = None,
                                      multirobot=True) -> List[Tuple[List[Variable], List[Factor]]]:
    """Replay the graph as a robot would have built it: per time step the new pose(s) of every robot
    (pose names are `<robot letter><time index>`), their odometry / prior / observation factors and
    the landmarks seen for the first time; `incremental_step` time steps form one update
    (reference: multirbt_group_nodes_factors_incrementally, :226-336)."""
    robots = {}
    max_t = 0
    for idx, v in enumerate(nodes):
        if v.type == VariableType.Pose:
            rid, t = str(v.name)[0], int(str(v.name)[1:])
            robots.setdefault(rid, {})[t] = v
            max_t = max(max_t, t)
    by_var = {}

    def attach(var, kind, fidx):
        by_var.setdefault(var, {}).setdefault(kind, []).append(fidx)

    for fidx, f in enumerate(factors):
        if isinstance(f, UnaryFactor):
            attach(f.vars[0], "prior", fidx)
        elif isinstance(f, BinaryFactor):
            v1, v2 = f.var1, f.var2
            if v1.type == v2.type == VariableType.Pose:
                consecutive = isinstance(f, OdomFactor) and \
                    str(v1.name)[0] == str(v2.name)[0] and int(str(v2.name)[1:]) - int(str(v1.name)[1:]) == 1
                attach(v2 if consecutive else v1, "odom" if consecutive else "pose_obsv", fidx)
            elif v1.type == VariableType.Pose and v2.type == VariableType.Landmark:
                attach(v1, "lmk_obsv", fidx)
            else:
                raise ValueError("Unknown factors: " + str(f))
        elif isinstance(f, AmbiguousDataAssociationFactor):
            kind = "pose_obsv" if f.child_vars[0].type == VariableType.Pose else "lmk_obsv"
            attach(f.root_var, kind, fidx)
    if incremental_step is None or incremental_step > max_t + 1 or incremental_step <= 0:
        incremental_step = max_t + 1
    out, new_vars, new_factors, seen_lmks = [], [], [], set()
    for t in range(max_t + 1):
        for rid, poses in robots.items():
            if t not in poses:
                continue
            v = poses[t]
            new_vars.append(v)
            for kind, idxs in by_var.get(v, {}).items():
                new_factors += idxs
            for fidx in by_var.get(v, {}).get("lmk_obsv", []):
                for lm in factors[fidx].vars[1:]:
                    if lm not in seen_lmks:
                        seen_lmks.add(lm)
                        new_vars.append(lm)
                        new_factors += by_var.get(lm, {}).get("prior", [])
        if (t + 1) % incremental_step == 0 or t == max_t:
            out.append([list(new_vars), [factors[j] for j in new_factors]])
            new_vars, new_factors = [], []
    return out
