"""slam.FactorGraphSimulator — `.fg` text reader/writer (reference: src/slam/FactorGraphSimulator.py:20-74).
Line formats:  `Variable <Pose|Landmark> <SE2|R2|...> <name> <truth...>`  /  `Factor <ClassName> <args...>`."""
from typing import Dict, Iterable, List, Tuple

import numpy as np

from factors.Factors import Factor
from slam.Variables import Variable


def read_variable_and_truth_from_line(line: str) -> Tuple[Variable, np.ndarray]:
    var = Variable.construct_from_text(line)
    tok = line.strip().split()
    return var, np.array([float(t) for t in tok[4:4 + var.dim]])


def write_variable_and_truth_to_line(var: Variable, truth: np.ndarray = None) -> str:
    line = str(var)
    if truth is not None:
        line += " " + " ".join(str(v) for v in truth)
    return line


def factor_graph_to_string(variables: Iterable[Variable], factors: Iterable[Factor],
                           var_truth: Dict[Variable, np.ndarray] = None) -> str:
    var_truth = var_truth or {}
    return "\n".join([write_variable_and_truth_to_line(v, var_truth.get(v)) for v in variables] +
                     [str(f) for f in factors])


def read_factor_graph_from_file(file_name: str) -> Tuple[List[Variable], Dict[Variable, np.ndarray], List[Factor]]:
    variables, truth, factors = [], {}, []
    with open(file_name) as fh:
        for line in fh:
            tok = line.strip().split()
            if not tok:
                continue
            if tok[0] == "Variable":
                var, val = read_variable_and_truth_from_line(line)
                variables.append(var)
                truth[var] = val
            elif tok[0] == "Factor":
                factors.append(Factor.construct_from_text(line, variables))
    return variables, truth, factors
