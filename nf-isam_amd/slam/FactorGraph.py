"""slam.FactorGraph — factor graph container + symbolic elimination into a Bayes tree
(reference: src/slam/FactorGraph.py:11-263).  Host-side bookkeeping."""
from typing import Dict, List, Set

from slam.BayesTree import BayesTree, BayesTreeNode
from slam.Variables import Variable, VariableType


class FactorGraph(object):
    def __init__(self) -> None:
        self._vars: List[Variable] = []       # insertion order (the orderings are derived from it)
        self._var_set: Set[Variable] = set()  # membership: the incremental solver rebuilds sub graphs of hundreds of variables
        self._factors: List = []              # every update -- a list scan per add was 11 M __eq__ calls on Plaza1
        self._by_var: Dict[Variable, List[int]] = {}   # variable -> indices of its factors (round 6: the sub graph of an update is cut
                                                       # out of the factors of the AFFECTED variables, not out of all of them)

    def add_node(self, var: Variable) -> "FactorGraph":
        if var in self._var_set:
            raise KeyError("The node has already existed in the graph")
        self._vars.append(var)
        self._var_set.add(var)
        return self

    def add_factor(self, factor) -> "FactorGraph":
        for v in factor.vars:
            if v not in self._var_set:
                raise KeyError("factor %s refers to a variable that is not in the graph: %s" % (factor, v.name))
        for v in factor.vars:
            self._by_var.setdefault(v, []).append(len(self._factors))
        self._factors.append(factor)
        return self

    @property
    def vars(self) -> List[Variable]:
        return self._vars

    @property
    def factors(self) -> List:
        return self._factors

    def get_adjacent_factors_from_node(self, key: Variable) -> List:
        return [f for f in self._factors if key in f.vars]

    def get_neighbors_in_factor_graph(self, key: Variable) -> Set[Variable]:
        return {v for f in self._factors if key in f.vars for v in f.vars if v != key}

    # ---- symbolic elimination --------------------------------------------------------------
    def eliminate_for_analysis(self, ordering: List[Variable]) -> Dict[Variable, Set[Variable]]:
        """Parents of every variable in the Bayes net obtained by eliminating in `ordering`
        (structure only; reference: eliminate_from_factor_graph_for_analysis, :72-95)."""
        adj = {v: set() for v in self._vars}
        for f in self._factors:
            vs = f.vars
            for a in vs:
                for b in vs:
                    if a != b:
                        adj[a].add(b)
        parents = {}
        for v in ordering:
            nb = set(adj[v])
            parents[v] = nb
            for a in nb:
                adj[a].discard(v)
                adj[a] |= (nb - {a})          # fill-in: the separator becomes a clique
            del adj[v]
        return parents

    def analyze_elimination_ordering(self, method: str = "natural", last_vars: List[Variable] = None) -> List[Variable]:
        if method == "natural":
            return sorted(self._vars)
        if method == "pose_first":
            return self.generate_pose_first_ordering(self._vars)
        raise ValueError("Unrecognized method for analyzing elimination order (ccolamd is dead code in the "
                         "reference as well, SURVEY.md Appendix B)")

    def get_bayes_tree(self, ordering: List[Variable] = None, method: str = "natural",
                       last_vars: List[Variable] = None) -> BayesTree:
        if ordering is None:
            ordering = self.analyze_elimination_ordering(method=method, last_vars=last_vars)
        parents = self.eliminate_for_analysis(ordering)
        tree = BayesTree(frontal=ordering[-1])
        tree.reverse_elimination_order = ordering[::-1]
        for frontal in ordering[:-1][::-1]:
            tree.add_node(frontal=frontal, parents=parents[frontal])
        return tree

    # ---- sub graphs used by the incremental solver ---------------------------------------------
    def get_sub_factor_graph_with_prior(self, variables: Set[Variable], sub_trees: List[BayesTree],
                                        clique_prior_dict: Dict[BayesTreeNode, object]) -> "FactorGraph":
        """Factors among `variables` that are not already summarised by a detached sub tree, plus one
        separator prior per detached sub tree (reference :204-228)."""
        sub = FactorGraph()
        for v in self._vars:
            if v in variables:
                sub.add_node(v)
        # (a factor among `variables` touches one of them: the candidates are the factors of those variables, in the graph's own order)
        by_var = self._by_var
        candidates = sorted({i for v in variables for i in by_var.get(v, ())})
        for i in candidates:
            f = self._factors[i]
            fv = set(f.vars)
            if fv.issubset(variables) and not any(fv.issubset(t.root.vars) for t in sub_trees):
                sub.add_factor(f)
        for t in sub_trees:
            sub.add_factor(clique_prior_dict[t.root])
        return sub

    def eliminate_clique_variables(self, clique: BayesTreeNode, new_factor) -> "FactorGraph":
        """Drop the clique's frontal variables and every factor inside the clique; add the clique's
        separator factor (reference :230-247)."""
        sub = FactorGraph()
        for v in self._vars:
            if v not in clique.frontal:
                sub.add_node(v)
        cv = clique.vars
        for f in self._factors:
            if not set(f.vars).issubset(cv):
                sub.add_factor(f)
        if new_factor is not None:
            sub.add_factor(new_factor)
        return sub

    def get_clique_factor_graph(self, clique: BayesTreeNode) -> "FactorGraph":
        sub = FactorGraph()
        cv = clique.vars
        for v in self._vars:
            if v in cv:
                sub.add_node(v)
        for f in self._factors:
            if set(f.vars).issubset(cv):
                sub.add_factor(f)
        return sub

    @staticmethod
    def generate_pose_first_ordering(nodes) -> List[Variable]:
        poses = [v for v in nodes if v.type != VariableType.Landmark]
        lmks = [v for v in nodes if v.type == VariableType.Landmark]
        return poses + lmks
