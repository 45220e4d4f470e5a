"""slam.BayesTree — Bayes tree bookkeeping around the flow hot path
(reference: src/slam/BayesTree.py:6-384).

Same class / method names as the reference; restated with ordered containers so that clique
order, attachment points and therefore training order are deterministic (the reference keeps
children in a `set`, whose iteration order depends on PYTHONHASHSEED).
Pure host-side Python: a few dozen set operations per incremental update.
"""
from typing import Iterable, List, Set, Tuple, Union

from slam.Variables import Variable


class BayesTreeNode(object):
    def __init__(self, frontal: Union[Variable, Set[Variable]], separator: Set[Variable] = None,
                 children: Iterable["BayesTreeNode"] = None, parent: "BayesTreeNode" = None) -> None:
        if isinstance(frontal, Variable):
            self.frontal = {frontal}
        elif isinstance(frontal, set):
            self.frontal = frontal
        else:
            raise ValueError("The frontal must be either the set of all frontal variables, or a frontal variable")
        self.separator = set(separator) if separator else set()
        self._hash = None               # cached __hash__ (the Bayes-tree bookkeeping hashes a clique ~1000 times per update)
        self.parent = parent
        self.children: List["BayesTreeNode"] = list(children) if children else []

    # ---- structure ---------------------------------------------------------------------------
    def append_child(self, child: "BayesTreeNode") -> "BayesTreeNode":
        if not any(c is child for c in self.children):
            self.children.append(child)
        child.parent = self
        return self

    def create_child(self, frontal: Variable, separator: Set[Variable] = None) -> "BayesTreeNode":
        child = BayesTreeNode(frontal=frontal, separator=separator)
        self.append_child(child)
        return child

    def add_frontal(self, frontal: Variable) -> "BayesTreeNode":
        self.frontal.add(frontal)
        self._hash = None
        return self

    def remove_child(self, child: "BayesTreeNode") -> "BayesTreeNode":
        self.children = [c for c in self.children if c is not child]
        child.parent = None
        return self

    # ---- properties --------------------------------------------------------------------------
    @property
    def is_leaf(self) -> bool:
        return len(self.children) == 0

    @property
    def is_root(self) -> bool:
        return self.parent is None

    @property
    def vars(self) -> Set[Variable]:
        return self.frontal | self.separator

    @property
    def num_vars(self) -> int:
        return len(self.frontal) + len(self.separator)

    @property
    def dim(self) -> int:
        return sum(v.dim for v in self.vars)

    @property
    def separator_dim(self) -> int:
        return sum(v.dim for v in self.separator)

    @property
    def frontal_dim(self) -> int:
        return sum(v.dim for v in self.frontal)

    def copy_without_parents_children(self) -> "BayesTreeNode":
        return BayesTreeNode(frontal=set(self.frontal), separator=set(self.separator))

    def __str__(self) -> str:
        names = lambda vs: "{" + ", ".join(sorted(str(v.name) for v in vs)) + "}"   # noqa: E731
        return "BayesTreeNode{frontal: %s, separator: %s}" % (names(self.frontal), names(self.separator))

    __repr__ = __str__

    def __eq__(self, other) -> bool:
        """Two cliques are equal iff they have the same frontal and separator variables."""
        return isinstance(other, BayesTreeNode) and self.frontal == other.frontal and self.separator == other.separator

    def __hash__(self) -> int:
        # order-independent over the two variable sets (consistent with __eq__); no sorting, no strings.  The sets are only
        # ever changed through add_frontal (which drops the cached value) -- and never while the clique is a dict key.
        h = self._hash
        if h is None:
            h = self._hash = hash(frozenset(self.frontal)) ^ (hash(frozenset(self.separator)) * 1000003)
        return h


class BayesTree(object):
    def __init__(self, root_clique: BayesTreeNode = None, frontal: Variable = None) -> None:
        if root_clique is not None:
            self.root = root_clique
            for child in root_clique.children:
                child.parent = root_clique
        elif frontal is not None:
            self.root = BayesTreeNode(frontal=frontal)
        else:
            raise ValueError("Either the root clique or a root frontal variable needs to be specified")
        self.reverse_elimination_order = None

    # ---- traversal (deterministic: breadth first, children in insertion order) ------------------
    def clique_ordering(self) -> List[BayesTreeNode]:
        order, queue = [], [self.root]
        while queue:
            c = queue.pop(0)
            order.append(c)
            queue.extend(c.children)
        return order

    @property
    def clique_nodes(self) -> Set[BayesTreeNode]:
        return set(self.clique_ordering())

    @property
    def leaves(self) -> Set[BayesTreeNode]:
        return {c for c in self.clique_ordering() if c.is_leaf}

    @property
    def frontal_vars(self) -> Set[Variable]:
        return set().union(*[c.frontal for c in self.clique_ordering()])

    # ---- construction ------------------------------------------------------------------------
    def add_node(self, frontal: Variable, parents: Set[Variable] = None) -> "BayesTree":
        """Insert the conditional p(frontal | parents) (variables are added in REVERSE elimination
        order).  It goes below the clique that holds its earliest-eliminated parent as a frontal
        variable (that clique contains all parents of a chordal elimination); if the parents are
        exactly that clique's variables the frontal joins the clique instead."""
        parents = set(parents) if parents else set()
        cliques = self.clique_ordering()
        target = None
        if parents and self.reverse_elimination_order is not None:
            first_parent = max(parents, key=lambda v: self.reverse_elimination_order.index(v))
            for c in cliques:
                if first_parent in c.frontal and parents.issubset(c.vars):
                    target = c
                    break
        if target is None:
            for c in cliques:
                if parents.issubset(c.vars):
                    target = c
                    break
        if target is None:
            raise ValueError("no clique contains the parents of %s" % frontal.name)
        if len(parents) == target.num_vars:
            target.add_frontal(frontal)
        else:
            target.create_child(frontal, parents)
        return self

    def append_clique(self, clique: BayesTreeNode, parent_clique: BayesTreeNode) -> "BayesTree":
        parent_clique.append_child(clique)
        return self

    def append_child_bayes_tree(self, child_tree: "BayesTree") -> "BayesTree":
        for attach_point in self.clique_ordering():
            if child_tree.root.separator.issubset(attach_point.vars):
                attach_point.append_child(child_tree.root)
                return self
        raise ValueError("no attachment point for sub tree rooted at %s" % child_tree.root)

    def append_child_bayes_trees(self, child_trees: Iterable["BayesTree"]) -> "BayesTree":
        for t in child_trees:
            self.append_child_bayes_tree(t)
        return self

    def __copy__(self) -> "BayesTree":
        new_tree = BayesTree(root_clique=self.root.copy_without_parents_children())
        new_tree.reverse_elimination_order = list(self.reverse_elimination_order) \
            if self.reverse_elimination_order else []
        stack = [(self.root, new_tree.root)]
        while stack:
            old, new = stack.pop()
            for oc in old.children:
                nc = oc.copy_without_parents_children()
                new.append_child(nc)
                stack.append((oc, nc))
        return new_tree

    def __str__(self) -> str:
        return "BayesTree{" + ", ".join(str(c) for c in self.clique_ordering()) + "}"

    # ---- incremental update support -----------------------------------------------------------
    def get_affected_vars_and_partial_bayes_trees(self, vars: Set[Variable], detach: bool = False
                                                  ) -> Tuple[Set[Variable], List["BayesTree"]]:
        """Cliques holding `vars` as frontal variables and all their ancestors are affected; every
        maximal unaffected subtree hanging off an affected clique is returned as a detached tree: a deep copy
        (as in the reference), or -- `detach=True`, for a caller that discards this tree afterwards -- the very
        same nodes cut off from their parent (no O(tree) copy per update).
        -> (frontal variables of affected cliques, detached sub trees)."""
        self.detached_edges = []
        frontal_of = {}
        for c in self.clique_ordering():
            for v in c.frontal:
                frontal_of[v] = c
        affected = []
        for v in sorted(set(vars) & set(frontal_of), key=lambda u: str(u.name)):
            c = frontal_of[v]
            while c is not None and not any(c is a for a in affected):
                affected.append(c)
                c = c.parent
        sub_trees = []
        for c in self.clique_ordering():
            if not any(c is a for a in affected):
                continue
            for child in list(c.children):
                if not any(child is a for a in affected):
                    if detach:
                        self.detached_edges.append((c, child, [x for x in c.children]))   # enough to undo the cut (reattach)
                        c.children = [x for x in c.children if x is not child]
                        child.parent = None
                        sub_trees.append(BayesTree(root_clique=child))
                    else:
                        sub_trees.append(BayesTree(root_clique=_deep_copy_subtree(child)))
        affected_vars = set().union(*[c.frontal for c in affected]) if affected else set()
        return affected_vars, sub_trees

    def reattach_detached(self) -> None:
        """Undo the cuts of the last `get_affected_vars_and_partial_bayes_trees(detach=True)`: the caller's update failed
        after the unaffected subtrees were moved out, and the tree has to be whole again for a retry."""
        for parent, child, children_before in reversed(getattr(self, "detached_edges", [])):
            parent.children = children_before
            child.parent = parent
        self.detached_edges = []

    def clique_variable_pattern(self, clique: BayesTreeNode) -> List[Variable]:
        """[separator variables, frontal variables], each in reverse elimination order."""
        key = lambda v: self.reverse_elimination_order.index(v)   # noqa: E731
        return sorted(clique.separator, key=key) + sorted(clique.frontal, key=key)


def _deep_copy_subtree(node: BayesTreeNode) -> BayesTreeNode:
    new = node.copy_without_parents_children()
    for ch in node.children:
        new.append_child(_deep_copy_subtree(ch))
    new.parent = None
    return new
