"""slam.Variables — variable types with the reference's constructor signatures and properties
(reference: src/slam/Variables.py:13-178).  Host-side bookkeeping only; what the hot path needs
from a variable is `dim`, `name` and `circular_dim_list` (which columns are angles)."""
from enum import Enum
from typing import Hashable, List, Set


class VariableType(Enum):
    Pose = "Pose"
    Landmark = "Landmark"
    Measurement = "Measurement"


class Variable(object):
    def __init__(self, name: Hashable, dim: int, variable_type: VariableType = VariableType.Pose,
                 rotational_dims: Set[int] = None) -> None:
        if dim <= 0:
            raise ValueError("Dimensionality must be positive")
        self._type = variable_type
        self._dim = dim
        self._name = name
        if not rotational_dims:
            self._rotational_dims = {}
        elif not 0 <= min(rotational_dims) <= max(rotational_dims) < dim:
            raise ValueError("rotational_dims is incorrect")
        else:
            self._rotational_dims = rotational_dims

    @classmethod
    def construct_from_text(cls, line: str) -> "Variable":
        """Parse a `.fg` variable line: `Variable <Pose|Landmark> <SE2|R2|...> <name>`."""
        tok = line.strip().split()
        cls_type = {"SE2": SE2Variable, "R2": R2Variable, "R1": R1Variable, "Bearing2D": Bearing2DVariable}[tok[2]]
        return cls_type(name=tok[3], variable_type=VariableType[tok[1]])

    @property
    def dim(self) -> int:
        return self._dim

    @property
    def name(self) -> Hashable:
        return self._name

    @property
    def type(self) -> VariableType:
        return self._type

    @property
    def translational_dim(self) -> int:
        return self._dim - len(self._rotational_dims)

    @property
    def rotational_dim(self) -> int:
        return len(self._rotational_dims)

    @property
    def circular_dim_list(self) -> List[bool]:
        """True for periodic (angle) dimensions; order is translation first, e.g. x y theta."""
        return [i in self._rotational_dims for i in range(self.dim)]

    @property
    def t_dim_indices(self):
        return list(range(self.translational_dim))

    @property
    def R_dim_indices(self):
        return list(range(self.dim))[self.translational_dim:]

    @staticmethod
    def file2vars(order_file: str, pose_space: str = "SE2") -> List["Variable"]:
        """Variables named in an ordering file (`step{i}_ordering` of a run folder: whitespace-separated names): names
        starting with 'L' are R2 landmarks, the others poses in `pose_space` ("SE2" or "R2"); reference Variables.py:142-154."""
        with open(order_file) as f:
            names = f.read().split()
        make_pose = {"SE2": SE2Variable, "R2": R2Variable}.get(pose_space)
        out = []
        for name in names:
            if name.startswith("L"):
                out.append(R2Variable(name=name, variable_type=VariableType.Landmark))
            elif make_pose is not None:                       # (an unknown pose space yields landmarks only, as the reference)
                out.append(make_pose(name=name, variable_type=VariableType.Pose))
        return out

    def __copy__(self) -> "Variable":
        return Variable(name=self._name, dim=self._dim)

    def __str__(self) -> str:
        return " ".join(["Variable", self.type.value, self.__class__.__name__.replace("Variable", ""), str(self.name)])

    __repr__ = __str__

    def __hash__(self) -> int:
        return hash(self._name)

    def __eq__(self, other) -> bool:
        return self._name == other._name

    def __ne__(self, other) -> bool:
        return self._name != other._name

    def __le__(self, other) -> bool:
        return self._name <= other._name

    def __lt__(self, other) -> bool:
        return self._name < other._name

    def __ge__(self, other) -> bool:
        return self._name >= other._name

    def __gt__(self, other) -> bool:
        return self._name > other._name


class R2Variable(Variable):
    def __init__(self, name: Hashable, variable_type: VariableType = VariableType.Pose) -> None:
        super().__init__(name=name, dim=2, variable_type=variable_type, rotational_dims=None)


class R1Variable(Variable):
    def __init__(self, name: Hashable, variable_type: VariableType = VariableType.Pose) -> None:
        super().__init__(name=name, dim=1, variable_type=variable_type, rotational_dims=None)


class Bearing2DVariable(Variable):
    def __init__(self, name: Hashable, variable_type: VariableType = VariableType.Pose) -> None:
        super().__init__(name=name, dim=1, variable_type=variable_type, rotational_dims={0})


class SE2Variable(Variable):
    def __init__(self, name: Hashable, variable_type: VariableType = VariableType.Pose) -> None:
        super().__init__(name=name, dim=3, variable_type=variable_type, rotational_dims={2})
