"""
Host-side mirror of the spaces the five pysim envs use (P/spaces/base.py, box.py, polar.py, compound.py).

These describe an environment to its callers (policies, samplers, wrappers); the device kernels carry their own copy of
the bounds as derived constants.  Same names, argument meaning and error behaviour as the reference.
"""
from copy import deepcopy
from typing import Sequence

import numpy as np

from .exceptions import ShapeErr, TypeErr, ValueErr


class Space:
    bound_lo: np.ndarray
    bound_up: np.ndarray

    @property
    def bounds(self):
        return self.bound_lo, self.bound_up

    @property
    def bound_abs_up(self) -> np.ndarray:  # P/spaces/base.py:66-69
        return np.max(np.stack([np.abs(self.bound_lo), np.abs(self.bound_up)], axis=0), axis=0)

    @property
    def ele_dim(self) -> int:
        return self.bound_lo.shape[0]

    @property
    def flat_dim(self) -> int:
        return int(np.prod(self.shape))

    def create_mask(self, *idcs) -> np.ndarray:
        """boolean mask with True at the given indices or labels (P/spaces/base.py:103-136)"""
        mask = np.zeros(self.shape, dtype=np.bool_)
        if len(idcs) == 1 and hasattr(idcs[0], "__iter__") and not isinstance(idcs[0], str):
            idcs = idcs[0]
        for idx in idcs:
            if isinstance(idx, str):
                hits = [i for i, lab in np.ndenumerate(self.labels) if lab == idx]
                if not hits:
                    raise ValueErr(msg=f"Label {idx} not found in {self}")
                idx = hits[0]
            if np.all(mask[idx] == 1):
                raise ValueErr(msg=f"Duplicate index {idx}")
            mask[idx] = 1
        return mask

    def copy(self):
        return deepcopy(self)

    def __eq__(self, other):
        if type(other) is not type(self):
            return False
        return bool(np.all([np.array_equal(s, o) for s, o in zip(self._members(), other._members())]))


class BoxSpace(Space):
    """P/spaces/box.py:38-232"""

    def __init__(self, bound_lo, bound_up, shape=None, labels: Sequence[str] = None):
        if shape is not None:
            self.bound_lo = np.ones(shape) * bound_lo
            self.bound_up = np.ones(shape) * bound_up
        else:
            try:
                self.bound_lo = np.atleast_1d(np.array(bound_lo, dtype=np.float64))
                self.bound_up = np.atleast_1d(np.array(bound_up, dtype=np.float64))
            except (TypeError, ValueError):
                raise TypeErr(given=bound_lo, expected_type=[float, list, np.ndarray])
            if self.bound_lo.shape != self.bound_up.shape:
                raise ShapeErr(given=self.bound_lo, expected_match=self.bound_up)
        if labels is not None:
            labels_np = np.array(labels, dtype=object)
            if not labels_np.shape == self.shape:
                raise ShapeErr(given=labels_np, expected_match=self)
            self._labels = labels_np
        else:
            self._labels = np.empty(self.shape, dtype=object)
            self._labels.fill(None)

    def _members(self):
        return self.bound_lo, self.bound_up, self._labels

    @property
    def shape(self) -> tuple:
        return self.bound_lo.shape

    @property
    def labels(self) -> np.ndarray:
        return self._labels

    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        """Inclusive bounds; raises ShapeErr on a shape mismatch and ValueErr on NaN (box.py:138-167, quirk Q9)."""
        cand = np.asarray(cand)
        if not cand.shape == self.shape:
            raise ShapeErr(given=cand, expected_match=self)
        if np.isnan(cand).any():
            raise ValueErr(msg="At least one value is NaN!")
        ok = bool(np.all((cand >= self.bound_lo) & (cand <= self.bound_up)))
        if not ok and verbose:
            print(f"lower {self.bound_lo}\ncand  {cand}\nupper {self.bound_up}")
        return ok

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        bl, bu = self.bound_lo.copy(), self.bound_up.copy()
        bl[bl == -np.inf] = -concrete_inf
        bu[bu == np.inf] = concrete_inf
        return np.random.uniform(bl, bu)  # NumPy global RNG, as the reference (Q14)

    def project_to(self, ele: np.ndarray) -> np.ndarray:
        """Returns the SAME object when inside, a clipped copy otherwise (box.py:180-184)."""
        if not self.contains(ele):
            return np.clip(ele, self.bound_lo, self.bound_up)
        return ele

    def subspace(self, idcs):
        idcs = np.atleast_1d(idcs)
        return BoxSpace(self.bound_lo[idcs], self.bound_up[idcs], labels=self._labels[idcs])

    @staticmethod
    def cat(spaces):
        spaces = [s for s in spaces if s is not None]
        lo, up, lab = [], [], []
        for s in spaces:
            if not isinstance(s, BoxSpace):
                raise TypeErr(given=s, expected_type=BoxSpace)
            lo.extend(s.bound_lo)
            up.extend(s.bound_up)
            lab.extend(s.labels)
        return BoxSpace(lo, up, labels=lab)


class Polar2DPosVelSpace(BoxSpace):
    """[r, phi, x_dot, y_dot] box sampled and returned in cartesian coordinates (P/spaces/polar.py:80-127)"""

    def __init__(self, bound_lo, bound_up, shape=None, labels=None):
        super().__init__(bound_lo, bound_up, shape, labels=labels)
        assert self.bound_lo.size == self.bound_up.size == 4

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        sample = super().sample_uniform()
        sample[:2] = np.array([sample[0] * np.cos(sample[1]), sample[0] * np.sin(sample[1])])
        return sample

    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        cand = np.asarray(cand, dtype=np.float64)
        assert cand.size == 4
        x, y = cand[0], cand[1]
        polar = np.array([np.sqrt(x ** 2 + y ** 2), np.arctan2(y, x), cand[2], cand[3]])
        return super().contains(polar, verbose=verbose)


class CompoundSpace(Space):
    """Union of sub-spaces; sampling first picks one of them (P/spaces/compound.py:38-87)"""

    def __init__(self, spaces: Sequence[Space]):
        self._spaces = deepcopy(list(spaces))

    @property
    def shape(self):
        return self._spaces[0].shape

    @property
    def flat_dim(self) -> int:
        return sum(s.flat_dim for s in self._spaces)

    def _members(self):
        return tuple(self._spaces)

    def subspace(self, idcs):
        return self._spaces[idcs]

    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        return any(s.contains(cand) for s in self._spaces)

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        idx = np.random.randint(len(self._spaces))
        return self._spaces[idx].sample_uniform()


class SingularStateSpace(BoxSpace):
    """always returns the same initial state (P/spaces/singular.py:37-52)"""

    def __init__(self, fixed_state: np.ndarray, labels=None):
        super().__init__(fixed_state, fixed_state, labels=labels)
        self._fixed_state = np.asarray(fixed_state, dtype=np.float64)

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        return self._fixed_state.copy()


class DiscreteSpace(Space):
    """finite set of elements, one per row (P/spaces/discrete.py:38-131)"""

    def __init__(self, eles, labels=None):
        eles = np.asarray(eles, dtype=np.float64)
        self.eles = np.atleast_2d(eles if eles.ndim == 2 else eles.reshape(-1, 1))
        self.bound_lo = np.min(self.eles, axis=0)
        self.bound_up = np.max(self.eles, axis=0)
        self._labels = np.array(labels, dtype=object) if labels is not None else np.full(self.shape, None, dtype=object)

    def _members(self):
        return self.eles, self._labels

    @property
    def shape(self):
        return self.bound_lo.shape

    @property
    def labels(self):
        return self._labels

    @property
    def num_ele(self) -> int:
        return self.eles.shape[0]

    @staticmethod
    def cat(spaces):
        """the union of the element sets of several DiscreteSpaces, in order (P/spaces/discrete.py:133-160)"""
        spaces = [s for s in spaces if s is not None]
        for sp in spaces:
            if not isinstance(sp, DiscreteSpace):
                raise TypeErr(given=sp, expected_type=DiscreteSpace)
        return DiscreteSpace(np.concatenate([sp.eles for sp in spaces], axis=0))

    @property
    def flat_dim(self) -> int:
        return self.eles.shape[1]

    @property
    def bound_abs_up(self) -> np.ndarray:
        return np.max(np.abs(self.eles), axis=0)

    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        cand = np.asarray(cand)
        if not cand.shape == self.shape:
            raise ShapeErr(given=cand, expected_match=self)
        if np.isnan(cand).any():
            raise ValueErr(msg="At least one value is NaN!")
        return bool(np.any(np.isclose(self.eles, cand.astype(self.eles.dtype))))

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        idx = np.random.randint(self.num_ele, size=1)
        return self.eles[idx, :].flatten()

    def project_to(self, ele: np.ndarray) -> np.ndarray:
        if not self.contains(ele):
            return self.eles[np.argmin(np.abs(ele - self.eles)), :]
        return ele


class EnvSpec:
    """P/utils/data_types.py:45-50"""

    def __init__(self, obs_space, act_space, state_space=None):
        self.obs_space, self.act_space, self.state_space = obs_space, act_space, state_space

    def __iter__(self):
        return iter((self.obs_space, self.act_space, self.state_space))
