"""The scenarios of the reference's own space tests (Pyrado/tests/test_spaces.py:39-189) on this package's spaces:
same fixtures (box shapes, discrete element sets), same checks, same known answers (torus samples).  CPU only."""
import math

import numpy as np
import pytest

from simurlacra_amd.spaces import BoxSpace, DiscreteSpace, Polar2DPosVelSpace

BOXES = [(-np.ones((7,)), np.ones((7,))), (-np.ones((7, 1)), np.ones((7, 1))),
         (np.array([5, -math.pi / 2, -math.pi]), np.array([5, math.pi / 2, math.pi]))]
DISCRETE = [np.array([1]), np.array([[1]]), np.array([1, 2, 3], dtype=np.int32), np.array([-2, -1, 0, 1, 2], dtype=np.int64),
            np.array([4, -3, 5, 0, 1, 2, 6, -7], dtype=np.int32), np.array([4.0, -3, 5, 0, 1, 2, 6, -7], dtype=np.float64)]


@pytest.fixture(params=BOXES, ids=["box_flatdim", "box", "half_sphere"])
def bs(request):
    return BoxSpace(request.param[0], request.param[1])


@pytest.fixture(params=DISCRETE, ids=["scalar1dim", "scalar2dim", "pos", "pos_neg", "prandom", "prandom_float"])
def ds(request):
    return DiscreteSpace(request.param)


def test_box_sample_contains_copy_project(bs):
    np.random.seed(0)
    for _ in range(10):
        assert bs.contains(bs.sample_uniform())
    assert not BoxSpace([-1, -2, -3], [1, 2, 3]).contains(np.array([-4, 0, 4]), verbose=True)
    twin = bs.copy()
    assert twin is not bs
    bs.bound_lo *= -3
    assert np.all(bs.bound_lo != twin.bound_lo)
    twin.bound_up *= 5
    assert np.all(bs.bound_up != twin.bound_up)
    fresh = BoxSpace(twin.bound_lo / 1, twin.bound_up / 5)
    for _ in range(100):  # inside w.p. 1/5, outside w.p. 4/5
        assert fresh.contains(fresh.project_to(fresh.sample_uniform() * 5.0))
    assert fresh.flat_dim == int(np.prod(fresh.shape)) and str(fresh)


@pytest.mark.parametrize("idcs", [[0, 1, 2], [0, 2]], ids=["3_wo_gap", "2_w_gap"])
def test_box_subspace(bs, idcs):
    sub = bs.subspace(idcs)
    if len(bs.shape) == 1:
        assert sub.flat_dim == len(idcs)
        np.testing.assert_equal(sub.bound_lo, bs.bound_lo[idcs])
        np.testing.assert_equal(sub.bound_up, bs.bound_up[idcs])
        np.testing.assert_equal(sub.labels, bs.labels[idcs])
    else:
        assert sub.flat_dim == len(idcs) * bs.shape[1]
        np.testing.assert_equal(sub.bound_lo, bs.bound_lo[idcs, :])
        np.testing.assert_equal(sub.bound_up, bs.bound_up[idcs, :])
        np.testing.assert_equal(sub.labels, bs.labels[idcs, :])


@pytest.mark.parametrize("parts", [[BoxSpace([-1, -2, -3], [1, 2, 3]), BoxSpace([-11, -22, -33], [11, 22, 33])],
                                   [BoxSpace([-1], [1]), BoxSpace([-22, 33], [22, 33])]], ids=["identical", "different"])
def test_box_cat(parts):
    cat = BoxSpace.cat(parts)
    assert isinstance(cat, BoxSpace) and cat.flat_dim == sum(p.flat_dim for p in parts)


@pytest.mark.parametrize("parts", [[DiscreteSpace([-1, -2, -3]), DiscreteSpace([11, 22, 33])],
                                   [DiscreteSpace([-1]), DiscreteSpace([22, 33])]], ids=["identical", "different"])
def test_discrete_cat(parts):
    cat = DiscreteSpace.cat(parts)
    assert isinstance(cat, DiscreteSpace) and cat.num_ele == sum(p.num_ele for p in parts)


def test_discrete_sample_contains_copy_project(ds):
    np.random.seed(0)
    for _ in range(10):
        assert ds.contains(ds.sample_uniform())
    twin = ds.copy()
    assert twin is not ds
    for _ in range(100):
        assert twin.contains(twin.project_to(twin.sample_uniform() * 5.0))
    assert str(twin)


def test_polar_space_known_samples():
    flat = Polar2DPosVelSpace(np.array([1, 0, -0.1, -0.1]), np.array([1, 0, 0.1, 0.1])).sample_uniform()
    assert flat[0] == 1 and flat[1] == 0  # r = 1 at 0 deg
    up = Polar2DPosVelSpace(np.array([1, np.pi / 2, 0, 0]), np.array([1, np.pi / 2, 0, 0])).sample_uniform()
    assert np.all(np.isclose(up, np.array([0, 1, 0, 0])))  # r = 1 at 90 deg
