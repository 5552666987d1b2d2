"""TrajectoryBatch: the flat-array stand-in for RolloutBuffer (icem/misc/rolloutbuffer.py:112-123, 156-172, 277) keeps the
sequence interface reference-style consumers use -- len / iteration / indexing / as_array / extend / append / flat /
is_empty -- with the reference's error for trajectories of unequal length."""
import numpy as np
import pytest

from icem_amd.models import TrajectoryBatch


def _batch(n, h=5, o=3, d=2, seed=0):
    rs = np.random.RandomState(seed)
    return TrajectoryBatch(observations=rs.randn(n, h, o), next_observations=rs.randn(n, h, o), actions=rs.randn(n, h, d),
                           rewards=rs.randn(n, h, 1))


def test_empty_extend_append_flat():
    b = TrajectoryBatch()
    assert b.is_empty and len(b) == 0 and not b
    a = _batch(4)
    b.extend(a)
    assert not b.is_empty and len(b) == 4
    np.testing.assert_array_equal(b.as_array("actions"), a.as_array("actions"))
    b.extend(_batch(3, seed=1))
    assert len(b) == 7
    b.append(a[2])                       # a per-trajectory dict view, as iteration hands them out
    assert len(b) == 8
    np.testing.assert_array_equal(b[7]["observations"], a[2]["observations"])
    b.extend(r for r in _batch(2, seed=2))   # any iterable of views
    assert len(b) == 10
    flat = b.flat
    assert flat["observations"].shape == (50, 3) and flat["actions"].shape == (50, 2)
    np.testing.assert_array_equal(flat["rewards"][:5], a.as_array("rewards")[0])
    np.testing.assert_array_equal(b["actions"], flat["actions"])    # the string index is the flat view too
    b.extend(TrajectoryBatch())          # nothing to add
    b.extend([])
    assert len(b) == 10


def test_unequal_lengths_raise_the_reference_error():
    b = _batch(2, h=5)
    with pytest.raises(TypeError, match="unequal length"):
        b.extend(_batch(2, h=6))
    with pytest.raises(TypeError):
        b.extend([{"observations": np.zeros((5, 3))}, {"observations": np.zeros((4, 3))}])
    with pytest.raises(TypeError, match="unequal length"):
        TrajectoryBatch(observations=np.zeros((2, 5, 3)), actions=np.zeros((3, 5, 2)))
