"""``OpenLoopPolicy`` -- the object the controller hands to ``forward_model.predict_n_steps`` (mpc.py:56-67) -- replayed
against the hand-out order recorded from the reference's own class (tests/golden/make_golden.py::
open_loop_policy_vectors): one trajectory at a time (``GroundTruthModel``), k rows at a time, all rows per call, and the
sub-policies ``ParallelGroundTruthModel`` cuts with ``get_parallel_policy_copy``.  No device needed."""
import os

import numpy as np
import pytest

from icem_amd.controllers import OpenLoopPolicy
from golden_util import GOLDEN

Z = np.load(os.path.join(GOLDEN, "open_loop_policy_vectors.npz"))
ERRORS = {"AttributeError": AttributeError, "IndexError": IndexError, "AssertionError": AssertionError}


@pytest.mark.parametrize("case", [str(c) for c in Z["cases"]])
def test_hand_out_order_and_exhaustion_match_the_reference(case):
    p, h, d, k = [int(v) for v in Z[case + "_cfg"]]
    seq = np.arange(p * h * d, dtype=np.float64).reshape(p, h, d)
    obs = np.zeros(4) if k == 1 else np.zeros((k, 4))
    want_ndim, want_rows, want_flat = Z[case + "_ndim"], Z[case + "_rows"], Z[case + "_flat"]
    err = ERRORS[str(Z[case + "_err"])]
    outs = []
    with pytest.raises(err):
        pol = OpenLoopPolicy(seq)
        for _ in range(p * h + 2):
            outs.append(np.array(pol.get_action(obs, None)))
    assert len(outs) == len(want_ndim)
    assert [o.ndim for o in outs] == list(want_ndim)
    assert [1 if o.ndim == 1 else o.shape[0] for o in outs] == list(want_rows)
    got = np.concatenate([o.reshape(-1) for o in outs]) if outs else np.zeros(0)
    np.testing.assert_array_equal(got, want_flat)


def test_sub_policies_of_parallel_workers():
    """gt_par_model.py:77-80: ``array_split`` chunks -> ``get_parallel_policy_copy(chunk)`` -> each walked row by row."""
    p, h, d, workers = [int(v) for v in Z["olp_chunks_cfg"]]
    seq = np.arange(p * h * d, dtype=np.float64).reshape(p, h, d)
    pol = OpenLoopPolicy(seq)
    flat = []
    for c in [c for c in np.array_split(range(p), workers) if len(c) > 0]:
        sub = pol.get_parallel_policy_copy(c)
        assert isinstance(sub, OpenLoopPolicy) and sub.action_sequences.shape == (len(c), h, d)
        for _ in range(len(c) * h):
            flat.append(np.array(sub.get_action(np.zeros(4), None)))
    np.testing.assert_array_equal(np.concatenate(flat), Z["olp_chunks_flat"])


def test_results_are_views_not_copies():
    seq = np.zeros((3, 2, 2))
    pol = OpenLoopPolicy(seq)
    a = pol.get_action(np.zeros(4), None)
    assert a.shape == (2,) and np.shares_memory(a, seq)
