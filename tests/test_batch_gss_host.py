"""Batched golden-section search (host logic): every problem's evaluation sequence must equal the
scalar reference-shaped search run on its own."""
import numpy as np

from ffsubsync_amd.batch_gss import gss_batch
from ffsubsync_amd.golden_section_search import gss


def test_lockstep_sequences_equal_scalar_searches():
    rng = np.random.RandomState(1)
    centres = rng.uniform(0.9, 1.1, size=9)
    # bumpy unimodal-ish objectives with plateaus, to exercise both branches and exact ties
    fs = [lambda x, last, c=c: np.round(abs(x - c), 3) + 0.01 * np.sin(40 * x) for c in centres]

    def evaluate(xs, last):
        return np.array([f(x, last) for f, x in zip(fs, xs)])

    lo, hi, trace = gss_batch(evaluate, len(fs))
    for i, f in enumerate(fs):
        seen = []

        def g(x, last, f=f):
            seen.append((x, last))
            return f(x, last)

        a, b = gss(g, 0.9, 1.1)
        assert [repr(x) for x, _ in seen] == [repr(float(t[0][i])) for t in trace]
        assert [l for _, l in seen] == [t[1] for t in trace]
        assert (repr(a), repr(b)) == (repr(float(lo[i])), repr(float(hi[i])))
    assert [t[1] for t in trace].count(True) == 1 and trace[-1][1]


def test_degenerate_interval_and_empty_batch():
    lo, hi, trace = gss_batch(lambda x, last: x, 3, 1.0, 1.00001, 1e-4)
    assert trace == [] and np.all(lo == 1.0)
    lo, hi, trace = gss_batch(lambda x, last: x, 0)
    assert trace == [] and lo.size == 0
