"""Random small shapes of the training / scoring step against the oracles (scripts/fuzz_step.py): widths that are
not the golden 40, lengths 1..50, 1..64 positives, 2..10 rows per positive, all encoders and sibling models.  The
seeds below include the cases that exposed real limits (groups of more than 8 rows in the attention backward, bpr on
wide embeddings, DIN with D > T)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, seed, kinds, precision="fp32"):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_step

    fuzz_step.PRECISION = precision
    rng = np.random.default_rng(seed)
    failures = []
    for i in range(n):
        desc, problems = fuzz_step.one_case(rng, i, kinds[i % len(kinds)])
        if problems:
            failures.append((desc, problems[:6]))
    assert not failures, failures


@pytest.mark.parametrize("precision", ["fp32", "fp32x3"])
def test_clsr_random_shapes(precision):
    """precision="fp32": every product at fp32 accuracy (the reference's arithmetic); "fp32x3": the two-piece split products,
    held to the same tolerances."""
    _run(16, 0, ["clsr"], precision)


def test_sibling_random_shapes():
    _run(30, 7, ["gru4rec", "din", "sli_rec", "a2svd", "dien"])


def test_model_api_random_batch_shapes(tmp_path):
    """fit / run_weighted_eval / predict with random batch sizes, negatives per positive and file sizes (ragged last
    batches, evaluation batches that cut groups apart): the product defaults (de-duplicated histories, compact feeds,
    launch plans, look-ahead staging) against the plain row-by-row configuration."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_model

    rng = np.random.default_rng(0)
    failures = []
    for i in range(12):
        desc, problems = fuzz_model.one_case(rng, i, str(tmp_path))
        if problems:
            failures.append((desc, problems[:6]))
    assert not failures, failures
