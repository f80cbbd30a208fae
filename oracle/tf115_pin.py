"""Shared pieces of the TF-1.15 pin harness  --  TEST INFRASTRUCTURE ONLY (numpy only: importable under Python 3.6/3.7
next to TensorFlow 1.15 as well as here).

``scripts/capture_tf115.py`` (run by whoever has a TensorFlow-1.15 environment) builds the REFERENCE's ``CLSRModel`` on
the committed synthetic slice (tests/golden/data), loads the deterministic weight set F2 below, feeds the committed
reference-captured batch (tests/golden/iterator_train_sa.npz) and dumps what the reference computes into
``tests/golden/tf115_clsr_step.npz``.  ``tests/test_tf115_pin.py`` compares the oracle (CPU) and the HIP path (GPU)
against that file when it exists and reports "parity unpinned" when it does not.  Only the .npz travels; nothing of
the reference does.

Round 3: the pin file is SELF-CONTAINED -- it carries the weights the TF run actually used (``before/<name>`` for every
global variable right after the assignment) next to ``meta/variables`` (TF's own ``v.op.name`` list) and
``meta/no_grad`` (trainables whose ``tf.gradients`` entry is None).  The tests load the weights FROM THE FILE and
compare the two variable inventories first (``check_inventory``: any name present on one side only fails with both
lists printed), so a naming difference can neither be skipped silently nor masquerade as an arithmetic mismatch.

F2: every trainable variable is filled from ``numpy.random.RandomState(crc32(name) ^ seed)`` (the legacy generator:
its ``randn`` stream is frozen across numpy versions), scaled by the variable's role -- so that the TF script and
the tests build bit-identical float32 weights from nothing but (name, shape).
"""
import zlib

import numpy as np

PIN_FILE = "tf115_clsr_step.npz"
SEED = 20220425
#: hparams overrides of the pinned run == tests/conftest.py::golden_hparams (the capture script passes the same)
HPARAMS = dict(max_seq_length=10, batch_size=64, train_num_ngs=4, time_unit="s", contrastive_loss="triplet",
               contrastive_length_threshold=5, is_clip_norm=1, embed_l2=1e-6, layer_l2=1e-6,
               discrepancy_loss_weight=0.01, contrastive_loss_weight=0.1, epochs=2, EARLY_STOP=5, show_step=1000,
               save_model=False, write_tfevents=False)
#: what each group of keys in the pin file confirms about TF 1.15 (oracle/README.md repeats this list)
CONFIRMS = {
    "logit, alpha": "graph wiring of clsr.py:137-277 incl. the time_to_now[:, -1] quirk; GRUCell gate order r,u and "
                    "h' = u*h + (1-u)*c; dynamic_rnn zero output / state copy-through past sequence_length; Time4LSTM "
                    "gate formulas; non-fused batch_normalization over all-but-last axes with biased variance; softmax "
                    "padding constant",
    "loss/*": "group-softmax data loss, triplet contrastive terms and their masks, discrepancy over UNIQUE users, "
              "L2 over unique involved rows + every non-embedding trainable",
    "grad/*": "tf.gradients of the total loss (IndexedSlices densified by segment sum)",
    "slices_norm/*": "tf.clip_by_norm on IndexedSlices takes the norm of the concatenated, un-deduplicated slices",
    "after/*": "AdamOptimizer step 1: lr_t, dense apply for dense variables, sparse apply == dedup + whole-table decay "
               "(every row moves on step 1 only where it has gradient); moving_mean / moving_variance update with "
               "momentum 0.95 (UPDATE_OPS run with the train op)",
    "eval_pred": "inference-mode batch_normalization (moving statistics) and sigmoid(logit)",
}


def f2_value(name, shape, seed=SEED):
    """Deterministic float32 value of trainable variable ``name`` (TF name without the ':0')."""
    rs = np.random.RandomState((zlib.crc32(name.encode("utf-8")) ^ seed) & 0xFFFFFFFF)
    x = rs.randn(*shape)
    leaf = name.rsplit("/", 1)[-1]
    if name.startswith("sequential/embedding/"):
        v = 0.08 * x
    elif leaf == "gamma":
        v = 1.0 + 0.1 * x
    elif leaf == "beta" or leaf.startswith("b_nn_") or "bias" in leaf:
        v = 0.1 * x + (1.0 if name.endswith("gates/bias") else 0.0)
    elif len(shape) == 1:
        v = 0.3 * x                                   # Time4LSTM's per-feature time input weights
    else:
        v = x * np.sqrt(2.0 / (shape[0] + shape[-1]))
    return v.astype(np.float32)


def f2_weights(named_shapes, seed=SEED):
    return {name: f2_value(name, tuple(shape), seed) for name, shape in named_shapes}


def feed_arrays(npz, batch=0):
    """The committed reference-captured batch as {iterator attribute name: array}."""
    pre = "b%d_" % batch
    return {k[len(pre):]: npz[k] for k in npz.files if k.startswith(pre)}


def ref_names(ref, prefix):
    """Names stored under ``prefix`` (e.g. "grad/") in a loaded pin file."""
    return sorted(k[len(prefix):] for k in ref.files if k.startswith(prefix))


def check_inventory(theirs, ours, what):
    """Fail loudly -- with BOTH name lists -- when the reference graph and the restatement disagree on a variable
    name.  (A silent skip of the odd one out is how a renamed variable would otherwise go unnoticed.)"""
    theirs, ours = sorted(set(theirs)), sorted(set(ours))
    only_t = [n for n in theirs if n not in set(ours)]
    only_o = [n for n in ours if n not in set(theirs)]
    if only_t or only_o:
        raise AssertionError(
            "%s: the variable inventories differ.\n  only in the TensorFlow reference (%d): %s\n  only in this repo (%d): %s"
            "\n  -- all reference names: %s\n  -- all names of this repo: %s" % (
                what, len(only_t), only_t, len(only_o), only_o, theirs, ours))


def require_keys(ref, prefix, names, what, allow_missing=()):
    """Every name must have ``prefix + name`` in the pin file (no silent skips), except ``allow_missing``."""
    have = set(ref.files)
    missing = [n for n in names if prefix + n not in have and n not in set(allow_missing)]
    if missing:
        raise AssertionError("%s: the pin file has no '%s<name>' entry for %s\n  -- entries it has: %s"
                             % (what, prefix, missing, ref_names(ref, prefix)))
