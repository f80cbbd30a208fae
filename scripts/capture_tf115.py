#!/usr/bin/env python
"""Pin the oracle against the REAL reference: run tsinghua-fib-lab/CLSR's own CLSRModel under TensorFlow 1.15.

    # in an environment with python 3.6/3.7, tensorflow==1.15.2, numpy, pandas, scikit-learn, pyyaml
    python scripts/capture_tf115.py --reference /path/to/CLSR [--out tests/golden/tf115_clsr_step.npz]

This cannot run in the build container (no TensorFlow; the reference needs TF 1.x APIs).  It is the one command
that turns "parity unpinned" into a pinned oracle: it imports the reference from ``--reference``, builds
``CLSRModel`` + ``SASequentialIterator`` on the committed synthetic slice (tests/golden/data) with the hparams of
tests/conftest.py::golden_hparams, assigns the deterministic weight set F2 (oracle/tf115_pin.py), feeds the
committed reference-captured batch tests/golden/iterator_train_sa.npz[b0] and writes

    meta/variables, meta/no_grad      TF's own names of the trainables; those whose tf.gradients entry is None
    before/<variable>                 every global variable right after the F2 assignment (the tests load THESE weights)
    logit, alpha                      forward values of the training graph (is_train_stage = True)
    loss/{loss,data_loss,regular_loss,contrastive_loss,discrepancy_loss}
    grad/<variable>                   tf.gradients(loss, variable), IndexedSlices densified
    slices_norm/<table>               norm of the raw IndexedSlices values (what tf.clip_by_norm sees)
    after/<variable>                  every global variable after ONE model.train step (Adam + BN moving statistics)
    eval_pred                         model.infer on tests/golden/iterator_eval_sa.npz[b0] after that step

Only that .npz travels back into the repo (tests/golden/); tests/test_tf115_pin.py consumes it.  What each key
confirms about TF 1.15 semantics: oracle/tf115_pin.py::CONFIRMS.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tf115_pin as P  # noqa: E402  (numpy only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of tsinghua-fib-lab/CLSR")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", P.PIN_FILE))
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    import tensorflow as tf

    assert tf.__version__.startswith("1.15"), "the reference is pinned to tensorflow 1.15.2 (README.md:7), got " + tf.__version__
    from reco_utils.recommender.deeprec.deeprec_utils import prepare_hparams
    from reco_utils.recommender.deeprec.io.sequential_iterator import SASequentialIterator
    from reco_utils.recommender.deeprec.models.sequential.clsr import CLSRModel

    gold = os.path.join(ROOT, "tests", "golden")
    d = os.path.join(gold, "data")
    yaml_file = os.path.join(args.reference, "reco_utils", "recommender", "deeprec", "config", "clsr.yaml")
    hparams = prepare_hparams(yaml_file, user_vocab=os.path.join(d, "user_vocab.pkl"),
                              item_vocab=os.path.join(d, "item_vocab.pkl"),
                              cate_vocab=os.path.join(d, "category_vocab.pkl"), MODEL_DIR=None, SUMMARIES_DIR=None,
                              **P.HPARAMS)
    model = CLSRModel(hparams, SASequentialIterator, seed=1)
    it, sess = model.iterator, model.sess
    with model.graph.as_default():
        tvars = tf.trainable_variables()
        names = [v.op.name for v in tvars]
        w = P.f2_weights([(n, v.shape.as_list()) for n, v in zip(names, tvars)])
        sess.run([tf.assign(v, w[n]) for n, v in zip(names, tvars)])
        grads = tf.gradients(model.loss, tvars)
        dense_g, slice_norm, no_grad = {}, {}, []
        for n, g in zip(names, grads):
            if g is None:
                no_grad.append(n)
                continue
            if isinstance(g, tf.IndexedSlices):
                slice_norm[n] = tf.sqrt(tf.reduce_sum(tf.square(g.values)))
                g = tf.convert_to_tensor(g)
            dense_g[n] = g
        gvars = tf.global_variables()

    def feed_of(path, training):
        a = P.feed_arrays(np.load(path))
        fd = {getattr(it, k): a[k] for k in ("labels", "users", "items", "cates", "item_history", "item_cate_history",
                                             "mask", "time", "time_diff", "time_from_first_action", "time_to_now")}
        if hasattr(it, "attn_labels"):
            fd[it.attn_labels] = a["attn_labels"]
        fd[model.layer_keeps] = model.keep_prob_train if training else model.keep_prob_test
        fd[model.embedding_keeps] = model.embedding_keep_prob_train if training else model.embedding_keep_prob_test
        fd[model.is_train_stage] = training
        return fd

    fd = feed_of(os.path.join(gold, "iterator_train_sa.npz"), True)
    out = {"meta/tf_version": np.array(tf.__version__), "meta/variables": np.array(names),
           "meta/no_grad": np.array(no_grad, dtype=str)}
    for v, val in zip(gvars, sess.run(gvars)):                    # the weights this run really used (+ BN moving stats)
        out["before/" + v.op.name] = val
    fetch = dict(logit=model.logit, alpha=model.alpha_output, loss=model.loss, data_loss=model.data_loss,
                 regular_loss=model.regular_loss, contrastive_loss=model.contrastive_loss,
                 discrepancy_loss=model.discrepancy_loss)
    vals = sess.run(fetch, feed_dict=fd)
    out["logit"], out["alpha"] = vals["logit"], vals["alpha"]
    for k in ("loss", "data_loss", "regular_loss", "contrastive_loss", "discrepancy_loss"):
        out["loss/" + k] = np.asarray(vals[k], dtype=np.float64)
    for n, v in sess.run(dense_g, feed_dict=fd).items():
        out["grad/" + n] = v
    for n, v in sess.run(slice_norm, feed_dict=fd).items():
        out["slices_norm/" + n] = np.asarray(v, dtype=np.float64)
    model.train(sess, dict(fd))                                   # ONE optimisation step (Adam + UPDATE_OPS)
    for v, val in zip(gvars, sess.run(gvars)):
        out["after/" + v.op.name] = val
    pred = model.infer(sess, feed_of(os.path.join(gold, "iterator_eval_sa.npz"), False))[0]
    out["eval_pred"] = np.asarray(pred)
    np.savez_compressed(args.out, **out)
    print("wrote %s (%d arrays) -- commit it; tests/test_tf115_pin.py now pins the oracle and the HIP path" % (args.out, len(out)))


if __name__ == "__main__":
    main()
