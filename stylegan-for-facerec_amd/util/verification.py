"""k-fold face-verification metrics on pair embeddings (reference: util/verification.py:37-172, used by
``perform_val`` at util/utils.py:303).

Same protocol and results as the reference's ``evaluate``: squared Euclidean distance between the two embeddings
of every pair, thresholds 0.00 .. 3.99 in steps of 0.01, ``nrof_folds`` contiguous folds (``KFold(shuffle=False)``
semantics: the first ``n % k`` folds hold one pair more); per fold the threshold with the best training accuracy
(first maximum) is applied to the test pairs; TPR / FPR curves are fold means over the test pairs.

Written for whole threshold grids at once instead of the reference's threshold-by-threshold Python loop: a pair is
predicted "same" at threshold index t iff t >= rank(d), rank(d) = number of grid values <= d, so one histogram of
ranks per fold and class gives every confusion count by a cumulative sum (400 thresholds x 6000 pairs x 10 folds
take microseconds instead of seconds).  Pinned by tests/golden/g10_verification.npz, captured from the reference.
"""
import numpy as np

__all__ = ["calculate_roc", "calculate_accuracy", "evaluate", "kfold_bounds"]


def kfold_bounds(n, k):
    """[(start, stop)] of the k contiguous test folds over range(n) -- sklearn KFold(n_splits=k, shuffle=False)."""
    if k < 2 or k > n:
        raise ValueError("Cannot have number of splits n_splits=%d greater than the number of samples: n_samples=%d."
                         % (k, n) if k > n else "k-fold cross-validation requires at least one train/test split")
    sizes = np.full(k, n // k, dtype=np.int64)
    sizes[: n % k] += 1
    stops = np.cumsum(sizes)
    return [(int(e - s), int(e)) for s, e in zip(sizes, stops)]


def calculate_accuracy(threshold, dist, actual_issame):
    """(tpr, fpr, accuracy) of the rule ``dist < threshold`` (reference :95-106)."""
    dist = np.asarray(dist)
    same = np.asarray(actual_issame, dtype=bool)
    pred = dist < threshold
    tp = int(np.count_nonzero(pred & same))
    fp = int(np.count_nonzero(pred & ~same))
    tn = int(np.count_nonzero(~pred & ~same))
    fn = int(np.count_nonzero(~pred & same))
    tpr = 0 if tp + fn == 0 else float(tp) / float(tp + fn)
    fpr = 0 if fp + tn == 0 else float(fp) / float(fp + tn)
    return tpr, fpr, float(tp + tn) / dist.size


def calculate_roc(thresholds, embeddings1, embeddings2, actual_issame, nrof_folds=10, pca=0):
    """Returns (tpr[nthr], fpr[nthr], accuracy[nrof_folds], best_thresholds[nrof_folds]) -- reference :37-92."""
    if pca:
        raise NotImplementedError("verification with a per-fold PCA (pca > 0) is not used by perform_val and not built")
    embeddings1, embeddings2 = np.asarray(embeddings1), np.asarray(embeddings2)
    assert embeddings1.shape[0] == embeddings2.shape[0]
    assert embeddings1.shape[1] == embeddings2.shape[1]
    thresholds = np.asarray(thresholds)
    same_all = np.asarray(actual_issame, dtype=bool)
    n = min(len(same_all), embeddings1.shape[0])
    nthr = len(thresholds)
    diff = np.subtract(embeddings1, embeddings2)
    dist = np.sum(np.square(diff), 1)
    rank = np.searchsorted(thresholds, dist[:n], side="right").astype(np.int64)  # accepted iff t >= rank
    rank = np.minimum(rank, nthr)  # distances beyond the grid are never accepted
    same = same_all[:n]
    tprs = np.zeros((nrof_folds, nthr))
    fprs = np.zeros((nrof_folds, nthr))
    accuracy = np.zeros(nrof_folds)
    best_thresholds = np.zeros(nrof_folds)
    tp_all, fp_all = _accept_counts(rank, same, nthr)
    n_diff_all = int(n - np.count_nonzero(same))
    for f, (lo, hi) in enumerate(kfold_bounds(n, nrof_folds)):
        tp_te, fp_te = _accept_counts(rank[lo:hi], same[lo:hi], nthr)
        ns_te = int(np.count_nonzero(same[lo:hi]))
        nd_te = (hi - lo) - ns_te
        tp_tr, fp_tr = tp_all - tp_te, fp_all - fp_te
        nd_tr = n_diff_all - nd_te
        # training accuracy per threshold = (tp + tn) / size, tn = n_diff - fp
        acc_tr = (tp_tr + (nd_tr - fp_tr)) / float(n - (hi - lo))
        best = int(np.argmax(acc_tr))  # first maximum, as np.argmax over the reference's loop results
        best_thresholds[f] = thresholds[best]
        tprs[f] = tp_te / float(ns_te) if ns_te else 0.0
        fprs[f] = fp_te / float(nd_te) if nd_te else 0.0
        accuracy[f] = (tp_te[best] + (nd_te - fp_te[best])) / float(hi - lo)
    return np.mean(tprs, 0), np.mean(fprs, 0), accuracy, best_thresholds


def _accept_counts(rank, same, nthr):
    """tp[t], fp[t]: pairs of each class accepted at threshold index t (accepted iff t >= rank)."""
    h_same = np.bincount(rank[same], minlength=nthr + 1)[:nthr]
    h_diff = np.bincount(rank[~same], minlength=nthr + 1)[:nthr]
    return np.cumsum(h_same), np.cumsum(h_diff)


def evaluate(embeddings, actual_issame, nrof_folds=10, pca=0):
    """Pairs are interleaved rows (2i, 2i+1) of ``embeddings`` -- reference :160-172."""
    thresholds = np.arange(0, 4, 0.01)
    embeddings = np.asarray(embeddings)
    return calculate_roc(thresholds, embeddings[0::2], embeddings[1::2], np.asarray(actual_issame),
                         nrof_folds=nrof_folds, pca=pca)
