#!/usr/bin/env python3
"""End-to-end check in the style of the reference's only CI test (QData/FastSK
test/run_check.py:37-64): gapped-k-mer kernel on the GPU -> LinearSVC + 5-fold calibration ->
test AUC >= 0.9 on EP300. Same user code as the reference; only the import resolves to the
MI355X engine (this repo's `fastsk/` alias package).

    python examples/run_check.py --train EP300.train.fasta --test EP300.test.fasta
"""
import argparse
import os
import sys
import time

import numpy as np
from sklearn.calibration import CalibratedClassifierCV
from sklearn.metrics import roc_auc_score
from sklearn.svm import LinearSVC

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastsk import FastSK, FastaUtility  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", default="/root/reference/data/EP300.train.fasta")
    ap.add_argument("--test", default="/root/reference/data/EP300.test.fasta")
    ap.add_argument("-g", type=int, default=10)
    ap.add_argument("-m", type=int, default=6)
    ap.add_argument("--exact", action="store_true", help="exact kernel instead of approx=True, t=1")
    args = ap.parse_args()

    reader = FastaUtility()
    Xtrain, Ytrain = reader.read_data(args.train)
    Xtest, Ytest = reader.read_data(args.test)

    t0 = time.time()
    k = FastSK(g=args.g, m=args.m) if args.exact else FastSK(g=args.g, m=args.m, t=1, approx=True)
    k.compute_kernel(Xtrain, Xtest)
    Ktr, Kte = k.get_train_kernel_np(), k.get_test_kernel_np()  # get_train_kernel() gives lists, as upstream
    print("kernel: %d x %d train, %d x %d test in %.3f s (%s)" % (*Ktr.shape, *Kte.shape, time.time() - t0, k.stats()["path_used"]))

    clf = CalibratedClassifierCV(LinearSVC(C=1), cv=5).fit(Ktr, Ytrain)
    acc = clf.score(Kte, np.array(Ytest))
    auc = roc_auc_score(Ytest, clf.predict_proba(Kte)[:, 1])
    print("calibrated linear SVM on the kernel rows: accuracy %.4f, test AUC %.4f" % (acc, auc))
    if auc < 0.9:  # the reference's acceptance threshold (test/run_check.py:64)
        raise SystemExit("test AUC %.4f is below 0.9" % auc)


if __name__ == "__main__":
    main()
