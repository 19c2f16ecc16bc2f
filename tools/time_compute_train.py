#!/usr/bin/env python3
"""The reference's timing harness for `compute_train`, on this engine.

Mirrors /root/reference/test/utils.py:15-66 (`time_fastsk`) and :393-417 (`FastskRunner`): same names,
arguments, defaults and meaning —

  * `FastskRunner(prefix, data_location)` reads `<data_location>/<prefix>.train.fasta` / `.test.fasta` through
    `FastaUtility` when it is constructed (reading is NOT timed, utils.py:31);
  * `time_fastsk(g, m, t, data_location, prefix, approx, max_iters, timeout, skip_variance)` returns the wall
    seconds of ONE `FastSK(g, m, t, approx, max_iters=I, delta, skip_variance).compute_train(train_seq)`
    (utils.py:405-417); with `timeout` the call runs in a child process that is terminated when the timeout
    expires, and the clock — as in the reference, utils.py:33-66 — runs from before the child is started until it
    has ended (process start-up included).

What differs, because the work happens on a GPU: the child is a SPAWNED interpreter (a forked copy of a process
that has a HIP context cannot use the device), and this parent never touches the GPU — it only reads FASTA files
(host code) — so the child is started before anything in this process could have initialised the device, and the
device is free for it. The spawned child re-reads nothing: the token lists travel to it as the runner object.

CLI:  python tools/time_compute_train.py --data DIR --prefix EP300 -g 10 -m 6 [-t 1] [--approx] [-I 50]
                                         [--skip-variance] [--timeout 60]
prints one JSON line {"seconds": ..., "timed_out": ...}.
"""
import argparse
import json
import multiprocessing
import os
import os.path as osp
import sys
import time

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class FastskRunner:
    """reference test/utils.py:393-417 (the kernel-timing half; `train_and_test` is the SVM stage, out of scope)"""

    def __init__(self, prefix, data_location="../data"):
        from fastsk import FastaUtility  # host code only: no GPU call
        self.prefix = prefix
        self.train_file = osp.join(data_location, prefix + ".train.fasta")
        self.test_file = osp.join(data_location, prefix + ".test.fasta")
        reader = FastaUtility()
        self.train_seq, self.Ytrain = reader.read_data(self.train_file)
        if osp.exists(self.test_file):
            self.test_seq, self.Ytest = reader.read_data(self.test_file)
        else:  # (the reference requires both files; timing compute_train only needs the train set)
            self.test_seq, self.Ytest = [], []

    def compute_train_kernel(self, g, m, t=20, approx=True, I=100, delta=0.025, skip_variance=False):
        from fastsk import FastSK
        kernel = FastSK(g=g, m=m, t=t, approx=approx, max_iters=I, delta=delta, skip_variance=skip_variance)
        kernel.compute_train(self.train_seq)


def _child(runner, g, m, kwargs):
    runner.compute_train_kernel(g, m, **kwargs)


def time_fastsk(g, m, t, data_location, prefix, approx=False, max_iters=None, timeout=None, skip_variance=False):
    """Run FastSK kernel computation. If a timeout is provided, it runs as a child process, which is killed when
    the timeout is reached (reference test/utils.py:15-66). Returns the elapsed seconds."""
    fastsk = FastskRunner(prefix, data_location)
    if max_iters:
        args = {"t": t, "approx": approx, "skip_variance": skip_variance, "I": max_iters}
    else:
        args = {"t": t, "approx": approx, "skip_variance": skip_variance}
    start = time.time()
    if timeout:
        ctx = multiprocessing.get_context("spawn")  # a fresh interpreter: started before any GPU call of this process
        p = ctx.Process(target=_child, name="TimeFastSK", args=(fastsk, g, m, args))
        p.start()
        p.join(timeout)
        time_fastsk.timed_out = p.is_alive()
        if p.is_alive():
            p.terminate()
            p.join()
        time_fastsk.exitcode = p.exitcode
    else:
        time_fastsk.timed_out = False
        time_fastsk.exitcode = 0
        fastsk.compute_train_kernel(g, m, **args)
    end = time.time()
    return end - start


time_fastsk.timed_out = False
time_fastsk.exitcode = 0


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--data", required=True, help="directory holding <prefix>.train.fasta")
    ap.add_argument("--prefix", required=True)
    ap.add_argument("-g", type=int, required=True)
    ap.add_argument("-m", type=int, required=True)
    ap.add_argument("-t", type=int, default=20)
    ap.add_argument("--approx", action="store_true")
    ap.add_argument("-I", "--max-iters", type=int, default=None)
    ap.add_argument("--skip-variance", action="store_true")
    ap.add_argument("--timeout", type=float, default=None)
    a = ap.parse_args()
    secs = time_fastsk(a.g, a.m, a.t, a.data, a.prefix, approx=a.approx, max_iters=a.max_iters, timeout=a.timeout,
                       skip_variance=a.skip_variance)
    print(json.dumps({"seconds": secs, "timed_out": time_fastsk.timed_out, "exitcode": time_fastsk.exitcode,
                      "prefix": a.prefix, "g": a.g, "m": a.m, "t": a.t, "approx": a.approx, "max_iters": a.max_iters,
                      "timeout": a.timeout}))
    if time_fastsk.exitcode not in (0, None) and not time_fastsk.timed_out:
        sys.exit(1)


if __name__ == "__main__":
    main()
