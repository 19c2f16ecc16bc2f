"""tools/time_compute_train.py: the reference's `time_fastsk` / `FastskRunner` harness (test/utils.py:15-66,
393-417) on this engine — compute_train only, in a spawned child with a timeout, the parent never on the GPU."""
import os
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, GOLD

sys.path.insert(0, os.path.join(ROOT, "tools"))


def write_fasta(path, X, labels):
    with open(path, "w") as f:
        for x, y in zip(X, labels):
            f.write(">%d\n%s\n" % (y, "".join("acgt"[v] for v in x)))


def test_runner_reads_like_the_reference_and_parent_stays_off_the_gpu(tmp_path):
    import time_compute_train as tct
    rng = np.random.default_rng(5)
    X = rng.integers(0, 4, size=(40, 50))
    write_fasta(tmp_path / "toy.train.fasta", X, rng.integers(0, 2, size=40))
    r = tct.FastskRunner("toy", str(tmp_path))
    assert len(r.train_seq) == 40 and len(r.train_seq[0]) == 50 and r.test_seq == []
    assert set(np.unique(np.array(r.train_seq))) <= {1, 2, 3, 4}      # ids from 1 in first-seen order
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device leg below is for the CPU container")
    # no device here: the spawned child fails loudly (no CPU fallback), the harness still returns the elapsed time
    secs = tct.time_fastsk(8, 4, 1, str(tmp_path), "toy", timeout=120)
    assert secs > 0 and not tct.time_fastsk.timed_out and tct.time_fastsk.exitcode not in (0, None)


@pytest.mark.gpu
def test_time_fastsk_on_the_gpu(tmp_path):
    import time_compute_train as tct
    rng = np.random.default_rng(6)
    X = rng.integers(0, 4, size=(3000, 200))
    write_fasta(tmp_path / "toy.train.fasta", X, rng.integers(0, 2, size=3000))
    # child process, generous timeout: finishes by itself
    secs = tct.time_fastsk(10, 6, 1, str(tmp_path), "toy", approx=False, timeout=300)
    assert 0 < secs < 300 and not tct.time_fastsk.timed_out and tct.time_fastsk.exitcode == 0
    # approx + max_iters, the form the reference's experiments use (I=...)
    secs = tct.time_fastsk(10, 6, 1, str(tmp_path), "toy", approx=True, max_iters=5, timeout=300)
    assert 0 < secs < 300 and tct.time_fastsk.exitcode == 0
    # a timeout far below the start-up time of the child: it is terminated, the clock says ~timeout
    t0 = time.time()
    secs = tct.time_fastsk(10, 6, 1, str(tmp_path), "toy", timeout=0.05)
    assert tct.time_fastsk.timed_out and secs < 30 and time.time() - t0 < 30
    # and the CLI
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_compute_train.py"), "--data", str(tmp_path), "--prefix", "toy",
                        "-g", "10", "-m", "6", "-t", "1", "--timeout", "300"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["seconds"] > 0 and not out["timed_out"]
