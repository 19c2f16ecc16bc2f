"""The N>1 host path on the CPU: world_size 2 over gloo, emulated engine, against the golden
vectors. Checks the combo partition, the single all-reduce and that every rank normalises the
same reduced triangle."""
import os
import subprocess
import sys

import numpy as np

from conftest import GOLD, ROOT, load_golden


def test_shard_partition():
    from fastsk_amd.distributed import shard
    combos = np.arange(495)
    parts = [shard(combos, r, 8) for r in range(8)]
    assert sorted(np.concatenate(parts).tolist()) == combos.tolist()
    assert [len(p) for p in parts] == [62, 62, 62, 62, 62, 62, 62, 61]
    assert shard(combos, 3, 8)[:3].tolist() == [3, 11, 19]


def test_two_rank_gloo_all_reduce(tmp_path):
    name = "f3_ragged_sigma7_g6m3"
    d = load_golden(name)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"),
                                       os.path.join(GOLD, name + ".npz"), str(tmp_path)], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    done = 0
    for rank in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % rank))
        assert int(z["world"]) == 2
        assert np.array_equal(z["counts"], d["counts"])  # identical on every rank after the all-reduce
        assert np.array_equal(z["tri"], d["tri"])
        done += int(z["done"])
    assert done == len(d["combos"])  # every combo processed exactly once across the ranks
