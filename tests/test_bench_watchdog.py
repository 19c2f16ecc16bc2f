"""bench.py's fail-fast watchdog (no GPU needed): a stage that outlives its bound — the main thread stuck in a C call,
as inside a collective — produces ONE JSON error line on rank 0 and exit code 2; the launcher's SIGTERM produces the
same line and exit code 143. Never a hang, never a re-exec."""
import json
import os
import signal
import subprocess
import sys
import time

from conftest import ROOT

CHILD = r"""
import sys, types, ctypes
sys.path.insert(0, %r)
import bench
args = types.SimpleNamespace(steps=3, warmup=1)
wd = bench.Watchdog(int(sys.argv[1]), 8, args, grace_s=1.0)
wd.partial["combos_ms_per_step_this_rank"] = 123.4
wd.stage("init_process_group(nccl)", 30.0)
wd.stage("combos: 3 timed steps", float(sys.argv[2]))
print("READY", flush=True)
libc = ctypes.CDLL(None)
for _ in range(40):              # the main thread is inside C calls (a signal only cuts one short)
    libc.sleep(1)
print("woke up", flush=True)
"""


def run_child(rank, bound, kill_after=None):
    p = subprocess.Popen([sys.executable, "-c", CHILD % ROOT, str(rank), str(bound)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.perf_counter()
    if kill_after is not None:
        assert p.stdout.readline().strip() == "READY"
        time.sleep(kill_after)
        p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    return p.returncode, out, err, time.perf_counter() - t0


def test_overrun_prints_one_json_line_and_exits_2():
    rc, out, err, dt = run_child(0, 1.0)
    assert rc == 2 and dt < 20
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["stage"] == "combos: 3 timed steps" and d["n_gpus"] == 8 and d["value"] is None and "exceeded its bound" in d["error"]
    assert d["timings"]["init_process_group(nccl)"] >= 0 and d["partial"]["combos_ms_per_step_this_rank"] == 123.4
    assert d["steps"] == 3 and d["warmup"] == 1 and "woke up" not in out


def test_other_ranks_stay_quiet_on_stdout_and_wait_for_rank_0():
    rc, out, err, dt = run_child(3, 1.0)
    assert rc == 2 and not [ln for ln in out.splitlines() if ln.startswith("{")] and "rank 3" in err
    assert dt >= 1.8   # bound + grace: rank 0's line is out first


def test_sigterm_from_the_launcher_still_leaves_the_line():
    rc, out, err, dt = run_child(0, 300.0, kill_after=0.5)
    assert rc == 143 and dt < 20
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert "SIGTERM" in d["error"] and d["stage"] == "combos: 3 timed steps"
