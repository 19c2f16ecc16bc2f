#!/usr/bin/env python3
"""What a FastSK user waits for: `FastSK(...)` + `compute_kernel` + getter, wall clock, for the BASELINE configs
1-4 and synthetic DNA at N = 16k — three input forms (Python lists as FastaUtility.read_data returns them; 2-D numpy / flat
tokens + offsets; `compute_kernel_flat`) and three getters (lists, numpy, DLPack), COLD (a fresh process: nothing GPU-side
before the clock starts — import, library load, HIP initialisation, first-call allocations included) and WARM (the same call
again in that process). Split: import + construct / input conversion + GPU / result boxing.
    tools/time_dropin.py [--cases cfg2,cfg1,...] > profiles/r06_dropin_wall.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
t_start = time.perf_counter()
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
case, form, getter = sys.argv[1], sys.argv[2], sys.argv[3]
# ---- the inputs, prepared before anything of the package is touched (a user has them from FastaUtility.read_data)
from conftest import load_golden, load_tokens
if case.startswith("synthetic"):
    N = int(case.split("_")[1]); rng = np.random.Generator(np.random.PCG64(1)); X = rng.integers(1, 5, size=(N, 300), dtype=np.int32)
    tokens, offsets, ntr, nte = X.reshape(-1), np.arange(N + 1, dtype=np.int64) * 300, N - N // 4, N // 4
    kw = dict(g=12, m=8)
    order = None
else:
    d = load_golden(case)
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    kw = dict(g=d["g"], m=d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"], skip_variance=bool(d["skip_variance"]))
    order = d["order"].tolist() if d["approx"] else None
if form == "lists":
    rows = [tokens[offsets[i]:offsets[i + 1]].tolist() for i in range(ntr + nte)]
    Xtr, Xte = rows[:ntr], rows[ntr:]
elif form == "numpy":
    L = int(offsets[1] - offsets[0])
    same = bool(np.all(np.diff(offsets) == L))
    if same:
        A = np.ascontiguousarray(tokens.reshape(ntr + nte, L)); Xtr, Xte = A[:ntr], A[ntr:]
    else:
        rows = [np.ascontiguousarray(tokens[offsets[i]:offsets[i + 1]]) for i in range(ntr + nte)]; Xtr, Xte = rows[:ntr], rows[ntr:]
t_inputs = time.perf_counter()
out = {"case": case, "input": form, "getter": getter, "n_train": ntr, "n_test": nte}
for phase in ("cold", "warm"):
    t0 = time.perf_counter()
    from fastsk import FastSK           # (cold: the import, the library load)
    f = FastSK(**kw)
    if order is not None:
        f.set_combo_order(order)
    t1 = time.perf_counter()
    if form == "flat":
        f.compute_kernel_flat(tokens, offsets, ntr)
    else:
        f.compute_kernel(Xtr, Xte)
    t2 = time.perf_counter()
    if getter == "lists":
        a, b = f.get_train_kernel(), f.get_test_kernel()
        probe = a[0][0] + b[0][0]
    elif getter == "numpy":
        a, b = f.get_train_kernel_np(), f.get_test_kernel_np()
        probe = float(a[0, 0] + b[0, 0])
    else:
        a, b = f.get_train_kernel_dlpack(), f.get_test_kernel_dlpack()
        probe = 0.0
    t3 = time.perf_counter()
    out[phase] = {"import_and_construct_ms": round(1e3 * (t1 - t0), 2), "compute_kernel_ms": round(1e3 * (t2 - t1), 2),
                  "getters_ms": round(1e3 * (t3 - t2), 2), "total_ms": round(1e3 * (t3 - t0), 2)}
    del a, b, f
print(json.dumps(out))
''' % ROOT

CASES = {"cfg1": "f7_cfg1_prot11_approx_t1", "cfg2": "f7_cfg2_ep300_exact", "cfg3": "f7_cfg3_ep47848_100combos", "cfg4": "f7_cfg4_prot219_exact",
         "syn16k": "synthetic_16000"}


def main():
    cases = ["cfg2", "cfg1", "cfg3", "cfg4", "syn16k"]
    if "--cases" in sys.argv:
        cases = sys.argv[sys.argv.index("--cases") + 1].split(",")
    rows = []
    for c in cases:
        for form, getter in (("lists", "lists"), ("lists", "numpy"), ("numpy", "numpy"), ("flat", "numpy"), ("flat", "dlpack")):
            r = subprocess.run([sys.executable, "-c", CHILD, CASES[c], form, getter], capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"case": c, "input": form, "getter": getter, "error": r.stderr[-300:]})
            rows.append(json.loads(line))
            print(line, file=sys.stderr, flush=True)
    print(json.dumps({"what": "wall clock of FastSK(...) + compute_kernel + getters, one MI355X, fresh process per row (cold = first call in it, "
                              "warm = the same again); GPU work of these calls: 2-8 ms (profiles/r06_configs1-4_gpu_timings.jsonl)", "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
