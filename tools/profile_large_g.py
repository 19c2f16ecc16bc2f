#!/usr/bin/env python3
"""One exact kernel in the paper's large-g regime (EP300, k = 6, g = 16: 8008 combos) for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_tokens
from fastsk_amd import _native
tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
g, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 10)
e = _native.Engine(g, m)
e.compute(tokens, offsets, ntr, nte)
print(e.stats()["combos_done"], e.stats()["cell_updates"])
