#!/usr/bin/env python3
"""Rewrite the tuning-key table of INTEGRATION.md (section 4b) from fsk_tuning_keys() of the built library, so that the
documented keys, defaults and ranges are the ones the engine has.   python3 tools/gen_tuning_table.py"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsk_amd import _native

rows = ["| key | default | range | effect |", "|---|---|---|---|"]
for line in _native.library().L.fsk_tuning_keys().decode().splitlines():
    m = re.match(r"(\w+)=(-?\d+) \[(-?\d+)\.\.(-?\d+)\] (.*)", line)
    rows.append("| `%s` | %s | %s … %s | %s |" % (m.group(1), m.group(2), m.group(3), m.group(4), m.group(5).replace("|", "\\|")))
path = os.path.join(ROOT, "INTEGRATION.md")
text = open(path).read()
a = text.index("| key | default | range | effect |")
b = text.index("\n\n", a)
open(path, "w").write(text[:a] + "\n".join(rows) + text[b:])
print("%d keys" % (len(rows) - 2))
