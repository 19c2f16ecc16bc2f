#!/usr/bin/env python3
"""Generate tests/golden/fasta/*: small FASTA-like files OF OUR OWN (mixed case, CRLF and lone-CR
line ends, padding white space, an empty sequence line, labels -1/0/1, no final newline) and the
token arrays the REFERENCE's reader (imported from /root/reference/src/fastsk/utils.py; build
container only) produces for them. Only our inputs and the reference's outputs are stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ref_reader  # noqa: E402

OUT = os.path.join(HERE, "golden", "fasta")

FILES = {
    "messy.train.fasta": b">1\nACGTacgtNN\n>0\r\n  ttgaCCa \r\n>-1\n\n> 1 \nAC-GT*x\r>0\rgattaca\n",
    "messy.test.fasta": b">0\nnnacgu\n>1\nBZXacgt",
    "protein.train.fasta": b">1\nMKVLAAGIVGLLLAQ\n>0\nmkvwaagyvgl\n",
}


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, data in FILES.items():
        open(os.path.join(OUT, name), "wb").write(data)
    rd = ref_reader()  # one vocabulary over messy.train then messy.test, as a user would
    out = {}
    for name in ("messy.train.fasta", "messy.test.fasta"):
        X, Y = rd.read_data(os.path.join(OUT, name))
        out[name + ":tokens"] = np.array([t for x in X for t in x], dtype=np.int32)
        out[name + ":lengths"] = np.array([len(x) for x in X], dtype=np.int64)
        out[name + ":labels"] = np.array(Y, dtype=np.int64)
    rd = ref_reader()
    X, Y = rd.read_data(os.path.join(OUT, "protein.train.fasta"))
    out["protein.train.fasta:tokens"] = np.array([t for x in X for t in x], dtype=np.int32)
    out["protein.train.fasta:lengths"] = np.array([len(x) for x in X], dtype=np.int64)
    out["protein.train.fasta:labels"] = np.array(Y, dtype=np.int64)
    np.savez(os.path.join(OUT, "expected.npz"), **out)
    print({k: v.tolist() for k, v in out.items()})


if __name__ == "__main__":
    main()
