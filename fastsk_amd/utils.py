"""FASTA-like reader with the reference's tokenisation.

Same surface as ``fastsk.utils`` of QData/FastSK (reference ``src/fastsk/utils.py:5-104``):
``Vocabulary`` and ``FastaUtility.read_data / shortest_seq``. Input format: alternating ``>label``
/ sequence lines, every line stripped and lower-cased (``utils.py:78``), ids handed out in
first-seen order starting at 1 with id 0 reserved (``utils.py:13``), one vocabulary per
``FastaUtility`` so that a train and a test file read through the same object agree on ids.

The work is done by the native reader ``fsk_read_fasta`` (``csrc/fsk_fasta.cpp``) behind
``read_packed``, which returns flat arrays ready for the C ABI; ``read_data`` only reshapes them
into the reference's nested lists; without the built library the same arrays come from a numpy pass
over the text (``_read_text``). Pinned against the reference's own reader by
``tests/golden/tokens_*.npz`` (``tests/test_tokeniser.py``).
"""
import ctypes as C

import numpy as np


class Vocabulary(object):
    """Token -> id map; id 0 is reserved (reference ``utils.py:5-36``)."""

    def __init__(self):
        self._token2idx = {0: 0}
        self._size = 1

    def add(self, token):
        """Id of ``token``; a token seen for the first time gets the next free id."""
        idx = self._token2idx.get(token)
        return self._assign(token) if idx is None else idx

    def _assign(self, token):
        idx = self._token2idx[token] = self._size
        self._size += 1
        return idx

    def size(self):
        return self._size

    def __str__(self):
        return str(self._token2idx)

    # ---- the byte table the native reader works on
    def _table(self):
        lut = np.zeros(256, dtype=np.int32)
        for tok, idx in self._token2idx.items():
            if isinstance(tok, str) and len(tok) == 1 and ord(tok) < 128:
                lut[ord(tok)] = idx
        return lut

    def _absorb(self, lut):
        """Take over the ids the native reader assigned (in id order, so ``add`` reproduces them)."""
        fresh = [(int(lut[b]), chr(b)) for b in range(128) if lut[b] > 0 and chr(b) not in self._token2idx]
        for idx, tok in sorted(fresh):
            got = self._assign(tok)
            assert got == idx, "vocabulary changed under the reader"


class MalformedFasta(AssertionError, ValueError):
    """The file is not alternating ``>label`` / sequence lines with labels in {-1, 0, 1} (where the
    reference's reader fails an ``assert``, ``utils.py:80-94``)."""


class FastaUtility:
    def __init__(self, vocab=None):
        self._vocab = Vocabulary() if vocab is None else vocab

    def read_packed(self, data_file):
        """``(tokens int32[total], offsets int64[n+1], labels int64[n])`` of ``data_file`` — the
        flat form the C ABI and ``FastSK.compute_kernel_flat`` take — with the tokenisation of
        ``read_data`` and this object's shared vocabulary."""
        from . import _native
        try:
            L = _native.library().L
        except (ImportError, OSError):
            # no hipcc-built library on this host: the tokeniser (host code, no GPU involved) stays usable
            # through the whole-file numpy pass below — same tokens, same errors
            return self._read_text(data_file)
        lut = self._vocab._table()
        nxt = C.c_int32(self._vocab.size())
        n_seq, n_tok = C.c_int64(0), C.c_int64(0)
        err = C.create_string_buffer(256)
        path = str(data_file).encode()
        rc = L.fsk_read_fasta(path, lut.ctypes.data, C.byref(nxt), None, 0, None, None, 0, C.byref(n_seq), C.byref(n_tok), err, 256)
        if rc == 0:
            tokens = np.empty(n_tok.value, dtype=np.int32)
            offsets = np.empty(n_seq.value + 1, dtype=np.int64)
            labels = np.empty(n_seq.value, dtype=np.int32)
            rc = L.fsk_read_fasta(path, lut.ctypes.data, C.byref(nxt), tokens.ctypes.data, n_tok.value, offsets.ctypes.data,
                                  labels.ctypes.data, n_seq.value, C.byref(n_seq), C.byref(n_tok), err, 256)
        if rc == -6:  # FSK_EUNSUPPORTED: non-ASCII text, characters are not bytes
            return self._read_text(data_file)
        if rc != 0:
            if not err.value.startswith(b"cannot open"):
                raise MalformedFasta(err.value.decode())
            open(data_file).close()  # raises the OSError a Python reader would
            raise OSError(err.value.decode())
        self._vocab._absorb(lut)
        return tokens, offsets, labels.astype(np.int64)

    def _read_text(self, data_file, want_labels=True):
        """Any text (non-ASCII alphabets): whole-file numpy pass over code points."""
        with open(data_file, "r") as f:
            lines = f.read().split("\n")
        if lines and lines[-1] == "":
            lines.pop()  # the text after the last newline is a line only when it is not empty
        lines = [ln.strip().lower() for ln in lines]
        if len(lines) % 2:
            raise MalformedFasta("the file ends after a label line")
        labels = self._labels(lines[0::2]) if want_labels else None
        seqs = lines[1::2]
        lens = np.fromiter(map(len, seqs), dtype=np.int64, count=len(seqs))
        offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        cps = np.array([ord(c) for c in "".join(seqs)], dtype=np.int64)  # code points
        vals, first = np.unique(cps, return_index=True)
        ids = {int(v): self._vocab.add(chr(int(v))) for v in vals[np.argsort(first)]}
        tokens = np.zeros(len(cps), dtype=np.int32)
        for v, i in ids.items():
            tokens[cps == v] = i
        return tokens, offsets, labels

    @staticmethod
    def _labels(label_lines, regression=False):
        out = []
        for ln in label_lines:
            parts = ln.split(">")
            if len(parts) != 2:
                raise MalformedFasta("expected a label line: %r" % ln)
            if regression:
                out.append(parts[1])
                continue
            try:
                v = int(parts[1])
            except ValueError:
                raise MalformedFasta("expected a label line: %r" % ln) from None
            if v not in (-1, 0, 1):
                raise MalformedFasta("label %d not in {-1, 0, 1}" % v)
            out.append(v)
        return out if regression else np.asarray(out, dtype=np.int64)

    def read_data(self, data_file, vocab="inferred", regression=False):
        """``(X, Y)``: token-id lists and labels of ``data_file`` — the reference's return value
        (``utils.py:50-96``). Labels are ints in {-1, 0, 1}; with ``regression`` the label text."""
        assert vocab.lower() in ["dna", "protein", "inferred"]
        if regression:  # free-text labels: tokens as usual, labels re-read as text
            with open(data_file, "r") as f:
                label_lines = [ln.strip().lower() for ln in f.read().split("\n")[0::2]]
            tokens, offsets, _ = self._read_text(data_file, want_labels=False)
            Y = self._labels(label_lines[:len(offsets) - 1], regression=True)
        else:
            tokens, offsets, labels = self.read_packed(data_file)
            Y = labels.tolist()
        flat = tokens.tolist()
        X = [flat[a:b] for a, b in zip(offsets[:-1].tolist(), offsets[1:].tolist())]
        return X, Y

    def shortest_seq(self, data_file):
        _, offsets, _ = self.read_packed(data_file)
        return int(np.diff(offsets).min())
