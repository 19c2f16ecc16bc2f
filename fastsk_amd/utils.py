"""FASTA-like reader with the reference's tokenisation.

Mirrors ``fastsk.utils`` of QData/FastSK (reference ``src/fastsk/utils.py:5-104``): alternating
``>label`` / sequence lines, every line stripped and lower-cased (``utils.py:78``), token ids
handed out in first-seen order starting at 1 with id 0 reserved (``utils.py:13``), and one
vocabulary shared by every ``read_data`` call on the same ``FastaUtility`` so that train and test
agree on ids (``utils.py:50-96``). Pinned against the reference's own reader by
``tests/golden/tokens_*.npz`` (see ``tests/make_golden.py``).
"""


class Vocabulary(object):
    """Token -> id map; id 0 is reserved for the unknown token (reference ``utils.py:5-36``)."""

    def __init__(self):
        self._token2idx = {0: 0}
        self._size = 1

    def add(self, token):
        """Return the id of ``token``, assigning the next free id on first sight."""
        idx = self._token2idx.get(token)
        if idx is None:
            idx = self._size
            self._token2idx[token] = idx
            self._size += 1
        return idx

    def size(self):
        return self._size

    def __str__(self):
        return str(self._token2idx)


class FastaUtility:
    def __init__(self, vocab=None):
        self._vocab = Vocabulary() if vocab is None else vocab

    def read_data(self, data_file, vocab="inferred", regression=False):
        """Read ``data_file``; returns ``(X, Y)`` = token-id lists and labels.

        Labels are ints in {-1, 0, 1} unless ``regression`` (then the raw label string), as in
        the reference (``utils.py:80-87``).
        """
        assert vocab.lower() in ["dna", "protein", "inferred"]
        X, Y = [], []
        add = self._vocab.add
        with open(data_file, "r") as f:
            expect_label = True
            for line in f:
                line = line.strip().lower()
                if expect_label:
                    parts = line.split(">")
                    assert len(parts) == 2
                    if regression:
                        label = parts[1]
                    else:
                        label = int(parts[1])
                        assert label in [-1, 0, 1]
                    Y.append(label)
                else:
                    X.append([add(ch) for ch in line])
                expect_label = not expect_label
        assert len(X) == len(Y)
        return X, Y

    def read_packed(self, data_file):
        """Vectorised reader for large files: same tokenisation as ``read_data`` (same shared
        vocabulary, ids in first-seen order) but returns flat arrays
        ``(tokens int32[total], offsets int64[n+1], labels int64[n])`` ready for the C ABI /
        ``FastSK.compute_kernel_flat`` instead of nested Python lists.
        """
        import numpy as np
        with open(data_file, "rb") as f:
            lines = [ln.strip().lower() for ln in f.read().splitlines()]
        if lines and not lines[-1] and len(lines) % 2:
            lines.pop()  # trailing blank line
        assert len(lines) % 2 == 0
        labels = np.empty(len(lines) // 2, dtype=np.int64)
        for i, ln in enumerate(lines[0::2]):
            parts = ln.split(b">")
            assert len(parts) == 2
            labels[i] = int(parts[1])
            assert labels[i] in (-1, 0, 1)
        seqs = lines[1::2]
        lens = np.fromiter((len(x) for x in seqs), dtype=np.int64, count=len(seqs))
        offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        raw = np.frombuffer(b"".join(seqs), dtype=np.uint8)
        if raw.size and raw.max() >= 128:  # non-ASCII: characters are not bytes, take the slow path
            X, Y = self.read_data(data_file)
            toks = np.fromiter((t for x in X for t in x), dtype=np.int32, count=int(offsets[-1]))
            return toks, offsets, np.asarray(Y, dtype=np.int64)
        # ids in first-seen order for the bytes not in the vocabulary yet
        vals, first = np.unique(raw, return_index=True)
        for b in vals[np.argsort(first)]:
            self._vocab.add(chr(int(b)))
        lut = np.zeros(256, dtype=np.int32)
        for b in vals:
            lut[int(b)] = self._vocab.add(chr(int(b)))
        return lut[raw], offsets, labels

    def shortest_seq(self, data_file):
        X, _ = self.read_data(data_file)
        return min(len(x) for x in X)
