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

    def shortest_seq(self, data_file):
        X, _ = self.read_data(data_file)
        return min(len(x) for x in X)
