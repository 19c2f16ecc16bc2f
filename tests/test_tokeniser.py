"""The native FASTA reader (fsk_read_fasta behind fastsk.utils.FastaUtility) against the
reference's reader: token arrays captured from the reference for the four FASTA configs of
BASELINE.json (tests/golden/tokens_*.npz; needs the data files, build container only) and for
small messy files of our own (tests/golden/fasta/, travels everywhere). SURVEY 8f-2."""
import os

import numpy as np
import pytest

from conftest import GOLD

REF_DATA = "/root/reference/data"
FASTA = os.path.join(GOLD, "fasta")


def flat(X):
    return np.array([t for x in X for t in x], dtype=np.int32), np.array([len(x) for x in X], dtype=np.int64)


@pytest.mark.parametrize("name", ["EP300", "EP300_47848", "1.1", "2.19", "small"])
@pytest.mark.parametrize("api", ["read_data", "read_packed"])
def test_reference_datasets(name, api):
    tr, te = (os.path.join(REF_DATA, "%s.%s.fasta" % (name, part)) for part in ("train", "test"))
    if not (os.path.exists(tr) and os.path.exists(te)):
        pytest.skip("reference data files not present on this box")
    from fastsk.utils import FastaUtility  # the reference's import path
    z = np.load(os.path.join(GOLD, "tokens_%s.npz" % name))
    rd = FastaUtility()
    if api == "read_data":
        (Xa, Ya), (Xb, Yb) = rd.read_data(tr), rd.read_data(te)
        (ta, la), (tb, lb) = flat(Xa), flat(Xb)
        ya, yb = np.array(Ya), np.array(Yb)
        assert isinstance(Xa, list) and isinstance(Xa[0], list) and isinstance(Ya[0], int)
    else:
        (ta, oa, ya), (tb, ob, yb) = rd.read_packed(tr), rd.read_packed(te)
        la, lb = np.diff(oa), np.diff(ob)
        assert ta.dtype == np.int32 and oa.dtype == np.int64 and oa[0] == 0
    assert len(la) == int(z["n_train"]) and len(lb) == int(z["n_test"])
    assert np.array_equal(np.concatenate([ta, tb]), z["tokens"])
    assert np.array_equal(np.concatenate([[0], np.cumsum(np.concatenate([la, lb]))]), z["offsets"])
    assert np.array_equal(ya, z["y_train"]) and np.array_equal(yb, z["y_test"])


@pytest.mark.parametrize("api", ["read_data", "read_packed"])
def test_messy_files_match_the_reference_reader(api):
    """Mixed case, CRLF / lone CR, padding, an empty sequence, labels -1/0/1, symbols outside the
    alphabet, no final newline; one vocabulary shared by the train and the test file."""
    from fastsk_amd.utils import FastaUtility
    want = np.load(os.path.join(FASTA, "expected.npz"))
    for group in (["messy.train.fasta", "messy.test.fasta"], ["protein.train.fasta"]):
        rd = FastaUtility()
        for name in group:
            path = os.path.join(FASTA, name)
            if api == "read_data":
                X, Y = rd.read_data(path)
                toks, lens = flat(X)
                labels = np.array(Y, dtype=np.int64)
            else:
                toks, off, labels = rd.read_packed(path)
                lens = np.diff(off)
            assert np.array_equal(toks, want[name + ":tokens"]), name
            assert np.array_equal(lens, want[name + ":lengths"]), name
            assert np.array_equal(labels, want[name + ":labels"]), name
        assert rd.shortest_seq(os.path.join(FASTA, group[0])) == int(want[group[0] + ":lengths"].min())
    assert rd._vocab.size() == 11  # id 0 + ten distinct residues


def test_malformed_files_fail_where_the_reference_asserts(tmp_path):
    from fastsk_amd.utils import FastaUtility, MalformedFasta
    cases = {"trailing_blank": b">1\nacgt\n\n", "label_2": b">2\nacgt\n", "no_marker": b"1\nacgt\n",
             "two_markers": b">>1\nacgt\n", "odd_lines": b">1\nacgt\n>0\n", "text_label": b">x\nacgt\n"}
    for name, data in cases.items():
        p = tmp_path / (name + ".fasta")
        p.write_bytes(data)
        with pytest.raises(AssertionError):  # the reference: a failed assert (int(): ValueError, also covered)
            try:
                FastaUtility().read_data(str(p))
            except ValueError as exc:
                assert isinstance(exc, MalformedFasta)
                raise
    with pytest.raises(OSError):
        FastaUtility().read_packed(str(tmp_path / "missing.fasta"))


def test_regression_labels_and_non_ascii(tmp_path):
    from fastsk_amd.utils import FastaUtility
    p = tmp_path / "reg.fasta"
    p.write_bytes(b">0.75\nACGT\n>-1.5e3\nggca\n")
    X, Y = FastaUtility().read_data(str(p), regression=True)
    assert X == [[1, 2, 3, 4], [3, 3, 2, 1]] and Y == ["0.75", "-1.5e3"]
    q = tmp_path / "greek.fasta"
    q.write_text(">1\nαβαA\n>0\nβa\n", encoding="utf-8")
    rd = FastaUtility()
    X, Y = rd.read_data(str(q))
    assert X == [[1, 2, 1, 3], [2, 3]] and Y == [1, 0]
    toks, off, lab = rd.read_packed(str(q))  # same object: ids stay
    assert toks.tolist() == [1, 2, 1, 3, 2, 3] and off.tolist() == [0, 4, 6] and lab.tolist() == [1, 0]


def test_reader_without_the_native_library(monkeypatch):
    """On a host where libfastsk_amd.so is not built the tokeniser falls back to its numpy pass over the
    text: same tokens, offsets, labels and vocabulary as the native reader."""
    from fastsk_amd import _native
    from fastsk_amd.utils import FastaUtility
    want = np.load(os.path.join(FASTA, "expected.npz"))

    def missing():
        raise ImportError("libfastsk_amd.so not found")
    monkeypatch.setattr(_native, "library", missing)
    rd = FastaUtility()
    for name in ("messy.train.fasta", "messy.test.fasta"):
        toks, off, labels = rd.read_packed(os.path.join(FASTA, name))
        assert np.array_equal(toks, want[name + ":tokens"]) and np.array_equal(np.diff(off), want[name + ":lengths"])
        assert np.array_equal(labels, want[name + ":labels"])
    assert rd._vocab.size() == 12  # id 0 + the eleven symbols of the two messy files
