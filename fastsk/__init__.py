"""Drop-in alias: ``from fastsk import FastSK, FastaUtility`` (reference ``src/fastsk/__init__.py:1-2``)
resolves to the MI355X-native engine in ``fastsk_amd``."""
from fastsk_amd import FastaUtility  # noqa: F401
from fastsk_amd import FastSK  # noqa: F401  (lazy attribute: settles which HIP runtime the process uses first)
