"""Drop-in alias of the reference's ``fastsk.utils`` (``src/fastsk/utils.py``):
``from fastsk.utils import FastaUtility, Vocabulary``."""
from fastsk_amd.utils import FastaUtility, Vocabulary  # noqa: F401
