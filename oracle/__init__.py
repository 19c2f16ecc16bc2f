"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU checkers for the gapped-k-mer kernel path. Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product (``fastsk_amd``) never
does and fails loudly when its HIP library is missing.

Two checkers live here:

* ``oracle.port``  – ``liboracle.so``: our own plain-C restatement of the reference algorithm
  (``fastsk_oracle.c``; every function cites the reference file:line it follows).
* ``oracle.ref``   – ``_ref/libfastsk_ref.so``: the real reference compiled from its own sources
  (see ``Makefile`` / ``ref_harness.cpp``). Present when built in the container that holds
  ``/root/reference``; the prebuilt file travels to the GPU box.

Parity status: PINNED. ``liboracle.so`` is checked bit-for-bit against ``libfastsk_ref.so`` and
against the committed golden vectors in ``tests/golden`` (generated from the compiled reference by
``tests/make_golden.py``).
"""
from . import loader  # noqa: F401
from .loader import port, ref, have_ref, build  # noqa: F401
