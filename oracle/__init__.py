"""CPU oracle for the cross-domain hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A dependency-free (torch-CPU fp32 + numpy) restatement of the reference's
algorithm for every row of SURVEY.md section 8a.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker / the timed CPU baseline.  The
product (``recbole-cdr_amd/``) never imports it and has no CPU fallback.

Pinning status
--------------
* Pinned against the reference itself: every function here is asserted (tests/
  test_oracle_golden.py) against golden vectors produced by importing the
  reference's own model / dataset / dataloader classes from /root/reference
  (tests/golden/make_golden.py, committed with its output).
* "Parity unpinned" at one boundary: the reference depends on the un-vendored
  third-party package recbole==1.0.1 (requirements.txt:1), absent from this
  image.  ``BPRLoss``, ``EmbLoss``, ``MLPLayers`` and
  ``xavier_normal_initialization`` are restated from recbole 1.0.1's published
  source both in the generator's stand-in and here; no
  reference test or fixture pins them, so parity *of those four symbols* is
  unpinned.  All stock-torch semantics (MSELoss, BCELoss, TripletMarginLoss,
  Embedding, Linear, matmul, sparse.mm) are pinned.
"""
from . import losses, emcdr, cmf, conet, sscdr, bitgcf, remap  # noqa: F401
