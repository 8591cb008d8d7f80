"""CPU oracle for the AMPConv hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.  The product
path (``ampnet_amd``) never imports this package and raises if its HIP
extension is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imported the reference's own
``src/ampnet/conv/amp_conv.py`` (unmodified, by file path) in the build
container and wrote the vectors under ``tests/golden/``; both restatements in
this package are checked against those vectors by ``tests/test_oracle.py``.
"""
