"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (torch fp32) restatement of the reference's contrastive training hot path
(Sense-GVT/DeCLIP, ``/root/reference``), plus the harness that imports the
*unmodified* reference in the build container to pin the restatement and to
generate the golden fixtures under ``tests/golden/``.

Nothing in the product (``declip_amd/``, ``prototype/``, ``linklink/``) may
import this package.  Allowed importers: ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker
or the timed CPU baseline, never as the thing measured or shipped.

Parity status: the reference ships no tests / golden vectors, so
parity is pinned by fixtures generated from the reference itself run on CPU in
the build container (``oracle/gen_golden.py`` -> ``tests/golden/*.pt``).
"""
