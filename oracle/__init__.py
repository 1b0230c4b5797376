"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the per-protein ODE hot path.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker (never as the thing measured as the product or shipped).

Parity status: the reference (bibymaths/phoskintime) holds NO golden vectors or
known-answer tests for this path (its single test file collects zero tests), so the
oracle is pinned against outputs of the reference itself, generated in the build
container by ``tools/make_golden.py`` (which imports ``/root/reference`` read-only)
and committed as data under ``tests/golden/``.
"""
