"""CPU oracle for the fingerprint/match hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / the CPU timing
baseline.  The product (``shazam_amd``) never imports this package and fails
loudly when the HIP library is missing.

Parity pin: the reference (CarlosArturoMe/shazam) ships no tests or golden
vectors for this path.  The oracle is therefore pinned against
outputs of the reference itself, run in the build container by
``tests/golden/make_golden.py`` (numpy 2.2.6 / scipy 1.15.3 / matplotlib
3.10.8) and committed as fixtures under ``tests/golden/``.
"""
