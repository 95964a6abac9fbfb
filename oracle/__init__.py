"""CPU oracle for the RMCKF hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(AI-SPARC/uncalibrated-visual-servoing: experiment.py / noise.py /
ur10_simulation.py / utils.py) that exists to *check* the HIP path.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package never imports it and fails
loudly when its HIP library is missing.

Parity pin: the reference ships no golden vectors or known-answer tests for
this path, so the oracle is pinned against outputs of the
reference itself, run unmodified in the build container by
``oracle/gen_golden.py`` and committed as ``tests/golden/*.npz``.
"""
