"""CPU oracle for the receive-chain hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``sdr_amd/`` imports this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
only as the checker.  See ``oracle/dsp_oracle.c`` (GNU Radio block
restatement, parity unpinned) and ``oracle/decoders.py`` (numpy restatement of
the reference's own Python decoders, pinned by the reference's known-answer
vectors in ``tests/golden/``).
"""
