"""CPU oracle for the xmca solve()/rotate()/rule_n() hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the
product package ``xmca_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.
"""
