"""CPU oracle for the LynseDB FLAT / IVF-Flat hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``lynsedb_amd``) never does.
"""
from .oracle import *  # noqa: F401,F403
