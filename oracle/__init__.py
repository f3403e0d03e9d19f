"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithm for the alignment hot path.  Nothing in the
product package ``ffsubsync_amd`` imports from here; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only as
the checker / the reported CPU baseline -- never as the thing measured or shipped.
"""
