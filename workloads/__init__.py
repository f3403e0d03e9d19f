"""Seeded synthetic workloads (benchmark / test inputs).  Not part of the product package:
``ffsubsync_amd`` never imports this; bench.py, tests/, profiles/ and oracle/ helpers do."""
