"""CPU oracle of the reference's allocate/backfill path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (kube-batch_amd) never imports it.
"""
from .oracle import Oracle, OracleRes, build, lib  # noqa: F401
