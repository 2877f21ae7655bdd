"""CPU oracle for the collective hot path (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package; the shipped library never does.  See oracle/oracle.c for the citations.
"""
from .oracle import *  # noqa: F401,F403
