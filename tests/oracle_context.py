"""the VoContext look-alike over the CPU oracle lives with the oracle (oracle/oracle_context.py: bench.py's cpu_baseline leg uses it too)"""
from oracle_context_impl import *  # noqa: F401,F403
