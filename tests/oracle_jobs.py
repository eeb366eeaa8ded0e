"""The oracle-phase worker functions live in oracle/jobs.py (bench.py's checker section uses them too)."""
from oracle.jobs import *  # noqa: F401,F403
from oracle.jobs import _kw  # noqa: F401
