"""Flat-import shim: the reference's scripts do `import LogHandler` with src/ on
sys.path (src/trainer.py:20-31).  The launcher
ss_asr_amd/run_reference.py puts this directory at sys.path[0] and runs the reference's
src/train.py unchanged against the MI355X implementation (PYTHONPATH alone is not enough: the
script's own directory comes first on sys.path -- INTEGRATION.md section 1)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from ss_asr_amd.LogHandler import *  # noqa: E402,F401,F403
from ss_asr_amd import LogHandler as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
