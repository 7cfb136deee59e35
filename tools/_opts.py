"""Tool-side switchboard: `CMDGEN_OPTIONS="edge_mt=128,node64=0" python tools/<tool>.py ...` (or bench.py --option k=v) fills
hip_backend.DEFAULT_OPTIONS, which every new Handle applies through cmdgen_set_option.  The library itself reads no environment
variable; this file is the only place outside tests/ that turns one into options."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdgen_amd import hip_backend  # noqa: E402

hip_backend.DEFAULT_OPTIONS.update(hip_backend.parse_options(os.environ.get('CMDGEN_OPTIONS', '')))
