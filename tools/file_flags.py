#!/usr/bin/env python3
"""The extra hipcc flags __graft_entry__.build() gives one source file (the single table of per-file flags: tools/build_*.sh ask here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
print(' '.join(dict(g.SOURCES).get(os.path.basename(sys.argv[1]), [])))
