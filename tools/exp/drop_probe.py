#!/usr/bin/env python3
"""PRIZE PROBE (results garbage by design): bench.py with the launches named in DROP=<abi name>,... replaced by nothing -- the upper
bound of what folding those launches into their neighbours could return.   DROP=bn_act_bwd_reduce python tools/exp/drop_probe.py --steps 100"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hypelcnn_amd import backend  # noqa: E402

drop = set(filter(None, os.environ.get("DROP", "").split(",")))
orig = backend.HipBackend.bind


def bind(self, name, args, stream=None):
    if name in drop:
        return lambda: 0
    return orig(self, name, args, stream)


backend.HipBackend.bind = bind
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
