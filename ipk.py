#!/usr/bin/env python3
"""`ipk.py build ...` -- the reference's entry point (ipk.py:203-349), served by the MI355X engine.

The reference wrapper only assembles an argv for the ipk-dna / ipk-aa binaries; here the same options go to
ipk_amd.cli, which runs the stages this repository owns (see its docstring).  Several GPUs:
    torchrun --nproc-per-node 8 ipk.py build ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ipk_amd.cli import ipk  # noqa: E402

if __name__ == "__main__":
    ipk()
