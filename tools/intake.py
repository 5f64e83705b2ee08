#!/usr/bin/env python3
"""Launcher for tests/intake.py -- the one-command intake for real circom / snarkjs artefacts.  The tool itself
is test infrastructure (it runs the oracle's circom interpreter as the checker), so it lives under tests/; this file only forwards:

    python tools/intake.py --node-modules NM --build-dir BUILD --input input.json [--wtns witness.wtns] ...
"""
import os
import runpy
import sys

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    sys.argv[0] = os.path.join(os.path.dirname(here), "tests", "intake.py")
    runpy.run_path(sys.argv[0], run_name="__main__")
