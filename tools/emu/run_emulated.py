#!/usr/bin/env python3
"""Runs test files written for the GPU against the library built for the host (no GPU):

    tools/emu/run_emulated.py tests/test_gpu_projection.py [pytest arguments]

The `gpu` marker is only a selection label; the tests call through the same C ABI.  Environment:
EMU_SHUFFLE=<seed> / EMU_REVERSE=1 (thread order), EMU_STRICT_LDS=1, OVS_EMU_UNITS=a.hip,b.hip.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import pytest  # noqa: E402

if __name__ == "__main__":
    sys.exit(pytest.main(["-p", "host_lib", "-q"] + sys.argv[1:]))
