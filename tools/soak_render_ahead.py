#!/usr/bin/env python3
"""Regression soak of render-ahead: the random call sequences of tests/test_gpu_parity.py for 80 more seeds, invariants armed."""
import os, sys
os.environ["CT_DEBUG_INVARIANTS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import test_gpu_parity as t
class MP:
    def setenv(self, k, v): os.environ[k] = v
bad = 0
for seed in range(4, 84):
    try:
        t.test_render_ahead_under_random_call_sequences(seed, MP())
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED", repr(e)[:300], flush=True)
print("soak done, failures:", bad)
