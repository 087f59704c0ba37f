#!/usr/bin/env python3
"""Writes tests/golden/go_reference_input.json: the seeded scene go/pkg/scheduler/core/golden_dump_test.go feeds to the
REFERENCE's own functions (run that test next to core.go with Go 1.14 + modules; it writes go_reference_dump.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import go_golden  # noqa: E402

out = os.path.join(ROOT, "tests", "golden", "go_reference_input.json")
json.dump(go_golden.build_input(), open(out, "w"), indent=1)
print("wrote", out)
