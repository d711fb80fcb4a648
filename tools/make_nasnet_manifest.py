#!/usr/bin/env python3
"""Writes tests/golden/nasnet_mobile_manifest.json: every variable (keras name, shape) of
tf.keras.applications.NASNetMobile(include_top=False) in creation order, from the independent restatement of the
published architecture in oracle/ref_nasnet.py (no TensorFlow needed, none available), plus the structural facts the
tests pin (4,269,716 elements, 188 unnamed activations, the five tapped activations' shapes at 128 x 416)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_nasnet  # noqa: E402

variables, info = ref_nasnet.manifest(128, 416)
out = {"model": "tf.keras.applications.NASNetMobile(include_top=False), tensorflow==2.4.1 (requirements.txt:41)",
       "generated_by": "tools/make_nasnet_manifest.py from oracle/ref_nasnet.py",
       "input": [128 + 2, 416 + 2, 3], **info,
       "variables": [[name, list(shape)] for name, shape in variables.items()]}
path = os.path.join(ROOT, "tests", "golden", "nasnet_mobile_manifest.json")
with open(path, "w") as f:                     # one variable per line (reviewable diffs)
    head = {k: v for k, v in out.items() if k != "variables"}
    f.write(json.dumps(head)[:-1] + ', "variables": [\n')
    f.write(",\n".join(json.dumps(v) for v in out["variables"]))
    f.write("\n]}\n")
print(path, len(variables), "variables,", info["total_elements"], "elements")
