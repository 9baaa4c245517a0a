"""The debug library with LDS guard bands (make -C csrc canary -> variants/lib_canary.so, built by __graft_entry__.build()): every
dynamic LDS region of the Fat-Llama kernels has a 64-byte sentinel band on either side, armed at kernel start, checked at exit.
Run in a subprocess with EGREGORA_AMD_LIB pointing at it: first the self-test (a write one element past / before the payload
MUST be counted, the last legal element must not), then one call of every plan kind of tests/test_gpu_fatllama_repeat.py --
packed two- and three-level, the scheduled C3 plan, relative threshold, all paired chirp-z forms, the legacy chirp-z, graph replay --
with results equal to the regular library's bit for bit and zero guard failures."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "variants" / "lib_canary.so"

CHILD = r'''
import hashlib, json, sys
import numpy as np, torch
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from packload import load_pack
load_pack()
from egregora_amd import fatllama_engine as fe, native
from test_gpu_fatllama import synth
from test_gpu_fatllama_repeat import CONFIGS
L = native.lib()
out = {"selftest": [int(L.egr_lds_canary_selftest(w)) for w in (0, 1, 2)], "digests": {}}
for name, C, n, f, iters, thr, kw in CONFIGS:
    if n > 3000000:
        continue
    x = torch.from_numpy(synth(C, n, seed=n + iters)).cuda()
    y = fe.enhance_device(x, f, iters, thr, True, False, True, True, **kw)
    out["digests"][name] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()
out["failures"] = int(L.egr_lds_canary_failures())
print("RESULT " + json.dumps(out))
'''


def _run(lib):
    env = dict(os.environ)
    if lib:
        env["EGREGORA_AMD_LIB"] = str(lib)
    else:
        env.pop("EGREGORA_AMD_LIB", None)
    r = subprocess.run([sys.executable, "-c", CHILD.replace("ROOT", repr(str(ROOT)))], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_guard_bands_are_placed_right_and_stay_intact_over_every_plan_kind():
    assert LIB.exists(), f"{LIB} missing: run __graft_entry__.build() (make -C csrc canary)"
    can = _run(LIB)
    ref = _run(None)
    assert ref["selftest"] == [-1, -1, -1] and ref["failures"] == -1         # the product library has no canaries
    assert can["selftest"] == [0, 1, 1], can["selftest"]                     # legal write: clean; past the end / before the start: caught
    assert can["failures"] == 2, can                                          # ... and nothing else tripped a guard
    assert can["digests"] == ref["digests"] and len(can["digests"]) >= 10
