"""The fuzz scripts of tools/ as a (short) part of the GPU tier: a dozen seeded random cases each, so that the scripts
stay runnable and the shapes they draw keep meeting the parity gates.  The long runs are in profiles/r03_fuzz.txt."""
import os
import subprocess
import sys

import pytest

from conftest import need_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script,cases,seed", [("fuzz_encoder.py", 12, 1234), ("fuzz_dropout.py", 10, 2468),
                                               ("fuzz_head.py", 40, 99), ("fuzz_dist.py", 15, 4321)])
def test_fuzz_script_cases_meet_the_gates(script, cases, seed):
    need_gpu()
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(cases), str(seed)], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    lines = run.stdout.splitlines()
    failed = [ln for ln in lines if ln.startswith("FAIL")]
    assert run.returncode == 0, run.stderr[-2000:]
    assert not failed, "\n".join(failed)
    assert sum(ln.startswith("ok") for ln in lines) == cases
