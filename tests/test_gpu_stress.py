"""Randomised sweeps over odd shapes, ranks and strided views (tools/stress_*.py) with fixed seeds: every case must agree with
the oracle / the fp64 device evaluation.  The seeds are the ones the sweeps were developed with (deterministic kernels: the
outcome does not depend on the box)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed,cases", [("stress_kernels.py", 0, 60), ("stress_tensor.py", 1, 60),
                                             ("stress_parity.py", 2, 60)])
def test_randomised_sweep(built_lib, tool, seed, cases):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed), str(cases)], capture_output=True,
                         text=True, timeout=600)
    tail = "\n".join(out.stdout.strip().splitlines()[-12:])
    assert out.returncode == 0, tail + out.stderr[-2000:]
    assert f"{cases} cases, 0 flagged" in out.stdout, tail
