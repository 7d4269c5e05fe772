"""Randomised sweeps over odd shapes, ranks and strided views (tools/stress_*.py) with fixed seeds: every case must agree with
the oracle / the fp64 device evaluation (deterministic kernels: the outcome does not depend on the box).  Seeds 0 / 1 / 2 are
the ones the sweeps were developed with; seed 81 of stress_parity.py holds the one case found so far (case 43: 128 x 130,
rank 128, sparse HALS) where an inner solve stops one sweep later than in the fp64 oracle -- threshold noise, analysed in
tools/probes/seed81_probe.py and DESIGN.md section 4: the tool classifies it from the ORACLE's own eps / (delta eps0) at that
sweep (0.99968: within 2e-3 of the threshold) and applies the documented looser bound from that solve on; anything else
-- a count off by more than one, or off by one away from the threshold -- is still flagged.  stress_stop.py: the iteration a HALS run
stops at (`tol` placed between two cost differences of a pilot run) with the Gram-identity cost against NNF_COST=direct -- same
length, bitwise equal factors, the two costs the test fired on bitwise those of the direct run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed,cases", [("stress_kernels.py", 0, 60), ("stress_tensor.py", 1, 60),
                                             ("stress_parity.py", 2, 60), ("stress_parity.py", 81, 60), ("stress_parity.py", 5, 60),
                                             ("stress_hals.py", 0, 150), ("stress_hals.py", 3, 150),
                                             ("stress_stop.py", 0, 40), ("stress_stop.py", 2, 40)])
def test_randomised_sweep(built_lib, tool, seed, cases):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed), str(cases)], capture_output=True,
                         text=True, timeout=600)
    tail = "\n".join(out.stdout.strip().splitlines()[-12:])
    assert out.returncode == 0, tail + out.stderr[-2000:]
    assert f"{cases} cases, 0 flagged" in out.stdout, tail
