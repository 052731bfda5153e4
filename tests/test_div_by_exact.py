"""The multiply + two-fma division by 9 / 3 of the packed photometric forward (csrc/loss.hip: div_by<C>) is the IEEE division:
tools/probes/div_by_check.c compares them bit for bit (here every 61st non-negative float; stride 1 = all of them, 30 s)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_div_by_constant_is_the_ieee_division(tmp_path):
    exe = str(tmp_path / "div_by_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tools", "probes", "div_by_check.c"), "-o", exe, "-lm"],
                   check=True)
    out = subprocess.run([exe, "61"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout
