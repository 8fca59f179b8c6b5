"""Randomised differential run of the GPU path against the oracle (tools/fuzz_gpu.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fuzz_against_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "400", "20251111"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fails 0" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode,trials,seed", [("lds", 120, 20261112), ("origin", 150, 20261113)])
def test_fuzz_other_shapes(mode, trials, seed):
    """The same differential run aimed at the shapes the in-LDS kernels take (blocks of 16-256 KiB through the
    batch path, small and byte alphabets) and at the origin-pointer format (sorted rotations, forward and back)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), str(trials), str(seed), mode], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fails 0" in out.stdout
