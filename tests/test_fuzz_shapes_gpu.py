"""Random image sizes around the conv kernel's tile boundaries (tools/fuzz_shapes.py): small networks of the three
architectures, f32 engine against the oracle at the f32 bound and f16 engine at <= 1 LSB / >= 50 dB, 90 cases;
150 standalone convolutions (random channels, tiles, activations, residuals, dtypes) against torch conv2d and 150
bilinear u8 resizes against the oracle, bit-exact."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_against_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_shapes.py"), "90", "3", "150"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-1500:])
    assert r.returncode == 0
    assert "90 cases, seed 3, 0 failed" in r.stdout and "150 standalone convolutions and resizes, 0 failed" in r.stdout


def test_every_path_through_a_model_gives_the_same_bytes():
    """tools/fuzz_paths.py: lanes x frames per launch, the host-buffer path and single frames, 30 random batches."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_paths.py"), "30", "4"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-1500:])
    assert r.returncode == 0
    assert "30 cases, seed 4, 0 differences" in r.stdout
