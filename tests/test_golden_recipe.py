"""The pin is self-checking in the build container: when the reference is present (/root/reference -- never on the GPU box),
re-run tools/make_golden.py end to end AS A SCRIPT into a scratch directory and require every committed fixture to come out
bit for bit (arrays) / value for value (curriculums.json).  Skipped where the reference does not exist."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("FENERF_REFERENCE_ROOT", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference only exists in the build container")
def test_committed_fixtures_regenerate_from_the_reference(tmp_path):
    out = str(tmp_path / "golden")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_golden.py"), out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    committed = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert len(committed) >= 19
    for f in committed:
        g = os.path.join(out, os.path.basename(f))
        assert os.path.exists(g), f"make_golden.py no longer writes {os.path.basename(f)}"
        a, b = np.load(f, allow_pickle=True), np.load(g, allow_pickle=True)
        assert set(a.files) == set(b.files), os.path.basename(f)
        for k in a.files:
            assert np.array_equal(a[k], b[k]), f"{os.path.basename(f)}:{k} differs from what the reference produces today"
    assert json.load(open(os.path.join(GOLDEN, "curriculums.json"))) == json.load(open(os.path.join(out, "curriculums.json")))
    assert os.path.exists(os.path.join(out, "ref_generator_tiny.pth"))
