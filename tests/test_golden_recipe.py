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


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference only exists in the build container")
def test_reference_itself_rejects_hierarchical_resampling_of_fewer_than_three_samples():
    """fenerf_render_forward / fenerf_resample return FENERF_E_INVALID for hierarchical num_steps < 3.  So does the reference: the pdf
    is built on weights[:, 1:-1] (generators.py:492-497), which is empty for N <= 2, and sample_pdf's gather raises
    (volumetric_rendering.py:289-292).  Run in a child so that the reference's module names do not leak into this process."""
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from generators.volumetric_rendering import sample_pdf\n"
            "for N in (1, 2):\n"
            "    z = torch.linspace(0.9, 1.1, N).reshape(1, N); w = torch.rand(1, N) + 1e-5\n"
            "    try:\n"
            "        sample_pdf(0.5 * (z[:, :-1] + z[:, 1:]), w[:, 1:-1], N, det=False); print('runs', N)\n"
            "    except RuntimeError as e:\n"
            "        print('raises', N)\n"
            "z = torch.linspace(0.9, 1.1, 3).reshape(1, 3); w = torch.rand(1, 3) + 1e-5\n"
            "print('three', tuple(sample_pdf(0.5 * (z[:, :-1] + z[:, 1:]), w[:, 1:-1], 3, det=False).shape))\n") % REFERENCE
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split("\n")[:3] == ["raises 1", "raises 2", "three (1, 3)"], r.stdout
