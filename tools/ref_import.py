"""Import the read-only reference (/root/reference) in THIS container only.

Test infrastructure for generating golden fixtures (tools/make_golden.py).  The
reference is pure Python but drags in packages this image lacks (torchvision,
cv2, kornia, pytorch_fid) and a few imports that no longer exist in numpy 2 /
torch 2.10.  We inject empty stand-in *modules* for those
unrelated imports -- nothing on the rendering path touches them -- and then
import the reference's own, unmodified sources from where they lie.

Nothing in here (or anything it imports) ever travels to the GPU box: the `-m gpu`
tests, smoke() and bench.py only read the .npz fixtures this tooling emits.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FENERF_REFERENCE_ROOT", "/root/reference")


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__path__ = []  # behave like a package so "import a.b" works
    sys.modules[name] = mod
    return mod


def install_stubs():
    import numpy.lib  # noqa: F401
    import torch

    # siren/siren.py:2  `from numpy.lib.type_check import imag` (gone in numpy 2)
    if "numpy.lib.type_check" not in sys.modules:
        _stub("numpy.lib.type_check", imag=lambda x: x)
    # siren/siren.py:5  `from torch.functional import align_tensors`
    if not hasattr(torch.functional, "align_tensors"):
        torch.functional.align_tensors = lambda *a, **k: a
    # siren/siren.py:7  `from fid_evaluation import output_images` (torchvision, pytorch_fid)
    _stub("fid_evaluation", output_images=None)
    # generators/util.py:1,4,27
    _stub("cv2", COLORMAP_HOT=11, applyColorMap=None)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils", save_image=None, make_grid=None)
    # generators/neural_rendering.py:4
    k = _stub("kornia")
    k.filters = _stub("kornia.filters", filter2D=None)


def import_reference():
    """Returns (siren.siren, generators.generators, generators.volumetric_rendering, curriculums)."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT} (only exists in the build container)")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # make sure we do not pick up the repo's own drop-in packages of the same name
    for name in ("siren", "generators", "curriculums"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", "").startswith(REFERENCE_ROOT):
            del sys.modules[name]
    siren = importlib.import_module("siren.siren")
    gens = importlib.import_module("generators.generators")
    vr = importlib.import_module("generators.volumetric_rendering")
    cur = importlib.import_module("curriculums")
    return siren, gens, vr, cur
