import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def spec_from_golden(g):
    spec = {}
    for k, v in g.items():
        if k.startswith("spec_"):
            v = v.item() if hasattr(v, "item") and v.ndim == 0 else v
            spec[k[5:]] = str(v) if isinstance(v, (str, np.str_)) else (int(v) if k[5:] != "kind" else str(v))
    return spec


def film_from_golden(g, spec, batch=None):
    """The raw FiLM parameters a fixture was recorded with: procedural.film_params at the fixture's seed / scale and, for the
    fixtures made beyond the init range (round 4), its phase_rev / freq0_gain."""
    from fenerf_amd import procedural as proc
    extra = {}
    if "meta_film_phase_rev" in g:
        extra = dict(phase_rev=float(g["meta_film_phase_rev"]), freq0_gain=float(g["meta_film_freq0_gain"]))
    return proc.film_params(spec, int(g["meta_B"]) if batch is None else batch, seed=int(g["meta_seed"]), scale=float(g["meta_film_scale"]), **extra)


def kwargs_from_golden(g):
    kw = {}
    for k, v in g.items():
        if k.startswith("kw_"):
            v = v.item()
            kw[k[3:]] = v
    return kw


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
