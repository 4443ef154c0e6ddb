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


def state_from_golden(g):
    """Fixtures recorded at a state an optimiser produced (round 5, tools/make_golden.py::run_trained_state) name it in `meta_state`:
    -> (render weights {reference parameter name: array}, raw FiLM parameters) of that state fixture, or None."""
    if "meta_state" not in g:
        return None
    st = load_golden(str(g["meta_state"]))
    return ({k[3:]: v for k, v in st.items() if k.startswith("sd_")}, {k[5:]: v for k, v in st.items() if k.startswith("film_")})


def weights_from_golden(g, spec, with_mapping=True):
    """The reference-named state dict a fixture was recorded with: procedural weights at the fixture's seed / sigma_gain, with the render
    weights of its trained state (if it names one) on top -- the mapping networks stay procedural, as in tools/make_golden.py."""
    from fenerf_amd import procedural as proc
    sd = proc.make_state_dict(spec, seed=int(g["meta_seed"]), sigma_gain=float(g["meta_sigma_gain"]), with_mapping=with_mapping)
    st = state_from_golden(g)
    if st is not None:
        assert set(st[0]) <= set(sd), sorted(set(st[0]) - set(sd))
        sd.update(st[0])
    return sd


def film_from_golden(g, spec, batch=None):
    """The raw FiLM parameters a fixture was recorded with: procedural.film_params at the fixture's seed / scale and, for the
    fixtures made beyond the init range (round 4), its phase_rev / freq0_gain; the trained state's own (round 5) if it names one."""
    from fenerf_amd import procedural as proc
    st = state_from_golden(g)
    if st is not None:
        assert batch is None or batch == next(iter(st[1].values())).shape[0]
        return {k: v.copy() for k, v in st[1].items()}
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
