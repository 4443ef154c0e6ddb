"""Import-path aliases so code (and pickles) written against the reference resolve to this package:
`import curriculums`, `from generators import generators`, `from siren import siren`
(pickled checkpoints embed `generators.generators.DoubleImplicitGenerator3d` / `siren.siren.*`); the pickled
`*_ema.pth` objects embed `torch_ema.ema.ExponentialMovingAverage`, served by fenerf_amd.ema when torch_ema is absent."""
import importlib.util
import sys


def install_aliases():
    from . import curriculums, generators, siren
    sys.modules.setdefault("curriculums", curriculums)
    sys.modules.setdefault("generators", generators)
    sys.modules.setdefault("generators.generators", generators.generators)
    sys.modules.setdefault("generators.volumetric_rendering", generators.volumetric_rendering)
    sys.modules.setdefault("generators.math_utils_torch", generators.math_utils_torch)
    sys.modules.setdefault("siren", siren)
    sys.modules.setdefault("siren.siren", siren.siren)
    # pickled SPATIALSIRENGRID modules embed siren.latent_grid.StyleGenerator2D and the siren.layers classes it is built from
    from .siren import latent_grid
    sys.modules.setdefault("siren.latent_grid", latent_grid)
    sys.modules.setdefault("siren.layers", latent_grid)
    sys.modules.setdefault("siren.op", latent_grid)
    sys.modules.setdefault("siren.op.native_ops", latent_grid)
    # a checkpoint pickled where the reference's CUDA ops did import (siren/op/__init__.py:1-4) names FusedLeakyReLU under
    # siren.op.fused_act; the only nn.Module those two files define is that one (the autograd Functions are never pickled)
    sys.modules.setdefault("siren.op.fused_act", latent_grid)
    sys.modules.setdefault("siren.op.upfirdn2d", latent_grid)
    if "torch_ema" not in sys.modules and importlib.util.find_spec("torch_ema") is None:
        from . import ema
        sys.modules["torch_ema"] = ema
        sys.modules["torch_ema.ema"] = ema
