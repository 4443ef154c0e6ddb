from . import generators, volumetric_rendering, math_utils_torch  # noqa: F401
