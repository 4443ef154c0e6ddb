from . import siren  # noqa: F401
