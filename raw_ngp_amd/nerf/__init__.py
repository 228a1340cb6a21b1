from .options import Options  # noqa: F401
