from .grid import GridEncoder, grid_encode  # noqa: F401
