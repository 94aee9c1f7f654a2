from .corner import create_corner_treatment  # noqa: F401
