from . import multiatlas  # noqa: F401
