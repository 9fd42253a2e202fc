from . import multiatlas  # noqa: F401
from . import cardiac  # noqa: F401
