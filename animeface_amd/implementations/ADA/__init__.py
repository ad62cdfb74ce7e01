from .utils import main  # noqa: F401
