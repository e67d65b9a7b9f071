"""Module name of the reference (poreplex/io.py); the implementation lives in sinks.py."""
from .sinks import *  # noqa: F401,F403
from .sinks import __all__  # noqa: F401
