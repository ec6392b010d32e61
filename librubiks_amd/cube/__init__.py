"""`from librubiks_amd import cube` mirrors `from librubiks import cube` (reference: librubiks/cube/__init__.py:2)."""
from .cube import *  # noqa: F401,F403
from .cube import device  # noqa: F401
