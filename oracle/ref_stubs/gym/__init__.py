"""Import stub for `gym` (agent.py:5 only needs gym.spaces.Discrete/Box with ==)."""
from . import spaces  # noqa: F401
