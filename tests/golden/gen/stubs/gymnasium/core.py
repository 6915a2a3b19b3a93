"""Stub of gymnasium.core: only the typing aliases the reference imports."""
from typing import Any
ObsType = Any
ActType = Any
