"""pred-occ-planner_amd — MI355X-native SOGM replan hot path (SOGM update -> hybrid A* ->
corridors -> Bezier QP), batched over agents, behind the C ABI in include/sogm_abi.h.

The directory name contains '-', so import it with::

    import importlib; pop = importlib.import_module("pred-occ-planner_amd")

Importing requires the in-tree HIP extension (libsogm_hip.so); there is no CPU fallback.
"""
from . import _abi, config, scene  # noqa: F401
from ._abi import SogmError, lib, load_library  # noqa: F401

__all__ = ["_abi", "config", "scene", "lib", "load_library", "SogmError"]
