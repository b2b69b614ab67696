"""pred-occ-planner_amd — MI355X-native SOGM replan hot path (SOGM update -> hybrid A* ->
corridors -> Bezier QP), batched over agents, behind the C ABI in include/sogm_abi.h.

The directory name contains '-', so import it with::

    import importlib; pop = importlib.import_module("pred-occ-planner_amd")

Importing requires the in-tree HIP extension (libsogm_hip.so); there is no CPU fallback.
"""
import os as _os

# sogm_replan runs its kernels on up to nine HIP streams; streams beyond the number of hardware queues share a queue
# and serialise (ROCm's default is four; RCCL's channels take queues as well: 16 was too few beside it).  Must be in the environment before HIP initialises, i.e. before the first
# torch.cuda / hip call of the process: a host that initialises HIP earlier sets it itself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

from . import _abi, config, scene  # noqa: E402,F401
from ._abi import SogmError, lib, load_library  # noqa: E402,F401

__all__ = ["_abi", "config", "scene", "lib", "load_library", "SogmError"]
