"""Stand-in for the reference's pybind module ``dgsparse._C`` (src/version.cpp:11-21): on ROCm there is no
CUDA toolkit version to report, so ``cuda_version()`` returns -1, which makes the reference's version check
(dgsparse/__init__.py:28-42) a no-op.  ``hip_arch()`` / ``abi_version()`` describe the HIP build instead."""
from . import _capi


def cuda_version() -> int:
    return -1


def hip_arch() -> str:
    return _capi.arch()


def abi_version() -> int:
    return _capi.version()
