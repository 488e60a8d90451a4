"""path_optimizer_amd — MI355X-native batched QP path optimisation behind the reference's OsqpSolver boundary.

The product is `libpo_hip.so` (C ABI in include/po_hip.h, HIP kernels in csrc/); this package only holds the
ctypes plumbing (`binding`), the struct mirror (`abi`) and the synthetic BASELINE workloads (`synth`).
"""
from . import abi, synth  # noqa: F401

__all__ = ["abi", "synth", "binding"]
