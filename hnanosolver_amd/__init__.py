"""hnanosolver_amd -- MI355X-native (HIP, gfx950) implementation of HNanoSolver's per-substep hot path.

The product is ``lib/libhns.so`` (C ABI declared in ``include/hns.h``); this package is the thin host-side mirror
of the reference's operator interface used by the tests, the benchmark and the multi-GPU driver:

* :mod:`hnanosolver_amd.api`     -- ``GridIndexedData`` + ``CreateIndexGrid`` / ``Compute_Sim`` / ``AdvectIndexGrid`` /
  ``AdvectIndexGridVelocity`` / ``ProjectNonDivergent`` / ``Divergence`` (same names, argument meaning and error
  behaviour as the reference's ``extern "C"`` entry points, reference src/Cuda/*.cu).
* :mod:`hnanosolver_amd.device`  -- kernel-level calls on device-resident torch tensors.
* :mod:`hnanosolver_amd.dist`    -- leaf partition + halo exchange over torch.distributed (RCCL / gloo).
* :mod:`hnanosolver_amd.fields`  -- closed-form synthetic inputs and leaf sets for the benchmark configurations.

There is no CPU fallback: importing works anywhere, computing needs a HIP device and the built library.
"""
from ._lib import lib, load_library, HNSError, library_path, set_option, get_option  # noqa: F401

__all__ = ["lib", "load_library", "HNSError", "library_path", "set_option", "get_option"]
