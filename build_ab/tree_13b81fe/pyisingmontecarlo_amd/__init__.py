"""MI355X-native classical Ising Metropolis engine (hot path of Renmusxd/PyIsingMonteCarlo).

  _capi            ctypes binding of the C ABI (include/isingmc.h, lib/libisingmc.so)
  _py_monte_carlo  C++ host shim with the reference's `py_monte_carlo` classes (Lattice, ClassicIsing)
  distributed      one-process-per-GPU replica sharding + parallel tempering over torch.distributed

The HIP library is mandatory: nothing in this package computes a spin flip on the CPU.
"""
from . import _capi  # noqa: F401

__all__ = ["_capi", "load_extension"]


def load_extension():
    """Import the C++ host shim; raises with a build hint when it has not been built."""
    _capi.preload_hip_runtime()  # before the extension pulls in libisingmc.so -> libamdhip64
    try:
        from . import _py_monte_carlo
    except ImportError as exc:  # pragma: no cover - build problem
        raise ImportError(
            "pyisingmontecarlo_amd._py_monte_carlo is not built: run `python -m pyisingmontecarlo_amd.build` "
            f"({exc})") from exc
    return _py_monte_carlo
