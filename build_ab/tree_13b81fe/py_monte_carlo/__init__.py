"""Drop-in module name of the reference (src/lib.rs:14-22): `import py_monte_carlo`.

Classical classes only (Lattice, ClassicIsing); the quantum SSE classes of the reference
(QmcIsing, QmcRunner, LatticeTempering) are out of scope of this build.  ClassicalTempering is the
build's classical beta-ladder shaped after LatticeTempering (tempering.rs).
"""
from pyisingmontecarlo_amd import load_extension as _load

_ext = _load()
Lattice = _ext.Lattice
ClassicIsing = _ext.ClassicIsing

from pyisingmontecarlo_amd.tempering import ClassicalTempering  # noqa: E402

__all__ = ["Lattice", "ClassicIsing", "ClassicalTempering"]

_QUANTUM_ONLY = ("QmcIsing", "QmcRunner", "LatticeTempering")  # src/lib.rs:16-21


def __getattr__(name):
    if name in _QUANTUM_ONLY:
        raise NotImplementedError(f"py_monte_carlo.{name}: quantum (SSE) Monte Carlo is not part of this build "
                                  "(classical Metropolis only: Lattice, ClassicIsing, ClassicalTempering)")
    raise AttributeError(f"module 'py_monte_carlo' has no attribute {name!r}")
