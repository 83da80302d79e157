"""In-tree build of the native parts (no pip, no JIT cache): the built files travel with the tree.

  lib/libisingmc.so                     HIP kernels + C ABI   (hipcc --offload-arch=gfx950)
  _py_monte_carlo.<abi>.so              C++ host shim: the reference's pyo3 surface (g++ + pybind11)
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libisingmc.so")
EXT = os.path.join(HERE, "_py_monte_carlo" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))

HIP_SOURCES = ["core.hip", "graph.hip", "isingmc.hip", "sampling.hip", "tempering.hip", "debug.hip", "strip_kernels.hip", "spread_kernels.hip", "mc_kernels.hip", "packed_uni_kernels.hip", "real_kernels.hip", "host_logic.cpp"]
HIP_DEPS = HIP_SOURCES + ["internal.hpp", "philox.hpp", "lattice_kernels.hpp", "strip_kernels.hpp", "strip_types.hpp", "spread_kernels.hpp", "spread_types.hpp", "mc_kernels.hpp", "mc_quad_body.inc", "mc_types.hpp", "general_kernels.hpp", "packed_kernels.hpp", "packed_types.hpp", "packed_uni_kernels.hpp", "real_kernels.hpp", "real_types.hpp", "host_logic.hpp",
                          os.path.join(ROOT, "include", "isingmc.h")]
EXT_SOURCES = ["py_monte_carlo.cpp"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d if os.path.isabs(d) else os.path.join(CSRC, d)) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_lib(force=False):
    """-ffp-contract=off: the general path's f64 arithmetic must round as written (oracle parity)."""
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _stale(LIB, HIP_DEPS):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
              "-Wall", "-Wextra", "-o", LIB] + [os.path.join(CSRC, s) for s in HIP_SOURCES])
    return LIB


def build_ext(force=False):
    build_lib(force)
    if force or _stale(EXT, EXT_SOURCES + [os.path.join(ROOT, "include", "isingmc.h"), LIB]):
        import pybind11
        inc = sysconfig.get_paths()["include"]
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
              "-I" + inc, "-I" + pybind11.get_include(), "-I" + os.path.join(ROOT, "include"),
              "-o", EXT] + [os.path.join(CSRC, s) for s in EXT_SOURCES] +
             ["-L" + LIBDIR, "-lisingmc", "-Wl,-rpath,$ORIGIN/lib"])
    return EXT


def build_all(force=False):
    build_lib(force)
    build_ext(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
