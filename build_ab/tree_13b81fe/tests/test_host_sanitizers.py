"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side logic of the library (seeds, schedules, lattice
recogniser, adjacency, greedy colouring, quantisation of real couplings, tempering swap round, bit-plane expansion):
`tests/host_fuzz.cpp` drives `csrc/host_logic.cpp` with randomised inputs -- damaged lattices, multigraphs with self
loops, stars, complete graphs, couplings over 80 binary orders of magnitude, NaN stops, unaligned outputs -- and checks
the invariants the device code relies on.  GPU sanitizers are not available on the pool; this is the CPU build."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyisingmontecarlo_amd", "csrc")


def _build(tmp_path, flags):
    exe = str(tmp_path / "host_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-ffp-contract=off", "-I" + CSRC] + flags + [
        os.path.join(ROOT, "tests", "host_fuzz.cpp"), os.path.join(CSRC, "host_logic.cpp"), "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if out.returncode != 0 and ("cannot find -lasan" in out.stderr or "cannot find -lubsan" in out.stderr or "libasan" in out.stderr):
        pytest.skip("sanitizer runtimes are not installed with this g++")
    assert out.returncode == 0, out.stderr[-3000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_logic_under_asan_and_ubsan(tmp_path):
    exe = _build(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", ISINGMC_NO_AVX512="")
    env.pop("ISINGMC_NO_AVX512")
    for seed, extra_env in ((1, {}), (2, {"ISINGMC_NO_AVX512": "1"})):  # both bit-plane expanders
        out = subprocess.run([exe, "1500", str(seed)], env=dict(env, **extra_env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
        assert "host_fuzz: 1500 cases" in out.stdout and "runtime error" not in out.stderr


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_engines_under_asan_and_ubsan(tmp_path):
    """The checker itself: oracle/ising_oracle.c rebuilt with the sanitizers, and the oracle's own pins (Philox and xoshiro vectors,
    README energies, enumeration of engines A-E on small graphs, every lattice mode of engine B, golden vectors of the real-coupling spec) re-run against that
    build in a child interpreter with the ASan runtime preloaded."""
    lib = str(tmp_path / "liboracle_san.so")
    build = subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-Wall", "-Wextra",
                            "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared", "-o", lib,
                            os.path.join(ROOT, "oracle", "ising_oracle.c"), "-lm"], capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "asan" in build.stderr:
        pytest.skip("sanitizer runtimes are not installed with this gcc")
    assert build.returncode == 0, build.stderr[-3000:]
    runtime = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, ISING_ORACLE_LIB=lib, LD_PRELOAD=runtime, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run(["python", "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                          os.path.join(ROOT, "tests", "test_oracle_pins.py"), os.path.join(ROOT, "tests", "test_real_path_host.py"),
                          os.path.join(ROOT, "tests", "test_oracle_engines.py"),
                          "-k", "not kaufman_matches_enumeration"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_python_shim_under_asan_and_ubsan(tmp_path):
    """The pybind11 host shim (csrc/py_monte_carlo.cpp: edge ingest off the CPython objects, argument validation, bias
    bookkeeping, device lists) rebuilt with ASan + UBSan into a scratch copy of the package, and the API-surface tests -- which
    need no GPU -- re-run against it in a child interpreter with the ASan runtime preloaded."""
    import sysconfig
    pybind11 = pytest.importorskip("pybind11")
    pkg_src = os.path.join(ROOT, "pyisingmontecarlo_amd")
    lib = os.path.join(pkg_src, "lib", "libisingmc.so")
    if not os.path.exists(lib):
        pytest.skip("libisingmc.so is not built")
    scratch = tmp_path / "site"
    pkg = scratch / "pyisingmontecarlo_amd"
    pkg.mkdir(parents=True)
    for name in os.listdir(pkg_src):
        if name.endswith(".py"):
            shutil.copy(os.path.join(pkg_src, name), pkg / name)
    os.symlink(os.path.join(pkg_src, "lib"), pkg / "lib")
    shutil.copytree(os.path.join(ROOT, "py_monte_carlo"), scratch / "py_monte_carlo", ignore=shutil.ignore_patterns("__pycache__"))
    ext = pkg / ("_py_monte_carlo" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-fno-omit-frame-pointer",
                            "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include(), "-I" + os.path.join(ROOT, "include"),
                            "-o", str(ext), os.path.join(CSRC, "py_monte_carlo.cpp"), "-L" + os.path.join(pkg_src, "lib"), "-lisingmc",
                            "-Wl,-rpath," + os.path.join(pkg_src, "lib")], capture_output=True, text=True, timeout=900)
    if build.returncode != 0 and "asan" in build.stderr:
        pytest.skip("sanitizer runtimes are not installed with this g++")
    assert build.returncode == 0, build.stderr[-3000:]
    runtime = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("libasan.so not found")
    # libstdc++ must be loaded at start-up too: ASan's __cxa_throw interceptor looks the real function up when it initialises,
    # and the interpreter itself does not link the C++ runtime (the first exception of the shim would otherwise abort the child)
    cxx = subprocess.run(["g++", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
    if os.path.isabs(cxx) and os.path.exists(cxx):
        runtime = runtime + " " + cxx
    env = dict(os.environ, LD_PRELOAD=runtime, PYTHONPATH=os.pathsep.join([str(scratch), ROOT]),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    probe = subprocess.run(["python", "-c", "import py_monte_carlo, pyisingmontecarlo_amd._py_monte_carlo as m; print(m.__file__)"],
                           env=env, capture_output=True, text=True, timeout=600, cwd=str(scratch))     # cwd leads sys.path
    assert probe.returncode == 0 and str(scratch) in probe.stdout, (probe.stdout + probe.stderr)[-3000:]
    out = subprocess.run(["python", "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                          os.path.join(ROOT, "tests", "test_api_surface.py")], env=env, capture_output=True, text=True, timeout=1500,
                         cwd=str(scratch))
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
