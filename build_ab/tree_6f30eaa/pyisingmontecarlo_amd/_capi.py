"""ctypes binding of include/isingmc.h (libisingmc.so).  No fallback of any kind: if the HIP library
is missing or no device is usable, calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ISINGMC_LIB_PATH") or os.path.join(_HERE, "lib", "libisingmc.so")  # override: A/B builds

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC = range(5)
KIND_GENERAL, KIND_LATTICE2D = 0, 1
FLAG_FORCE_GENERAL = 1
FLAG_STABLE_PATH = 2


class GraphInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("device", C.c_int32), ("nvars", C.c_uint64), ("n_edges", C.c_uint64),
                ("width", C.c_int32), ("height", C.c_int32), ("jabs", C.c_double), ("uniform_sign", C.c_int32),
                ("n_colours", C.c_uint32), ("state_words", C.c_uint64), ("fast_path", C.c_int32), ("open_x", C.c_int32),
                ("open_y", C.c_int32), ("field", C.c_double), ("jabs_y", C.c_double),
                ("field_signs", C.c_int32), ("packed_degree", C.c_int32), ("real_slots", C.c_int32),
                ("real_quantum_log2", C.c_int32), ("real_energy_log2", C.c_int32), ("real_heavy_sites", C.c_int32),
                ("stable_path", C.c_int32)]


_vp = C.c_void_p
_PROTOTYPES = {
    "isingmc_last_error": (C.c_char_p, []),
    "isingmc_abi_version": (C.c_int, []),
    "isingmc_release_cached_resources": (C.c_size_t, []),
    "isingmc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "isingmc_host_make_seeds": (C.c_int, [C.c_int, C.c_uint64, C.c_size_t, _vp]),
    "isingmc_host_expand_schedule": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_int, _vp]),
    "isingmc_host_recognise_lattice2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_int),
                                                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                                   C.POINTER(C.c_int)]),
    "isingmc_host_colour_graph": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, C.POINTER(C.c_uint32)]),
    "isingmc_host_pt_swap_round": (C.c_int, [C.c_uint64, C.c_uint64, C.c_size_t, _vp, _vp, _vp,
                                             C.POINTER(C.c_uint64)]),
    "isingmc_host_rj_quantise": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp, C.POINTER(C.c_int),
                                           C.POINTER(C.c_int)]),
    "isingmc_host_rj_energy_levels": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int)]),
    "isingmc_host_rj_beta": (C.c_int, [C.c_double, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "isingmc_host_rj_log_table": (C.c_int, [_vp]),
    "isingmc_graph_create": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_int, C.c_uint, C.POINTER(_vp)]),
    "isingmc_graph_info": (C.c_int, [_vp, C.POINTER(GraphInfo)]),
    "isingmc_graph_destroy": (None, [_vp]),
    "isingmc_states_create": (C.c_int, [_vp, C.c_size_t, _vp, _vp, C.POINTER(_vp)]),
    "isingmc_states_create_range": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, _vp, C.POINTER(_vp)]),
    "isingmc_states_append": (C.c_int, [_vp, C.c_uint64, _vp]),
    "isingmc_states_set_state": (C.c_int, [_vp, C.c_size_t, _vp]),
    "isingmc_states_count": (C.c_size_t, [_vp]),
    "isingmc_states_destroy": (None, [_vp]),
    "isingmc_states_set_betas": (C.c_int, [_vp, _vp]),
    "isingmc_do_time_steps": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "isingmc_do_time_steps_timed": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_float)]),
    "isingmc_get_energies": (C.c_int, [_vp, _vp]),
    "isingmc_get_magnetisations": (C.c_int, [_vp, _vp]),
    "isingmc_get_states": (C.c_int, [_vp, _vp, C.c_size_t]),
    "isingmc_get_packed_states": (C.c_int, [_vp, _vp]),
    "isingmc_states_timestep": (C.c_uint64, [_vp]),
    "isingmc_states_set_timestep": (C.c_int, [_vp, C.c_uint64]),
    "isingmc_run_sampling": (C.c_int, [_vp, C.c_double, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp]),
    "isingmc_pt_attach": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64]),
    "isingmc_pt_can_attach": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]),
    "isingmc_pt_detach": (C.c_int, [_vp]),
    "isingmc_pt_buffers": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "isingmc_pt_time_steps": (C.c_int, [_vp, C.c_size_t]),
    "isingmc_pt_measure": (C.c_int, [_vp]),
    "isingmc_pt_run": (C.c_int, [_vp, C.c_size_t, C.c_size_t]),
    "isingmc_pt_swap": (C.c_int, [_vp]),
    "isingmc_pt_state": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "isingmc_states_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "isingmc_synchronize": (C.c_int, [_vp]),
    "isingmc_debug_shader_clock": (C.c_int, [_vp, C.c_size_t, C.c_double, C.c_double, C.POINTER(C.c_double)]),
}
EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None
_hip_preloaded = False


def preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Whichever copy a
    process loads first serves both torch and libisingmc.so; if the system copy wins, torch's other bundled
    ROCm libraries no longer match it and torch.cuda reports no GPU.  So when torch is installed its copy
    is loaded first (without importing torch); libisingmc.so then binds to that one."""
    global _hip_preloaded
    if _hip_preloaded:
        return
    _hip_preloaded = True
    if os.environ.get("ISINGMC_NO_TORCH_HIP_PRELOAD"):  # diagnostics: run on the system ROCm runtime
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:  # pragma: no cover - torch absent or laid out differently: use the system runtime
        pass


def lib():
    global _lib
    if _lib is None:
        preload_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m pyisingmontecarlo_amd.build` "
                "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.isingmc_abi_version() != 4:
            raise RuntimeError("libisingmc.so ABI version mismatch")
        _lib = L
    return _lib


def _check(rc):
    if rc == OK:
        return
    msg = (lib().isingmc_last_error() or b"").decode()
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_ALLOC:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _arr(x, dtype):
    return None if x is None else np.ascontiguousarray(x, dtype=dtype)


def device_count():
    n = C.c_int(0)
    rc = lib().isingmc_device_count(C.byref(n))
    return n.value if rc == OK else 0


def make_seeds(seed_gen, n):
    out = np.zeros(n, dtype=np.uint64)
    _check(lib().isingmc_host_make_seeds(int(seed_gen is not None), C.c_uint64(seed_gen or 0), n, _p(out)))
    return out


def expand_schedule(stops, timesteps, compat_constant_beta=False):
    t = _arr([s[0] for s in stops], np.uint64)
    b = _arr([s[1] for s in stops], np.float64)
    out = np.zeros(timesteps, dtype=np.float64)
    _check(lib().isingmc_host_expand_schedule(_p(t), _p(b), len(stops), timesteps, int(compat_constant_beta), _p(out)))
    return out


def recognise_lattice2d(ea, eb, ej, nvars):
    ea, eb, ej = _arr(ea, np.uint64), _arr(eb, np.uint64), _arr(ej, np.float64)
    ok, w, h, u, jabs = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
    _check(lib().isingmc_host_recognise_lattice2d(_p(ea), _p(eb), _p(ej), len(ea), nvars, C.byref(ok), C.byref(w),
                                                  C.byref(h), C.byref(jabs), C.byref(u)))
    out = dict(is_lattice=bool(ok.value), width=w.value, height=h.value, jabs=jabs.value, uniform_sign=bool(u.value))
    if ok.value > 0 and (ok.value - 1) & 6:  # open boundaries (all wrap-around bonds of a direction absent)
        out.update(open_x=bool((ok.value - 1) & 2), open_y=bool((ok.value - 1) & 4))
    if ok.value > 0 and (ok.value - 1) & 8:  # jabs is the horizontal bonds' |J|, the vertical bonds have another
        out.update(anisotropic=True)
    return out


def rj_quantise(ea, eb, ej, nvars, biases=None):
    """(k, jq[n_edges, 2] (each bond as seen from its two ends), hq per site, dshift per site, eligible) of the real-coupling
    packed path."""
    ea, eb, ej, b = _arr(ea, np.uint64), _arr(eb, np.uint64), _arr(ej, np.float64), _arr(biases, np.float64)
    jq, hq, d = np.zeros((len(ea), 2), dtype=np.int32), np.zeros(nvars, dtype=np.int32), np.zeros(nvars, dtype=np.uint8)
    k, ok = C.c_int(), C.c_int()
    _check(lib().isingmc_host_rj_quantise(_p(ea), _p(eb), _p(ej), len(ea), nvars, _p(b), _p(jq), _p(hq), _p(d), C.byref(k), C.byref(ok)))
    return k.value, jq, hq, d, bool(ok.value)


def rj_energy_levels(ea, eb, ej, nvars, biases=None):
    """(kE, jhi, jlo per input edge, hhi, hlo per site): the two integer levels of that path's energies."""
    ea, eb, ej, b = _arr(ea, np.uint64), _arr(eb, np.uint64), _arr(ej, np.float64), _arr(biases, np.float64)
    jhi, jlo = np.zeros(len(ea), dtype=np.int32), np.zeros(len(ea), dtype=np.int32)
    hhi, hlo, k = np.zeros(nvars, dtype=np.int32), np.zeros(nvars, dtype=np.int32), C.c_int()
    _check(lib().isingmc_host_rj_energy_levels(_p(ea), _p(eb), _p(ej), len(ea), nvars, _p(b), _p(jhi), _p(jlo), _p(hhi), _p(hlo), C.byref(k)))
    return k.value, jhi, jlo, hhi, hlo


def rj_beta(beta, k):
    sh, mant = C.c_uint32(), C.c_uint32()
    _check(lib().isingmc_host_rj_beta(float(beta), int(k), C.byref(sh), C.byref(mant)))
    return sh.value, mant.value


def rj_log_table():
    out = np.zeros(2049, dtype=np.uint32)
    _check(lib().isingmc_host_rj_log_table(_p(out)))
    return out


def colour_graph(ea, eb, nvars):
    ea, eb = _arr(ea, np.uint64), _arr(eb, np.uint64)
    colours = np.zeros(nvars, dtype=np.uint32)
    nc = C.c_uint32()
    _check(lib().isingmc_host_colour_graph(_p(ea), _p(eb), len(ea), nvars, _p(colours), C.byref(nc)))
    return nc.value, colours


def pt_swap_round(seed, rnd, betas, slot_energy, perm):
    """One exchange round of the beta ladder; perm (uint32, rung -> slot) is updated in place."""
    betas, slot_energy = _arr(betas, np.float64), _arr(slot_energy, np.float64)
    assert perm.dtype == np.uint32 and perm.flags.c_contiguous
    swaps = C.c_uint64()
    _check(lib().isingmc_host_pt_swap_round(C.c_uint64(int(seed)), C.c_uint64(int(rnd)), len(betas), _p(betas),
                                            _p(slot_energy), _p(perm), C.byref(swaps)))
    return swaps.value


def release_cached_resources():
    """Hand the library's idle device blocks, pinned host blocks, streams and events back to the runtime; bytes released."""
    return int(lib().isingmc_release_cached_resources())


class Graph:
    """isingmc_graph: edges (+ biases) resident on one device."""

    def __init__(self, ea, eb, ej, nvars=None, biases=None, device=0, force_general=False, stable_path=False):
        self._h = _vp()
        ea, eb, ej = _arr(ea, np.uint64), _arr(eb, np.uint64), _arr(ej, np.float64)
        if nvars is None:
            nvars = int(max(ea.max(), eb.max())) + 1 if len(ea) else 0
        biases = _arr(biases, np.float64)
        _check(lib().isingmc_graph_create(_p(ea), _p(eb), _p(ej), len(ea), nvars, _p(biases), device,
                                          (FLAG_FORCE_GENERAL if force_general else 0) | (FLAG_STABLE_PATH if stable_path else 0),
                                          C.byref(self._h)))
        info = GraphInfo()
        _check(lib().isingmc_graph_info(self._h, C.byref(info)))
        self.info = info
        self.nvars = int(info.nvars)
        self.kind = int(info.kind)
        self.state_words = int(info.state_words)

    def close(self):
        if self._h:
            lib().isingmc_graph_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class States:
    """isingmc_states: R replicas of the graph's spins on the device."""

    def __init__(self, graph, seeds, initial_state=None, replica_range=None):
        """seeds: one u64 per experiment.  replica_range=(lo, hi): this object is the shard [lo, hi) of the
        len(seeds) experiments (isingmc_states_create_range: results do not depend on the cut)."""
        self.graph = graph  # keeps the graph alive
        self._h = _vp()
        seeds = _arr(seeds, np.uint64)
        ini = _arr(initial_state, np.uint8)
        if ini is not None and ini.size != graph.nvars:
            raise ValueError("Initial state must be of the same size as biases, or 0.")
        if replica_range is None:
            _check(lib().isingmc_states_create(graph._h, len(seeds), _p(seeds), _p(ini), C.byref(self._h)))
        else:
            lo, hi = int(replica_range[0]), int(replica_range[1])
            if not 0 <= lo <= hi <= len(seeds):
                raise ValueError("replica_range out of bounds")
            _check(lib().isingmc_states_create_range(graph._h, len(seeds), _p(seeds), lo, hi - lo, _p(ini),
                                                     C.byref(self._h)))

    @property
    def count(self):
        return int(lib().isingmc_states_count(self._h))

    @property
    def timestep(self):
        return int(lib().isingmc_states_timestep(self._h))

    @timestep.setter
    def timestep(self, t):
        _check(lib().isingmc_states_set_timestep(self._h, C.c_uint64(int(t))))

    def append(self, seed, initial_state=None):
        ini = _arr(initial_state, np.uint8)
        _check(lib().isingmc_states_append(self._h, C.c_uint64(int(seed)), _p(ini)))

    def set_state(self, replica, state):
        st = _arr(state, np.uint8)
        if st.size != self.graph.nvars:
            raise ValueError("Initial state must be of the same size as biases, or 0.")
        _check(lib().isingmc_states_set_state(self._h, replica, _p(st)))

    def set_betas(self, betas):
        b = _arr(betas, np.float64)
        if b is not None and b.size != self.count:
            raise ValueError("one beta per replica expected")
        _check(lib().isingmc_states_set_betas(self._h, _p(b)))

    def do_time_steps(self, timesteps, beta=None, per_step_energies=False):
        """beta: float (constant), sequence of length timesteps, or None when per-replica betas are set."""
        R = self.count
        if beta is None:
            b, stride = None, 0
        elif np.ndim(beta) == 0:
            b, stride = np.array([beta], dtype=np.float64), 0
        else:
            b, stride = _arr(beta, np.float64), 1
            if b.size != timesteps:
                raise ValueError("need one beta per timestep")
        out = np.zeros((R, timesteps), dtype=np.float64) if per_step_energies else None
        _check(lib().isingmc_do_time_steps(self._h, timesteps, _p(b), stride, _p(out)))
        return out

    def do_time_steps_timed(self, timesteps, beta):
        b = np.array([beta], dtype=np.float64) if np.ndim(beta) == 0 else _arr(beta, np.float64)
        ms = C.c_float()
        _check(lib().isingmc_do_time_steps_timed(self._h, timesteps, _p(b), 0 if b.size == 1 else 1, C.byref(ms)))
        return ms.value

    def run_sampling(self, beta, thermalization, sampling_freq, n_samples):
        """(energies float64[R, S], states bool[R, S, nvars]) -- the sampling loop of lattice.rs:271-287."""
        R, N = self.count, self.graph.nvars
        energies = np.zeros((R, n_samples), dtype=np.float64)
        states = np.zeros((R, n_samples, N), dtype=np.bool_)
        _check(lib().isingmc_run_sampling(self._h, float(0.0 if beta is None else beta), thermalization, sampling_freq,
                                          n_samples, _p(energies), states.ctypes.data_as(_vp)))
        return energies, states

    def energies(self):
        out = np.zeros(self.count, dtype=np.float64)
        _check(lib().isingmc_get_energies(self._h, _p(out)))
        return out

    def magnetisations(self):
        out = np.zeros(self.count, dtype=np.int64)
        _check(lib().isingmc_get_magnetisations(self._h, _p(out)))
        return out

    def states(self, out=None):
        """bool[R, nvars]; `out` may be a C-contiguous (R, ..., nvars)-strided uint8/bool view base."""
        R, N = self.count, self.graph.nvars
        if out is None:
            out = np.zeros((R, N), dtype=np.bool_)
            stride = N
        else:
            stride = out.strides[0]
        _check(lib().isingmc_get_states(self._h, out.ctypes.data_as(_vp), stride))
        return out

    def packed(self):
        out = np.zeros((self.count, self.graph.state_words), dtype=np.uint32)
        _check(lib().isingmc_get_packed_states(self._h, _p(out)))
        return out

    # ---- on-stream parallel tempering (lattice path): every call below only ENQUEUES on the engine's stream
    def pt_attach(self, ladder_betas, slot_offset, slots_per_rank, world_size, seed):
        b = _arr(ladder_betas, np.float64)
        _check(lib().isingmc_pt_attach(self._h, _p(b), len(b), slot_offset, slots_per_rank, world_size,
                                       C.c_uint64(int(seed))))
        self._pt_rungs = len(b)
        self._pt_world = world_size

    def pt_can_attach(self, n_rungs, slot_offset, slots_per_rank, world_size):
        """Would pt_attach accept this container and ladder geometry?  No side effects."""
        ok = C.c_int()
        _check(lib().isingmc_pt_can_attach(self._h, n_rungs, slot_offset, slots_per_rank, world_size, C.byref(ok)))
        return bool(ok.value)

    def pt_detach(self):
        _check(lib().isingmc_pt_detach(self._h))

    def pt_buffers(self):
        """(local, all) as torch CUDA tensors viewing the engine's device buffers (no copy)."""
        import torch
        loc, al, per = _vp(), _vp(), C.c_size_t()
        _check(lib().isingmc_pt_buffers(self._h, C.byref(loc), C.byref(al), C.byref(per)))

        class _View:  # __cuda_array_interface__ view of a raw device pointer
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        dev = torch.device("cuda", self.graph.info.device)
        world = self._pt_world
        return (torch.as_tensor(_View(loc.value, per.value), device=dev),
                torch.as_tensor(_View(al.value, per.value * world), device=dev))

    def pt_stream(self):
        import torch
        st = _vp()
        _check(lib().isingmc_states_stream(self._h, C.byref(st)))
        return torch.cuda.ExternalStream(st.value, device=torch.device("cuda", self.graph.info.device))

    def pt_time_steps(self, timesteps):
        _check(lib().isingmc_pt_time_steps(self._h, timesteps))

    def pt_run(self, timesteps, swap_every):
        _check(lib().isingmc_pt_run(self._h, timesteps, swap_every))

    def pt_measure(self):
        _check(lib().isingmc_pt_measure(self._h))

    def pt_swap(self):
        _check(lib().isingmc_pt_swap(self._h))

    def pt_state(self):
        perm = np.zeros(self._pt_rungs, dtype=np.uint32)
        rnd, swaps = C.c_uint64(), C.c_uint64()
        _check(lib().isingmc_pt_state(self._h, _p(perm), C.byref(rnd), C.byref(swaps)))
        return perm, rnd.value, swaps.value

    def synchronize(self):
        _check(lib().isingmc_synchronize(self._h))

    def shader_clock_ghz(self, timesteps, beta, probe_ms=10.0):
        """Shader clock held while `timesteps` sweeps run (measurement hook of bench.py)."""
        ghz = C.c_double()
        _check(lib().isingmc_debug_shader_clock(self._h, timesteps, float(beta), float(probe_ms), C.byref(ghz)))
        return ghz.value

    def close(self):
        if self._h:
            lib().isingmc_states_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
