"""ctypes front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of ising_oracle.c).  PARITY UNPINNED against the reference's `qmc` crate; pinned
by published RNG vectors, README.md:45-46 energies, exact enumeration and Kaufman's solution.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("ISING_ORACLE_LIB")  # tests/test_host_sanitizers.py: a build with -fsanitize
        if not path:
            path = os.path.join(_HERE, "liboracle.so")
            if not os.path.exists(path):
                build()
        L = C.CDLL(path)
        L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
        L.orc_xoshiro_from_state.argtypes = [u64p, C.c_size_t, u64p]
        L.orc_make_seeds.argtypes = [C.c_uint64, C.c_size_t, u64p]
        L.orc_energy.restype = C.c_double
        L.orc_energy.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, C.c_void_p, u8p]
        L.orc_ref_run.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, C.c_void_p, u64p,
                                  C.c_size_t, C.c_void_p, f64p, C.c_size_t, u8p, C.c_void_p,
                                  C.c_void_p]
        L.orc_ref_averages.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, C.c_void_p, u64p, C.c_size_t, C.c_void_p,
                                       C.c_double, C.c_size_t, C.c_size_t, f64p, f64p]
        L.orc_ref_bench.restype = C.c_uint64
        L.orc_ref_bench.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, u64p, C.c_size_t,
                                    C.c_double, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        L.orc_threshold_fixed.restype = C.c_uint64
        L.orc_threshold_fixed.argtypes = [C.c_double, C.c_double]
        L.orc_lat_supported.argtypes = [C.c_int, C.c_int]
        L.orc_lat_state_words.restype = C.c_size_t
        L.orc_lat_state_words.argtypes = [C.c_int, C.c_int]
        L.orc_lat_init.argtypes = [C.c_int, C.c_int, C.c_uint64, u32p]
        L.orc_lat_pack.argtypes = [C.c_int, C.c_int, u8p, u32p]
        L.orc_lat_unpack.argtypes = [C.c_int, C.c_int, u32p, u8p]
        L.orc_lat_sweep.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                    u32p, C.c_uint64, C.c_uint64, C.c_double]
        L.orc_lat_energy_mag.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                         C.c_void_p, u32p, C.POINTER(C.c_double),
                                         C.POINTER(C.c_int64)]
        L.orc_lat_sweep_ex.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                       C.c_int, u32p, C.c_uint64, C.c_uint64, C.c_double]
        L.orc_lat_energy_mag_ex.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                            C.c_int, C.c_int, u32p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.orc_lat_sweep_ex2.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                        C.c_void_p, C.c_int, C.c_int, u32p, C.c_uint64, C.c_uint64, C.c_double]
        L.orc_lat_energy_mag_ex2.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_double, C.c_void_p, C.c_int, C.c_int, u32p, C.POINTER(C.c_double),
                                             C.POINTER(C.c_int64)]
        L.orc_det_exp.restype = C.c_double
        L.orc_det_exp.argtypes = [C.c_double]
        L.orc_gen_colouring.restype = C.c_uint32
        L.orc_gen_colouring.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, C.c_void_p,
                                        C.c_void_p]
        L.orc_gen_run.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, C.c_void_p,
                                  C.c_uint64, C.c_void_p, C.c_uint64, f64p, C.c_size_t, u8p,
                                  C.c_void_p, C.c_void_p]
        L.orc_pk_run.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, u64p, C.c_size_t, C.c_int, C.c_uint64,
                                 C.c_void_p, C.c_void_p, C.c_size_t, u8p, C.c_void_p, C.c_void_p]
        L.orc_rj_log_table.argtypes = [u32p]
        L.orc_rj_lambda.restype = C.c_uint32
        L.orc_rj_lambda.argtypes = [C.c_uint32]
        L.orc_rj_quantise.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_int)]
        L.orc_rj_eligible.restype = C.c_int
        L.orc_rj_eligible.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, C.c_void_p]
        L.orc_rj_beta.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_rj_accept.restype = C.c_int
        L.orc_rj_accept.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rj_run.argtypes = [C.c_size_t, u64p, u64p, f64p, C.c_size_t, C.c_void_p, u64p, C.c_size_t, C.c_int,
                                 C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, u8p, C.c_void_p, C.c_void_p]
        L.orc_pt_swap_round.restype = C.c_uint64
        L.orc_pt_swap_round.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, f64p, f64p, u32p]
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def split_edges(edges):
    """[((a, b), j), ...] (the reference's edge-list form, lattice.rs:47) -> three arrays."""
    ea = np.ascontiguousarray([e[0][0] for e in edges], dtype=np.uint64)
    eb = np.ascontiguousarray([e[0][1] for e in edges], dtype=np.uint64)
    ej = np.ascontiguousarray([e[1] for e in edges], dtype=np.float64)
    return ea, eb, ej


def philox(ctr, key):
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(np.asarray(ctr, dtype=np.uint32), np.asarray(key, dtype=np.uint32), out)
    return out


def make_seeds(seed_gen, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().orc_make_seeds(C.c_uint64(seed_gen), n, out)
    return out


def xoshiro_from_state(s, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().orc_xoshiro_from_state(np.asarray(s, dtype=np.uint64), n, out)
    return out


def energy(ea, eb, ej, nvars, state, biases=None):
    b = None if biases is None else np.ascontiguousarray(biases, dtype=np.float64)
    st = np.ascontiguousarray(state, dtype=np.uint8)
    return lib().orc_energy(len(ea), ea, eb, ej, nvars, _ptr(b), st)


def ref_run(ea, eb, ej, nvars, seeds, betas, biases=None, initial=None, per_step=False):
    """Reference-faithful engine (random-site sequential Metropolis), R = len(seeds) chains."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    R, T = len(seeds), len(betas)
    states = np.zeros((R, nvars), dtype=np.uint8)
    energies = np.zeros(R, dtype=np.float64)
    eps = np.zeros((R, T), dtype=np.float64) if per_step else None
    b = None if biases is None else np.ascontiguousarray(biases, dtype=np.float64)
    ini = None if initial is None else np.ascontiguousarray(initial, dtype=np.uint8)
    lib().orc_ref_run(len(ea), ea, eb, ej, nvars, _ptr(b), seeds, R, _ptr(ini), betas, T, states,
                      _ptr(energies), _ptr(eps))
    return (energies, states, eps) if per_step else (energies, states)


def ref_averages(ea, eb, ej, nvars, seeds, beta, therm, steps, biases=None, initial=None):
    """(mean E, mean |M|) per chain of engine A: `therm` timesteps, then averages over `steps` timesteps."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    b = None if biases is None else np.ascontiguousarray(biases, dtype=np.float64)
    ini = None if initial is None else np.ascontiguousarray(initial, dtype=np.uint8)
    e, m = np.zeros(len(seeds)), np.zeros(len(seeds))
    lib().orc_ref_averages(len(ea), ea, eb, ej, nvars, _ptr(b), seeds, len(seeds), _ptr(ini), float(beta), therm, steps, e, m)
    return e, m


def ref_bench(ea, eb, ej, nvars, seeds, beta, timesteps, threads):
    """Timed sweep loop of the reference-faithful engine; returns (seconds, attempts)."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    sec = C.c_double()
    lib().orc_ref_bench(len(ea), ea, eb, ej, nvars, seeds, len(seeds), float(beta), timesteps,
                        int(threads), C.byref(sec))
    return sec.value, len(seeds) * nvars * timesteps


def threshold_fixed(beta, dE):
    return int(lib().orc_threshold_fixed(beta, dE))


def det_exp(x):
    return lib().orc_det_exp(x)


class Lat:
    """Checkerboard spec engine (engine B) for one replica of a periodic W x H lattice."""

    def __init__(self, W, H, jabs=1.0, jpos_uniform=0, jright=None, jdown=None, field=0.0, open_x=False, open_y=False, jabs_y=None,
                 field_neg=None):
        """field: uniform h of E = sum J s s - h sum s; open_x / open_y: no bonds between columns W-1 and 0 / rows H-1 and 0;
        jabs_y: |J| of the vertical bonds when it differs from the horizontal bonds' (jabs)."""
        self.jabs_y = -1.0 if jabs_y is None else float(jabs_y)
        # field_neg: uint8[H*W], 1 where the site's field is -field instead of +field
        self.field_neg = None if field_neg is None else np.ascontiguousarray(field_neg, dtype=np.uint8).ravel()
        assert lib().orc_lat_supported(W, H), (W, H)
        self.W, self.H, self.jabs, self.jpos = W, H, float(jabs), int(jpos_uniform)
        self.field, self.open_x, self.open_y = float(field), int(bool(open_x)), int(bool(open_y))
        self.jright = None if jright is None else np.ascontiguousarray(jright, dtype=np.uint8)
        self.jdown = None if jdown is None else np.ascontiguousarray(jdown, dtype=np.uint8)
        self.words = lib().orc_lat_state_words(W, H)

    def init(self, seed):
        st = np.zeros(self.words, dtype=np.uint32)
        lib().orc_lat_init(self.W, self.H, C.c_uint64(int(seed)), st)
        return st

    def pack(self, spins):
        st = np.zeros(self.words, dtype=np.uint32)
        lib().orc_lat_pack(self.W, self.H, np.ascontiguousarray(spins, dtype=np.uint8).ravel(), st)
        return st

    def unpack(self, st):
        out = np.zeros(self.W * self.H, dtype=np.uint8)
        lib().orc_lat_unpack(self.W, self.H, st, out)
        return out

    def sweep(self, st, seed, t, beta):
        lib().orc_lat_sweep_ex2(self.W, self.H, self.jabs, self.jabs_y, self.jpos, _ptr(self.jright), _ptr(self.jdown), self.field,
                                _ptr(self.field_neg), self.open_x, self.open_y, st, C.c_uint64(int(seed)), C.c_uint64(int(t)), float(beta))

    def energy_mag(self, st):
        e, m = C.c_double(), C.c_int64()
        lib().orc_lat_energy_mag_ex2(self.W, self.H, self.jabs, self.jabs_y, self.jpos, _ptr(self.jright), _ptr(self.jdown), self.field,
                                     _ptr(self.field_neg), self.open_x, self.open_y, st, C.byref(e), C.byref(m))
        return e.value, m.value


def gen_colouring(ea, eb, ej, nvars):
    colours = np.zeros(nvars, dtype=np.uint32)
    pos = np.zeros(nvars, dtype=np.uint64)
    nc = lib().orc_gen_colouring(len(ea), ea, eb, ej, nvars, _ptr(colours), _ptr(pos))
    return nc, colours, pos


def gen_run(ea, eb, ej, nvars, seed, betas, biases=None, initial=None, t0=0, state=None,
            per_step=False):
    """General-path spec engine (engine C), one replica. Returns (energy, state[, per-step E])."""
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    T = len(betas)
    st = np.zeros(nvars, dtype=np.uint8) if state is None else np.ascontiguousarray(state, dtype=np.uint8)
    e = C.c_double()
    eps = np.zeros(T, dtype=np.float64) if per_step else None
    b = None if biases is None else np.ascontiguousarray(biases, dtype=np.float64)
    ini = None if initial is None else np.ascontiguousarray(initial, dtype=np.uint8)
    lib().orc_gen_run(len(ea), ea, eb, ej, nvars, _ptr(b), C.c_uint64(int(seed)), _ptr(ini),
                      C.c_uint64(int(t0)), betas, T, st, C.cast(C.byref(e), C.c_void_p), _ptr(eps))
    return (e.value, st, eps) if per_step else (e.value, st)


def pt_swap_round(seed, rnd, betas, slot_energy, perm):
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    slot_energy = np.ascontiguousarray(slot_energy, dtype=np.float64)
    return int(lib().orc_pt_swap_round(C.c_uint64(int(seed)), C.c_uint64(int(rnd)), len(betas),
                                       betas, slot_energy, perm))


def pk_run(ea, eb, ej, nvars, seeds, timesteps, betas=None, beta_replica=None, states=None, t0=0, per_step=False):
    """Replica-packed spec engine (engine D).  states: None (random start) or uint8[32*ceil(R/32), nvars]
    carried over from a previous call.  Returns (energies[R], states[32G, nvars][, per-step energies])."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    R = len(seeds)
    G = (R + 31) // 32
    random_start = states is None
    st = np.zeros((32 * G, nvars), dtype=np.uint8) if states is None else np.ascontiguousarray(states, dtype=np.uint8)
    b = None if betas is None else np.ascontiguousarray(betas, dtype=np.float64)
    br = None if beta_replica is None else np.ascontiguousarray(beta_replica, dtype=np.float64)
    e = np.zeros(R, dtype=np.float64)
    eps = np.zeros((R, timesteps), dtype=np.float64) if per_step else None
    lib().orc_pk_run(len(ea), ea, eb, ej, nvars, seeds, R, int(random_start), C.c_uint64(int(t0)), _ptr(b), _ptr(br),
                     timesteps, st, _ptr(e), _ptr(eps))
    return (e, st, eps) if per_step else (e, st)


# ---- engine E: replica-packed real-coupling path ------------------------------------------
def rj_log_table():
    out = np.zeros(2049, dtype=np.uint32)
    lib().orc_rj_log_table(out)
    return out


def rj_lambda(u):
    return int(lib().orc_rj_lambda(C.c_uint32(int(u))))


def rj_quantise(ea, eb, ej, nvars, biases=None):
    """(k, jq per input edge, hq per site): couplings as integers in units of 2^k."""
    b = None if biases is None else np.ascontiguousarray(biases, dtype=np.float64)
    jq, hq, k = np.zeros(len(ea), dtype=np.int32), np.zeros(nvars, dtype=np.int32), C.c_int()
    lib().orc_rj_quantise(len(ea), ea, eb, ej, nvars, _ptr(b), _ptr(jq), _ptr(hq), C.byref(k))
    return k.value, jq, hq


def rj_eligible(ea, eb, ej, nvars, biases=None):
    b = None if biases is None else np.ascontiguousarray(biases, dtype=np.float64)
    return bool(lib().orc_rj_eligible(len(ea), ea, eb, ej, nvars, _ptr(b)))


def rj_beta(beta, k):
    sh, mant = C.c_uint32(), C.c_uint32()
    lib().orc_rj_beta(float(beta), int(k), C.byref(sh), C.byref(mant))
    return sh.value, mant.value


def rj_accept(X, u, shift, mant):
    return bool(lib().orc_rj_accept(C.c_int32(int(X)), C.c_uint32(int(u)), C.c_uint32(shift), C.c_uint32(mant)))


def rj_run(ea, eb, ej, nvars, seeds, timesteps, betas=None, beta_replica=None, biases=None, states=None, t0=0, per_step=False):
    """Real-coupling spec engine (engine E); arguments and results as pk_run, plus site biases."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    R = len(seeds)
    G = (R + 31) // 32
    random_start = states is None
    st = np.zeros((32 * G, nvars), dtype=np.uint8) if states is None else np.ascontiguousarray(states, dtype=np.uint8)
    b = None if betas is None else np.ascontiguousarray(betas, dtype=np.float64)
    br = None if beta_replica is None else np.ascontiguousarray(beta_replica, dtype=np.float64)
    h = None if biases is None else np.ascontiguousarray(biases, dtype=np.float64)
    e = np.zeros(R, dtype=np.float64)
    eps = np.zeros((R, timesteps), dtype=np.float64) if per_step else None
    lib().orc_rj_run(len(ea), ea, eb, ej, nvars, _ptr(h), seeds, R, int(random_start), C.c_uint64(int(t0)), _ptr(b), _ptr(br),
                     timesteps, st, _ptr(e), _ptr(eps))
    return (e, st, eps) if per_step else (e, st)
