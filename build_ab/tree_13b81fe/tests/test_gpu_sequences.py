"""Random CALL SEQUENCES on one states object against an oracle-backed model (hypothesis, derandomised): timesteps with a scalar
beta / a schedule / per-replica betas, with and without the energy after every timestep, set_state, sampling runs, jumps of the
timestep counter (also across 2^32), interleaved with reads of energies, magnetisations and configurations -- on every path:
LDS-resident, strip and streaming checkerboard kernels, a multi-class (field) lattice, the f64 CSR path, and the two
replica-packed paths.  The single-call parity tests cannot see state that leaks from one call into the next (stale measurement
caches, stream lanes left forked, per-replica tables not rebuilt, counters)."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu
COMMON = dict(deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
SCALE = int(os.environ.get("ISINGMC_HYP_SCALE", "1"))
BETAS = [0.0, 0.25, 0.4407, 0.8, 3.0]


class LatticeModel:
    def __init__(self, oracle, lat, nvars):
        self.o, self.lat, self.nvars = oracle, lat, nvars

    def start(self, seeds):
        self.seeds = [int(s) for s in seeds]
        self.st = [self.lat.init(s) for s in self.seeds]

    def step(self, t, betas):
        for r, s in enumerate(self.seeds):
            self.lat.sweep(self.st[r], s, t, betas[r])

    def energy(self, r):
        return self.lat.energy_mag(self.st[r])[0]

    def mag(self, r):
        return self.lat.energy_mag(self.st[r])[1]

    def spins(self, r):
        return self.lat.unpack(self.st[r])

    def set_state(self, r, spins):
        self.st[r] = self.lat.pack(spins)


class CsrModel:
    def __init__(self, oracle, ea, eb, ej, nvars, biases):
        self.o, self.g, self.nvars, self.biases = oracle, (ea, eb, ej), nvars, biases

    def start(self, seeds):
        self.seeds = [int(s) for s in seeds]
        self.st = [self.o.gen_run(*self.g, self.nvars, s, [], biases=self.biases)[1] for s in self.seeds]
        self.e = [None] * len(self.seeds)

    def step(self, t, betas):
        for r, s in enumerate(self.seeds):
            self.e[r], self.st[r] = self.o.gen_run(*self.g, self.nvars, s, [betas[r]], biases=self.biases, initial=self.st[r], t0=t)

    def energy(self, r):
        return self.o.energy(*self.g, self.nvars, self.st[r], self.biases)

    def mag(self, r):
        return 2 * int(self.st[r].sum()) - self.nvars

    def spins(self, r):
        return self.st[r]

    def set_state(self, r, spins):
        self.st[r] = np.ascontiguousarray(spins, dtype=np.uint8)


class PackedModel:
    """Engines D / E: whole groups of 32 replicas, the unused ones of the last group included."""

    def __init__(self, oracle, run, ea, eb, ej, nvars, biases):
        self.o, self.run, self.g, self.nvars, self.biases = oracle, run, (ea, eb, ej), nvars, biases

    def _call(self, T, **kw):
        if self.biases is not None:
            kw["biases"] = self.biases
        return self.run(*self.g, self.nvars, self.seeds, T, **kw)

    def start(self, seeds):
        self.seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        self.e, self.st = self._call(0, betas=[])

    def step(self, t, betas):
        self.e, self.st = self._call(1, beta_replica=np.array(betas, dtype=np.float64), states=self.st, t0=t)

    def energy(self, r):
        # the engine's own energy (on the real-coupling path: of the couplings rounded to 2^k, an exact integer sum)
        return self._call(0, betas=[], states=self.st)[0][r]

    def mag(self, r):
        return 2 * int(self.st[r].sum()) - self.nvars

    def spins(self, r):
        return self.st[r]

    def set_state(self, r, spins):
        self.st[r] = np.ascontiguousarray(spins, dtype=np.uint8)


def _build(path, capi, oracle, exact, monkeypatch):
    rng = np.random.default_rng(17)
    if path in ("resident", "strips", "streaming"):
        W, H = (64, 8) if path == "resident" else (1024, 512)
        if path == "streaming":
            monkeypatch.setenv("ISINGMC_STRIP", "0")
        ea, eb, ej = exact.square_lattice_edges(W, H, 1.0, rng)
        g = capi.Graph(ea, eb, ej)
        assert g.kind == capi.KIND_LATTICE2D
        lat = oracle.Lat(W, H, 1.0, 0, (ej[0::2] > 0).astype(np.uint8), (ej[1::2] > 0).astype(np.uint8))
        return g, LatticeModel(oracle, lat, W * H), (2 if path != "resident" else 5), True
    if path == "field":
        W, H, h = 256, 16, -0.6
        ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
        g = capi.Graph(ea, eb, ej, biases=np.full(W * H, h))
        assert g.kind == capi.KIND_LATTICE2D and g.info.fast_path == 1
        return g, LatticeModel(oracle, oracle.Lat(W, H, 1.0, 0, field=h), W * H), 4, True
    if path == "csr":
        n = 50
        ea = rng.integers(0, n, 120).astype(np.uint64)
        eb = rng.integers(0, n, 120).astype(np.uint64)
        ej = np.round(rng.normal(size=120), 3)
        biases = np.round(rng.normal(size=n) * 0.3, 3)
        g = capi.Graph(ea, eb, ej, nvars=n, biases=biases)
        return g, CsrModel(oracle, ea, eb, ej, n, biases), 4, False
    if path == "packed":
        monkeypatch.setenv("ISINGMC_FORCE_PACKED", "1")
        monkeypatch.setenv("ISINGMC_DISABLE_REAL", "1")
        ea, eb, ej = exact.cubic_lattice_edges(6, -1.0)
        ej = ej * rng.choice([-1.0, 1.0], len(ej))
        g = capi.Graph(ea, eb, ej, force_general=True)
        assert g.info.packed_degree == 6
        return g, PackedModel(oracle, oracle.pk_run, ea, eb, ej, 216, None), 37, True
    monkeypatch.setenv("ISINGMC_FORCE_REAL", "1")
    ea, eb, _ = exact.square_lattice_edges(12, 10, 1.0)
    ej = rng.normal(size=len(ea))
    biases = rng.normal(size=120) * 0.3
    g = capi.Graph(ea, eb, ej, nvars=120, biases=biases, force_general=True)
    assert g.info.real_slots == 4
    return g, PackedModel(oracle, oracle.rj_run, ea, eb, ej, 120, biases), 37, True


OPS = st.one_of(
    st.tuples(st.just("steps"), st.integers(0, 3), st.sampled_from(BETAS), st.booleans()),
    st.tuples(st.just("schedule"), st.integers(1, 3), st.integers(0, 10 ** 6), st.booleans()),
    st.tuples(st.just("set_betas"), st.integers(0, 10 ** 6)),
    st.tuples(st.just("clear_betas")),
    st.tuples(st.just("set_state"), st.integers(0, 10 ** 6)),
    st.tuples(st.just("sampling"), st.integers(0, 2), st.integers(1, 2), st.integers(1, 2), st.sampled_from(BETAS)),
    st.tuples(st.just("jump"), st.sampled_from([0, 5, 2 ** 32 - 2, 2 ** 32 - 1, 2 ** 40 + 3])),
    st.tuples(st.just("check")),
)


def _check(states, model, R, exact_energy):
    e, m, s = states.energies(), states.magnetisations(), states.states()
    for r in range(R):
        assert np.array_equal(s[r].astype(np.uint8), model.spins(r)), r
        assert m[r] == model.mag(r)
        if exact_energy:
            assert e[r] == model.energy(r)
        else:
            np.testing.assert_allclose(e[r], model.energy(r), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("path", ["resident", "strips", "streaming", "field", "csr", "packed", "real"])
def test_call_sequences(capi, oracle, exact, monkeypatch, path):
    g, model, R, exact_energy = _build(path, capi, oracle, exact, monkeypatch)
    examples = (30 if path in ("strips", "streaming") else 60) * SCALE

    @settings(max_examples=examples, **COMMON)
    @given(ops=st.lists(OPS, min_size=1, max_size=7), seed=st.integers(0, 2 ** 63))
    def run(ops, seed):
        seeds = capi.make_seeds(seed, R)
        states = capi.States(g, seeds)
        model.start(seeds)
        t, per_replica = 0, None
        rng = np.random.default_rng(seed % 2 ** 32)
        for op in ops:
            kind = op[0]
            if kind in ("steps", "schedule"):
                T = op[1]
                betas = [op[2]] * T if kind == "steps" else list(np.random.default_rng(op[2]).choice(BETAS, T))
                arg = (op[2] if kind == "steps" else np.array(betas)) if per_replica is None else None
                out = states.do_time_steps(T, arg, per_step_energies=op[3])
                for k in range(T):
                    model.step(t, per_replica if per_replica is not None else [betas[k]] * R)
                    t += 1
                    if op[3]:
                        for r in range(R):
                            if exact_energy:
                                assert out[r, k] == model.energy(r), (kind, k, r)
                            else:
                                np.testing.assert_allclose(out[r, k], model.energy(r), rtol=1e-9, atol=1e-9)
            elif kind == "set_betas":
                per_replica = list(np.random.default_rng(op[1]).choice(BETAS, R))
                states.set_betas(per_replica)
            elif kind == "clear_betas":
                per_replica = None
                states.set_betas(None)
            elif kind == "set_state":
                r = op[1] % R
                spins = rng.integers(0, 2, model.nvars).astype(np.uint8)
                states.set_state(r, spins)
                model.set_state(r, spins)
            elif kind == "sampling":
                therm, freq, n, beta = op[1:]
                e, s = states.run_sampling(beta, therm, freq, n)
                for _ in range(therm):
                    model.step(t, per_replica if per_replica is not None else [beta] * R)
                    t += 1
                for k in range(n):
                    for _ in range(freq):
                        model.step(t, per_replica if per_replica is not None else [beta] * R)
                        t += 1
                    for r in range(R):
                        assert np.array_equal(s[r, k].astype(np.uint8), model.spins(r)), (k, r)
                        if exact_energy:
                            assert e[r, k] == model.energy(r)
                        else:
                            np.testing.assert_allclose(e[r, k], model.energy(r), rtol=1e-9, atol=1e-9)
            elif kind == "jump":
                t = op[1]
                states.timestep = t
            else:
                _check(states, model, R, exact_energy)
            assert states.timestep == t
        _check(states, model, R, exact_energy)

    run()


@pytest.fixture(scope="module")
def mod():
    import py_monte_carlo
    return py_monte_carlo


CI_OPS = st.one_of(
    st.tuples(st.just("run"), st.sampled_from(BETAS), st.integers(0, 3), st.sampled_from([None, 1.0, 0.5, 2.0, 1.37, 0.01])),
    st.tuples(st.just("add"), st.booleans(), st.integers(0, 10 ** 6)),
    st.tuples(st.just("sample"), st.sampled_from(BETAS), st.integers(1, 4), st.integers(0, 2), st.integers(1, 2)),
    st.tuples(st.just("check")),
)


@pytest.mark.parametrize("kind", ["lattice", "general graph"])
def test_classic_ising_call_sequences(mod, capi, oracle, exact, kind):
    """The persistent container of classicising.rs:27-179 through the Python surface: run_monte_carlo with any nspinupdates
    (attempts accumulate on a cursor across calls, every nvars of them run as one sweep), add_graph with and without an initial
    state between runs (a new replica starts at the container's current timestep, its seed is the container rng's next draw),
    sampling runs, reads -- against the model."""
    rng0 = np.random.default_rng(5)
    if kind == "lattice":
        W, H = 64, 8
        ea, eb, ej = exact.square_lattice_edges(W, H, -1.0)
        n = W * H
        model = LatticeModel(oracle, oracle.Lat(W, H), n)
        exact_energy = True
    else:
        n = 40
        ea = rng0.integers(0, n, 90).astype(np.uint64)
        eb = (ea + rng0.integers(1, n, 90).astype(np.uint64)) % n
        ej = np.round(rng0.normal(size=90), 3)
        ea[-1], eb[-1] = n - 1, 0                                           # nvars = max index + 1
        model = CsrModel(oracle, ea, eb, ej, n, None)
        exact_energy = False
    edges = [((int(a), int(b)), float(j)) for a, b, j in zip(ea, eb, ej)]

    @settings(max_examples=40 * SCALE, **COMMON)
    @given(ops=st.lists(CI_OPS, min_size=1, max_size=7), seed=st.integers(0, 2 ** 63), first=st.integers(1, 3))
    def run(ops, seed, first):
        ci = mod.ClassicIsing(edges, None, first, seed)
        R = first
        model.start(capi.make_seeds(seed, R))
        t, pending = 0, 0
        rng = np.random.default_rng(seed % 2 ** 32)

        def sweeps(k, beta):
            nonlocal t
            for _ in range(k):
                model.step(t, [beta] * R)
                t += 1

        for op in ops:
            if op[0] == "run":
                _, beta, T, per = op
                nspin = None if per is None else max(1, int(per * n))
                ci.run_monte_carlo(beta, T, nspin)
                pending += T * (n if nspin is None else nspin)
                sweeps(pending // n, beta)
                pending %= n
            elif op[0] == "add":
                initial = rng.integers(0, 2, n).astype(bool) if op[1] else None
                ci.add_graph(None if initial is None else [bool(x) for x in initial])
                R += 1
                new_seed = int(capi.make_seeds(seed, R)[-1])
                model.seeds.append(new_seed)
                if kind == "lattice":
                    model.st.append(model.lat.init(new_seed) if initial is None else model.lat.pack(initial.astype(np.uint8)))
                else:
                    model.st.append(oracle.gen_run(ea, eb, ej, n, new_seed, [])[1] if initial is None else initial.astype(np.uint8))
                    model.e.append(None)
            elif op[0] == "sample":
                _, beta, T, therm, freq = op
                e, s = ci.run_monte_carlo_sampling(beta, T, None, None, None, None, therm, freq)
                S = T // freq
                assert e.shape == (R, S) and s.shape == (R, S, n)
                sweeps(therm, beta)
                for k in range(S):
                    sweeps(freq, beta)
                    for r in range(R):
                        assert np.array_equal(s[r, k].astype(np.uint8), model.spins(r))
            else:
                pass
            states, energies = ci.get_states(), ci.get_energies()
            assert ci.get_num_graphs() == R
            for r in range(R):
                assert np.array_equal(states[r].astype(np.uint8), model.spins(r)), (op, r)
                if exact_energy:
                    assert energies[r] == model.energy(r)
                else:
                    np.testing.assert_allclose(energies[r], model.energy(r), rtol=1e-9, atol=1e-9)

    run()
