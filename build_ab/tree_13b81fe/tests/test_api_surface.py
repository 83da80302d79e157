"""The Python surface of `py_monte_carlo` must equal the reference's pyo3 classes on the hot path
(SURVEY.md 2.1): names, positional order, keyword names, None-defaults, error types and messages.
Validation paths only -- nothing here needs a GPU."""
import numpy as np
import pytest

EDGES = [((0, 1), 1.0), ((1, 2), -1.0)]  # README.md:50-53


@pytest.fixture(scope="module")
def mod():
    import py_monte_carlo
    return py_monte_carlo


def test_module_classes(mod):
    assert {"Lattice", "ClassicIsing"} <= set(dir(mod))


def _sig(fn):
    return fn.__doc__.splitlines()[0]


def test_quantum_entry_points_fail_with_a_reason(mod):
    """lattice.rs:478-1036 and lib.rs:16-21: out of scope, but present by name."""
    import py_monte_carlo
    lat = mod.Lattice([((0, 1), 1.0)])
    for name in ("run_quantum_monte_carlo", "run_quantum_monte_carlo_sampling", "run_quantum_monte_carlo_and_measure_spins",
                 "get_offset", "average_on_and_off_diagonal_and_consts"):
        with pytest.raises(NotImplementedError, match="quantum"):
            getattr(lat, name)(1.0, 10, 2)
    for name in ("QmcIsing", "QmcRunner", "LatticeTempering"):
        with pytest.raises(NotImplementedError, match="quantum"):
            getattr(py_monte_carlo, name)
    with pytest.raises(AttributeError):
        py_monte_carlo.no_such_thing


def test_lattice_signatures(mod):
    L = mod.Lattice
    # lattice.rs:46-50, 171-179, 231-241, 309-317, 395-403: order and names of the reference's parameters
    assert "edges" in _sig(L.__init__) and "seed_gen" in _sig(L.__init__) and "use_allocator" in _sig(L.__init__)
    order = lambda s, names: [s.index(n + ":") for n in names]
    s = _sig(L.run_monte_carlo)
    idx = order(s, ["beta", "timesteps", "num_experiments", "only_basic_moves", "edge_move_importance_sampling"])
    assert idx == sorted(idx)
    s = _sig(L.run_monte_carlo_sampling)
    idx = order(s, ["beta", "timesteps", "num_experiments", "only_basic_moves", "thermalization_time",
                    "sampling_freq", "edge_move_importance_sampling"])
    assert idx == sorted(idx)
    for name in ("run_monte_carlo_annealing", "run_monte_carlo_annealing_and_get_energies"):
        s = _sig(getattr(L, name))
        idx = order(s, ["betas", "timesteps", "num_experiments", "only_basic_moves", "edge_move_importance_sampling"])
        assert idx == sorted(idx)
    for name in ("set_seed_gen", "make_seeds", "set_enable_rvb_update", "set_enable_heatbath_update",
                 "set_individual_bias", "set_global_bias", "set_transverse_field", "set_initial_state", "clone"):
        assert hasattr(L, name)


def test_classic_ising_signatures(mod):
    C = mod.ClassicIsing
    s = _sig(C.__init__)
    idx = [s.index(n + ":") for n in ["edges", "longitudinal", "num_experiments", "seed", "use_basic_moves"]]
    assert idx == sorted(idx)  # classicising.rs:27-33
    s = _sig(C.run_monte_carlo)
    idx = [s.index(n + ":") for n in ["beta", "timesteps", "nspinupdates", "nedgeupdates", "nwormupdates",
                                      "only_basic_moves"]]
    assert idx == sorted(idx)  # classicising.rs:88-96
    s = _sig(C.run_monte_carlo_sampling)
    idx = [s.index(n + ":") for n in ["beta", "timesteps", "nspinupdates", "nedgeupdates", "nwormupdates",
                                      "only_basic_moves", "thermalization_time", "sampling_freq"]]
    assert idx == sorted(idx)  # classicising.rs:119-130
    assert "initial_state" in _sig(C.add_graph) and "edge_move_importance_sampling" in _sig(C.add_graph)


def test_lattice_validation_errors(mod):
    with pytest.raises(ValueError, match="Must supply some edges for graph"):   # lattice.rs:70-72
        mod.Lattice([])
    lat = mod.Lattice(EDGES)
    with pytest.raises(ValueError, match="Index out of bounds: variable 3 out of 3"):  # lattice.rs:119-124
        lat.set_individual_bias(3, 1.0)
    lat.set_individual_bias(2, 0.5)
    lat.set_global_bias(0.0)
    with pytest.raises(ValueError, match="Transverse field must be positive"):  # lattice.rs:141-144
        lat.set_transverse_field(-1.0)
    with pytest.raises(ValueError, match="Initial state must be of the same size as biases, or 0."):  # :157-159
        lat.set_initial_state([True])
    lat.set_initial_state([True, False, True])
    lat.set_initial_state([])
    lat.set_transverse_field(0.5)
    for call in (lambda: lat.run_monte_carlo(1.0, 10, 2),
                 lambda: lat.run_monte_carlo_sampling(1.0, 10, 2),
                 lambda: lat.run_monte_carlo_annealing([(0, 1.0)], 10, 2),
                 lambda: lat.run_monte_carlo_annealing_and_get_energies([(0, 1.0)], 10, 2)):
        with pytest.raises(ValueError, match="Cannot run classic monte carlo with transverse field"):  # :217-219
            call()
    lat.set_transverse_field(0.0)  # == 0 clears it (lattice.rs:138-140)


def test_make_seeds_public_and_reproducible(mod, oracle):
    lat = mod.Lattice(EDGES, seed_gen=1234)
    seeds = lat.make_seeds(5)
    assert seeds == [int(x) for x in oracle.make_seeds(1234, 5)] == lat.make_seeds(5)  # lattice.rs:76-91
    lat.set_seed_gen(None)
    assert lat.make_seeds(3) != lat.make_seeds(3)
    clone = mod.Lattice(EDGES, 7).clone()
    assert clone.make_seeds(2) == mod.Lattice(EDGES, seed_gen=7).make_seeds(2)


def test_keywords_and_none_defaults(mod):
    lat = mod.Lattice(edges=EDGES, seed_gen=None, use_allocator=None)
    lat.set_transverse_field(1.0)
    with pytest.raises(ValueError, match="transverse"):
        lat.run_monte_carlo(beta=0.3, timesteps=5, num_experiments=2, only_basic_moves=None,
                            edge_move_importance_sampling=None)
    with pytest.raises(ValueError, match="transverse"):
        lat.run_monte_carlo_sampling(0.3, 5, 2, None, None, None, None)


def test_from_arrays_extension(mod):
    lat = mod.Lattice.from_arrays(np.array([0, 1]), np.array([1, 2]), np.array([1.0, -1.0]), seed_gen=3)
    assert lat.make_seeds(2) == mod.Lattice(EDGES, 3).make_seeds(2)
    with pytest.raises(ValueError):
        mod.Lattice.from_arrays(np.array([0]), np.array([1, 2]), np.array([1.0]))


def test_device_list_of_the_in_process_fan_out(mod, monkeypatch):
    """ISINGMC_DEVICES / set_devices (extension): the device list Lattice.run_monte_carlo* fans its experiments out over.
    Host-side parsing only -- no device is touched before the first run."""
    edges = [((0, 1), 1.0), ((1, 2), -1.0)]
    monkeypatch.delenv("ISINGMC_DEVICES", raising=False)
    monkeypatch.delenv("ISINGMC_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert mod.Lattice(edges).get_devices() == [0]
    monkeypatch.setenv("ISINGMC_DEVICE", "3")
    assert mod.Lattice(edges).get_devices() == [3]
    monkeypatch.setenv("ISINGMC_DEVICES", "0, 2,2,5")
    lat = mod.Lattice(edges)
    assert lat.get_devices() == [0, 2, 2, 5]                       # the list wins over ISINGMC_DEVICE; an ordinal may repeat
    lat.set_device(1)
    assert lat.get_devices() == [1]
    lat.set_devices([4, 4])
    assert lat.clone().get_devices() == [4, 4]
    with pytest.raises(ValueError):
        lat.set_devices([])
