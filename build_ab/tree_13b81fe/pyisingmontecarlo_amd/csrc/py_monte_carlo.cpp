// Host shim with the Python surface of the reference's pyo3 module `py_monte_carlo`
// (src/lib.rs:14-22) for the classical hot path: classes Lattice (src/lattice.rs:27-470) and
// ClassicIsing (src/classicising.rs:11-180).  Same method names, positional order, keyword names,
// None-defaults, return tuple order, dtypes (float64 / bool) and ValueError messages.  The reference
// host is Rust/pyo3; no Rust toolchain exists in this image, so the shim is C++/pybind11 over the C
// ABI of include/isingmc.h -- exactly the calls a pyo3 maintainer would bind.
// All Monte-Carlo work happens in libisingmc.so's HIP kernels; nothing here computes a spin flip.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <memory>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "isingmc.h"

namespace py = pybind11;

using Edge = std::pair<std::pair<size_t, size_t>, double>;

namespace {

void check(int rc)
{
    if (rc == ISINGMC_OK) return;
    const std::string msg = isingmc_last_error();
    if (rc == ISINGMC_ERR_INVALID) throw py::value_error(msg);
    if (rc == ISINGMC_ERR_ALLOC) throw std::bad_alloc();
    throw std::runtime_error(msg);
}

int default_device()
{
    if (const char *d = std::getenv("ISINGMC_DEVICE")) return std::atoi(d);
    if (const char *lr = std::getenv("LOCAL_RANK")) { // one process per GPU under torchrun
        int count = 0;
        if (isingmc_device_count(&count) == ISINGMC_OK && std::atoi(lr) < count) return std::atoi(lr);
    }
    return 0;
}

// ISINGMC_DEVICES=0,1,...: the devices Lattice.run_monte_carlo* fans its experiments out over (one host thread and
// one isingmc_states per entry; an ordinal may repeat).  Unset: the one device of default_device().
std::vector<int> default_devices()
{
    std::vector<int> out;
    if (const char *e = std::getenv("ISINGMC_DEVICES")) {
        const std::string str(e);
        size_t pos = 0;
        while (pos <= str.size()) {
            const size_t comma = std::min(str.find(',', pos), str.size());
            const std::string tok = str.substr(pos, comma - pos);
            if (!tok.empty()) {
                if (tok == "all") {
                    int count = 0;
                    if (isingmc_device_count(&count) == ISINGMC_OK)
                        for (int d = 0; d < count; d++) out.push_back(d);
                } else {
                    out.push_back(std::atoi(tok.c_str()));
                }
            }
            pos = comma + 1;
        }
    }
    if (out.empty()) out.push_back(default_device());
    return out;
}

bool compat_anneal_bug()
{
    const char *e = std::getenv("ISINGMC_COMPAT_ANNEAL_BUG");
    return e && e[0] && e[0] != '0';
}

struct GraphHandle {
    isingmc_graph *g = nullptr;
    ~GraphHandle() { isingmc_graph_destroy(g); }
};

struct StatesHandle {
    isingmc_states *s = nullptr;
    std::shared_ptr<GraphHandle> graph; // the graph must outlive the states
    ~StatesHandle() { isingmc_states_destroy(s); }
};

struct EdgeArrays {
    std::vector<uint64_t> a, b;
    std::vector<double> j;
    size_t nvars = 0;
};

// [((a, b), j), ...] -> three arrays, straight off the CPython objects: the generic pybind11 caster
// takes tens of seconds on the 3.4e7 edges of a 4096^2 lattice, this loop about one.
EdgeArrays split_edges(const py::object &edges)
{
    PyObject *seq = PySequence_Fast(edges.ptr(), "edges must be a sequence of ((a, b), j) entries");
    if (!seq) throw py::error_already_set();
    const py::object guard = py::reinterpret_steal<py::object>(seq);
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    EdgeArrays E;
    E.a.resize(n);
    E.b.resize(n);
    E.j.resize(n);
    auto pair_of = [](PyObject *o, PyObject *&x, PyObject *&y) {
        if (PyTuple_Check(o) && PyTuple_GET_SIZE(o) == 2) { x = PyTuple_GET_ITEM(o, 0); y = PyTuple_GET_ITEM(o, 1); return true; }
        if (PyList_Check(o) && PyList_GET_SIZE(o) == 2) { x = PyList_GET_ITEM(o, 0); y = PyList_GET_ITEM(o, 1); return true; }
        return false;
    };
    for (Py_ssize_t k = 0; k < n; k++) {
        PyObject *ab = nullptr, *j = nullptr, *a = nullptr, *b = nullptr;
        if (!pair_of(items[k], ab, j) || !pair_of(ab, a, b))
            throw py::type_error("edges must be ((a, b), j) entries: bad entry " + std::to_string(k));
        const unsigned long long ua = PyLong_AsUnsignedLongLong(a), ub = PyLong_AsUnsignedLongLong(b);
        const double dj = PyFloat_AsDouble(j);
        if (PyErr_Occurred()) throw py::error_already_set();
        E.a[k] = ua;
        E.b[k] = ub;
        E.j[k] = dj;
        E.nvars = std::max<size_t>(E.nvars, std::max(ua, ub) + 1); // lattice.rs:51-55
    }
    return E;
}

std::shared_ptr<GraphHandle> make_graph(const EdgeArrays &E, const std::vector<double> *biases, int device,
                                        bool force_general)
{
    auto h = std::make_shared<GraphHandle>();
    py::gil_scoped_release nogil;
    // The sign of the bias term lives in the un-vendored crate (unverified convention).  The library's
    // Hamiltonian is E = sum J s s - sum h s (a positive bias favours True); ISINGMC_COMPAT_BIAS_SIGN=-1 hands it -h, i.e.
    // E = sum J s s + sum h s, should the crate turn out to use that sign.
    std::vector<double> flipped;
    const char *sign = std::getenv("ISINGMC_COMPAT_BIAS_SIGN");
    if (biases && sign && std::atoi(sign) < 0) {
        flipped.resize(biases->size());
        for (size_t i = 0; i < flipped.size(); i++) flipped[i] = -(*biases)[i];
        biases = &flipped;
    }
    check(isingmc_graph_create(E.a.data(), E.b.data(), E.j.data(), E.a.size(), E.nvars,
                               biases ? biases->data() : nullptr, device,
                               force_general ? ISINGMC_FLAG_FORCE_GENERAL : 0u, &h->g));
    return h;
}

std::vector<uint8_t> to_bytes(const std::vector<bool> &v)
{
    return std::vector<uint8_t>(v.begin(), v.end());
}

using Range = std::optional<std::pair<size_t, size_t>>;

// ------------------------------------------------------------------------------------------------
// Lattice (src/lattice.rs:27-470, classical methods)
// ------------------------------------------------------------------------------------------------
class Lattice {
public:
    Lattice(const py::object &edges, std::optional<uint64_t> seed_gen, std::optional<bool> use_allocator)
        : E_(std::make_shared<EdgeArrays>(split_edges(edges))), seed_gen_(seed_gen),
          use_allocator_(use_allocator.value_or(true)), devices_(default_devices())
    {
        if (E_->a.empty()) throw py::value_error("Must supply some edges for graph"); // lattice.rs:70-72
    }

    // extension: numpy ingest without building 10^7 Python tuples
    static Lattice from_arrays(py::array_t<uint64_t, py::array::c_style | py::array::forcecast> a,
                               py::array_t<uint64_t, py::array::c_style | py::array::forcecast> b,
                               py::array_t<double, py::array::c_style | py::array::forcecast> j,
                               std::optional<uint64_t> seed_gen)
    {
        if (a.ndim() != 1 || a.size() != b.size() || a.size() != j.size())
            throw py::value_error("edge arrays must be 1-d and of equal length");
        if (a.size() == 0) throw py::value_error("Must supply some edges for graph");
        Lattice L;
        auto E = std::make_shared<EdgeArrays>();
        E->a.assign(a.data(), a.data() + a.size());
        E->b.assign(b.data(), b.data() + b.size());
        E->j.assign(j.data(), j.data() + j.size());
        for (ssize_t k = 0; k < a.size(); k++) E->nvars = std::max<size_t>(E->nvars, std::max(E->a[k], E->b[k]) + 1);
        L.E_ = E;
        L.seed_gen_ = seed_gen;
        L.devices_ = default_devices();
        return L;
    }

    void set_seed_gen(std::optional<uint64_t> seed_gen) { seed_gen_ = seed_gen; } // lattice.rs:78-80

    std::vector<uint64_t> make_seeds(size_t num_experiments) const // lattice.rs:83-91
    {
        std::vector<uint64_t> seeds(num_experiments);
        check(isingmc_host_make_seeds(seed_gen_.has_value(), seed_gen_.value_or(0), num_experiments, seeds.data()));
        return seeds;
    }

    void set_enable_rvb_update(bool v) { enable_rvb_ = v; }      // lattice.rs:94-96 (QMC only; stored)
    void set_enable_heatbath_update(bool v) { enable_heatbath_ = v; } // lattice.rs:99-101

    void set_individual_bias(size_t var, double bias) // lattice.rs:104-126
    {
        if (var >= E_->nvars)
            throw py::value_error("Index out of bounds: variable " + std::to_string(var) + " out of " +
                                  std::to_string(E_->nvars));
        if (biases_.empty()) biases_.assign(E_->nvars, global_bias_);
        biases_[var] = bias;
        graphs_.clear();
    }

    void set_global_bias(double bias) // lattice.rs:129-131
    {
        biases_.clear();
        global_bias_ = bias;
        graphs_.clear();
    }

    void set_transverse_field(double transverse) // lattice.rs:134-146
    {
        if (transverse > 0.0) transverse_ = transverse;
        else if (transverse == 0.0) transverse_.reset();
        else throw py::value_error("Transverse field must be positive");
    }

    void set_initial_state(const std::vector<bool> &initial_state) // lattice.rs:149-161
    {
        if (initial_state.size() == E_->nvars) initial_state_ = to_bytes(initial_state);
        else if (initial_state.empty()) initial_state_.clear();
        else throw py::value_error("Initial state must be of the same size as biases, or 0.");
    }

    // extensions: device ordinal, and forcing the general edge-list path (BASELINE config c5)
    void set_device(int device) { devices_ = {device}; graphs_.clear(); }
    void set_devices(const std::vector<int> &devices) // extension: the device list of the in-process fan-out
    {
        if (devices.empty()) throw py::value_error("the device list must not be empty");
        devices_ = devices;
        graphs_.clear();
    }
    std::vector<int> get_devices() const { return devices_; }
    void set_force_general_path(bool v) { force_general_ = v; graphs_.clear(); }
    py::dict engine_info()
    {
        isingmc_graph_info_t info;
        check(isingmc_graph_info(graph()->g, &info));
        py::dict d;
        d["kind"] = info.kind == ISINGMC_KIND_LATTICE2D ? "lattice2d" : "general";
        d["device"] = info.device;
        d["nvars"] = info.nvars;
        d["width"] = info.width;
        d["height"] = info.height;
        d["n_colours"] = info.n_colours;
        d["uniform_sign"] = bool(info.uniform_sign);
        d["field"] = info.field;
        d["open_x"] = bool(info.open_x);
        d["open_y"] = bool(info.open_y);
        d["packed_degree"] = info.packed_degree;
        d["real_slots"] = info.real_slots; // 4 / 7 / 11 / 15: the real-coupling packed path applies (graphs of >= 8 000 sites: from 6 experiments on)
        d["real_quantum_log2"] = info.real_quantum_log2;
        return d;
    }

    // lattice.rs:171-221
    py::tuple run_monte_carlo(double beta, size_t timesteps, size_t num_experiments, std::optional<bool>,
                              std::optional<bool>, Range replica_range)
    {
        require_classical();
        const auto [lo, hi] = bounds(num_experiments, replica_range);
        const size_t R = hi - lo, N = E_->nvars;
        py::array_t<double> energies(std::vector<ssize_t>{ssize_t(R)});
        py::array_t<bool> states(std::vector<ssize_t>{ssize_t(R), ssize_t(N)});
        double *e = energies.mutable_data();
        uint8_t *st = reinterpret_cast<uint8_t *>(states.mutable_data());
        fan_out(num_experiments, lo, hi, [&](isingmc_states *s, size_t off) {
            int rc = isingmc_do_time_steps(s, timesteps, &beta, 0, nullptr);
            if (rc == ISINGMC_OK) rc = isingmc_get_energies(s, e + off);
            if (rc == ISINGMC_OK) rc = isingmc_get_states(s, st + off * N, N);
            return rc;
        });
        return py::make_tuple(energies, states);
    }

    // lattice.rs:231-299
    py::tuple run_monte_carlo_sampling(double beta, size_t timesteps, size_t num_experiments, std::optional<bool>,
                                       std::optional<size_t> thermalization_time, std::optional<size_t> sampling_freq,
                                       std::optional<bool>, Range replica_range)
    {
        require_classical();
        const size_t therm = thermalization_time.value_or(0), freq = sampling_freq.value_or(1);
        if (freq == 0) throw py::value_error("sampling_freq must be positive");
        const size_t S = timesteps / freq; // lattice.rs:247
        const auto [lo, hi] = bounds(num_experiments, replica_range);
        const size_t R = hi - lo, N = E_->nvars;
        py::array_t<double> energies(std::vector<ssize_t>{ssize_t(R), ssize_t(S)});
        py::array_t<bool> states(std::vector<ssize_t>{ssize_t(R), ssize_t(S), ssize_t(N)});
        double *e = energies.mutable_data();
        uint8_t *st = reinterpret_cast<uint8_t *>(states.mutable_data());
        // lattice.rs:271-287: thermalise, then S x { freq steps; record state + energy } -- one library call per shard
        fan_out(num_experiments, lo, hi, [&](isingmc_states *s, size_t off) {
            return isingmc_run_sampling(s, beta, therm, freq, S, e + off * S, st + off * S * N);
        });
        return py::make_tuple(energies, states);
    }

    // lattice.rs:309-385
    py::tuple run_monte_carlo_annealing(const std::vector<std::pair<size_t, double>> &betas, size_t timesteps,
                                        size_t num_experiments, std::optional<bool>, std::optional<bool>,
                                        Range replica_range)
    {
        require_classical();
        const std::vector<double> schedule = expand(betas, timesteps);
        const auto [lo, hi] = bounds(num_experiments, replica_range);
        const size_t R = hi - lo, N = E_->nvars;
        py::array_t<double> energies(std::vector<ssize_t>{ssize_t(R)});
        py::array_t<bool> states(std::vector<ssize_t>{ssize_t(R), ssize_t(N)});
        double *e = energies.mutable_data();
        uint8_t *st = reinterpret_cast<uint8_t *>(states.mutable_data());
        fan_out(num_experiments, lo, hi, [&](isingmc_states *s, size_t off) {
            int rc = isingmc_do_time_steps(s, timesteps, schedule.data(), 1, nullptr);
            if (rc == ISINGMC_OK) rc = isingmc_get_energies(s, e + off);
            if (rc == ISINGMC_OK) rc = isingmc_get_states(s, st + off * N, N);
            return rc;
        });
        return py::make_tuple(energies, states);
    }

    // lattice.rs:395-470
    py::tuple run_monte_carlo_annealing_and_get_energies(const std::vector<std::pair<size_t, double>> &betas,
                                                         size_t timesteps, size_t num_experiments,
                                                         std::optional<bool>, std::optional<bool>, Range replica_range)
    {
        require_classical();
        const std::vector<double> schedule = expand(betas, timesteps);
        const auto [lo, hi] = bounds(num_experiments, replica_range);
        const size_t R = hi - lo, N = E_->nvars;
        py::array_t<double> energies(std::vector<ssize_t>{ssize_t(R), ssize_t(timesteps)});
        py::array_t<bool> states(std::vector<ssize_t>{ssize_t(R), ssize_t(N)});
        double *e = energies.mutable_data();
        uint8_t *st = reinterpret_cast<uint8_t *>(states.mutable_data());
        fan_out(num_experiments, lo, hi, [&](isingmc_states *s, size_t off) {
            int rc = isingmc_do_time_steps(s, timesteps, schedule.data(), 1, e + off * timesteps);
            if (rc == ISINGMC_OK) rc = isingmc_get_states(s, st + off * N, N);
            return rc;
        });
        return py::make_tuple(energies, states);
    }

    Lattice clone() const { return *this; } // lattice.rs:1038-1040 (the immutable device graph is shared)

    static void sample_into(isingmc_states *s, double beta, size_t therm, size_t freq, size_t S, size_t, size_t,
                            double *energies, uint8_t *states)
    {
        // lattice.rs:271-287: thermalise, then S x { freq steps; record state + energy } -- one library call,
        // enqueued end to end on the device stream
        check(isingmc_run_sampling(s, beta, therm, freq, S, energies, states));
    }

private:
    Lattice() = default;

    void require_classical() const
    {
        if (transverse_) // lattice.rs:217-219
            throw py::value_error("Cannot run classic monte carlo with transverse field");
    }

    // one graph per entry of the device list, built on first use (the first one serves engine_info)
    std::shared_ptr<GraphHandle> graph(size_t slot = 0)
    {
        if (graphs_.size() != devices_.size()) graphs_.assign(devices_.size(), nullptr);
        if (!graphs_[slot]) {
            std::vector<double> b;
            const std::vector<double> *bp = nullptr;
            if (!biases_.empty()) bp = &biases_;
            else if (global_bias_ != 0.0) { b.assign(E_->nvars, global_bias_); bp = &b; } // lattice.rs:186-189
            graphs_[slot] = make_graph(*E_, bp, devices_[slot], force_general_);
        }
        return graphs_[slot];
    }

    static std::pair<size_t, size_t> bounds(size_t num_experiments, const Range &range)
    {
        if (!range) return {0, num_experiments};
        if (range->first > range->second || range->second > num_experiments) throw py::value_error("replica_range out of bounds");
        return {range->first, range->second};
    }

    // The rayon fan-out of lattice.rs:192-197, over devices: experiments [lo, hi) are cut into contiguous blocks,
    // one per entry of the device list (ISINGMC_DEVICES / set_devices; aligned to 32 experiments once a block
    // holds that many), and every block runs on its device from a host thread of its own:
    //   seed -> rng; GraphState::new; set_state(initial)  (lattice.rs:198-203) = isingmc_states_create_range,
    // then body(states, offset of the block in the output arrays).  Philox keys, replica groups and the path
    // choice follow the GLOBAL experiment index, so the arrays do not depend on the device list.
    template <typename Body>
    void fan_out(size_t num_experiments, size_t lo, size_t hi, Body &&body)
    {
        const std::vector<uint64_t> seeds = make_seeds(num_experiments);
        const size_t n = hi - lo, D = devices_.size();
        size_t per = (n + D - 1) / std::max<size_t>(D, 1);
        if (per >= 32) per = (per + 31) / 32 * 32;
        struct Block { size_t slot, lo, hi; int rc = ISINGMC_OK; std::string msg; };
        std::vector<Block> blocks;
        for (size_t d = 0; d < D; d++) {
            const size_t b = std::min(hi, lo + d * per), e = std::min(hi, b + per);
            if (e > b || (d == 0 && n == 0)) blocks.push_back({d, b, e});
        }
        for (const Block &blk : blocks) (void)graph(blk.slot); // may raise: with the GIL, before any thread starts
        const uint8_t *ini = initial_state_.empty() ? nullptr : initial_state_.data();
        const auto run_block = [&](Block &blk) {
            isingmc_states *st = nullptr;
            blk.rc = isingmc_states_create_range(graphs_[blk.slot]->g, num_experiments, seeds.data(), blk.lo, blk.hi - blk.lo, ini, &st);
            if (blk.rc == ISINGMC_OK) blk.rc = body(st, blk.lo - lo);
            if (blk.rc != ISINGMC_OK) blk.msg = isingmc_last_error(); // per thread: read it where it was set
            isingmc_states_destroy(st);
        };
        {
            py::gil_scoped_release nogil;
            if (blocks.size() == 1) run_block(blocks[0]);
            else {
                std::vector<std::thread> pool;
                for (Block &blk : blocks) pool.emplace_back([&run_block, &blk] { run_block(blk); });
                for (auto &th : pool) th.join();
            }
        }
        for (const Block &blk : blocks) {
            if (blk.rc == ISINGMC_OK) continue;
            if (blk.rc == ISINGMC_ERR_INVALID) throw py::value_error(blk.msg);
            if (blk.rc == ISINGMC_ERR_ALLOC) throw std::bad_alloc();
            throw std::runtime_error(blk.msg);
        }
    }

    static std::vector<double> expand(const std::vector<std::pair<size_t, double>> &betas, size_t timesteps)
    {
        std::vector<uint64_t> t;
        std::vector<double> b;
        for (const auto &s : betas) { t.push_back(s.first); b.push_back(s.second); }
        std::vector<double> out(timesteps);
        check(isingmc_host_expand_schedule(t.data(), b.data(), t.size(), timesteps, compat_anneal_bug(), out.data()));
        return out;
    }

    std::shared_ptr<EdgeArrays> E_;
    std::vector<double> biases_; // empty = BiasType::Global(global_bias_)
    double global_bias_ = 0.0;
    std::optional<double> transverse_;
    std::vector<uint8_t> initial_state_;
    bool enable_rvb_ = false, enable_heatbath_ = false;
    std::optional<uint64_t> seed_gen_;
    bool use_allocator_ = true;
    std::vector<int> devices_;  // one block of experiments per entry (ISINGMC_DEVICES; an ordinal may repeat)
    bool force_general_ = false;
    std::vector<std::shared_ptr<GraphHandle>> graphs_;
};

// ------------------------------------------------------------------------------------------------
// ClassicIsing (src/classicising.rs:11-180): replicas persist on the device between calls
// ------------------------------------------------------------------------------------------------
class ClassicIsing {
public:
    ClassicIsing(const py::object &edges, std::optional<double> longitudinal,
                 std::optional<size_t> num_experiments, std::optional<uint64_t> seed, std::optional<bool> use_basic_moves)
        : E_(split_edges(edges)), longitudinal_(longitudinal.value_or(0.0)),
          use_basic_moves_(use_basic_moves.value_or(false))
    {
        // the reference unwraps None here and aborts the process (classicising.rs:34-39)
        if (E_.a.empty()) throw py::value_error("Must supply some edges for graph");
        if (seed) master_seed_ = *seed;
        else check(isingmc_host_make_seeds(0, 0, 1, &master_seed_)); // SmallRng::from_entropy()
        std::vector<double> bias;
        if (longitudinal_ != 0.0) bias.assign(E_.nvars, longitudinal_); // classicising.rs:69
        graph_ = make_graph(E_, bias.empty() ? nullptr : &bias, default_device(), false);
        st_ = std::make_shared<StatesHandle>();
        st_->graph = graph_;
        // all experiments of the constructor at once (seed i = the i-th draw of the container's rng, as add_graph would
        // draw them one by one): the library picks its path from the count and the graph size -- a graph that is not a
        // recognised lattice usually runs on the replica-packed kernels (isingmc.hip packed_worth_it) -- and later add_graph calls grow that container
        const size_t n = num_experiments.value_or(1);
        drawn_.resize(n);
        check(isingmc_host_make_seeds(1, master_seed_, n, drawn_.data()));
        check(isingmc_states_create(graph_->g, n, drawn_.data(), nullptr, &st_->s));
    }

    // classicising.rs:62-79: seed = self.rng.gen(); GraphState::new / new_with_state_and_rng
    void add_graph(std::optional<std::vector<bool>> initial_state, std::optional<bool>)
    {
        std::vector<uint8_t> ini;
        if (initial_state) {
            if (initial_state->size() != E_.nvars)
                throw py::value_error("Initial state must be of the same size as biases, or 0.");
            ini = to_bytes(*initial_state);
        }
        drawn_.resize(drawn_.size() + 1);
        check(isingmc_host_make_seeds(1, master_seed_, drawn_.size(), drawn_.data())); // n-th draw of the master rng
        py::gil_scoped_release nogil;
        check(isingmc_states_append(st_->s, drawn_.back(), initial_state ? ini.data() : nullptr));
    }

    // classicising.rs:88-110
    void run_monte_carlo(double beta, size_t timesteps, std::optional<size_t> nspinupdates, std::optional<size_t>,
                         std::optional<size_t>, std::optional<bool>)
    {
        const size_t sweeps = sweeps_for(timesteps, nspinupdates);
        py::gil_scoped_release nogil;
        check(isingmc_do_time_steps(st_->s, sweeps, &beta, 0, nullptr));
    }

    // classicising.rs:119-179
    py::tuple run_monte_carlo_sampling(double beta, size_t timesteps, std::optional<size_t> nspinupdates,
                                       std::optional<size_t>, std::optional<size_t>, std::optional<bool>,
                                       std::optional<size_t> thermalization_time, std::optional<size_t> sampling_freq)
    {
        const size_t therm = thermalization_time.value_or(0), freq = sampling_freq.value_or(1);
        if (freq == 0) throw py::value_error("sampling_freq must be positive");
        const size_t S = timesteps / freq, R = isingmc_states_count(st_->s), N = E_.nvars;
        py::array_t<double> energies(std::vector<ssize_t>{ssize_t(R), ssize_t(S)});
        py::array_t<bool> states(std::vector<ssize_t>{ssize_t(R), ssize_t(S), ssize_t(N)});
        if (whole_sweeps(nspinupdates)) { // every timestep is made of whole sweeps: one pipelined library call
            const size_t mult = nspinupdates ? *nspinupdates / N : 1;
            double *e_out = energies.mutable_data();
            uint8_t *s_out = reinterpret_cast<uint8_t *>(states.mutable_data());
            {
                py::gil_scoped_release nogil;
                Lattice::sample_into(st_->s, beta, therm * mult, freq * mult, S, R, N, e_out, s_out);
            }
            return py::make_tuple(energies, states);
        }
        // attempts per timestep that are not whole sweeps: the blocks between samples hold varying numbers of sweeps
        std::vector<size_t> block(S + 1);
        block[0] = sweeps_for(therm, nspinupdates);
        for (size_t k = 0; k < S; k++) block[k + 1] = sweeps_for(freq, nspinupdates);
        double *e = energies.mutable_data();
        uint8_t *st = reinterpret_cast<uint8_t *>(states.mutable_data());
        {
            py::gil_scoped_release nogil;
            std::vector<double> e_k(R);
            check(isingmc_do_time_steps(st_->s, block[0], &beta, 0, nullptr));
            for (size_t k = 0; k < S; k++) {
                check(isingmc_do_time_steps(st_->s, block[k + 1], &beta, 0, nullptr));
                check(isingmc_get_energies(st_->s, e_k.data()));
                for (size_t r = 0; r < R; r++) e[r * S + k] = e_k[r];
                if (R) check(isingmc_get_states(st_->s, st + k * N, S * N)); // replica r's sample k sits at (r S + k) N
            }
        }
        return py::make_tuple(energies, states);
    }

    // extensions: read the persistent replicas without advancing them
    py::array_t<double> get_energies()
    {
        py::array_t<double> e(std::vector<ssize_t>{ssize_t(isingmc_states_count(st_->s))});
        check(isingmc_get_energies(st_->s, e.mutable_data()));
        return e;
    }
    py::array_t<bool> get_states()
    {
        py::array_t<bool> s(std::vector<ssize_t>{ssize_t(isingmc_states_count(st_->s)), ssize_t(E_.nvars)});
        check(isingmc_get_states(st_->s, reinterpret_cast<uint8_t *>(s.mutable_data()), E_.nvars));
        return s;
    }
    size_t get_num_graphs() const { return isingmc_states_count(st_->s); }

private:
    // nspinupdates = single-spin attempts per timestep (classicising.rs:88-110 hands it to do_time_step; crate default:
    // nvars).  The engine attempts every site once per sweep, in the colour order, so attempts are executed sweep by
    // sweep: `timesteps` timesteps of n attempts add timesteps x n attempts to a cursor that persists across calls,
    // every nvars accumulated attempts run as one sweep, and the remainder (< nvars attempts) stays pending for the next
    // call.  Whole multiples of nvars are exactly that many sweeps per timestep; any other positive count is honoured on
    // average (the total number of attempts is exact up to the pending remainder), with a one-off warning.
    bool whole_sweeps(const std::optional<size_t> &nspinupdates) const
    {
        return !nspinupdates || (*nspinupdates > 0 && *nspinupdates % E_.nvars == 0 && pending_attempts_ == 0);
    }
    size_t sweeps_for(size_t timesteps, const std::optional<size_t> &nspinupdates)
    {
        if (!nspinupdates) return timesteps;
        if (*nspinupdates == 0) throw py::value_error("nspinupdates must be positive");
        if (*nspinupdates % E_.nvars != 0 && !warned_partial_) {
            warned_partial_ = true;
            if (PyErr_WarnEx(PyExc_UserWarning,
                             "nspinupdates is not a multiple of the number of variables: attempts are executed sweep by sweep "
                             "(every site once, in the colour order); the attempts that do not complete a sweep stay pending "
                             "and count towards the next timesteps", 1) < 0)
                throw py::error_already_set();
        }
        const unsigned __int128 total = (unsigned __int128)timesteps * *nspinupdates + pending_attempts_;
        const unsigned __int128 sweeps = total / E_.nvars;
        if (sweeps > (unsigned __int128)std::numeric_limits<size_t>::max() / 4) throw py::value_error("too many spin updates");
        pending_attempts_ = size_t(total % E_.nvars);
        return size_t(sweeps);
    }

    size_t pending_attempts_ = 0;
    bool warned_partial_ = false;
    EdgeArrays E_;
    double longitudinal_;
    bool use_basic_moves_; // stored, never read -- as in the reference (classicising.rs:45,54)
    uint64_t master_seed_ = 0;
    std::vector<uint64_t> drawn_;
    std::shared_ptr<GraphHandle> graph_;
    std::shared_ptr<StatesHandle> st_;
};

} // namespace

PYBIND11_MODULE(_py_monte_carlo, m)
{
    m.doc() = "MI355X-native classical Ising Metropolis engine behind the py_monte_carlo API";
    using namespace py::literals;

    py::class_<Lattice>(m, "Lattice")
        .def(py::init<const py::object &, std::optional<uint64_t>, std::optional<bool>>(), "edges"_a,
             "seed_gen"_a = py::none(), "use_allocator"_a = py::none())
        .def_static("from_arrays", &Lattice::from_arrays, "edge_a"_a, "edge_b"_a, "edge_j"_a, "seed_gen"_a = py::none())
        .def("set_seed_gen", &Lattice::set_seed_gen, "seed_gen"_a = py::none())
        .def("make_seeds", &Lattice::make_seeds, "num_experiments"_a)
        .def("set_enable_rvb_update", &Lattice::set_enable_rvb_update, "enable_updates"_a)
        .def("set_enable_heatbath_update", &Lattice::set_enable_heatbath_update, "enable_heatbath"_a)
        .def("set_individual_bias", &Lattice::set_individual_bias, "var"_a, "bias"_a)
        .def("set_global_bias", &Lattice::set_global_bias, "bias"_a)
        .def("set_transverse_field", &Lattice::set_transverse_field, "transverse"_a)
        .def("set_initial_state", &Lattice::set_initial_state, "initial_state"_a)
        .def("set_device", &Lattice::set_device, "device"_a)
        .def("set_devices", &Lattice::set_devices, "devices"_a)
        .def("get_devices", &Lattice::get_devices)
        .def("set_force_general_path", &Lattice::set_force_general_path, "force"_a)
        .def("engine_info", &Lattice::engine_info)
        .def("run_monte_carlo", &Lattice::run_monte_carlo, "beta"_a, "timesteps"_a, "num_experiments"_a,
             "only_basic_moves"_a = py::none(), "edge_move_importance_sampling"_a = py::none(),
             "replica_range"_a = py::none())
        .def("run_monte_carlo_sampling", &Lattice::run_monte_carlo_sampling, "beta"_a, "timesteps"_a,
             "num_experiments"_a, "only_basic_moves"_a = py::none(), "thermalization_time"_a = py::none(),
             "sampling_freq"_a = py::none(), "edge_move_importance_sampling"_a = py::none(),
             "replica_range"_a = py::none())
        .def("run_monte_carlo_annealing", &Lattice::run_monte_carlo_annealing, "betas"_a, "timesteps"_a,
             "num_experiments"_a, "only_basic_moves"_a = py::none(), "edge_move_importance_sampling"_a = py::none(),
             "replica_range"_a = py::none())
        .def("run_monte_carlo_annealing_and_get_energies", &Lattice::run_monte_carlo_annealing_and_get_energies,
             "betas"_a, "timesteps"_a, "num_experiments"_a, "only_basic_moves"_a = py::none(),
             "edge_move_importance_sampling"_a = py::none(), "replica_range"_a = py::none())
        .def("clone", &Lattice::clone);
    // the reference's quantum (SSE) entry points (lattice.rs:478-1036) live in the un-vendored qmc crate and are out of
    // scope: present by name, so a script written for the reference fails with a reason instead of an AttributeError
    for (const char *name : {"run_quantum_monte_carlo", "run_quantum_monte_carlo_sampling",
                             "run_quantum_monte_carlo_and_measure_variable_autocorrelation",
                             "run_quantum_monte_carlo_and_measure_spin_product_autocorrelation",
                             "run_quantum_monte_carlo_and_measure_bond_autocorrelation",
                             "run_quantum_monte_carlo_and_measure_spins", "get_offset", "average_on_and_off_diagonal_and_consts"}) {
        const std::string what = std::string("Lattice.") + name + ": quantum (SSE) Monte Carlo is not part of this build (classical Metropolis only)";
        py::setattr(m.attr("Lattice"), name, py::cpp_function([what](const py::args &, const py::kwargs &) -> py::object {
                        PyErr_SetString(PyExc_NotImplementedError, what.c_str());
                        throw py::error_already_set();
                    }, py::is_method(m.attr("Lattice"))));
    }

    py::class_<ClassicIsing>(m, "ClassicIsing")
        .def(py::init<const py::object &, std::optional<double>, std::optional<size_t>, std::optional<uint64_t>,
                      std::optional<bool>>(),
             "edges"_a, "longitudinal"_a = py::none(), "num_experiments"_a = py::none(), "seed"_a = py::none(),
             "use_basic_moves"_a = py::none())
        .def("add_graph", &ClassicIsing::add_graph, "initial_state"_a = py::none(),
             "edge_move_importance_sampling"_a = py::none())
        .def("run_monte_carlo", &ClassicIsing::run_monte_carlo, "beta"_a, "timesteps"_a, "nspinupdates"_a = py::none(),
             "nedgeupdates"_a = py::none(), "nwormupdates"_a = py::none(), "only_basic_moves"_a = py::none())
        .def("run_monte_carlo_sampling", &ClassicIsing::run_monte_carlo_sampling, "beta"_a, "timesteps"_a,
             "nspinupdates"_a = py::none(), "nedgeupdates"_a = py::none(), "nwormupdates"_a = py::none(),
             "only_basic_moves"_a = py::none(), "thermalization_time"_a = py::none(), "sampling_freq"_a = py::none())
        .def("get_energies", &ClassicIsing::get_energies)
        .def("get_states", &ClassicIsing::get_states)
        .def("get_num_graphs", &ClassicIsing::get_num_graphs);
}
