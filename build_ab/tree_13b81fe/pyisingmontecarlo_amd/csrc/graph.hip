// libisingmc.so: graphs -- edge-list checks, the host-only helpers of the C ABI, and the construction of the device views of a
// graph (checkerboard lattice, coloured CSR, replica-packed ELL, real-coupling ELL).  Host code only: no kernel is launched here.
#include "internal.hpp"

// ------------------------------------------------------------------------------------------------
// small host helpers
// ------------------------------------------------------------------------------------------------
// acceptance probability as a THR_BITS-bit fixed-point threshold: accept iff u < T, u uniform on
// [0, 2^THR_BITS); T = 2^THR_BITS accepts always (dE <= 0, or beta < 0)
uint64_t threshold_fixed(double beta, double dE)
{
    const uint64_t ONE = uint64_t(1) << THR_BITS;
    if (dE <= 0.0) return ONE;
    const double p = std::exp(-beta * dE);
    if (!(p < 1.0)) return ONE;
    return uint64_t(std::floor(std::ldexp(p, THR_BITS)));
}

LatThr lattice_thresholds(double beta, double jabs)
{
    return LatThr{threshold_fixed(beta, 4.0 * jabs), threshold_fixed(beta, 8.0 * jabs)};
}

// thresholds of the multi-class kernels (classes: mc_types.hpp); same fixed-point rule, same exp as the two-class ones
LatThrMC lattice_thresholds_mc(const isingmc_graph *g, double beta)
{
    LatThrMC t{};
    const int nc = g->mc_mode == MC_FIELD_OPEN ? 9 : g->mc_mode == MC_FIELD ? 6 : g->mc_mode == MC_ANISO ? 5 : 4;
    for (int c = 0; c < nc; c++) {
        double dE;
        if (g->mc_mode == MC_FIELD_OPEN) { // classes by m = sat - unsat and sigma = spin x sign of the site's field: |h| here
            const int m = c < 8 ? 1 + c / 2 : 0;
            const double sval = (c == 8 || (c & 1)) ? 1.0 : -1.0;
            dE = 2.0 * g->jabs * double(m) + 2.0 * std::fabs(g->field) * sval;
        } else if (g->mc_mode == MC_ANISO) {
            static const int mx[5] = {2, 2, 0, 2, -2}, my[5] = {2, 0, 2, -2, 2}; // (sat - unsat) per direction of the classes
            dE = 2.0 * g->jabs * double(mx[c]) + 2.0 * g->jabs_y * double(my[c]); // the oracle's expression, term by term
        } else if (g->mc_mode == MC_FIELD) {
            const int k = 2 + c / 2;
            const double sval = (c & 1) ? 1.0 : -1.0;
            dE = 2.0 * g->jabs * double(2 * k - 4) + 2.0 * g->field * sval; // the oracle's expression: 2|J|(sat - unsat) + 2 h s
        } else {
            dE = 2.0 * g->jabs * double(c + 1);
        }
        const uint64_t T = threshold_fixed(beta, dE);
        if (!(T >> THR_BITS)) t.costly |= 1u << c;
        t.hi[c] = uint32_t(T >> 32) & ((1u << N_PLANES) - 1);
        t.lo[c] = uint32_t(T);
    }
    return t;
}

// lattice energy from the integer counters: E = |J| (bonds - 2 satisfied) - h (2 up - N)   (exact in f64 for h = 0)
double lattice_energy(const isingmc_graph *g, unsigned long long sat, unsigned long long up)
{
    if (g->mc_mode == MC_ANISO) { // sat = satisfied horizontal | satisfied vertical << 32; N bonds per direction
        const int64_t n = int64_t(g->nvars), sx = int64_t(sat & 0xFFFFFFFFull), sy = int64_t(sat >> 32);
        return g->jabs * double(n - 2 * sx) + g->jabs_y * double(n - 2 * sy);
    }
    if (g->d_fneg) { // sat = satisfied bonds | spins along their site's field << 32; field = |h|
        const int64_t k = int64_t(sat & 0xFFFFFFFFull), along = int64_t(sat >> 32);
        return g->jabs * double(int64_t(g->n_edges) - 2 * k) - g->field * double(2 * along - int64_t(g->nvars));
    }
    const double bonds = g->jabs * double(int64_t(g->n_edges) - 2 * int64_t(sat));
    if (g->mc_mode != MC_FIELD && g->mc_mode != MC_FIELD_OPEN) return bonds;
    return bonds - g->field * double(2 * int64_t(up) - int64_t(g->nvars));
}

// bytes (site order) -> packed words of one replica
void pack_state(const isingmc_graph *g, const uint8_t *spins, uint32_t *words)
{
    std::fill(words, words + g->state_words, 0u);
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        const LatGeom &L = g->geom;
        for (uint32_t y = 0; y < L.H; y++)
            for (uint32_t x = 0; x < L.W; x++)
                if (spins[size_t(y) * L.W + x]) {
                    const uint32_t c = (x + y) & 1, i = x >> 1;
                    words[size_t(c) * L.wpp + size_t(y) * L.wpr + (i >> 5)] |= 1u << (i & 31);
                }
    } else {
        for (uint64_t i = 0; i < g->nvars; i++)
            if (spins[i]) words[g->pos[i] >> 5] |= 1u << (g->pos[i] & 31);
    }
}

// packed words of one replica -> bytes (site order)
void unpack_state(const isingmc_graph *g, const uint32_t *words, uint8_t *spins)
{
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        unpack_lattice(g->geom.W, g->geom.H, words, spins);
    } else {
        for (uint64_t i = 0; i < g->nvars; i++) spins[i] = (words[g->pos[i] >> 5] >> (g->pos[i] & 31)) & 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI: host-only helpers
// ------------------------------------------------------------------------------------------------
extern "C" int isingmc_host_make_seeds(int has_seed, uint64_t seed_gen, size_t n, uint64_t *seeds_out)
{
    if (n && !seeds_out) return fail(ISINGMC_ERR_INVALID, "seeds_out is NULL");
    const auto seeds = make_seeds(has_seed != 0, seed_gen, n);
    std::copy(seeds.begin(), seeds.end(), seeds_out);
    return ISINGMC_OK;
}

extern "C" int isingmc_host_expand_schedule(const uint64_t *stop_t, const double *stop_beta, size_t n_stops,
                                            size_t timesteps, int compat_constant_beta, double *betas_out)
{
    if ((n_stops && (!stop_t || !stop_beta)) || (timesteps && !betas_out))
        return fail(ISINGMC_ERR_INVALID, "NULL schedule argument");
    const std::string msg = expand_schedule(stop_t, stop_beta, n_stops, timesteps, compat_constant_beta != 0, betas_out);
    return msg.empty() ? ISINGMC_OK : fail(ISINGMC_ERR_INVALID, msg);
}

static int check_edges(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges, size_t nvars)
{
    if (n_edges == 0) return fail(ISINGMC_ERR_INVALID, "Must supply some edges for graph"); // lattice.rs:70-72
    if (!ea || !eb || !ej) return fail(ISINGMC_ERR_INVALID, "NULL edge array");
    if (nvars == 0 || nvars > 0xFFFFFFF0ull) return fail(ISINGMC_ERR_INVALID, "nvars out of range (1 .. 2^32-16)");
    for (size_t k = 0; k < n_edges; k++) {
        if (ea[k] >= nvars || eb[k] >= nvars)
            return fail(ISINGMC_ERR_INVALID, "Index out of bounds: edge " + std::to_string(k) + " touches variable " +
                                                 std::to_string(std::max(ea[k], eb[k])) + " out of " + std::to_string(nvars));
        if (!std::isfinite(ej[k])) return fail(ISINGMC_ERR_INVALID, "edge couplings must be finite");
    }
    return ISINGMC_OK;
}

extern "C" int isingmc_host_recognise_lattice2d(const uint64_t *ea, const uint64_t *eb, const double *ej,
                                                size_t n_edges, size_t nvars, int *is_lattice, int *width,
                                                int *height, double *jabs, int *uniform_sign)
{
    if (!is_lattice) return fail(ISINGMC_ERR_INVALID, "is_lattice is NULL");
    TRY(check_edges(ea, eb, ej, n_edges, nvars));
    const Lattice2D L = recognise_lattice2d(ea, eb, ej, n_edges, nvars);
    // bit 0: a W x H lattice; bits 1, 2: open in x, y; bit 3: |J| differs between the directions
    *is_lattice = L.ok ? 1 + 2 * int(L.open_x) + 4 * int(L.open_y) + 8 * int(L.jabs != L.jabs_y) : 0;
    if (width) *width = L.W;
    if (height) *height = L.H;
    if (jabs) *jabs = L.jabs;
    if (uniform_sign) *uniform_sign = L.uniform_sign;
    return ISINGMC_OK;
}

extern "C" int isingmc_host_colour_graph(const uint64_t *ea, const uint64_t *eb, size_t n_edges, size_t nvars,
                                         uint32_t *colours_out, uint32_t *n_colours_out)
{
    std::vector<double> ones(n_edges, 1.0);
    TRY(check_edges(ea, eb, ones.data(), n_edges, nvars));
    const Adjacency A = build_adjacency(ea, eb, ones.data(), n_edges, nvars);
    const Colouring C = greedy_colouring(A, nvars);
    if (colours_out) std::copy(C.colour.begin(), C.colour.end(), colours_out);
    if (n_colours_out) *n_colours_out = C.n_colours;
    return ISINGMC_OK;
}

extern "C" int isingmc_host_pt_swap_round(uint64_t seed, uint64_t round, size_t n_rungs, const double *betas,
                                          const double *slot_energy, uint32_t *perm, uint64_t *swaps_out)
{
    if (n_rungs && (!betas || !slot_energy || !perm)) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    for (size_t i = 0; i < n_rungs; i++)
        if (perm[i] >= n_rungs) return fail(ISINGMC_ERR_INVALID, "perm is not a permutation of the rungs");
    const uint64_t swaps = pt_swap_round(seed, round, n_rungs, betas, slot_energy, perm);
    if (swaps_out) *swaps_out = swaps;
    return ISINGMC_OK;
}

// adjacency order -> input-edge order: edge e is the next unfilled entry of both its ends' rows
template <typename F>
static void for_each_input_edge(const Adjacency &A, const uint64_t *ea, const uint64_t *eb, size_t n_edges, F &&f)
{
    std::vector<uint64_t> fill(A.ptr.begin(), A.ptr.end());
    for (size_t e = 0; e < n_edges; e++) {
        if (ea[e] == eb[e]) { f(e, false, 0, 0); continue; }
        f(e, true, fill[ea[e]], fill[eb[e]]);
        fill[ea[e]]++;
        fill[eb[e]]++;
    }
}

extern "C" int isingmc_host_rj_energy_levels(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges, size_t nvars,
                                             const double *biases, int32_t *jhi_out, int32_t *jlo_out, int32_t *hhi_out,
                                             int32_t *hlo_out, int *k_energy_out)
{
    TRY(check_edges(ea, eb, ej, n_edges, nvars));
    if (biases)
        for (size_t i = 0; i < nvars; i++)
            if (!std::isfinite(biases[i])) return fail(ISINGMC_ERR_INVALID, "biases must be finite");
    const Adjacency A = build_adjacency(ea, eb, ej, n_edges, nvars);
    const RjQuant Q = rj_quantise(A, nvars, biases);
    if (k_energy_out) *k_energy_out = Q.k_energy;
    if (hhi_out) std::copy(Q.hhi.begin(), Q.hhi.end(), hhi_out);
    if (hlo_out) std::copy(Q.hlo.begin(), Q.hlo.end(), hlo_out);
    for_each_input_edge(A, ea, eb, n_edges, [&](size_t e, bool bond, uint64_t ia, uint64_t) {
        if (jhi_out) jhi_out[e] = bond ? Q.jhi[ia] : 0;
        if (jlo_out) jlo_out[e] = bond ? Q.jlo[ia] : 0;
    });
    return ISINGMC_OK;
}

extern "C" int isingmc_host_rj_quantise(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges, size_t nvars,
                                        const double *biases, int32_t *jq_out, int32_t *hq_out, uint8_t *dshift_out, int *k_out,
                                        int *eligible_out)
{
    TRY(check_edges(ea, eb, ej, n_edges, nvars));
    if (biases)
        for (size_t i = 0; i < nvars; i++)
            if (!std::isfinite(biases[i])) return fail(ISINGMC_ERR_INVALID, "biases must be finite");
    const Adjacency A = build_adjacency(ea, eb, ej, n_edges, nvars);
    const RjQuant Q = rj_quantise(A, nvars, biases);
    if (k_out) *k_out = Q.k;
    if (eligible_out) *eligible_out = Q.eligible;
    if (hq_out) std::copy(Q.hq.begin(), Q.hq.end(), hq_out);
    if (dshift_out) std::copy(Q.dshift.begin(), Q.dshift.end(), dshift_out);
    if (jq_out)
        for_each_input_edge(A, ea, eb, n_edges, [&](size_t e, bool bond, uint64_t ia, uint64_t ib) {
            jq_out[2 * e] = bond ? Q.jq[ia] : 0;
            jq_out[2 * e + 1] = bond ? Q.jq[ib] : 0;
        });
    return ISINGMC_OK;
}

extern "C" int isingmc_host_rj_beta(double beta, int k, uint32_t *shift_out, uint32_t *mant_out)
{
    if (!shift_out || !mant_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    if (!std::isfinite(beta)) return fail(ISINGMC_ERR_INVALID, "beta must be finite");
    rj_beta(beta, k, shift_out, mant_out);
    return ISINGMC_OK;
}

extern "C" int isingmc_host_rj_log_table(uint32_t *table_out)
{
    if (!table_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    rj_log_table(table_out);
    return ISINGMC_OK;
}

// ------------------------------------------------------------------------------------------------
// graph
// ------------------------------------------------------------------------------------------------
// the uniform field h (0 without); fields of one size and both signs (h_i = +-h): |h| and *signs = true;
// NaN when the biases differ from site to site in any other way
static double uniform_bias(const double *biases, size_t nvars, bool *signs)
{
    *signs = false;
    if (!biases) return 0.0;
    bool equal = true, same_size = true;
    for (size_t i = 1; i < nvars; i++) {
        equal &= biases[i] == biases[0];
        same_size &= std::fabs(biases[i]) == std::fabs(biases[0]);
    }
    if (equal) return biases[0];
    if (!same_size) return std::numeric_limits<double>::quiet_NaN();
    *signs = true;
    return std::fabs(biases[0]);
}

// Periodic and field-free: the two-class kernels of lattice_kernels.hpp.  A field |h| <= 2|J| (uniform, or +-h from
// site to site) on a periodic lattice, open boundaries without a field or with a field |h| <= |J|, anisotropic couplings
// (periodic, no field): the multi-class kernels (whole quads per row needed).  Anything else (other site-dependent
// biases, larger fields, anisotropy with a field or open boundaries): the general path.
static bool lattice_fast_path_ok(const Lattice2D &L, double h, bool field_signs)
{
    if (!L.ok || L.W % 64 != 0) return false;
    if (std::isnan(h)) return false;
    const bool open = L.open_x || L.open_y, aniso = L.jabs != L.jabs_y;
    if (aniso && (open || h != 0.0)) return false; // anisotropic couplings: periodic and field-free only
    if (field_signs && uint64_t(L.W) * uint64_t(L.H) >= (uint64_t(1) << 31)) return false; // two 32-bit counters in one word
    if (h != 0.0 && !(std::fabs(h) <= 2.0 * L.jabs)) return false;
    // a boundary site with one more unsatisfied than satisfied bond (m = -1) must still flip outright: |h| <= |J| there
    if (open && h != 0.0 && !(std::fabs(h) <= L.jabs)) return false;
    if ((open || h != 0.0 || aniso) && (L.W / 64) % 4 != 0) return false;
    if (aniso && uint64_t(L.W) * uint64_t(L.H) >= (uint64_t(1) << 32)) return false; // two 32-bit bond counters in one word
    const uint64_t wpp = uint64_t(L.H) * uint64_t(L.W / 64);
    // the kernels address a replica through ONE buffer descriptor (int num_records) and 32-bit byte offsets:
    // both planes must fit below 2^31 bytes; larger lattices take the general path
    return wpp % 4 == 0 && 2 * wpp * sizeof(uint32_t) < (uint64_t(1) << 31);
}

static int build_lattice(isingmc_graph *g, const Lattice2D &L, double h, const double *biases, bool field_signs)
{
    g->kind = ISINGMC_KIND_LATTICE2D;
    const bool open = L.open_x || L.open_y;
    g->mc_mode = h != 0.0 ? (open ? MC_FIELD_OPEN : MC_FIELD) : open ? MC_OPEN : L.jabs != L.jabs_y ? MC_ANISO : MC_NONE;
    g->jabs_y = L.jabs_y;
    g->field = h; // signed for a uniform field, |h| with sign planes
    g->open = McOpen{uint32_t(L.open_x), uint32_t(L.open_y), (!field_signs && h < 0.0) ? 0xFFFFFFFFu : 0u};
    LatGeom &G = g->geom;
    G.W = L.W;
    G.H = L.H;
    G.wpr = L.W / 64;
    G.wpp = G.H * G.wpr;
    G.nquads = G.wpp / 4;
    g->vec = G.wpr % 4 == 0;
    G.cols_log2 = -1;
    if (g->vec) { // division-free, parity-uniform thread mapping (thread_to_quad)
        const uint32_t cols = G.wpr / 4;
        if ((cols & (cols - 1)) == 0) {
            int cl = 0;
            while ((1u << cl) < cols) cl++;
            const uint32_t rows_per_pair = cl >= 6 ? 1 : 2 * (64u >> cl);
            if (G.H % rows_per_pair == 0 && G.nquads % 64 == 0) G.cols_log2 = cl;
        }
    }
    g->jabs = L.jabs;
    g->uniform_sign = L.uniform_sign;
    g->jneg_uniform = L.jpos_uniform ? 0u : 0xFFFFFFFFu;
    g->state_words = 2 * uint64_t(G.wpp);
    g->n_colours = 2;
    if (!L.uniform_sign) { // per-bond sign planes in each colour's compact layout
        std::vector<uint32_t> jneg(size_t(8) * G.wpp, 0u);
        const uint32_t W = G.W, H = G.H;
        for (uint32_t c = 0; c < 2; c++)
            for (uint32_t y = 0; y < H; y++) {
                const uint32_t yu = (y + H - 1) % H, o = (y + c) & 1;
                for (uint32_t i = 0; i < W / 2; i++) {
                    const uint32_t x = 2 * i + o, xl = (x + W - 1) % W;
                    const bool up = L.jdown[size_t(yu) * W + x], dn = L.jdown[size_t(y) * W + x];
                    const bool left = L.jright[size_t(y) * W + xl], right = L.jright[size_t(y) * W + x];
                    const bool ce = o ? left : right, si = o ? right : left;
                    const size_t w = size_t(y) * G.wpr + (i >> 5);
                    const uint32_t bit = 1u << (i & 31);
                    uint32_t *base = jneg.data() + size_t(c) * 4 * G.wpp;
                    if (!up) base[w] |= bit;
                    if (!dn) base[G.wpp + w] |= bit;
                    if (!ce) base[2 * size_t(G.wpp) + w] |= bit;
                    if (!si) base[3 * size_t(G.wpp) + w] |= bit;
                }
            }
        const uint32_t *d = nullptr;
        TRY(graph_upload(g, &d, jneg));
        g->d_jneg = const_cast<uint32_t *>(d);
    }
    if (field_signs) { // bit set where h_i < 0, each colour's compact layout
        std::vector<uint32_t> fneg(size_t(2) * G.wpp, 0u);
        for (uint32_t c = 0; c < 2; c++)
            for (uint32_t y = 0; y < G.H; y++) {
                const uint32_t o = (y + c) & 1;
                for (uint32_t i = 0; i < G.W / 2; i++)
                    if (biases[size_t(y) * G.W + 2 * i + o] < 0.0) fneg[size_t(c) * G.wpp + size_t(y) * G.wpr + (i >> 5)] |= 1u << (i & 31);
            }
        const uint32_t *d = nullptr;
        TRY(graph_upload(g, &d, fneg));
        g->d_fneg = const_cast<uint32_t *>(d);
    }
    return ISINGMC_OK;
}

static int build_general(isingmc_graph *g, const uint64_t *ea, const uint64_t *eb, const double *ej,
                         size_t n_edges, size_t nvars, const double *biases)
{
    g->kind = ISINGMC_KIND_GENERAL;
    const Adjacency A = build_adjacency(ea, eb, ej, n_edges, nvars);
    if (A.nbr.size() >= 0xFFFFFFFFull) return fail(ISINGMC_ERR_INVALID, "too many edges for the general path (2^32 directed)");
    const Colouring C = greedy_colouring(A, nvars);
    if (C.n_pos >= 0xFFFFFFC0ull) return fail(ISINGMC_ERR_INVALID, "too many sites for the general path");
    g->self_energy = A.self_energy;
    g->n_colours = C.n_colours;
    g->class_base = C.class_base;
    g->pos = C.pos;
    g->state_words = C.n_pos / 32;

    const uint32_t n_pos = uint32_t(C.n_pos);
    std::vector<uint32_t> site(n_pos, PAD_SITE), rowptr(size_t(n_pos) + 1, 0);
    for (size_t i = 0; i < nvars; i++) site[C.pos[i]] = uint32_t(i);
    for (uint32_t p = 0; p < n_pos; p++)
        rowptr[p + 1] = rowptr[p] + (site[p] == PAD_SITE ? 0u : uint32_t(A.ptr[site[p] + 1] - A.ptr[site[p]]));
    std::vector<uint32_t> nbr(A.nbr.size());
    std::vector<double> w(A.w.size());
    bool lossless = true;
    for (uint32_t p = 0; p < n_pos; p++) {
        if (site[p] == PAD_SITE) continue;
        uint32_t o = rowptr[p];
        for (uint64_t e = A.ptr[site[p]]; e < A.ptr[site[p] + 1]; e++, o++) {
            nbr[o] = uint32_t(C.pos[A.nbr[e]]);
            w[o] = A.w[e];
            lossless &= double(float(A.w[e])) == A.w[e];
        }
    }
    GenGraphDev &D = g->gdev;
    D.n_pos = n_pos;
    D.n_words = n_pos / 32;
    g->gen_edges2 = uint32_t(nbr.size());
    TRY(graph_upload(g, &D.rowptr, rowptr));
    TRY(graph_upload(g, &D.nbr, nbr));
    TRY(graph_upload(g, &D.site, site));
    {
        std::vector<uint32_t> cb(C.class_base.begin(), C.class_base.end());
        TRY(graph_upload(g, &D.class_base, cb));
        D.n_colours = C.n_colours;
    }
    g->w_is_float = lossless;
    if (lossless) { // stream 4-byte couplings when that loses nothing (e.g. J = +-1)
        std::vector<float> wf(w.begin(), w.end());
        const float *d = nullptr;
        TRY(graph_upload(g, &d, wf));
        D.w = d;
    } else {
        const double *d = nullptr;
        TRY(graph_upload(g, &d, w));
        D.w = d;
    }
    // replica-packed eligibility: one |J| for every bond, no fields, degree <= PK_MAX_DEG
    {
        uint64_t maxdeg = 0;
        for (size_t i = 0; i < nvars; i++) maxdeg = std::max(maxdeg, A.ptr[i + 1] - A.ptr[i]);
        bool uniform = !w.empty() && !g->has_bias && maxdeg <= PK_MAX_DEG && n_pos < 0x80000000u;
        const double jabs = w.empty() ? 0.0 : std::fabs(w[0]);
        for (double x : w) uniform &= std::fabs(x) == jabs;
        uniform &= jabs > 0.0;
        if (uniform) {
            std::vector<uint32_t> ell(size_t(PK_MAX_DEG) * n_pos, PK_NO_NBR); // slot-major: coalesced per slot
            for (uint32_t p = 0; p < n_pos; p++)
                for (uint32_t e = rowptr[p]; e < rowptr[p + 1]; e++)
                    ell[size_t(e - rowptr[p]) * n_pos + p] = nbr[e] | (w[e] > 0.0 ? 0x80000000u : 0u);
            // block headers: a slot whose 64 entries of a block are one translation (or all unused) needs no table read
            const size_t n_blocks = n_pos / 64;
            std::vector<uint2> hdr(n_blocks * PK_MAX_DEG);
            parallel_for(n_blocks, [&](size_t B) {
                for (uint32_t i = 0; i < uint32_t(PK_MAX_DEG); i++) {
                    const uint32_t *e = ell.data() + size_t(i) * n_pos + 64 * B;
                    const uint32_t p0 = uint32_t(64 * B);
                    bool unused = true, uniform = e[0] != PK_NO_NBR;
                    const uint32_t sign = e[0] & 0x80000000u, delta = (e[0] & 0x7FFFFFFFu) - p0;
                    for (uint32_t l = 0; l < 64; l++) {
                        unused &= e[l] == PK_NO_NBR;
                        uniform &= e[l] != PK_NO_NBR && (e[l] & 0x80000000u) == sign && (e[l] & 0x7FFFFFFFu) - (p0 + l) == delta;
                    }
                    hdr[B * PK_MAX_DEG + i] = unused ? make_uint2(PK_HDR_UNUSED, 0) : uniform ? make_uint2(PK_HDR_UNIFORM | sign, delta)
                                                                                           : make_uint2(PK_HDR_MIXED, 0);
                }
            });
            PkGraphDev &P = g->pk;
            TRY(graph_upload(g, &P.ell_hdr, hdr));
            TRY(graph_upload(g, &P.nbr_ell, ell));
            P.site = D.site;
            P.class_base = D.class_base;
            P.n_colours = D.n_colours;
            P.n_pos = n_pos;
            g->packed_ok = true;
            g->jabs = jabs;
            g->n_directed = nbr.size();
            // one degree, one sign?  (isolated sites have degree 0: they rule the uniform kernel out too)
            uint64_t mindeg = maxdeg;
            for (size_t i = 0; i < nvars; i++) mindeg = std::min(mindeg, A.ptr[i + 1] - A.ptr[i]);
            bool one_sign = true;
            for (double x : w) one_sign &= (x > 0.0) == (w[0] > 0.0);
            if (mindeg == maxdeg && maxdeg >= 3) {
                g->pk_uni_deg = int(maxdeg);
                g->pk_uni_pmj = !one_sign;
                g->pk_uni.negmask = w[0] > 0.0 ? 0u : 0xFFFFFFFFu;
                // this kernel's block headers: translations whatever the signs, the signs as one 64-bit mask per block and slot
                std::vector<uint2> shift(n_blocks * PK_MAX_DEG, make_uint2(PK_HDR_MIXED, 0)), sign(n_blocks * PK_MAX_DEG, make_uint2(0, 0));
                parallel_for(n_blocks, [&](size_t B) {
                    for (uint32_t i = 0; i < uint32_t(maxdeg); i++) {
                        const uint32_t *e = ell.data() + size_t(i) * n_pos + 64 * B;
                        const uint32_t p0 = uint32_t(64 * B);
                        const auto off = [&](uint32_t l) { return (e[l] & 0x7FFFFFFFu) - (p0 + l); };
                        // the translation of the block: what two of its first three lanes agree on
                        const uint32_t delta = off(1) == off(2) ? off(1) : off(0);
                        uint32_t odd_lanes = 0, odd_lane = 0;
                        uint64_t mask = 0;
                        for (uint32_t l = 0; l < 64; l++) {
                            if (e[l] == PK_NO_NBR || off(l) != delta) { odd_lanes++; odd_lane = l; }
                            mask |= uint64_t(e[l] != PK_NO_NBR && (e[l] >> 31)) << l;
                        }
                        if (odd_lanes == 0) shift[B * PK_MAX_DEG + i] = make_uint2(PK_HDR_UNIFORM, delta);
                        else if (odd_lanes == 1 && e[odd_lane] != PK_NO_NBR) { // a translation but for one lane (a row's wrap-around)
                            const int32_t ex = int32_t(off(odd_lane) - delta);
                            if (ex >= -(1 << 23) && ex < (1 << 23))
                                shift[B * PK_MAX_DEG + i] = make_uint2(PK_HDR_UNIFORM_BUT_ONE | (odd_lane << 2) | (uint32_t(ex) << 8), delta);
                        }
                        sign[B * PK_MAX_DEG + i] = make_uint2(uint32_t(mask), uint32_t(mask >> 32));
                    }
                });
                for (const uint2 &hd : shift) g->pk_uni_but_one += (hd.x & 3u) == PK_HDR_UNIFORM_BUT_ONE && hd.x != PK_HDR_UNIFORM;
                TRY(graph_upload(g, &g->pk_uni.shift, shift));
                TRY(graph_upload(g, &g->pk_uni.sign, sign));
                g->pk_class_full.resize(C.n_colours);
                for (uint32_t c = 0; c < C.n_colours; c++) { // real sites come first in a class, the padding after them
                    uint32_t real = 0;
                    while (uint32_t(C.class_base[c]) + real < uint32_t(C.class_base[c + 1]) && site[uint32_t(C.class_base[c]) + real] != PAD_SITE) real++;
                    g->pk_class_full[c] = uint32_t(C.class_base[c]) + real / 256 * 256;
                }
                g->pk_class_table.assign(C.n_colours, 0);
                for (uint32_t c = 0; c < C.n_colours; c++)
                    for (size_t B = C.class_base[c] / 64; B < g->pk_class_full[c] / 64 && !g->pk_class_table[c]; B++)
                        for (uint32_t i = 0; i < uint32_t(maxdeg); i++) g->pk_class_table[c] |= (shift[B * PK_MAX_DEG + i].x & 3u) == PK_HDR_MIXED;
            }
        }
    }
    g->class_real_end.resize(C.n_colours);
    for (uint32_t c = 0; c < C.n_colours; c++) { // real sites come first in a class
        uint32_t real = 0;
        while (uint32_t(C.class_base[c]) + real < uint32_t(C.class_base[c + 1]) && site[uint32_t(C.class_base[c]) + real] != PAD_SITE) real++;
        g->class_real_end[c] = uint32_t(C.class_base[c]) + real;
    }
    // real-coupling packed path: whatever the bit-sliced packed path cannot take (couplings of several sizes, site
    // biases), degree <= 15, quantisation faithful (rj_quantise)
    // (ISINGMC_FORCE_REAL=1 at graph creation builds it for graphs the bit-sliced path takes, too: the same Hamiltonian through
    // the other acceptance rule, for cross-checks such as tools/highstat.py)
    if ((!g->packed_ok || g->opt.force_real) && n_pos < 0x80000000u) {
        const RjQuant Q = rj_quantise(A, nvars, biases);
        if (Q.eligible) {
            const uint32_t slots = Q.max_degree <= 4 ? 4u : Q.max_degree <= 7 ? 7u : Q.max_degree <= 11 ? 11u : Q.max_degree <= 15 ? 15u
                                   : Q.max_degree <= 23 ? 23u : 31u;
            std::vector<uint32_t> enbr(size_t(slots) * n_pos);
            std::vector<int32_t> ejq(size_t(slots) * n_pos, 0), ehq(n_pos, 0);
            std::vector<int32_t> ejhi(size_t(slots) * n_pos, 0), ejlo(size_t(slots) * n_pos, 0), ehhi(n_pos, 0), ehlo(n_pos, 0);
            std::vector<uint8_t> edsh(n_pos, 0);
            for (uint32_t i = 0; i < slots; i++)
                for (uint32_t p = 0; p < n_pos; p++) enbr[size_t(i) * n_pos + p] = p; // unused slots point at the own position
            for (uint32_t p = 0; p < n_pos; p++) {
                if (site[p] == PAD_SITE) continue;
                ehq[p] = Q.hq[site[p]];
                ehhi[p] = Q.hhi[site[p]];
                ehlo[p] = Q.hlo[site[p]];
                edsh[p] = Q.dshift[site[p]];
                g->rj_heavy_sites += Q.dshift[site[p]] != 0;
                uint32_t i = 0;
                for (uint64_t e = A.ptr[site[p]]; e < A.ptr[site[p] + 1]; e++, i++) {
                    enbr[size_t(i) * n_pos + p] = uint32_t(C.pos[A.nbr[e]]);
                    ejq[size_t(i) * n_pos + p] = Q.jq[e];
                    ejhi[size_t(i) * n_pos + p] = Q.jhi[e];
                    ejlo[size_t(i) * n_pos + p] = Q.jlo[e];
                }
            }
            uint32_t lt[RJ_LOG_INTERVALS + 1];
            rj_log_table(lt);
            std::vector<uint2> logtab(RJ_LOG_INTERVALS);
            for (int i = 0; i < RJ_LOG_INTERVALS; i++) logtab[i] = make_uint2(lt[i], lt[i + 1] - lt[i]);
            RjGraphDev &J = g->rj;
            TRY(graph_upload(g, &J.nbr, enbr));
            TRY(graph_upload(g, &J.jq, ejq));
            TRY(graph_upload(g, &J.hq, ehq));
            TRY(graph_upload(g, &J.logtab, logtab));
            J.dshift = nullptr;
            if (Q.heavy) TRY(graph_upload(g, &J.dshift, edsh));
            J.n_pos = n_pos;
            J.slots = slots;
            g->rj_hi = g->rj_lo = J;
            TRY(graph_upload(g, &g->rj_hi.jq, ejhi));
            TRY(graph_upload(g, &g->rj_hi.hq, ehhi));
            TRY(graph_upload(g, &g->rj_lo.jq, ejlo));
            TRY(graph_upload(g, &g->rj_lo.hq, ehlo));
            g->rj_k = Q.k;
            g->rj_k_energy = Q.k_energy;
            g->rj_ok = true;
            // the packed containers' common parts (random start, set_state, copy-out) read these
            g->pk.site = D.site;
            g->pk.class_base = D.class_base;
            g->pk.n_colours = D.n_colours;
            g->pk.n_pos = n_pos;
        }
    }
    D.bias = nullptr;
    if (g->has_bias) {
        std::vector<double> bias(n_pos, 0.0);
        for (size_t i = 0; i < nvars; i++) bias[C.pos[i]] = biases[i];
        TRY(graph_upload(g, &D.bias, bias));
    }
    return ISINGMC_OK;
}

extern "C" int isingmc_graph_create(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges,
                                    size_t nvars, const double *biases, int device, unsigned flags,
                                    isingmc_graph **graph_out)
{
    if (!graph_out) return fail(ISINGMC_ERR_INVALID, "graph_out is NULL");
    *graph_out = nullptr;
    TRY(check_edges(ea, eb, ej, n_edges, nvars));
    bool has_bias = false;
    if (biases)
        for (size_t i = 0; i < nvars; i++) {
            if (!std::isfinite(biases[i])) return fail(ISINGMC_ERR_INVALID, "biases must be finite");
            has_bias |= biases[i] != 0.0;
        }
    TRY(use_device(device));
    auto g = std::make_unique<isingmc_graph>();
    g->device = device;
    g->nvars = nvars;
    g->n_edges = n_edges;
    g->has_bias = has_bias;
    g->opt = Options::from_env();
    g->stable_path = (flags & ISINGMC_FLAG_STABLE_PATH) != 0 || env_flag("ISINGMC_STABLE_PATH");
    Lattice2D L;
    bool field_signs = false;
    const double h = has_bias ? uniform_bias(biases, nvars, &field_signs) : 0.0;
    if (!(flags & ISINGMC_FLAG_FORCE_GENERAL) && !std::isnan(h)) L = recognise_lattice2d(ea, eb, ej, n_edges, nvars);
    if (lattice_fast_path_ok(L, h, field_signs)) TRY(build_lattice(g.get(), L, h, biases, field_signs));
    else TRY(build_general(g.get(), ea, eb, ej, n_edges, nvars, biases));
    *graph_out = g.release();
    return ISINGMC_OK;
}

extern "C" int isingmc_graph_info(const isingmc_graph *g, isingmc_graph_info_t *info)
{
    if (!g || !info) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    std::memset(info, 0, sizeof *info);
    info->kind = g->kind;
    info->device = g->device;
    info->nvars = g->nvars;
    info->n_edges = g->n_edges;
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        info->width = int32_t(g->geom.W);
        info->height = int32_t(g->geom.H);
        info->jabs = g->jabs;
        info->jabs_y = g->mc_mode == MC_ANISO ? g->jabs_y : g->jabs;
        info->uniform_sign = g->uniform_sign;
        info->fast_path = g->mc_mode;
        info->field = g->field;
        info->open_x = int32_t(g->open.open_x);
        info->open_y = int32_t(g->open.open_y);
        info->field_signs = g->d_fneg ? 1 : 0;
    }
    info->n_colours = g->n_colours;
    info->packed_degree = g->packed_ok ? g->pk_uni_deg : 0;
    info->packed_but_one_headers = g->packed_ok ? int32_t(g->pk_uni_but_one) : 0;
    info->real_slots = g->rj_ok ? int32_t(g->rj.slots) : 0;
    info->real_quantum_log2 = g->rj_ok ? g->rj_k : 0;
    info->real_energy_log2 = g->rj_ok ? g->rj_k_energy : 0;
    info->real_heavy_sites = g->rj_ok ? int32_t(g->rj_heavy_sites) : 0;
    info->stable_path = g->stable_path ? 1 : 0;
    info->state_words = g->state_words;
    return ISINGMC_OK;
}

extern "C" void isingmc_graph_destroy(isingmc_graph *g) { delete g; }

