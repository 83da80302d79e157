// Host-side logic (no device code).  Reference lines per function: see host_logic.hpp / isingmc.h.
#include "host_logic.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <random>
#include <unordered_map>

namespace isingmc {

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

SmallRng::SmallRng(uint64_t state)
{
    for (auto &word : s) { // SplitMix64 fills the 256-bit state
        state += 0x9e3779b97f4a7c15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        word = z ^ (z >> 31);
    }
}

uint64_t SmallRng::next_u64()
{
    const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}

std::vector<uint64_t> make_seeds(bool has_seed, uint64_t seed_gen, size_t n)
{
    if (!has_seed) { // SmallRng::from_entropy()
        std::random_device rd;
        seed_gen = (uint64_t(rd()) << 32) ^ uint64_t(rd());
    }
    SmallRng rng(seed_gen);
    std::vector<uint64_t> out(n);
    for (auto &v : out) v = rng.next_u64();
    return out;
}

std::string expand_schedule(const uint64_t *stop_t, const double *stop_beta, size_t n_stops,
                            size_t timesteps, bool compat_constant_beta, double *betas_out)
{
    std::vector<std::pair<uint64_t, double>> betas(n_stops);
    for (size_t k = 0; k < n_stops; k++) {
        if (!std::isfinite(stop_beta[k])) return "beta schedule values must be finite";
        betas[k] = {stop_t[k], stop_beta[k]};
    }
    std::stable_sort(betas.begin(), betas.end(),
                     [](const auto &a, const auto &b) { return a.first < b.first; });
    if (betas.empty()) {
        betas.push_back({0, 1.0});
        betas.push_back({timesteps, 1.0});
    }
    if (betas.front().first > 0) betas.insert(betas.begin(), {0, betas.front().second});
    const uint64_t last_user_t = betas.back().first;
    if (betas.back().first < timesteps) betas.push_back({timesteps, betas.back().second});

    size_t idx = 0;
    for (size_t step = 0; step < timesteps; step++) {
        // the reference evaluates the interpolation at a captured constant (the last stop's time)
        const uint64_t i = compat_constant_beta ? last_user_t : step;
        while (idx + 2 < betas.size() && i > betas[idx + 1].first) idx++;
        const auto [ia, va] = betas[idx];
        const auto [ib, vb] = betas[idx + 1];
        const double frac = (ib == ia) ? 0.0 : double(int64_t(i) - int64_t(ia)) / double(ib - ia);
        betas_out[step] = (vb - va) * frac + va;
    }
    return "";
}

Lattice2D recognise_lattice2d(const uint64_t *ea, const uint64_t *eb, const double *ej,
                              size_t n_edges, size_t nvars)
{
    Lattice2D out;
    const uint64_t N = nvars;
    if (N < 16 || n_edges != 2 * N || N > (uint64_t(1) << 40)) return out;
    // |a-b| is 1 or W-1 for horizontal bonds, W or N-W for vertical ones; N-W of the 2N bonds
    // have |a-b| == W, which makes W the most frequent difference other than 1.
    std::unordered_map<uint64_t, uint64_t> hist;
    for (size_t k = 0; k < n_edges; k++) {
        const uint64_t d = ea[k] > eb[k] ? ea[k] - eb[k] : eb[k] - ea[k];
        if (d != 1 && ++hist[d] && hist.size() > 64) return out; // a lattice has <= 3 such values
    }
    uint64_t W = 0, best = 0;
    for (const auto &[d, cnt] : hist)
        if (cnt > best || (cnt == best && d < W)) { best = cnt; W = d; }
    if (W < 4 || N % W != 0) return out;
    const uint64_t H = N / W;
    if (H < 4 || (H & 1) || (W & 1) || W > (1u << 30) || H > (1u << 30)) return out;

    std::vector<uint8_t> seen(2 * N, 0), jpos(2 * N, 0);
    const double jabs = std::fabs(ej[0]);
    bool any_pos = false, any_neg = false;
    for (size_t k = 0; k < n_edges; k++) {
        const uint64_t lo = std::min(ea[k], eb[k]), hi = std::max(ea[k], eb[k]);
        if (hi >= N) return out;
        const uint64_t d = hi - lo;
        uint64_t slot;
        if (d == 1 && lo % W != W - 1) slot = 2 * lo;               // right bond of lo
        else if (d == W - 1 && lo % W == 0) slot = 2 * hi;          // right bond of hi wraps to lo
        else if (d == W) slot = 2 * lo + 1;                         // down bond of lo
        else if (d == N - W && lo < W) slot = 2 * hi + 1;           // down bond of hi wraps to lo
        else return out;
        if (seen[slot]) return out;
        seen[slot] = 1;
        if (!(std::fabs(ej[k]) == jabs)) return out; // uniform |J| only (NaN fails too)
        const bool pos = ej[k] > 0.0;
        jpos[slot] = pos;
        (pos ? any_pos : any_neg) = true;
    }
    out.ok = true;
    out.W = int(W);
    out.H = int(H);
    out.jabs = jabs;
    out.uniform_sign = !(any_pos && any_neg);
    out.jpos_uniform = any_pos && !any_neg;
    if (!out.uniform_sign) {
        out.jright.resize(N);
        out.jdown.resize(N);
        for (uint64_t i = 0; i < N; i++) {
            out.jright[i] = jpos[2 * i];
            out.jdown[i] = jpos[2 * i + 1];
        }
    }
    return out;
}

Adjacency build_adjacency(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges,
                          size_t nvars)
{
    Adjacency A;
    A.ptr.assign(nvars + 1, 0);
    for (size_t k = 0; k < n_edges; k++) {
        if (ea[k] == eb[k]) { A.self_energy += ej[k]; continue; }
        A.ptr[ea[k] + 1]++;
        A.ptr[eb[k] + 1]++;
    }
    for (size_t i = 0; i < nvars; i++) A.ptr[i + 1] += A.ptr[i];
    A.nbr.resize(A.ptr[nvars]);
    A.w.resize(A.ptr[nvars]);
    std::vector<uint64_t> fill(A.ptr.begin(), A.ptr.end() - 1);
    for (size_t k = 0; k < n_edges; k++) { // neighbours of a site stay in edge-list order
        if (ea[k] == eb[k]) continue;
        A.nbr[fill[ea[k]]] = uint32_t(eb[k]);
        A.w[fill[ea[k]]++] = ej[k];
        A.nbr[fill[eb[k]]] = uint32_t(ea[k]);
        A.w[fill[eb[k]]++] = ej[k];
    }
    return A;
}

Colouring greedy_colouring(const Adjacency &A, size_t nvars)
{
    Colouring C;
    C.colour.resize(nvars);
    uint64_t maxdeg = 0;
    for (size_t i = 0; i < nvars; i++) maxdeg = std::max(maxdeg, A.ptr[i + 1] - A.ptr[i]);
    std::vector<uint8_t> used(maxdeg + 2);
    uint32_t nc = 1;
    for (size_t i = 0; i < nvars; i++) {
        const uint64_t deg = A.ptr[i + 1] - A.ptr[i];
        std::fill(used.begin(), used.begin() + deg + 2, 0);
        for (uint64_t e = A.ptr[i]; e < A.ptr[i + 1]; e++) {
            const uint32_t j = A.nbr[e];
            if (j < i && C.colour[j] <= deg) used[C.colour[j]] = 1;
        }
        uint32_t c = 0;
        while (used[c]) c++;
        C.colour[i] = c;
        nc = std::max(nc, c + 1);
    }
    C.n_colours = nc;
    std::vector<uint64_t> count(nc, 0);
    for (size_t i = 0; i < nvars; i++) count[C.colour[i]]++;
    C.class_base.assign(nc + 1, 0);
    for (uint32_t c = 0; c < nc; c++) C.class_base[c + 1] = C.class_base[c] + (count[c] + 63) / 64 * 64;
    C.n_pos = C.class_base[nc];
    C.pos.resize(nvars);
    std::fill(count.begin(), count.end(), 0);
    for (size_t i = 0; i < nvars; i++) C.pos[i] = C.class_base[C.colour[i]] + count[C.colour[i]]++;
    return C;
}

} // namespace isingmc
