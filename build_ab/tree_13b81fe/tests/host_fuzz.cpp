// Randomised driver for the host-side logic of libisingmc.so (csrc/host_logic.cpp), meant to be built with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// and run by tests/test_host_sanitizers.py (GPU AddressSanitizer is not available on the pool: the sanitizers run on
// the CPU build only).  Every case also checks the invariants the device code relies on: a recognised lattice accounts
// for every edge, the adjacency is symmetric and in range, the colouring is proper and its classes are whole
// 256-position blocks, quantised couplings fit 31 bits per site, schedules are finite.
// usage: host_fuzz [cases] [seed]
#include "host_logic.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

using namespace isingmc;

namespace {

struct Rng {
    uint64_t s;
    uint64_t next()
    {
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    double unit() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    bool coin(double p = 0.5) { return unit() < p; }
};

struct Edges {
    std::vector<uint64_t> a, b;
    std::vector<double> j;
    size_t nvars = 0;
    void add(uint64_t x, uint64_t y, double w)
    {
        a.push_back(x);
        b.push_back(y);
        j.push_back(w);
    }
    void finish()
    {
        nvars = 0;
        for (size_t k = 0; k < a.size(); k++) nvars = std::max<size_t>(nvars, std::max(a[k], b[k]) + 1); // lattice.rs:51-55
    }
};

[[noreturn]] void die(const char *what, uint64_t case_no)
{
    std::fprintf(stderr, "host_fuzz: invariant broken in case %llu: %s\n", (unsigned long long)case_no, what);
    std::exit(1);
}
#define CHECK(cond) do { if (!(cond)) die(#cond, case_no); } while (0)

double draw_coupling(Rng &r, int style)
{
    switch (style) {
    case 0: return -1.0;
    case 1: return r.coin() ? 1.0 : -1.0;
    case 2: return (r.coin() ? 1.0 : -1.0) * (0.25 + r.unit());
    case 3: return std::ldexp(r.unit() - 0.5, int(r.below(80)) - 40); // wide dynamic range
    default: return r.coin(0.1) ? 0.0 : double(int64_t(r.below(7)) - 3);
    }
}

// a W x H lattice (periodic or open per direction), then damaged in one of several ways
Edges lattice_case(Rng &r, bool *intact)
{
    Edges e;
    const uint64_t W = 2 * (2 + r.below(7)), H = 2 * (2 + r.below(7));
    const bool open_x = r.coin(0.25), open_y = r.coin(0.25);
    const int style = int(r.below(3));
    const double jx = draw_coupling(r, style == 2 ? 2 : 0), jy = r.coin(0.3) ? 2.0 * jx : jx;
    for (uint64_t y = 0; y < H; y++)
        for (uint64_t x = 0; x < W; x++) {
            const uint64_t i = y * W + x;
            if (x + 1 < W || !open_x) {
                const double w = style == 1 ? (r.coin() ? jx : -jx) : jx;
                if (r.coin()) e.add(i, y * W + (x + 1) % W, w);
                else e.add(y * W + (x + 1) % W, i, w);
            }
            if (y + 1 < H || !open_y) {
                const double w = style == 1 ? (r.coin() ? jy : -jy) : jy;
                e.add(i, ((y + 1) % H) * W + x, w);
            }
        }
    *intact = true;
    const int damage = int(r.below(8));
    if (damage == 1 && !e.a.empty()) { // one bond missing
        const size_t k = r.below(e.a.size());
        e.a.erase(e.a.begin() + k); e.b.erase(e.b.begin() + k); e.j.erase(e.j.begin() + k);
        *intact = false;
    } else if (damage == 2) { // one bond twice
        const size_t k = r.below(e.a.size());
        e.add(e.a[k], e.b[k], e.j[k]);
        *intact = false;
    } else if (damage == 3) { // a self loop
        const uint64_t i = r.below(W * H);
        e.add(i, i, 0.5);
        *intact = false;
    } else if (damage == 4) { // one coupling of another size
        e.j[r.below(e.j.size())] *= 3.0;
        *intact = false;
    } else if (damage == 5) { // a long-range bond
        e.add(0, W * H / 2 + 1, -1.0);
        *intact = false;
    } else if (damage == 6) { // an extra isolated tail of sites
        e.add(W * H - 1, W * H + r.below(3), -1.0);
        *intact = false;
    }
    if (r.coin(0.3)) { // edge order must not matter
        for (size_t k = e.a.size(); k > 1; k--) {
            const size_t m = r.below(k);
            std::swap(e.a[k - 1], e.a[m]); std::swap(e.b[k - 1], e.b[m]); std::swap(e.j[k - 1], e.j[m]);
        }
    }
    e.finish();
    return e;
}

Edges graph_case(Rng &r)
{
    Edges e;
    const int kind = int(r.below(5));
    const int style = int(r.below(5));
    if (kind == 0) { // sparse random graph, multi-edges and self loops allowed
        const uint64_t n = 1 + r.below(300);
        const uint64_t m = 1 + r.below(3 * n + 1);
        for (uint64_t k = 0; k < m; k++) e.add(r.below(n), r.below(n), draw_coupling(r, style));
    } else if (kind == 1) { // star: one site of huge degree
        const uint64_t n = 2 + r.below(600);
        for (uint64_t k = 1; k < n; k++) e.add(0, k, draw_coupling(r, style));
    } else if (kind == 2) { // complete graph
        const uint64_t n = 2 + r.below(24);
        for (uint64_t x = 0; x < n; x++)
            for (uint64_t y = x + 1; y < n; y++) e.add(x, y, draw_coupling(r, style));
    } else if (kind == 3) { // ring with a gap in the numbering (sites without bonds)
        const uint64_t n = 3 + r.below(500), stride = 1 + r.below(3);
        for (uint64_t k = 0; k < n; k++) e.add(k * stride, ((k + 1) % n) * stride, draw_coupling(r, style));
    } else { // one edge, possibly a lone self loop
        const uint64_t x = r.below(40);
        e.add(x, r.coin() ? x : r.below(40), draw_coupling(r, style));
    }
    e.finish();
    return e;
}

void check_case(Rng &r, const Edges &e, bool expect_lattice, bool known_lattice_shape, uint64_t case_no)
{
    const size_t n = e.nvars, m = e.a.size();
    const Lattice2D L = recognise_lattice2d(e.a.data(), e.b.data(), e.j.data(), m, n);
    if (known_lattice_shape && expect_lattice) CHECK(L.ok);
    if (known_lattice_shape && !expect_lattice) CHECK(!L.ok);
    if (L.ok) {
        CHECK(uint64_t(L.W) * uint64_t(L.H) == n);
        CHECK(L.W >= 4 && L.H >= 4 && L.W % 2 == 0 && L.H % 2 == 0);
        const uint64_t bonds = uint64_t(L.W - (L.open_x ? 1 : 0)) * L.H + uint64_t(L.H - (L.open_y ? 1 : 0)) * L.W;
        CHECK(bonds == m); // every edge is a lattice bond, every bond once
        CHECK(L.jabs >= 0.0 && L.jabs_y >= 0.0);
        if (!L.uniform_sign) CHECK(L.jright.size() == n && L.jdown.size() == n);
        for (size_t k = 0; k < m; k++) CHECK(e.a[k] != e.b[k]);
    }

    const Adjacency A = build_adjacency(e.a.data(), e.b.data(), e.j.data(), m, n);
    CHECK(A.ptr.size() == n + 1 && A.ptr[0] == 0);
    size_t self_loops = 0;
    double self = 0.0;
    for (size_t k = 0; k < m; k++)
        if (e.a[k] == e.b[k]) { self_loops++; self += e.j[k]; }
    CHECK(A.ptr[n] == 2 * (m - self_loops));
    CHECK(A.self_energy == self);
    CHECK(A.nbr.size() == A.ptr[n] && A.w.size() == A.ptr[n]);
    double sum_w = 0.0, sum_e = 0.0;
    for (size_t i = 0; i < n; i++) {
        CHECK(A.ptr[i] <= A.ptr[i + 1]);
        for (uint64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) {
            CHECK(A.nbr[q] < n && A.nbr[q] != i);
            sum_w += std::fabs(A.w[q]);
        }
    }
    for (size_t k = 0; k < m; k++)
        if (e.a[k] != e.b[k]) sum_e += 2.0 * std::fabs(e.j[k]);
    CHECK(std::fabs(sum_w - sum_e) <= 1e-9 * (1.0 + sum_e));

    const Colouring C = greedy_colouring(A, n);
    CHECK(C.colour.size() == n && C.pos.size() == n && C.class_base.size() == size_t(C.n_colours) + 1);
    CHECK(C.n_colours >= 1 && C.n_pos == C.class_base[C.n_colours] && C.n_pos % 256 == 0);
    std::vector<uint8_t> taken(C.n_pos, 0);
    for (size_t i = 0; i < n; i++) {
        const uint32_t c = C.colour[i];
        CHECK(c < C.n_colours);
        CHECK(C.pos[i] >= C.class_base[c] && C.pos[i] < C.class_base[c + 1]);
        CHECK(!taken[C.pos[i]]);
        taken[C.pos[i]] = 1;
        for (uint64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) CHECK(C.colour[A.nbr[q]] != c); // independent sets
        // greedy: every smaller colour is taken by a lower-numbered neighbour
        for (uint32_t lower = 0; lower < c; lower++) {
            bool found = false;
            for (uint64_t q = A.ptr[i]; q < A.ptr[i + 1] && !found; q++) found = A.nbr[q] < i && C.colour[A.nbr[q]] == lower;
            CHECK(found);
        }
    }
    for (uint32_t c = 0; c < C.n_colours; c++) CHECK((C.class_base[c + 1] - C.class_base[c]) % 256 == 0 && C.class_base[c + 1] > C.class_base[c]);

    // biases: none, uniform, per site, one enormous entry
    std::vector<double> bias;
    const int bias_kind = int(r.below(4));
    if (bias_kind == 1) bias.assign(n, 0.5);
    if (bias_kind == 2) { bias.resize(n); for (auto &h : bias) h = r.unit() - 0.5; }
    if (bias_kind == 3) { bias.assign(n, 0.0); bias[r.below(n)] = 1e9; }
    const RjQuant Q = rj_quantise(A, n, bias.empty() ? nullptr : bias.data());
    CHECK(Q.jq.size() == A.w.size() && Q.hq.size() == n);
    uint32_t maxdeg = 0;
    double fmax = 0.0;
    for (size_t i = 0; i < n; i++) {
        int64_t total = std::llabs(int64_t(Q.hq[i]));
        double f = bias.empty() ? 0.0 : std::fabs(bias[i]);
        for (uint64_t q = A.ptr[i]; q < A.ptr[i + 1]; q++) {
            total += std::llabs(int64_t(Q.jq[q]));
            f += std::fabs(A.w[q]);
        }
        // |X| <= sum of the rounded magnitudes: each rounds by at most 1/2, Fmax 2^-k < 2^30
        CHECK(total <= (int64_t(1) << 30) + int64_t(A.ptr[i + 1] - A.ptr[i] + 1) / 2 + 1);
        maxdeg = std::max<uint32_t>(maxdeg, uint32_t(A.ptr[i + 1] - A.ptr[i]));
        fmax = std::max(fmax, f);
    }
    CHECK(Q.max_degree == maxdeg);
    if (Q.eligible) {
        CHECK(maxdeg <= 15 && fmax > 0.0);
        CHECK(std::ldexp(fmax, -Q.k) >= double(1 << 29) && std::ldexp(fmax, -Q.k) < double(int64_t(1) << 30));
        for (size_t q = 0; q < A.w.size(); q++) // the same integer on both ends of a bond, rounding error <= 1/2
            CHECK(std::fabs(std::ldexp(A.w[q], -Q.k) - double(Q.jq[q])) <= 0.5);
    }
    if (bias_kind == 3 && sum_e > 0.0 && sum_e < 1e6) CHECK(!Q.eligible); // the enormous bias must not set the quantum

    // acceptance scales of arbitrary betas
    const double betas[] = {0.0, -1.0, 1e-300, 1e-12, 0.4407, 3.0, 1e12, 1e300, HUGE_VAL, r.unit() * 10.0};
    for (double beta : betas) {
        uint32_t shift = 77, mant = 0;
        rj_beta(beta, Q.k, &shift, &mant);
        CHECK(shift <= 31);
        if (!(beta > 0.0)) CHECK(shift == 31 && mant == 0xFFFFFFFFu);
    }
}

void check_schedules(Rng &r, uint64_t case_no)
{
    const size_t T = r.below(200);
    const size_t stops = r.below(6);
    std::vector<uint64_t> t(stops);
    std::vector<double> v(stops);
    for (size_t k = 0; k < stops; k++) {
        t[k] = r.coin(0.1) ? T + r.below(50) : r.below(T + 1);
        v[k] = r.coin(0.05) ? NAN : r.unit() * 4.0;
    }
    bool finite = true;
    for (double x : v) finite &= std::isfinite(x);
    for (int compat = 0; compat < 2; compat++) {
        std::vector<double> out(T + 1, -7.0);
        const std::string err = expand_schedule(t.data(), v.data(), stops, T, compat != 0, out.data());
        CHECK(err.empty() == finite);
        CHECK(out[T] == -7.0); // writes exactly T entries
        if (!finite) continue;
        double lo = 1.0, hi = 1.0;
        if (stops) { lo = *std::min_element(v.begin(), v.end()); hi = *std::max_element(v.begin(), v.end()); }
        for (size_t s = 0; s < T; s++) CHECK(std::isfinite(out[s]) && (compat || (out[s] >= lo - 1e-12 && out[s] <= hi + 1e-12)));
    }

    const size_t n = r.below(70);
    const auto seeds = make_seeds(true, r.next(), n);
    CHECK(seeds.size() == n);
    const auto again = make_seeds(false, 0, 3);
    CHECK(again.size() == 3);

    // tempering swap round: perm stays a permutation, swaps are counted
    const size_t rungs = r.below(40);
    std::vector<double> b(rungs), en(rungs);
    std::vector<uint32_t> perm(rungs);
    for (size_t i = 0; i < rungs; i++) {
        b[i] = 0.1 + 0.05 * double(i);
        en[i] = r.coin(0.2) ? -100.0 : -100.0 - 50.0 * r.unit() * double(i);
        perm[i] = uint32_t(i);
    }
    uint64_t total = 0;
    for (uint64_t round = 0; round < 6; round++) {
        const std::vector<uint32_t> before = perm;
        const uint64_t swaps = pt_swap_round(r.next(), round + (r.coin(0.2) ? (uint64_t(1) << 33) : 0), rungs, b.data(), en.data(), perm.data());
        uint64_t moved = 0;
        for (size_t i = 0; i < rungs; i++) moved += before[i] != perm[i];
        CHECK(moved == 2 * swaps);
        total += swaps;
    }
    std::set<uint32_t> distinct(perm.begin(), perm.end());
    CHECK(distinct.size() == rungs);
    (void)total;

    // bit planes -> bytes: every byte 0 or 1, and the round trip through the S2 layout
    const uint32_t W = 64 * uint32_t(1 + r.below(3)), H = 2 * uint32_t(1 + r.below(5));
    const uint32_t wpr = W / 64;
    std::vector<uint32_t> words(2 * size_t(H) * wpr);
    for (auto &w : words) w = uint32_t(r.next());
    const size_t offset = r.below(3) * 16 + (r.coin(0.3) ? 1 : 0); // aligned and unaligned destinations
    std::vector<uint8_t> bytes(size_t(W) * H + 64 + offset, 0xEE);
    unpack_lattice(W, H, words.data(), bytes.data() + offset);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            const uint32_t c = (x + y) & 1, i = x >> 1; // x = 2i + ((y+c)&1)
            const uint32_t bit = (words[size_t(c) * H * wpr + size_t(y) * wpr + (i >> 5)] >> (i & 31)) & 1;
            CHECK(bytes[offset + size_t(y) * W + x] == bit);
        }
    for (size_t k = 0; k < offset; k++) CHECK(bytes[k] == 0xEE);
    for (size_t k = offset + size_t(W) * H; k < bytes.size(); k++) CHECK(bytes[k] == 0xEE);
}

} // namespace

int main(int argc, char **argv)
{
    const uint64_t cases = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 2000;
    Rng r{argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1};
    uint64_t lattices = 0, recognised_checks = 0;
    for (uint64_t case_no = 0; case_no < cases; case_no++) {
        if (case_no % 3 == 0) {
            bool intact = false;
            const Edges e = lattice_case(r, &intact);
            lattices++;
            recognised_checks += intact;
            check_case(r, e, intact, true, case_no);
        } else {
            check_case(r, graph_case(r), false, false, case_no);
        }
        check_schedules(r, case_no);
    }
    uint32_t table[2049];
    rj_log_table(table);
    for (int i = 0; i < 2048; i++)
        if (table[i] >= table[i + 1]) die("log table is increasing", 0);
    if (table[0] != 0 || table[2048] > (1u << 24) + 1 || table[2048] < (1u << 24) - 1) die("log table spans [0, 2^24]", 0);
    std::printf("host_fuzz: %llu cases (%llu lattices, %llu intact) ok\n", (unsigned long long)cases, (unsigned long long)lattices,
                (unsigned long long)recognised_checks);
    return 0;
}
