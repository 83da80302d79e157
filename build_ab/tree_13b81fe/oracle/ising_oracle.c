/*
 * ising_oracle.c -- CPU ORACLE for the classical Ising Metropolis hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The product path (pyisingmontecarlo_amd/, libisingmc.so) never links, imports
 * or falls back to it.
 *
 * PARITY UNPINNED against the reference's arithmetic: the reference
 * (/root/reference/src/lattice.rs:171-470, src/classicising.rs:62-179) delegates every
 * Metropolis operation to the third-party crate `qmc ^2.20` (Cargo.toml:23-25), whose source is
 * not in the tree, and ships no tests or golden vectors; no Rust toolchain exists here.  What
 * pins this oracle instead: Random123's published Philox4x32-10 known-answer vectors, the
 * xoshiro256++ reference vector, the README's Hamiltonian (README.md:45-46, energy = J*Sza*Szb,
 * positive J antiferromagnetic) on hand-checkable states, exact enumeration on small graphs and
 * Kaufman's exact finite-torus solution (oracle/exact.py, tests/).
 *
 * Three engines live here:
 *
 *  A. orc_ref_*   "reference-faithful" restatement of what lattice.rs:192-212 drives: one
 *                 replica = one sequential chain; adjacency list; one bool per spin; f64 dE;
 *                 a timestep = nvars single-spin Metropolis attempts at uniformly random sites,
 *                 accept if dE <= 0 else with probability exp(-beta dE); per-replica
 *                 xoshiro256++ (rand 0.8 SmallRng on 64-bit targets) seeded from the u64 that
 *                 make_seeds (lattice.rs:83-91) hands out.  The crate-internal details
 *                 (order of rng draws, edge/worm moves) are [UNVERIFIED] recollection; this
 *                 engine is the timed CPU baseline and a statistical cross-check, never a
 *                 bit-level authority.
 *
 *  B. orc_lat_*   serial, per-spin restatement of the build's own 2-colour checkerboard
 *                 algorithm on a periodic W x H square lattice with uniform |J| (DESIGN.md
 *                 "Algorithm specification", S3).  Same Philox counters, same fixed-point
 *                 acceptance thresholds => the HIP kernels must reproduce its spin
 *                 configurations BIT FOR BIT.  Written spin-by-spin (no bit-slicing) on purpose,
 *                 so that it checks the kernel's bit-sliced logic independently.
 *
 *  C. orc_gen_*   serial restatement of the general edge-list path (greedy colouring,
 *                 f64 local fields, deterministic exp), DESIGN.md S4.  Also bit-exact.
 *
 * Hamiltonian (all engines):  E = sum_edges J_ab s_a s_b  -  sum_i h_i s_i,  s = +1 for True.
 * The edge term follows README.md:45-46; the sign of the bias term is [UNVERIFIED]
 * (crate-internal), chosen as -h s.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DOM_LAT_SWEEP 0x4C415453u /* "LATS" */
#define DOM_LAT_INIT 0x4C415449u  /* "LATI" */
#define DOM_GEN_SWEEP 0x47454E53u /* "GENS" */
#define DOM_GEN_INIT 0x47454E49u  /* "GENI" */
#define DOM_PT_SWAP 0x50545357u   /* "PTSW" */

#ifndef N_PLANES
#define N_PLANES 7 /* bit-planes of the acceptance uniform drawn before the residual stage */
#endif
#define THR_BITS (N_PLANES + 32) /* fixed-point bits of an acceptance probability */

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123).  Pinned by the Random123
 * known-answer vectors in tests/golden/philox_kat.json.
 * ---------------------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox_seeded(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t out[4])
{
    uint32_t ctr[4] = {c0, c1, c2, c3};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

/* counter word 2: (t >> 32) in the top 16 bits, colour in bits 8..15, call index in bits 0..7 */
static uint32_t ctr2(uint64_t t, uint32_t colour, uint32_t call)
{
    return (uint32_t)(((t >> 32) & 0xFFFFu) << 16) | ((colour & 0xFFu) << 8) | (call & 0xFFu);
}

/* ------------------------------------------------------------------------------------------
 * xoshiro256++ with SplitMix64 seeding = rand 0.8 `SmallRng::seed_from_u64` on 64-bit targets
 * (published algorithm; the rand crate itself is not in the tree => [UNVERIFIED] that the
 * reference's Cargo resolution picks exactly this generator).  Used by make_seeds
 * (lattice.rs:83-91) and by the reference-faithful engine (lattice.rs:198).
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t s[4]; } xoshiro;

static uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static uint64_t xo_next(xoshiro *g)
{
    uint64_t *s = g->s;
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

static void xo_seed_from_u64(xoshiro *g, uint64_t state)
{
    for (int i = 0; i < 4; i++) {
        state += 0x9e3779b97f4a7c15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        g->s[i] = z ^ (z >> 31);
    }
}

/* raw generator from an explicit state: for the xoshiro256++ reference vector */
void orc_xoshiro_from_state(const uint64_t s[4], size_t n, uint64_t *out)
{
    xoshiro g;
    memcpy(g.s, s, sizeof g.s);
    for (size_t i = 0; i < n; i++) out[i] = xo_next(&g);
}

/* lattice.rs:83-91: master rng seeded from seed_gen, one u64 per experiment */
void orc_make_seeds(uint64_t seed_gen, size_t n, uint64_t *out)
{
    xoshiro g;
    xo_seed_from_u64(&g, seed_gen);
    for (size_t i = 0; i < n; i++) out[i] = xo_next(&g);
}

static double xo_f64(xoshiro *g) { return (double)(xo_next(g) >> 11) * (1.0 / 9007199254740992.0); }

static int xo_bool(xoshiro *g) { return (int32_t)(uint32_t)(xo_next(g) >> 32) < 0; }

/* rand 0.8 UniformInt::sample_single for u64 ranges [0, range) (widening-multiply rejection) */
static uint64_t xo_below(xoshiro *g, uint64_t range)
{
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        unsigned __int128 m = (unsigned __int128)xo_next(g) * range;
        if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
    }
}

/* ------------------------------------------------------------------------------------------
 * Energy of an explicit configuration: E = sum J s_a s_b - sum h s   (README.md:45-46).
 * state: one byte per spin, nonzero = True = +1.
 * ---------------------------------------------------------------------------------------- */
double orc_energy(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej,
                  size_t nvars, const double *biases, const uint8_t *state)
{
    double e = 0.0;
    for (size_t k = 0; k < n_edges; k++) {
        int sa = state[ea[k]] ? 1 : -1, sb = state[eb[k]] ? 1 : -1;
        e += ej[k] * (double)(sa * sb);
    }
    if (biases)
        for (size_t i = 0; i < nvars; i++) e -= biases[i] * (state[i] ? 1.0 : -1.0);
    return e;
}

/* ==========================================================================================
 * A. reference-faithful engine
 * ======================================================================================== */
typedef struct {
    size_t nvars;
    size_t *ptr;     /* CSR row pointers, nvars+1 */
    uint32_t *nbr;   /* neighbour ids */
    double *w;       /* couplings */
} adjacency;

static void adj_build(adjacency *A, size_t n_edges, const uint64_t *ea, const uint64_t *eb,
                      const double *ej, size_t nvars)
{
    A->nvars = nvars;
    A->ptr = calloc(nvars + 1, sizeof(size_t));
    for (size_t k = 0; k < n_edges; k++)
        if (ea[k] != eb[k]) { A->ptr[ea[k] + 1]++; A->ptr[eb[k] + 1]++; }
    for (size_t i = 0; i < nvars; i++) A->ptr[i + 1] += A->ptr[i];
    size_t nnz = A->ptr[nvars];
    A->nbr = malloc((nnz ? nnz : 1) * sizeof(uint32_t));
    A->w = malloc((nnz ? nnz : 1) * sizeof(double));
    size_t *fill = malloc((nvars + 1) * sizeof(size_t));
    memcpy(fill, A->ptr, (nvars + 1) * sizeof(size_t));
    for (size_t k = 0; k < n_edges; k++) { /* neighbours of i appear in edge-list order */
        if (ea[k] == eb[k]) continue;
        A->nbr[fill[ea[k]]] = (uint32_t)eb[k]; A->w[fill[ea[k]]++] = ej[k];
        A->nbr[fill[eb[k]]] = (uint32_t)ea[k]; A->w[fill[eb[k]]++] = ej[k];
    }
    free(fill);
}

static void adj_free(adjacency *A) { free(A->ptr); free(A->nbr); free(A->w); }

/*
 * One experiment of Lattice::run_monte_carlo (lattice.rs:197-212): seed -> rng, random initial
 * state unless `initial` is given, `timesteps` x do_time_step(beta), then energy + state.
 * betas: per-timestep beta (length timesteps) so that the annealing variants
 * (lattice.rs:358-368) can reuse it; energies_per_step may be NULL (lattice.rs:445-455).
 */
static void ref_run_one(const adjacency *A, const double *biases, uint64_t seed,
                        const uint8_t *initial, const double *betas, size_t timesteps,
                        uint8_t *state, double *energy_out, double *energies_per_step,
                        size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej)
{
    size_t n = A->nvars;
    xoshiro g;
    xo_seed_from_u64(&g, seed);
    if (initial) memcpy(state, initial, n);
    else for (size_t i = 0; i < n; i++) state[i] = (uint8_t)xo_bool(&g);
    for (size_t t = 0; t < timesteps; t++) {
        double beta = betas[t];
        for (size_t a = 0; a < n; a++) {
            size_t i = (size_t)xo_below(&g, n);
            double si = state[i] ? 1.0 : -1.0;
            double field = 0.0;
            for (size_t e = A->ptr[i]; e < A->ptr[i + 1]; e++)
                field += A->w[e] * (state[A->nbr[e]] ? 1.0 : -1.0);
            double dE = 2.0 * si * ((biases ? biases[i] : 0.0) - field);
            if (dE <= 0.0 || xo_f64(&g) < exp(-beta * dE)) state[i] = !state[i];
        }
        if (energies_per_step)
            energies_per_step[t] = orc_energy(n_edges, ea, eb, ej, n, biases, state);
    }
    if (energy_out) *energy_out = orc_energy(n_edges, ea, eb, ej, n, biases, state);
}

/*
 * R experiments, one per OpenMP thread (the rayon fan-out of lattice.rs:192-197).
 * states_out: uint8[R][nvars]; energies_out: double[R]; energies_per_step: double[R][T] or NULL.
 */
void orc_ref_run(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej,
                 size_t nvars, const double *biases, const uint64_t *seeds, size_t R,
                 const uint8_t *initial, const double *betas, size_t timesteps,
                 uint8_t *states_out, double *energies_out, double *energies_per_step)
{
    adjacency A;
    adj_build(&A, n_edges, ea, eb, ej, nvars);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t r = 0; r < R; r++)
        ref_run_one(&A, biases, seeds[r], initial, betas, timesteps, states_out + r * nvars,
                    energies_out ? energies_out + r : NULL,
                    energies_per_step ? energies_per_step + r * timesteps : NULL, n_edges, ea, eb,
                    ej);
    adj_free(&A);
}

/*
 * Equilibrium averages of engine A for the observables table (tests/observables.py): per chain, `therm` timesteps at beta,
 * then `steps` timesteps during which the energy (updated incrementally by the accepted dE) and |M| are accumulated after
 * every timestep.  mean_e_out / mean_absm_out: double[R].
 */
void orc_ref_averages(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej, size_t nvars,
                      const double *biases, const uint64_t *seeds, size_t R, const uint8_t *initial, double beta,
                      size_t therm, size_t steps, double *mean_e_out, double *mean_absm_out)
{
    adjacency A;
    adj_build(&A, n_edges, ea, eb, ej, nvars);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t r = 0; r < R; r++) {
        uint8_t *state = malloc(nvars ? nvars : 1);
        xoshiro g;
        xo_seed_from_u64(&g, seeds[r]);
        if (initial) memcpy(state, initial, nvars);
        else for (size_t i = 0; i < nvars; i++) state[i] = (uint8_t)xo_bool(&g);
        double e = orc_energy(n_edges, ea, eb, ej, nvars, biases, state), sum_e = 0.0, sum_m = 0.0;
        int64_t m = 0;
        for (size_t i = 0; i < nvars; i++) m += state[i] ? 1 : -1;
        for (size_t t = 0; t < therm + steps; t++) {
            for (size_t a = 0; a < nvars; a++) {
                size_t i = (size_t)xo_below(&g, nvars);
                double si = state[i] ? 1.0 : -1.0;
                double field = 0.0;
                for (size_t k = A.ptr[i]; k < A.ptr[i + 1]; k++)
                    field += A.w[k] * (state[A.nbr[k]] ? 1.0 : -1.0);
                double dE = 2.0 * si * ((biases ? biases[i] : 0.0) - field);
                if (dE <= 0.0 || xo_f64(&g) < exp(-beta * dE)) {
                    state[i] = !state[i];
                    e += dE;
                    m += state[i] ? 2 : -2;
                }
            }
            if (t >= therm) { sum_e += e; sum_m += (double)(m < 0 ? -m : m); }
        }
        mean_e_out[r] = steps ? sum_e / (double)steps : e;
        mean_absm_out[r] = steps ? sum_m / (double)steps : (double)(m < 0 ? -m : m);
        free(state);
    }
    adj_free(&A);
}

/*
 * Timed variant for bench.py's cpu_baseline leg: the same R chains on `threads` OpenMP threads,
 * constant beta, no outputs kept; *seconds_out = wall time of the sweep loop only (adjacency
 * construction and the random start are excluded, as on the GPU side).  Returns a checksum of the
 * final configurations so the work cannot be optimised away.
 */
uint64_t orc_ref_bench(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej,
                       size_t nvars, const uint64_t *seeds, size_t R, double beta,
                       size_t timesteps, int threads, double *seconds_out)
{
    adjacency A;
    adj_build(&A, n_edges, ea, eb, ej, nvars);
    uint8_t *states = malloc(R * nvars);
    xoshiro *rngs = malloc(R * sizeof(xoshiro));
    for (size_t r = 0; r < R; r++) {
        xo_seed_from_u64(&rngs[r], seeds[r]);
        for (size_t i = 0; i < nvars; i++) states[r * nvars + i] = (uint8_t)xo_bool(&rngs[r]);
    }
    double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (size_t r = 0; r < R; r++) {
        uint8_t *state = states + r * nvars;
        xoshiro g = rngs[r];
        for (size_t t = 0; t < timesteps; t++)
            for (size_t a = 0; a < nvars; a++) {
                size_t i = (size_t)xo_below(&g, nvars);
                double si = state[i] ? 1.0 : -1.0;
                double field = 0.0;
                for (size_t e = A.ptr[i]; e < A.ptr[i + 1]; e++)
                    field += A.w[e] * (state[A.nbr[e]] ? 1.0 : -1.0);
                double dE = 2.0 * si * (0.0 - field);
                if (dE <= 0.0 || xo_f64(&g) < exp(-beta * dE)) state[i] = !state[i];
            }
    }
    *seconds_out = omp_get_wtime() - t0;
    uint64_t sum = 0;
    for (size_t k = 0; k < R * nvars; k++) sum += states[k];
    free(states); free(rngs);
    adj_free(&A);
    return sum;
}

/* ==========================================================================================
 * Acceptance threshold, THR_BITS-bit fixed point: accept iff u < T, u uniform on [0, 2^THR_BITS).
 * T = 2^THR_BITS (always) when dE <= 0 or exp(-beta dE) >= 1.
 * ======================================================================================== */
uint64_t orc_threshold_fixed(double beta, double dE)
{
    const uint64_t ONE = (uint64_t)1 << THR_BITS;
    if (dE <= 0.0) return ONE;
    double p = exp(-beta * dE);
    if (!(p < 1.0)) return ONE;
    return (uint64_t)floor(ldexp(p, THR_BITS));
}

/* ==========================================================================================
 * B. checkerboard lattice engine (uniform |J|, per-bond sign), periodic W x H.
 *
 * Layout restated from DESIGN.md S2: colour c = (x+y)&1; the colour-c sites of row y are
 * x = 2i + ((y+c)&1), i = 0..W/2-1; plane c stores row y as W/64 words, bit (i&31) of word
 * y*(W/64) + (i>>5).  state = plane 0 followed by plane 1.  A quad = 4 consecutive words of a
 * plane (row-major linear word index / 4).
 * ======================================================================================== */
typedef struct { int W, H, wpr; size_t wpp; } lat_geom;

static lat_geom lat_make(int W, int H)
{
    lat_geom g = {W, H, W / 64, (size_t)H * (size_t)(W / 64)};
    return g;
}

int orc_lat_supported(int W, int H)
{
    return W >= 64 && W % 64 == 0 && H >= 2 && H % 2 == 0 && ((size_t)H * (size_t)(W / 64)) % 4 == 0;
}

size_t orc_lat_state_words(int W, int H) { return 2 * lat_make(W, H).wpp; }

static int lat_get(const lat_geom *g, const uint32_t *state, int y, int x)
{
    int c = (x + y) & 1, i = x >> 1;
    return (state[(size_t)c * g->wpp + (size_t)y * g->wpr + (i >> 5)] >> (i & 31)) & 1;
}

/* S2: random initial state, word w of plane c = Philox(key, (0, w>>2, c<<8, DOM_LAT_INIT))[w&3] */
void orc_lat_init(int W, int H, uint64_t seed, uint32_t *state)
{
    lat_geom g = lat_make(W, H);
    for (uint32_t c = 0; c < 2; c++)
        for (size_t w = 0; w < g.wpp; w++) {
            uint32_t r[4];
            philox_seeded(seed, 0, (uint32_t)(w >> 2), ctr2(0, c, 0), DOM_LAT_INIT, r);
            state[c * g.wpp + w] = r[w & 3];
        }
}

void orc_lat_pack(int W, int H, const uint8_t *spins, uint32_t *state)
{
    lat_geom g = lat_make(W, H);
    memset(state, 0, 2 * g.wpp * sizeof(uint32_t));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            if (spins[(size_t)y * W + x]) {
                int c = (x + y) & 1, i = x >> 1;
                state[(size_t)c * g.wpp + (size_t)y * g.wpr + (i >> 5)] |= 1u << (i & 31);
            }
}

void orc_lat_unpack(int W, int H, const uint32_t *state, uint8_t *spins)
{
    lat_geom g = lat_make(W, H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) spins[(size_t)y * W + x] = (uint8_t)lat_get(&g, state, y, x);
}

/*
 * Couplings: jabs = |J| of every bond; jright[y*W+x] / jdown[y*W+x] = 1 if the bond from
 * (y,x) to (y,x+1) / (y+1,x) has J > 0 (antiferromagnetic), else 0.  NULL => uniform sign
 * given by jpos_uniform.
 */
static int bond_pos(const uint8_t *plane, int jpos_uniform, int W, int y, int x)
{
    return plane ? plane[(size_t)y * W + x] : jpos_uniform;
}

/* satisfied (J s s < 0) and unsatisfied bonds of site (y,x) among the bonds that EXIST: with open boundaries in x
 * (open_x) the bonds between columns W-1 and 0 are absent, with open_y those between rows H-1 and 0 */
static void lat_bond_counts(const lat_geom *g, const uint32_t *state, const uint8_t *jright,
                            const uint8_t *jdown, int jpos_uniform, int open_x, int open_y, int y, int x,
                            int *sat, int *unsat)
{
    /* sat[0], unsat[0]: the horizontal bonds; sat[1], unsat[1]: the vertical bonds */
    int W = g->W, H = g->H;
    int s = lat_get(g, state, y, x);
    int xl = (x + W - 1) % W, xr = (x + 1) % W, yu = (y + H - 1) % H, yd = (y + 1) % H;
    int ok;
    sat[0] = sat[1] = unsat[0] = unsat[1] = 0;
    /* a bond with J>0 is satisfied when the spins differ, with J<0 when they agree */
    if (!(open_x && x == W - 1)) { ok = (s != lat_get(g, state, y, xr)) == bond_pos(jright, jpos_uniform, W, y, x); sat[0] += ok; unsat[0] += !ok; }
    if (!(open_x && x == 0)) { ok = (s != lat_get(g, state, y, xl)) == bond_pos(jright, jpos_uniform, W, y, xl); sat[0] += ok; unsat[0] += !ok; }
    if (!(open_y && y == H - 1)) { ok = (s != lat_get(g, state, yd, x)) == bond_pos(jdown, jpos_uniform, W, y, x); sat[1] += ok; unsat[1] += !ok; }
    if (!(open_y && y == 0)) { ok = (s != lat_get(g, state, yu, x)) == bond_pos(jdown, jpos_uniform, W, yu, x); sat[1] += ok; unsat[1] += !ok; }
}

/*
 * S3: one timestep t = colour 0 pass then colour 1 pass.  A spin s (+1 for a set bit) with `sat` satisfied and
 * `unsat` unsatisfied bonds in a uniform field h (E = sum J s s - h sum s, lattice.rs:129-131 set_global_bias,
 * classicising.rs:69 longitudinal) costs dE = 2|J|(sat - unsat) + 2 h s to flip; with T = fixed(exp(-beta dE))
 * (2^THR_BITS = always, in particular whenever dE <= 0) it flips iff u < T, where
 * u = (N_PLANES-bit prefix from the quad's bit-planes) << 32 | (32-bit residual word, drawn only when T is not
 * "always" and the prefix equals the threshold's top N_PLANES bits: a "tie"; the n-th tie of the quad in (word, bit)
 * order takes word n%4 of call N_PLANES + n/4).  Periodic, h = 0: dE = 2|J|(2k-4), the k = 3, 4 classes of round 1.
 * Anisotropic (jabs_y >= 0: the vertical bonds' |J|, jabs then the horizontal bonds'):
 * dE = 2|Jx|(sat_x - unsat_x) + 2|Jy|(sat_y - unsat_y) + 2 h s.
 */
void orc_lat_sweep_ex2(int W, int H, double jabs, double jabs_y, int jpos_uniform, const uint8_t *jright,
                       const uint8_t *jdown, double h, const uint8_t *hneg, int open_x, int open_y, uint32_t *state,
                       uint64_t seed, uint64_t t, double beta)
{
    /* hneg (optional, [H*W]): 1 where the site's field is -h instead of +h (fields of one size and both signs) */
    lat_geom g = lat_make(W, H);
    const uint64_t ONE = (uint64_t)1 << THR_BITS;
    size_t nquads = g.wpp / 4;
    for (uint32_t c = 0; c < 2; c++) {
        uint32_t *own = state + c * g.wpp;
        for (size_t Q = 0; Q < nquads; Q++) {
            uint32_t planes[N_PLANES][4];
            for (uint32_t p = 0; p < N_PLANES; p++)
                philox_seeded(seed, (uint32_t)t, (uint32_t)Q, DOM_LAT_SWEEP, ctr2(t, c, p),
                              planes[p]);
            uint32_t resid[4];
            unsigned n_undecided = 0;
            for (int q = 0; q < 4; q++) {
                size_t w = 4 * Q + q;
                int y = (int)(w / g.wpr), xw = (int)(w % g.wpr);
                uint32_t flip = 0;
                for (int b = 0; b < 32; b++) {
                    int i = 32 * xw + b;
                    int x = 2 * i + ((y + (int)c) & 1);
                    int sat[2], unsat[2];
                    lat_bond_counts(&g, state, jright, jdown, jpos_uniform, open_x, open_y, y, x, sat, unsat);
                    double sval = lat_get(&g, state, y, x) ? 1.0 : -1.0;
                    double hi_ = (hneg && hneg[(size_t)y * W + x]) ? -h : h;
                    double dE = jabs_y < 0.0
                        ? 2.0 * jabs * (double)(sat[0] + sat[1] - unsat[0] - unsat[1]) + 2.0 * hi_ * sval
                        : 2.0 * jabs * (double)(sat[0] - unsat[0]) + 2.0 * jabs_y * (double)(sat[1] - unsat[1]) + 2.0 * hi_ * sval;
                    uint64_t T = orc_threshold_fixed(beta, dE);
                    int accept;
                    if (T == ONE) accept = 1;
                    else {
                        uint32_t hi = (uint32_t)(T >> 32), lo = (uint32_t)T;
                        uint32_t upre = 0;
                        for (int p = 0; p < N_PLANES; p++)
                            upre = (upre << 1) | ((planes[p][q] >> b) & 1u);
                        if (upre < hi) accept = 1;
                        else if (upre > hi) accept = 0;
                        else {
                            if ((n_undecided & 3) == 0)
                                philox_seeded(seed, (uint32_t)t, (uint32_t)Q, DOM_LAT_SWEEP,
                                              ctr2(t, c, N_PLANES + n_undecided / 4), resid);
                            accept = resid[n_undecided & 3] < lo;
                            n_undecided++;
                        }
                    }
                    flip |= (uint32_t)accept << b;
                }
                /* same-colour spins are never neighbours: in-place update is the simultaneous one */
                own[w] ^= flip;
            }
        }
    }
}

void orc_lat_sweep_ex(int W, int H, double jabs, int jpos_uniform, const uint8_t *jright,
                      const uint8_t *jdown, double h, int open_x, int open_y, uint32_t *state, uint64_t seed,
                      uint64_t t, double beta)
{
    orc_lat_sweep_ex2(W, H, jabs, -1.0, jpos_uniform, jright, jdown, h, NULL, open_x, open_y, state, seed, t, beta);
}

void orc_lat_sweep(int W, int H, double jabs, int jpos_uniform, const uint8_t *jright,
                   const uint8_t *jdown, uint32_t *state, uint64_t seed, uint64_t t, double beta)
{
    orc_lat_sweep_ex(W, H, jabs, jpos_uniform, jright, jdown, 0.0, 0, 0, state, seed, t, beta);
}

/* E = sum over existing bonds of J s s (J = +-jabs; anisotropic, jabs_y >= 0: +-jabs horizontally, +-jabs_y vertically)
 * - h sum s, and M = sum s, from the packed state */
void orc_lat_energy_mag_ex2(int W, int H, double jabs, double jabs_y, int jpos_uniform, const uint8_t *jright,
                            const uint8_t *jdown, double h, const uint8_t *hneg, int open_x, int open_y,
                            const uint32_t *state, double *energy, int64_t *mag)
{
    lat_geom g = lat_make(W, H);
    int64_t ums[2] = {0, 0}, m = 0, mh = 0; /* mh = sum of s_i x sign of the site's field */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int s = lat_get(&g, state, y, x);
            m += s ? 1 : -1;
            mh += (s != 0) != (hneg && hneg[(size_t)y * W + x]) ? 1 : -1;
            int sr = lat_get(&g, state, y, (x + 1) % W), sd = lat_get(&g, state, (y + 1) % H, x);
            if (!(open_x && x == W - 1))
                ums[0] += ((s != sr) == bond_pos(jright, jpos_uniform, W, y, x)) ? -1 : 1;
            if (!(open_y && y == H - 1))
                ums[1] += ((s != sd) == bond_pos(jdown, jpos_uniform, W, y, x)) ? -1 : 1;
        }
    /* the terms separately, then one subtraction: the engine forms the same expression from its counters */
    if (energy)
        *energy = jabs_y < 0.0 ? jabs * (double)(ums[0] + ums[1]) - h * (double)mh
                               : jabs * (double)ums[0] + jabs_y * (double)ums[1] - h * (double)mh;
    if (mag) *mag = m;
}

void orc_lat_energy_mag_ex(int W, int H, double jabs, int jpos_uniform, const uint8_t *jright,
                           const uint8_t *jdown, double h, int open_x, int open_y, const uint32_t *state,
                           double *energy, int64_t *mag)
{
    orc_lat_energy_mag_ex2(W, H, jabs, -1.0, jpos_uniform, jright, jdown, h, NULL, open_x, open_y, state, energy, mag);
}

void orc_lat_energy_mag(int W, int H, double jabs, int jpos_uniform, const uint8_t *jright,
                        const uint8_t *jdown, const uint32_t *state, double *energy, int64_t *mag)
{
    orc_lat_energy_mag_ex(W, H, jabs, jpos_uniform, jright, jdown, 0.0, 0, 0, state, energy, mag);
}

/* ==========================================================================================
 * C. general edge-list engine
 * ======================================================================================== */

/* deterministic exp for the acceptance test: IEEE f64 ops + fma only, so that gcc on x86 and
 * hipcc on gfx950 produce identical bits.  Domain of use: x = -beta dE. */
double orc_det_exp(double x)
{
    if (x >= 0.0) return 1.0;
    if (x < -40.0) return 0.0; /* below 2^-53: can never beat a 53-bit uniform */
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    double kf = floor(fma(x, LOG2E, 0.5));
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    /* Taylor to degree 13 on |r| <= 0.35: truncation < 5e-18 */
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int64_t k = (int64_t)kf; /* in [-58, 0] */
    uint64_t bits = (uint64_t)(1023 + k) << 52;
    double scale;
    memcpy(&scale, &bits, 8);
    return p * scale;
}

/* greedy colouring in site order: smallest colour unused by already-coloured neighbours */
typedef struct {
    adjacency A;
    uint32_t *colour;   /* per site */
    uint32_t ncolours;
    size_t *class_base; /* first packed position of each colour class, ncolours+1 */
    size_t *pos;        /* site -> packed position (colour-major, classes padded to 256) */
    size_t npos;
} gen_graph;

static void gen_build(gen_graph *G, size_t n_edges, const uint64_t *ea, const uint64_t *eb,
                      const double *ej, size_t nvars)
{
    adj_build(&G->A, n_edges, ea, eb, ej, nvars);
    G->colour = malloc(nvars * sizeof(uint32_t));
    uint32_t nc = 1;
    size_t maxdeg = 0;
    for (size_t i = 0; i < nvars; i++)
        if (G->A.ptr[i + 1] - G->A.ptr[i] > maxdeg) maxdeg = G->A.ptr[i + 1] - G->A.ptr[i];
    uint8_t *used = calloc(maxdeg + 2, 1);
    for (size_t i = 0; i < nvars; i++) {
        size_t deg = G->A.ptr[i + 1] - G->A.ptr[i];
        memset(used, 0, deg + 2);
        for (size_t e = G->A.ptr[i]; e < G->A.ptr[i + 1]; e++) {
            uint32_t j = G->A.nbr[e];
            if (j < i && G->colour[j] <= deg) used[G->colour[j]] = 1;
        }
        uint32_t c = 0;
        while (used[c]) c++;
        G->colour[i] = c;
        if (c + 1 > nc) nc = c + 1;
    }
    free(used);
    G->ncolours = nc;
    size_t *count = calloc(nc, sizeof(size_t));
    for (size_t i = 0; i < nvars; i++) count[G->colour[i]]++;
    G->class_base = malloc((nc + 1) * sizeof(size_t));
    G->class_base[0] = 0;
    for (uint32_t c = 0; c < nc; c++) G->class_base[c + 1] = G->class_base[c] + ((count[c] + 255) / 256) * 256;
    G->npos = G->class_base[nc];
    G->pos = malloc(nvars * sizeof(size_t));
    memset(count, 0, nc * sizeof(size_t));
    for (size_t i = 0; i < nvars; i++) G->pos[i] = G->class_base[G->colour[i]] + count[G->colour[i]]++;
    free(count);
}

static void gen_free(gen_graph *G)
{
    adj_free(&G->A);
    free(G->colour); free(G->class_base); free(G->pos);
}

/* expose the colouring for host-logic tests: colours[nvars], returns the number of colours */
uint32_t orc_gen_colouring(size_t n_edges, const uint64_t *ea, const uint64_t *eb,
                           const double *ej, size_t nvars, uint32_t *colours, uint64_t *positions)
{
    gen_graph G;
    gen_build(&G, n_edges, ea, eb, ej, nvars);
    for (size_t i = 0; i < nvars; i++) {
        if (colours) colours[i] = G.colour[i];
        if (positions) positions[i] = G.pos[i];
    }
    uint32_t nc = G.ncolours;
    gen_free(&G);
    return nc;
}

/*
 * One experiment on the general path.  state: one byte per spin in SITE order.
 * initial == NULL: S2' random start -- packed word w (positions 32w..32w+31) =
 * Philox(key, (0, w>>2, 0, DOM_GEN_INIT))[w&3]; site i takes bit pos(i)&31 of word pos(i)>>5.
 * Timestep t (absolute, t0 + local index): colour classes in order; site i draws
 * Philox(key, (t_lo, i>>1, t_hi16<<16, DOM_GEN_SWEEP)) words 2(i&1), 2(i&1)+1 -> 53-bit u.
 */
void orc_gen_run(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej,
                 size_t nvars, const double *biases, uint64_t seed, const uint8_t *initial,
                 uint64_t t0, const double *betas, size_t timesteps, uint8_t *state,
                 double *energy_out, double *energies_per_step)
{
    gen_graph G;
    gen_build(&G, n_edges, ea, eb, ej, nvars);
    if (initial) memcpy(state, initial, nvars);
    else if (t0 == 0)
        for (size_t i = 0; i < nvars; i++) {
            size_t p = G.pos[i], w = p >> 5;
            uint32_t r[4];
            philox_seeded(seed, 0, (uint32_t)(w >> 2), 0, DOM_GEN_INIT, r);
            state[i] = (uint8_t)((r[w & 3] >> (p & 31)) & 1u);
        }
    /* (t0 != 0 with initial == NULL: continue from the state already in `state`) */
    uint8_t *next = malloc(nvars ? nvars : 1);
    for (size_t k = 0; k < timesteps; k++) {
        uint64_t t = t0 + k;
        double beta = betas[k];
        for (uint32_t c = 0; c < G.ncolours; c++) {
            memcpy(next, state, nvars);
            for (size_t i = 0; i < nvars; i++) {
                if (G.colour[i] != c) continue;
                double si = state[i] ? 1.0 : -1.0;
                double field = 0.0;
                for (size_t e = G.A.ptr[i]; e < G.A.ptr[i + 1]; e++)
                    field += G.A.w[e] * (state[G.A.nbr[e]] ? 1.0 : -1.0);
                double dE = 2.0 * si * ((biases ? biases[i] : 0.0) - field);
                int accept = dE <= 0.0;
                if (!accept) {
                    uint32_t r[4];
                    philox_seeded(seed, (uint32_t)t, (uint32_t)(i >> 1), ctr2(t, 0, 0),
                                  DOM_GEN_SWEEP, r);
                    uint64_t x = ((uint64_t)r[2 * (i & 1) + 1] << 32) | r[2 * (i & 1)];
                    double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
                    accept = u < orc_det_exp(-beta * dE);
                }
                if (accept) next[i] = !state[i];
            }
            memcpy(state, next, nvars);
        }
        if (energies_per_step)
            energies_per_step[k] = orc_energy(n_edges, ea, eb, ej, nvars, biases, state);
    }
    free(next);
    if (energy_out) *energy_out = orc_energy(n_edges, ea, eb, ej, nvars, biases, state);
    gen_free(&G);
}

/* ==========================================================================================
 * Parallel-tempering swap decision, shaped after the countdown loop of
 * tempering.rs:172-212 (quantum in the reference; the classical ladder is the build's own).
 * Round `round`, parity = round & 1: pairs (i, i+1) for i = parity, parity+2, ...; rung i holds
 * replica slot perm[i].  Swap iff u53 < exp((beta_i - beta_j)(E_i - E_j)) with
 * u = Philox(key = seed, (i, round_lo, round_hi, DOM_PT_SWAP)) words 0,1.  Swaps exchange the
 * betas (perm entries), never the configurations.  Returns the number of accepted swaps.
 * ======================================================================================== */
uint64_t orc_pt_swap_round(uint64_t seed, uint64_t round, size_t n_rungs, const double *betas,
                           const double *slot_energy, uint32_t *perm)
{
    uint64_t swaps = 0;
    for (size_t i = round & 1; i + 1 < n_rungs; i += 2) {
        double d = (betas[i] - betas[i + 1]) * (slot_energy[perm[i]] - slot_energy[perm[i + 1]]);
        int accept = d >= 0.0;
        if (!accept) {
            uint32_t r[4];
            philox_seeded(seed, (uint32_t)i, (uint32_t)round, (uint32_t)(round >> 32), DOM_PT_SWAP, r);
            uint64_t x = ((uint64_t)r[1] << 32) | r[0];
            double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
            accept = u < orc_det_exp(d);
        }
        if (accept) {
            uint32_t tmp = perm[i]; perm[i] = perm[i + 1]; perm[i + 1] = tmp;
            swaps++;
        }
    }
    return swaps;
}

/* ==========================================================================================
 * D. replica-packed general engine: graphs with one |J|, no fields, degree <= 6.
 * Same colouring and positions as engine C; replicas live in groups of 32 that share Philox calls
 * (replica 32g+b takes bit b of every random word of group g, whose key is the seed of replica 32g).
 * A spin with k satisfied bonds out of deg flips always when m = 2k - deg <= 0, else iff
 * u < T_m = fixed(exp(-beta 2|J| m)), u built exactly as in S3: N_PLANES prefix bits from the
 * position-quad's plane words, 32 more bits for ties, which are numbered over the whole
 * position-quad in (position, replica bit) order -- over all 32 bits, so the unused replicas of a
 * last, partial group are simulated too (they take tie words).
 * states: uint8[32*G][nvars], G = ceil(R/32); replicas >= R are the padding of the last group.
 * betas: per timestep (beta_replica == NULL) or beta_replica[R] (then padding replicas use
 * beta_replica[R-1]).
 * ======================================================================================== */
#define DOM_PK_SWEEP 0x504B5357u
#define DOM_PK_INIT 0x504B494Eu

void orc_pk_run(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej,
                size_t nvars, const uint64_t *seeds, size_t R, int random_start, uint64_t t0,
                const double *betas, const double *beta_replica, size_t timesteps,
                uint8_t *states, double *energies_out, double *energies_per_step)
{
    gen_graph G;
    gen_build(&G, n_edges, ea, eb, ej, nvars);
    double jabs = 0.0;
    for (size_t k = 0; k < n_edges; k++)
        if (ea[k] != eb[k]) { jabs = fabs(ej[k]); break; }
    size_t groups = (R + 31) / 32;
    /* site of each position (SIZE_MAX on padding) */
    size_t *site_of = malloc(G.npos * sizeof(size_t));
    for (size_t p = 0; p < G.npos; p++) site_of[p] = (size_t)-1;
    for (size_t i = 0; i < nvars; i++) site_of[G.pos[i]] = i;

    for (size_t g = 0; g < groups; g++) {
        uint64_t key = seeds[32 * g];
        uint8_t *S = states + 32 * g * nvars; /* S[b*nvars + i] */
        if (random_start)
            for (size_t i = 0; i < nvars; i++) {
                size_t p = G.pos[i];
                uint32_t r[4];
                size_t q = (p & 255) >> 6; /* position p is word q of the quad led by p - 64 q */
                philox_seeded(key, 0, (uint32_t)(p - 64 * q), 0, DOM_PK_INIT, r);
                for (int b = 0; b < 32; b++) S[(size_t)b * nvars + i] = (uint8_t)((r[q] >> b) & 1u);
            }
        for (size_t k = 0; k < timesteps; k++) {
            uint64_t t = t0 + k;
            for (uint32_t c = 0; c < G.ncolours; c++)
                /* position-quads: leader p0 (offset < 64 inside its 256-block), members p0 + 64 q */
                for (size_t p0 = G.class_base[c]; p0 < G.class_base[c + 1]; p0 += ((p0 & 63) == 63 ? 193 : 1)) {
                    uint32_t planes[N_PLANES][4], tie_words[4];
                    unsigned n_ties = 0;
                    for (uint32_t pl = 0; pl < N_PLANES; pl++)
                        philox_seeded(key, (uint32_t)t, (uint32_t)p0, DOM_PK_SWEEP, ctr2(t, 0, pl), planes[pl]);
                    for (int q = 0; q < 4; q++) {
                        size_t i = site_of[p0 + 64 * q];
                        if (i == (size_t)-1) continue;
                        int deg = (int)(G.A.ptr[i + 1] - G.A.ptr[i]);
                        for (int b = 0; b < 32; b++) {
                            uint8_t *s = S + (size_t)b * nvars;
                            size_t r = 32 * g + b;
                            double beta = beta_replica ? beta_replica[r < R ? r : R - 1] : betas[k];
                            int sat = 0;
                            for (size_t e = G.A.ptr[i]; e < G.A.ptr[i + 1]; e++) {
                                int differ = s[i] != s[G.A.nbr[e]];
                                sat += (G.A.w[e] > 0.0) ? differ : !differ;
                            }
                            int m = 2 * sat - deg, accept;
                            if (m <= 0) accept = 1;
                            else {
                                uint64_t T = orc_threshold_fixed(beta, 2.0 * jabs * (double)m);
                                uint32_t hi = (uint32_t)(T >> 32), lo = (uint32_t)T, upre = 0;
                                for (int pl = 0; pl < N_PLANES; pl++) upre = (upre << 1) | ((planes[pl][q] >> b) & 1u);
                                if (upre < hi) accept = 1;
                                else if (upre > hi) accept = 0;
                                else {
                                    if ((n_ties & 3) == 0)
                                        philox_seeded(key, (uint32_t)t, (uint32_t)p0, DOM_PK_SWEEP,
                                                      ctr2(t, 0, N_PLANES + n_ties / 4), tie_words);
                                    accept = tie_words[n_ties & 3] < lo;
                                    n_ties++;
                                }
                            }
                            /* the sites of a colour class are independent: in place == simultaneous */
                            if (accept) s[i] = !s[i];
                        }
                    }
                }
            if (energies_per_step)
                for (int b = 0; b < 32 && 32 * g + b < R; b++)
                    energies_per_step[(32 * g + b) * timesteps + k] =
                        orc_energy(n_edges, ea, eb, ej, nvars, NULL, S + (size_t)b * nvars);
        }
        if (energies_out)
            for (int b = 0; b < 32 && 32 * g + b < R; b++)
                energies_out[32 * g + b] = orc_energy(n_edges, ea, eb, ej, nvars, NULL, S + (size_t)b * nvars);
    }
    free(site_of);
    gen_free(&G);
}

/* ==========================================================================================
 * E. replica-packed REAL-COUPLING engine: any real J and any site biases
 * (lattice.rs:46-50 edge list, :104-131 set_individual_bias / set_global_bias, :186-189), graphs of
 * degree <= 31.  Same colouring, positions, replica groups, group keys and random start as engine D.
 * The acceptance test runs in the LOG domain on integers, so that nothing per attempt needs exp():
 *
 *   scales (once per graph)         F_i = |h_i| + sum_e |J_e| (adjacency order), Fmax = max_i F_i, med = the lower median
 *                                   of the nonzero |J_e| (every bond once) and |h_i|;
 *                                   k0 = ilogb(min(Fmax, 64 med)) + 1 - 30      (the graph's quantum 2^k0)
 *                                   k_i = max(k0, ilogb(F_i) + 1 - 30), d_i = min(k_i - k0, 31)   (a HEAVY site: d_i > 0)
 *   quantisation AS SEEN FROM SITE i  Jq_e = rint(J_e 2^-k_i), hq_i = rint(h_i 2^-k_i)   (int32, |X_i| < 2^30 + 32)
 *   half energy change              X_i = s_i (hq_i - sum_e Jq_e s_j)           (dE = 2 X_i 2^k_i)
 *   uniform                         u = word (b & 3) of Philox(key_g, (t_lo, p, "RJSW", ctr2(t,0,b>>2)))
 *                                   for replica bit b of the group at position p
 *   Lambda_q(u) ~ -log2(u / 2^32) in Q24 from the bits of (float)u: exponent field + a 2048-interval
 *                                   table of log2(1 + m) with linear interpolation
 *   per beta                        kappa = ln2 / (2 beta 2^k0) (quanta per unit of -log2 u);
 *                                   r = max(0, ilogb(kappa) - 23), mant = floor(kappa 2^(8 - r))
 *   accept                          iff max(X_i >> (r - m), 0) <= ((Lambda_q * mant) >> 32) >> (d_i - m),  m = min(r, d_i)
 *                                   (u / 2^32 < exp(-beta dE) with ~2^-23 relative resolution in beta dE; d_i = 0: m = 0)
 *   eligible                        degree <= 31, Fmax > 0, and every heavy site is DOMINATED by one term:
 *                                   4 max(|h_i|, max_e |J_e|) >= 3 F_i  (then |X_i| >= F_i / 2 whatever the spins: the coarser
 *                                   quantum 2^k_i of a heavy site -- one pinning bias, one enormous bond -- cannot change a
 *                                   decision that f64 arithmetic would take differently by more than 2^-29 of its exponent)
 *   energy                          of the ORIGINAL couplings, two integer levels: kE = ilogb(Fmax) + 2 - 30,
 *                                   hi = rint(x 2^-kE), lo = rint((x - hi 2^kE) 2^(24 - kE)) for every J_e (one value per bond)
 *                                   and h_i;  E = 2^kE S(hi) + 2^(kE-24) S(lo) + sum of self-loop J,  S(q) = sum_bonds q s s -
 *                                   sum_i q_i s_i as an exact int64 -- any summation order gives the same bits, and every
 *                                   term is within Fmax 2^-54 of its f64 value
 *
 * Written spin by spin with a direct integer field sum: the HIP kernel's per-site tables, bit
 * transposition and carry tricks are checked against this independently.
 * ======================================================================================== */
#define DOM_RJ_SWEEP 0x524A5357u /* "RJSW" */
#define RJ_MAX_DEG 31
#define RJ_LOG_INTERVALS 2048
#define RJ_LO_BITS 24

/* LT[i] ~ log2(1 + x_i) 2^24, x_i = i / 2048, i = 0 .. 2048, centred for the interpolation that uses it: the chord of
 * the concave log2 lies below the curve by up to h^2 log2(e) / (8 (1 + x)^2) (h = 1/2048) and the interpolation rounds
 * down, so every entry but the first carries half that gap plus half a unit: LT[i] = rint((log2(1 + x_i) + h^2 log2(e) /
 * (16 (1 + x_i)^2)) 2^24 + 0.5).  LT[0] = 0 exactly: Lambda_q of a u that rounds to 2^32 must be 0, not negative. */
void orc_rj_log_table(uint32_t *out)
{
    const double h = 1.0 / RJ_LOG_INTERVALS, LOG2E = 1.4426950408889634074;
    out[0] = 0;
    for (int i = 1; i <= RJ_LOG_INTERVALS; i++) {
        const double x = (double)i * h;
        out[i] = (uint32_t)nearbyint(ldexp(log2(1.0 + x) + h * h * LOG2E / (16.0 * (1.0 + x) * (1.0 + x)), 24) + 0.5);
    }
}

uint32_t orc_rj_lambda(uint32_t u)
{
    static uint32_t LT[RJ_LOG_INTERVALS + 1];
    static int have = 0;
    if (!have) { orc_rj_log_table(LT); have = 1; } /* engine E runs on one thread */
    float f = (float)u; /* round to nearest even: 24 significant bits of u */
    uint32_t bits;
    memcpy(&bits, &f, 4);
    uint32_t E = bits >> 23, idx = (bits >> 12) & 0x7FFu, frac = bits & 0xFFFu;
    uint32_t val = LT[idx] + (uint32_t)(((uint64_t)(LT[idx + 1] - LT[idx]) * frac) >> 12);
    return (159u << 24) - (E << 24) - val;
}

typedef struct {
    double fmax, median;
    size_t maxdeg;
    int k0, kE;
    int heavy;     /* some site has d_i > 0 */
    int dominated; /* every heavy site is dominated by one term */
    int *ksite;    /* k_i per site */
} rj_scales;

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

static void rj_analyse(const adjacency *A, size_t nvars, const double *biases, rj_scales *S)
{
    size_t nnz = A->ptr[nvars], terms = 0;
    double *mags = malloc((nnz / 2 + nvars + 1) * sizeof(double)), *F = malloc((nvars ? nvars : 1) * sizeof(double));
    S->fmax = 0.0;
    S->maxdeg = 0;
    for (size_t i = 0; i < nvars; i++) {
        double f = biases ? fabs(biases[i]) : 0.0;
        if (f != 0.0) mags[terms++] = f;
        for (size_t e = A->ptr[i]; e < A->ptr[i + 1]; e++) {
            f += fabs(A->w[e]);
            if (A->nbr[e] > i && A->w[e] != 0.0) mags[terms++] = fabs(A->w[e]); /* every bond once, from its lower end */
        }
        F[i] = f;
        if (f > S->fmax) S->fmax = f;
        if (A->ptr[i + 1] - A->ptr[i] > S->maxdeg) S->maxdeg = A->ptr[i + 1] - A->ptr[i];
    }
    S->median = 0.0;
    if (terms) {
        qsort(mags, terms, sizeof(double), cmp_double);
        S->median = mags[(terms - 1) / 2];
    }
    free(mags);
    double fbase = S->fmax < 64.0 * S->median ? S->fmax : 64.0 * S->median;
    S->k0 = fbase > 0.0 ? ilogb(fbase) + 1 - 30 : 0;
    S->kE = S->fmax > 0.0 ? ilogb(S->fmax) + 2 - 30 : 0;
    S->ksite = malloc((nvars ? nvars : 1) * sizeof(int));
    S->heavy = 0;
    S->dominated = 1;
    for (size_t i = 0; i < nvars; i++) {
        int ki = F[i] > 0.0 ? ilogb(F[i]) + 1 - 30 : S->k0;
        if (ki < S->k0) ki = S->k0;
        S->ksite[i] = ki;
        if (ki > S->k0) {
            S->heavy = 1;
            double m = biases ? fabs(biases[i]) : 0.0;
            for (size_t e = A->ptr[i]; e < A->ptr[i + 1]; e++)
                if (fabs(A->w[e]) > m) m = fabs(A->w[e]);
            if (!(4.0 * m >= 3.0 * F[i])) S->dominated = 0;
        }
    }
    free(F);
}

/* Quantisation for the dynamics: k0; per INPUT edge e the coupling as seen from its two ends, jq_out[2e] (from ea[e]) and
 * jq_out[2e+1] (from eb[e]) (0, 0 for self-loops); per site the bias hq_out[i] and the shift d_i dshift_out[i] */
void orc_rj_quantise(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej, size_t nvars,
                     const double *biases, int32_t *jq_out, int32_t *hq_out, uint8_t *dshift_out, int *k_out)
{
    adjacency A;
    adj_build(&A, n_edges, ea, eb, ej, nvars);
    rj_scales S;
    rj_analyse(&A, nvars, biases, &S);
    adj_free(&A);
    if (k_out) *k_out = S.k0;
    for (size_t e = 0; e < n_edges && jq_out; e++) {
        jq_out[2 * e] = ea[e] == eb[e] ? 0 : (int32_t)nearbyint(ldexp(ej[e], -S.ksite[ea[e]]));
        jq_out[2 * e + 1] = ea[e] == eb[e] ? 0 : (int32_t)nearbyint(ldexp(ej[e], -S.ksite[eb[e]]));
    }
    for (size_t i = 0; i < nvars; i++) {
        if (hq_out) hq_out[i] = biases ? (int32_t)nearbyint(ldexp(biases[i], -S.ksite[i])) : 0;
        if (dshift_out) dshift_out[i] = (uint8_t)(S.ksite[i] - S.k0 > 31 ? 31 : S.ksite[i] - S.k0);
    }
    free(S.ksite);
}

/* the two integer levels of the energy: per INPUT edge (0 for self-loops) and per site */
static void rj_two_levels(double x, int kE, int32_t *hi, int32_t *lo)
{
    double h = nearbyint(ldexp(x, -kE));
    *hi = (int32_t)h;
    *lo = (int32_t)nearbyint(ldexp(x - ldexp(h, kE), RJ_LO_BITS - kE)); /* x - hi 2^kE is exact in f64 */
}

void orc_rj_energy_levels(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej, size_t nvars,
                          const double *biases, int32_t *jhi, int32_t *jlo, int32_t *hhi, int32_t *hlo, int *kE_out)
{
    adjacency A;
    adj_build(&A, n_edges, ea, eb, ej, nvars);
    rj_scales S;
    rj_analyse(&A, nvars, biases, &S);
    adj_free(&A);
    free(S.ksite);
    if (kE_out) *kE_out = S.kE;
    for (size_t e = 0; e < n_edges; e++) {
        if (ea[e] == eb[e]) { jhi[e] = jlo[e] = 0; continue; }
        rj_two_levels(ej[e], S.kE, &jhi[e], &jlo[e]);
    }
    for (size_t i = 0; i < nvars; i++) {
        hhi[i] = hlo[i] = 0;
        if (biases) rj_two_levels(biases[i], S.kE, &hhi[i], &hlo[i]);
    }
}

/* eligibility for this path (see the header of this engine) */
int orc_rj_eligible(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej, size_t nvars,
                    const double *biases)
{
    adjacency A;
    adj_build(&A, n_edges, ea, eb, ej, nvars);
    rj_scales S;
    rj_analyse(&A, nvars, biases, &S);
    adj_free(&A);
    free(S.ksite);
    return S.maxdeg <= RJ_MAX_DEG && S.fmax > 0.0 && S.dominated;
}

void orc_rj_beta(double beta, int k, uint32_t *shift_out, uint32_t *mant_out)
{
    uint32_t shift = 31, mant = 0xFFFFFFFFu; /* beta <= 0: every attempt is accepted */
    if (beta > 0.0) {
        double kappa = ldexp(0.69314718055994530942 / (2.0 * beta), -k);
        int e = kappa > 0.0 && isfinite(kappa) ? ilogb(kappa) : (kappa > 0.0 ? 2000 : -2000);
        int r = e - 23 > 0 ? e - 23 : 0;
        if (r <= 31) {
            shift = (uint32_t)r;
            mant = (uint32_t)floor(ldexp(kappa, 8 - r)); /* kappa 2^-r < 2^24 */
        }
    }
    *shift_out = shift;
    *mant_out = mant;
}

/* d: the site's shift d_i (0 unless the site is heavy).  X is in units of 2^(k0 + d): of the d binary places between the
 * site's scale and the graph's, as many as possible come off the right shift of X (m = min(shift, d)), the rest off the bound */
int orc_rj_accept(int32_t X, uint32_t u, uint32_t shift, uint32_t mant, uint32_t d)
{
    uint32_t m = shift < d ? shift : d;
    int32_t xs = X >> (shift - m); /* arithmetic */
    uint32_t xpos = xs > 0 ? (uint32_t)xs : 0u;
    uint32_t y = (uint32_t)(((uint64_t)orc_rj_lambda(u) * mant) >> 32);
    return xpos <= (y >> (d - m));
}

/* E = 2^kE S(hi) + 2^(kE - 24) S(lo) + self-loop constant, S(q) = sum_bonds q s s - sum_i q_i s_i: exact integer sums, so any
 * order gives the same bits */
static double rj_energy(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const int32_t *jhi, const int32_t *jlo,
                        size_t nvars, const int32_t *hhi, const int32_t *hlo, int kE, double self_energy, const uint8_t *s)
{
    int64_t shi = 0, slo = 0;
    for (size_t e = 0; e < n_edges; e++)
        if (ea[e] != eb[e]) {
            int par = (s[ea[e]] != 0) == (s[eb[e]] != 0) ? 1 : -1;
            shi += (int64_t)jhi[e] * par;
            slo += (int64_t)jlo[e] * par;
        }
    for (size_t i = 0; i < nvars; i++) {
        shi -= (int64_t)hhi[i] * (s[i] ? 1 : -1);
        slo -= (int64_t)hlo[i] * (s[i] ? 1 : -1);
    }
    return (ldexp((double)shi, kE) + ldexp((double)slo, kE - RJ_LO_BITS)) + self_energy;
}

double orc_rj_energy(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej, size_t nvars,
                     const double *biases, const uint8_t *state)
{
    int32_t *jhi = malloc((n_edges ? n_edges : 1) * 4), *jlo = malloc((n_edges ? n_edges : 1) * 4);
    int32_t *hhi = malloc((nvars ? nvars : 1) * 4), *hlo = malloc((nvars ? nvars : 1) * 4);
    int kE;
    orc_rj_energy_levels(n_edges, ea, eb, ej, nvars, biases, jhi, jlo, hhi, hlo, &kE);
    double self_energy = 0.0;
    for (size_t e = 0; e < n_edges; e++)
        if (ea[e] == eb[e]) self_energy += ej[e];
    double E = rj_energy(n_edges, ea, eb, jhi, jlo, nvars, hhi, hlo, kE, self_energy, state);
    free(jhi); free(jlo); free(hhi); free(hlo);
    return E;
}

/* states: uint8[32*G][nvars] as engine D; betas per timestep, or beta_replica[R] (padding replicas of the
 * last group then use beta_replica[R-1]) */
void orc_rj_run(size_t n_edges, const uint64_t *ea, const uint64_t *eb, const double *ej,
                size_t nvars, const double *biases, const uint64_t *seeds, size_t R, int random_start, uint64_t t0,
                const double *betas, const double *beta_replica, size_t timesteps,
                uint8_t *states, double *energies_out, double *energies_per_step)
{
    gen_graph G;
    gen_build(&G, n_edges, ea, eb, ej, nvars);
    rj_scales SC;
    rj_analyse(&G.A, nvars, biases, &SC);
    const int k = SC.k0;
    /* couplings in adjacency order, each as seen from the row's site; biases; shifts */
    size_t nnz = G.A.ptr[nvars];
    int32_t *jq = malloc((nnz ? nnz : 1) * sizeof(int32_t)), *hq = malloc((nvars ? nvars : 1) * sizeof(int32_t));
    uint32_t *dsh = malloc((nvars ? nvars : 1) * sizeof(uint32_t));
    for (size_t i = 0; i < nvars; i++) {
        for (size_t e = G.A.ptr[i]; e < G.A.ptr[i + 1]; e++) jq[e] = (int32_t)nearbyint(ldexp(G.A.w[e], -SC.ksite[i]));
        hq[i] = biases ? (int32_t)nearbyint(ldexp(biases[i], -SC.ksite[i])) : 0;
        dsh[i] = (uint32_t)(SC.ksite[i] - k > 31 ? 31 : SC.ksite[i] - k);
    }
    int32_t *jhi = malloc((n_edges ? n_edges : 1) * 4), *jlo = malloc((n_edges ? n_edges : 1) * 4);
    int32_t *hhi = malloc((nvars ? nvars : 1) * 4), *hlo = malloc((nvars ? nvars : 1) * 4);
    int kE;
    orc_rj_energy_levels(n_edges, ea, eb, ej, nvars, biases, jhi, jlo, hhi, hlo, &kE);
    double self_energy = 0.0;
    for (size_t e = 0; e < n_edges; e++)
        if (ea[e] == eb[e]) self_energy += ej[e];
    size_t groups = (R + 31) / 32;
    for (size_t g = 0; g < groups; g++) {
        uint64_t key = seeds[32 * g];
        uint8_t *S = states + 32 * g * nvars;
        if (random_start)
            for (size_t i = 0; i < nvars; i++) {
                size_t p = G.pos[i];
                uint32_t r[4];
                size_t q = (p & 255) >> 6;
                philox_seeded(key, 0, (uint32_t)(p - 64 * q), 0, DOM_PK_INIT, r);
                for (int b = 0; b < 32; b++) S[(size_t)b * nvars + i] = (uint8_t)((r[q] >> b) & 1u);
            }
        for (size_t step = 0; step < timesteps; step++) {
            uint64_t t = t0 + step;
            uint32_t shift[32], mant[32]; /* the group's acceptance scales at this timestep */
            for (int b = 0; b < 32; b++) {
                size_t rr = 32 * g + b;
                orc_rj_beta(beta_replica ? beta_replica[rr < R ? rr : R - 1] : betas[step], k, &shift[b], &mant[b]);
            }
            for (uint32_t c = 0; c < G.ncolours; c++)
                for (size_t i = 0; i < nvars; i++) {
                    if (G.colour[i] != c) continue;
                    size_t p = G.pos[i];
                    uint32_t words[8][4];
                    for (uint32_t j = 0; j < 8; j++)
                        philox_seeded(key, (uint32_t)t, (uint32_t)p, DOM_RJ_SWEEP, ctr2(t, 0, j), words[j]);
                    for (int b = 0; b < 32; b++) {
                        uint8_t *s = S + (size_t)b * nvars;
                        int64_t field = 0;
                        for (size_t e = G.A.ptr[i]; e < G.A.ptr[i + 1]; e++)
                            field += (int64_t)jq[e] * (s[G.A.nbr[e]] ? 1 : -1);
                        int64_t X = (s[i] ? 1 : -1) * ((int64_t)hq[i] - field);
                        /* the sites of a colour class are independent: in place == simultaneous */
                        if (orc_rj_accept((int32_t)X, words[b >> 2][b & 3], shift[b], mant[b], dsh[i])) s[i] = !s[i];
                    }
                }
            if (energies_per_step)
                for (int b = 0; b < 32 && 32 * g + b < R; b++)
                    energies_per_step[(32 * g + b) * timesteps + step] =
                        rj_energy(n_edges, ea, eb, jhi, jlo, nvars, hhi, hlo, kE, self_energy, S + (size_t)b * nvars);
        }
        if (energies_out)
            for (int b = 0; b < 32 && 32 * g + b < R; b++)
                energies_out[32 * g + b] = rj_energy(n_edges, ea, eb, jhi, jlo, nvars, hhi, hlo, kE, self_energy, S + (size_t)b * nvars);
    }
    free(jq); free(hq); free(dsh); free(jhi); free(jlo); free(hhi); free(hlo); free(SC.ksite);
    gen_free(&G);
}
