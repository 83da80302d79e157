"""Test-side helpers: an oracle-backed stand-in for the HIP engine (same duck-typed interface as
pyisingmontecarlo_amd._capi.States) so that the sharding / tempering host logic can be exercised on
CPU-only machines.  TEST INFRASTRUCTURE: the product never uses this."""
import numpy as np

from oracle import oracle as O


class OracleLatStates:
    def __init__(self, lat, seeds):
        self.lat = lat
        self.seeds = [int(s) for s in seeds]
        self.st = [lat.init(s) for s in self.seeds]
        self.t = 0
        self.betas = None

    @property
    def count(self):
        return len(self.seeds)

    def set_betas(self, betas):
        self.betas = None if betas is None else [float(b) for b in betas]

    def do_time_steps(self, timesteps, beta=None, per_step_energies=False):
        out = np.zeros((self.count, timesteps)) if per_step_energies else None
        for k in range(timesteps):
            for r in range(self.count):
                b = self.betas[r] if self.betas is not None else (beta if np.ndim(beta) == 0 else beta[k])
                self.lat.sweep(self.st[r], self.seeds[r], self.t, b)
                if per_step_energies:
                    out[r, k] = self.lat.energy_mag(self.st[r])[0]
            self.t += 1
        return out

    def energies(self):
        return np.array([self.lat.energy_mag(s)[0] for s in self.st])

    def states(self, out=None):
        res = np.stack([self.lat.unpack(s) for s in self.st]).astype(np.bool_) if self.st else np.zeros((0, 0), bool)
        if out is None:
            return res
        out[...] = res
        return out


class OracleLatEngine:
    def __init__(self, W, H, jabs=1.0, jpos=0):
        self.lat = O.Lat(W, H, jabs, jpos)
        self.nvars = W * H

    def make_states(self, seeds, replica_range=None):
        lo, hi = replica_range if replica_range is not None else (0, len(seeds))
        return OracleLatStates(self.lat, seeds[lo:hi])


class OracleRjStates:
    """Oracle engine E (real-coupling packed spec) -- or, with eng.run = O.pk_run, engine D (bit-sliced packed spec) -- behind
    the States interface the tempering host logic uses."""

    def __init__(self, eng, seeds, lo, hi):
        self.eng, self.all_seeds, self.lo, self.hi = eng, np.asarray(seeds, dtype=np.uint64), lo, hi
        self.t = 0
        self.betas = None
        self.st = None          # uint8[32 G, nvars] of ALL replicas (the oracle simulates whole groups)

    @property
    def count(self):
        return self.hi - self.lo

    def set_betas(self, betas):
        self.betas = None if betas is None else np.asarray(betas, dtype=np.float64)

    def do_time_steps(self, timesteps, beta=None, per_step_energies=False):
        e = self.eng
        R = len(self.all_seeds)
        kw = {}
        if self.betas is not None:
            full = np.zeros(R)
            full[self.lo:self.hi] = self.betas
            kw["beta_replica"] = full
        else:
            kw["betas"] = [beta] * timesteps if np.ndim(beta) == 0 else beta
        out = e.run(e.ea, e.eb, e.ej, e.nvars, self.all_seeds, timesteps, states=self.st, t0=self.t,
                    per_step=per_step_energies, **kw, **e.extra)
        self.st = out[1]
        self.t += timesteps
        return out[2][self.lo:self.hi] if per_step_energies else None

    def energies(self):
        e = self.eng
        if self.st is None:
            self.do_time_steps(0, 0.0)
        return e.run(e.ea, e.eb, e.ej, e.nvars, self.all_seeds, 0, betas=[], states=self.st, t0=self.t, **e.extra)[0][self.lo:self.hi]

    def states(self, out=None):
        res = self.st[self.lo:self.hi].astype(np.bool_)
        if out is None:
            return res
        out[...] = res
        return out


class OracleRjEngine:
    def __init__(self, ea, eb, ej, nvars, biases=None, bit_sliced=False):
        self.ea, self.eb, self.ej, self.nvars, self.biases = ea, eb, ej, nvars, biases
        self.run = O.pk_run if bit_sliced else O.rj_run
        self.extra = {} if bit_sliced else {"biases": biases}

    def make_states(self, seeds, replica_range=None):
        lo, hi = replica_range if replica_range is not None else (0, len(seeds))
        return OracleRjStates(self, seeds, lo, hi)
