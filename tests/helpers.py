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
