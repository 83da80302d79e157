"""Classical parallel tempering on a beta ladder, with the API shape of the reference's
LatticeTempering (src/tempering.rs:29-299; quantum SSE there -- the classical ladder is this build's
own): add_graph(beta), timesteps(t), timesteps_sample(timesteps, replica_swap_freq,
sampling_freq) -> (bool[G,S,N] states, float64[G] time-averaged energies), get_total_swaps().

Replica slots are sharded over the ranks of a torch.distributed group (one process per GPU).  The swap
step exchanges TEMPERATURES, never configurations: every rank all-gathers one float64 energy per slot
(RCCL over xGMI with backend nccl), evaluates the same Philox-keyed decisions in libisingmc's host code
(isingmc_host_pt_swap_round) and re-labels its own slots' betas.  No spin ever crosses a link.
"""
import numpy as np

from . import _capi
from . import distributed as D


class HipEngine:
    """Default engine: the HIP library through the C ABI.  (Tests inject a CPU-oracle engine with the
    same two methods to exercise the sharding logic without a GPU; the product never does.)"""

    def __init__(self, ea, eb, ej, nvars, device=0):
        self.graph = _capi.Graph(ea, eb, ej, nvars=nvars, device=device)
        self.nvars = self.graph.nvars
        # sweeps, measurement, exchange decisions and beta relabelling all on the engine's HIP stream: periodic field-free
        # lattices, and graphs on the replica-packed paths (whether a container is packed depends on the graph and its
        # size: pt_attach refuses otherwise and the ladder falls back to the host swap step)
        self.supports_on_stream_pt = ((self.graph.kind == _capi.KIND_LATTICE2D and self.graph.info.fast_path == 0) or
                                      self.graph.kind == _capi.KIND_GENERAL)

    def make_states(self, seeds, replica_range=None):
        """seeds of ALL slots + this rank's [lo, hi): group membership on the replica-packed path follows the global
        slot index, so the ladder's trajectories do not depend on the number of ranks."""
        return _capi.States(self.graph, seeds, replica_range=replica_range)


def _split(edges):
    if isinstance(edges, tuple) and len(edges) == 3 and hasattr(edges[0], "__len__") and not isinstance(edges[0], tuple):
        ea, eb, ej = edges
    else:
        ea = [e[0][0] for e in edges]
        eb = [e[0][1] for e in edges]
        ej = [e[1] for e in edges]
    ea = np.ascontiguousarray(ea, dtype=np.uint64)
    eb = np.ascontiguousarray(eb, dtype=np.uint64)
    ej = np.ascontiguousarray(ej, dtype=np.float64)
    if ea.size == 0:
        raise ValueError("Must supply some edges for graph")
    return ea, eb, ej


class ClassicalTempering:
    def __init__(self, edges, seed=None, *, group=None, device=None, engine_factory=None):
        self._ea, self._eb, self._ej = _split(edges)
        self.nvars = int(max(self._ea.max(), self._eb.max())) + 1  # tempering.rs:44-49
        self._group = group
        self._world, self._rank = D.world_rank(group)
        if seed is None:
            if self._world > 1:
                raise ValueError("a seed is required when the ladder is sharded over several ranks")
            seed = int(_capi.make_seeds(None, 1)[0])
        self._seed = int(seed)
        self._betas = []
        self._slot_seeds = []
        self._n_drawn = 0
        self._device = device
        self._engine_factory = engine_factory or (lambda: HipEngine(self._ea, self._eb, self._ej, self.nvars,
                                                                    device=self._default_device()))
        self._states = None
        self._perm = None       # rung -> slot
        self._round = 0
        self._total_swaps = 0
        self._on_stream = False

    def _default_device(self):
        import os
        if self._device is not None:
            return self._device
        return int(os.environ.get("ISINGMC_DEVICE", os.environ.get("LOCAL_RANK", "0")))

    # -- tempering.rs:70-113 -------------------------------------------------------------------
    def add_graph(self, beta, seed=None):
        """Add a rung at inverse temperature `beta` (betas must be added in ladder order: the swap
        step pairs neighbouring rungs)."""
        if self._states is not None:
            raise ValueError("add every rung before the first timestep")
        if not np.isfinite(beta):
            raise ValueError("beta must be finite")
        if seed is None:  # seed.unwrap_or_else(|| self.tempering.rng_mut().gen())
            self._n_drawn += 1
            seed = int(_capi.make_seeds(self._seed, self._n_drawn)[-1])
        self._betas.append(float(beta))
        self._slot_seeds.append(int(seed))

    def get_num_graphs(self):
        return len(self._betas)

    def get_total_swaps(self):  # tempering.rs:297-299
        if self._on_stream:
            return int(self._states.pt_state()[2]) if self._hi > self._lo else self._total_swaps
        return self._total_swaps

    def get_betas(self):
        return np.array(self._betas)

    def get_permutation(self):
        """rung -> replica slot currently holding that temperature."""
        self._materialise()
        if self._on_stream and self._hi > self._lo:
            self._perm = self._states.pt_state()[0]
        return self._perm.copy()

    # -------------------------------------------------------------------------------------------
    def _materialise(self):
        if self._states is not None:
            return
        G = len(self._betas)
        if G == 0:
            raise ValueError("no graphs: call add_graph(beta) first")
        self._per = D.block_size(G, self._world)
        self._lo, self._hi = D.shard_bounds(G, self._world, self._rank)
        self._engine = self._engine_factory()
        self._states = self._engine.make_states(np.array(self._slot_seeds, dtype=np.uint64), (self._lo, self._hi))
        self._perm = np.arange(G, dtype=np.uint32)
        import os
        self._on_stream = (bool(getattr(self._engine, "supports_on_stream_pt", False)) and self._hi > self._lo and
                           os.environ.get("ISINGMC_PT_HOST", "0") in ("", "0"))  # ISINGMC_PT_HOST=1: the host swap step (A/B runs)
        # eligibility first, WITHOUT side effects (e.g. a real-coupling graph too small for the packed kernels, a bit-sliced shard
        # that cuts a replica group: the host swap step serves those) ...
        if self._on_stream:
            self._on_stream = self._states.pt_can_attach(G, self._lo, self._per, self._world)
        if self._world > 1:  # ... then every rank must take the same path (the collective differs) ...
            flags = D.all_gather_f64(np.array([float(self._on_stream)]), 1, self._group)  # (a rank without slots has no buffers to gather into: host path for all)
            self._on_stream = bool(flags.min() > 0)
        if self._on_stream:  # ... and only then does anyone attach
            self._states.pt_attach(self._betas, self._lo, self._per, self._world, self._seed)
        if self._on_stream:
            self._pt_local, self._pt_all = self._states.pt_buffers()
            self._pt_stream = self._states.pt_stream() if self._world > 1 else None
        else:
            self._push_betas()

    def _push_betas(self):
        G = len(self._betas)
        beta_of_slot = np.empty(G, dtype=np.float64)
        beta_of_slot[self._perm] = self._betas
        if self._hi > self._lo:
            self._states.set_betas(beta_of_slot[self._lo:self._hi])

    def _slot_energies(self, local):
        """float64[G] in slot order on every rank: the swap step's all-gather."""
        G = len(self._betas)
        gathered = D.all_gather_f64(local, self._per, self._group)
        out = np.empty(G, dtype=np.float64)
        for r in range(self._world):
            lo, hi = D.shard_bounds(G, self._world, r)
            out[lo:hi] = gathered[r * self._per:r * self._per + (hi - lo)]
        return out

    def _swap_step(self, need_perm=False):
        if self._on_stream:
            # measure -> [RCCL all-gather on the engine's stream] -> exchange kernel: nothing waits on the host
            self._states.pt_measure()
            if self._world > 1:
                import torch
                import torch.distributed as dist
                if dist.get_backend(self._group) == "nccl":
                    with torch.cuda.stream(self._pt_stream):
                        dist.all_gather_into_tensor(self._pt_all, self._pt_local, group=self._group)
                else:
                    # gloo (several ranks rehearsing on one GPU): the same exchange through the host -- wait for the
                    # measurement, gather on the CPU, put the result where the exchange kernel reads it
                    self._states.synchronize()
                    send = self._pt_local.cpu()
                    recv = torch.empty(self._world * self._per, dtype=torch.float64)
                    dist.all_gather_into_tensor(recv, send, group=self._group)
                    self._pt_all.copy_(recv)
                    torch.cuda.synchronize()
            self._states.pt_swap()
            if need_perm:
                self._perm = self._states.pt_state()[0]
            return
        local = self._states.energies() if self._hi > self._lo else np.zeros(0)
        slot_e = self._slot_energies(local)
        self._total_swaps += _capi.pt_swap_round(self._seed, self._round, self._betas, slot_e, self._perm)
        self._round += 1
        self._push_betas()

    def timesteps(self, t, replica_swap_freq=None):
        """tempering.rs:150-152 (parallel_timesteps): t sweeps on every rung.  replica_swap_freq (extension):
        an exchange round after every `replica_swap_freq` sweeps, as in the loop of tempering.rs:177-194."""
        self._materialise()
        if t <= 0:
            return
        if not replica_swap_freq:
            if self._hi > self._lo:
                self._states.do_time_steps(t)
            return
        if self._on_stream and self._world == 1 and self._hi > self._lo:
            # single rank: the whole loop is one library call (one persistent launch on mid-size lattices)
            self._states.pt_run(int(t), int(replica_swap_freq))
            self._states.synchronize()
            return
        done = 0
        while done < t:
            b = min(int(replica_swap_freq), t - done)
            if self._hi > self._lo:
                if self._on_stream:
                    self._states.pt_time_steps(b)
                else:
                    self._states.do_time_steps(b)
            done += b
            if b == replica_swap_freq:
                self._swap_step()
        if self._on_stream:
            self._states.synchronize()

    def timesteps_sample(self, timesteps, replica_swap_freq=None, sampling_freq=None):
        """tempering.rs:156-222: countdown scheduler of run / swap / sample.

        Returns (states, energies[, rungs]): energies float64[G] = time average per rung (identical on
        every rank).  Single process: states bool[G, S, N] indexed by rung.  Sharded: states
        bool[G_local, S, N] of this rank's slots plus rungs int[G_local, S] = the rung each local slot
        held at each sample (configurations never leave their GPU).
        """
        self._materialise()
        if self._on_stream and self._hi > self._lo:
            self._perm = self._states.pt_state()[0]  # the device owns the ladder state
        sampling_freq = 1 if sampling_freq is None else int(sampling_freq)
        replica_swap_freq = 1 if replica_swap_freq is None else int(replica_swap_freq)
        if sampling_freq <= 0:
            raise ValueError("sampling_freq must be positive")
        G, N = len(self._betas), self.nvars
        n_local = self._hi - self._lo
        S = timesteps // sampling_freq
        states = np.zeros((n_local, S, N), dtype=np.bool_)
        rungs = np.zeros((n_local, S), dtype=np.int64)
        energy_acc = np.zeros(G, dtype=np.float64)

        remaining, to_swap, to_sample, k = timesteps, replica_swap_freq, sampling_freq, 0
        while remaining > 0:
            t = min(to_sample, remaining) if replica_swap_freq <= 0 else min(to_sample, to_swap, remaining)
            local_sum = np.zeros(n_local)
            if n_local:
                local_sum = self._states.do_time_steps(t, per_step_energies=True).sum(axis=1)
            # energy_acc[rung] += te * t  (tempering.rs:185), te = mean energy over the block
            energy_acc += self._slot_energies(local_sum)[self._perm]
            to_sample -= t
            to_swap -= t
            remaining -= t
            if to_swap == 0 and replica_swap_freq > 0:
                self._swap_step(need_perm=True)
                to_swap = replica_swap_freq
            if to_sample == 0:
                if n_local and k < S:
                    self._states.states(out=states[:, k, :])
                    inv = np.empty(G, dtype=np.int64)
                    inv[self._perm] = np.arange(G)
                    rungs[:, k] = inv[self._lo:self._hi]
                k += 1
                to_sample = sampling_freq
        energies = energy_acc / max(timesteps, 1)
        if self._world == 1:
            by_rung = np.empty_like(states)
            for s in range(S):
                by_rung[rungs[:, s], s, :] = states[:, s, :]
            return by_rung, energies
        return states, energies, rungs
