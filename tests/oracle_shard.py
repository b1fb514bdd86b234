"""CPU stand-in for one rank's block of a split temperature ladder, built on the oracle.
It has the interface procgen_amd.distributed.SplitLadder drives (run_to_swap_point,
local_energies, swap_pass), so the multi-rank control flow -- swap schedule, energy
all-gather, replicated label permutation, slot-addressed draws -- can be tested with gloo on
CPUs against a single-process oracle run of the whole ladder."""
import numpy as np
import torch

import oracle_binding as ob


class OracleShard:
    def __init__(self, pos, edges, ecount, kind, value, ladder_temps, slot_base, swap_freq, key, group_base=0,
                 cfg=None):
        self.pos = np.array(pos, np.float32, copy=True)          # [T_local, N, 3], replica order
        self.edges = np.array(edges, np.int32, copy=True)
        self.ecount = np.array(ecount, np.int32, copy=True)
        self.kind, self.value = kind, value
        self.ladder = np.asarray(ladder_temps, np.float64)
        self.local_count, self.ladder_count = self.pos.shape[0], len(self.ladder)
        self.slot_base, self.f, self.key, self.group_base = slot_base, swap_freq, key, group_base
        self.cfg = cfg if cfg is not None else ob.default_config(ob.SUM_CANONICAL, ob.ALGO_OPTIMISED)
        self.slot_of_rep = np.arange(self.ladder_count)
        self.rep_of_slot = np.arange(self.ladder_count)
        self.step = 0
        self.cost = np.array([ob.cost(self.pos[t], self.edges[t, :self.ecount[t]], kind, value, self.cfg).total
                              for t in range(self.local_count)])
        self.swap_log = []

    def run_to_swap_point(self, max_steps):
        s, f = self.step, self.f
        nxt = ((s + f - 1) // f) * f
        last = min(s + max_steps - 1, nxt)
        n = last - s + 1
        if n <= 0:
            return 0, False
        slots = self.slot_of_rep[self.slot_base:self.slot_base + self.local_count]
        res = ob.run(self.pos, self.edges, self.ecount, self.kind, self.value, self.ladder[slots], n, 0, self.cfg,
                     key=self.key, stream_ids=[int(self.group_base * self.ladder_count + q) for q in slots],
                     first_step=s, trace=False)
        assert res["err"] == 0
        self.pos, self.edges, self.ecount, self.cost = res["pos"], res["edges"], res["ecount"], res["cost"]
        self.step = last + 1
        return n, last == nxt

    def local_energies(self, out):
        out.copy_(torch.from_numpy(self.cost))

    def swap_pass(self, all_energies):
        """attemptSwap over the whole ladder (reference src/ParallelTempering.Core.fs:112-119,168-175)."""
        en = all_energies.cpu().numpy()
        Tl = self.ladder_count
        p_idx = (self.step - 1) // self.f
        u = ob.philox_stream(self.key, self.group_base, 1, (p_idx + 1) * (Tl - 1))[p_idx * (Tl - 1):]
        for i in range(Tl - 1):
            r1, r2 = self.rep_of_slot[i], self.rep_of_slot[i + 1]
            p = min(1.0, ob.lib().og_swap_probability(en[r1], self.ladder[i], en[r2], self.ladder[i + 1]))
            sw = bool(u[i] < p)
            if sw:
                self.rep_of_slot[i], self.rep_of_slot[i + 1] = r2, r1
                self.slot_of_rep[r2], self.slot_of_rep[r1] = i, i + 1
            self.swap_log.append(sw)
