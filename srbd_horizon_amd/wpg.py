"""Walking-pattern scheduler: re-creation of the behaviour of reference python/wpg.py (``steps_phase``), checked
tick by tick against fixtures produced by the reference's own file (tests/golden/wpg_*.npz).

Same constructor and ``set(action)`` call shape (wpg.py:4, :68).  The receding "shift the contact plan back by one
node" (wpg.py:74-77) is one slice move per parameter instead of a Python loop over nodes x contacts.
Deliberately reproduced quirks: the swing profile is ``0.1*sin(linspace(0, pi))`` -- numpy's default 50 samples, of
which only indices 1..8 are used, i.e. a rising ramp rather than a bump (wpg.py:28, :37).
"""
from __future__ import annotations

import numpy as np


def step_tables(c_init_z: float, step_duration=0.5, dt=0.05, ss_share=0.8, ds_share=0.2):
    """Left/right swing-height and contact-switch cycles (wpg.py:19-64): 2*step_nodes+1 entries each."""
    step_nodes = int(step_duration / dt)
    ss = int(ss_share * step_nodes)
    ds = int(ds_share * step_nodes)
    ramp = 0.1 * np.sin(np.linspace(0, np.pi))[1:ss + 1]       # 50-sample default, indices 1..ss
    z0 = float(c_init_z)
    flat = lambda n: np.full(n, z0)
    ones = lambda n: np.ones(n)
    l_cycle = np.concatenate([flat(ds), z0 + ramp, flat(ds), flat(ss), flat(1)])
    l_switch = np.concatenate([ones(ds), np.zeros(ss), ones(ds), ones(ss), ones(1)])
    r_cycle = np.concatenate([flat(ds), flat(ss), flat(ds), z0 + ramp, flat(1)])
    r_switch = np.concatenate([ones(ds), ones(ss), ones(ds), np.zeros(ss), ones(1)])
    return step_nodes, l_cycle, l_switch, r_cycle, r_switch


class steps_phase:
    def __init__(self, f, c, cdot, c_init_z, c_ref, w_ref, orientation_tracking_gain, cdot_switch, nodes,
                 number_of_legs, contact_model):
        self.f, self.c, self.cdot = f, c, cdot
        self.c_ref, self.cdot_switch = c_ref, cdot_switch
        self.w_ref, self.orientation_tracking_gain = w_ref, orientation_tracking_gain
        self.number_of_legs, self.contact_model = number_of_legs, contact_model
        self.nodes = nodes
        self.step_counter = 0
        self.step_duration, self.dt, self.ss_share, self.ds_share = 0.5, 0.05, 0.8, 0.2
        (self.step_nodes, self.l_cycle, self.l_cdot_switch,
         self.r_cycle, self.r_cdot_switch) = step_tables(c_init_z, self.step_duration, self.dt, self.ss_share, self.ds_share)
        self.action = ""

    def set(self, action):
        self.action = action
        ref_id = self.step_counter % (2 * self.step_nodes)
        nc = self.contact_model * self.number_of_legs
        last = self.nodes
        for i in range(nc):                                     # shift the contact plan back by one node
            for par in (self.cdot_switch[i], self.c_ref[i]):
                v = par.values
                v[:, :last] = v[:, 1:last + 1]
        self.w_ref.assign([0.0, 0.0, 0.0], nodes=last)
        if action == "step":
            self.orientation_tracking_gain.assign(1e2, nodes=last)
            for i in range(nc):
                left = i < self.contact_model
                self.cdot_switch[i].assign((self.l_cdot_switch if left else self.r_cdot_switch)[ref_id], nodes=last)
                self.c_ref[i].assign((self.l_cycle if left else self.r_cycle)[ref_id], nodes=last)
        elif action == "jump":
            self.orientation_tracking_gain.assign(0.0, nodes=last)
            for i in range(len(self.c)):
                self.cdot_switch[i].assign(0.0, nodes=last)
        else:                                                   # stance (the examples pass "standing")
            self.orientation_tracking_gain.assign(1e2, nodes=last)
            for i in range(len(self.c)):
                self.cdot_switch[i].assign(1.0, nodes=last)
                self.c_ref[i].assign(0.0, nodes=last)
        self.step_counter += 1
