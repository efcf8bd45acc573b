"""small helpers of the reference's utils module that the hot path's users rely on"""
import enum
import sys

import numpy as np

gamma_1H = 42.576 * 1e3  # kHz/T   (utils.py:8)
gamma_23Na = 11.262 * 1e3  # kHz/T


def Axes(*names):
    """IntEnum of named grid axes, e.g. Axes("FA", "T2") (utils.py:134-145)"""
    return enum.IntEnum("Axes", names, start=0)


def check_states(states):
    """EPG symmetry of a full [*, 2n+1, 3] array (utils.py:118-121)"""
    states = np.asarray(states)
    return np.allclose(states, states[..., ::-1, [1, 0, 2]].conj())


def get_norm(states):
    """utils.py:152-154"""
    states = np.asarray(states)
    return np.sqrt(np.sum(np.abs(states[..., 1:]) ** 2, axis=(-2, -1)))


def get_wavenumber(grad, duration, gamma=gamma_1H):
    """wavenumber (rad/m) of a gradient lobe: mT/m x ms (utils.py:157-169)"""
    return 2 * np.pi * gamma * np.asarray(grad) * 1e-3 * np.asarray(duration)


class Progress:
    """text progress display behind `simulate(disp=True)` (the reference wraps its operator loop in a progress
    bar, functions.py:175-176 / utils.py:219-236).  Here the unit of progress is a device launch: one per
    ADC-to-ADC segment in the per-timestep mode, one per operator batch in the stepwise mode, a single one for a
    state-resident run -- the bar then jumps from 0 to done when the kernel has finished."""

    def __init__(self, total, prefix="Simulating: ", width=60, out=None):
        self.total, self.prefix, self.width = max(int(total), 1), prefix, width
        self.out = out if out is not None else sys.stdout
        self.done = 0
        self._show()

    def _show(self):
        fill = self.width * self.done // self.total
        print(f"{self.prefix}[{'#' * fill}{'.' * (self.width - fill)}] {self.done}/{self.total}", end="\r", file=self.out, flush=True)

    def step(self, n=1):
        self.done = min(self.total, self.done + n)
        self._show()

    def close(self):
        self.done = self.total
        self._show()
        print("", file=self.out, flush=True)
