"""Diffusion-weighted spin echo with ONE gradient vector per voxel class: six directions x three b-values along the first
grid axis, a T2 range along the second -- `S(k)` with a vectorised integer `k` (epgpy/shift.py:38-41) and a diffusion
tensor.  The coordinates of the phase states then differ from voxel to voxel while their row structure is shared; the
host plans ONE gather table for all voxels, the per-voxel b-values go into the `D` tables (epgpy_amd/kspace.py).

    python examples/dwi_directions.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
from epgpy_amd import epg  # noqa: E402

dirs = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1]])
amps = np.array([1, 2, 3])
k = (dirs[:, None, :] * amps[None, :, None]).reshape(-1, 1, 3)            # [18, 1, 3]: axis 0 = gradient, axis 1 = T2
T2 = np.linspace(40, 120, 5)[None, :]
Dt = np.array([[1.7, 0.1, 0.0], [0.1, 0.4, 0.0], [0.0, 0.0, 0.3]]) * 1e-3  # mm^2/s: a fibre along x
kvalue = 2.5e4                                                             # rad/m per unit of k

seq = [epg.T(90, 90), epg.S(k), epg.D(10, Dt), epg.E(10, 1000, T2), epg.T(180, 0),
       epg.S(k), epg.D(10, Dt), epg.E(10, 1000, T2), epg.ADC]
signal = np.abs(epg.simulate(seq, kvalue=kvalue))[0]                       # [18, 5]
s0 = np.exp(-20 / T2)                                                      # without diffusion weighting
print("direction  amplitude  attenuation (T2 = %.0f ms)" % T2[0, 2])
for row, (d, a) in enumerate((d, a) for d in dirs for a in amps):
    print(f"{d}   {a}          {signal[row, 2] / s0[0, 2]:.4f}")
