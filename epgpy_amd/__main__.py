"""`python -m epgpy_amd`: library / device information and a one-second self check (the README sequence
against its published values)."""
import sys
import time

import numpy as np

from . import epg, _lib, __version__


def main():
    print(f"epgpy_amd {__version__}   library: {_lib.library_path()}")
    ctx = _lib.get_context(None)
    info = ctx.info()
    print(f"device: {info['name']} ({info['arch']}), {info['compute_units']} CUs, wavefront {info['wavefront_size']}, "
          f"{info['hbm_bytes'] / 2 ** 30:.0f} GiB, ABI v{ctx.lib.epgx_abi_version()}")
    seq = [epg.T(90, 90)] + [[epg.S(1, duration=5), epg.E(5, 150, [30, 40, 50]), epg.T(120, 0),
                              epg.S(1, duration=5), epg.E(5, 150, [30, 40, 50]), epg.ADC]] * 20
    t0 = time.perf_counter()
    sig = epg.simulate(seq)
    dt = time.perf_counter() - t0
    expect = np.array([0.537398482930342, 0.5841005873035536, 0.6140480648084863])   # epgpy README, first echo
    err = float(np.max(np.abs(np.abs(sig[0]) - expect)))
    print(f"README multi-spin-echo: {sig.shape} in {1e3 * dt:.1f} ms, |first echo - reference| = {err:.1e}")
    return 0 if err < 1e-12 else 1


if __name__ == "__main__":
    sys.exit(main())
