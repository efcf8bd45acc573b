"""one-rank RCCL communicator through libepgx (epgx_comm_*), with and without torch in the process"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch" in sys.argv:
    import torch  # noqa: F401
import numpy as np
from epgpy_amd import _lib
ctx = _lib.get_context(0)
print("ctx ok", flush=True)
comm = _lib.Comm(ctx, 0, 1, lambda raw: raw)
print("comm ok", flush=True)
src, dst = _lib.DeviceBuffer(ctx, 4096), _lib.DeviceBuffer(ctx, 4096)
data = np.arange(512, dtype=np.float64)
src.upload(data)
comm.gather(src.ptr.value, dst.ptr.value, 4096, 0)
print("gather ok", np.array_equal(dst.download(np.float64, (512,)), data), flush=True)
comm.destroy()
