"""RCCL communicator created in a helper thread (as bench.py does under its watchdog), used from the main thread."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import _lib, sharding
uid = _lib.comm_unique_id()
box = {}
def init():
    box["comm"] = sharding.RcclComm(0, 0, 1, uid)
th = threading.Thread(target=init, daemon=True); th.start(); th.join(60)
assert not th.is_alive() and "comm" in box
comm = box["comm"]
x = np.arange(5, dtype=np.int64)
print("allreduce", comm.allreduce_sum(x), "allgather", comm.allgather(np.array([1.5, 2.5])).tolist())
comm.barrier(); comm.close(); print("ok")
