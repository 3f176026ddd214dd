import ctypes as C, numpy as np, sys
sys.path.insert(0,'.')
from multiverse_amd import _lib
lib=_lib.load()
n=4096
out=np.zeros(n,dtype=np.int32)
rc=lib.mv_debug_xcc_map(0, n, out.ctypes.data_as(C.POINTER(C.c_int32)))
print(rc, out[:32].tolist())
print("match id%8:", float((out == (np.arange(n)%8)).mean()), "distinct", np.unique(out).tolist())
