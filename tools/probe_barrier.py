import ctypes, sys
sys.path.insert(0,'.'); import _pkg; _pkg.load_package()
from minigpt4_cpp_amd import minigpt4_library as ML
lib=ML.load_library(); L=lib.library
L.minigpt4_amd_probe_grid_barrier.restype=ctypes.c_float
L.minigpt4_amd_probe_grid_barrier.argtypes=[ctypes.c_int,ctypes.c_int,ctypes.POINTER(ctypes.c_uint)]
for nb in (32,64,128,256):
    e=ctypes.c_uint(0)
    us=L.minigpt4_amd_probe_grid_barrier(nb,2000,ctypes.byref(e))
    print("blocks",nb,"barrier_us",round(us,3),"errors",e.value, flush=True)
