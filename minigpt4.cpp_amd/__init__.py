"""minigpt4.cpp_amd -- MI355X-native drop-in for the hot path of Maknee/minigpt4.cpp.

Holds only what the path needs:
  csrc/                HIP kernels (gfx950) + host engine + the C-ABI (`libminigpt4.so`, include/minigpt4.h)
  minigpt4_library.py  host-side mirror of the reference's ctypes binding (same class / method names)
  modelgen.py, quants.py  synthetic model files in the reference's two on-disk formats (tooling)

The directory name contains a dot, so it is imported through `__graft_entry__.load_package()`
(registers it as `minigpt4_cpp_amd`).
"""
