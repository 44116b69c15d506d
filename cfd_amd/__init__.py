"""cfd_amd -- the finite-volume field update as fp64 HIP kernels for MI355X (gfx950).

cfd_amd.csrc/    HIP kernels and the C ABI (include/cfd_hip.h) -> libcfdhip.so
cfd_amd.solver   host-side mirror of the reference's step interface (ctypes over the C ABI)
cfd_amd.cases    the shipped initial condition in closed form and the synthetic benchmark grids
"""
from .solver import (OUTPUT_PROPERTIES, AdjGraph, adjGraphToGrid, applyDiffEq_adj, correctPrevTime, make_params, runSingleStep,  # noqa: F401
                     runTimeSteps_Print, totalDensity)
