"""MI355X-native hot path of LeastSquaresOptim.jl: host mirror of the reference API over the C ABI
(include/lsqhip.h) of hand-written gfx950 HIP kernels.  Import as `lsq_amd` (root shim)."""
from . import _lib
from ._lib import (ArgumentError, DimensionMismatch, HipError, IsFiniteException, LsqError,
                   PeerAborted, PosDefException, RankDeficientException, build, declared_symbols, lib)
from .api import (axpy_, box_clip_, clamp_, copyto_, ediv_, fill_, first_nonfinite, rmul_, vsum,
                  AllocatedSolver, Cholesky, Context, PinnedBuffer, DeviceMatrix, DeviceOperator, DeviceVector, Dogleg, LSMR,
                  LeastSquaresProblem, LeastSquaresProblemAllocated, LeastSquaresResult, LevenbergMarquardt, OptimizationState, QR,
                  colsumabs2_, rowsumabs2_, converged, default_context, default_optimizer, default_solver,
                  maxabs, maxabs_projected_gradient, mul_, norm, optimize, optimize_, set_exact, debug_set, debug_get, sumsq, wdot,
                  wnorm)
from . import loops, rowshard, sharding, synthetic
from .loops import optimize_operator_level

__all__ = [n for n in dir() if not n.startswith("_")]
