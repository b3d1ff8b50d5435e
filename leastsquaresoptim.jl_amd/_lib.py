"""ctypes binding of liblsqhip.so (the C ABI declared in include/lsqhip.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing, or no HIP
device is visible when a context is created, this raises -- it never silently computes elsewhere.
"""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
# LSQ_LIB_PATH: another build of the same library (same-box A/B measurements of kernel variants, tools/ab_bench.sh)
LIB_PATH = os.environ.get("LSQ_LIB_PATH") or os.path.join(_HERE, "liblsqhip.so")
HEADER = os.path.join(ROOT, "include", "lsqhip.h")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)

F_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
G_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
ALLREDUCE_CALLBACK = C.CFUNCTYPE(C.c_int, c_dp, C.c_int, C.c_void_p)
OP_MUL_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)   # (trans, d_x, d_out, user)
OP_COLSUM_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)                  # (d_out, user)
PRECOND_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)   # (d_P, J, d_damp, user)
ROW_ALLREDUCE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)   # (d_buf, count, hip_stream, user)
PRECOND_UPDATE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)              # (J, d_damp, user)
PRECOND_LDIV_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)                # (d_out, d_in, user)

OK, EDIM, ENOTPD, ERANK, ENONFINITE, EBOUNDS, EHIP, EARG, ECALLBACK, ERCCL = range(10)
QR, CHOLESKY, LSMR = 0, 1, 2
DOGLEG, LEVENBERG_MARQUARDT = 0, 1


class Options(C.Structure):
    _fields_ = [("x_tol", C.c_double), ("f_tol", C.c_double), ("g_tol", C.c_double),
                ("iterations", C.c_int), ("delta", C.c_double), ("h_lower", c_dp), ("h_upper", c_dp),
                ("allreduce", ALLREDUCE_CALLBACK), ("allreduce_user", C.c_void_p),
                ("trace_cap", C.c_int), ("trace_ssr", c_dp), ("trace_gnorm", c_dp),
                ("trace_delta", c_dp), ("trace_rho", c_dp), ("trace_inner", c_ip),
                ("trace_accept", c_ip), ("trace_x", c_dp),
                ("preconditioner", PRECOND_CALLBACK), ("preconditioner_user", C.c_void_p),
                ("row_allreduce", ROW_ALLREDUCE_CALLBACK), ("row_allreduce_user", C.c_void_p),
                ("global_rows", C.c_longlong),
                ("precond_update", PRECOND_UPDATE_CALLBACK), ("precond_ldiv", PRECOND_LDIV_CALLBACK),
                ("precond_general_user", C.c_void_p)]


class Result(C.Structure):
    _fields_ = [("optimizer", C.c_int), ("ssr", C.c_double), ("iterations", C.c_int),
                ("converged", C.c_int), ("x_converged", C.c_int), ("f_converged", C.c_int),
                ("g_converged", C.c_int), ("f_calls", C.c_int), ("g_calls", C.c_int),
                ("mul_calls", C.c_int), ("status", C.c_int), ("bad_index", C.c_int),
                ("seconds", C.c_double), ("lsmr_iterations", C.c_longlong), ("ssr0", C.c_double)]


def build(verbose=False):
    """Compile liblsqhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def declared_symbols():
    """Every function name include/lsqhip.h declares (used by the CPU-side export test)."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(lsq_[a-z0-9_]+)\s*\(", txt)
    skip = {"lsq_f_callback", "lsq_g_callback", "lsq_allreduce_callback", "lsq_device_allreduce_callback",
            "lsq_precond_update_callback", "lsq_precond_ldiv_callback"}
    return sorted(set(n for n in names if n not in skip))


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "liblsqhip.so is not built (%s). Run __graft_entry__.build() or `make -C "
            "leastsquaresoptim.jl_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    pvp = C.POINTER(C.c_void_p)
    sig = {
        "lsq_last_error": (C.c_char_p, []),
        "lsq_version": (i, []),
        "lsq_ctx_create": (i, [i, vp, pvp]),
        "lsq_ctx_destroy": (i, [vp]),
        "lsq_ctx_sync": (i, [vp]),
        "lsq_ctx_stream": (vp, [vp]),
        "lsq_malloc": (i, [vp, sz, pvp]),
        "lsq_free": (i, [vp, vp]),
        "lsq_h2d": (i, [vp, vp, vp, sz]),
        "lsq_d2h": (i, [vp, vp, vp, sz]),
        "lsq_d2d": (i, [vp, vp, vp, sz]),
        "lsq_dense_create": (i, [vp, i, i, pvp]),
        "lsq_csc_create": (i, [vp, i, i, c_ip, c_ip, pvp]),
        "lsq_mat_destroy": (i, [vp]),
        "lsq_mat_size": (i, [vp, c_ip, c_ip, C.POINTER(C.c_longlong)]),
        "lsq_mat_set_values": (i, [vp, c_dp]),
        "lsq_mat_get_values": (i, [vp, c_dp]),
        "lsq_mat_set_values_async": (i, [vp, c_dp]),
        "lsq_mat_upload_wait": (i, [vp]),
        "lsq_host_alloc": (i, [vp, sz, pvp]),
        "lsq_host_free": (i, [vp, vp]),
        "lsq_dot": (i, [vp, i, vp, vp, c_dp]),
        "lsq_emul": (i, [vp, i, vp, vp, vp]),
        "lsq_mat_values": (vp, [vp]),
        "lsq_mat_refresh": (i, [vp]),
        "lsq_mat_set_colscale": (i, [vp, vp]),
        "lsq_mat_colscale_changed": (i, [vp]),
        "lsq_mul": (i, [vp, i, d, vp, d, vp]),
        "lsq_solver_set_preconditioner": (i, [vp, PRECOND_CALLBACK, vp]),
        "lsq_solver_set_general_preconditioner": (i, [vp, PRECOND_UPDATE_CALLBACK, PRECOND_LDIV_CALLBACK, vp]),
        "lsq_op_create": (i, [vp, i, i, OP_MUL_CALLBACK, OP_COLSUM_CALLBACK, vp, C.POINTER(vp)]),
        "lsq_colsumabs2": (i, [vp, vp]),
        "lsq_rowsumabs2": (i, [vp, vp]),
        "lsq_axpy": (i, [vp, i, d, vp, vp]),
        "lsq_scal": (i, [vp, i, d, vp]),
        "lsq_copy": (i, [vp, i, vp, vp]),
        "lsq_fill": (i, [vp, i, d, vp]),
        "lsq_sumsq": (i, [vp, i, vp, c_dp]),
        "lsq_sum": (i, [vp, i, vp, c_dp]),
        "lsq_nrm2": (i, [vp, i, vp, c_dp]),
        "lsq_wdot": (i, [vp, i, vp, vp, vp, c_dp]),
        "lsq_amax": (i, [vp, i, vp, c_dp]),
        "lsq_amax_projected": (i, [vp, i, vp, vp, vp, vp, c_dp]),
        "lsq_clamp": (i, [vp, i, d, d, vp]),
        "lsq_ediv": (i, [vp, i, vp, vp, vp]),
        "lsq_box_clip": (i, [vp, i, vp, vp, vp, vp]),
        "lsq_first_nonfinite": (i, [vp, i, vp, c_ip]),
        "lsq_solver_create": (i, [vp, vp, i, i, pvp]),
        "lsq_solver_destroy": (i, [vp]),
        "lsq_ldiv": (i, [vp, vp, vp, vp, c_ip]),
        "lsq_ldiv_damped": (i, [vp, vp, vp, vp, vp, c_ip]),
        "lsq_solver_set_row_allreduce": (i, [vp, ROW_ALLREDUCE_CALLBACK, vp, C.c_longlong]),
        "lsq_solver_info": (i, [vp, c_ip, c_ip, c_ip]),
        "lsq_solver_qr_path": (i, [vp, c_ip]),
        "lsq_solver_qr_panel": (i, [vp, c_ip]),
        "lsq_solver_chol_path": (i, [vp, c_ip]),
        "lsq_solver_stats": (i, [vp, c_ip, c_ip]),
        "lsq_ctx_fallback_stats": (i, [vp, c_ip]),
        "lsq_ctx_device_info": (i, [vp, c_ip, c_ip, C.c_char_p, i]),
        "lsq_ctx_tail_stats": (i, [vp, C.POINTER(C.c_longlong)]),
        "lsq_bench_occupy": (i, [vp, i, i, d]),
        "lsq_bench_occupy_wait": (i, [vp]),
        "lsq_options_default": (None, [C.POINTER(Options)]),
        "lsq_optimize": (i, [vp, i, i, vp, vp, vp, F_CALLBACK, G_CALLBACK, vp, C.POINTER(Options),
                             C.POINTER(Result)]),
        "lsq_model_tanh_create": (i, [vp, vp, c_dp, c_dp, pvp]),
        "lsq_model_destroy": (i, [vp]),
        "lsq_model_f": (F_CALLBACK, []),
        "lsq_model_g": (G_CALLBACK, []),
        "lsq_synth_sparse": (i, [i, i, i, C.c_ulonglong, c_ip, c_ip, c_dp]),
        "lsq_synth_dense": (i, [i, i, C.c_ulonglong, c_dp]),
        "lsq_synth_uniform": (i, [i, C.c_ulonglong, d, d, c_dp]),
        "lsq_synth_normal": (i, [i, C.c_ulonglong, c_dp]),
        "lsq_set_exact": (i, [i]),
        "lsq_debug_set": (i, [i, i]),
        "lsq_debug_get": (i, [c_ip, c_ip, C.POINTER(C.c_longlong)]),
        "lsq_prof_begin": (i, [vp, i]),
        "lsq_prof_select": (i, [vp, i]),
        "lsq_prof_end": (i, [vp, c_dp, c_ip]),
        "lsq_prof_overhead": (i, [vp, i, c_dp]),
        "lsq_bench_mul": (i, [vp, i, i, vp, vp, d, C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._signatures = sig
    _LIB = L
    return L


class LsqError(RuntimeError):
    """Base of the exceptions mirroring the reference's (SURVEY 8b 'Errors')."""

    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


class DimensionMismatch(LsqError):
    pass


class PosDefException(LsqError):
    pass


class RankDeficientException(LsqError):
    pass


class IsFiniteException(LsqError):
    pass


class ArgumentError(LsqError):
    pass


class HipError(LsqError):
    pass


class PeerAborted(LsqError):
    """Sharded run: another rank left its loop with an error (LSQ_ERCCL)."""


_EXC = {EDIM: DimensionMismatch, ENOTPD: PosDefException, ERANK: RankDeficientException,
        ENONFINITE: IsFiniteException, EBOUNDS: ArgumentError, EHIP: HipError, EARG: ArgumentError,
        ECALLBACK: LsqError, ERCCL: PeerAborted}


def check(status):
    if status != OK:
        msg = lib().lsq_last_error().decode("utf-8", "replace")
        raise _EXC.get(status, LsqError)(status, msg or "lsq status %d" % status)
