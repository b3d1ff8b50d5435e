"""The sanitizer canary of tools/asan_run.sh: a deliberately short colptr handed to lsq_csc_create (which reads colptr[n] in
instrumented host code) must end the process with an AddressSanitizer heap-buffer-overflow report.  Run as
    tools/asan_run.sh --canary
and expect a non-zero exit with 'heap-buffer-overflow' on stderr; exit 0 means the library under test is NOT instrumented."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.getcwd())
import lsq_amd as lsq   # the importable alias of leastsquaresoptim.jl_amd/ (tests/conftest.py)
L = lsq.lib()
print("library:", L._name, flush=True)
maps = open("/proc/self/maps").read()
print("asan runtime mapped:", "libasan" in maps or "clang_rt.asan" in maps, flush=True)
n = 1000
libc = ctypes.CDLL(None)
libc.malloc.restype = ctypes.c_void_p
p = libc.malloc(ctypes.c_size_t(4 * n))          # n ints, not n + 1
ctypes.memset(p, 0, 4 * n)
colptr = ctypes.cast(p, ctypes.POINTER(ctypes.c_int))
rowval = (ctypes.c_int * 1)()
ctx = ctypes.c_void_p()
rc = L.lsq_ctx_create(0, None, ctypes.byref(ctx))
assert rc == 0, rc
out = ctypes.c_void_p()
rc = L.lsq_csc_create(ctx, 10, n, colptr, rowval, ctypes.byref(out))
print("lsq_csc_create returned", rc, "-- the overflow was NOT caught", flush=True)
sys.exit(0)
