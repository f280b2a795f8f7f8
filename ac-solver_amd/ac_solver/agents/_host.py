"""ctypes binding of libacx_trainer.so (include/acx_trainer.h): host utilities of the PPO trainer -- NumPy's legacy shuffle and
CPython's random.Random restated so that they run off the interpreter lock beside a rollout.  Not part of libacx.so: no device work."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "lib", "libacx_trainer.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: build it first (python __graft_entry__.py, or make -C ac-solver_amd/csrc)")
lib = C.CDLL(LIB_PATH)
_i64p, _i32p, _u8p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
SIGNATURES = {
    "acxt_np_shuffle_epochs": (C.c_int, [C.c_uint32, C.c_int64, C.c_int, _i64p]),
    "acxt_py_curriculum_draws": (C.c_int, [C.POINTER(C.c_uint32), _i32p, C.c_int64, C.c_int64, C.c_int64, C.c_double, _u8p, _i64p]),
    "acxt_last_error": (C.c_char_p, []),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype, _fn.argtypes = _res, _args
E_INVAL = -1


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {(lib.acxt_last_error() or b'').decode()}")
