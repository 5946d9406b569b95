"""ctypes driver for oracle/rbpf_oracle_c.c (test / baseline infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "librbpf_oracle_c.so")


def build(native_dir=None):
    """make -C oracle (portable flags); with native_dir, a -march=native build for timing."""
    if native_dir:
        os.makedirs(native_dir, exist_ok=True)
        out = os.path.join(native_dir, "librbpf_oracle_c_native.so")
        cmd = ["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=c11", "-shared", "-o", out,
               os.path.join(ORACLE_DIR, "rbpf_oracle_c.c"), "-lm"]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        return out
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "rbpf_oracle_c.c")):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB


def build_arbiter():
    """The extended-precision build (-DRBPF_ORACLE_LONG_DOUBLE, x87 long double: see the header of oracle/rbpf_oracle_c.c)."""
    out = os.path.join(ORACLE_DIR, "_build", "librbpf_oracle_c_ld.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(os.path.join(ORACLE_DIR, "rbpf_oracle_c.c")):
        subprocess.run(["make", "-C", ORACLE_DIR, "arbiter"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return out


def particle_filter(rbpf, model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, dt, rng, n_threads=0,
                    want_full=True, lib_path=None):
    """Runs the C restatement with the product package's marshalling (same structs as include/rbpf.h).
    Returns (dict of outputs, loop_seconds)."""
    import importlib
    host = importlib.import_module(rbpf.__name__ + ".host")
    ffi = importlib.import_module(rbpf.__name__ + "._ffi")
    lib = C.CDLL(lib_path or build())
    prob = host._Problem(model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, dt)
    blk, keep = host._rng_block(rng, prob.N_P, prob.N_T, model.nw, 1)
    opt = ffi.rbpf_options(keep_history=1 if want_full else 0, trace=1, fix_p_mean=0, reserved=0, jitter=0.0)
    nN, n, N, T = model.nNonLin, model.nLin, prob.N_P, prob.N_T
    o = ffi.rbpf_filter_out()
    b = dict(traj_max=np.empty((nN, T), order="F"), traj_mean=np.empty((nN, T), order="F"), xl_max=np.empty(n),
             xl_mean=np.empty(n), iw_max=np.zeros(1, dtype=np.int32))
    if want_full:
        b.update(P_max=np.empty((n, n), order="F"), P_mean=np.empty((n, n), order="F"),
                 traj_sample_iwmax=np.empty((nN, T), order="F"), xn_traj=np.empty((nN, N, T), order="F"),
                 trace_logw=np.empty((N, T), order="F"), trace_w=np.empty((N, T), order="F"),
                 trace_ai=np.zeros((N, T), dtype=np.int32, order="F"), final_xn=np.empty((nN, N), order="F"),
                 final_xl=np.empty((n, N), order="F"), final_P=np.empty((n, n, N), order="F"))
    for k, v in b.items():
        setattr(o, k, v.ctypes.data_as(ffi.c_int32_p if v.dtype == np.int32 else ffi.c_double_p))
    secs = C.c_double(0.0)
    mdesc = model.descriptor()
    lib.rbpf_oracle_particle_filter.argtypes = [C.POINTER(ffi.rbpf_model), C.POINTER(ffi.rbpf_problem),
                                                C.POINTER(ffi.rbpf_rng), C.POINTER(ffi.rbpf_options),
                                                C.POINTER(ffi.rbpf_filter_out), C.c_int, C.POINTER(C.c_double)]
    st = lib.rbpf_oracle_particle_filter(C.byref(mdesc), C.byref(prob.c), C.byref(blk), C.byref(opt), C.byref(o),
                                         int(n_threads), C.byref(secs))
    if st != 0:
        raise RuntimeError(f"C oracle failed with status {st}")
    return b, secs.value


def max_threads(lib_path=None):
    lib = C.CDLL(lib_path or build())
    return int(lib.rbpf_oracle_max_threads())


def particle_smoother(rbpf, model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, N_K, dt, rng, info_form, use_dyn_res_norm=True,
                      n_threads=0, lib_path=None):
    """The C restatement of src/particleSmoother.m / src/particleSmootherInformationForm.m (dense families, replayed
    randomness).  Returns (dict XNK, XLK, PK, w, ai, paNt, ak in the layout of the numpy oracle's traces, seconds)."""
    import importlib
    host = importlib.import_module(rbpf.__name__ + ".host")
    ffi = importlib.import_module(rbpf.__name__ + "._ffi")
    lib = C.CDLL(lib_path or build())
    prob = host._Problem(model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, dt)
    blk, keep = host._rng_block(rng, prob.N_P, prob.N_T, model.nw, N_K)
    opt = ffi.rbpf_options(keep_history=1, trace=1, fix_p_mean=0, lazy_depth=0, jitter=0.0)
    nN, n, N, T = model.nNonLin, model.nLin, prob.N_P, prob.N_T
    o = ffi.rbpf_smoother_out()
    b = dict(XNK=np.empty((nN, T, N_K), order="F"), XLK=np.empty((n, N_K), order="F"), PK=np.empty((n, n, N_K), order="F"),
             trace_logw=np.empty((N, T, N_K), order="F"), trace_w=np.empty((N, T, N_K), order="F"),
             trace_ai=np.zeros((N, T, N_K), dtype=np.int32, order="F"), trace_paNt=np.full((N, T, N_K), np.nan, order="F"),
             trace_ak=np.zeros(N_K, dtype=np.int32))
    for k, v in b.items():
        setattr(o, k, v.ctypes.data_as(ffi.c_int32_p if v.dtype == np.int32 else ffi.c_double_p))
    secs = C.c_double(0.0)
    mdesc = model.descriptor(use_dyn_res_norm=use_dyn_res_norm)
    lib.rbpf_oracle_particle_smoother.argtypes = [C.POINTER(ffi.rbpf_model), C.POINTER(ffi.rbpf_problem), C.POINTER(ffi.rbpf_rng),
                                                  C.POINTER(ffi.rbpf_options), C.c_int, C.c_int, C.POINTER(ffi.rbpf_smoother_out),
                                                  C.c_int, C.POINTER(C.c_double)]
    st = lib.rbpf_oracle_particle_smoother(C.byref(mdesc), C.byref(prob.c), C.byref(blk), C.byref(opt), int(N_K), 1 if info_form else 0,
                                           C.byref(o), int(n_threads), C.byref(secs))
    if st != 0:
        raise RuntimeError(f"C oracle smoother failed with status {st}")
    tr = lambda a: np.transpose(a, (2, 1, 0)).copy()                    # noqa: E731  -> [N_K, T, N]
    return dict(XNK=b["XNK"], XLK=b["XLK"], PK=b["PK"], logw=tr(b["trace_logw"]), w=tr(b["trace_w"]), ai=tr(b["trace_ai"]),
                paNt=tr(b["trace_paNt"]), ak=b["trace_ak"]), secs.value
