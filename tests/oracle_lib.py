"""ctypes binding of the CPU oracle (oracle/libvits_oracle.so). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
(vits.cpp_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libvits_oracle.so")

MODE_REFERENCE, MODE_HF = 0, 1
ARITH_F32, ARITH_BF16, ARITH_F16 = 0, 1, 2
SCOPE_FLOW_VOCODER, SCOPE_ALL_CONVS = 0, 1
NOISE_REFERENCE, NOISE_COUNTER, NOISE_EXPLICIT = 0, 1, 2

TAPS = ["enc_out", "prior_mean", "prior_logvar", "log_duration", "durations", "noise_dur", "noise_prior", "z_p", "z_flow",
        "pre_tanh", "waveform"]


class Opts(C.Structure):
    _fields_ = [("mode", C.c_int32), ("noise_kind", C.c_int32), ("noise_seed", C.c_uint64), ("noise_dur", C.c_void_p),
                ("noise_prior", C.c_void_p), ("noise_prior_stride", C.c_int64), ("fixed_duration", C.c_int32),
                ("threads", C.c_int32), ("arith", C.c_int32), ("arith_scope", C.c_int32), ("ggml_tables", C.c_int32)]


class Conv1dDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("t", C.c_int32), ("t_stride", C.c_int32),
                ("k", C.c_int32), ("dilation", C.c_int32), ("pad_left", C.c_int32), ("pre_act", C.c_int32),
                ("pre_slope", C.c_float), ("post_act", C.c_int32), ("out_scale", C.c_float), ("arith", C.c_int32)]


class ConvT1dDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("t", C.c_int32), ("t_stride", C.c_int32),
                ("t_out_stride", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("crop", C.c_int32),
                ("pre_slope", C.c_float), ("arith", C.c_int32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    L.vo_last_error.restype = C.c_char_p
    L.vo_load.restype = vp
    L.vo_load.argtypes = [C.c_char_p, sz]
    L.vo_free.argtypes = [vp]
    L.vo_num_tensors.restype = i32
    L.vo_num_tensors.argtypes = [vp]
    L.vo_tensor.restype = i64
    L.vo_tensor.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    L.vo_config.restype = i64
    L.vo_config.argtypes = [vp, C.c_char_p, C.c_char_p, sz]
    L.vo_process_ids.restype = vp
    L.vo_process_ids.argtypes = [vp, vp, i32, C.POINTER(Opts)]
    L.vo_log_durations.restype = i32
    L.vo_log_durations.argtypes = [vp, vp, i32, C.POINTER(Opts), vp, vp]
    L.vo_run_tap.restype = i64
    L.vo_run_tap.argtypes = [vp, C.c_char_p, vp, sz]
    L.vo_run_free.argtypes = [vp]
    L.vo_tokenize.restype = i64
    L.vo_tokenize.argtypes = [vp, C.c_char_p, vp, sz]
    L.vo_reference_noise_seed.argtypes = [C.c_uint32]
    L.vo_reference_noise_draw.argtypes = [vp, sz]
    L.vo_conv1d.restype = i32
    L.vo_conv1d.argtypes = [C.POINTER(Conv1dDesc), vp, vp, vp, vp, vp, vp, vp, i32]
    L.vo_conv_transpose1d.restype = i32
    L.vo_conv_transpose1d.argtypes = [C.POINTER(ConvT1dDesc), vp, vp, vp, vp, vp]
    L.vo_rel_attention.restype = i32
    L.vo_rel_attention.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.vo_add_layer_norm.restype = i32
    L.vo_add_layer_norm.argtypes = [i32, i32, i32, i32, C.c_float, vp, vp, vp, vp, vp]
    L.vo_max.restype = C.c_float
    L.vo_max.argtypes = [vp, i64]
    L.vo_masked_get_compact.restype = i64
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class OracleError(RuntimeError):
    pass


class Model:
    def __init__(self, data):
        self._h = lib().vo_load(data, len(data))
        if not self._h:
            raise OracleError(lib().vo_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().vo_free(self._h)
            self._h = None

    __del__ = close

    def num_tensors(self):
        return lib().vo_num_tensors(self._h)

    def tensor(self, name):
        dt, rk, dims = C.c_int32(), C.c_int32(), (C.c_int64 * 4)()
        n = lib().vo_tensor(self._h, name.encode(), None, 0, C.byref(dt), C.byref(rk), dims)
        if n < 0:
            raise KeyError(name)
        out = np.zeros(n, np.float32)
        lib().vo_tensor(self._h, name.encode(), _ptr(out), n, None, None, None)
        ne = [dims[i] for i in range(rk.value)]
        return out.reshape(ne[::-1]), dt.value  # torch shape

    def config(self, key):
        buf = C.create_string_buffer(1024)
        n = lib().vo_config(self._h, key.encode(), buf, 1024)
        return None if n < 0 else buf.value.decode()

    def tokenize(self, text):
        buf = np.zeros(4 * len(text.encode()) + 8, np.int32)
        n = lib().vo_tokenize(self._h, text.encode(), _ptr(buf), buf.size)
        return buf[:n].copy()

    def process_ids(self, ids, mode=MODE_REFERENCE, noise_kind=NOISE_COUNTER, noise_seed=4321, noise_dur=None,
                    noise_prior=None, fixed_duration=0, threads=0, taps=TAPS, arith=0, arith_scope=SCOPE_FLOW_VOCODER, ggml_tables=False):
        """Runs the full restated graph for ONE utterance. Returns {tap: np.ndarray} (flat [C*len] arrays reshaped
        to [C, len] where C is known)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        nd, npr = _f32(noise_dur), _f32(noise_prior)
        o = Opts(mode, noise_kind, noise_seed, _ptr(nd), _ptr(npr), 0 if npr is None else npr.shape[-1], fixed_duration,
                 threads, arith, arith_scope, int(ggml_tables))
        r = lib().vo_process_ids(self._h, _ptr(ids), ids.size, C.byref(o))
        if not r:
            raise OracleError(lib().vo_last_error().decode())
        try:
            out = {}
            for name in taps:
                n = lib().vo_run_tap(r, name.encode(), None, 0)
                a = np.zeros(n, np.float32)
                lib().vo_run_tap(r, name.encode(), _ptr(a), n)
                out[name] = a
            return out
        finally:
            lib().vo_run_free(r)


def _log_durations(self, ids, mode=MODE_REFERENCE, noise_kind=NOISE_COUNTER, noise_seed=4321, threads=0, arith=0, arith_scope=SCOPE_FLOW_VOCODER, noise_dur=None,
                   ggml_tables=False):
    """Stage one only (text encoder + duration predictor): (log_duration [T], durations [T]) of one utterance."""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    nd = _f32(noise_dur)
    o = Opts(mode, noise_kind, noise_seed, _ptr(nd), None, 0, 0, threads, arith, arith_scope, int(ggml_tables))
    logw, dur = np.zeros(ids.size, np.float32), np.zeros(ids.size, np.float32)
    if lib().vo_log_durations(self._h, _ptr(ids), ids.size, C.byref(o), _ptr(logw), _ptr(dur)) != 0:
        raise OracleError(lib().vo_last_error().decode())
    return logw, dur


Model.log_durations = _log_durations


def outside_latents():
    """latents outside the spline interval in the last process_ids / log_durations call of this thread (vo_outside_latents)"""
    f = lib().vo_outside_latents
    f.restype = C.c_int64
    return int(f())


def reference_noise(n, seed=None):
    if seed is not None:
        lib().vo_reference_noise_seed(seed)
    a = np.zeros(n, np.float32)
    lib().vo_reference_noise_draw(_ptr(a), n)
    return a


def conv1d(x, w, bias=None, dilation=1, pad_left=None, pre_slope=None, post_act=0, residual=None, accum=None, out_scale=1.0,
           lens=None, threads=0, arith=0):
    x, w = _f32(x), _f32(w)
    B, cin, T = x.shape
    cout, _, k = w.shape
    d = Conv1dDesc(B, cin, cout, T, T, k, dilation, (k - 1) * dilation // 2 if pad_left is None else pad_left,
                   0 if pre_slope is None else 1, 0.0 if pre_slope is None else pre_slope, post_act, out_scale, arith)
    cy = cout // 2 if post_act == 2 else cout
    y = np.zeros((B, cy, T), np.float32)
    bias, residual, accum = _f32(bias), _f32(residual), _f32(accum)
    lens = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    rc = lib().vo_conv1d(C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(residual), _ptr(accum), _ptr(lens), _ptr(y), threads)
    assert rc == 0
    return y


def conv_transpose1d(x, w, bias, stride, crop, pre_slope=1.0, lens=None, arith=0):
    x, w, bias = _f32(x), _f32(w), _f32(bias)
    B, cin, T = x.shape
    _, cout, k = w.shape
    To = stride * T + k - stride - 2 * crop
    d = ConvT1dDesc(B, cin, cout, T, T, To, k, stride, crop, pre_slope, arith)
    y = np.zeros((B, cout, To), np.float32)
    lens = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    assert lib().vo_conv_transpose1d(C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(lens), _ptr(y)) == 0
    return y


def rel_attention(q, k, v, rel_k, rel_v, heads, window, lens=None):
    q, k, v, rel_k, rel_v = map(_f32, (q, k, v, rel_k, rel_v))
    B, HD, T = q.shape
    out = np.zeros_like(q)
    lens = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    assert lib().vo_rel_attention(B, heads, HD // heads, T, T, window, _ptr(q), _ptr(k), _ptr(v), _ptr(rel_k), _ptr(rel_v),
                                  _ptr(lens), _ptr(out)) == 0
    return out


def add_layer_norm(x, residual, gamma, beta, eps=1e-5):
    x, residual, gamma, beta = map(_f32, (x, residual, gamma, beta))
    B, Cc, T = x.shape
    y = np.zeros_like(x)
    assert lib().vo_add_layer_norm(B, Cc, T, T, eps, _ptr(x), _ptr(residual), _ptr(gamma), _ptr(beta), _ptr(y)) == 0
    return y
