"""Host-side Python mirror of the C ABI in ``include/vits.h`` (ctypes; no compute happens here).

The product is ``csrc/libvits_hip.so`` (hand-written HIP kernels for gfx950 + a C++ host orchestrator). This
module only loads it and marshals plain pointers, mirroring the reference's operator interface
(``vits_model_load_from_*`` / ``vits_model_process`` / ``vits_free_*``, /root/reference/src/include/vits.h:87-102)
plus the extensions declared in ``include/vits.h``. There is NO fallback: if the shared library (or a GPU,
for the compute entry points) is missing, calls fail loudly.

The directory is named ``vits.cpp_amd`` (with a dot), so it is imported by path; see ``load_package()`` in
``tests/conftest.py`` / ``bench.py``:  ``importlib.util.spec_from_file_location("vits_cpp_amd", ".../__init__.py")``.
"""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITS_HIP_LIB", os.path.join(_HERE, "csrc", "libvits_hip.so"))

MODE_DEFAULT, MODE_REFERENCE, MODE_HF = -1, 0, 1
NOISE_REFERENCE, NOISE_COUNTER, NOISE_EXPLICIT = 0, 1, 2
SYNTH_FULL, SYNTH_TINY, SYNTH_BF16 = 0, 1, 0x100
ARITH_F32, ARITH_BF16, ARITH_F16, ARITH_F32_SPLIT = 0, 1, 2, 3
SCOPE_FLOW_VOCODER, SCOPE_ALL_CONVS = 0, 1

#: every symbol include/vits.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "vits_model_load_from_bytes", "vits_model_load_from_file", "vits_free_model", "vits_free_result",
    "vits_model_process", "vits_last_error", "vits_model_set_mode", "vits_model_get_mode",
    "vits_reference_noise_seed", "vits_model_process_ids", "vits_model_process_batch", "vits_free_batch_result",
    "vits_model_sync", "vits_model_tokenize", "vits_model_sampling_rate", "vits_model_vocab_size",
    "vits_model_weight_bytes", "vits_model_get_tap", "vits_synth_model_bytes", "vits_free_bytes",
    "vits_prof_enable", "vits_prof_reset", "vits_prof_report", "vits_op_conv1d", "vits_op_conv_transpose1d",
    "vits_op_rel_attention", "vits_op_add_layer_norm", "vits_device_info", "vits_set_device", "vits_model_file_reserialize",
    "vits_model_file_tokenize", "vits_pcm16_from_float", "vits_write_wav16", "vits_pcm16_from_float_device",
    "vits_model_set_arith", "vits_model_get_arith", "vits_model_file_validate", "vits_op_set_arith",
    "vits_model_set_arith_scope", "vits_model_get_arith_scope", "vits_model_submit_batch", "vits_model_wait", "vits_model_pending",
    "vits_model_set_ggml_tables", "vits_model_get_ggml_tables",
    "vits_pcm_gather_unique_id", "vits_pcm_gather_init", "vits_pcm_gather", "vits_pcm_gather_destroy", "vits_pcm_gather_verdict",
]


class VitsResult(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("size", C.c_size_t)]


# int on_chunk(void* user, int32 utt, size_t offset, const float* pcm, size_t n)  (include/vits.h vits_chunk_callback)
ChunkCallback = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_size_t, C.POINTER(C.c_float), C.c_size_t)


class ProcessOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("mode", C.c_int32), ("noise_kind", C.c_int32), ("noise_seed", C.c_uint64),
        ("noise_dur", C.c_void_p), ("noise_prior", C.c_void_p), ("noise_prior_stride", C.c_int64),
        ("fixed_duration", C.c_int32), ("collect_taps", C.c_int32), ("out_device", C.c_void_p),
        ("out_device_stride", C.c_int64), ("skip_host_copy", C.c_int32), ("async_", C.c_int32),
        ("vocoder_chunk_frames", C.c_int32), ("frames_only", C.c_int32), ("on_chunk", ChunkCallback), ("on_chunk_user", C.c_void_p),
        ("noise_seed_offsets", C.c_void_p),
    ]


class BatchResult(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("stride", C.c_size_t), ("lengths", C.POINTER(C.c_int64)),
                ("frames", C.POINTER(C.c_int64)), ("batch", C.c_size_t)]


class Conv1dDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("t", C.c_int32), ("t_stride", C.c_int32),
                ("k", C.c_int32), ("dilation", C.c_int32), ("pad_left", C.c_int32), ("pre_act", C.c_int32),
                ("pre_slope", C.c_float), ("post_act", C.c_int32), ("out_scale", C.c_float)]


class ConvT1dDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("t", C.c_int32), ("t_stride", C.c_int32),
                ("t_out_stride", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("crop", C.c_int32),
                ("pre_slope", C.c_float)]


_lib = None


def source_sha16():
    """sha256 (first 16 hex digits) over the library's sources (csrc/*.hip|cpp|h, include/*.h): identifies the kernel build
    that a profile artefact (profiles/*_pmc_*.json) was collected with. The GPU box has no .git, so a commit hash is not
    available there; this changes exactly when the code that runs changes."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def lib():
    """Load libvits_hip.so (once). Raises if it has not been built — there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, sz, f32p = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t, C.c_void_p
    L.vits_model_load_from_bytes.restype = vp
    L.vits_model_load_from_bytes.argtypes = [C.c_char_p, sz]
    L.vits_model_load_from_file.restype = vp
    L.vits_model_load_from_file.argtypes = [C.c_char_p]
    L.vits_free_model.restype = None
    L.vits_free_model.argtypes = [vp]
    L.vits_free_result.restype = None
    L.vits_free_result.argtypes = [VitsResult]
    L.vits_model_process.restype = VitsResult
    L.vits_model_process.argtypes = [vp, C.c_char_p]
    L.vits_last_error.restype = C.c_char_p
    L.vits_last_error.argtypes = []
    L.vits_model_set_mode.restype = i32
    L.vits_model_set_mode.argtypes = [vp, i32]
    L.vits_model_get_mode.restype = i32
    L.vits_model_get_mode.argtypes = [vp]
    L.vits_model_set_arith.restype = i32
    L.vits_model_set_arith.argtypes = [vp, i32]
    L.vits_model_get_arith.restype = i32
    L.vits_model_get_arith.argtypes = [vp]
    L.vits_model_set_arith_scope.restype = i32
    L.vits_model_set_arith_scope.argtypes = [vp, i32]
    L.vits_model_get_arith_scope.restype = i32
    L.vits_model_get_arith_scope.argtypes = [vp]
    L.vits_reference_noise_seed.restype = None
    L.vits_reference_noise_seed.argtypes = [C.c_uint32]
    L.vits_model_process_ids.restype = VitsResult
    L.vits_model_process_ids.argtypes = [vp, vp, sz]
    L.vits_model_process_batch.restype = i32
    L.vits_model_process_batch.argtypes = [vp, vp, vp, i32, i32, C.POINTER(ProcessOpts), C.POINTER(BatchResult)]
    L.vits_free_batch_result.restype = None
    L.vits_free_batch_result.argtypes = [C.POINTER(BatchResult)]
    L.vits_model_sync.restype = i32
    L.vits_model_sync.argtypes = [vp]
    L.vits_model_set_ggml_tables.restype = i32
    L.vits_model_set_ggml_tables.argtypes = [vp, i32]
    L.vits_model_get_ggml_tables.restype = i32
    L.vits_model_get_ggml_tables.argtypes = [vp]
    L.vits_model_submit_batch.restype = i32
    L.vits_model_submit_batch.argtypes = [vp, vp, vp, i32, i32, C.POINTER(ProcessOpts)]
    L.vits_model_wait.restype = i32
    L.vits_model_wait.argtypes = [vp, C.POINTER(BatchResult)]
    L.vits_model_pending.restype = i32
    L.vits_model_pending.argtypes = [vp]
    L.vits_model_tokenize.restype = i64
    L.vits_model_tokenize.argtypes = [vp, C.c_char_p, vp, sz]
    L.vits_model_sampling_rate.restype = i32
    L.vits_model_sampling_rate.argtypes = [vp]
    L.vits_model_vocab_size.restype = i32
    L.vits_model_vocab_size.argtypes = [vp]
    L.vits_model_weight_bytes.restype = i64
    L.vits_model_weight_bytes.argtypes = [vp]
    L.vits_model_get_tap.restype = i64
    L.vits_model_get_tap.argtypes = [vp, C.c_char_p, i32, f32p, sz]
    L.vits_synth_model_bytes.restype = i32
    L.vits_synth_model_bytes.argtypes = [u64, i32, C.POINTER(C.c_void_p), C.POINTER(sz)]
    L.vits_free_bytes.restype = None
    L.vits_free_bytes.argtypes = [vp]
    L.vits_prof_enable.restype = i32
    L.vits_prof_enable.argtypes = [vp, i32]
    L.vits_prof_reset.restype = i32
    L.vits_prof_reset.argtypes = [vp]
    L.vits_prof_report.restype = i64
    L.vits_prof_report.argtypes = [vp, C.c_char_p, sz]
    L.vits_op_conv1d.restype = i32
    L.vits_op_conv1d.argtypes = [C.POINTER(Conv1dDesc), vp, vp, vp, vp, vp, vp, vp]
    L.vits_op_conv_transpose1d.restype = i32
    L.vits_op_conv_transpose1d.argtypes = [C.POINTER(ConvT1dDesc), vp, vp, vp, vp, vp]
    L.vits_op_rel_attention.restype = i32
    L.vits_op_rel_attention.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.vits_op_add_layer_norm.restype = i32
    L.vits_op_add_layer_norm.argtypes = [i32, i32, i32, i32, C.c_float, vp, vp, vp, vp, vp]
    L.vits_model_file_reserialize.restype = i32
    L.vits_model_file_reserialize.argtypes = [C.c_char_p, sz, C.POINTER(C.c_void_p), C.POINTER(sz)]
    L.vits_model_file_tokenize.restype = i64
    L.vits_model_file_tokenize.argtypes = [C.c_char_p, sz, C.c_char_p, vp, sz]
    L.vits_pcm16_from_float.restype = None
    L.vits_pcm16_from_float.argtypes = [vp, sz, vp]
    L.vits_write_wav16.restype = i32
    L.vits_write_wav16.argtypes = [C.c_char_p, vp, sz, i32]
    L.vits_set_device.restype = i32
    L.vits_set_device.argtypes = [i32]
    L.vits_device_info.restype = i32
    L.vits_device_info.argtypes = [C.c_char_p, sz, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    _lib = L
    return L


class VitsError(RuntimeError):
    pass


def last_error():
    return lib().vits_last_error().decode("utf-8", "replace")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def synth_model_bytes(seed=0x5EED, arch=SYNTH_FULL):
    """Deterministic synthetic model file in the reference's on-disk format (host-only, no GPU needed)."""
    p, n = C.c_void_p(), C.c_size_t()
    if lib().vits_synth_model_bytes(seed, arch, C.byref(p), C.byref(n)) != 0:
        raise VitsError(last_error())
    try:
        return C.string_at(p, n.value)
    finally:
        lib().vits_free_bytes(p)


def reserialize(data):
    """parse + write back a model file (host only)"""
    p, n = C.c_void_p(), C.c_size_t()
    if lib().vits_model_file_reserialize(data, len(data), C.byref(p), C.byref(n)) != 0:
        raise VitsError(last_error())
    try:
        return C.string_at(p, n.value)
    finally:
        lib().vits_free_bytes(p)


def validate(data):
    """Host-only load check of a model file (vits_model_file_validate); raises VitsError with the reason."""
    f = lib().vits_model_file_validate
    f.restype, f.argtypes = C.c_int32, [C.c_char_p, C.c_size_t]
    if f(data, len(data)) != 0:
        raise VitsError(last_error())


def file_tokenize(data, text):
    buf = np.zeros(4 * len(text.encode("utf-8")) + 8, np.int32)
    n = lib().vits_model_file_tokenize(data, len(data), text.encode("utf-8"), _ptr(buf), buf.size)
    if n < 0:
        raise VitsError(last_error())
    return buf[:n].copy()


def synth_ids(batch, n_ids, vocab=38, ids_seed=1234):
    """Synthetic phoneme ids of include/vits_synth_noise.h (blank-interleaved, uniform over the vocabulary)."""
    M = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    def hash3(seed, stream, index):
        h = mix(seed ^ 0xD1B54A32D192ED03)
        h = mix(h ^ ((stream * 0x9E3779B97F4A7C15) & M))
        return mix(h ^ index)

    out = np.zeros((batch, n_ids), np.int32)
    for u in range(batch):
        for t in range(1, n_ids, 2):
            out[u, t] = 1 + (hash3((ids_seed + u) & M, 3, t) >> 33) % (vocab - 1)
    return out


class Model:
    """One loaded model on the current HIP device (mirror of the reference's opaque ``vits_model*``)."""

    def __init__(self, data=None, path=None):
        L = lib()
        if path is not None:
            self._h = L.vits_model_load_from_file(os.fsencode(path))
        else:
            self._h = L.vits_model_load_from_bytes(data, len(data))
        if not self._h:
            raise VitsError(last_error())

    def close(self):
        if getattr(self, "_h", None):
            lib().vits_free_model(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- reference entry points ---------------------------------------------------------------------
    def process(self, text):
        r = lib().vits_model_process(self._h, text.encode("utf-8"))
        if not r.data:
            raise VitsError(last_error())
        try:
            return np.ctypeslib.as_array(r.data, shape=(r.size,)).copy()
        finally:
            lib().vits_free_result(r)

    def process_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        r = lib().vits_model_process_ids(self._h, _ptr(ids), ids.size)
        if not r.data:
            raise VitsError(last_error())
        try:
            return np.ctypeslib.as_array(r.data, shape=(r.size,)).copy()
        finally:
            lib().vits_free_result(r)

    # -- extensions -----------------------------------------------------------------------------------
    def set_mode(self, mode):
        if lib().vits_model_set_mode(self._h, mode) != 0:
            raise VitsError(last_error())

    def set_arith(self, arith):
        """ARITH_F32 (exact, default) | ARITH_BF16 | ARITH_F16: conv operand precision (include/vits.h VITS_ARITH_*)"""
        if lib().vits_model_set_arith(self._h, arith) != 0:
            raise VitsError(last_error())

    @property
    def arith(self):
        return lib().vits_model_get_arith(self._h)

    def set_arith_scope(self, scope):
        """SCOPE_FLOW_VOCODER (default: stage one stays exact fp32, durations bit-identical to the fp32 path) | SCOPE_ALL_CONVS
        (the literal Q7 arithmetic: every conv of the path) — include/vits.h VITS_ARITH_SCOPE_*"""
        if lib().vits_model_set_arith_scope(self._h, scope) != 0:
            raise VitsError(last_error())

    def set_ggml_tables(self, on):
        """EMULATED ggml fp16 lookup tables for ggml_gelu / ggml_soft_max (Q8; inferred from upstream ggml, the reference's fork is absent)"""
        if lib().vits_model_set_ggml_tables(self._h, int(on)) != 0:
            raise VitsError(last_error())

    @property
    def ggml_tables(self):
        return bool(lib().vits_model_get_ggml_tables(self._h))

    @property
    def ggml_tables_mode(self):
        """0 off, 1 tables + stage one in the exact order shared with the oracle, 2 tables inside the throughput kernels"""
        return int(lib().vits_model_get_ggml_tables(self._h))

    @property
    def arith_scope(self):
        return lib().vits_model_get_arith_scope(self._h)

    @property
    def mode(self):
        return lib().vits_model_get_mode(self._h)

    @property
    def sampling_rate(self):
        return lib().vits_model_sampling_rate(self._h)

    @property
    def vocab_size(self):
        return lib().vits_model_vocab_size(self._h)

    @property
    def weight_bytes(self):
        return lib().vits_model_weight_bytes(self._h)

    def tokenize(self, text):
        buf = np.zeros(4 * len(text.encode("utf-8")) + 8, np.int32)
        n = lib().vits_model_tokenize(self._h, text.encode("utf-8"), _ptr(buf), buf.size)
        if n < 0:
            raise VitsError(last_error())
        return buf[:n].copy()

    def process_batch(self, ids, id_lengths=None, mode=MODE_DEFAULT, noise_kind=NOISE_COUNTER, noise_seed=4321,
                      noise_dur=None, noise_prior=None, fixed_duration=0, collect_taps=False, out_device=None,
                      out_device_stride=0, skip_host_copy=False, async_=False, vocoder_chunk_frames=0, on_chunk=None, frames_only=False, noise_seed_offsets=None, keep_pcm=True):
        """ids: int32 [B, id_stride]. Returns (list of per-utterance PCM arrays or None, lengths, frames).
        vocoder_chunk_frames > 0 runs the vocoder window by window (bit-identical PCM, bounded activations);
        on_chunk(utt, offset, pcm ndarray) is then called as each window's samples reach the host (return True to abort)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if ids.ndim == 1:
            ids = ids[None, :]
        B, stride = ids.shape
        lens = np.full(B, stride, np.int32) if id_lengths is None else np.ascontiguousarray(id_lengths, dtype=np.int32)
        o = ProcessOpts()
        o.struct_size = C.sizeof(ProcessOpts)
        o.mode, o.noise_kind, o.noise_seed = mode, noise_kind, noise_seed
        nd, npr = _f32(noise_dur), _f32(noise_prior)
        o.noise_dur, o.noise_prior = _ptr(nd), _ptr(npr)
        o.noise_prior_stride = 0 if npr is None else npr.shape[-1]
        o.fixed_duration, o.collect_taps = fixed_duration, int(collect_taps)
        o.out_device = out_device
        o.out_device_stride = out_device_stride
        o.skip_host_copy, o.async_ = int(skip_host_copy), int(async_)
        o.vocoder_chunk_frames = int(vocoder_chunk_frames)
        o.frames_only = int(frames_only)
        nso = None if noise_seed_offsets is None else np.ascontiguousarray(noise_seed_offsets, dtype=np.int32)
        if nso is not None and nso.size != B:
            raise ValueError("noise_seed_offsets needs one entry per utterance")
        o.noise_seed_offsets = _ptr(nso)
        cb_error = []
        if on_chunk is not None:
            def _cb(_user, utt, offset, pcm, n):
                try:
                    return 1 if on_chunk(int(utt), int(offset), np.ctypeslib.as_array(pcm, shape=(n,)).copy()) else 0
                except BaseException as e:  # never let an exception unwind through the C frames
                    cb_error.append(e)
                    return 1
            o.on_chunk = ChunkCallback(_cb)
        res = BatchResult()
        if lib().vits_model_process_batch(self._h, _ptr(ids), _ptr(lens), B, stride, C.byref(o), C.byref(res)) != 0:
            if cb_error:
                raise cb_error[0]
            raise VitsError(last_error())
        try:
            lengths = np.ctypeslib.as_array(res.lengths, shape=(B,)).copy()
            frames = np.ctypeslib.as_array(res.frames, shape=(B,)).copy()
            pcm = None
            if res.data and keep_pcm:
                full = np.ctypeslib.as_array(res.data, shape=(B, res.stride))
                pcm = [full[b, : lengths[b]].copy() for b in range(B)]
            return pcm, lengths, frames
        finally:
            lib().vits_free_batch_result(C.byref(res))

    def submit_batch(self, ids, id_lengths=None, mode=MODE_DEFAULT, noise_seed=4321, fixed_duration=0, out_device=None, out_device_stride=0,
                     skip_host_copy=False, vocoder_chunk_frames=0, noise_seed_offsets=None):
        """vits_model_submit_batch: queue one batch on this handle's pipeline (at most two in flight); its stage one runs under the
        previous batch's vocoder. Results come from wait(), in submission order, bit-identical to process_batch."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if ids.ndim == 1:
            ids = ids[None, :]
        B, stride = ids.shape
        lens = np.full(B, stride, np.int32) if id_lengths is None else np.ascontiguousarray(id_lengths, dtype=np.int32)
        o = ProcessOpts()
        o.struct_size = C.sizeof(ProcessOpts)
        o.mode, o.noise_kind, o.noise_seed = mode, NOISE_COUNTER, noise_seed
        o.fixed_duration = fixed_duration
        o.out_device = out_device
        o.out_device_stride = out_device_stride
        o.skip_host_copy = int(skip_host_copy)
        o.vocoder_chunk_frames = int(vocoder_chunk_frames)
        nso = None if noise_seed_offsets is None else np.ascontiguousarray(noise_seed_offsets, dtype=np.int32)
        if nso is not None and nso.size != B:
            raise ValueError("noise_seed_offsets needs one entry per utterance")
        o.noise_seed_offsets = _ptr(nso)
        if lib().vits_model_submit_batch(self._h, _ptr(ids), _ptr(lens), B, stride, C.byref(o)) != 0:
            raise VitsError(last_error())

    def wait(self, keep_pcm=True):
        """vits_model_wait: (pcm list or None, lengths, frames) of the oldest submitted batch."""
        res = BatchResult()
        if lib().vits_model_wait(self._h, C.byref(res)) != 0:
            raise VitsError(last_error())
        try:
            B = res.batch
            lengths = np.ctypeslib.as_array(res.lengths, shape=(B,)).copy()
            frames = np.ctypeslib.as_array(res.frames, shape=(B,)).copy()
            pcm = None
            if res.data and keep_pcm:
                full = np.ctypeslib.as_array(res.data, shape=(B, res.stride))
                pcm = [full[b, : lengths[b]].copy() for b in range(B)]
            return pcm, lengths, frames
        finally:
            lib().vits_free_batch_result(C.byref(res))

    @property
    def pending(self):
        return lib().vits_model_pending(self._h)

    def sync(self):
        if lib().vits_model_sync(self._h) != 0:
            raise VitsError(last_error())

    def tap(self, name, utt=0):
        n = lib().vits_model_get_tap(self._h, name.encode(), utt, None, 0)
        if n <= 0:
            raise VitsError(f"no tap '{name}': {last_error()}")
        out = np.zeros(n, np.float32)
        lib().vits_model_get_tap(self._h, name.encode(), utt, _ptr(out), n)
        return out

    def prof_enable(self, on=True):
        lib().vits_prof_enable(self._h, int(on))

    def prof_reset(self):
        lib().vits_prof_reset(self._h)

    def prof_report(self):
        buf = C.create_string_buffer(1 << 20)
        n = lib().vits_prof_report(self._h, buf, len(buf))
        if n < 0:
            raise VitsError(last_error())
        return json.loads(buf.value.decode())


# ---- operator-level wrappers (parity tests) ----------------------------------------------------------
def op_set_arith(arith):
    """Arithmetic of op_conv1d / op_conv_transpose1d on this thread (ARITH_F32 | ARITH_BF16 | ARITH_F16)."""
    f = lib().vits_op_set_arith
    f.restype, f.argtypes = C.c_int32, [C.c_int32]
    if f(arith) != 0:
        raise VitsError(last_error())


def op_conv1d(x, w, bias=None, dilation=1, pad_left=None, pre_slope=None, post_act=0, residual=None, accum=None,
              out_scale=1.0, lens=None):
    x, w = _f32(x), _f32(w)
    B, cin, T = x.shape
    cout, _, k = w.shape
    d = Conv1dDesc(B, cin, cout, T, T, k, dilation, (k - 1) * dilation // 2 if pad_left is None else pad_left,
                   0 if pre_slope is None else 1, 0.0 if pre_slope is None else pre_slope, post_act, out_scale)
    cy = cout // 2 if post_act == 2 else cout
    y = np.zeros((B, cy, T), np.float32)
    bias, residual, accum = _f32(bias), _f32(residual), _f32(accum)
    lens = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    if lib().vits_op_conv1d(C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(residual), _ptr(accum), _ptr(lens), _ptr(y)) != 0:
        raise VitsError(last_error())
    return y


def op_conv_transpose1d(x, w, bias, stride, crop, pre_slope=1.0, lens=None):
    x, w, bias = _f32(x), _f32(w), _f32(bias)
    B, cin, T = x.shape
    _, cout, k = w.shape
    To = stride * T + k - stride - 2 * crop
    d = ConvT1dDesc(B, cin, cout, T, T, To, k, stride, crop, pre_slope)
    y = np.zeros((B, cout, To), np.float32)
    lens = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    if lib().vits_op_conv_transpose1d(C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(lens), _ptr(y)) != 0:
        raise VitsError(last_error())
    return y


def op_rel_attention(q, k, v, rel_k, rel_v, heads, window, lens=None):
    q, k, v, rel_k, rel_v = map(_f32, (q, k, v, rel_k, rel_v))
    B, HD, T = q.shape
    out = np.zeros_like(q)
    lens = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    if lib().vits_op_rel_attention(B, heads, HD // heads, T, T, window, _ptr(q), _ptr(k), _ptr(v), _ptr(rel_k), _ptr(rel_v),
                                   _ptr(lens), _ptr(out)) != 0:
        raise VitsError(last_error())
    return out


def op_add_layer_norm(x, residual, gamma, beta, eps=1e-5):
    x, residual, gamma, beta = map(_f32, (x, residual, gamma, beta))
    B, Cc, T = x.shape
    y = np.zeros_like(x)
    if lib().vits_op_add_layer_norm(B, Cc, T, T, eps, _ptr(x), _ptr(residual), _ptr(gamma), _ptr(beta), _ptr(y)) != 0:
        raise VitsError(last_error())
    return y


def pcm16(pcm):
    pcm = _f32(pcm)
    out = np.zeros(pcm.size, np.int16)
    lib().vits_pcm16_from_float(_ptr(pcm), pcm.size, _ptr(out))
    return out


def pcm16_device(src_ptr, src_stride, dst_ptr, dst_stride, rows, cols, lengths_ptr=None, stream=None):
    """Device fp32 -> int16 rows (vits_pcm16_from_float_device). All pointers are integer device addresses; lengths_ptr
    (optional) is a device int64 [rows] array; stream a hipStream_t value (None = default stream). Asynchronous."""
    f = lib().vits_pcm16_from_float_device
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
    f.restype = C.c_int
    if f(src_ptr, src_stride, dst_ptr, dst_stride, lengths_ptr, rows, cols, stream) != 0:
        raise VitsError(last_error())


GATHER_ID_BYTES = 128


class GatherResult(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_int64), ("lengths", C.POINTER(C.c_int64)), ("rows_total", C.c_int32)]


def gather_unique_id():
    """vits_pcm_gather_unique_id: the 128 bytes rank 0 creates and hands to the other ranks (the RCCL unique id)."""
    f = lib().vits_pcm_gather_unique_id
    f.restype, f.argtypes = C.c_int, [C.c_char_p]
    buf = C.create_string_buffer(GATHER_ID_BYTES)
    if f(buf) != 0:
        raise VitsError(last_error())
    return buf.raw


def gather_verdict(table, world, rows):
    """vits_pcm_gather_verdict: the decision every rank takes on the table of the first all-gather ([row_capacity, lengths...] per rank, -1 = a row
    its rank could not use). Returns the common row width; raises VitsError with the message every rank would report."""
    t = np.ascontiguousarray(table, dtype=np.int64).reshape(world, rows + 1)
    f = lib().vits_pcm_gather_verdict
    f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    out = C.c_int64(0)
    if f(_ptr(t), world, rows, C.byref(out)) != 0:
        raise VitsError(last_error())
    return int(out.value)


class PcmGather:
    """The C ABI's PCM all-gather (include/vits.h vits_pcm_gather_*: RCCL through dlopen, no torch): what a C / C++ / Swift host calls;
    multi_gpu.PcmExchange is the torch.distributed form of the same exchange."""

    def __init__(self, unique_id, rank, world, rows, row_capacity, elem_bytes=4):
        L = lib()
        L.vits_pcm_gather_init.restype = C.c_void_p
        L.vits_pcm_gather_init.argtypes = [C.c_char_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]
        L.vits_pcm_gather.restype = C.c_int
        L.vits_pcm_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(GatherResult)]
        L.vits_pcm_gather_destroy.restype = None
        L.vits_pcm_gather_destroy.argtypes = [C.c_void_p]
        self._h = L.vits_pcm_gather_init(unique_id, len(unique_id) if unique_id is not None else 0, rank, world, rows, row_capacity, elem_bytes)
        if not self._h:
            raise VitsError(last_error())

    def gather(self, pcm_ptr, pcm_stride, lengths, stream=None):
        """pcm_ptr: integer device address of [rows][pcm_stride]; lengths: host int64 [rows]. Returns (device address of the gathered
        [rows_total][stride] block, stride, lengths of all rows as a numpy array)."""
        lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        r = GatherResult()
        if lib().vits_pcm_gather(self._h, pcm_ptr, pcm_stride, _ptr(lengths), stream, C.byref(r)) != 0:
            raise VitsError(last_error())
        return r.data, int(r.stride), np.ctypeslib.as_array(r.lengths, shape=(r.rows_total,)).copy()

    def close(self):
        if self._h:
            lib().vits_pcm_gather_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def write_wav16(path, pcm, sample_rate=16000):
    pcm = _f32(pcm)
    if lib().vits_write_wav16(os.fsencode(path), _ptr(pcm), pcm.size, sample_rate) != 0:
        raise VitsError(last_error())


def set_device(index):
    if lib().vits_set_device(index) != 0:
        raise VitsError(last_error())


def device_info():
    name = C.create_string_buffer(256)
    cu, mhz, hbm = C.c_int32(), C.c_int32(), C.c_int64()
    if lib().vits_device_info(name, 256, C.byref(cu), C.byref(mhz), C.byref(hbm)) != 0:
        raise VitsError(last_error())
    return {"name": name.value.decode(), "cu_count": cu.value, "clock_mhz": mhz.value, "hbm_bytes": hbm.value}
