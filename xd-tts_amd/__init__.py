"""xd-tts hot path on MI355X: Python mirror of the reference's Rust surface over the C ABI.

The product is ``libxdtts_hip.so`` (HIP kernels + ``extern "C"`` entry points, declared in
``include/xdtts.h``).  This module is the thin host-side binding used by tests and bench.py; it
mirrors the reference's operator interface for the hot path:

* ``Tacotron2.load(path)`` / ``Tacotron2.infer(ids)``   -- src/tacotron2/mod.rs:242,398
* ``create_mel_filter_bank(...)``                       -- griffin_lim::mel, src/tacotron2/mod.rs:453
* ``GriffinLim(mel_basis, noverlap, power, iter, momentum)`` / ``.infer(mel)``
                                                        -- src/tacotron2/mod.rs:456, src/lib.rs:141
* ``create_griffin_lim()``                              -- src/tacotron2/mod.rs:441-458

There is no CPU fallback: if the shared library is missing, importing fails; if no HIP device is
visible every call raises ``XdttsError`` (status XDTTS_ERR_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XDTTS_LIB") or os.path.join(_HERE, "libxdtts_hip.so")  # XDTTS_LIB: developer builds (profiling)

N_MEL = 80
EMB = 512
ATT_DIM = 128

STATUS = {0: "OK", 1: "BAD_ARG", 2: "IO", 3: "HIP", 4: "OOM", 5: "TOO_LONG", 6: "NO_DEVICE"}
XDTTS_ERR_BAD_ARG, XDTTS_ERR_IO, XDTTS_ERR_HIP, XDTTS_ERR_OOM, XDTTS_ERR_TOO_LONG, XDTTS_ERR_NO_DEVICE = 1, 2, 3, 4, 5, 6


class XdttsError(RuntimeError):
    """Non-zero xdtts_status; the Rust shim maps this to anyhow::Error (mod.rs:249,254,259)."""

    def __init__(self, status, message):
        super().__init__("xdtts status %d (%s): %s" % (status, STATUS.get(status, "?"), message))
        self.status = status


class GriffinLimOpts(C.Structure):
    """xdtts_griffinlim_opts: the conventions of GriffinLim::infer's mel->linear step as switches."""

    _fields_ = [("nnls_iters", C.c_int32), ("power_mode", C.c_int32), ("mel_decompress", C.c_int32), ("output_normalise", C.c_int32), ("batch_shape", C.c_int32), ("rms_target", C.c_float)]


class InferOpts(C.Structure):
    _fields_ = [
        ("gate_threshold", C.c_float),
        ("max_steps", C.c_int32),
        ("fixed_steps", C.c_int32),
        ("dropout_mode", C.c_int32),
        ("dropout_seed", C.c_uint32),
        ("max_chunk", C.c_int32),
        ("item_base", C.c_uint32),
        ("fixed_frames_per_id", C.c_float),
        ("dropout_masks", C.c_void_p),      # dropout_mode 2: keep bytes [chunk][steps][2][256]
        ("dropout_mask_steps", C.c_int32),
    ]


# name -> (restype, argtypes).  Every symbol include/xdtts.h declares is listed here; the CPU-only
# tests check that the library exports all of them.
_VP, _I32, _U32, _SZ, _F = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t, C.c_float
_PF = C.POINTER(C.c_float)
SYMBOLS = {
    "xdtts_infer_opts_default": (None, [C.POINTER(InferOpts)]),
    "xdtts_tacotron2_load": (_I32, [C.c_char_p, _I32, C.POINTER(_VP)]),
    "xdtts_model_dir_read": (_I32, [C.c_char_p, _VP, _SZ]),
    "xdtts_model_dir_describe": (_I32, [C.c_char_p, C.c_char_p, _SZ, C.POINTER(_SZ)]),
    "xdtts_tacotron2_load_synthetic": (_I32, [_U32, _F, _I32, C.POINTER(_VP)]),
    "xdtts_tacotron2_load_blob": (_I32, [_VP, _SZ, _I32, C.POINTER(_VP)]),
    "xdtts_tacotron2_save": (_I32, [_VP, C.c_char_p]),
    "xdtts_tensor_count": (_I32, []),
    "xdtts_tensor_name": (C.c_char_p, [_I32]),
    "xdtts_tensor_ndim": (_I32, [_I32]),
    "xdtts_tensor_dim": (_I32, [_I32, _I32]),
    "xdtts_tensor_offset": (_SZ, [_I32]),
    "xdtts_tensor_total": (_SZ, []),
    "xdtts_tacotron2_get_tensor": (_I32, [_VP, _I32, _VP]),
    "xdtts_tacotron2_infer_ids": (_I32, [_VP, _VP, _SZ, _VP, _SZ, C.POINTER(InferOpts), C.POINTER(_PF), C.POINTER(_SZ)]),
    "xdtts_tacotron2_infer_batch": (_I32, [_VP, _VP, _VP, _I32, _I32, C.POINTER(InferOpts), _VP, C.POINTER(_PF), C.POINTER(_SZ)]),
    "xdtts_tacotron2_encoder": (_I32, [_VP, _VP, _I32, _VP, _VP]),
    "xdtts_tacotron2_decoder": (_I32, [_VP, _VP, _VP, _I32, _I32, C.POINTER(InferOpts), _VP, _VP, C.POINTER(_SZ)]),
    "xdtts_tacotron2_decoder_step": (_I32, [_VP, _VP, _VP, _I32, _I32, C.POINTER(InferOpts), _U32] + [_VP] * 10),
    "xdtts_tacotron2_decoder_steps": (_I32, [_VP, _I32, _I32, _VP, _VP, _I32, _VP, C.POINTER(InferOpts), _U32, _I32] + [_VP] * 10),
    "xdtts_tacotron2_engine_state": (_I32, [_VP, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "xdtts_tacotron2_small_batch_engine_state": (_I32, [_VP, C.POINTER(_I32)]),
    "xdtts_tacotron2_engine_reset": (_I32, [_VP]),
    "xdtts_tacotron2_postnet": (_I32, [_VP, _VP, _I32, _VP]),
    "xdtts_tacotron2_last_timings": (_I32, [_VP, C.POINTER(C.c_float * 4), C.POINTER(_I32)]),
    "xdtts_tacotron2_free": (None, [_VP]),
    "xdtts_tacotron2_sync": (_I32, [_VP]),
    "xdtts_mel_filter_bank": (_I32, [_F, _SZ, _SZ, _F, _F, _VP]),
    "xdtts_griffinlim_new": (_I32, [_VP, _SZ, _SZ, _SZ, _F, _SZ, _F, _I32, C.POINTER(_VP)]),
    "xdtts_griffinlim_opts_default": (None, [_VP]),
    "xdtts_griffinlim_set_opts": (_I32, [_VP, _VP]),
    "xdtts_griffinlim_get_opts": (_I32, [_VP, _VP]),
    "xdtts_griffinlim_set_seed": (_I32, [_VP, _U32]),
    "xdtts_griffinlim_infer": (_I32, [_VP, _VP, _SZ, _SZ, C.POINTER(_PF), C.POINTER(_SZ)]),
    "xdtts_griffinlim_infer_batch": (_I32, [_VP, _VP, _SZ, _VP, _I32, _VP, _VP]),
    "xdtts_griffinlim_infer_linear": (_I32, [_VP, _VP, _VP, _SZ, _SZ, C.POINTER(_PF), C.POINTER(_SZ)]),
    "xdtts_griffinlim_mel_to_linear": (_I32, [_VP, _VP, _SZ, _SZ, _VP]),
    "xdtts_griffinlim_step": (_I32, [_VP, _VP, _VP, _VP, _SZ, _SZ]),
    "xdtts_griffinlim_last_timings": (_I32, [_VP, C.POINTER(C.c_float * 3)]),
    "xdtts_griffinlim_free": (None, [_VP]),
    "xdtts_synthesize_ids": (_I32, [_VP, _VP, _VP, _SZ, _VP, _SZ, C.POINTER(InferOpts), C.POINTER(_PF), C.POINTER(_SZ), C.POINTER(_PF), C.POINTER(_SZ)]),
    "xdtts_synthesize_batch": (_I32, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _I32, C.POINTER(InferOpts), _VP, _VP, _VP, _VP, _VP]),
    "xdtts_synthesize_sequence": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _I32, C.POINTER(InferOpts), _VP, _VP, _VP, _VP]),
    "xdtts_symbol_count": (_I32, []),
    "xdtts_symbol_token": (C.c_char_p, [_I32]),
    "xdtts_unit_id": (C.c_int64, [C.c_char_p, _I32]),
    "xdtts_split_score": (_I32, [C.c_int64]),
    "xdtts_find_splits": (_I32, [_VP, _SZ, _SZ, _VP, _SZ, C.POINTER(_SZ)]),
    "xdtts_audio_to_i16": (_I32, [_VP, _SZ, _VP]),
    "xdtts_build_info": (C.c_char_p, []),
    "xdtts_edge_floor_us": (_I32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "xdtts_default_device": (C.c_int32, []),
    "xdtts_silence_samples": (_SZ, [C.c_double, _U32]),
    "xdtts_silence_samples_duration": (_SZ, [C.c_uint64, _U32, _U32]),
    "xdtts_wav_write": (_I32, [C.c_char_p, _VP, _SZ, _U32]),
    "xdtts_npy_write_f32": (_I32, [C.c_char_p, _VP, _SZ, _SZ]),
    "xdtts_real_time_factor": (C.c_double, [C.c_double, _SZ]),
    "xdtts_free": (None, [_VP]),
    "xdtts_last_error": (C.c_char_p, []),
    "xdtts_device_count": (_I32, []),
}


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64
    and ask for them by file name, so a process that loads libxdtts_hip.so first (system ROCm runtime,
    soname libamdhip64.so.7) and imports torch later ends up with TWO runtimes driving the same GPU
    (observed: a device-side error word the host never sees).  If torch is installed but not imported
    yet, map ITS runtime first -- without importing torch -- so that our DT_NEEDED resolves to that copy
    and a later `import torch` finds it already loaded.  With torch imported first nothing is needed;
    without torch installed the system runtime is the only one."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    lib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(lib):
        try:
            C.CDLL(lib, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load():
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C %s` (or __graft_entry__.build()); there is no CPU fallback" % (LIB_PATH, _HERE)
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def _check(status):
    if status != 0:
        raise XdttsError(status, lib.xdtts_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def default_opts(**kw):
    """xdtts_infer_opts with the reference's defaults; `dropout_masks=` takes a uint8 array (chunks, steps, 2, 256) of keep
    bytes and selects dropout_mode 2 (the array is kept alive by the returned struct)."""
    o = InferOpts()
    lib.xdtts_infer_opts_default(C.byref(o))
    masks = kw.pop("dropout_masks", None)
    for k, v in kw.items():
        setattr(o, k, v)
    if masks is not None:
        m = np.ascontiguousarray(masks, dtype=np.uint8)
        if m.ndim != 4 or m.shape[2:] != (2, 256):
            raise ValueError("dropout_masks must be (chunks, steps, 2, 256)")
        o._masks_keepalive = m
        o.dropout_masks = m.ctypes.data
        o.dropout_mask_steps = m.shape[1]
        o.dropout_mode = 2
    return o


def device_count():
    return lib.xdtts_device_count()


def tensor_table():
    out = []
    for i in range(lib.xdtts_tensor_count()):
        shape = tuple(lib.xdtts_tensor_dim(i, d) for d in range(lib.xdtts_tensor_ndim(i)))
        out.append((lib.xdtts_tensor_name(i).decode(), shape, lib.xdtts_tensor_offset(i)))
    return out


def read_model_dir(path):
    """The host half of Tacotron2::load(path): the reference's model directory (encoder.onnx,
    decoder_iter.onnx, postnet.onnx -- src/tacotron2/mod.rs:246-259) or a tacotron2.xdtw container as a
    dict name -> array of the canonical tensors.  Needs no device."""
    blob = np.empty(lib.xdtts_tensor_total(), dtype=np.float32)
    _check(lib.xdtts_model_dir_read(os.fsencode(path), _ptr(blob), blob.size))
    return {n: blob[off : off + int(np.prod(shape))].reshape(shape) for n, shape, off in tensor_table()}


def describe_model_dir(path):
    """{file: (inputs, outputs)} of the three ONNX graphs of a model directory -- the names the reference binds at
    src/tacotron2/mod.rs:284-296,306-307,332-339,349."""
    need = C.c_size_t()
    _check(lib.xdtts_model_dir_describe(os.fsencode(path), None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _check(lib.xdtts_model_dir_describe(os.fsencode(path), buf, need.value, None))
    out = {}
    for line in buf.value.decode().splitlines():
        f, rest = line.split(": inputs ", 1)
        i, o = rest.split(" ; outputs ", 1)
        out[f] = ([x for x in i.split(",") if x], [x for x in o.split(",") if x])
    return out


class _Pinned:
    """Owner of one library buffer (pinned host memory): xdtts_free when the last view of it goes."""

    def __init__(self, addr):
        self.addr = addr

    def __del__(self):
        if self.addr and lib is not None:
            lib.xdtts_free(C.cast(self.addr, _PF))
            self.addr = None


def _take(ptr, n, shape):
    """A numpy view of a library-owned pinned buffer (no copy: a batch of audio is tens of MB); the buffer
    goes back to the library's pool when the array and every view of it are gone."""
    addr = C.cast(ptr, C.c_void_p).value
    if not n or not addr:
        if addr:
            lib.xdtts_free(ptr)
        return np.zeros(shape, dtype=np.float32)
    buf = (C.c_float * int(n)).from_address(addr)
    buf._owner = _Pinned(addr)  # numpy keeps `buf` as the array's base, `buf` keeps the owner
    return np.frombuffer(buf, dtype=np.float32).reshape(shape)


def generate_id_list():
    """generate_id_list() -- src/tacotron2/mod.rs:90-122, as token strings."""
    return [lib.xdtts_symbol_token(i).decode() for i in range(lib.xdtts_symbol_count())]


def unit_id(token, as_character=False):
    """Unit::from_str + best_match_for_unit (src/phonemes.rs:450-487,627-660); None if no id."""
    i = lib.xdtts_unit_id(token.encode(), 1 if as_character else 0)
    return None if i < 0 else int(i)


def units_to_ids(tokens, as_character=False):
    """The filter_map of src/tacotron2/mod.rs:403-406: units without an id are dropped."""
    out = [unit_id(t, as_character) for t in tokens]
    return np.array([i for i in out if i is not None], dtype=np.int64)


SAMPLE_RATE = 22050  # WAV_SPEC, src/lib.rs:25-30


def build_info():
    """xdtts_build_info() as a dict (src_sha256, arch, built_utc, compiler)."""
    return dict(kv.split("=", 1) for kv in lib.xdtts_build_info().decode().split(" ") if "=" in kv)


def source_hash():
    """The hash the Makefile computes: sha256 over csrc/* and include/xdtts.h in name order."""
    import glob
    import hashlib

    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*")) + [os.path.join(here, "..", "include", "xdtts.h")],
                   key=lambda p: os.path.relpath(p, here))
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


DEVICE_DEFAULT = -1  # XDTTS_DEVICE_DEFAULT: the process's default GPU (environment variable XDTTS_DEVICE, 0 without it)


def default_device():
    """What device_id = DEVICE_DEFAULT resolves to now (xdtts_default_device)."""
    v = lib.xdtts_default_device()
    if v < 0:
        raise XdttsError(2, lib.xdtts_last_error().decode())
    return int(v)


def edge_floor_us(device_id=0, steps=2000, T=100, tuned=True):
    """Latency floor of one persistent-decoder step on this device now (xdtts_edge_floor_us): us per step."""
    us = C.c_double()
    _check(lib.xdtts_edge_floor_us(device_id, steps, T, 1 if tuned else 0, C.byref(us)))
    return us.value


def audio_to_i16(audio):
    """`(sample * i16::MAX as f32) as i16` (src/lib.rs:153-155): truncating, saturating, NaN -> 0."""
    a = np.ascontiguousarray(audio, dtype=np.float32).ravel()
    out = np.empty(a.size, dtype=np.int16)
    _check(lib.xdtts_audio_to_i16(_ptr(a), a.size, _ptr(out)))
    return out


def silence_samples(seconds, sample_rate=SAMPLE_RATE):
    """write_silence (src/lib.rs:162-176)."""
    return int(lib.xdtts_silence_samples(float(seconds), int(sample_rate)))


def silence_samples_duration(secs, nanos, sample_rate=SAMPLE_RATE):
    """write_silence (src/lib.rs:162-176) from a Rust Duration's (secs, subsec_nanos): f32 throughout."""
    return int(lib.xdtts_silence_samples_duration(int(secs), int(nanos), int(sample_rate)))


def write_wav(path, audio, sample_rate=SAMPLE_RATE):
    """f32 audio (or ready int16 PCM) -> mono 16-bit WAV with the reference's WAV_SPEC."""
    pcm = np.ascontiguousarray(audio) if np.asarray(audio).dtype == np.int16 else audio_to_i16(audio)
    _check(lib.xdtts_wav_write(os.fsencode(path), _ptr(pcm), pcm.size, int(sample_rate)))


def write_mel_npy(path, mel):
    """ndarray_npy::write_npy of the (80, F) spectrogram (src/lib.rs:128-141)."""
    m = np.ascontiguousarray(mel, dtype=np.float32)
    _check(lib.xdtts_npy_write_f32(os.fsencode(path), _ptr(m), m.shape[0], m.shape[1]))


def real_time_factor(compute_seconds, n_samples):
    return float(lib.xdtts_real_time_factor(float(compute_seconds), int(n_samples)))


def find_splits(ids, max_size=100):
    """find_splits(units, max_size) -- src/phonemes.rs:681-753."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    out = np.empty(max(ids.size, 1) + 2, dtype=np.uintp)
    n = C.c_size_t()
    _check(lib.xdtts_find_splits(_ptr(ids), ids.size, max_size, _ptr(out), out.size, C.byref(n)))
    return out[: n.value].astype(np.int64)


class Tacotron2:
    """Mirror of ``Tacotron2`` (src/tacotron2/mod.rs:139-148): ``load`` then ``infer``."""

    def __init__(self, handle):
        self._h = handle

    # -- constructors ---------------------------------------------------------------------------
    @classmethod
    def load(cls, path, device_id=0):
        """Tacotron2::load(path) -- src/tacotron2/mod.rs:242."""
        h = C.c_void_p()
        _check(lib.xdtts_tacotron2_load(os.fsencode(path), device_id, C.byref(h)))
        return cls(h)

    @classmethod
    def synthetic(cls, seed=20240327, rec_scale=1.0, device_id=0):
        h = C.c_void_p()
        _check(lib.xdtts_tacotron2_load_synthetic(seed, rec_scale, device_id, C.byref(h)))
        return cls(h)

    @classmethod
    def from_blob(cls, blob, device_id=0):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        h = C.c_void_p()
        _check(lib.xdtts_tacotron2_load_blob(_ptr(blob), blob.size, device_id, C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            lib.xdtts_tacotron2_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def save(self, path):
        _check(lib.xdtts_tacotron2_save(self._h, os.fsencode(path)))

    def blob(self):
        """The canonical flat fp32 weight blob of this handle (what from_blob / xdtts_tacotron2_load_blob take)."""
        out = np.zeros(lib.xdtts_tensor_total(), dtype=np.float32)
        for i, (_n, shape, off) in enumerate(tensor_table()):
            t = np.empty(shape, dtype=np.float32)
            _check(lib.xdtts_tacotron2_get_tensor(self._h, i, _ptr(t)))
            out[off : off + t.size] = t.ravel()
        return out

    def get_tensor(self, name):
        for i, (n, shape, _off) in enumerate(tensor_table()):
            if n == name:
                out = np.empty(shape, dtype=np.float32)
                _check(lib.xdtts_tacotron2_get_tensor(self._h, i, _ptr(out)))
                return out
        raise KeyError(name)

    # -- the reference surface --------------------------------------------------------------------
    def infer(self, ids, splits=None, opts=None):
        """Tacotron2::infer -- src/tacotron2/mod.rs:398; ids after best_match_for_unit, splits from
        find_splits.  Returns the (80, F) mel like the reference's Array2<f32>."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        sp = None if splits is None else np.ascontiguousarray(splits, dtype=np.uintp)
        mel, nf = _PF(), C.c_size_t()
        _check(
            lib.xdtts_tacotron2_infer_ids(
                self._h, _ptr(ids), ids.size, None if sp is None else _ptr(sp), 0 if sp is None else sp.size, C.byref(opts) if opts else None, C.byref(mel), C.byref(nf)
            )
        )
        return _take(mel, N_MEL * nf.value, (N_MEL, nf.value))

    def infer_units(self, tokens, opts=None, as_character=False):
        """Tacotron2::infer(&[Unit]) end to end (mod.rs:398-437): unit tokens -> ids (units with no
        id dropped) -> find_splits(.., window) -> chunks -> mel."""
        ids = units_to_ids(tokens, as_character)
        window = opts.max_chunk if opts else 100
        return self.infer(ids, splits=find_splits(ids, window), opts=opts)

    def infer_batch(self, ids_list, opts=None, fixed_steps=None):
        """B independent chunks (infer_chunk, mod.rs:361-393) decoded in lock-step."""
        B = len(ids_list)
        lens = np.array([len(x) for x in ids_list], dtype=np.int32)
        stride = int(lens.max()) if B else 1
        ids = np.zeros((B, stride), dtype=np.int64)
        for b, x in enumerate(ids_list):
            ids[b, : len(x)] = x
        fs = None if fixed_steps is None else np.ascontiguousarray(fixed_steps, dtype=np.int32)
        mels = (_PF * B)()
        nf = (C.c_size_t * B)()
        _check(lib.xdtts_tacotron2_infer_batch(self._h, _ptr(ids), _ptr(lens), B, stride, C.byref(opts) if opts else None, None if fs is None else _ptr(fs), mels, nf))
        return [_take(mels[b], N_MEL * nf[b], (N_MEL, nf[b])) for b in range(B)]

    # -- parity hooks: the three graphs one at a time ---------------------------------------------
    def encoder(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        T = ids.size
        memory = np.empty((T, EMB), dtype=np.float32)
        pmem = np.empty((T, ATT_DIM), dtype=np.float32)
        _check(lib.xdtts_tacotron2_encoder(self._h, _ptr(ids), T, _ptr(memory), _ptr(pmem)))
        return memory, pmem

    def decoder(self, memory, pmem, n_valid, opts=None):
        memory = np.ascontiguousarray(memory, dtype=np.float32)
        pmem = np.ascontiguousarray(pmem, dtype=np.float32)
        o = opts if opts else default_opts()
        frames = np.empty((o.max_steps, N_MEL), dtype=np.float32)
        gates = np.empty(o.max_steps, dtype=np.float32)
        nf = C.c_size_t()
        _check(lib.xdtts_tacotron2_decoder(self._h, _ptr(memory), _ptr(pmem), memory.shape[0], n_valid, C.byref(o), _ptr(frames), _ptr(gates), C.byref(nf)))
        return frames[: nf.value].copy(), gates[: nf.value].copy()

    def decoder_step(self, memory, pmem, n_valid, state, decoder_input, step, opts=None):
        """ONE decoder_iter call (mod.rs:304).  `state` = dict with the reference's tensor names
        (attention_hidden, attention_cell, decoder_hidden, decoder_cell, attention_weights,
        attention_weights_cum, attention_context); returns (decoder_output, gate_prediction, new state)."""
        memory = np.ascontiguousarray(memory, dtype=np.float32)
        pmem = np.ascontiguousarray(pmem, dtype=np.float32)
        names = ("attention_hidden", "attention_cell", "decoder_hidden", "decoder_cell", "attention_weights", "attention_weights_cum", "attention_context")
        st = {k: np.array(state[k], dtype=np.float32, order="C") for k in names}
        din = np.ascontiguousarray(decoder_input, dtype=np.float32)
        out, gate = np.empty(N_MEL, dtype=np.float32), np.empty(1, dtype=np.float32)
        _check(lib.xdtts_tacotron2_decoder_step(self._h, _ptr(memory), _ptr(pmem), memory.shape[0], n_valid, C.byref(opts) if opts else None, step,
                                                _ptr(din), *[_ptr(st[k]) for k in names], _ptr(out), _ptr(gate)))
        return out, float(gate[0]), st

    ENGINES = {"launch": 0, "persistent": 1, "batched": 2, "persistent8": 3}

    def decoder_steps(self, engine, memory, pmem, n_valid, states, decoder_input, step0, n_steps=1, opts=None):
        """n_steps decoder_iter calls for B chunks through the named engine ("launch" / "persistent" / "batched").
        memory (B, T, 512), pmem (B, T, 128), n_valid (B,), decoder_input (B, 80); `states` = dict of (B, ...) arrays with the
        reference's tensor names.  Returns (decoder_output (B, n_steps, 80), gate_prediction (B, n_steps), new states)."""
        memory = np.ascontiguousarray(memory, dtype=np.float32)
        pmem = np.ascontiguousarray(pmem, dtype=np.float32)
        B, T = memory.shape[0], memory.shape[1]
        nv = np.ascontiguousarray(n_valid, dtype=np.int32)
        names = ("attention_hidden", "attention_cell", "decoder_hidden", "decoder_cell", "attention_weights", "attention_weights_cum", "attention_context")
        st = {k: np.array(states[k], dtype=np.float32, order="C") for k in names}
        din = np.ascontiguousarray(decoder_input, dtype=np.float32)
        out, gate = np.empty((B, n_steps, N_MEL), dtype=np.float32), np.empty((B, n_steps), dtype=np.float32)
        _check(lib.xdtts_tacotron2_decoder_steps(self._h, self.ENGINES[engine], B, _ptr(memory), _ptr(pmem), T, _ptr(nv), C.byref(opts) if opts else None,
                                                 step0, n_steps, _ptr(din), *[_ptr(st[k]) for k in names], _ptr(out), _ptr(gate)))
        return out, gate, st

    def engine_state(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib.xdtts_tacotron2_engine_state(self._h, C.byref(a), C.byref(b), C.byref(c)))
        e = C.c_int32()
        _check(lib.xdtts_tacotron2_small_batch_engine_state(self._h, C.byref(e)))
        return {"decoder_persistent": a.value, "encoder_cooperative": b.value, "batched_attention": c.value, "decoder_persistent8": e.value}

    def engine_reset(self):
        _check(lib.xdtts_tacotron2_engine_reset(self._h))

    def postnet(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        F = frames.shape[0]
        out = np.empty((N_MEL, F), dtype=np.float32)
        _check(lib.xdtts_tacotron2_postnet(self._h, _ptr(frames), F, _ptr(out)))
        return out

    def last_timings(self):
        ms = (C.c_float * 4)()
        steps = C.c_int32()
        _check(lib.xdtts_tacotron2_last_timings(self._h, C.byref(ms), C.byref(steps)))
        return {"encoder_ms": ms[0], "decoder_ms": ms[1], "postnet_ms": ms[2], "total_ms": ms[3], "steps": steps.value}


def create_mel_filter_bank(sample_rate, n_fft, n_mels, fmin, fmax=None):
    """griffin_lim::mel::create_mel_filter_bank -- src/tacotron2/mod.rs:453 (fmax: Option<f32>)."""
    out = np.empty((n_mels, n_fft // 2 + 1), dtype=np.float32)
    _check(lib.xdtts_mel_filter_bank(sample_rate, n_fft, n_mels, fmin, float("nan") if fmax is None else fmax, _ptr(out)))
    return out


class GriffinLim:
    """Mirror of griffin_lim::GriffinLim: ``new(mel_basis, noverlap, power, iter, momentum)``
    (src/tacotron2/mod.rs:456) and ``infer(&mel)`` (src/lib.rs:141)."""

    def __init__(self, mel_basis, noverlap, power, iters, momentum, device_id=0, seed=0):
        basis = np.ascontiguousarray(mel_basis, dtype=np.float32)
        self._h = C.c_void_p()
        _check(lib.xdtts_griffinlim_new(_ptr(basis), basis.shape[0], basis.shape[1], noverlap, power, iters, momentum, device_id, C.byref(self._h)))
        self.n_bins = basis.shape[1]
        self.set_seed(seed)

    def set_seed(self, seed):
        _check(lib.xdtts_griffinlim_set_seed(self._h, seed))

    def set_opts(self, **kw):
        """nnls_iters, power_mode (0 inverse / 1 direct / 2 none), mel_decompress (0 exp / 1 none / 2 10^x),
        output_normalise (0 none / 1 peak / 2 rms / 3 rms limited to a peak of 1), rms_target, batch_shape (0 auto / 4 the single-utterance shape: bit-identical batches); unspecified
        fields keep their current value."""
        o = self.get_opts()
        for k, v in kw.items():
            setattr(o, k, v)
        _check(lib.xdtts_griffinlim_set_opts(self._h, C.byref(o)))

    def get_opts(self):
        o = GriffinLimOpts()
        _check(lib.xdtts_griffinlim_get_opts(self._h, C.byref(o)))
        return o

    def close(self):
        if self._h:
            lib.xdtts_griffinlim_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def infer(self, mel):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        audio, n = _PF(), C.c_size_t()
        _check(lib.xdtts_griffinlim_infer(self._h, _ptr(mel), mel.shape[0], mel.shape[1], C.byref(audio), C.byref(n)))
        return _take(audio, n.value, (n.value,))

    def infer_batch(self, mels):
        """GriffinLim::infer for a list of (80, F_u) mels in one call; returns the list of audios."""
        ms = [np.ascontiguousarray(m, dtype=np.float32) for m in mels]
        n = len(ms)
        ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in ms])
        nf = (C.c_size_t * n)(*[m.shape[1] for m in ms])
        audios = (_PF * n)()
        ns = (C.c_size_t * n)()
        _check(lib.xdtts_griffinlim_infer_batch(self._h, ptrs, ms[0].shape[0], nf, n, audios, ns))
        return [_take(audios[u], ns[u], (ns[u],)) for u in range(n)]

    def infer_linear(self, S, phase0=None, iters=0):
        S = np.ascontiguousarray(S, dtype=np.float32)
        p0 = None if phase0 is None else np.ascontiguousarray(phase0, dtype=np.float32)
        audio, n = _PF(), C.c_size_t()
        _check(lib.xdtts_griffinlim_infer_linear(self._h, _ptr(S), None if p0 is None else _ptr(p0), S.shape[1], iters, C.byref(audio), C.byref(n)))
        return _take(audio, n.value, (n.value,))

    def step(self, S, angles, rebuilt, n_iter=1):
        """Parity hook: n_iter iterations from the state (angles, rebuilt), both (n_bins, F, 2); returns
        the new (angles, rebuilt)."""
        S = np.ascontiguousarray(S, dtype=np.float32)
        a = np.array(angles, dtype=np.float32, order="C")
        r = np.array(rebuilt, dtype=np.float32, order="C")
        _check(lib.xdtts_griffinlim_step(self._h, _ptr(S), _ptr(a), _ptr(r), S.shape[1], n_iter))
        return a, r

    def mel_to_linear(self, mel):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        S = np.empty((self.n_bins, mel.shape[1]), dtype=np.float32)
        _check(lib.xdtts_griffinlim_mel_to_linear(self._h, _ptr(mel), mel.shape[0], mel.shape[1], _ptr(S)))
        return S

    def last_timings(self):
        ms = (C.c_float * 3)()
        _check(lib.xdtts_griffinlim_last_timings(self._h, C.byref(ms)))
        return {"mel_to_linear_ms": ms[0], "iterations_ms": ms[1], "total_ms": ms[2]}


def create_griffin_lim(device_id=0, iters=30, seed=0):
    """create_griffin_lim() -- src/tacotron2/mod.rs:441-458 (sr 22050, n_fft 1024, 80 mels, fmin 0,
    fmax 8000, noverlap 768, power 1.7, 30 iterations, momentum 0.99)."""
    mel_basis = create_mel_filter_bank(22050.0, 1024, 80, 0.0, 8000.0)
    return GriffinLim(mel_basis, 1024 - 256, 1.7, iters, 0.99, device_id=device_id, seed=seed)


def synthesize(tacotron2, vocoder, ids, splits=None, opts=None):
    """XdTts::infer (src/lib.rs:110-159): mel-gen then vocoder, mel kept in HBM in between."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    sp = None if splits is None else np.ascontiguousarray(splits, dtype=np.uintp)
    mel, nf, audio, ns = _PF(), C.c_size_t(), _PF(), C.c_size_t()
    _check(
        lib.xdtts_synthesize_ids(
            tacotron2._h, vocoder._h, _ptr(ids), ids.size, None if sp is None else _ptr(sp), 0 if sp is None else sp.size, C.byref(opts) if opts else None, C.byref(mel), C.byref(nf), C.byref(audio), C.byref(ns)
        )
    )
    return _take(mel, N_MEL * nf.value, (N_MEL, nf.value)), _take(audio, ns.value, (ns.value,))


def synthesize_batch(tacotron2, vocoder, utterance_chunks, opts=None, fixed_steps=None, want_mels=True):
    """XdTts::infer (src/lib.rs:110-159) for several utterances in one call: `utterance_chunks[u]` is the
    list of <=window-id chunks of utterance u (find_splits + the trailing split, mod.rs:399,412-414);
    `fixed_steps`, if given, holds one frame count per chunk in the same nested shape.  All chunks decode
    in one lock-step batch and the vocoder batch reads the concatenated mel in HBM.  Returns
    (mels or None, audios), one entry per utterance."""
    chunks = [c for u in utterance_chunks for c in u]
    B, n_utt = len(chunks), len(utterance_chunks)
    lens = np.array([len(x) for x in chunks], dtype=np.int32)
    stride = int(lens.max()) if B else 1
    ids = np.zeros((B, stride), dtype=np.int64)
    for b, x in enumerate(chunks):
        ids[b, : len(x)] = x
    uc = np.array([len(u) for u in utterance_chunks], dtype=np.int32)
    fs = None if fixed_steps is None else np.ascontiguousarray([s for u in fixed_steps for s in u], dtype=np.int32)
    mels = (_PF * n_utt)() if want_mels else None
    audios = (_PF * n_utt)()
    nf, ns = (C.c_size_t * n_utt)(), (C.c_size_t * n_utt)()
    _check(
        lib.xdtts_synthesize_batch(
            tacotron2._h, vocoder._h, _ptr(ids), _ptr(lens), B, stride, _ptr(uc), n_utt, C.byref(opts) if opts else None, None if fs is None else _ptr(fs), mels, nf, audios, ns
        )
    )
    out_m = [_take(mels[u], N_MEL * nf[u], (N_MEL, nf[u])) for u in range(n_utt)] if want_mels else None
    return out_m, [_take(audios[u], ns[u], (ns[u],)) for u in range(n_utt)]


def synthesize_sequence(tacotron2, vocoder, ids_list, splits_list=None, opts=None, want_mels=True):
    """XdTts::infer (src/lib.rs:110-159) for a sequence of utterances, each decoded alone as `synthesize` does, the vocoder of one
    overlapped with the encoder of the next (xdtts_synthesize_sequence).  Returns (mels or None, audios)."""
    n = len(ids_list)
    idv = [np.ascontiguousarray(x, dtype=np.int64) for x in ids_list]
    spv = [None if (splits_list is None or splits_list[u] is None) else np.ascontiguousarray(splits_list[u], dtype=np.uintp) for u in range(n)]
    ids_p = (C.c_void_p * n)(*[x.ctypes.data for x in idv])
    n_ids = (C.c_size_t * n)(*[x.size for x in idv])
    sp_p = (C.c_void_p * n)(*[(None if s is None else s.ctypes.data) for s in spv])
    n_sp = (C.c_size_t * n)(*[(0 if s is None else s.size) for s in spv])
    mels = (_PF * n)() if want_mels else None
    audios = (_PF * n)()
    nf, ns = (C.c_size_t * n)(), (C.c_size_t * n)()
    _check(lib.xdtts_synthesize_sequence(tacotron2._h, vocoder._h, ids_p, n_ids, sp_p, n_sp, n, C.byref(opts) if opts else None, mels, nf, audios, ns))
    out_m = [_take(mels[u], N_MEL * nf[u], (N_MEL, nf[u])) for u in range(n)] if want_mels else None
    return out_m, [_take(audios[u], ns[u], (ns[u],)) for u in range(n)]
