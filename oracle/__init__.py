"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see xdtts_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (xd-tts_amd/) never does.  PARITY UNPINNED: see the header of
xdtts_oracle.h for what is and is not pinned against the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
N_MEL, EMB, ATT_DIM, T_MAX = 80, 512, 128, 512


def build(force=False):
    """gcc-compile liboracle_f32.so / liboracle_f64.so next to the sources."""
    src = [os.path.join(_HERE, f) for f in ("xdtts_oracle.c", "xdtts_oracle.h")]
    out = [os.path.join(_HERE, f) for f in ("liboracle_f32.so", "liboracle_f64.so", "liboracle_f32_omp.so")]
    stale = force or any(
        (not os.path.exists(o)) or os.path.getmtime(o) < max(os.path.getmtime(s) for s in src) for o in out
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return out


class DecoderOpts(C.Structure):
    _fields_ = [
        ("gate_threshold", C.c_float),
        ("max_steps", C.c_int32),
        ("fixed_steps", C.c_int32),
        ("dropout_mode", C.c_int32),
        ("dropout_seed", C.c_uint32),
        ("item", C.c_uint32),
        ("masks", C.c_void_p),       # dropout_mode 2: [mask_steps][2][256] keep bytes of this chunk
        ("mask_steps", C.c_int32),
    ]


def _state_struct(ct):
    class DecoderState(C.Structure):
        _fields_ = [
            ("att_h", ct * 1024),
            ("att_c", ct * 1024),
            ("dec_h", ct * 1024),
            ("dec_c", ct * 1024),
            ("aw", ct * T_MAX),
            ("awc", ct * T_MAX),
            ("ctx", ct * 512),
            ("dec_in", ct * 80),
        ]

    return DecoderState


class Oracle:
    """One precision flavour of the oracle: Oracle('f32') or Oracle('f64')."""

    def __init__(self, precision="f32", omp=False):
        """omp=True: the same f32 source compiled with -fopenmp (all host cores; identical results) --
        bench.py's all-cores CPU baseline.  Thread count: OMP_NUM_THREADS / the runtime's default."""
        build()
        if omp and precision != "f32":
            raise ValueError("the OpenMP build exists for f32 only")
        self.precision = precision
        self.dtype = np.float32 if precision == "f32" else np.float64
        self.ct = C.c_float if precision == "f32" else C.c_double
        self.lib = C.CDLL(os.path.join(_HERE, "liboracle_%s%s.so" % (precision, "_omp" if omp else "")))
        self.State = _state_struct(self.ct)
        L = self.lib
        L.orc_rng_u32.restype = C.c_uint32
        L.orc_rng_u32.argtypes = [C.c_uint32] * 3
        L.orc_rng_uniform.restype = C.c_float
        L.orc_rng_uniform.argtypes = [C.c_uint32] * 3
        L.orc_tensor_name.restype = C.c_char_p
        L.orc_tensor_numel.restype = C.c_size_t
        L.orc_tensor_offset.restype = C.c_size_t
        L.orc_total_floats.restype = C.c_size_t
        L.orc_tensor_index.argtypes = [C.c_char_p]
        L.orc_weights_synthetic.argtypes = [C.c_uint32, C.c_float, C.c_void_p]
        L.orc_sigmoid.restype = self.ct
        L.orc_sigmoid.argtypes = [self.ct]
        L.orc_run_decoder.restype = C.c_int
        L.orc_infer_chunk.restype = C.c_int
        L.orc_pinv.restype = C.c_int
        L.orc_mel_filter_bank.argtypes = [C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.orc_mel_to_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, self.ct, C.c_void_p]
        L.orc_griffinlim.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, self.ct, C.c_void_p]
        L.orc_output_normalise.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double]
        L.orc_output_normalise.restype = None
        L.orc_nnls_lipschitz.restype = C.c_double
        L.orc_nnls_lipschitz.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_mel_to_linear_opts.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, self.ct, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_griffinlim_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, self.ct]
        L.orc_dropout_keep.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]

    # ---- helpers -------------------------------------------------------------------------
    def _arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    # ---- weights -------------------------------------------------------------------------
    def tensor_table(self):
        L = self.lib
        out = []
        for i in range(L.orc_num_tensors()):
            shape = tuple(L.orc_tensor_dim(i, d) for d in range(L.orc_tensor_ndim(i)))
            out.append((L.orc_tensor_name(i).decode(), shape, L.orc_tensor_offset(i), L.orc_tensor_numel(i)))
        return out

    def weights_synthetic(self, seed=20240327, rec_scale=1.0):
        blob = np.empty(self.lib.orc_total_floats(), dtype=np.float32)
        self.lib.orc_weights_synthetic(seed, rec_scale, self._p(blob))
        return blob

    def tensor(self, blob, name):
        for n, shape, off, numel in self.tensor_table():
            if n == name:
                return blob[off : off + numel].reshape(shape)
        raise KeyError(name)

    # ---- tacotron2 -----------------------------------------------------------------------
    def default_opts(self, **kw):
        o = DecoderOpts()
        self.lib.orc_decoder_opts_default(C.byref(o))
        masks = kw.pop("masks", None)
        for k, v in kw.items():
            setattr(o, k, v)
        if masks is not None:  # explicit prenet keep masks (steps, 2, 256) uint8 of this chunk: dropout_mode 2
            m = np.ascontiguousarray(masks, dtype=np.uint8)
            assert m.ndim == 3 and m.shape[1:] == (2, 256)
            o._masks_keepalive = m
            o.masks = m.ctypes.data
            o.mask_steps = m.shape[0]
            o.dropout_mode = 2
        return o

    def encoder(self, blob, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        T = len(ids)
        memory = np.empty((T, EMB), dtype=self.dtype)
        pmem = np.empty((T, ATT_DIM), dtype=self.dtype)
        self.lib.orc_encoder(self._p(blob), self._p(ids), T, self._p(memory), self._p(pmem))
        return memory, pmem

    def new_state(self):
        s = self.State()
        self.lib.orc_decoder_state_init(C.byref(s))
        return s

    def decoder_step(self, blob, memory, pmem, n_valid, state, opts, step):
        memory, pmem = self._arr(memory), self._arr(pmem)
        mel = np.empty(N_MEL, dtype=self.dtype)
        gate = self.ct(0)
        self.lib.orc_decoder_step(
            self._p(blob), self._p(memory), self._p(pmem), memory.shape[0], n_valid, C.byref(state), C.byref(opts), C.c_uint32(step), self._p(mel), C.byref(gate)
        )
        return mel, gate.value

    def run_decoder(self, blob, memory, pmem, n_valid, opts):
        memory, pmem = self._arr(memory), self._arr(pmem)
        limit = opts.fixed_steps if opts.fixed_steps > 0 else opts.max_steps
        frames = np.empty((limit, N_MEL), dtype=self.dtype)
        gates = np.empty(limit, dtype=self.dtype)
        F = self.lib.orc_run_decoder(self._p(blob), self._p(memory), self._p(pmem), memory.shape[0], n_valid, C.byref(opts), self._p(frames), self._p(gates))
        return frames[:F].copy(), gates[:F].copy()

    def postnet(self, blob, frames):
        frames = self._arr(frames)
        F = frames.shape[0]
        out = np.empty((N_MEL, F), dtype=self.dtype)
        self.lib.orc_postnet(self._p(blob), self._p(frames), F, self._p(out))
        return out

    def infer_chunk(self, blob, ids, opts, window=100):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        limit = opts.fixed_steps if opts.fixed_steps > 0 else opts.max_steps
        out = np.empty(N_MEL * limit, dtype=self.dtype)
        F = self.lib.orc_infer_chunk(self._p(blob), self._p(ids), len(ids), window, C.byref(opts), self._p(out))
        return out[: N_MEL * F].reshape(N_MEL, F).copy()

    def sigmoid(self, x):
        return self.lib.orc_sigmoid(self.ct(x))

    def dropout_keep(self, seed, item, step, layer, j):
        return self.lib.orc_dropout_keep(seed, item, step, layer, j)

    # ---- griffin-lim ---------------------------------------------------------------------
    def mel_filter_bank(self, sr=22050.0, n_fft=1024, n_mels=80, fmin=0.0, fmax=8000.0):
        out = np.empty((n_mels, n_fft // 2 + 1), dtype=np.float32)
        self.lib.orc_mel_filter_bank(sr, n_fft, n_mels, fmin, fmax, self._p(out))
        return out

    def pinv(self, basis):
        basis = np.ascontiguousarray(basis, dtype=np.float32)
        out = np.empty((basis.shape[1], basis.shape[0]), dtype=np.float32)
        rc = self.lib.orc_pinv(self._p(basis), basis.shape[0], basis.shape[1], self._p(out))
        if rc:
            raise ValueError("basis is not full row rank")
        return out

    def mel_to_linear(self, pinv, mel, power=1.7):
        mel = self._arr(mel)
        pinv = np.ascontiguousarray(pinv, dtype=np.float32)
        S = np.empty((pinv.shape[0], mel.shape[1]), dtype=self.dtype)
        self.lib.orc_mel_to_linear(self._p(pinv), mel.shape[0], pinv.shape[0], self._p(mel), mel.shape[1], power, self._p(S))
        return S

    def mel_to_linear_opts(self, pinv, basis, mel, power=1.7, nnls_iters=0, power_mode=0, decompress=0):
        """Step 1 of GriffinLim::infer with the switches of xdtts_griffinlim_opts."""
        mel = self._arr(mel)
        pinv = np.ascontiguousarray(pinv, dtype=np.float32)
        basis = np.ascontiguousarray(basis, dtype=np.float32)
        S = np.empty((pinv.shape[0], mel.shape[1]), dtype=self.dtype)
        self.lib.orc_mel_to_linear_opts(self._p(pinv), self._p(basis), mel.shape[0], pinv.shape[0], self._p(mel), mel.shape[1], power, nnls_iters, power_mode, decompress, self._p(S))
        return S

    def nnls_lipschitz(self, basis):
        basis = np.ascontiguousarray(basis, dtype=np.float32)
        return float(self.lib.orc_nnls_lipschitz(self._p(basis), basis.shape[0], basis.shape[1]))

    def phase_init(self, seed, n_bins, F):
        out = np.empty((n_bins, F, 2), dtype=self.dtype)
        self.lib.orc_phase_init(C.c_uint32(seed), n_bins, F, self._p(out))
        return out

    def stft(self, y, n_fft=1024, hop=256):
        y = self._arr(y)
        F = len(y) // hop + 1
        out = np.empty((n_fft // 2 + 1, F, 2), dtype=self.dtype)
        self.lib.orc_stft(self._p(y), len(y), n_fft, hop, self._p(out), F)
        return out

    def istft(self, spec, n_fft=1024, hop=256):
        spec = self._arr(spec)
        F = spec.shape[1]
        y = np.empty(hop * (F - 1), dtype=self.dtype)
        self.lib.orc_istft(self._p(spec), F, n_fft, hop, self._p(y))
        return y

    def griffinlim_step(self, S, angles, rebuilt, n_fft=1024, hop=256, iters=1, momentum=0.99):
        """`iters` iterations from the state (angles, rebuilt), both (n_bins, F, 2); returns the new pair."""
        S = self._arr(S)
        a = np.array(angles, dtype=self.dtype, order="C")
        r = np.array(rebuilt, dtype=self.dtype, order="C")
        self.lib.orc_griffinlim_step(self._p(S), self._p(a), self._p(r), S.shape[1], n_fft, hop, iters, momentum)
        return a, r

    def output_normalise(self, audio, mode=2, target=0.1):
        """G6: what GriffinLim::infer does to the final ISTFT's samples (0 none / 1 peak / 2 rms); returns a copy."""
        y = np.array(audio, dtype=self.dtype, order="C")
        self.lib.orc_output_normalise(self._p(y), y.size, mode, target)
        return y

    def griffinlim(self, S, phase0=None, seed=0, n_fft=1024, hop=256, iters=30, momentum=0.99):
        S = self._arr(S)
        F = S.shape[1]
        audio = np.empty(hop * (F - 1), dtype=self.dtype)
        p0 = None if phase0 is None else self._arr(phase0)
        self.lib.orc_griffinlim(self._p(S), None if p0 is None else self._p(p0), seed, F, n_fft, hop, iters, momentum, self._p(audio))
        return audio
