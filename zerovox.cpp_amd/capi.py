"""ctypes binding of libzerovox_amd.so (the C-ABI of include/zerovox_amd.h).

Used by the parity tests, bench.py and smoke(): the same entry points a C/C++ host would call.
There is NO CPU fallback: if the library is missing or no gfx950 device is present, loading fails.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZEROVOX_AMD_LIB") or os.path.join(HERE, "libzerovox_amd.so")   # env override: A/B builds

# every symbol include/zerovox_amd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "zv_last_error", "zv_version", "zv_model_load", "zv_model_free", "zv_model_get_hparams", "zv_model_reserve",
    "zv_encode", "zv_encode_taps", "zv_decode", "zv_vocode", "zv_synthesize", "zv_synthesize_batch", "zv_synthesize_batch_begin", "zv_synthesize_batch_end", "zv_device_alloc", "zv_device_free",
    "zv_memcpy_h2d", "zv_memcpy_d2h", "zv_vocode_device", "zv_vocode_stream", "zv_vocoder_halo_frames", "zv_decode_device", "zv_synchronize", "zv_set_graph_mode",
    "zv_profile_begin", "zv_profile_end", "zv_write_wav", "zv_gguf_inspect", "zv_max_frames", "zv_demo_utterance", "zv_debug_layer", "zv_debug_set",
    "zv_debug_get", "zv_batch_timeline",
]


WAV_SINK = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_uint64, C.c_uint64)
BATCH_LANES = 4      # ZV_BATCH_LANES of include/zerovox_amd.h


class HParams(C.Structure):
    _fields_ = [("max_seq_len", C.c_uint32), ("emb_dim", C.c_uint32), ("punct_emb_dim", C.c_uint32),
                ("decoder_n_head", C.c_uint32), ("conv_filter_size", C.c_uint32), ("conv_kernel_size", C.c_uint32 * 2),
                ("encoder_layer", C.c_uint32), ("encoder_head", C.c_uint32), ("encoder_vp_filter_size", C.c_uint32),
                ("encoder_vp_kernel_size", C.c_uint32), ("encoder_ve_n_bins", C.c_uint32),
                ("audio_sampling_rate", C.c_uint32), ("audio_num_mels", C.c_uint32), ("audio_hop_size", C.c_uint32),
                ("voc_channels", C.c_uint32), ("voc_num_upsamples", C.c_uint32), ("voc_upsample_scales", C.c_uint32 * 8),
                ("voc_num_resblocks", C.c_uint32), ("voc_resblock_kernels", C.c_uint32 * 8)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint32), ("total_ms", C.c_double),
                ("algo_bytes", C.c_double), ("algo_flops", C.c_double)]


class ZvError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"zv_status {status}: {msg}")
        self.status = status


_lib = None


def load_library(path: Optional[str] = None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ZvError(-1, f"{p} not found: build it with `make -C zerovox.cpp_amd/csrc` (no CPU fallback exists)")
    lib = C.CDLL(p)
    vp, u32, i32p, fp = C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p
    lib.zv_last_error.restype = C.c_char_p
    lib.zv_version.restype = C.c_char_p
    lib.zv_model_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.zv_model_free.argtypes = [vp]
    lib.zv_model_free.restype = None
    lib.zv_model_get_hparams.argtypes = [vp, C.POINTER(HParams)]
    lib.zv_model_reserve.argtypes = [vp, u32, u32]
    lib.zv_encode.argtypes = [vp, i32p, i32p, fp, u32, u32, fp, C.POINTER(u32)]
    lib.zv_encode_taps.argtypes = [vp, i32p, i32p, fp, u32, u32, u32, fp, C.POINTER(u32), fp, fp, fp, fp, i32p, i32p]
    lib.zv_debug_layer.argtypes = [vp, C.c_int, C.c_int, fp, u32, fp, fp]
    lib.zv_debug_set.argtypes = [C.c_char_p, C.c_int]
    lib.zv_debug_get.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.zv_batch_timeline.argtypes = [vp, u32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(u32)]
    lib.zv_max_frames.argtypes = [vp]
    lib.zv_max_frames.restype = u32
    lib.zv_demo_utterance.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)]
    lib.zv_demo_utterance.restype = None
    lib.zv_decode.argtypes = [vp, fp, fp, u32, fp]
    lib.zv_vocode.argtypes = [vp, fp, u32, fp]
    lib.zv_synthesize.argtypes = [vp, i32p, i32p, fp, u32, u32, fp, C.POINTER(u32)]
    lib.zv_synthesize_batch.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u32), C.POINTER(u32),
                                        C.POINTER(vp), C.POINTER(u32)]
    lib.zv_synthesize_batch_begin.argtypes = [vp, u32, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u32), C.POINTER(u32),
                                              C.POINTER(vp), C.POINTER(u32)]
    lib.zv_synthesize_batch_end.argtypes = [vp, u32]
    lib.zv_device_alloc.argtypes = [vp, C.c_size_t]
    lib.zv_device_alloc.restype = vp
    lib.zv_device_free.argtypes = [vp, vp]
    lib.zv_device_free.restype = None
    lib.zv_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.zv_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.zv_vocode_device.argtypes = [vp, vp, u32, vp]
    lib.zv_vocode_stream.argtypes = [vp, fp, u32, u32, WAV_SINK, vp]
    lib.zv_vocoder_halo_frames.argtypes = [vp]
    lib.zv_vocoder_halo_frames.restype = u32
    lib.zv_decode_device.argtypes = [vp, vp, vp, u32, vp]
    lib.zv_synchronize.argtypes = [vp]
    lib.zv_set_graph_mode.argtypes = [vp, C.c_int]
    lib.zv_profile_begin.argtypes = [vp]
    lib.zv_profile_end.argtypes = [vp, C.POINTER(KernelStat), u32, C.POINTER(u32)]
    lib.zv_write_wav.argtypes = [C.c_char_p, fp, C.c_size_t, u32]
    lib.zv_gguf_inspect.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.c_int, C.c_char_p, C.POINTER(u32),
                                    C.POINTER(C.c_int64)]
    if path is None:
        _lib = lib
    _forward_env_switches(lib)
    return lib


def _forward_env_switches(lib):
    """The shipped library never reads the environment.  This test / bench binding forwards ZV_* variables that name a switch
    (csrc/knobs.h) through zv_debug_set, once per load, so that `ZV_PAIR_MT=4 python bench.py` still A/Bs a run; variables that
    name no switch of this build (ZV_BENCH_LANES, a diagnostic switch on a shipped build ...) are left alone."""
    for k, v in sorted(os.environ.items()):
        if not k.startswith("ZV_"):
            continue
        try:
            val = int(v)
        except ValueError:
            continue
        cur = C.c_int(0)
        if lib.zv_debug_get(k.encode(), C.byref(cur)) == 0:
            lib.zv_debug_set(k.encode(), val)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def debug_set(name: Optional[str], value: int = 0):
    """zv_debug_set: one test / measurement switch (csrc/knobs.h); name None resets every switch to its default"""
    lib = load_library()
    st = lib.zv_debug_set(None if name is None else name.encode(), int(value))
    if st != 0:
        raise ZvError(st, lib.zv_last_error().decode())


def debug_get(name: str) -> int:
    lib = load_library()
    v = C.c_int(0)
    st = lib.zv_debug_get(name.encode(), C.byref(v))
    if st != 0:
        raise ZvError(st, lib.zv_last_error().decode())
    return int(v.value)


class switches:
    """`with capi.switches(ZV_NO_FUSE=1, ...):` — the switches hold inside the block (schedule switches are sampled by
    Model(), kernel-regime switches at every launch; captured graphs are re-captured) and go back to the values they had
    before it afterwards (not to their built-in defaults: a value set for the whole run, e.g. ZV_ARENA_FILL=255, survives,
    and blocks nest)"""

    def __init__(self, **kw):
        self.kw = kw
        self.saved = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.saved[k] = debug_get(k)
            debug_set(k, int(v))
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            debug_set(k, v)
        return False


class Model:
    """One loaded checkpoint on one MI355X (zv_model)."""

    def __init__(self, gguf_path: str, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.zv_model_load(gguf_path.encode(), device, C.byref(h))
        if st != 0:
            raise ZvError(st, self.lib.zv_last_error().decode())
        self.h = h
        hp = HParams()
        self._chk(self.lib.zv_model_get_hparams(self.h, C.byref(hp)))
        self.hp = hp
        self.E = hp.emb_dim + hp.punct_emb_dim

    def close(self):
        if getattr(self, "h", None):
            self.lib.zv_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st != 0:
            raise ZvError(st, self.lib.zv_last_error().decode())

    # ---- host-buffer API (same call protocol as the reference's eval methods) ----
    def vocode(self, mel: np.ndarray) -> np.ndarray:
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        T = mel.shape[0]
        wav = np.empty(T * self.hp.audio_hop_size, np.float32)
        self._chk(self.lib.zv_vocode(self.h, _ptr(mel), T, _ptr(wav)))
        return wav

    def vocode_stream(self, mel: np.ndarray, chunk_frames: int):
        """chunked vocoding with halo (zv_vocode_stream): list of (first_sample, samples) in delivery order"""
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        chunks = []

        def sink(_user, wav, first, n):
            chunks.append((int(first), np.ctypeslib.as_array(wav, shape=(int(n),)).copy()))

        cb = WAV_SINK(sink)
        self._chk(self.lib.zv_vocode_stream(self.h, _ptr(mel), mel.shape[0], chunk_frames, cb, None))
        return chunks

    def vocoder_halo_frames(self) -> int:
        return int(self.lib.zv_vocoder_halo_frames(self.h))

    def decode(self, hidden: np.ndarray, style: np.ndarray) -> np.ndarray:
        hidden = np.ascontiguousarray(hidden, dtype=np.float32)
        style = np.ascontiguousarray(style, dtype=np.float32)
        T = hidden.shape[0]
        mel = np.empty((T, self.hp.audio_num_mels), np.float32)
        self._chk(self.lib.zv_decode(self.h, _ptr(hidden), _ptr(style), T, _ptr(mel)))
        return mel

    LAYER_VOC_RESBLOCK, LAYER_ENC_FFT, LAYER_DEC_BLOCK, LAYER_VAR_PRED = 0, 1, 2, 3
    LAYER_VOC_UPSAMPLE, LAYER_VOC_INPUT, LAYER_VOC_OUTPUT, LAYER_DEC_ASR_RES, LAYER_DEC_TO_OUT, LAYER_ENC_EMBED = 4, 5, 6, 7, 8, 9
    LAYER_ENC_MHA, LAYER_ENC_FFN, LAYER_DEC_ADAIN = 10, 11, 12

    def debug_layer(self, kind: int, index: int, x: np.ndarray, out_cols: int, style=None, out_rows: Optional[int] = None) -> np.ndarray:
        """zv_debug_layer: one layer of the production schedule on the given input (time-major [rows][cin]); out_rows when the
        layer changes the rate (transposed conv: rows x scale)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        orows = x.shape[0] if out_rows is None else out_rows
        out = np.empty((orows, out_cols) if out_cols else (orows,), np.float32)
        st = None if style is None else np.ascontiguousarray(style, dtype=np.float32)
        self._chk(self.lib.zv_debug_layer(self.h, kind, index, _ptr(x), x.shape[0], _ptr(st), _ptr(out)))
        return out

    def voc_rate(self, stage: int) -> int:
        r = 1
        for i in range(stage + 1):
            r *= self.hp.voc_upsample_scales[i]
        return r

    def voc_channels(self, stage: int) -> int:
        return self.hp.voc_channels >> (stage + 1)

    def max_frames(self) -> int:
        return int(self.lib.zv_max_frames(self.h))

    def encode(self, ids, puncts, style, T: int, num_phonemes: Optional[int] = None) -> dict:
        """num_phonemes < len(ids): all ids are encoded, the length regulator walks the first num_phonemes (FS2Encoder::eval)"""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        puncts = np.ascontiguousarray(puncts, dtype=np.int32)
        style = np.ascontiguousarray(style, dtype=np.float32)
        N, E = len(ids), self.E
        out = dict(hidden=np.empty((T, E), np.float32), features=np.empty((N, E), np.float32),
                   logdur=np.empty(N, np.float32), pitch=np.empty(N, np.float32), energy=np.empty(N, np.float32),
                   pitch_bucket=np.empty(N, np.int32), energy_bucket=np.empty(N, np.int32))
        nf = C.c_uint32(0)
        self._chk(self.lib.zv_encode_taps(self.h, _ptr(ids), _ptr(puncts), _ptr(style), N,
                                          N if num_phonemes is None else num_phonemes, T, _ptr(out["hidden"]),
                                          C.byref(nf), _ptr(out["features"]), _ptr(out["logdur"]), _ptr(out["pitch"]),
                                          _ptr(out["energy"]), _ptr(out["pitch_bucket"]), _ptr(out["energy_bucket"])))
        out["n_frames"] = int(nf.value)
        return out

    def synthesize(self, ids, puncts, style, T: int):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        puncts = np.ascontiguousarray(puncts, dtype=np.int32)
        style = np.ascontiguousarray(style, dtype=np.float32)
        wav = np.empty(T * self.hp.audio_hop_size, np.float32)
        nf = C.c_uint32(0)
        self._chk(self.lib.zv_synthesize(self.h, _ptr(ids), _ptr(puncts), _ptr(style), len(ids), T, _ptr(wav), C.byref(nf)))
        return wav, int(nf.value)

    def prepare_batch(self, utterances) -> "BatchCall":
        """argument arrays and output buffers of one zv_synthesize_batch call, built once (a C host would keep its
        buffers too): .run() is exactly one call of the C entry point, .results() the (wav, n_frames) list"""
        return BatchCall(self, utterances)

    def synthesize_batch(self, utterances):
        """utterances: list of (ids, puncts, style, T) -> list of (wav, n_frames); each utterance keeps its own (N, T)"""
        call = BatchCall(self, utterances)
        call.run()
        return call.results()

    # ---- device-resident API ----
    def device_alloc(self, nbytes: int) -> int:
        p = self.lib.zv_device_alloc(self.h, nbytes)
        if not p:
            raise ZvError(7, "zv_device_alloc failed")
        return p

    def device_free(self, p: int):
        self.lib.zv_device_free(self.h, p)

    def h2d(self, dst: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.zv_memcpy_h2d(self.h, dst, _ptr(arr), arr.nbytes))

    def d2h(self, arr: np.ndarray, src: int):
        self._chk(self.lib.zv_memcpy_d2h(self.h, _ptr(arr), src, arr.nbytes))

    def vocode_device(self, d_mel: int, T: int, d_wav: int):
        self._chk(self.lib.zv_vocode_device(self.h, d_mel, T, d_wav))

    def decode_device(self, d_hidden: int, d_style: int, T: int, d_mel: int):
        self._chk(self.lib.zv_decode_device(self.h, d_hidden, d_style, T, d_mel))

    def synchronize(self):
        self._chk(self.lib.zv_synchronize(self.h))

    def set_graph_mode(self, on: bool):
        self._chk(self.lib.zv_set_graph_mode(self.h, int(on)))

    def reserve(self, max_phonemes: int, max_frames: int):
        self._chk(self.lib.zv_model_reserve(self.h, max_phonemes, max_frames))

    def batch_timeline(self, cap: int = 64):
        """zv_batch_timeline: [(start_ms, end_ms)] of the most recent batches, oldest first"""
        a, b, n = (C.c_double * cap)(), (C.c_double * cap)(), C.c_uint32(0)
        self._chk(self.lib.zv_batch_timeline(self.h, cap, a, b, C.byref(n)))
        return [(a[i], b[i]) for i in range(n.value)]

    def profile_begin(self):
        self._chk(self.lib.zv_profile_begin(self.h))

    def profile_end(self) -> list:
        cap = 64
        arr = (KernelStat * cap)()
        n = C.c_uint32(0)
        self._chk(self.lib.zv_profile_end(self.h, arr, cap, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                     algo_bytes=arr[i].algo_bytes, algo_flops=arr[i].algo_flops) for i in range(min(cap, n.value))]


class BatchCall:
    def __init__(self, model: Model, utterances):
        self.model = model
        n = self.n = len(utterances)
        self.keep, self.wavs = [], []
        P = C.c_void_p * n
        self.ids_p, self.pun_p, self.sty_p, self.wav_p = P(), P(), P(), P()
        self.Ns, self.Ts, self.nf = (C.c_uint32 * n)(), (C.c_uint32 * n)(), (C.c_uint32 * n)()
        for i, (ids, puncts, style, T) in enumerate(utterances):
            a = np.ascontiguousarray(ids, dtype=np.int32)
            b = np.ascontiguousarray(puncts, dtype=np.int32)
            c = np.ascontiguousarray(style, dtype=np.float32)
            w = np.zeros(T * model.hp.audio_hop_size, np.float32)
            self.keep += [a, b, c]
            self.wavs.append(w)
            self.ids_p[i], self.pun_p[i], self.sty_p[i], self.wav_p[i] = a.ctypes.data, b.ctypes.data, c.ctypes.data, w.ctypes.data
            self.Ns[i], self.Ts[i] = len(a), T

    def run(self):
        m = self.model
        m._chk(m.lib.zv_synthesize_batch(m.h, self.n, self.ids_p, self.pun_p, self.sty_p, self.Ns, self.Ts, self.wav_p, self.nf))

    def begin(self, lane: int):
        """zv_synthesize_batch_begin on `lane`: returns once everything is enqueued; results are valid after end(lane)"""
        m = self.model
        m._chk(m.lib.zv_synthesize_batch_begin(m.h, lane, self.n, self.ids_p, self.pun_p, self.sty_p, self.Ns, self.Ts, self.wav_p, self.nf))

    def end(self, lane: int):
        m = self.model
        m._chk(m.lib.zv_synthesize_batch_end(m.h, lane))

    def results(self):
        return [(self.wavs[i], int(self.nf[i])) for i in range(self.n)]


def demo_utterance():
    """(ids[120], puncts[120], style[528]) of the reference's ZeroVOXModel::eval() (needs no GPU)"""
    lib = load_library()
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    n, ns = C.c_uint32(0), C.c_uint32(0)
    lib.zv_demo_utterance(C.byref(a), C.byref(b), C.byref(c), C.byref(n), C.byref(ns))
    ids = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_int32)), shape=(n.value,)).copy()
    pun = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_int32)), shape=(n.value,)).copy()
    sty = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_float)), shape=(ns.value,)).copy()
    return ids, pun, sty


def gguf_inspect(path: str, tensor_index: int = -1):
    """(n_tensors, max_seq_len[, name, ggml_type, ne]) as parsed by the product's C++ GGUF reader (no GPU needed)."""
    lib = load_library()
    n, T, typ = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    name = C.create_string_buffer(64)
    ne = (C.c_int64 * 4)()
    st = lib.zv_gguf_inspect(path.encode(), C.byref(n), C.byref(T), tensor_index, name, C.byref(typ), ne)
    if st != 0:
        raise ZvError(st, lib.zv_last_error().decode())
    if tensor_index < 0:
        return n.value, T.value
    return n.value, T.value, name.value.decode(), typ.value, list(ne)


def write_wav(path: str, wav: np.ndarray, sampling_rate: int):
    lib = load_library()
    wav = np.ascontiguousarray(wav, dtype=np.float32)
    st = lib.zv_write_wav(path.encode(), _ptr(wav), wav.size, sampling_rate)
    if st != 0:
        raise ZvError(st, lib.zv_last_error().decode())
