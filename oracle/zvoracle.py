"""ctypes front-end of the CPU oracle (oracle/zv_oracle.c) and driver for the compiled reference
(oracle/_ref/zvref).  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import tempfile
from typing import Dict, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libzvoracle.so")
REF_BIN = os.path.join(HERE, "_ref", "zvref")

ORDER_GGML_AVX2, ORDER_SEQ_F32, ORDER_SEQ_F64 = 0, 1, 2


def build(native: bool = False, out_dir: Optional[str] = None) -> str:
    """Compile the oracle library.  native=True builds a -march=native copy (for the timed CPU
    baseline on the GPU box's host) into out_dir; the default x86-64-v3 build goes to oracle/_build."""
    if not native:
        subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
        return LIB_PATH
    out_dir = out_dir or tempfile.gettempdir()
    path = os.path.join(out_dir, "libzvoracle_native.so")
    subprocess.run(["gcc", "-std=c11", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
                    "-o", path, os.path.join(HERE, "zv_oracle.c"), "-lm"], check=True)
    return path


class _EncParams(C.Structure):
    _fields_ = [("n_phonemes", C.c_int), ("max_seq_len", C.c_int), ("emb_dim", C.c_int), ("punct_emb_dim", C.c_int),
                ("n_layers", C.c_int), ("n_heads", C.c_int), ("ffn_kernel", C.c_int * 2), ("vp_kernel", C.c_int),
                ("ve_n_bins", C.c_int)]


def _load(path: str):
    lib = C.CDLL(path)
    lib.zvo_new.restype = C.c_void_p
    lib.zvo_free.argtypes = [C.c_void_p]
    lib.zvo_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    lib.zvo_set_order.argtypes = [C.c_void_p, C.c_int]
    lib.zvo_set_f16_inputs.argtypes = [C.c_void_p, C.c_int]
    lib.zvo_set_threads.argtypes = [C.c_void_p, C.c_int]
    lib.zvo_last_error.restype = C.c_char_p
    lib.zvo_vocoder.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.zvo_decoder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.zvo_encoder.argtypes = [C.c_void_p, C.POINTER(_EncParams)] + [C.c_void_p] * 11
    lib.zvo_length_regulator.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.zvo_conv1d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_void_p, C.c_void_p]
    lib.zvo_norm_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
    lib.zvo_layer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                              C.POINTER(C.c_int), C.c_void_p]
    return lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """CPU oracle bound to one set of weights ({gguf name: ndarray in numpy/C order})."""

    def __init__(self, tensors: Dict[str, np.ndarray], lib_path: Optional[str] = None, threads: int = 0,
                 order: int = ORDER_GGML_AVX2):
        path = lib_path or LIB_PATH
        if not os.path.exists(path):
            build()
        self.lib = _load(path)
        self.ctx = C.c_void_p(self.lib.zvo_new())
        self._keep = []
        for name, arr in tensors.items():
            arr = np.ascontiguousarray(arr)
            self._keep.append(arr)
            dt = {np.dtype(np.float32): 0, np.dtype(np.float16): 1}[arr.dtype]
            ne = (C.c_int64 * 4)(*(list(arr.shape[::-1]) + [1] * (4 - arr.ndim)))
            if self.lib.zvo_set_tensor(self.ctx, name.encode(), _p(arr), dt, arr.ndim, ne) != 0:
                raise RuntimeError(self.lib.zvo_last_error().decode())
        self.lib.zvo_set_order(self.ctx, order)
        # OpenMP's default is one thread per visible CPU; on a box whose CPU share is far below that (a GPU box shows
        # every host core but grants ~16) hundreds of spinning threads turn a millisecond call into half a minute.
        # Results do not depend on the thread count.
        if not threads:
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            threads = max(1, min(16, avail))
        self.lib.zvo_set_threads(self.ctx, threads)

    def __del__(self):
        try:
            self.lib.zvo_free(self.ctx)
        except Exception:
            pass

    def set_order(self, order: int):
        self.lib.zvo_set_order(self.ctx, order)

    def set_f16_inputs(self, on: bool):
        self.lib.zvo_set_f16_inputs(self.ctx, int(on))

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + self.lib.zvo_last_error().decode())

    def vocoder(self, mel: np.ndarray, hop: int = 300) -> np.ndarray:
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        T = mel.shape[0]
        wav = np.empty(T * hop, dtype=np.float32)
        self._chk(self.lib.zvo_vocoder(self.ctx, _p(mel), T, _p(wav)))
        return wav

    def decoder(self, hidden: np.ndarray, style: np.ndarray, num_mels: int = 80) -> np.ndarray:
        hidden = np.ascontiguousarray(hidden, dtype=np.float32)
        style = np.ascontiguousarray(style, dtype=np.float32)
        T = hidden.shape[0]
        mel = np.empty((T, num_mels), dtype=np.float32)
        self._chk(self.lib.zvo_decoder(self.ctx, _p(hidden), _p(style), T, _p(mel)))
        return mel

    def encoder(self, geom, ids, puncts, style, T: int, num_phonemes: Optional[int] = None) -> dict:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        puncts = np.ascontiguousarray(puncts, dtype=np.int32)
        style = np.ascontiguousarray(style, dtype=np.float32)
        N, E = len(ids), geom.E
        p = _EncParams(N, T, geom.emb_dim, geom.punct_emb_dim, geom.encoder_layer, geom.encoder_head,
                       (C.c_int * 2)(*geom.conv_kernel_size), geom.vp_kernel_size, geom.ve_n_bins)
        out = dict(hidden=np.empty((T, E), np.float32), features=np.empty((N, E), np.float32),
                   logdur=np.empty(N, np.float32), pitch=np.empty(N, np.float32), energy=np.empty(N, np.float32),
                   pitch_bucket=np.empty(N, np.int32), energy_bucket=np.empty(N, np.int32))
        nf = C.c_int32(0)
        self._chk(self.lib.zvo_encoder(self.ctx, C.byref(p), _p(ids), _p(puncts), _p(style), _p(out["hidden"]),
                                       C.cast(C.byref(nf), C.c_void_p), _p(out["features"]), _p(out["logdur"]),
                                       _p(out["pitch"]), _p(out["energy"]), _p(out["pitch_bucket"]),
                                       _p(out["energy_bucket"])))
        out["n_frames"] = int(nf.value)
        if num_phonemes is not None and num_phonemes < N:
            # FS2Encoder::eval(num_phonemes < max_n_phonemes): every token is encoded, the regulator walks the first
            # num_phonemes (reference src/fs2encoder.cpp:622)
            out["hidden"], out["n_frames"] = self.length_regulator(out["features"][:num_phonemes], out["logdur"][:num_phonemes], T)
        return out

    def length_regulator(self, features: np.ndarray, logdur: np.ndarray, T: int):
        features = np.ascontiguousarray(features, dtype=np.float32)
        logdur = np.ascontiguousarray(logdur, dtype=np.float32)
        N, E = features.shape
        hidden = np.empty((T, E), np.float32)
        nf = self.lib.zvo_length_regulator(_p(features), _p(logdur), N, E, T, _p(hidden))
        return hidden, nf

    def conv1d(self, x_cf: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], pad: int, dil: int) -> np.ndarray:
        """x_cf [IC][L] f32, w [OC][IC][K] f16 -> [OC][OL] f32"""
        x_cf = np.ascontiguousarray(x_cf, dtype=np.float32)
        w = np.ascontiguousarray(w, dtype=np.float16)
        IC, L = x_cf.shape
        OC, IC2, K = w.shape
        assert IC == IC2
        OL = L + 2 * pad - dil * (K - 1)
        out = np.empty((OC, OL), np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        self._chk(self.lib.zvo_conv1d(self.ctx, _p(x_cf), L, IC, _p(w), OC, K, pad, dil, _p(b), _p(out)))
        return out

    LAYER_VOC_RESBLOCK, LAYER_ENC_FFT, LAYER_DEC_BLOCK, LAYER_VAR_PRED = 0, 1, 2, 3
    LAYER_VOC_UPSAMPLE, LAYER_VOC_INPUT, LAYER_VOC_OUTPUT, LAYER_DEC_ASR_RES, LAYER_DEC_TO_OUT, LAYER_ENC_EMBED = 4, 5, 6, 7, 8, 9
    LAYER_ENC_MHA, LAYER_ENC_FFN, LAYER_DEC_ADAIN = 10, 11, 12

    def layer(self, kind: int, index: int, x: np.ndarray, out_cols: int, style=None, heads: int = 2, ksz=(9, 1),
              out_rows: Optional[int] = None) -> np.ndarray:
        """one layer of the reference semantics on a given input (time-major [rows][cols]): HiFi-GAN residual block,
        FFT block, decoder residual block, variance predictor, transposed conv (out_rows = rows x scale), vocoder input /
        output conv, asr_res, to_out, embedding (out_cols = 0 -> a vector of out_rows values)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        rows, cols = x.shape
        orows = rows if out_rows is None else out_rows
        out = np.empty((orows, out_cols) if out_cols else (orows,), np.float32)
        st = None if style is None else np.ascontiguousarray(style, dtype=np.float32)
        k = (C.c_int * 2)(*(list(ksz) + [1])[:2])
        self._chk(self.lib.zvo_layer(self.ctx, kind, index, _p(x), rows, cols, _p(st), 0 if st is None else len(st), heads, k, _p(out)))
        return out

    def norm_rows(self, x: np.ndarray, eps: float = 1e-5) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        self.lib.zvo_norm_rows(_p(x), x.shape[0], x.shape[1], eps, _p(y))
        return y


# ---------------------------------------------------------------------------------------------
# compiled reference (oracle/_ref/zvref) — exists only where oracle/Makefile `ref` was run

def have_reference() -> bool:
    return os.path.exists(REF_BIN)


def run_reference_chain(gguf_path: str, ids, puncts, style, *, T: int, threads: int = 4, num_phonemes: Optional[int] = None,
                        num_mels: int = 80, hop: int = 300) -> dict:
    """The three stage evals back to back on one utterance, as ZeroVOXModel::eval does (reference src/zerovox.cpp:326-334):
    the decoder reads the encoder's hidden, the vocoder the decoder's mel.  Returns every stage's outputs + timings."""
    return run_reference(gguf_path, N=len(ids), T=T, threads=threads, enc=(ids, puncts, style), dec=(None, style), voc=None,
                         chain=True, num_phonemes=num_phonemes, num_mels=num_mels, hop=hop)


def run_reference(gguf_path: str, *, N: Optional[int] = None, T: Optional[int] = None, threads: int = 4, reps: int = 1,
                  enc=None, dec=None, voc=None, E: Optional[int] = None, num_mels: int = 80, hop: int = 300,
                  chain: bool = False, num_phonemes: Optional[int] = None) -> dict:
    """enc=(ids, puncts, style), dec=(hidden[T,E], style), voc=mel[T,80].  Returns outputs + timings.
    chain=True: dec / voc inputs come from the previous stage (their input arrays may be None)."""
    if not have_reference():
        raise RuntimeError("oracle/_ref/zvref not built (run `make -C oracle ref` where /root/reference exists)")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        cmd = [REF_BIN, gguf_path, "--threads", str(threads), "--reps", str(reps)]
        if chain:
            cmd += ["--chain"]
        if num_phonemes is not None:
            cmd += ["--num", str(num_phonemes)]
        if N is not None:
            cmd += ["--N", str(N)]
        if T is not None:
            cmd += ["--T", str(T)]
        if enc is not None:
            ids, puncts, style = enc
            np.asarray(ids, np.int32).tofile(td + "/ids")
            np.asarray(puncts, np.int32).tofile(td + "/puncts")
            np.asarray(style, np.float32).tofile(td + "/style_e")
            cmd += ["--enc", td + "/ids", td + "/puncts", td + "/style_e", td + "/enc"]
        if dec is not None:
            hidden, style = dec
            if hidden is not None:
                np.asarray(hidden, np.float32).tofile(td + "/hidden")
            np.asarray(style, np.float32).tofile(td + "/style_d")
            cmd += ["--dec", td + "/hidden", td + "/style_d", td + "/mel_out"]
        if voc is not None or chain:
            if voc is not None:
                np.asarray(voc, np.float32).tofile(td + "/mel_in")
            cmd += ["--voc", td + "/mel_in", td + "/wav_out"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"zvref failed ({r.returncode}): {r.stderr[-2000:]}")
        out["timing"] = json.loads(r.stderr.strip().splitlines()[-1])
        if enc is not None:
            n = len(enc[0])
            e = E if E is not None else len(enc[2])
            out["hidden"] = np.fromfile(td + "/enc.hidden.f32", np.float32).reshape(-1, e)
            out["features"] = np.fromfile(td + "/enc.features.f32", np.float32).reshape(n, e)
            out["logdur"] = np.fromfile(td + "/enc.logdur.f32", np.float32)
            out["pitch"] = np.fromfile(td + "/enc.pitch.f32", np.float32)
            out["energy"] = np.fromfile(td + "/enc.energy.f32", np.float32)
            out["pitch_bucket"] = np.fromfile(td + "/enc.pitch_bucket.i32", np.int32)
            out["energy_bucket"] = np.fromfile(td + "/enc.energy_bucket.i32", np.int32)
            out["n_frames"] = int(np.fromfile(td + "/enc.nframes.i32", np.int32)[0])
        if dec is not None:
            out["mel"] = np.fromfile(td + "/mel_out", np.float32).reshape(-1, num_mels)
        if voc is not None or chain:
            out["wav"] = np.fromfile(td + "/wav_out", np.float32)
    return out
