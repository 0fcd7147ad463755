"""Deterministic synthetic ZeroVox checkpoints with the reference's exact weight contract.

No real checkpoint exists offline, so every parity case runs on a synthetic GGUF
whose tensor names / shapes / dtypes are those the reference's stage classes look up
(src/fs2encoder.cpp:29-62,152-171,344-382,504-505; src/stylettsdec.cpp:33-66,163-168,220-239,
334-340; src/hifigan.cpp:34-39,84-95,208-218) and whose KV keys are those of src/zerovox.h:17-33.

Values come from a counter-based integer hash (splitmix64) -> float, so a (geometry, seed) pair
produces bit-identical files on every machine: the GPU box regenerates the checkpoints the golden
fixtures were made from instead of shipping 193 MB files.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

ARCH = "zerovox-resnet-fs2-styletts"
NUM_PHONEMES = 154          # reference src/zerovox.h:35
NUM_PUNCTS = 6              # reference src/zerovox.h:36

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u01(seed: int, name: str, n: int) -> np.ndarray:
    """n uniform doubles in [0,1) that depend only on (seed, name, index)."""
    base = np.uint64((_fnv1a64(name) ^ (seed * 0x2545F4914F6CDD1D)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    return (_splitmix64(idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def sym(seed: int, name: str, shape, amp: float) -> np.ndarray:
    n = int(np.prod(shape))
    return ((u01(seed, name, n) * 2.0 - 1.0) * amp).astype(np.float32).reshape(shape)


def normal(seed: int, name: str, shape, std: float = 1.0) -> np.ndarray:
    """Box-Muller on two hashed uniforms (float64 math, cast to f32)."""
    n = int(np.prod(shape))
    a = u01(seed, name + "/a", n)
    b = u01(seed, name + "/b", n)
    z = np.sqrt(-2.0 * np.log(1.0 - a)) * np.cos(2.0 * np.pi * b)
    return (z * std).astype(np.float32).reshape(shape)


@dataclass
class Geometry:
    name: str = "medium"
    emb_dim: int = 512
    punct_emb_dim: int = 16
    conv_filter_size: int = 1024            # encoder FFN width (KV key says "decoder.", src/zerovox.cpp:111)
    conv_kernel_size: Tuple[int, int] = (9, 1)
    encoder_layer: int = 4
    encoder_head: int = 2
    vp_filter_size: int = 256
    vp_kernel_size: int = 3
    ve_n_bins: int = 256
    decoder_n_head: int = 2                 # read by the reference, unused
    max_seq_len: int = 1500
    num_mels: int = 80
    hop_size: int = 300
    sampling_rate: int = 22050
    residual_dim: int = 64                  # hard-coded in the reference (src/zerovox.cpp:124)
    voc_channels: int = 512
    voc_kernel_size: int = 7
    upsample_scales: Tuple[int, ...] = (5, 5, 4, 3)        # hard-coded, src/zerovox.cpp:129
    upsample_kernels: Tuple[int, ...] = (10, 10, 8, 6)
    resblock_kernels: Tuple[int, ...] = (3, 7, 11)
    resblock_dilations: Tuple[int, ...] = (1, 3, 5)        # hard-coded, src/zerovox.cpp:132

    @property
    def E(self) -> int:
        return self.emb_dim + self.punct_emb_dim

    def kv(self) -> Dict[str, int]:
        p = ARCH + "."
        return {
            p + "max_seq_len": self.max_seq_len,
            p + "emb_dim": self.emb_dim,
            p + "punct_emb_dim": self.punct_emb_dim,
            p + "decoder.n_head": self.decoder_n_head,
            p + "encoder.layer": self.encoder_layer,
            p + "encoder.head": self.encoder_head,
            p + "encoder.vp_filter_size": self.vp_filter_size,
            p + "encoder.vp_kernel_size": self.vp_kernel_size,
            p + "encoder.ve_n_bins": self.ve_n_bins,
            p + "decoder.conv_filter_size": self.conv_filter_size,
            p + "decoder.conv_kernel_size.0": self.conv_kernel_size[0],
            p + "decoder.conv_kernel_size.1": self.conv_kernel_size[1],
            p + "audio.sampling_rate": self.sampling_rate,
            p + "audio.num_mels": self.num_mels,
            p + "audio.hop_size": self.hop_size,
        }


MEDIUM = Geometry()
# Same topology, every width shrunk; channel counts deliberately NOT multiples of 16/32 so the
# kernels' padding paths are exercised (SURVEY.md §8c "tiny-geometry GGUF with identical topology").
TINY = Geometry(name="tiny", emb_dim=48, punct_emb_dim=16, conv_filter_size=72, encoder_layer=2,
                encoder_head=2, vp_filter_size=40, ve_n_bins=16, max_seq_len=64, voc_channels=48)
# A mid-size geometry: big enough that every MFMA tile shape is hit, small enough for quick CPU runs.
SMALL = Geometry(name="small", emb_dim=112, punct_emb_dim=16, conv_filter_size=256, encoder_layer=2,
                 encoder_head=2, vp_filter_size=64, ve_n_bins=64, max_seq_len=256, voc_channels=128)

# MEDIUM with 8 pitch / energy bins (`encoder.ve_n_bins` is a free KV of the checkpoint format): a bin is 1/7 wide, 300 times
# the summation-order noise of the predictions, so an utterance whose predictions all sit >= 0.1 bin from a boundary makes
# the same integer decisions under any re-association — the un-forced end-to-end golden (tests/golden/make_golden.py)
MEDIUM8 = Geometry(name="medium8", ve_n_bins=8)

GEOMETRIES = {g.name: g for g in (MEDIUM, TINY, SMALL, MEDIUM8)}


def sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """Same formula as utils/zv2gguf.py:41-62 (angles evaluated in float64, table cast to f32
    BEFORE sin/cos are applied in f32 — the converter builds the angle array with dtype=float32
    and then overwrites slices with np.sin/np.cos of those f32 angles)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    ang = (pos / np.power(10000.0, 2 * (j // 2) / d_hid)).astype(np.float32)
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return ang


def make_tensors(g: Geometry, seed: int) -> List[Tuple[str, np.ndarray]]:
    """All tensors of SURVEY.md Appx A in numpy (C-order) shape = reversed ggml ne."""
    E, F = g.E, g.conv_filter_size
    T: List[Tuple[str, np.ndarray]] = []

    def add(name, arr):
        T.append((name, np.ascontiguousarray(arr)))

    def conv_w(name, oc, ic, k, gain=1.0):           # f16, ggml ne [k, ic, oc]
        amp = gain * np.sqrt(3.0 / (ic * k))
        add(name, sym(seed, name, (oc, ic, k), amp).astype(np.float16))

    def lin_w(name, out, inp, gain=1.0):             # f32, ggml ne [in, out]
        add(name, sym(seed, name, (out, inp), gain * np.sqrt(3.0 / inp)))

    def vec(name, n, amp=0.1, center=0.0):
        add(name, (sym(seed, name, (n,), amp) + np.float32(center)).astype(np.float32))

    # ---- vocoder statistics
    vec("hifigan.mean", g.num_mels, amp=1.0, center=-1.0)
    vec("hifigan.scale", g.num_mels, amp=0.5, center=1.0)

    # ---- FastSpeech2 encoder
    add("sinusoid_encoding_table", sinusoid_table(g.max_seq_len + 1, E))
    add("_pe._enc.src_word_emb.w", sym(seed, "_pe._enc.src_word_emb.w", (NUM_PHONEMES + 1, g.emb_dim), 1.0))
    add("_pe._enc.punct_embed.w", sym(seed, "_pe._enc.punct_embed.w", (NUM_PUNCTS + 1, g.punct_emb_dim), 1.0))
    for i in range(g.encoder_layer):
        p = f"_pe._enc.laystk.{i}."
        for nm in ("w_qs", "w_ks", "w_vs", "fc"):
            lin_w(p + f"slf_attn.{nm}.w", E, E)
            vec(p + f"slf_attn.{nm}.b", E)
        vec(p + "slf_attn.layer_norm.w", E, center=1.0)
        vec(p + "slf_attn.layer_norm.b", E)
        conv_w(p + "pos_ffn.w_1.w", F, E, g.conv_kernel_size[0], gain=1.4)
        vec(p + "pos_ffn.w_1.b", F)
        conv_w(p + "pos_ffn.w_2.w", E, F, g.conv_kernel_size[1], gain=1.4)
        vec(p + "pos_ffn.w_2.b", E)
        vec(p + "pos_ffn.layer_norm.w", E, center=1.0)
        vec(p + "pos_ffn.layer_norm.b", E)
    V = g.vp_filter_size
    for pred, lin_gain, lin_bias in (("duration_predictor", 0.25, 1.6), ("pitch_predictor", 0.15, 0.5),
                                     ("engy_pred", 0.15, 0.5)):
        p = f"_pe._var_adapt.{pred}."
        conv_w(p + "conv_layer.conv1d_1.conv.w", V, E, g.vp_kernel_size, gain=1.4)
        vec(p + "conv_layer.conv1d_1.conv.b", V)
        vec(p + "conv_layer.layer_norm_1.w", V, center=1.0)
        vec(p + "conv_layer.layer_norm_1.b", V)
        conv_w(p + "conv_layer.conv1d_2.conv.w", V, V, g.vp_kernel_size, gain=1.4)
        vec(p + "conv_layer.conv1d_2.conv.b", V)
        vec(p + "conv_layer.layer_norm_2.w", V, center=1.0)
        vec(p + "conv_layer.layer_norm_2.b", V)
        lin_w(p + "linear_layer.w", 1, V, gain=lin_gain)
        add(p + "linear_layer.b", np.array([lin_bias], dtype=np.float32))
    add("_pe._var_adapt.pitch_embedding.w", sym(seed, "pitch_embedding", (g.ve_n_bins, E), 0.5))
    add("_pe._var_adapt.energy_embedding.w", sym(seed, "energy_embedding", (g.ve_n_bins, E), 0.5))

    # ---- StyleTTS mel decoder
    R = g.residual_dim
    for i, (ci, co) in enumerate(((E, 2 * E), (2 * E, 2 * E))):
        p = f"_mel_decoder.encode.{i}."
        conv_w(p + "conv1.w", ci, ci, 3, gain=1.4)
        vec(p + "conv1.b", ci)
        conv_w(p + "conv2.w", co, ci, 3, gain=1.4)
        vec(p + "conv2.b", co)
        if ci != co:
            conv_w(p + "conv1x1.w", co, ci, 1)
        for n in ("norm1", "norm2"):
            vec(p + n + ".w", ci, center=1.0)
            vec(p + n + ".b", ci)
    conv_w("_mel_decoder.asr_res.0.w", R, E, 1)
    vec("_mel_decoder.asr_res.0.b", R)
    vec("_mel_decoder.asr_res.1.w", R, center=1.0)
    vec("_mel_decoder.asr_res.1.b", R)
    dims = ((2 * E + R, 2 * E), (2 * E + R, 2 * E), (2 * E + R, E), (E, E), (E, E))
    for i, (ci, co) in enumerate(dims):
        p = f"_mel_decoder.decode.{i}."
        lin_w(p + "norm1.fc.w", 2 * ci, E)
        vec(p + "norm1.fc.b", 2 * ci)
        lin_w(p + "norm2.fc.w", 2 * co, E)
        vec(p + "norm2.fc.b", 2 * co)
        conv_w(p + "conv1.w", co, ci, 3, gain=1.4)
        vec(p + "conv1.b", co)
        conv_w(p + "conv2.w", co, co, 3, gain=1.4)
        vec(p + "conv2.b", co)
        if ci != co:
            conv_w(p + "conv1x1.w", co, ci, 1)
    conv_w("_mel_decoder.to_out.0.w", g.num_mels, E, 1)
    vec("_mel_decoder.to_out.0.b", g.num_mels)

    # ---- HiFi-GAN vocoder (channels halve per upsample stage).  Gains are chosen so that the waveform has a
    # speech-like level (RMS ~0.1, peaks ~0.35): the reference's re-association noise floor is a fixed
    # *relative* ~4.5e-4 of the signal (f16 operand rounding, SURVEY.md Appx D), so the absolute 1e-4 RMS
    # gate only means something at a realistic amplitude.
    C = g.voc_channels
    conv_w("_meldec.input_conv.w", C, g.num_mels, g.voc_kernel_size)
    vec("_meldec.input_conv.b", C)
    for i, (s, k) in enumerate(zip(g.upsample_scales, g.upsample_kernels)):
        ci, co = C >> i, C >> (i + 1)
        # stored already flipped + permuted to (out, in, k) (utils/zv2gguf.py:175-178); a transposed
        # conv of stride s uses k/s taps per output sample, hence fan-in ci*k/s
        amp_gain = np.sqrt(s)
        conv_w(f"_meldec.upsamples.{i}.1.w", co, ci, k, gain=1.0 * amp_gain)
        vec(f"_meldec.upsamples.{i}.1.b", co)
        for j, rk in enumerate(g.resblock_kernels):
            n = i * len(g.resblock_kernels) + j
            for d in range(len(g.resblock_dilations)):
                conv_w(f"_meldec.blocks.{n}.convs1.{d}.1.w", co, co, rk, gain=1.0)
                vec(f"_meldec.blocks.{n}.convs1.{d}.1.b", co)
                conv_w(f"_meldec.blocks.{n}.convs2.{d}.1.w", co, co, rk, gain=0.2)
                vec(f"_meldec.blocks.{n}.convs2.{d}.1.b", co)
    cl = C >> len(g.upsample_scales)
    conv_w("_meldec.output_conv.1.w", 1, cl, g.voc_kernel_size, gain=0.5)
    vec("_meldec.output_conv.1.b", 1, amp=0.01)
    return T


def write_checkpoint(path: str, g: Geometry, seed: int, trim_dims: bool = False) -> None:
    from .gguf import write_gguf
    write_gguf(path, g.kv(), make_tensors(g, seed), arch=ARCH, trim_dims=trim_dims)


# ---- seeded synthetic inputs -----------------------------------------

def encoder_inputs(g: Geometry, seed: int, n_phonemes: int):
    ids = 1 + (u01(seed, "ids", n_phonemes) * NUM_PHONEMES).astype(np.int32)          # U{1..154}
    puncts = (u01(seed, "puncts", n_phonemes) * (NUM_PUNCTS + 1)).astype(np.int32)    # U{0..6}
    style = normal(seed, "style", (g.E,), 0.05)
    return np.minimum(ids, NUM_PHONEMES).astype(np.int32), np.minimum(puncts, NUM_PUNCTS).astype(np.int32), style


def vocoder_mel(g: Geometry, tensors: Dict[str, np.ndarray], seed: int, T: int) -> np.ndarray:
    """mel[T, num_mels] = mean + scale * N(0,1) so the normalised vocoder input is N(0,1)."""
    z = normal(seed, "mel", (T, g.num_mels), 1.0)
    return (tensors["hifigan.mean"][None, :] + tensors["hifigan.scale"][None, :] * z).astype(np.float32)


def decoder_hidden(g: Geometry, seed: int, T: int, frames_per_phoneme: int = 4, fill: float = 0.8) -> np.ndarray:
    """Piece-wise constant hidden[T, E] (what the length regulator emits) with a zero-padded tail."""
    n_used = int(T * fill)
    n_ph = (n_used + frames_per_phoneme - 1) // frames_per_phoneme
    feat = normal(seed, "hidden", (n_ph, g.E), 1.5)
    h = np.zeros((T, g.E), dtype=np.float32)
    h[:n_used] = np.repeat(feat, frames_per_phoneme, axis=0)[:n_used]
    return h
