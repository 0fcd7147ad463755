"""Checkpoint converter: ZeroVox training checkpoint (+ HiFi-GAN generator + mel statistics) -> GGUF v3.

SURVEY.md §8(f) rank 1.  Produces the file format `zv_model_load` and the reference loader
(src/zerovox.cpp:21-179) read, applying the same transforms as the reference's converter
(utils/zv2gguf.py), restated here from their description:

  * tensor names: ordered substring shortening (utils/zv2gguf.py:22-39)
  * 0-dim tensors are dropped (:150-152)
  * `pos_ffn.w_1.w`, `pos_ffn.w_2.w` and `*conv.w` are stored as F16 (:154-159)
  * torch weight-norm pairs (`weight_g`, `weight_v`) are folded into one tensor
    w = g * v / ||v||  (norm over every dim but 0, i.e. torch._weight_norm(v, g, 0)) and stored as F16
    under the name  <key with `weight_v` -> `w`>  — the name shortening is NOT applied to these (:162-173, 180)
  * the folded weight of a transposed conv (`_meldec.upsamples.N.1.w`, torch layout (in, out, k)) is flipped
    along k and permuted to (out, in, k): the runtime evaluates ConvTranspose1d as zero-stuffing + plain
    conv (:175-178, src/hifigan.cpp:22-71)
  * `_meldec.*` tensors of the acoustic checkpoint are replaced by the stand-alone HiFi-GAN generator (:96-107)
  * `hifigan.mean`, `hifigan.scale` from the vocoder's stats file (:68-73, 141-142)
  * `sinusoid_encoding_table` [(max_seq_len + 1), E] (:41-62, 184-185)
  * the 15 u32 hyper-parameters (:119-139)

Inputs are plain mappings name -> array (numpy arrays or torch tensors), so the module needs neither the
author's directory layout nor `gguf`/`h5py`; `main()` adds the file handling for a real model directory.
"""
from __future__ import annotations

import re
from typing import Dict, List, Mapping, Optional, Tuple

import numpy as np

from .gguf import write_gguf
from .synth import ARCH, sinusoid_table

# applied in this order to every occurrence (the first entry must run before the second: it contains it)
SHORTNAMES: Tuple[Tuple[str, str], ...] = (
    ("_phoneme_encoder", "_pe"),
    ("_encoder", "_enc"),
    ("layer_stack", "laystk"),
    ("weight", "w"),
    ("_variance_adaptor", "_var_adapt"),
    ("energy_predictor", "engy_pred"),
    ("bias", "b"),
)

_F16_SUFFIXES = ("pos_ffn.w_1.w", "pos_ffn.w_2.w", "conv.w")
_DECONV_RE = re.compile(r"^_meldec\.upsamples\.[0-9]\.1\.w$")


def shorten_tensor_name(long_name: str) -> str:
    s = long_name
    for long, short in SHORTNAMES:
        s = s.replace(long, short)
    return s


def _np(t) -> np.ndarray:
    if hasattr(t, "detach"):          # torch tensor
        t = t.detach().cpu().numpy()
    return np.asarray(t)


def fold_weight_norm(v, g) -> np.ndarray:
    """w = g * v / ||v||, the norm taken over every dim except 0 (torch.nn.utils.weight_norm's default dim).
    Uses torch's own kernel when torch is importable so that the f32 result — and therefore its F16 rounding —
    is the one the reference converter stores; the numpy path differs from it by at most one f32 ulp."""
    v32, g32 = _np(v).astype(np.float32), _np(g).astype(np.float32)
    try:
        import torch
        return torch._weight_norm(torch.from_numpy(np.ascontiguousarray(v32)), torch.from_numpy(np.ascontiguousarray(g32)), 0).numpy()
    except ImportError:
        axes = tuple(range(1, v32.ndim))
        norm = np.sqrt(np.sum(v32 * v32, axis=axes, keepdims=True, dtype=np.float32))
        return (v32 * (g32.reshape(norm.shape) / norm)).astype(np.float32)


def deconv_to_conv_weight(w: np.ndarray) -> np.ndarray:
    """torch ConvTranspose1d weight (in, out, k) -> the (out, in, k) kernel of the equivalent plain conv over the
    zero-stuffed input: flip along k, swap the channel axes."""
    return np.ascontiguousarray(np.flip(w, axis=2).transpose(1, 0, 2))


def hyperparameters(cfg: Mapping) -> Dict[str, int]:
    m, a = cfg["model"], cfg["audio"]
    p = ARCH + "."
    return {
        p + "max_seq_len": int(m["max_seq_len"]),
        p + "emb_dim": int(m["emb_dim"]),
        p + "punct_emb_dim": int(m["punct_emb_dim"]),
        p + "decoder.n_head": int(m["decoder"]["n_head"]),
        p + "encoder.layer": int(m["encoder"]["fs2_layer"]),
        p + "encoder.head": int(m["encoder"]["fs2_head"]),
        p + "encoder.vp_filter_size": int(m["encoder"]["vp_filter_size"]),
        p + "encoder.vp_kernel_size": int(m["encoder"]["vp_kernel_size"]),
        p + "encoder.ve_n_bins": int(m["encoder"]["ve_n_bins"]),
        p + "decoder.conv_filter_size": int(m["decoder"]["conv_filter_size"]),
        p + "decoder.conv_kernel_size.0": int(m["decoder"]["conv_kernel_size"][0]),
        p + "decoder.conv_kernel_size.1": int(m["decoder"]["conv_kernel_size"][1]),
        p + "audio.sampling_rate": int(a["sampling_rate"]),
        p + "audio.num_mels": int(a["num_mels"]),
        p + "audio.hop_size": int(a["hop_size"]),
    }


def convert_tensors(state_dict: Mapping[str, object], cfg: Mapping, stats: Mapping[str, object],
                    meldec_generator: Optional[Mapping[str, object]] = None) -> List[Tuple[str, np.ndarray]]:
    """The tensor list of the GGUF file, in file order: statistics, the state dict in sorted key order,
    the position table."""
    sd = dict(state_dict)
    if meldec_generator is not None:
        for k in [k for k in sd if k.startswith("_meldec.")]:
            del sd[k]
        for k, t in meldec_generator.items():
            sd["_meldec." + k] = t

    out: List[Tuple[str, np.ndarray]] = [
        ("hifigan.mean", _np(stats["mean"]).astype(np.float32)),
        ("hifigan.scale", _np(stats["scale"]).astype(np.float32)),
    ]
    for key in sorted(sd):
        if key.endswith("weight_g"):
            continue
        if key.endswith("weight_v"):
            g_key = key.replace(".weight_v", ".weight_g")
            if g_key not in sd:
                raise KeyError(f"{key}: weight-norm magnitude {g_key} is missing")
            w = fold_weight_norm(sd[key], sd[g_key])
            name = key.replace("weight_v", "w")
            if _DECONV_RE.match(name):
                w = deconv_to_conv_weight(w)
            out.append((name, np.ascontiguousarray(w.astype(np.float16))))
            continue
        t = _np(sd[key])
        if t.ndim == 0:
            continue
        name = shorten_tensor_name(key)
        if name.endswith(_F16_SUFFIXES):
            t = t.astype(np.float16)
        out.append((name, np.ascontiguousarray(t)))
    m = cfg["model"]
    out.append(("sinusoid_encoding_table", sinusoid_table(int(m["max_seq_len"]) + 1, int(m["emb_dim"]) + int(m["punct_emb_dim"]))))
    for name, _ in out:
        if len(name.encode()) >= 64:            # GGML_MAX_NAME, ggml/include/ggml.h:224-225
            raise ValueError(f"tensor name too long for GGUF: {name}")
    return out


def convert(state_dict, cfg, stats, out_path: str, meldec_generator=None) -> List[Tuple[str, np.ndarray]]:
    tensors = convert_tensors(state_dict, cfg, stats, meldec_generator)
    write_gguf(out_path, hyperparameters(cfg), tensors, arch=ARCH)
    return tensors


def _load_stats(path: str) -> Dict[str, np.ndarray]:
    if path.endswith(".npz"):
        z = np.load(path)
        return {"mean": z["mean"], "scale": z["scale"]}
    try:
        import h5py
    except ImportError as e:
        raise SystemExit(f"{path}: reading HDF5 statistics needs h5py, which is not installed — "
                         "export mean/scale to an .npz and pass that instead") from e
    with h5py.File(path, "r") as f:
        return {"mean": f["mean"][:], "scale": f["scale"][:]}


def main(argv=None) -> int:
    import argparse
    import glob
    import os

    ap = argparse.ArgumentParser(description="ZeroVox checkpoint -> GGUF (same transforms as the reference's utils/zv2gguf.py)")
    ap.add_argument("--model-dir", required=True, help="directory with modelcfg.yaml and checkpoints/*.ckpt")
    ap.add_argument("--hifigan-dir", required=True, help="directory with checkpoint.pkl and stats.h5 (or stats.npz)")
    ap.add_argument("-o", "--out", default="medium-ldec.gguf")
    a = ap.parse_args(argv)

    import torch
    import yaml

    with open(os.path.join(a.model_dir, "modelcfg.yaml")) as f:
        cfg = yaml.safe_load(f)
    ckpts = glob.glob(os.path.join(a.model_dir, "checkpoints", "*.ckpt"))
    if not ckpts:
        raise SystemExit(f"no checkpoints/*.ckpt under {a.model_dir}")
    ckpt = max(ckpts, key=os.path.getctime)
    state_dict = torch.load(ckpt, map_location="cpu", weights_only=False)["state_dict"]
    generator = torch.load(os.path.join(a.hifigan_dir, "checkpoint.pkl"), map_location="cpu", weights_only=False)["model"]["generator"]
    stats_path = os.path.join(a.hifigan_dir, "stats.h5")
    if not os.path.exists(stats_path):
        stats_path = os.path.join(a.hifigan_dir, "stats.npz")
    tensors = convert(state_dict, cfg, _load_stats(stats_path), a.out, meldec_generator=generator)
    print(f"{a.out}: {len(tensors)} tensors from {ckpt}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
