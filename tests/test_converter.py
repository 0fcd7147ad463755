"""checkpoint converter equivalence.

The reference's utils/zv2gguf.py cannot run here (needs `gguf`, `h5py` and the author's checkpoints), so the
converter is checked against what its transforms MEAN: a training-style checkpoint (long names, torch layouts,
weight-norm pairs) is built for the tiny geometry, converted, and the result must (a) have exactly the tensor
inventory of SURVEY.md Appx A that the loader accepts, (b) carry folded weights equal to torch's own weight-norm,
and (c) make zero-stuffing + plain conv with the converted kernel equal torch's ConvTranspose1d."""
import os

import numpy as np
import pytest

LONG = {"_pe": "_phoneme_encoder", "_enc": "_encoder", "laystk": "layer_stack", "w": "weight", "b": "bias",
        "_var_adapt": "_variance_adaptor", "engy_pred": "energy_predictor"}


def _lengthen(short):
    return ".".join(LONG.get(c, c) for c in short.split("."))


def _cfg(g):
    return {"model": {"max_seq_len": g.max_seq_len, "emb_dim": g.emb_dim, "punct_emb_dim": g.punct_emb_dim,
                      "decoder": {"n_head": g.decoder_n_head, "conv_filter_size": g.conv_filter_size,
                                  "conv_kernel_size": list(g.conv_kernel_size)},
                      "encoder": {"fs2_layer": g.encoder_layer, "fs2_head": g.encoder_head,
                                  "vp_filter_size": g.vp_filter_size, "vp_kernel_size": g.vp_kernel_size,
                                  "ve_n_bins": g.ve_n_bins}},
            "audio": {"sampling_rate": g.sampling_rate, "num_mels": g.num_mels, "hop_size": g.hop_size}}


def _training_checkpoint(g, seed):
    """(state_dict, generator, stats, expected) in the shapes a torch training run leaves behind"""
    from zerovox_cpp_amd import synth
    rng = np.random.default_rng(seed)
    sd, gen, expected = {}, {}, {}
    stats = {}
    for name, arr in synth.make_tensors(g, seed):
        if name in ("hifigan.mean", "hifigan.scale"):
            stats[name.split(".")[1]] = arr.astype(np.float64)        # HDF5 statistics are float64 upstream
            expected[name] = arr
            continue
        if name == "sinusoid_encoding_table":
            expected[name] = arr
            continue
        weight_normed = arr.ndim == 3 and (name.startswith("_meldec.") or name.startswith("_mel_decoder."))
        target = gen if name.startswith("_meldec.") else sd
        key = name[len("_meldec."):] if name.startswith("_meldec.") else name
        if weight_normed:
            oc, ic, k = arr.shape
            deconv = ".upsamples." in name
            v = rng.standard_normal((ic, oc, k) if deconv else (oc, ic, k)).astype(np.float32)
            gm = (0.5 + rng.random((v.shape[0], 1, 1))).astype(np.float32)
            assert key.endswith(".w")
            target[key[:-1] + "weight_v"] = v
            target[key[:-1] + "weight_g"] = gm
            norm = np.sqrt(np.sum(v.astype(np.float64) ** 2, axis=(1, 2), keepdims=True))
            w = (v.astype(np.float64) * (gm.astype(np.float64) / norm)).astype(np.float32)
            if deconv:
                w = np.flip(w, 2).transpose(1, 0, 2)
            expected[name] = w                                        # f32; the file holds its f16 rounding
        else:
            long = _lengthen(key)
            target[long] = arr.astype(np.float32)                     # training keeps everything in f32
            expected[name] = arr
    sd["_phoneme_encoder._encoder.position_enc_steps"] = np.float32(3.0)          # 0-dim: dropped
    sd["_meldec.stale.weight"] = np.zeros((2, 2), np.float32)                     # replaced by the generator
    return sd, gen, stats, expected


def test_name_shortening_covers_the_inventory():
    from zerovox_cpp_amd import convert, synth
    for name, arr in synth.make_tensors(synth.TINY, 1):
        if name.startswith(("hifigan.", "sinusoid")):
            continue
        assert convert.shorten_tensor_name(_lengthen(name)) == name, name
    assert convert.shorten_tensor_name("_phoneme_encoder._encoder.layer_stack.0.slf_attn.w_qs.weight") == "_pe._enc.laystk.0.slf_attn.w_qs.w"
    assert convert.shorten_tensor_name("_phoneme_encoder._variance_adaptor.energy_predictor.linear_layer.bias") == \
        "_pe._var_adapt.engy_pred.linear_layer.b"


def test_converted_checkpoint_matches_the_loader_inventory(tmp_path):
    from zerovox_cpp_amd import capi, convert, gguf, synth
    g = synth.TINY
    sd, gen, stats, expected = _training_checkpoint(g, 77)
    out = str(tmp_path / "conv.gguf")
    tensors = convert.convert(sd, _cfg(g), stats, out, meldec_generator=gen)
    names = [n for n, _ in tensors]
    assert names[:2] == ["hifigan.mean", "hifigan.scale"] and names[-1] == "sinusoid_encoding_table"
    inv = dict(synth.make_tensors(g, 77))
    assert set(names) == set(inv), set(names) ^ set(inv)
    kv, read = gguf.read_gguf(out)
    for k, v in g.kv().items():
        assert kv[k] == v
    for name, ref in inv.items():
        got = read[name]
        assert got.shape == ref.shape and got.dtype == ref.dtype, (name, got.shape, ref.shape, got.dtype, ref.dtype)
        exp = expected[name]
        if exp.dtype == np.float32 and ref.dtype == np.float16:
            # weight-norm fold: f16 rounding of an f32 product; the f64 restatement may sit one f32 ulp away
            d = np.abs(got.astype(np.float32) - exp.astype(np.float16).astype(np.float32))
            ulp = np.spacing(np.abs(exp).astype(np.float16)).astype(np.float32)
            assert np.all(d <= ulp), name
            assert np.mean(d == 0) > 0.999, name
        else:
            assert np.array_equal(got, exp.astype(ref.dtype)), name
    # the C++ reader (the loader half of zv_model_load) accepts the file and sees the same inventory
    n, msl = capi.gguf_inspect(out)[:2]
    assert n == len(inv) and msl == g.max_seq_len


def test_weight_norm_fold_equals_torch():
    import torch
    from zerovox_cpp_amd import convert
    rng = np.random.default_rng(5)
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 10, 7, padding=3))
    with torch.no_grad():
        conv.weight_v.copy_(torch.from_numpy(rng.standard_normal((10, 6, 7)).astype(np.float32)))
        conv.weight_g.copy_(torch.from_numpy((0.5 + rng.random((10, 1, 1))).astype(np.float32)))
    w = convert.fold_weight_norm(conv.weight_v, conv.weight_g)
    x = torch.from_numpy(rng.standard_normal((1, 6, 33)).astype(np.float32))
    ref = conv(x)
    got = torch.nn.functional.conv1d(x, torch.from_numpy(w), conv.bias, padding=3)
    assert torch.allclose(ref, got, atol=1e-6)


@pytest.mark.parametrize("stride,k", [(5, 10), (4, 8), (3, 6)])
def test_converted_deconv_kernel_equals_conv_transpose(stride, k):
    """reference src/hifigan.cpp:22-71: ConvTranspose1d(stride s, kernel k, padding s//2 + s%2, output_padding s%2)
    evaluated as zero-stuffing + plain cross-correlation with the converter's flipped / permuted kernel"""
    import torch
    from zerovox_cpp_amd import convert
    rng = np.random.default_rng(stride)
    cin, cout, L = 6, 4, 17
    w_t = rng.standard_normal((cin, cout, k)).astype(np.float32)          # torch layout
    x = rng.standard_normal((cin, L)).astype(np.float32)
    p, op = stride // 2 + stride % 2, stride % 2
    ref = torch.nn.functional.conv_transpose1d(torch.from_numpy(x)[None], torch.from_numpy(w_t), stride=stride,
                                               padding=p, output_padding=op)[0].numpy()
    assert ref.shape == (cout, L * stride)
    w = convert.deconv_to_conv_weight(w_t)                                # (out, in, k)
    off = k - 1 - p
    stuffed = np.zeros((cin, (L - 1) * stride + 1 + 2 * off + op), np.float32)
    stuffed[:, off: off + (L - 1) * stride + 1: stride] = x
    got = np.zeros((cout, L * stride), np.float32)
    for t in range(L * stride):
        got[:, t] = np.einsum("oik,ik->o", w, stuffed[:, t: t + k])
    assert np.allclose(got, ref, atol=1e-5)


@pytest.mark.gpu
def test_converted_checkpoint_loads_and_runs(tmp_path):
    """the converter's output goes through zv_model_load (every name / shape / dtype check of the loader) and all
    three stages run on it"""
    from zerovox_cpp_amd import capi, convert, synth
    g = synth.TINY
    sd, gen, stats, _ = _training_checkpoint(g, 78)
    out = str(tmp_path / "conv.gguf")
    convert.convert(sd, _cfg(g), stats, out, meldec_generator=gen)
    m = capi.Model(out, 0)
    ids, puncts, style = synth.encoder_inputs(g, 3, 12)
    wav, nf = m.synthesize(ids, puncts, style, 32)
    assert wav.shape == (32 * g.hop_size,) and np.isfinite(wav).all() and 0 < nf <= 32
    m.close()
