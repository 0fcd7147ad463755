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


def _torch_generator(g, seed):
    """A HiFi-GAN generator built from torch modules the way the training code builds it (weight-normed Conv1d /
    ConvTranspose1d, ParallelWaveGAN `hifigan.v1` topology that the reference's graph restates, src/hifigan.cpp:187-356):
    the un-converted side of the end-to-end check."""
    import torch
    from torch import nn
    from torch.nn.utils import weight_norm
    torch.manual_seed(seed)

    class ResidualBlock(nn.Module):
        def __init__(self, k, ch, dils):
            super().__init__()
            self.convs1 = nn.ModuleList([nn.Sequential(nn.LeakyReLU(0.1), nn.Conv1d(ch, ch, k, dilation=d, padding=(k - 1) // 2 * d)) for d in dils])
            self.convs2 = nn.ModuleList([nn.Sequential(nn.LeakyReLU(0.1), nn.Conv1d(ch, ch, k, dilation=1, padding=(k - 1) // 2)) for d in dils])

    class Generator(nn.Module):
        def __init__(self):
            super().__init__()
            C = g.voc_channels
            self.input_conv = nn.Conv1d(g.num_mels, C, g.voc_kernel_size, padding=(g.voc_kernel_size - 1) // 2)
            self.upsamples, self.blocks = nn.ModuleList(), nn.ModuleList()
            for i, (s, k) in enumerate(zip(g.upsample_scales, g.upsample_kernels)):
                self.upsamples.append(nn.Sequential(nn.LeakyReLU(0.1), nn.ConvTranspose1d(C >> i, C >> (i + 1), k, s, padding=s // 2 + s % 2, output_padding=s % 2)))
                for rk in g.resblock_kernels:
                    self.blocks.append(ResidualBlock(rk, C >> (i + 1), g.resblock_dilations))
            self.output_conv = nn.Sequential(nn.LeakyReLU(0.01), nn.Conv1d(C >> len(g.upsample_scales), 1, g.voc_kernel_size, padding=(g.voc_kernel_size - 1) // 2), nn.Tanh())
            for m in self.modules():
                if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d)):
                    weight_norm(m)
            with torch.no_grad():               # speech-like output level: the 1e-4 RMS gate is absolute
                for b in self.blocks:
                    for c in b.convs2:
                        c[1].weight_g.mul_(0.3)

    return Generator().double()


def _generator_forward_ggml_semantics(gen, g, mel, mean, scale):
    """the torch modules' forward, un-converted weights, with operands rounded where the reference's ggml graph rounds them:
    every conv rounds its input activations (im2col) and its weight to f16, products accumulate in f32 or better
    (here float64); (mel - mean) / scale in front (src/hifigan.cpp:242-243)"""
    import torch
    import torch.nn.functional as F

    def h(t):                      # round to f16, keep computing in f64
        return t.float().half().double()

    def w_of(conv):                # the folded weight: g * v / ||v||, rounded to f16 like the stored tensor
        v, gm = conv.weight_v, conv.weight_g
        w = v * (gm / v.flatten(1).norm(dim=1).view(-1, 1, 1))
        return h(w.float())

    def conv(c, x):
        return F.conv1d(h(x.float()), w_of(c), c.bias, dilation=c.dilation, padding=c.padding)

    def deconv(c, x):
        return F.conv_transpose1d(h(x.float()), w_of(c), c.bias, stride=c.stride, padding=c.padding, output_padding=c.output_padding)

    lrelu = lambda t, s: F.leaky_relu(t.float(), s).double()          # noqa: E731  (activations are f32 tensors in the reference)
    with torch.no_grad():
        x = ((torch.from_numpy(mel).float() - torch.from_numpy(mean).float()) / torch.from_numpy(scale).float()).T[None].double()
        x = conv(gen.input_conv, x).float().double()
        nb = len(g.resblock_kernels)
        for i in range(len(g.upsample_scales)):
            x = deconv(gen.upsamples[i][1], lrelu(x, 0.1)).float().double()
            cs = None
            for j in range(nb):
                blk, y = gen.blocks[i * nb + j], x
                for c1, c2 in zip(blk.convs1, blk.convs2):
                    xt = conv(c1[1], lrelu(y, 0.1)).float().double()
                    xt = conv(c2[1], lrelu(xt, 0.1)).float().double()
                    y = (xt.float() + y.float()).double()
                cs = y if cs is None else (cs.float() + y.float()).double()
            x = (cs.float() * torch.tensor(1.0 / nb).float()).double()
        x = torch.tanh(conv(gen.output_conv[1], lrelu(x, 0.01)).float())
    return x[0, 0].numpy().astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", ["tiny", "small"])
def test_torch_generator_forward_equals_gpu_on_converted_checkpoint(tmp_path, geom):
    """f-1 end to end: torch forward of the UN-converted modules (weight_norm parametrisation, ConvTranspose1d) == the GPU
    vocoder on the file convert.py writes from their state dict (weight-norm fold, deconv flip + permute, name
    shortening, F16 casts; reference utils/zv2gguf.py:96-107,164-180, src/hifigan.cpp:22-71)"""
    from zerovox_cpp_amd import capi, convert, synth
    g = synth.GEOMETRIES[geom]
    sd, _, stats, _ = _training_checkpoint(g, 78)
    gen = _torch_generator(g, 5)
    gen_sd = {k: v.detach().float() for k, v in gen.state_dict().items()}
    assert any(k.endswith("weight_v") for k in gen_sd) and "upsamples.0.1.weight_g" in gen_sd
    out = str(tmp_path / "conv.gguf")
    convert.convert(sd, _cfg(g), stats, out, meldec_generator=gen_sd)
    m = capi.Model(out, 0)
    T = 40
    mean, scale = stats["mean"].astype(np.float32), stats["scale"].astype(np.float32)
    mel = (mean[None, :] + scale[None, :] * np.random.default_rng(3).standard_normal((T, g.num_mels))).astype(np.float32)
    wav = m.vocode(mel)
    m.close()
    ref = _generator_forward_ggml_semantics(gen, g, mel, mean, scale)
    assert ref.shape == wav.shape == (T * g.hop_size,)
    err = float(np.sqrt(np.mean((wav.astype(np.float64) - ref) ** 2)))
    sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    print(f"{geom}: torch generator vs GPU on the converted file: wav rms err {err:.3e} (signal rms {sig:.3f})")
    assert 0.02 < sig < 0.6 and err <= 1e-4
