"""CPU: the oracle (oracle/zv_oracle.c) against the golden vectors produced by the compiled reference.

The fixtures under tests/golden/ are OUTPUTS OF THE REFERENCE ITSELF (tests/golden/make_golden.py drives
oracle/_ref/zvref = the unmodified stage classes on ggml-CPU).  The oracle must reproduce them bit for bit
(ZVO_ORDER_GGML_AVX2).  Where oracle/_ref exists (it travels to the GPU box as a prebuilt binary) the same
is checked live on fresh seeds."""
import hashlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _inputs(g, tensors, z):
    from zerovox_cpp_amd import synth
    T, N = int(z["T"]), int(z["N"])
    mel = synth.vocoder_mel(g, tensors, int(z["seed_mel"]), T)
    hid = synth.decoder_hidden(g, int(z["seed_hidden"]), T)
    ids, puncts, style = synth.encoder_inputs(g, int(z["seed_enc"]), N)
    return T, N, mel, hid, ids, puncts, style


@pytest.mark.parametrize("fixture", ["tiny_T40_N10.npz", "small_T64_N16.npz"])
def test_oracle_reproduces_reference_full(ckpt, fixture):
    from oracle import zvoracle
    z = np.load(os.path.join(GOLD, fixture))
    path, g, tensors = ckpt(str(z["geometry"]), int(z["seed_w"]))
    T, N, mel, hid, ids, puncts, style = _inputs(g, tensors, z)
    orc = zvoracle.Oracle(tensors)
    assert np.array_equal(orc.vocoder(mel), z["wav"])
    assert np.array_equal(orc.decoder(hid, style), z["mel"])
    e = orc.encoder(g, ids, puncts, style, T)
    assert np.array_equal(e["hidden"], z["hidden"])
    assert np.array_equal(e["features"], z["features"])
    assert np.array_equal(e["logdur"], z["logdur"])
    assert np.array_equal(e["energy"], z["energy"])
    assert np.array_equal(e["pitch_bucket"], z["pitch_bucket"])
    assert np.array_equal(e["energy_bucket"], z["energy_bucket"])
    assert e["n_frames"] == int(z["n_frames"])


@pytest.mark.parametrize("fixture", ["medium_T512_N64.npz", "medium_T512_N128.npz", "medium_T1024_N256.npz"])
def test_oracle_reproduces_reference_full_size(ckpt, fixture):
    """BASELINE.json configs[0..3] at full size: SHA-256 of the whole buffers + strided samples"""
    from oracle import zvoracle
    z = np.load(os.path.join(GOLD, fixture))
    path, g, tensors = ckpt("medium", int(z["seed_w"]))
    T, N, mel, hid, ids, puncts, style = _inputs(g, tensors, z)
    s = int(z["stride"])
    orc = zvoracle.Oracle(tensors)
    e = orc.encoder(g, ids, puncts, style, T)
    assert e["n_frames"] == int(z["n_frames"])
    assert np.array_equal(e["pitch_bucket"], z["pitch_bucket"]) and np.array_equal(e["energy_bucket"], z["energy_bucket"])
    assert np.array_equal(e["logdur"], z["logdur"])
    assert np.array_equal(e["hidden"].reshape(-1)[::s], z["hidden_samples"])
    assert sha(e["hidden"]) == str(z["hidden_sha256"])
    if N != 128:       # decoder / vocoder inputs depend on T only: once per T (512: the N = 64 fixture; 1 024)
        d = orc.decoder(hid, style)
        assert np.array_equal(d.reshape(-1)[::s], z["mel_samples"]) and sha(d) == str(z["mel_sha256"])
        w = orc.vocoder(mel)
        assert np.array_equal(w[::s], z["wav_samples"]) and sha(w) == str(z["wav_sha256"])


def test_oracle_vs_live_reference_fresh_seed(tmp_path):
    """different weights seed + ragged sizes, compared live against oracle/_ref when it is present"""
    from zerovox_cpp_amd import gguf, synth
    from oracle import zvoracle
    if not zvoracle.have_reference():
        pytest.skip("oracle/_ref/zvref not built on this machine")
    g = synth.TINY
    path = str(tmp_path / "t.gguf")
    synth.write_checkpoint(path, g, 99, trim_dims=True)      # ggml-C-writer style n_dims
    _, tensors = gguf.read_gguf(path)
    orc = zvoracle.Oracle(tensors)
    for T, N in ((7, 3), (33, 17)):
        mel = synth.vocoder_mel(g, tensors, 21, T)
        hid = synth.decoder_hidden(g, 22, T)
        ids, puncts, style = synth.encoder_inputs(g, 23, N)
        assert np.array_equal(orc.vocoder(mel), zvoracle.run_reference(path, T=T, voc=mel)["wav"])
        assert np.array_equal(orc.decoder(hid, style), zvoracle.run_reference(path, T=T, dec=(hid, style))["mel"])
        r = zvoracle.run_reference(path, T=T, N=N, enc=(ids, puncts, style), E=g.E)
        e = orc.encoder(g, ids, puncts, style, T)
        for k in ("hidden", "features", "logdur", "energy", "pitch_bucket", "energy_bucket"):
            assert np.array_equal(e[k], r[k]), k
        assert e["n_frames"] == r["n_frames"]


def test_instancenorm_known_answer():
    """the reference repo's only own golden data: utils/norm1dexample.json (PyTorch InstanceNorm1d(affine) pair,
    weights rounded to 4 decimals in the file -> 2e-4 tolerance)"""
    from oracle import zvoracle
    z = np.load(os.path.join(GOLD, "instnorm1d_kat.npz"))
    orc = zvoracle.Oracle({})
    y = orc.norm_rows(z["x_in"]) * z["weight"][:, None] + z["bias"][:, None]
    assert np.max(np.abs(y - z["x_out"])) < 2e-4


def test_length_regulator_edge_cases():
    from oracle import zvoracle
    orc = zvoracle.Oracle({})
    feat = np.arange(12, dtype=np.float32).reshape(3, 4)
    # durations: exp(ld)-1 -> 0.4 (-> 0), 2.5 (-> 3: truncating x+0.5, not half-even), 100 (clipped at T)
    ld = np.log(np.array([1.4, 3.5, 101.0], np.float32)).astype(np.float32)
    hid, nf = orc.length_regulator(feat, ld, 6)
    assert nf == 6 and np.array_equal(hid[:3], np.repeat(feat[1:2], 3, 0)) and np.array_equal(hid[3:], np.repeat(feat[2:3], 3, 0))
    hid, nf = orc.length_regulator(feat, np.full(3, -5, np.float32), 4)
    assert nf == 0 and not hid.any()


def test_oracle_self_noise_floor(ckpt):
    """re-association noise of the reference semantics (SURVEY.md Appx D): bounded, and the reason GPU gates are
    noise-aware.  wav floor stays well under the 1e-4 RMS gate at speech-like amplitude."""
    from zerovox_cpp_amd import synth
    from oracle import zvoracle
    path, g, tensors = ckpt("small")
    mel = synth.vocoder_mel(g, tensors, 7, 32)
    orc = zvoracle.Oracle(tensors)
    a = orc.vocoder(mel)
    orc.set_order(zvoracle.ORDER_SEQ_F32)
    b = orc.vocoder(mel)
    floor = float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)))
    assert 0 < floor < 6e-5


@pytest.mark.parametrize("k,dil,ic,oc,L", [(3, 1, 24, 40, 50), (7, 3, 32, 32, 97), (11, 5, 16, 48, 130), (1, 1, 33, 17, 20)])
def test_oracle_conv_against_independent_torch_fp32(k, dil, ic, oc, L):
    """an implementation that shares no code with the oracle: torch's fp32 conv1d on operands rounded to f16 exactly where
    ggml rounds them (the im2col of the activations, the stored weights).  Products of two f16 values are exact in f32, so
    the two differ only by f32 summation order: a few ulp of the accumulated magnitude"""
    import torch
    from oracle import zvoracle
    rng = np.random.default_rng(k * 100 + dil)
    x = rng.standard_normal((ic, L)).astype(np.float32)               # channels-first, like the reference's conv input
    w = (rng.standard_normal((oc, ic, k)) / np.sqrt(ic * k)).astype(np.float16)
    b = rng.standard_normal(oc).astype(np.float32)
    pad = (k - 1) // 2 * dil
    got = zvoracle.Oracle({}).conv1d(x, w, b, pad, dil)
    x16 = torch.from_numpy(x.astype(np.float16).astype(np.float32))[None]
    ref = torch.nn.functional.conv1d(x16, torch.from_numpy(w.astype(np.float32)), torch.from_numpy(b), padding=pad, dilation=dil)[0].numpy()
    assert got.shape == ref.shape == (oc, L)
    assert np.max(np.abs(got - ref)) <= 2e-5 * max(1.0, float(np.max(np.abs(ref))))


def test_oracle_reproduces_reference_demo_utterance(ckpt):
    """ZeroVOXModel::eval() of the reference (its hard-coded 120-phoneme utterance + 528-float style vector, the three
    stages back to back at T = max_seq_len, src/zerovox.cpp:198-335) on the synthetic medium checkpoint: the oracle
    chain reproduces every stage of the compiled reference bit for bit.  The utterance data is served by the product
    library (zv_demo_utterance), which needs no GPU."""
    from zerovox_cpp_amd import capi
    from oracle import zvoracle
    z = np.load(os.path.join(GOLD, "demo_medium_T1500.npz"))
    path, g, tensors = ckpt("medium", int(z["seed_w"]))
    ids, puncts, style = capi.demo_utterance()
    assert len(ids) == 120 == int(z["N"]) and style.shape == (528,) and g.E == 528
    assert ids[0] == 69 and ids[-1] == 87 and puncts[-1] == 3            # first / last entries of the reference's literals
    T, s = int(z["T"]), int(z["stride"])
    assert T == g.max_seq_len
    orc = zvoracle.Oracle(tensors)
    e = orc.encoder(g, ids, puncts, style, T)
    assert e["n_frames"] == int(z["n_frames"])
    for k in ("logdur", "energy", "pitch_bucket", "energy_bucket"):
        assert np.array_equal(e[k], z[k]), k
    assert sha(e["features"]) == str(z["features_sha256"]) and sha(e["hidden"]) == str(z["hidden_sha256"])
    mel = orc.decoder(e["hidden"], style)
    assert np.array_equal(mel.reshape(-1)[::s], z["mel_samples"]) and sha(mel) == str(z["mel_sha256"])
    wav = orc.vocoder(mel)
    assert np.array_equal(wav[::s], z["wav_samples"]) and sha(wav) == str(z["wav_sha256"])


def test_oracle_reproduces_reference_num_phonemes_below_max(ckpt):
    """FS2Encoder::eval(num_phonemes < max_n_phonemes): all tokens are encoded, the regulator walks the first num
    (reference src/fs2encoder.cpp:594-650)"""
    from zerovox_cpp_amd import synth
    from oracle import zvoracle
    z = np.load(os.path.join(GOLD, "small_T64_N16_num9.npz"))
    path, g, tensors = ckpt("small", int(z["seed_w"]))
    ids, puncts, style = synth.encoder_inputs(g, int(z["seed_enc"]), int(z["N"]))
    e = zvoracle.Oracle(tensors).encoder(g, ids, puncts, style, int(z["T"]), num_phonemes=int(z["num"]))
    assert e["n_frames"] == int(z["n_frames"]) and np.array_equal(e["hidden"], z["hidden"])
    assert np.array_equal(e["logdur"], z["logdur"]) and sha(e["features"]) == str(z["features_sha256"])


def test_oracle_layers_compose_to_the_oracle_vocoder(tmp_path):
    """zvo_layer's vocoder kinds (input conv, transposed convs, residual blocks, output conv: the per-layer checkers of
    tests/test_gpu_layers.py) chained by hand reproduce zvo_vocoder — which reproduces the compiled reference — bit for bit"""
    from zerovox_cpp_amd import gguf, synth
    from oracle import zvoracle
    g = synth.TINY
    path = str(tmp_path / "tiny.gguf")
    synth.write_checkpoint(path, g, 1234)
    _, t = gguf.read_gguf(path)
    o = zvoracle.Oracle(t)
    mel = synth.vocoder_mel(g, t, 7, 16)
    x = o.layer(o.LAYER_VOC_INPUT, 0, mel, g.voc_channels)
    for i, s in enumerate(g.upsample_scales):
        up = o.layer(o.LAYER_VOC_UPSAMPLE, i, x, x.shape[1] // 2, out_rows=x.shape[0] * s)
        ys = [o.layer(o.LAYER_VOC_RESBLOCK, i * 3 + j, up, up.shape[1]) for j in range(3)]
        x = ((ys[0] + ys[1]) + ys[2]) * np.float32(1.0 / np.float32(3))          # src/hifigan.cpp:300-315
    assert np.array_equal(o.layer(o.LAYER_VOC_OUTPUT, 0, x, 0), o.vocoder(mel))
