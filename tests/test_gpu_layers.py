"""GPU: one layer at a time.  zv_debug_layer runs ONE layer of the production schedule (the fused kernels, the same
launch configurations) on a given input; the oracle's zvo_layer runs the reference semantics of the same layer on the same
input (the counterpart of the reference's tensor_dbg, src/utils.cpp:19-44).  Nothing compounds from layer to layer, so
these gates are far tighter than the stage-level ones: the only differences are f32 summation order and, inside a layer
with several convs, the occasional f16 re-rounding flip that the order causes — measured against the oracle's own
re-association noise on the same layer (AVX2 lane order vs sequential f32)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_M = {}


@pytest.fixture(scope="module")
def env(ckpt):
    from zerovox_cpp_amd import capi
    from oracle import zvoracle
    if "m" not in _M:
        path, g, tensors = ckpt("medium")
        _M.update(m=capi.Model(path, 0), g=g, t=tensors, o=zvoracle.Oracle(tensors))
    yield _M["m"], _M["g"], _M["t"], _M["o"]


def teardown_module(module):
    for k in ("m", "mb"):
        if k in _M:
            _M[k].close()
    _M.clear()


def _rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


def _check(name, got, ref, alt, rel_gate, floor_mult=2.0):
    """got: GPU, ref: oracle (ggml AVX2 order), alt: oracle (sequential f32).  rel = rms(err) / rms(signal)."""
    sig = _rms(ref)
    err, floor = _rms(got - ref) / sig, _rms(alt - ref) / sig
    mx = float(np.max(np.abs(got - ref))) / sig
    print(f"{name:28s} rel rms err {err:.2e} (oracle self-noise {floor:.2e}), max {mx:.2e}, signal rms {sig:.3f}")
    assert np.isfinite(got).all()
    assert err <= max(floor_mult * floor, 3e-7), name          # no worse than the reference's own re-association noise
    assert err <= rel_gate, name


def _oracle_pair(o, *args, **kw):
    from oracle import zvoracle
    o.set_order(zvoracle.ORDER_GGML_AVX2)
    ref = o.layer(*args, **kw)
    o.set_order(zvoracle.ORDER_SEQ_F32)
    alt = o.layer(*args, **kw)
    o.set_order(zvoracle.ORDER_GGML_AVX2)
    return ref, alt


@pytest.mark.parametrize("block", list(range(12)))
def test_every_hifigan_residual_block(env, block):
    """all 12 residual blocks (4 stages x 3 kernel sizes), each = 3 dilation pairs of 2 convs through the fused pair /
    whole-block kernels, 32 frames at the stage's rate"""
    m, g, t, o = env
    stage = block // 3
    C, rate = m.voc_channels(stage), m.voc_rate(stage)
    x = (0.5 * np.random.default_rng(100 + block).standard_normal((32 * rate, C))).astype(np.float32)
    got = m.debug_layer(m.LAYER_VOC_RESBLOCK, block, x, C)
    ref, alt = _oracle_pair(o, o.LAYER_VOC_RESBLOCK, block, x, C)
    _check(f"hifigan block {block} (C={C})", got, ref, alt, 2e-4)


@pytest.mark.parametrize("layer", [0, 1, 2, 3])
def test_every_encoder_fft_block(env, layer):
    """attention sublayer (f32 QKV / attention / fc / add + LayerNorm) + conv FFN (k9 / k1 with the f16 ReLU operand)"""
    m, g, t, o = env
    x = np.random.default_rng(200 + layer).standard_normal((96, g.E)).astype(np.float32)
    got = m.debug_layer(m.LAYER_ENC_FFT, layer, x, g.E)
    ref, alt = _oracle_pair(o, o.LAYER_ENC_FFT, layer, x, g.E, heads=g.encoder_head, ksz=g.conv_kernel_size)
    _check(f"encoder FFT block {layer}", got, ref, alt, 1e-4)


@pytest.mark.parametrize("block", list(range(7)))
def test_every_decoder_residual_block(env, block):
    """ResBlk1d (affine InstanceNorm) x 2 and AdainResBlk1d x 5, each: norm -> lrelu -> conv -> norm -> lrelu -> conv
    (+ learned 1x1 shortcut) / sqrt 2, statistics from the conv epilogues' partial sums"""
    m, g, t, o = env
    E, R = g.E, g.residual_dim
    cin = [E, 2 * E, 2 * E + R, 2 * E + R, 2 * E + R, E, E][block]
    cout = [2 * E, 2 * E, 2 * E, 2 * E, E, E, E][block]
    T = 96
    # distinct frames (InstanceNorm over near-identical frames only measures rounding of the statistics)
    x = (1.2 * np.random.default_rng(300 + block).standard_normal((T, cin))).astype(np.float32)
    style = (0.05 * np.random.default_rng(7).standard_normal(E)).astype(np.float32)
    from zerovox_cpp_amd import capi
    for pre in (0, 1):                           # both ways of feeding the convs their normalised operand
        with capi.switches(ZV_DEC_PREPASS=pre):
            got = m.debug_layer(m.LAYER_DEC_BLOCK, block, x, cout, style=style)
        ref, alt = _oracle_pair(o, o.LAYER_DEC_BLOCK, block, x, cout, style=style)
        _check(f"decoder block {block} ({cin}->{cout}) prepass={pre}", got, ref, alt, 3e-4)
        if pre:
            with capi.switches(ZV_DEC_PREPASS=1, ZV_CONV_GEMM=2):      # the batch form of the wide convs on this 96-frame block
                assert np.array_equal(m.debug_layer(m.LAYER_DEC_BLOCK, block, x, cout, style=style), got), "conv_gemm_kernel"


@pytest.mark.parametrize("p", [0, 1, 2])
def test_every_variance_predictor(env, p):
    m, g, t, o = env
    x = np.random.default_rng(400 + p).standard_normal((80, g.E)).astype(np.float32)
    got = m.debug_layer(m.LAYER_VAR_PRED, p, x, 0)
    ref, alt = _oracle_pair(o, o.LAYER_VAR_PRED, p, x, 0, ksz=(g.vp_kernel_size,))
    _check(f"variance predictor {p}", got, ref, alt, 1e-4)


def test_unknown_layers_are_errors(env):
    from zerovox_cpp_amd import capi
    m, g, t, o = env
    x = np.zeros((32, g.E), np.float32)
    for kind, idx in ((m.LAYER_ENC_FFT, 99), (m.LAYER_DEC_BLOCK, 7), (m.LAYER_VOC_UPSAMPLE, 4), (99, 0)):
        with pytest.raises(capi.ZvError):
            m.debug_layer(kind, idx, x, g.E)


# ---- round 3: the layers that only stage-level gates (at the noise floor) used to cover ------------------------------

@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_every_transposed_conv(env, idx):
    """leaky_relu(0.1) + conv_transpose1d (reference src/hifigan.cpp:22-71, 281-297; here a polyphase ordinary conv whose
    channels-last output IS the up-sampled sequence), 48 frames at the stage's input rate"""
    m, g, t, o = env
    cin = g.voc_channels >> idx
    cout, s = cin // 2, g.upsample_scales[idx]
    rate_in = 1 if idx == 0 else m.voc_rate(idx - 1)
    x = (0.7 * np.random.default_rng(400 + idx).standard_normal((48 * rate_in, cin))).astype(np.float32)
    got = m.debug_layer(m.LAYER_VOC_UPSAMPLE, idx, x, cout, out_rows=x.shape[0] * s)
    ref, alt = _oracle_pair(o, o.LAYER_VOC_UPSAMPLE, idx, x, cout, out_rows=x.shape[0] * s)
    _check(f"conv_transpose1d {idx} ({cin}->{cout}, x{s})", got, ref, alt, 1e-4)


def test_vocoder_input_conv(env):
    """(mel - mean) / scale + input conv k7 (src/hifigan.cpp:242-265)"""
    from zerovox_cpp_amd import synth
    m, g, t, o = env
    x = synth.vocoder_mel(g, t, 41, 96)
    got = m.debug_layer(m.LAYER_VOC_INPUT, 0, x, g.voc_channels)
    ref, alt = _oracle_pair(o, o.LAYER_VOC_INPUT, 0, x, g.voc_channels)
    _check("vocoder input conv", got, ref, alt, 1e-4)


def test_vocoder_output_conv_tanh(env):
    """leaky_relu(0.01) + conv k7 (32 -> 1) + tanh (src/hifigan.cpp:324-345) on a given MRF mean"""
    m, g, t, o = env
    C = g.voc_channels >> len(g.upsample_scales)
    x = (0.6 * np.random.default_rng(43).standard_normal((16 * g.hop_size, C))).astype(np.float32)
    got = m.debug_layer(m.LAYER_VOC_OUTPUT, 0, x, 0)
    ref, alt = _oracle_pair(o, o.LAYER_VOC_OUTPUT, 0, x, 0)
    _check("vocoder output conv + tanh", got, ref, alt, 1e-4)


def test_decoder_asr_res_and_to_out(env):
    """asr_res = InstanceNorm(conv1x1(enc_seq)) (src/stylettsdec.cpp:382-396) and to_out = conv1x1 + bias (:432-441)"""
    m, g, t, o = env
    T = 96
    x = (1.1 * np.random.default_rng(44).standard_normal((T, g.E))).astype(np.float32)
    got = m.debug_layer(m.LAYER_DEC_ASR_RES, 0, x, g.residual_dim)
    ref, alt = _oracle_pair(o, o.LAYER_DEC_ASR_RES, 0, x, g.residual_dim)
    _check("decoder asr_res", got, ref, alt, 1e-4)
    got = m.debug_layer(m.LAYER_DEC_TO_OUT, 0, x, g.num_mels)
    ref, alt = _oracle_pair(o, o.LAYER_DEC_TO_OUT, 0, x, g.num_mels)
    _check("decoder to_out", got, ref, alt, 1e-4)


def test_encoder_embedding_is_bit_exact(env):
    """word + punctuation embedding + positional encoding (src/fs2encoder.cpp:306-324): a gather and one f32 add per element"""
    from zerovox_cpp_amd import synth
    m, g, t, o = env
    ids, puncts, _ = synth.encoder_inputs(g, 45, 200)
    x = np.stack([ids, puncts], axis=1).astype(np.float32)
    got = m.debug_layer(m.LAYER_ENC_EMBED, 0, x, g.E)
    ref = o.layer(o.LAYER_ENC_EMBED, 0, x, g.E)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("layer", [0, 1, 2, 3])
def test_attention_sublayer_and_feed_forward_sublayer_alone(env, layer):
    """sub-block taps (round 4; the reference's tensor_dbg taps any node, src/utils.cpp:19-44): MultiHeadAttention alone (QKV,
    softmax(QK^T / sqrt d) V, fc, residual + LayerNorm: src/fs2encoder.cpp:71-140) and PositionwiseFeedForward alone (conv k9 ->
    ReLU -> conv k1, residual + LayerNorm: :174-228) — a regression in one of them no longer hides behind the other; both forms
    of the attention kernel"""
    from zerovox_cpp_amd import capi
    m, g, t, o = env
    x = np.random.default_rng(500 + layer).standard_normal((96, g.E)).astype(np.float32)
    for sw in ({"ZV_ATT_SCALAR": 1}, {"ZV_ATT_MFMA": 1}):
        with capi.switches(**sw):
            got = m.debug_layer(m.LAYER_ENC_MHA, layer, x, g.E)
        ref, alt = _oracle_pair(o, o.LAYER_ENC_MHA, layer, x, g.E, heads=g.encoder_head, ksz=g.conv_kernel_size)
        _check(f"attention sublayer {layer} {list(sw)[0]}", got, ref, alt, 1e-4)
    got = m.debug_layer(m.LAYER_ENC_FFN, layer, x, g.E)
    ref, alt = _oracle_pair(o, o.LAYER_ENC_FFN, layer, x, g.E, heads=g.encoder_head, ksz=g.conv_kernel_size)
    _check(f"feed-forward sublayer {layer}", got, ref, alt, 1e-4)
    # the two taps chained are the whole block
    y = m.debug_layer(m.LAYER_ENC_MHA, layer, x, g.E)
    assert np.array_equal(m.debug_layer(m.LAYER_ENC_FFN, layer, y, g.E), m.debug_layer(m.LAYER_ENC_FFT, layer, x, g.E))


@pytest.mark.parametrize("idx", list(range(10)))
def test_every_adain_alone(env, idx):
    """AdaIN1d alone (src/stylettsdec.cpp:171-200): gamma / beta from the production fc GEMM of all ten layers, the production
    statistics, ((x - mean) * rstd) * (1 + gamma) + beta — without the activation and the conv it is fused into in the schedule"""
    m, g, t, o = env
    E, R = g.E, g.residual_dim
    cin = [2 * E + R, 2 * E + R, 2 * E + R, E, E]
    cout = [2 * E, 2 * E, E, E, E]
    C = (cout if idx & 1 else cin)[idx // 2]
    T = 96
    x = (1.2 * np.random.default_rng(600 + idx).standard_normal((T, C)) + 0.3).astype(np.float32)
    style = (0.05 * np.random.default_rng(8).standard_normal(E)).astype(np.float32)
    got = m.debug_layer(m.LAYER_DEC_ADAIN, idx, x, C, style=style)
    ref, alt = _oracle_pair(o, o.LAYER_DEC_ADAIN, idx, x, C, style=style)
    _check(f"AdaIN decode.{idx // 2}.norm{1 + (idx & 1)} (C={C})", got, ref, alt, 1e-5, floor_mult=4.0)


# ---- round 4: the same per-layer gates with the kernels only BATCHES pick forced on (a regression in a batch kernel then
# ---- shows at its layer, against the reference semantics, instead of in a whole-vocoder RMS) --------------------------------
BATCH_REGIME = dict(ZV_BLOCK64=-11, ZV_CONV_STREAM=2, ZV_CONV_GEMM=2, ZV_UP_GEMM=2, ZV_PAIR64_RING=2, ZV_TRIPLE_V2=3, ZV_FUSE256=1,
                    ZV_PAIR_MT=0, ZV_DEC_PREPASS=1)


def _batch_model(ckpt):
    """a second model built with the regime on (ZV_FUSE256 is sampled when a model is loaded)"""
    from zerovox_cpp_amd import capi
    if "mb" not in _M:
        with capi.switches(**BATCH_REGIME):
            _M["mb"] = capi.Model(ckpt("medium")[0], 0)
    return _M["mb"]


@pytest.mark.parametrize("block", list(range(12)))
def test_every_hifigan_residual_block_on_the_batch_kernels(env, ckpt, block):
    """resblock_pair_kernel<256> (fused, 96-row tiles) / <128> / resblock_pair64_kernel (LDS weight ring) + resblock_block64_kernel
    (two pairs per launch, every tap count) / resblock_block32_kernel on 512-row tiles, one residual block at a time vs the oracle;
    and the bits of the default kernels"""
    from zerovox_cpp_amd import capi
    m, g, t, o = env
    stage = block // 3
    C, rate = m.voc_channels(stage), m.voc_rate(stage)
    x = (0.5 * np.random.default_rng(100 + block).standard_normal((32 * rate, C))).astype(np.float32)
    dflt = m.debug_layer(m.LAYER_VOC_RESBLOCK, block, x, C)
    mb = _batch_model(ckpt)
    with capi.switches(**BATCH_REGIME):
        got = mb.debug_layer(mb.LAYER_VOC_RESBLOCK, block, x, C)
    ref, alt = _oracle_pair(o, o.LAYER_VOC_RESBLOCK, block, x, C)
    _check(f"hifigan block {block} (C={C}) batch kernels", got, ref, alt, 2e-4)
    assert np.array_equal(got, dflt)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_every_transposed_conv_on_the_batch_kernels(env, idx):
    """conv_gemm_kernel behind act_f16_kernel (the two deep upsample convs) / conv_stream_kernel (the two memory-bound ones)"""
    from zerovox_cpp_amd import capi
    m, g, t, o = env
    cin = g.voc_channels >> idx
    cout, s = cin // 2, g.upsample_scales[idx]
    rate_in = 1 if idx == 0 else m.voc_rate(idx - 1)
    x = (0.7 * np.random.default_rng(400 + idx).standard_normal((48 * rate_in, cin))).astype(np.float32)
    dflt = m.debug_layer(m.LAYER_VOC_UPSAMPLE, idx, x, cout, out_rows=x.shape[0] * s)
    with capi.switches(**BATCH_REGIME):
        got = m.debug_layer(m.LAYER_VOC_UPSAMPLE, idx, x, cout, out_rows=x.shape[0] * s)
    ref, alt = _oracle_pair(o, o.LAYER_VOC_UPSAMPLE, idx, x, cout, out_rows=x.shape[0] * s)
    _check(f"conv_transpose1d {idx} batch kernels", got, ref, alt, 1e-4)
    assert np.array_equal(got, dflt)
