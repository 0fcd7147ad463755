"""-m gpu: BASELINE.json's full-size configurations.  The CPU oracle is too slow to be the live checker here on
every run, so these cases use (1) the committed golden samples produced by the compiled reference
(tests/golden/medium_T512_*.npz: strided samples of the reference's outputs), teacher-forced per stage, and
(2) size-independent properties of the path: determinism, prefix consistency (frames far from the cut do not
change when the utterance is truncated: exercises every tile/halo boundary), batching invariance of the graph
replay, zero-tail handling of the length regulator."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


@pytest.fixture(scope="module")
def medium(ckpt):
    from zerovox_cpp_amd import capi
    path, g, tensors = ckpt("medium")
    m = capi.Model(path, 0)
    yield m, g, tensors
    m.close()


@pytest.mark.parametrize("fixture", ["medium_T512_N64.npz", "medium_T1024_N256.npz"])
def test_config2_vocoder_512_frames_vs_reference_golden(medium, fixture):
    """configs[1]: 80-ch mel, 512 frames -> 153600 samples (and configs[3]'s T = 1 024); gate: wav RMS <= 1e-4 vs the
    reference's own output"""
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    z = np.load(os.path.join(GOLD, fixture))
    T, s = int(z["T"]), int(z["stride"])
    mel = synth.vocoder_mel(g, tensors, int(z["seed_mel"]), T)
    wav = model.vocode(mel)
    assert wav.shape == (T * 300,) and np.isfinite(wav).all()
    err = _rms(wav[::s] - z["wav_samples"])
    print(f"config2: wav rms err (strided vs reference) {err:.3e}; reference wav rms {float(z['wav_rms']):.3f}; "
          f"re-association floor of this fixture {float(z['floor_wav_rms']):.3e}")
    assert err <= 1e-4 and err <= 1.5 * float(z["floor_wav_rms"])
    # determinism + graph replay gives the same bits as eager launches
    model.set_graph_mode(True)
    w2 = model.vocode(mel)
    w3 = model.vocode(mel)
    model.set_graph_mode(False)
    assert np.array_equal(wav, w2) and np.array_equal(w2, w3)


@pytest.mark.parametrize("fixture", ["medium_T512_N64.npz", "medium_T1024_N256.npz"])
def test_config2_decoder_512_frames_vs_reference_golden(medium, fixture):
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    z = np.load(os.path.join(GOLD, fixture))
    T, s = int(z["T"]), int(z["stride"])
    hid = synth.decoder_hidden(g, int(z["seed_hidden"]), T)
    _, _, style = synth.encoder_inputs(g, int(z["seed_enc"]), int(z["N"]))
    mel = model.decode(hid, style)
    d = mel.reshape(-1)[::s] - z["mel_samples"]
    print(f"decoder T={T}: mel err max {np.max(np.abs(d)):.3e} rms {_rms(d):.3e} (mel rms {float(z['mel_rms']):.3f}; "
          f"reference self-noise floor at this size: max ~3e-3..4e-3, rms ~1e-3 — SURVEY.md Appx D)")
    assert np.isfinite(mel).all()
    # gate: 1.5 x the reference semantics' own re-association noise on THIS input, measured when the fixture was made
    # (tests/golden/make_golden.py add_floors) — north_star's absolute 1e-3 max-abs gate is below that floor on these weights
    fl_max, fl_rms = float(z["floor_mel_max"]), float(z["floor_mel_rms"])
    print(f"   floor of this fixture: max {fl_max:.3e} rms {fl_rms:.3e} -> ratios {np.max(np.abs(d)) / fl_max:.2f} / {_rms(d) / fl_rms:.2f}")
    assert _rms(d) <= 1.5 * fl_rms and np.max(np.abs(d)) <= 1.5 * fl_max
    # two ways to feed the convs their normalised operand (fused prologue for single utterances, one f16 operand pass
    # for batches): same bits, whichever the size picks
    from zerovox_cpp_amd import capi
    for v in (0, 1):
        with capi.switches(ZV_DEC_PREPASS=v):
            assert np.array_equal(model.decode(hid, style), mel), v
        # the single-utterance MFMA loop switched off: same chain per output element, same bits
        with capi.switches(ZV_DEC_PREPASS=v, ZV_CONV_SINGLE=0):
            assert np.array_equal(model.decode(hid, style), mel), ("no single loop", v)
    # the batches' 256 x 256-tile GEMM form of the wide convs (both operands through LDS by LDS-DMA, taps folded into the K loop)
    with capi.switches(ZV_DEC_PREPASS=1, ZV_CONV_GEMM=2):
        assert np.array_equal(model.decode(hid, style), mel), "conv_gemm_kernel"
    with capi.switches(ZV_DEC_PREPASS=1, ZV_CONV_GEMM=0):
        assert np.array_equal(model.decode(hid, style), mel), "no conv_gemm_kernel"


@pytest.mark.parametrize("fixture", ["medium_T512_N64.npz", "medium_T512_N128.npz", "medium_T1024_N256.npz"])
def test_config1_3_encoder_vs_reference_golden(medium, fixture):
    """configs[0] (N=64), configs[2] (N=128), configs[3]'s longest utterance (N=256): float predictions and integer decisions
    against the reference, gated on the fixture's own re-association floors with near-tie accounting"""
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    z = np.load(os.path.join(GOLD, fixture))
    T, N = int(z["T"]), int(z["N"])
    ids, puncts, style = synth.encoder_inputs(g, int(z["seed_enc"]), N)
    e = model.encode(ids, puncts, style, T)
    print(fixture)
    from parity_helpers import encoder_decisions_vs_reference
    encoder_decisions_vs_reference(e, z, g.ve_n_bins - 1, T)
    # zero tail behind the regulated frames, exactly
    assert not e["hidden"][e["n_frames"]:].any()
    assert e["hidden"][: e["n_frames"]].any()


def test_prefix_consistency_full_size(medium):
    """truncating the utterance must not change samples whose receptive field does not reach the cut"""
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    mel = synth.vocoder_mel(g, tensors, 31, 512)
    full = model.vocode(mel)
    half = model.vocode(mel[:301])
    margin = 24 * 300                 # vocoder receptive field is < 24 frames per side
    n = 301 * 300 - margin
    assert np.array_equal(full[:n], half[:n])
    assert not np.array_equal(full[n: 301 * 300], half[n:])


def test_host_and_device_entry_points_agree(medium):
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    T = 64
    mel = synth.vocoder_mel(g, tensors, 33, T)
    ref = model.vocode(mel)
    d_mel, d_wav = model.device_alloc(mel.nbytes), model.device_alloc(T * 300 * 4)
    model.h2d(d_mel, mel)
    model.vocode_device(d_mel, T, d_wav)
    out = np.empty(T * 300, np.float32)
    model.d2h(out, d_wav)
    model.device_free(d_mel)
    model.device_free(d_wav)
    assert np.array_equal(out, ref)


def test_fused_256_channel_stage_vs_reference_golden(ckpt):
    """long / batched utterances run HiFi-GAN stage 1 (256 channels) on the fused dilation-pair kernel; a single
    512-frame utterance does not (too few rows).  ZV_FUSE256=1 forces that path at 512 frames so that it is checked
    against the reference's own output like the default path"""
    from zerovox_cpp_amd import capi, synth
    path, g, tensors = ckpt("medium")
    z = np.load(os.path.join(GOLD, "medium_T512_N64.npz"))
    T, s = int(z["T"]), int(z["stride"])
    mel = synth.vocoder_mel(g, tensors, int(z["seed_mel"]), T)
    with capi.switches(ZV_FUSE256=1):      # a schedule switch: sampled when the model is built
        m = capi.Model(path, 0)
    wav = m.vocode(mel)
    m.close()
    err = _rms(wav[::s] - z["wav_samples"])
    print(f"fused 256-channel stage: wav rms err (strided vs reference) {err:.3e}")
    assert np.isfinite(wav).all() and err <= 1e-4


@pytest.mark.parametrize("fixture", ["medium_T512_N64.npz", "medium_T1024_N256.npz"])
def test_batch_regime_kernels_vs_reference_golden(ckpt, fixture):
    """the kernels only batches pick by themselves — whole-block kernel with its weights through LDS (two weight buffers),
    64-channel pair kernel with the LDS weight ring, fused 256-channel stage — forced onto one utterance and checked
    against the REFERENCE's waveform for that mel, like the default path (not only against our own other regimes)"""
    from zerovox_cpp_amd import capi, synth
    path, g, tensors = ckpt("medium")
    z = np.load(os.path.join(GOLD, fixture))
    T, s = int(z["T"]), int(z["stride"])
    mel = synth.vocoder_mel(g, tensors, int(z["seed_mel"]), T)
    with capi.switches(ZV_FUSE256=1, ZV_TRIPLE_V2=3, ZV_PAIR64_RING=2, ZV_MERGE_ALWAYS=1, ZV_UP_GEMM=2, ZV_CONV_GEMM=2, ZV_CONV_STREAM=2, ZV_BLOCK64=-11):
        m = capi.Model(path, 0)
        wav = m.vocode(mel)
        m.close()
    err = _rms(wav[::s] - z["wav_samples"])
    print(f"{fixture}: batch-regime kernels, wav rms err (strided vs reference) {err:.3e}")
    assert np.isfinite(wav).all() and err <= 1e-4


def test_vocoder_beyond_max_seq_len_and_regime_change(medium):
    """T is a run-time argument: 2 000 frames (> max_seq_len, and long enough that stage 1 switches to the fused
    kernel by itself).  Samples far from the cut must be the 512-frame run's, bit for bit: the two runs use different
    kernels for stage 1, but every kernel sums in the same order"""
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    T = 2000
    mel = synth.vocoder_mel(g, tensors, 33, T)
    long = model.vocode(mel)
    assert long.shape == (T * 300,) and np.isfinite(long).all()
    short = model.vocode(mel[:512])
    n = (512 - 24) * 300
    assert np.array_equal(long[:n], short[:n])
    assert np.array_equal(long, model.vocode(mel))        # deterministic


def test_repeatability_stress(medium):
    """races show up as run-to-run differences: every stage is repeated 40 times at full size (all kernel families:
    plain / split-K / fused pair / whole-block) and must return the same bits every time"""
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    T = 512
    mel = synth.vocoder_mel(g, tensors, 41, T)
    hid = synth.decoder_hidden(g, 42, T)
    ids, puncts, style = synth.encoder_inputs(g, 43, 96)
    w0, m0 = model.vocode(mel), model.decode(hid, style)
    e0 = model.encode(ids, puncts, style, T)
    s0, nf0 = model.synthesize(ids, puncts, style, T)
    for it in range(40):
        assert np.array_equal(model.vocode(mel), w0), f"vocoder differs at repetition {it}"
        assert np.array_equal(model.decode(hid, style), m0), f"decoder differs at repetition {it}"
        if it % 4 == 0:
            e = model.encode(ids, puncts, style, T)
            assert np.array_equal(e["hidden"], e0["hidden"]) and e["n_frames"] == e0["n_frames"]
            s, nf = model.synthesize(ids, puncts, style, T)
            assert nf == nf0 and np.array_equal(s, s0)


def test_kernel_regimes_give_the_same_bits(ckpt):
    """the vocoder picks kernels by sequence length (unfused / fused 256-channel stage, pair / whole-block kernel);
    every one of them sums each output element in the same order, so the choice must not change a single bit —
    this is what makes chunked (streaming) vocoding exact"""
    from zerovox_cpp_amd import capi, synth
    path, g, tensors = ckpt("medium")
    mel = synth.vocoder_mel(g, tensors, 51, 384)
    outs = {}
    for name, env in (("default", {}), ("fuse256", {"ZV_FUSE256": "1"}), ("no_triple", {"ZV_NO_TRIPLE": "1"}), ("no_fuse", {"ZV_NO_FUSE": "1"}),
                      ("no_merge", {"ZV_NO_MERGE": "1"}), ("fuse256_no_merge", {"ZV_FUSE256": "1", "ZV_NO_MERGE": "1"}),
                      ("merge", {"ZV_MERGE_ALWAYS": "1"}), ("fuse256_merge", {"ZV_FUSE256": "1", "ZV_MERGE_ALWAYS": "1"}),
                      ("merge_in_one_workgroup", {"ZV_MERGE_ALWAYS": "1", "ZV_MERGE_SEQ": "0"}),
                      ("fuse256_merge_in_one_workgroup", {"ZV_FUSE256": "1", "ZV_MERGE_ALWAYS": "1", "ZV_MERGE_SEQ": "0"}),
                      ("fuse256_merge_mt3", {"ZV_FUSE256": "1", "ZV_MERGE_ALWAYS": "1", "ZV_PAIR_MT": "3"}),
                      ("pair64_ring_merge", {"ZV_PAIR64_RING": "2", "ZV_MERGE_ALWAYS": "1"}),
                      ("single_loop_everywhere", {"ZV_CONV_SINGLE": "2"}), ("no_single_loop", {"ZV_CONV_SINGLE": "0"}),
                      ("block_v1", {"ZV_TRIPLE_V2": "0"}), ("block_v2", {"ZV_TRIPLE_V2": "2"}),
                      ("block_v2_512", {"ZV_TRIPLE_V2": "3"}), ("block_v2_512_one_weight_buffer", {"ZV_TRIPLE_V2": "3", "ZV_TRIPLE_DB": "0"}),
                      ("block_v2_not_interleaved", {"ZV_TRIPLE_V2": "2", "ZV_TRIPLE_INTERLEAVE": "0"}),
                      ("pair64_ring", {"ZV_PAIR64_RING": "2"}), ("pair64_ring_no_merge", {"ZV_PAIR64_RING": "2", "ZV_NO_MERGE": "1"}),
                      ("pair64_no_ring", {"ZV_PAIR64_RING": "0"}),
                      ("upsample_gemm", {"ZV_UP_GEMM": "2", "ZV_CONV_GEMM": "2"}), ("upsample_no_gemm", {"ZV_UP_GEMM": "0"}),
                      ("block64_3_merge", {"ZV_BLOCK64": "-3", "ZV_PAIR64_RING": "2", "ZV_MERGE_ALWAYS": "1"}),
                      ("block64_3_no_merge", {"ZV_BLOCK64": "-3", "ZV_PAIR64_RING": "2"}), ("block64_11", {"ZV_BLOCK64": "-11", "ZV_PAIR64_RING": "2"}),
                      ("no_block64", {"ZV_BLOCK64": "0", "ZV_PAIR64_RING": "2"}),
                      ("upsample_stream", {"ZV_CONV_STREAM": "2"}), ("upsample_no_stream", {"ZV_CONV_STREAM": "0"}),
                      # round 4: the fused batch kernels run on v_mfma_f32_16x16x32_f16, the generic conv kernel ("no_fuse") and the
                      # single-utterance whole-block kernel on 32x32x16: one k-ordered chain, two instruction shapes, the same bits
                      ("pair_mt4", {"ZV_PAIR_MT": "4", "ZV_FUSE256": "1"}),
                      ("pair_no_ring_no_triple", {"ZV_PAIR64_RING": "0", "ZV_NO_TRIPLE": "1", "ZV_MERGE_ALWAYS": "1"}),
                      # the single-utterance conv form (loader waves, two LDS tiles) with its channel groups dealt / not dealt over the XCDs
                      ("single_loop_everywhere_plain_grid", {"ZV_CONV_SINGLE": "2", "ZV_CONV_XCD": "0"}), ("plain_grid", {"ZV_CONV_XCD": "0"})):
        with capi.switches(**{k: int(v) for k, v in env.items()}):      # some switches are sampled when the model is built, some at every launch
            m = capi.Model(path, 0)
            outs[name] = m.vocode(mel)
            m.close()
    for name, w in outs.items():
        assert np.array_equal(w, outs["default"]), name


@pytest.mark.parametrize("chunk", [64, 100, 511, 512, 4096])
def test_streaming_vocoder_is_bit_exact(medium, chunk):
    """zv_vocode_stream: chunks of `chunk` frames with zv_vocoder_halo_frames() frames of context per side must
    reproduce zv_vocode of the whole utterance bit for bit, in order, without gaps"""
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    T = 512
    mel = synth.vocoder_mel(g, tensors, 52, T)
    full = model.vocode(mel)
    H = model.vocoder_halo_frames()
    assert 16 <= H <= 32
    chunks = model.vocode_stream(mel, chunk)
    assert [c[0] for c in chunks] == [a * g.hop_size for a in range(0, T, chunk)]
    got = np.concatenate([c[1] for c in chunks])
    assert got.shape == full.shape and np.array_equal(got, full)


def test_config4_batch_mixed_lengths_full_size(medium):
    """configs[3]: 32 mixed-length utterances (32..256 phonemes), T = 1 024 frames each, as ONE launch per kernel
    (segment tables in HBM); every utterance must equal its stand-alone run bit for bit (no batch padding, nothing
    leaks between neighbouring utterances of the concatenated buffers), eager and as a replayed hipGraph"""
    from zerovox_cpp_amd import sharding, synth
    model, g, tensors = medium
    lens = sharding.mixed_length_batch(3, 32)
    assert min(lens) >= 32 and max(lens) <= 256
    utts = []
    for u, n in enumerate(lens):
        ids, puncts, style = synth.encoder_inputs(g, 300 + u, n)
        utts.append((ids, puncts, style, 1024))
    got = model.synthesize_batch(utts)
    model.set_graph_mode(True)
    got_g = model.synthesize_batch(utts)          # capture
    got_g = model.synthesize_batch(utts)          # replay
    model.set_graph_mode(False)
    for (ids, puncts, style, T), (wav, nf), (wav_g, nf_g) in zip(utts, got, got_g):
        ref, nf_ref = model.synthesize(ids, puncts, style, T)
        assert nf == nf_ref == nf_g and 0 < nf <= T and np.isfinite(wav).all()
        assert np.array_equal(wav, ref) and np.array_equal(wav_g, ref)


def test_maximum_sizes(medium):
    """the largest shapes the checkpoint admits: as many phonemes as the position table has rows (max_seq_len + 1) and
    T = max_seq_len frames, end to end; finite output, regulator saturates at T, deterministic"""
    from zerovox_cpp_amd import synth
    model, g, tensors = medium
    N, T = g.max_seq_len + 1, g.max_seq_len
    ids, puncts, style = synth.encoder_inputs(g, 91, N)
    wav, nf = model.synthesize(ids, puncts, style, T)
    assert wav.shape == (T * g.hop_size,) and np.isfinite(wav).all()
    assert nf == T                                  # ~4 frames per phoneme: the frame budget is exhausted
    wav2, nf2 = model.synthesize(ids, puncts, style, T)
    assert nf2 == nf and np.array_equal(wav, wav2)
