"""-m gpu: HiFi-GAN vocoder through the C-ABI vs the CPU oracle (teacher-forced mel input).

Gate (BASELINE.json north_star): waveform RMS error <= 1e-4 against the reference semantics.  The
oracle is bit-exact against the compiled reference (tests/test_oracle_vs_reference.py), so distance to
the oracle IS distance to the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WAV_RMS_GATE = 1e-4


def _rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


@pytest.fixture(scope="module")
def models(ckpt):
    from zerovox_cpp_amd import capi
    cache = {}

    def get(name):
        if name not in cache:
            path, g, tensors = ckpt(name)
            cache[name] = (capi.Model(path, 0), g, tensors)
        return cache[name]

    yield get
    for m, _, _ in cache.values():
        m.close()


@pytest.mark.parametrize("geom,T", [("tiny", 16), ("tiny", 37), ("small", 64), ("small", 1), ("medium", 24)])
def test_vocoder_matches_oracle(models, geom, T):
    from zerovox_cpp_amd import synth
    from oracle import zvoracle
    model, g, tensors = models(geom)
    mel = synth.vocoder_mel(g, tensors, 7, T)
    wav = model.vocode(mel)
    ref = zvoracle.Oracle(tensors).vocoder(mel)
    assert wav.shape == ref.shape == (T * g.hop_size,)
    assert np.isfinite(wav).all()
    err = _rms(wav - ref)
    print(f"{geom} T={T}: wav rms err {err:.3e}, max {np.max(np.abs(wav - ref)):.3e}, signal rms {_rms(ref):.3f}")
    assert err <= WAV_RMS_GATE


def test_vocoder_length_sweep(models):
    """ragged lengths around every tile boundary of the kernels (32-row MFMA tiles, the fused kernels' 54..62-row and
    136..232-row output tiles, single-tile and multi-tile sequences): each length against the oracle, and each as a
    prefix of a longer utterance (bit-exact: the receptive field does not reach the cut)"""
    from zerovox_cpp_amd import synth
    from oracle import zvoracle
    for geom, lengths in (("tiny", (2, 3, 5, 7, 9, 11, 13, 27, 28, 29, 47, 55, 63)), ("small", (2, 3, 7, 9, 10, 11, 19, 25, 33))):
        model, g, tensors = models(geom)
        orc = zvoracle.Oracle(tensors)
        H = model.vocoder_halo_frames()
        Tmax = max(lengths) + H + 8
        mel = synth.vocoder_mel(g, tensors, 17, Tmax)
        full = model.vocode(mel)
        worst = 0.0
        for T in lengths:
            wav = model.vocode(mel[:T])
            ref = orc.vocoder(mel[:T])
            assert wav.shape == ref.shape and np.isfinite(wav).all()
            worst = max(worst, _rms(wav - ref))
            assert _rms(wav - ref) <= WAV_RMS_GATE, (geom, T)
            # chunk [0, T) of the long utterance with halo == the first T frames of vocode(mel[:T + H])
            ctx = model.vocode(mel[: T + H])
            assert np.array_equal(ctx[: T * g.hop_size], full[: T * g.hop_size]), (geom, T)
        print(f"{geom}: worst wav rms err over {len(lengths)} lengths {worst:.3e}")


def test_gpu_vs_live_reference_fresh_seed(ckpt, tmp_path):
    """no oracle, no committed fixture in between: the compiled reference itself (oracle/_ref/zvref, which travels to
    the GPU box as a prebuilt binary) is run on a fresh seed and all three stages are compared with it directly"""
    from zerovox_cpp_amd import capi, gguf, synth
    from oracle import zvoracle
    if not zvoracle.have_reference():
        pytest.skip("oracle/_ref/zvref not built on this machine")
    g = synth.SMALL
    path = str(tmp_path / "fresh.gguf")
    synth.write_checkpoint(path, g, 4321)
    _, tensors = gguf.read_gguf(path)
    model = capi.Model(path, 0)
    T, N = 72, 18
    mel = synth.vocoder_mel(g, tensors, 23, T)
    r = zvoracle.run_reference(path, T=T, voc=mel)
    wav = model.vocode(mel)
    err = _rms(wav - r["wav"])
    print(f"fresh seed, vocoder vs live reference: wav rms err {err:.3e} (signal rms {_rms(r['wav']):.3f})")
    assert err <= WAV_RMS_GATE
    ids, puncts, style = synth.encoder_inputs(g, 24, N)
    hid = synth.decoder_hidden(g, 25, T)
    r = zvoracle.run_reference(path, T=T, dec=(hid, style))
    m = model.decode(hid, style)
    d = m - r["mel"].reshape(m.shape)
    print(f"fresh seed, decoder vs live reference: mel rms err {_rms(d):.3e} max {np.max(np.abs(d)):.3e}")
    assert _rms(d) <= 3e-3 and np.max(np.abs(d)) <= 2e-2
    r = zvoracle.run_reference(path, T=T, N=N, enc=(ids, puncts, style), E=g.E)
    e = model.encode(ids, puncts, style, T)
    ld = float(np.max(np.abs(e["logdur"] - r["logdur"])))
    print(f"fresh seed, encoder vs live reference: logdur err {ld:.3e}, frames {e['n_frames']} vs {r['n_frames']}")
    assert ld <= 5e-3 and abs(int(e["n_frames"]) - int(r["n_frames"])) <= 2
    model.close()
