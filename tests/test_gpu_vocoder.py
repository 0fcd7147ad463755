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
