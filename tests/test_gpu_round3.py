"""-m gpu: round-3 parity cases.

(1) UN-FORCED end to end against the reference itself.  Stage-wise teacher forcing (every other GPU test) is what SURVEY.md
    §8d prescribes because on the synthetic weights 10-20 % of the pitch / energy buckets sit within summation-order noise
    of a boundary.  Here the utterance is chosen so that no integer decision does (geometry medium8, every prediction >= 0.1
    bin / 0.04 frames from a boundary — tests/golden/make_golden.py robust_case), so zv_synthesize's waveform can be compared
    with ZeroVOXModel::eval's directly: same frames, same buckets, and the float outputs within 1.5 x the reference
    semantics' own un-forced re-association noise stored in the fixture.  (That floor, 2.9e-4 RMS, is above north_star's
    1e-4: un-forced, the decoder's f16 operand rounding amplifies any re-association — SURVEY.md Appx C-H13.)
(2) The longest encoder the sinusoid table allows (N = max_seq_len + 1 = 1 501 phonemes, scalar-attention fallback) once
    against the oracle.
(3) zv_debug_layer kinds for the layers that stage-level gates only cover at the noise floor.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


def test_unforced_end_to_end_vs_reference_on_a_decision_robust_utterance(ckpt):
    from zerovox_cpp_amd import capi, synth
    z = np.load(os.path.join(GOLD, "robust_medium8_T96_N16.npz"))
    path, g, tensors = ckpt(str(z["geometry"]), int(z["seed_w"]))
    N, T = int(z["N"]), int(z["T"])
    ids, puncts, style = synth.encoder_inputs(g, int(z["seed_enc"]), N)
    m = capi.Model(path, 0)
    try:
        e = m.encode(ids, puncts, style, T)
        # integer decisions: exactly the reference's (margins stored in the fixture: >= 0.1 bin, >= 0.04 frames)
        assert e["n_frames"] == int(z["n_frames"])
        assert np.array_equal(e["pitch_bucket"], z["pitch_bucket"]) and np.array_equal(e["energy_bucket"], z["energy_bucket"])
        assert np.max(np.abs(e["pitch"] - z["pitch"])) * 7 < float(z["margin_pitch_bins"])
        assert np.max(np.abs(e["energy"] - z["energy"])) * 7 < float(z["margin_energy_bins"])
        wav, nf = m.synthesize(ids, puncts, style, T)
        assert nf == int(z["n_frames"]) and np.isfinite(wav).all()
        err, emax = _rms(wav - z["wav"]), float(np.max(np.abs(wav - z["wav"])))
        floor, fmax = float(z["floor_wav_rms"]), float(z["floor_wav_max"])
        print(f"un-forced zv_synthesize vs ZeroVOXModel::eval: wav rms err {err:.3e} max {emax:.3e} (signal rms {float(z['wav_rms']):.3f}); "
              f"reference semantics' own un-forced re-association noise {floor:.3e} / {fmax:.3e} -> ratio {err / floor:.2f}")
        assert err <= 1.5 * floor and emax <= 2.0 * fmax
        # the mel the chain went through (decoder on the GPU's own hidden) against the reference's
        mel = m.decode(e["hidden"], style)
        dm = mel - z["mel"]
        print(f"   mel: max {np.max(np.abs(dm)):.3e} rms {_rms(dm):.3e}; floor max {float(z['floor_mel_max']):.3e} rms {float(z['floor_mel_rms']):.3e}")
        assert _rms(dm) <= 1.5 * float(z["floor_mel_rms"]) and np.max(np.abs(dm)) <= 1.5 * float(z["floor_mel_max"])
        # a batch of 32 copies: every copy is the stand-alone call bit for bit, hence within the same distance of the reference
        res = m.synthesize_batch([(ids, puncts, style, T)] * 32)
        for w, nfb in res:
            assert nfb == nf and np.array_equal(w, wav)
        # ... and a ragged batch around it (other lengths before and behind) does not move it either
        other = [(*synth.encoder_inputs(g, 700 + k, 9 + 5 * k), 64 + 16 * k) for k in range(4)]
        res = m.synthesize_batch(other[:2] + [(ids, puncts, style, T)] + other[2:])
        assert res[2][1] == nf and np.array_equal(res[2][0], wav)
    finally:
        m.close()


def test_longest_encoder_vs_oracle(ckpt):
    """N = 1 501 phonemes = every row of the sinusoid table (reference src/fs2encoder.cpp:306-324 takes as many positions as
    the table has rows); the attention of 1 501 keys does not fit the matrix-core kernel's LDS and falls back to the
    scalar kernel.  Log-durations and features on the rows whose buckets agree, near-tie accounting for the rest."""
    from zerovox_cpp_amd import capi, synth
    from oracle import zvoracle
    path, g, tensors = ckpt("medium")
    N, T = g.max_seq_len + 1, 1500
    ids, puncts, style = synth.encoder_inputs(g, 77, N)
    m = capi.Model(path, 0)
    try:
        e = m.encode(ids, puncts, style, T)
        with pytest.raises(capi.ZvError):
            m.encode(np.concatenate([ids, ids[:1]]), np.concatenate([puncts, puncts[:1]]), style, T)      # 1 502 > table rows
    finally:
        m.close()
    orc = zvoracle.Oracle(tensors, threads=16)
    r = orc.encoder(g, ids, puncts, style, T)
    alt = zvoracle.Oracle(tensors, threads=16, order=zvoracle.ORDER_SEQ_F32).encoder(g, ids, puncts, style, T)
    ld, ld_floor = float(np.max(np.abs(e["logdur"] - r["logdur"]))), float(np.max(np.abs(alt["logdur"] - r["logdur"])))
    agree = (e["pitch_bucket"] == r["pitch_bucket"]) & (e["energy_bucket"] == r["energy_bucket"])
    agree_alt = (alt["pitch_bucket"] == r["pitch_bucket"]) & (alt["energy_bucket"] == r["energy_bucket"])
    fe = float(np.max(np.abs(e["features"][agree] - r["features"][agree])))
    fa = float(np.max(np.abs(alt["features"][agree_alt] - r["features"][agree_alt])))
    print(f"N = {N}: logdur err {ld:.3e} (floor {ld_floor:.3e}); bucket rows agreeing {int(agree.sum())} (oracle re-ordered: {int(agree_alt.sum())}); "
          f"features on agreeing rows {fe:.3e} (floor {fa:.3e}); frames {e['n_frames']} vs {r['n_frames']}")
    assert ld <= 2.0 * ld_floor + 1e-4
    assert agree.sum() >= 0.9 * agree_alt.sum()
    assert fe <= 2.0 * fa + 1e-4
    # every flipped pitch bucket is a near tie of the reference's own prediction; the energy predictor runs on the pitch-augmented
    # features (reference src/fs2encoder.cpp:569-572) through two k = 3 convs, so an energy flip is either a near tie too or sits
    # within two rows of a pitch flip
    nb = g.ve_n_bins - 1
    pflip = e["pitch_bucket"] != r["pitch_bucket"]
    xp = r["pitch"][pflip].astype(np.float64) * nb + 0.5
    assert np.all(np.abs(xp - np.round(xp)) <= 0.25) and np.all(np.abs(e["pitch_bucket"][pflip] - r["pitch_bucket"][pflip]) <= 1)
    near = np.convolve(pflip.astype(np.int32), np.ones(5, np.int32), mode="same") > 0
    eflip = e["energy_bucket"] != r["energy_bucket"]
    xe = r["energy"].astype(np.float64) * nb + 0.5
    tie = np.abs(xe - np.round(xe)) <= 0.25
    assert np.all(tie[eflip] | near[eflip]), "energy flips away from any pitch flip must be near ties"
    assert min(e["n_frames"], r["n_frames"]) == T or abs(e["n_frames"] - r["n_frames"]) <= max(3, N // 100)
