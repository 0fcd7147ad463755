"""shared parity gates of the GPU tests (no test in here)"""
import numpy as np


def encoder_decisions_vs_reference(e, z, nb, T):
    """Integer decisions of the encoder against a reference fixture, gated on the reference semantics' OWN re-association noise
    (tests/golden/make_golden.py add_encoder_margins: our oracle in sequential-f32 order against the reference, stored per
    fixture) — no hand-written tolerance:
      * float predictions within 2 x their floor (energy away from pitch flips: the energy predictor reads x + pitch embedding
        through two k = 3 convs, reference src/fs2encoder.cpp:569-572);
      * every flipped decision moved by ONE step and sat, in the reference, no further from its rounding boundary than 2 x the
        floor of the value it rounds (a flip elsewhere is an error, not noise); an energy flip may instead sit within two rows of
        a pitch flip;
      * no more flips than 2 x the floor's own count + 2."""
    N = len(z["logdur"])
    f_ld, f_p, f_e = float(z["floor_logdur_max"]), float(z["floor_pitch_max"]), float(z["floor_energy_max"])
    ld_err = float(np.max(np.abs(e["logdur"] - z["logdur"])))
    p_err = float(np.max(np.abs(e["pitch"] - z["pitch"])))
    dur_x = np.exp(z["logdur"].astype(np.float64)) - 1 + 0.5                     # the reference's value before (int)
    dur_g = (np.exp(e["logdur"].astype(np.float64)) - 1 + 0.5).astype(np.int64)
    dur_r = dur_x.astype(np.int64)
    dflip = dur_g != dur_r
    pflip = e["pitch_bucket"] != z["pitch_bucket"]
    eflip = e["energy_bucket"] != z["energy_bucket"]
    near = np.convolve(pflip.astype(np.int32), np.ones(5, np.int32), mode="same") > 0
    e_err = float(np.max(np.abs(e["energy"][~near] - z["energy"][~near]))) if (~near).any() else 0.0
    print(f"   N={N}: logdur err {ld_err:.2e} (floor {f_ld:.2e}), pitch err {p_err:.2e} ({f_p:.2e}), energy err away from pitch flips "
          f"{e_err:.2e} ({f_e:.2e}); flips dur {int(dflip.sum())} (floor {int(z['floor_dur_flips'])}) pitch {int(pflip.sum())} "
          f"({int(z['floor_pitch_flips'])}) energy {int(eflip.sum())} ({int(z['floor_energy_flips'])}); frames {e['n_frames']} vs {int(z['n_frames'])}")
    assert ld_err <= 2.0 * f_ld + 1e-5 and p_err <= 2.0 * f_p + 1e-5 and e_err <= 2.0 * f_e + 1e-5
    # distance of the reference's pre-rounding value from the boundary a flip crossed
    dist = lambda x: np.abs(x - np.round(x))
    d_dur = dist(dur_x)                                    # frames; d(dur)/d(logdur) = exp(logdur)
    band_dur = 2.0 * f_ld * np.exp(z["logdur"].astype(np.float64)) + 1e-6
    assert np.all(np.abs(dur_g - dur_r) <= 1) and np.all(d_dur[dflip] <= band_dur[dflip]), "a duration flipped away from a rounding boundary"
    xp = z["pitch"].astype(np.float64) * nb + 0.5
    assert np.all(np.abs(e["pitch_bucket"].astype(np.int64) - z["pitch_bucket"])[pflip] <= 1)
    assert np.all(dist(xp)[pflip] <= 2.0 * f_p * nb + 1e-6), "a pitch bucket flipped away from a boundary"
    xe = z["energy"].astype(np.float64) * nb + 0.5
    tie = dist(xe) <= 2.0 * f_e * nb + 1e-6
    assert np.all(tie[eflip] | near[eflip]), "an energy bucket flipped away from a boundary and away from any pitch flip"
    assert int(dflip.sum()) <= 2 * int(z["floor_dur_flips"]) + 2
    assert int(pflip.sum()) <= 2 * int(z["floor_pitch_flips"]) + 2
    assert int(eflip.sum()) <= 2 * int(z["floor_energy_flips"]) + 2
    assert min(e["n_frames"], int(z["n_frames"])) == T or abs(e["n_frames"] - int(z["n_frames"])) <= int(dflip.sum())
    return int(dflip.sum()), int(pflip.sum()), int(eflip.sum())
