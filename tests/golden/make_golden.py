#!/usr/bin/env python3
"""Generate the golden fixtures from the compiled reference (oracle/_ref/zvref).

Run in the build container only (it needs /root/reference to have been compiled by `make -C oracle ref`):
    python tests/golden/make_golden.py
Fixtures hold inputs' *recipes* (geometry, seeds — inputs are regenerated bit-identically by
zerovox.cpp_amd/synth.py) and the reference's OUTPUTS: in full for the small geometries, as strided
samples + SHA-256 of the full f32 buffer for the full-size configs of BASELINE.json.  ISA of the
reference build: x86-64-v3 (AVX2+FMA+F16C), 4 threads (results do not depend on the thread count).
Also slices the reference's only own known-answer data, utils/norm1dexample.json (InstanceNorm1d pair).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from zerovox_cpp_amd import gguf, synth  # noqa: E402
from oracle import zvoracle  # noqa: E402

SEED_W = 1234
STRIDE = 61      # prime stride for the sampled goldens


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case(geom_name, T, N, full, tmp="/tmp"):
    g = synth.GEOMETRIES[geom_name]
    path = os.path.join(tmp, f"golden_{geom_name}.gguf")
    synth.write_checkpoint(path, g, SEED_W)
    _, tensors = gguf.read_gguf(path)
    mel_in = synth.vocoder_mel(g, tensors, 7, T)
    hid_in = synth.decoder_hidden(g, 11, T)
    ids, puncts, style = synth.encoder_inputs(g, 5, N)
    v = zvoracle.run_reference(path, T=T, voc=mel_in)
    d = zvoracle.run_reference(path, T=T, dec=(hid_in, style))
    e = zvoracle.run_reference(path, T=T, N=N, enc=(ids, puncts, style), E=g.E)
    out = dict(geometry=geom_name, seed_w=SEED_W, T=T, N=N, seed_mel=7, seed_hidden=11, seed_enc=5, stride=STRIDE,
               wav_sha256=sha(v["wav"]), mel_sha256=sha(d["mel"]), hidden_sha256=sha(e["hidden"]),
               features_sha256=sha(e["features"]), logdur=e["logdur"], energy=e["energy"],
               pitch_bucket=e["pitch_bucket"], energy_bucket=e["energy_bucket"], n_frames=e["n_frames"],
               wav_rms=float(np.sqrt(np.mean(v["wav"].astype(np.float64) ** 2))),
               mel_rms=float(np.sqrt(np.mean(d["mel"].astype(np.float64) ** 2))))
    if full:
        out.update(wav=v["wav"], mel=d["mel"], hidden=e["hidden"], features=e["features"])
    else:
        out.update(wav_samples=v["wav"][::STRIDE].copy(), mel_samples=d["mel"].reshape(-1)[::STRIDE].copy(),
                   hidden_samples=e["hidden"].reshape(-1)[::STRIDE].copy(),
                   features_samples=e["features"].reshape(-1)[::STRIDE].copy())
    np.savez_compressed(os.path.join(HERE, f"{geom_name}_T{T}_N{N}.npz"), **out)
    print(geom_name, T, N, "frames", e["n_frames"], "wav rms", out["wav_rms"])
    os.remove(path)


def demo_case(tmp="/tmp"):
    """ZeroVOXModel::eval() of the reference (src/zerovox.cpp:198-335): its hard-coded utterance (120 phonemes, the
    528-float style vector) through encoder -> decoder -> vocoder back to back at T = max_seq_len, on the synthetic
    medium checkpoint.  The utterance comes from the product library's zv_demo_utterance (data extracted from the
    reference source by scripts/extract_demo_utterance.py)."""
    from zerovox_cpp_amd import capi
    g = synth.MEDIUM
    path = os.path.join(tmp, "golden_medium.gguf")
    synth.write_checkpoint(path, g, SEED_W)
    ids, puncts, style = capi.demo_utterance()
    T = g.max_seq_len
    r = zvoracle.run_reference_chain(path, ids, puncts, style, T=T)
    # (the pitch prediction's buffer is recycled by ggml's graph allocator before the graph ends: not a valid tap)
    out = dict(geometry="medium", seed_w=SEED_W, T=T, N=len(ids), stride=STRIDE, n_frames=r["n_frames"], logdur=r["logdur"],
               energy=r["energy"], pitch_bucket=r["pitch_bucket"], energy_bucket=r["energy_bucket"],
               hidden_sha256=sha(r["hidden"]), mel_sha256=sha(r["mel"]), wav_sha256=sha(r["wav"]),
               features_sha256=sha(r["features"]),
               wav_samples=r["wav"][::STRIDE].copy(), mel_samples=r["mel"].reshape(-1)[::STRIDE].copy(),
               wav_rms=float(np.sqrt(np.mean(r["wav"].astype(np.float64) ** 2))))
    np.savez_compressed(os.path.join(HERE, "demo_medium_T%d.npz" % T), **out)
    print("demo utterance: frames", r["n_frames"], "wav rms", out["wav_rms"])
    os.remove(path)


def numph_case(tmp="/tmp"):
    """FS2Encoder::eval with num_phonemes < max_n_phonemes (reference src/fs2encoder.cpp:594-650): the graph encodes all
    max_n_phonemes tokens, the length regulator walks the first num_phonemes."""
    g = synth.SMALL
    path = os.path.join(tmp, "golden_small.gguf")
    synth.write_checkpoint(path, g, SEED_W)
    N, num, T = 16, 9, 64
    ids, puncts, style = synth.encoder_inputs(g, 5, N)
    e = zvoracle.run_reference(path, T=T, N=N, enc=(ids, puncts, style), E=g.E, num_phonemes=num)
    full = zvoracle.run_reference(path, T=T, N=N, enc=(ids, puncts, style), E=g.E)
    assert np.array_equal(e["features"], full["features"]) and e["n_frames"] < full["n_frames"]
    np.savez_compressed(os.path.join(HERE, "small_T64_N16_num9.npz"), geometry="small", seed_w=SEED_W, T=T, N=N, num=num,
                        seed_enc=5, hidden=e["hidden"], n_frames=e["n_frames"], logdur=e["logdur"],
                        features_sha256=sha(e["features"]))
    print("num_phonemes case: frames", e["n_frames"], "of", full["n_frames"])
    os.remove(path)


def robust_case(tmp="/tmp", search=0):
    """An UN-FORCED end-to-end golden: ZeroVOXModel::eval's chain (src/zerovox.cpp:326-334) on an utterance whose integer
    decisions do not depend on summation order.  Geometry medium8 (8 variance bins); the utterance seed was found by
    search (search=N re-runs it over N seeds with our oracle, which reproduces the reference bit for bit): every pitch /
    energy prediction >= 0.1 bin from a bucket boundary, every duration >= 0.04 frames from a rounding boundary.  Besides
    the reference's outputs the fixture stores the reference semantics' own un-forced re-association noise (our oracle in
    sequential-f32 order against the reference): the floor the GPU is gated against."""
    g = synth.GEOMETRIES["medium8"]
    path = os.path.join(tmp, "golden_medium8.gguf")
    synth.write_checkpoint(path, g, SEED_W)
    _, tensors = gguf.read_gguf(path)
    N, T, seed = 16, 96, 11167
    nb = g.ve_n_bins

    def margins(e):
        def bm(p):
            x = p.astype(np.float64) * (nb - 1) + 0.5
            fr = x - np.floor(x)
            m = np.minimum(fr, 1 - fr)
            m = np.where(x < 0, -x, m)
            return np.where(x >= nb, x - nb + 1, m)
        d = np.exp(e["logdur"].astype(np.float64)) - 1.0 + 0.5
        fd = d - np.floor(d)
        return float(bm(e["pitch"]).min()), float(bm(e["energy"]).min()), float(np.where(d < 0, -d, np.minimum(fd, 1 - fd)).min())

    orc = zvoracle.Oracle(tensors, threads=8)
    if search:
        best = (0.0, seed)
        for sd in range(1000, 1000 + search):
            ids, puncts, style = synth.encoder_inputs(g, sd, N)
            mp, me, md = margins(orc.encoder(g, ids, puncts, style, T))
            best = max(best, (min(mp, me, 4 * md), sd))
        seed = best[1]
    ids, puncts, style = synth.encoder_inputs(g, seed, N)
    r = zvoracle.run_reference_chain(path, ids, puncts, style, T=T)
    e = orc.encoder(g, ids, puncts, style, T)        # (the reference's pitch tap is recycled by its graph allocator: ours)
    assert np.array_equal(e["hidden"], r["hidden"]) and e["n_frames"] == r["n_frames"]
    mp, me, md = margins(e)
    alt = zvoracle.Oracle(tensors, threads=8, order=zvoracle.ORDER_SEQ_F32)
    ea = alt.encoder(g, ids, puncts, style, T)
    mela = alt.decoder(ea["hidden"], style)
    wava = alt.vocoder(mela)
    assert ea["n_frames"] == r["n_frames"] and np.array_equal(ea["pitch_bucket"], r["pitch_bucket"]) and np.array_equal(ea["energy_bucket"], r["energy_bucket"])
    rms = lambda a: float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))
    out = dict(geometry="medium8", seed_w=SEED_W, T=T, N=N, seed_enc=seed, n_frames=r["n_frames"], logdur=r["logdur"],
               pitch=e["pitch"], energy=r["energy"], pitch_bucket=r["pitch_bucket"], energy_bucket=r["energy_bucket"],
               margin_pitch_bins=mp, margin_energy_bins=me, margin_duration_frames=md,
               hidden_sha256=sha(r["hidden"]), mel=r["mel"], wav=r["wav"], wav_rms=rms(r["wav"]), mel_rms=rms(r["mel"]),
               floor_wav_rms=rms(wava - r["wav"]), floor_wav_max=float(np.max(np.abs(wava - r["wav"]))),
               floor_mel_rms=rms(mela - r["mel"]), floor_mel_max=float(np.max(np.abs(mela - r["mel"]))),
               floor_hidden_max=float(np.max(np.abs(ea["hidden"] - r["hidden"]))))
    np.savez_compressed(os.path.join(HERE, "robust_medium8_T%d_N%d.npz" % (T, N)), **out)
    print("robust un-forced case: seed %d frames %d margins pitch %.3f energy %.3f bins, duration %.3f frames; un-forced floor wav rms %.3e mel max %.3e"
          % (seed, r["n_frames"], mp, me, md, out["floor_wav_rms"], out["floor_mel_max"]))
    os.remove(path)


def add_floors(tmp="/tmp"):
    """the reference semantics' own re-association noise on each full-size fixture's stage inputs (our oracle in sequential-f32
    order against the stored reference samples): the floor the GPU's stage outputs are gated against (1.5 x), stored in
    the fixture itself so that the tests carry no hand-written tolerance"""
    g = synth.MEDIUM
    path = os.path.join(tmp, "golden_medium.gguf")
    synth.write_checkpoint(path, g, SEED_W)
    _, tensors = gguf.read_gguf(path)
    alt = zvoracle.Oracle(tensors, threads=8, order=zvoracle.ORDER_SEQ_F32)
    rms = lambda a: float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))
    for name in ("medium_T512_N64.npz", "medium_T512_N128.npz", "medium_T1024_N256.npz"):
        f = os.path.join(HERE, name)
        z = dict(np.load(f))
        T, s_ = int(z["T"]), int(z["stride"])
        mel_in = synth.vocoder_mel(g, tensors, int(z["seed_mel"]), T)
        hid_in = synth.decoder_hidden(g, int(z["seed_hidden"]), T)
        _, _, style = synth.encoder_inputs(g, int(z["seed_enc"]), int(z["N"]))
        dm = alt.decoder(hid_in, style).reshape(-1)[::s_] - z["mel_samples"]
        dw = alt.vocoder(mel_in)[::s_] - z["wav_samples"]
        z.update(floor_mel_max=float(np.max(np.abs(dm))), floor_mel_rms=rms(dm), floor_wav_rms=rms(dw), floor_wav_max=float(np.max(np.abs(dw))))
        np.savez_compressed(f, **z)
        print(name, "floors: mel max %.3e rms %.3e, wav rms %.3e" % (z["floor_mel_max"], z["floor_mel_rms"], z["floor_wav_rms"]))
    # the reference's demo utterance (chain mode: the decoder's input is the reference's own hidden, not regenerable from a
    # seed): teacher-force our oracle with the reference's hidden / mel via the reference chain's outputs
    from zerovox_cpp_amd import capi
    f = os.path.join(HERE, "demo_medium_T1500.npz")
    z = dict(np.load(f))
    ids, puncts, style = capi.demo_utterance()
    T, s_ = int(z["T"]), int(z["stride"])
    r = zvoracle.run_reference_chain(path, ids, puncts, style, T=T)
    dm = alt.decoder(r["hidden"], style).reshape(-1)[::s_] - z["mel_samples"]
    dw = alt.vocoder(r["mel"])[::s_] - z["wav_samples"]
    z.update(floor_mel_max=float(np.max(np.abs(dm))), floor_mel_rms=rms(dm), floor_wav_rms=rms(dw), floor_wav_max=float(np.max(np.abs(dw))))
    np.savez_compressed(f, **z)
    print("demo floors (stages teacher-forced with the reference's own hidden / mel): mel max %.3e rms %.3e, wav rms %.3e"
          % (z["floor_mel_max"], z["floor_mel_rms"], z["floor_wav_rms"]))
    os.remove(path)


def add_encoder_margins(tmp="/tmp"):
    """what the reference semantics' own re-association noise does to the ENCODER of each full-size fixture (our oracle in
    sequential-f32 order against the reference on the same ids): the floors of the float predictions and of the integer
    decisions.  The fixtures also get the reference's float pitch predictions, so that a test can measure how far every flipped
    bucket / duration of the GPU sat from a rounding boundary IN THE REFERENCE — the gates of tests/test_gpu_full_size.py and
    tests/test_gpu_round2.py carry no hand-written tolerance any more."""
    from zerovox_cpp_amd import capi
    g = synth.MEDIUM
    path = os.path.join(tmp, "golden_medium.gguf")
    synth.write_checkpoint(path, g, SEED_W)
    _, tensors = gguf.read_gguf(path)
    alt = zvoracle.Oracle(tensors, threads=8, order=zvoracle.ORDER_SEQ_F32)
    exact = zvoracle.Oracle(tensors, threads=8)          # the reference's summation order: reproduces it bit for bit

    def margins(ref, a, ex):
        # the float pitch prediction comes from the bit-exact oracle: the reference's own pitch tensor is recycled by ggml's
        # graph allocator before the harness can read it (its buckets, energy, log-durations and hidden are read and agree)
        assert np.array_equal(ex["pitch_bucket"], ref["pitch_bucket"]) and np.array_equal(ex["energy"], ref["energy"])
        assert np.array_equal(ex["logdur"], ref["logdur"]) and np.array_equal(ex["energy_bucket"], ref["energy_bucket"])
        ref = dict(ref, pitch=ex["pitch"])
        dur_r = np.exp(ref["logdur"].astype(np.float64)) - 1 + 0.5
        dur_a = np.exp(a["logdur"].astype(np.float64)) - 1 + 0.5
        pflip = a["pitch_bucket"] != ref["pitch_bucket"]
        near = np.convolve(pflip.astype(np.int32), np.ones(5, np.int32), mode="same") > 0      # the energy predictor sees two k = 3 convs of x + pitch embedding
        clean = ~near
        return dict(pitch=ref["pitch"], energy=ref["energy"],
                    floor_logdur_max=float(np.max(np.abs(a["logdur"] - ref["logdur"]))),
                    floor_pitch_max=float(np.max(np.abs(a["pitch"] - ref["pitch"]))),
                    floor_energy_max=float(np.max(np.abs(a["energy"][clean] - ref["energy"][clean]))) if clean.any() else 0.0,
                    floor_dur_flips=int(np.sum(dur_r.astype(np.int64) != dur_a.astype(np.int64))),
                    floor_pitch_flips=int(pflip.sum()),
                    floor_energy_flips=int(np.sum(a["energy_bucket"] != ref["energy_bucket"])),
                    floor_frames=int(abs(int(a["n_frames"]) - int(ref["n_frames"]))))

    for name in ("medium_T512_N64.npz", "medium_T512_N128.npz", "medium_T1024_N256.npz"):
        f = os.path.join(HERE, name)
        z = dict(np.load(f))
        T, N = int(z["T"]), int(z["N"])
        ids, puncts, style = synth.encoder_inputs(g, int(z["seed_enc"]), N)
        ref = zvoracle.run_reference(path, T=T, N=N, enc=(ids, puncts, style), E=g.E)
        assert np.array_equal(ref["logdur"], z["logdur"]) and np.array_equal(ref["pitch_bucket"], z["pitch_bucket"])
        z.update(margins(ref, alt.encoder(g, ids, puncts, style, T), exact.encoder(g, ids, puncts, style, T)))
        np.savez_compressed(f, **z)
        print(name, {k: z[k] for k in z if k.startswith("floor_") and "mel" not in k and "wav" not in k})
    f = os.path.join(HERE, "demo_medium_T1500.npz")
    z = dict(np.load(f))
    ids, puncts, style = capi.demo_utterance()
    T = int(z["T"])
    ref = zvoracle.run_reference(path, T=T, N=len(ids), enc=(ids, puncts, style), E=g.E)
    assert np.array_equal(ref["logdur"], z["logdur"]) and np.array_equal(ref["pitch_bucket"], z["pitch_bucket"])
    z.update(margins(ref, alt.encoder(g, ids, puncts, style, T), exact.encoder(g, ids, puncts, style, T)))
    np.savez_compressed(f, **z)
    print("demo", {k: z[k] for k in z if k.startswith("floor_") and "mel" not in k and "wav" not in k})
    os.remove(path)


def norm_kat():
    src = "/root/reference/utils/norm1dexample.json"
    if not os.path.exists(src):
        return
    j = json.load(open(src))
    x_in, x_out = np.array(j["x_in"], np.float32)[0], np.array(j["x_out"], np.float32)[0]
    w, b = np.array(j["weight"], np.float32), np.array(j["bias"], np.float32)
    sel = np.arange(0, x_in.shape[0], 16)
    np.savez_compressed(os.path.join(HERE, "instnorm1d_kat.npz"), x_in=x_in[sel], x_out=x_out[sel], weight=w[sel], bias=b[sel],
                        channels=sel)
    print("instnorm KAT:", x_in[sel].shape)


if __name__ == "__main__":
    if not zvoracle.have_reference():
        sys.exit("oracle/_ref/zvref missing: run `make -C oracle ref` first")
    case("tiny", 40, 10, full=True)
    case("small", 64, 16, full=True)
    case("medium", 512, 64, full=False)      # BASELINE.json configs[0] (N=64) and [1] (vocoder, 512 frames)
    case("medium", 512, 128, full=False)     # configs[2]
    case("medium", 1024, 256, full=False)    # configs[3]: T = 1 024, its longest utterance (256 phonemes)
    norm_kat()
    demo_case()
    numph_case()
    robust_case()
    add_floors()
    add_encoder_margins()
