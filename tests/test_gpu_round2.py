"""GPU: round-2 parity items — integer decisions pinned bit-exact by teacher forcing, FS2Encoder::eval with
num_phonemes < max_n_phonemes, the reference's demo utterance, the frame limit, loader validation, the two attention
kernels, batches of ragged utterances.  Everything goes through the C-ABI (ctypes)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_MODELS = {}


@pytest.fixture(scope="module")
def models(ckpt):
    from zerovox_cpp_amd import capi

    def get(name, seed=1234):
        if (name, seed) not in _MODELS:
            path, g, tensors = ckpt(name, seed)
            _MODELS[(name, seed)] = (capi.Model(path, 0), g, tensors)
        return _MODELS[(name, seed)]
    yield get
    for m, _, _ in _MODELS.values():
        m.close()
    _MODELS.clear()


def _rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


def _bucket(pred, nbins):
    """ggml_zv_mul_clamp_to_i32 (reference src/fs2encoder.cpp:442-474): (int)((double)(x * bin_max) + 0.5), clamped"""
    p = pred.astype(np.float32) * np.float32(nbins - 1)
    return np.clip((p.astype(np.float64) + 0.5).astype(np.int64), 0, nbins - 1).astype(np.int32)      # astype truncates toward 0


@pytest.mark.parametrize("geom,N,T", [("tiny", 10, 40), ("small", 16, 64), ("medium", 64, 512), ("medium", 256, 1024)])
def test_integer_decisions_are_bit_exact_when_teacher_forced(models, geom, N, T):
    """The bar for integer / index work is bit-exact.  The float predictions differ from the reference's by summation
    order, so the integers derived from the GPU's OWN float taps are checked against the reference's integer rules applied
    to those same floats: buckets = the clamp rule on gpu.pitch / gpu.energy, frames = the regulator on gpu.features /
    gpu.logdur (the oracle's regulator = the reference's host loop, src/fs2encoder.cpp:611-654)."""
    from zerovox_cpp_amd import synth
    from oracle import zvoracle
    model, g, tensors = models(geom)
    ids, puncts, style = synth.encoder_inputs(g, 5, N)
    e = model.encode(ids, puncts, style, T)
    assert np.array_equal(e["pitch_bucket"], _bucket(e["pitch"], g.ve_n_bins))
    assert np.array_equal(e["energy_bucket"], _bucket(e["energy"], g.ve_n_bins))
    orc = zvoracle.Oracle(tensors)
    hid, nf = orc.length_regulator(e["features"], e["logdur"], T)
    assert nf == e["n_frames"] and np.array_equal(hid, e["hidden"])
    # and the features really are encoder output + the bucket rows: recompute the last two adds on the host
    pe, ee = tensors["_pe._var_adapt.pitch_embedding.w"], tensors["_pe._var_adapt.energy_embedding.w"]
    r = orc.encoder(g, ids, puncts, style, T)
    same = (e["pitch_bucket"] == r["pitch_bucket"]) & (e["energy_bucket"] == r["energy_bucket"])
    assert same.sum() >= N // 2
    d = np.abs(e["features"][same] - r["features"][same])
    assert d.max() <= 2e-2, d.max()
    assert pe.shape[0] == g.ve_n_bins == ee.shape[0]


def test_num_phonemes_below_max_vs_reference_golden(models):
    """FS2Encoder::eval(num_phonemes < max_n_phonemes) (reference src/fs2encoder.cpp:594-650): every token is encoded and
    attended to, the regulator walks the first num_phonemes — against the compiled reference's output"""
    from zerovox_cpp_amd import synth
    from oracle import zvoracle
    z = np.load(os.path.join(GOLD, "small_T64_N16_num9.npz"))
    model, g, tensors = models("small", int(z["seed_w"]))
    N, num, T = int(z["N"]), int(z["num"]), int(z["T"])
    ids, puncts, style = synth.encoder_inputs(g, int(z["seed_enc"]), N)
    e = model.encode(ids, puncts, style, T, num_phonemes=num)
    full = model.encode(ids, puncts, style, T)
    assert np.array_equal(e["features"], full["features"]) and np.array_equal(e["logdur"], full["logdur"])
    assert e["n_frames"] == int(z["n_frames"]) < full["n_frames"]
    assert np.max(np.abs(e["logdur"] - z["logdur"])) <= 2e-3
    # teacher-forced: the regulator over the first num tokens of the GPU's own taps, bit for bit
    hid, nf = zvoracle.Oracle(tensors).length_regulator(e["features"][:num], e["logdur"][:num], T)
    assert nf == e["n_frames"] and np.array_equal(hid, e["hidden"])
    # against the reference's hidden: rows of tokens whose buckets agree are close, the zero tail is exact
    assert not e["hidden"][e["n_frames"]:].any() and not z["hidden"][int(z["n_frames"]):].any()
    d = np.abs(e["hidden"] - z["hidden"]).max(axis=1)
    assert np.mean(d <= 2e-2) >= 0.5
    with pytest.raises(Exception):
        model.encode(ids, puncts, style, T, num_phonemes=N + 1)


def test_demo_utterance_stage_by_stage_vs_reference(models):
    """ZeroVOXModel::eval() (reference src/zerovox.cpp:198-335) with the reference's own utterance: every stage teacher-forced
    with the reference's values (the oracle reproduces the compiled reference bit for bit on this case:
    tests/test_oracle_golden.py::test_oracle_reproduces_reference_demo_utterance)"""
    from zerovox_cpp_amd import capi
    from oracle import zvoracle
    z = np.load(os.path.join(GOLD, "demo_medium_T1500.npz"))
    model, g, tensors = models("medium", int(z["seed_w"]))
    ids, puncts, style = capi.demo_utterance()
    T, s = int(z["T"]), int(z["stride"])
    e = model.encode(ids, puncts, style, T)
    # float predictions and integer decisions gated on the fixture's own re-association floors, near-tie accounting for every flip
    from parity_helpers import encoder_decisions_vs_reference
    dfl, pfl, efl = encoder_decisions_vs_reference(e, z, g.ve_n_bins - 1, T)
    flips = pfl + efl
    orc = zvoracle.Oracle(tensors)
    r = orc.encoder(g, ids, puncts, style, T)
    mel = model.decode(r["hidden"], style)                       # 386 live frames + 1 114 zero frames: all T are normalised
    dm = mel.reshape(-1)[::s] - z["mel_samples"]
    wav = model.vocode(orc.decoder(r["hidden"], style))
    dw = wav[::s] - z["wav_samples"]
    print(f"demo utterance: frames {e['n_frames']} (reference {int(z['n_frames'])}), bucket flips {flips}, mel err max "
          f"{np.max(np.abs(dm)):.3e} rms {_rms(dm):.3e}, wav rms err {_rms(dw):.3e} (wav rms {float(z['wav_rms']):.3f})")
    # gates: 1.5 x the re-association floor stored in the fixture (our oracle in sequential-f32 order against the reference
    # on the same teacher-forced stage inputs), and the absolute wav gate
    assert _rms(dm) <= 1.5 * float(z["floor_mel_rms"]) and np.max(np.abs(dm)) <= 1.5 * float(z["floor_mel_max"])
    assert _rms(dw) <= 1e-4 and _rms(dw) <= 1.5 * float(z["floor_wav_rms"])
    # the chained call is the three stage calls back to back
    w2, nf2 = model.synthesize(ids, puncts, style, T)
    assert nf2 == e["n_frames"] and np.array_equal(w2, model.vocode(model.decode(e["hidden"], style)))


def test_frame_limit_is_an_error_not_silent_garbage(models):
    from zerovox_cpp_amd import capi, synth
    model, g, tensors = models("tiny")
    lim = model.max_frames()
    assert 1024 <= lim <= 32768
    mel = np.zeros((lim + 1, g.num_mels), np.float32)
    for call in (lambda: model.vocode(mel), lambda: model.decode(np.zeros((lim + 1, g.E), np.float32), np.zeros(g.E, np.float32))):
        with pytest.raises(capi.ZvError) as ei:
            call()
        assert ei.value.status == 5 and "zv_max_frames" in str(ei.value)
    # streaming has no limit on the total, only on a chunk plus its context
    T = 300
    melr = synth.vocoder_mel(g, tensors, 3, T)
    chunks = model.vocode_stream(melr, 64)
    assert np.array_equal(np.concatenate([c for _, c in chunks]), model.vocode(melr))


def test_loader_rejects_tables_and_stage_widths_the_schedule_cannot_run(tmp_path):
    """a checkpoint with fewer embedding rows than ids in use must limit the accepted ids (not read past the
    table); an upsample stage that does not halve the channels must be rejected at load"""
    from zerovox_cpp_amd import capi, gguf, synth
    g = synth.TINY
    tensors = synth.make_tensors(g, 7)
    small = [(n, (a[:100] if n == "_pe._enc.src_word_emb.w" else a[:4] if n == "_pe._enc.punct_embed.w" else a)) for n, a in tensors]
    p1 = str(tmp_path / "small_tables.gguf")
    gguf.write_gguf(p1, g.kv(), small, arch=synth.ARCH)
    m = capi.Model(p1, 0)
    ids, puncts, style = synth.encoder_inputs(g, 5, 8)
    ids, puncts = np.minimum(ids, 99).astype(np.int32), np.minimum(puncts, 3).astype(np.int32)
    assert m.encode(ids, puncts, style, 32)["n_frames"] >= 0
    for bad_ids, bad_p in ((np.where(np.arange(8) == 2, 100, ids).astype(np.int32), puncts), (ids, np.where(np.arange(8) == 5, 4, puncts).astype(np.int32))):
        with pytest.raises(capi.ZvError) as ei:
            m.encode(bad_ids, bad_p, style, 32)
        assert ei.value.status == 5
    m.close()
    # upsample stage 1 keeps the channel count of stage 0 instead of halving it
    C0 = g.voc_channels >> 1
    wide = []
    for n, a in tensors:
        if n == "_meldec.upsamples.1.1.w":
            a = np.zeros((C0, C0, a.shape[2]), np.float16)
        if n == "_meldec.upsamples.1.1.b":
            a = np.zeros(C0, np.float32)
        wide.append((n, a))
    p2 = str(tmp_path / "wide_stage.gguf")
    gguf.write_gguf(p2, g.kv(), wide, arch=synth.ARCH)
    with pytest.raises(capi.ZvError) as ei:
        capi.Model(p2, 0)
    assert ei.value.status == 4 and "halve" in str(ei.value)


def test_matrix_core_attention_equals_scalar_attention_bit_for_bit(ckpt, tmp_path):
    """two attention kernels (f32 matrix cores for batches, scalar fma chains for a single short utterance): every dot
    product is the same k-ordered fma chain and the softmax sum is associated the same way, so whichever the size picks the
    encoder gives the same bits (this is what keeps a batch bit-equal to stand-alone calls)"""
    import subprocess
    import sys
    from zerovox_cpp_amd import synth
    path, g, _ = ckpt("medium")
    outs = {}
    for name, env in (("scalar", {"ZV_ATT_SCALAR": "1"}), ("mfma", {"ZV_ATT_MFMA": "1"})):
        out = str(tmp_path / (name + ".npz"))
        code = ("import sys, numpy as np\nsys.path.insert(0, %r)\nfrom __graft_entry__ import load_package\nload_package()\n"
                "from zerovox_cpp_amd import capi, synth\ng = synth.MEDIUM\nm = capi.Model(%r, 0)\nres = {}\n"
                "for N, T in ((200, 1024), (37, 160), (1, 8)):\n"
                "    ids, puncts, style = synth.encoder_inputs(g, 77 + N, N)\n    e = m.encode(ids, puncts, style, T)\n"
                "    res.update({'%%s_%%d' %% (k, N): np.asarray(v) for k, v in e.items()})\n"
                "np.savez(%r, **res)\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, out))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    assert set(outs["scalar"].files) == set(outs["mfma"].files) and len(outs["mfma"].files) >= 24
    for k in outs["scalar"].files:
        assert np.array_equal(outs["scalar"][k], outs["mfma"][k]), k


def test_ragged_batches_equal_stand_alone_calls(models):
    """utterances of different N AND different T in one launch per kernel (grids are sized for the longest; shorter
    segments leave workgroups that exit), more utterances than one group holds, graph replay across batches of
    different content under the same capacities"""
    from zerovox_cpp_amd import synth
    model, g, tensors = models("small")
    rng = np.random.default_rng(5)
    utts = []
    for u in range(70):                       # > 64: two groups
        n = int(rng.integers(1, 40))
        T = int(rng.integers(1, 200))
        ids, puncts, style = synth.encoder_inputs(g, 900 + u, n)
        utts.append((ids, puncts, style, T))
    ref = [model.synthesize(*u) for u in utts]
    got = model.synthesize_batch(utts)
    for (w, nf), (rw, rnf) in zip(got, ref):
        assert nf == rnf and np.array_equal(w, rw)
    model.set_graph_mode(True)
    try:
        a = model.synthesize_batch(utts[:20])
        b = model.synthesize_batch(utts[20:40])          # other content; same capacities only if the maxima round alike
        c = model.synthesize_batch(utts[:20])
    finally:
        model.set_graph_mode(False)
    for (w, nf), (rw, rnf) in zip(a + b + c, ref[:40] + ref[:20]):
        assert nf == rnf and np.array_equal(w, rw)


def test_batches_in_flight_on_two_lanes_equal_the_synchronous_call(models):
    """zv_synthesize_batch_begin / _end: a serving loop that keeps a batch in flight per lane (the next batch's upload and
    kernels start while the previous one's tail downloads) must return, for every batch, the bits of zv_synthesize_batch;
    large batches (tail groups + copy stream) and small ones, eager and graph replay; misuse is an error, not a hang"""
    from zerovox_cpp_amd import capi, synth
    model, g, tensors = models("medium")
    rng = np.random.default_rng(9)
    batches = []
    for b in range(3):
        utts = []
        for u in range(16):
            n = int(rng.integers(24, 96))
            ids, puncts, style = synth.encoder_inputs(g, 1300 + 16 * b + u, n)
            utts.append((ids, puncts, style, int(rng.integers(900, 1025))))      # > 16 MB of waveforms: four tail groups
        batches.append(utts)
    small = [(*synth.encoder_inputs(g, 1400 + u, 20 + u), 64 + u) for u in range(5)]
    batches.append(small)
    ref = [model.synthesize_batch(b) for b in batches]
    for graph in (False, True):
        model.set_graph_mode(graph)
        try:
            calls = [model.prepare_batch(b) for b in batches]
            order = [0, 1, 2, 3, 0, 2]
            got = {}
            prev = None
            for k, bi in enumerate(order):
                lane = k % 2
                calls[bi].begin(lane)
                if prev is not None:
                    calls[prev[0]].end(prev[1])
                    got[prev[0]] = [(w.copy(), nf) for w, nf in calls[prev[0]].results()]
                prev = (bi, lane)
            calls[prev[0]].end(prev[1])
            got[prev[0]] = [(w.copy(), nf) for w, nf in calls[prev[0]].results()]
        finally:
            model.set_graph_mode(False)
        for bi, res in got.items():
            for (w, nf), (rw, rnf) in zip(res, ref[bi]):
                assert nf == rnf and np.array_equal(w, rw), (graph, bi)
    c = model.prepare_batch(small)
    zero_mel = np.zeros((8, g.num_mels), np.float32)
    ref_v = model.vocode(zero_mel)
    with pytest.raises(capi.ZvError):
        c.end(1)                                   # nothing in flight on that lane
    c.begin(1)
    with pytest.raises(capi.ZvError):
        c.begin(1)                                 # the lane is taken
    # the synchronous entry points run on lane 0 whatever lane was touched last: lane 1 busy does not stop them ...
    assert np.array_equal(model.vocode(zero_mel), ref_v)
    c.end(1)
    assert np.array_equal(c.results()[0][0], ref[3][0][0])
    # ... and lane 0 busy does, also when the lane touched last (1) is idle again: begin(0), begin(1), end(1), then a
    # synchronous call of every kind must be refused (it would overwrite lane 0's I/O block under its downloads)
    c0, c1 = model.prepare_batch(batches[0]), model.prepare_batch(small)
    c0.begin(0)
    c1.begin(1)
    c1.end(1)
    ids, puncts, style = synth.encoder_inputs(g, 5, 12)
    d = model.device_alloc(8 * g.num_mels * 4 + 8 * g.hop_size * 4)
    for call in (lambda: model.vocode(zero_mel), lambda: model.synthesize(ids, puncts, style, 64),
                 lambda: model.encode(ids, puncts, style, 64), lambda: model.decode(np.zeros((8, g.E), np.float32), style),
                 lambda: model.vocode_stream(zero_mel, 4), lambda: model.vocode_device(d, 8, d + 8 * g.num_mels * 4),
                 lambda: model.synthesize_batch(small)):
        with pytest.raises(capi.ZvError):
            call()
    c0.end(0)
    for (w, nf), (rw, rnf) in zip(c0.results(), ref[0]):
        assert nf == rnf and np.array_equal(w, rw)           # the refused calls left the batch in flight untouched
    model.device_free(d)
    assert np.array_equal(model.vocode(zero_mel), ref_v)
    with pytest.raises(capi.ZvError):
        c.begin(capi.BATCH_LANES)                  # no such lane


def test_two_rank_sharded_bench_equals_single_process_batch(tmp_path):
    """the N > 1 path of bench.py (one process per rank, contiguous shards of the global utterance list, no data-path
    collective) rehearsed on ONE GPU: two ranks via torch.distributed.run + gloo, both on cuda:0 (ZV_BENCH_ONE_GPU).
    (a) the union of the ranks' waveforms is bit for bit the single-process batch over the same 16 utterances,
    (b) the reported whole-job rate is sum(audio) / max(wall)."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZV_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29581", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--utts-per-gpu", "8", "--frames", "256", "--no-extras", "--no-cpu-baseline", "--dump-dir", str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["utterances_total"] == 16
    assert "configs[3]" in j["config"]["workload"] and j["unit"] == "x_realtime"
    # (b) value = steps * audio of ALL ranks / max wall
    assert abs(j["value"] - j["config"]["audio_seconds_per_step"] / (j["ms_per_step"] * 1e-3)) <= 0.02 * j["value"]
    assert abs(j["config"]["audio_seconds_per_step"] - 16 * 256 * 300 / 22050) < 1e-2
    # (a) union of the shards == one process, one batch
    from zerovox_cpp_amd import capi, sharding, synth
    g = synth.MEDIUM
    lens = sharding.mixed_length_batch(3, 16)
    utts = []
    for u in range(16):
        ids, puncts, style = synth.encoder_inputs(g, 200 + u, lens[u])
        utts.append((ids, puncts, style, 256))
    ckpt_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "zerovox_medium_seed1234.gguf")
    assert os.path.exists(ckpt_path)                         # written by rank 0 of the run above
    m = capi.Model(ckpt_path, 0)
    ref = m.synthesize_batch(utts)
    m.close()
    seen = set()
    for rk in range(2):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rk))
        lo, hi = sharding.shard_utterances(16, 2, rk)
        assert list(z["index"]) == list(range(lo, hi))
        for k, u in enumerate(z["index"]):
            assert int(z["n_frames"][k]) == ref[u][1] and np.array_equal(z["wav%d" % u], ref[u][0]), u
            seen.add(int(u))
    assert seen == set(range(16))
