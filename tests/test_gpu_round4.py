"""-m gpu: round 4 — the serving loop with three batches in flight and its timeline, the lane rule of the copy / wait companions of the device-resident entry points, switch save / restore, graphs re-captured
after a switch change."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

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


def _batches(g, synth, nb=4, nu=12, seed=1700):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(nb):
        utts = []
        for u in range(nu):
            n = int(rng.integers(24, 96))
            ids, puncts, style = synth.encoder_inputs(g, seed + 16 * b + u, n)
            utts.append((ids, puncts, style, int(rng.integers(700, 1025))))
        out.append(utts)
    return out


@pytest.mark.parametrize("graph", [False, True])
def test_three_batches_in_flight_equal_the_synchronous_call(models, graph):
    """begin(k) / end(k - 2) over three lanes: every batch's results are those of zv_synthesize_batch bit for bit, and
    zv_batch_timeline reports one (start, last kernel) pair per batch, in order, batches of one lane never overlapping"""
    from zerovox_cpp_amd import capi, synth
    model, g, _ = models("medium")
    batches = _batches(g, synth)
    ref = [model.synthesize_batch(b) for b in batches]
    model.set_graph_mode(graph)
    try:
        calls = [model.prepare_batch(b) for b in batches] + [model.prepare_batch(b) for b in batches[:2]]
        which = [0, 1, 2, 3, 0, 1]
        depth = 3
        got = {}
        for k in range(len(calls)):
            calls[k].begin(k % depth)
            if k >= depth - 1:
                j = k - (depth - 1)
                calls[j].end(j % depth)
                got[j] = [(w.copy(), nf) for w, nf in calls[j].results()]
        for j in range(len(calls) - (depth - 1), len(calls)):
            calls[j].end(j % depth)
            got[j] = [(w.copy(), nf) for w, nf in calls[j].results()]
    finally:
        model.set_graph_mode(False)
    for j, res in got.items():
        for (w, nf), (rw, rnf) in zip(res, ref[which[j]]):
            assert nf == rnf and np.array_equal(w, rw), (graph, j)
    tl = model.batch_timeline(len(calls))
    assert len(tl) == len(calls)
    for i, (s, e) in enumerate(tl):
        assert e > s >= 0.0
        if i >= depth:
            assert s >= tl[i - depth][1] - 1e-3, (i, tl)    # same lane, same stream: batch i starts after batch i - depth
    print("timeline (ms):", [(round(s, 2), round(e, 2)) for s, e in tl])


def test_copies_and_waits_follow_lane0_whatever_lane_was_touched_last(models):
    """zv_vocode_device runs on lane 0; a batch begun on lane 1 afterwards makes lane 1 the lane touched last — the copy that
    fetches the vocoder's result (and zv_synchronize) must still be ordered behind lane 0's kernels"""
    from zerovox_cpp_amd import synth
    model, g, tensors = models("medium")
    T = 512
    mel = synth.vocoder_mel(g, tensors, 91, T)
    ref = model.vocode(mel)
    hop = g.hop_size
    d_mel = model.device_alloc(mel.nbytes)
    d_wav = model.device_alloc(T * hop * 4)
    small = [(*synth.encoder_inputs(g, 1500 + u, 20 + u), 64 + u) for u in range(5)]
    c = model.prepare_batch(small)
    ref_b = model.synthesize_batch(small)
    try:
        for trial in range(4):
            model.h2d(d_wav, np.zeros(T * hop, np.float32))
            model.h2d(d_mel, mel)
            model.vocode_device(d_mel, T, d_wav)      # enqueued on lane 0, not waited for
            c.begin(1)                                # lane 1 is now the lane touched last
            got = np.empty(T * hop, np.float32)
            if trial % 2:
                model.synchronize()                   # every lane
            model.d2h(got, d_wav)
            assert np.array_equal(got, ref), trial
            c.end(1)
            for (w, nf), (rw, rnf) in zip(c.results(), ref_b):
                assert nf == rnf and np.array_equal(w, rw)
    finally:
        model.device_free(d_mel)
        model.device_free(d_wav)


def test_switches_restore_previous_values_and_nest():
    from zerovox_cpp_amd import capi
    capi.debug_set("ZV_PAIR_MT", 4)
    try:
        with capi.switches(ZV_PAIR_MT=2, ZV_NO_FUSE=1):
            assert capi.debug_get("ZV_PAIR_MT") == 2 and capi.debug_get("ZV_NO_FUSE") == 1
            with capi.switches(ZV_NO_FUSE=0):
                assert capi.debug_get("ZV_NO_FUSE") == 0
            assert capi.debug_get("ZV_NO_FUSE") == 1
        assert capi.debug_get("ZV_PAIR_MT") == 4 and capi.debug_get("ZV_NO_FUSE") == 0
    finally:
        capi.debug_set("ZV_PAIR_MT", 0)
    with pytest.raises(capi.ZvError):
        capi.debug_get("ZV_NO_SUCH_SWITCH")
    # timing-only ablation switches (wrong results) do not exist in the shipped library
    with pytest.raises(capi.ZvError):
        capi.debug_set("ZV_DBG", 2)


def test_a_switch_change_recaptures_graphs(models):
    """a captured graph replays the kernel regime it was captured in: after zv_debug_set the same call captures anew, so an
    in-process regime comparison in graph mode compares two regimes — both must still give the same bits"""
    from zerovox_cpp_amd import capi, synth
    model, g, tensors = models("medium")
    mel = synth.vocoder_mel(g, tensors, 92, 384)
    ref = model.vocode(mel)
    d_mel = model.device_alloc(mel.nbytes)
    d_wav = model.device_alloc(384 * g.hop_size * 4)
    model.h2d(d_mel, mel)
    model.set_graph_mode(True)
    try:
        outs = []
        for sw in ({}, {"ZV_CONV_SINGLE": 0}, {"ZV_PAIR_MT": 4}, {}):
            with capi.switches(**sw):
                for _ in range(2):                    # capture, replay
                    model.vocode_device(d_mel, 384, d_wav)
                got = np.empty(384 * g.hop_size, np.float32)
                model.d2h(got, d_wav)
                outs.append(got)
        for o in outs:
            assert np.array_equal(o, ref)
    finally:
        model.set_graph_mode(False)
        model.device_free(d_mel)
        model.device_free(d_wav)


def test_fused_tails_and_one_launch_regulator_give_the_same_bits(ckpt):
    """round 4's launch savers for short utterances — LayerNorm launches that also do the style add / the predictor's linear
    layer / the bucket + embedding step, scan + gather of the length regulator in one launch — change which launch does an
    operation, never the operation: the bits of the plain schedule (ZV_LN_TAIL=0), eager and as a graph, for one utterance and
    for a batch, every tap of the encoder included"""
    from zerovox_cpp_amd import capi, synth
    path, g, _ = ckpt("medium")
    ids, puncts, style = synth.encoder_inputs(g, 61, 96)
    T = 384
    utts = [(*synth.encoder_inputs(g, 1600 + u, 30 + 9 * u), 200 + 16 * u) for u in range(6)]
    ref = None
    for name, sw in (("plain", dict(ZV_LN_TAIL=0)), ("default", {})):
        with capi.switches(**sw):
            m = capi.Model(path, 0)
            outs = []
            for graph in (False, True):
                m.set_graph_mode(graph)
                for rep in range(3 if graph else 1):
                    w, nf = m.synthesize(ids, puncts, style, T)
                    e = m.encode(ids, puncts, style, T)
                    b = m.synthesize_batch(utts)
                    outs.append((w, nf, e["hidden"], e["logdur"], e["pitch"], e["energy"], e["pitch_bucket"], e["energy_bucket"],
                                 e["features"], [x[0] for x in b], [x[1] for x in b]))
            m.close()
        if ref is None:
            ref = outs[0]
        for o in outs:
            assert o[1] == ref[1] and o[10] == ref[10], name
            for a_, b_ in zip(o[:1] + o[2:9], ref[:1] + ref[2:9]):
                assert np.array_equal(a_, b_), name
            for a_, b_ in zip(o[9], ref[9]):
                assert np.array_equal(a_, b_), name


def test_single_utterance_conv_forms_give_the_same_bits(ckpt):
    """one utterance end to end under every form its generic convs can take — the loader-wave form (four waves stage chunk c + 1 into
    the second LDS tile while four multiply chunk c) with its channel groups dealt over the XCDs or on a plain grid, for multi-chunk
    convs only (default) or every conv, or never (the batch's form: every wave stages, then every wave multiplies): one accumulation
    chain per output element whoever stages the tile, so the waveform, the frame count and every encoder tap are bit-equal"""
    from zerovox_cpp_amd import capi, synth
    path, g, _ = ckpt("medium")
    ids, puncts, style = synth.encoder_inputs(g, 77, 128)
    T = 512
    ref = None
    for name, sw in (("default", {}), ("plain_grid", dict(ZV_CONV_XCD=0)), ("every_conv", dict(ZV_CONV_SINGLE=2)),
                     ("cold_l2", dict(ZV_CONV_WARM=0)), ("cold_l2_plain_grid", dict(ZV_CONV_WARM=0, ZV_CONV_XCD=0)),
                     ("every_conv_plain_grid", dict(ZV_CONV_SINGLE=2, ZV_CONV_XCD=0)), ("never", dict(ZV_CONV_SINGLE=0))):
        with capi.switches(**sw):
            m = capi.Model(path, 0)
            outs = []
            for graph in (False, True):
                m.set_graph_mode(graph)
                w, nf = m.synthesize(ids, puncts, style, T)
                e = m.encode(ids, puncts, style, T)
                outs.append((w, nf, e["hidden"], e["logdur"], e["pitch"], e["energy"]))
            m.close()
        if ref is None:
            ref = outs[0]
        for o in outs:
            assert o[1] == ref[1], name
            for a_, b_ in zip(o[:1] + o[2:], ref[:1] + ref[2:]):
                assert np.array_equal(a_, b_), name


def test_single_utterance_forms_on_ragged_sizes(ckpt):
    """the loader-wave conv form against the batch's form on utterances whose sizes hit the tile edges (1 .. 300 phonemes, 1 .. 1 500 frames:
    one row tile or many, chunks of 256 / 32 / 16 channels, the operand pre-pass on either side of its 256-frame threshold), eager and as a graph"""
    from zerovox_cpp_amd import capi, synth
    path, g, _ = ckpt("medium")
    rng = np.random.default_rng(5)
    a = capi.Model(path, 0)
    try:
        with capi.switches(ZV_CONV_SINGLE=0):
            b = capi.Model(path, 0)
            cases = []
            for it, T in enumerate([1, 7, 31, 33, 64, 100, 255, 256, 257, 400, 777, 1500]):
                n = int(rng.integers(1, 300))
                ids, puncts, style = synth.encoder_inputs(g, 2100 + it, n)
                cases.append((ids, puncts, style, T) + b.synthesize(ids, puncts, style, T))
            b.close()
        for graph in (False, True):
            a.set_graph_mode(graph)
            for ids, puncts, style, T, wb, nfb in cases:
                w, nf = a.synthesize(ids, puncts, style, T)
                assert nf == nfb and np.array_equal(w, wb), (len(ids), T, graph)
    finally:
        a.close()
