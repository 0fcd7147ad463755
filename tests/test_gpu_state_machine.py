"""-m gpu: history independence.  A long random sequence of calls (all entry points, lengths that grow and shrink the
arenas, graph mode toggled, batches on the in-flight lanes, chunked vocoding) on ONE model must give, for every call,
exactly the bits a second model gives that sees each distinct call once, eagerly, in a different order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _key(op, *a):
    return (op,) + tuple(a)


def test_random_call_sequences_are_history_independent(ckpt):
    from zerovox_cpp_amd import capi, synth
    path, g, tensors = ckpt("small")
    a, b = capi.Model(path, 0), capi.Model(path, 0)
    import os
    rng = np.random.default_rng(int(os.environ.get("ZV_SM_SEED", "2024")))
    Ts = [1, 7, 33, 64, 100, 160, 250]
    Ns = [1, 5, 24, 40, 77]

    def run(m, key):
        op = key[0]
        if op == "vocode":
            return [m.vocode(synth.vocoder_mel(g, tensors, key[2], key[1]))]
        if op == "stream":
            ch = m.vocode_stream(synth.vocoder_mel(g, tensors, key[2], key[1]), key[3])
            return [np.concatenate([c[1] for c in ch])]
        if op == "decode":
            _, _, style = synth.encoder_inputs(g, key[2], 4)
            return [m.decode(synth.decoder_hidden(g, key[2], key[1], frames_per_phoneme=2, fill=0.9), style)]
        if op == "encode":
            ids, puncts, style = synth.encoder_inputs(g, key[3], key[1])
            e = m.encode(ids, puncts, style, key[2])
            return [e["hidden"], e["logdur"], np.array([e["n_frames"]])]
        if op == "synth":
            ids, puncts, style = synth.encoder_inputs(g, key[3], key[1])
            w, nf = m.synthesize(ids, puncts, style, key[2])
            return [w, np.array([nf])]
        raise AssertionError(op)

    seen = {}
    order = []
    for it in range(140):
        r = rng.integers(0, 100)
        T, N, seed = int(rng.choice(Ts)), int(rng.choice(Ns)), int(rng.integers(0, 3))
        if r < 8:
            a.set_graph_mode(bool(rng.integers(0, 2)))
            continue
        if r < 30:
            key = _key("vocode", T, seed)
        elif r < 40:
            key = _key("stream", T, seed, int(rng.choice([16, 50, 64])))
        elif r < 55:
            key = _key("decode", T, seed)
        elif r < 70:
            key = _key("encode", N, T, seed)
        elif r < 85:
            key = _key("synth", N, T, seed)
        else:
            # a batch of synth calls on the lanes: each element is checked like a stand-alone synth call
            keys = [_key("synth", int(rng.choice(Ns)), int(rng.choice(Ts)), int(rng.integers(0, 3))) for _ in range(int(rng.integers(1, 7)))]
            utts = []
            for k in keys:
                ids, puncts, style = synth.encoder_inputs(g, k[3], k[1])
                utts.append((ids, puncts, style, k[2]))
            for k, (w, nf) in zip(keys, a.synthesize_batch(utts)):
                seen.setdefault(k, []).append([w, np.array([nf])])
                order.append(k)
            continue
        seen.setdefault(key, []).append(run(a, key))
        order.append(key)
    a.set_graph_mode(False)
    # model b: every distinct call once, eagerly, sorted order (a different history)
    for key in sorted(seen, key=repr):
        ref = run(b, key)
        for got in seen[key]:
            assert len(got) == len(ref)
            for x, y in zip(got, ref):
                assert np.array_equal(np.asarray(x), np.asarray(y)), key
    print(f"{len(order)} calls, {len(seen)} distinct")
    a.close()
    b.close()
