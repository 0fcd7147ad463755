"""stress of the single-utterance path: random (phonemes, frames) pairs, the default conv forms against the batch's form (ZV_CONV_SINGLE=0),
bit for bit; every call also repeated under graph replay.  usage (GPU box): timeout -k 10 600 python scripts/single_stress.py [iterations] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from zerovox_cpp_amd import capi, synth
g = synth.MEDIUM
ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), "zerovox_medium_seed1234.gguf")
if not os.path.exists(ckpt):
    synth.write_checkpoint(ckpt, g, 1234)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
a = capi.Model(ckpt, 0)
with capi.switches(ZV_CONV_SINGLE=0):
    b = capi.Model(ckpt, 0)
    cases = []
    for it in range(iters):
        n = int(rng.integers(1, 300))
        T = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 400, 512, 777, 1024, 1500]))
        ids, puncts, style = synth.encoder_inputs(g, 900 + it, n)
        wb, nfb = b.synthesize(ids, puncts, style, T)
        cases.append((n, T, ids, puncts, style, wb, nfb))
for graph in (False, True):
    a.set_graph_mode(graph)
    for n, T, ids, puncts, style, wb, nfb in cases:
        w, nf = a.synthesize(ids, puncts, style, T)
        assert nf == nfb and np.isfinite(w).all() and np.array_equal(w, wb), (n, T, graph)
print(f"{iters} utterances x (eager, graph): bit-equal to the ZV_CONV_SINGLE=0 model")
