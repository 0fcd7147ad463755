"""configs[2] (one 128-phoneme utterance, T = 512) in a loop, for kernel traces: python scripts/single.py [reps] [graph] [N] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from zerovox_cpp_amd import capi, synth
g = synth.MEDIUM
ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), "zerovox_medium_seed1234.gguf")
if not os.path.exists(ckpt):
    synth.write_checkpoint(ckpt, g, 1234)
m = capi.Model(ckpt, 0)
N, T = 128, 512
if len(sys.argv) > 4: N, T = int(sys.argv[3]), int(sys.argv[4])
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ids, puncts, style = synth.encoder_inputs(g, 5, N)
m.reserve(N, T)
m.set_graph_mode(len(sys.argv) > 2 and sys.argv[2] == "graph")
for _ in range(3):
    m.synthesize(ids, puncts, style, T)
t0 = time.perf_counter()
for _ in range(R):
    m.synthesize(ids, puncts, style, T)
print(f"{(time.perf_counter() - t0) / R * 1e3:.3f} ms per utterance")
