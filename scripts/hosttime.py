"""host-side cost of the two halves of a pipelined batch step (zv_synthesize_batch_begin / _end), BASELINE configs[3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from zerovox_cpp_amd import capi, sharding, synth
g = synth.MEDIUM
ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), "zerovox_medium_seed1234.gguf")
if not os.path.exists(ckpt):
    synth.write_checkpoint(ckpt, g, 1234)
m = capi.Model(ckpt, 0)
lens = sharding.mixed_length_batch(3, 32)
utts = [synth.encoder_inputs(g, 200 + u, n) + (1024,) for u, n in enumerate(lens)]
m.set_graph_mode(True)
lanes = [m.prepare_batch(utts) for _ in range(2)]
for k in range(4):
    lanes[k % 2].begin(k % 2)
    if k >= 1: lanes[(k - 1) % 2].end((k - 1) % 2)
lanes[1].end(1)
m.synchronize()
tb, te = [], []
K = 16
t0 = time.perf_counter()
for k in range(K):
    a = time.perf_counter(); lanes[k % 2].begin(k % 2); b = time.perf_counter(); tb.append(b - a)
    if k >= 1:
        a = time.perf_counter(); lanes[(k - 1) % 2].end((k - 1) % 2); b = time.perf_counter(); te.append(b - a)
a = time.perf_counter(); lanes[(K - 1) % 2].end((K - 1) % 2); te.append(time.perf_counter() - a)
dt = time.perf_counter() - t0
print(f"step {dt / K * 1e3:.2f} ms; begin() mean {np.mean(tb) * 1e3:.2f} ms (max {np.max(tb) * 1e3:.2f}); end() mean {np.mean(te) * 1e3:.2f} ms (min {np.min(te) * 1e3:.2f})")
# how long does end() take when everything has arrived already?
lanes[0].begin(0); m.synchronize(); time.sleep(0.05)
a = time.perf_counter(); lanes[0].end(0); print(f"end() of a finished batch: {(time.perf_counter() - a) * 1e3:.2f} ms")
