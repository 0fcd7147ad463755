"""determinism soak of the batch kernels (LDS-DMA rings, raw LDS stores, tail groups): the configs[3] batch N times, every
waveform of every run must equal the first run's bit for bit; python scripts/soak_batch.py [runs]"""
import os, sys
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
utts = [synth.encoder_inputs(g, 200 + u, n) + (1024 - 37 * (u % 5),) for u, n in enumerate(lens)]
m.set_graph_mode(True)
call = m.prepare_batch(utts)
call.run()
ref = [(w.copy(), nf) for w, nf in call.results()]
bad = 0
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for it in range(runs):
    call.run()
    for k, ((w, nf), (r, nfr)) in enumerate(zip(call.results(), ref)):
        if nf != nfr or not np.array_equal(w, r):
            bad += 1
            print(f"run {it} utterance {k}: differs from the first run", flush=True)
print("mismatches:", bad, "of", runs * len(utts))
sys.exit(1 if bad else 0)
