"""decoder time vs T with per-family breakdown (host buffers, eager)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from zerovox_cpp_amd import capi, synth
g = synth.MEDIUM
ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), "zerovox_medium_seed1234.gguf")
if not os.path.exists(ckpt):
    synth.write_checkpoint(ckpt, g, 1234)
m = capi.Model(ckpt, 0)
_, _, style = synth.encoder_inputs(g, 5, 8)
for T in [int(x) for x in (sys.argv[1:] or ["256", "448", "512", "576", "1024", "512"])]:
    hid = synth.decoder_hidden(g, 3, T)
    m.reserve(1, T)
    for _ in range(3): m.decode(hid, style)
    R = 10
    t0 = time.perf_counter()
    for _ in range(R): m.decode(hid, style)
    dt = (time.perf_counter() - t0) / R
    m.profile_begin()
    for _ in range(5): m.decode(hid, style)
    st = m.profile_end()
    print(f"decoder T={T}: {dt*1e3:.3f} ms wall;", "  ".join(f"{s['name']} {s['launches']//5}x{1e3*s['total_ms']/s['launches']:.1f}us" for s in st),
          f"| kernel sum {sum(s['total_ms'] for s in st)/5:.3f} ms")
