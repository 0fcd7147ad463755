"""full-chain timing (BASELINE.json configs[2]: 128 phonemes -> T=512) with per-kernel-family breakdown"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from zerovox_cpp_amd import capi, gguf, synth
g = synth.MEDIUM
ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), "zerovox_medium_seed1234.gguf")
if not os.path.exists(ckpt):
    synth.write_checkpoint(ckpt, g, 1234)
m = capi.Model(ckpt, 0)
N, T = 128, 512
ids, puncts, style = synth.encoder_inputs(g, 5, N)
m.reserve(N, T)
for _ in range(3):
    wav, nf = m.synthesize(ids, puncts, style, T)
t0 = time.perf_counter(); R = 20
for _ in range(R):
    wav, nf = m.synthesize(ids, puncts, style, T)
dt = (time.perf_counter() - t0) / R
print(f"synthesize N={N} T={T}: {dt*1e3:.3f} ms/utt (host buffers, eager) -> {T*300/22050/dt:.0f} xRT, frames {nf}")
m.set_graph_mode(True)
for _ in range(3):
    wavg, nfg = m.synthesize(ids, puncts, style, T)
t0 = time.perf_counter()
for _ in range(R):
    wavg, nfg = m.synthesize(ids, puncts, style, T)
dtg = (time.perf_counter() - t0) / R
m.set_graph_mode(False)
print(f"synthesize N={N} T={T}: {dtg*1e3:.3f} ms/utt (host buffers, hipGraph replay) -> {T*300/22050/dtg:.0f} xRT, same bits as eager: {bool((wavg == wav).all() and nfg == nf)}")
m.profile_begin()
for _ in range(5):
    m.synthesize(ids, puncts, style, T)
st = m.profile_end()
tot = sum(s["total_ms"] for s in st)
for s in st:
    print(f"  {s['name']:22s} launches/utt {s['launches']//5:3d}  ms/utt {s['total_ms']/5:.4f}  share {s['total_ms']/tot:.3f}  TF {s['algo_flops']/(s['total_ms']*1e-3)/1e12 if s['total_ms'] else 0:.1f}")
print("  total kernel ms/utt (event-timed)", tot / 5)

# ---- BASELINE.json configs[3]: batch = 32 mixed-length utterances (32..256 phonemes), T = 1024 each
from zerovox_cpp_amd import sharding
lens = sharding.mixed_length_batch(3, 32)
utts = []
for u, n in enumerate(lens):
    i, p, s = synth.encoder_inputs(g, 200 + u, n)
    utts.append((i, p, s, 1024))
m.reserve(256, 1024)
m.synthesize_batch(utts[:8])
t0 = time.perf_counter()
out = m.synthesize_batch(utts)
dt = time.perf_counter() - t0
audio = sum(1024 * 300 / 22050 for _ in utts)
print(f"batch 32 x T=1024 (4 lanes): {dt*1e3:.1f} ms -> {audio/dt:.0f} xRT whole-batch audio-seconds/second (all T frames vocoded, like the reference)")
t0 = time.perf_counter()
for (i, p, s, T) in utts:
    m.synthesize(i, p, s, T)
dt1 = time.perf_counter() - t0
print(f"same 32 utterances one by one: {dt1*1e3:.1f} ms -> {audio/dt1:.0f} xRT")
