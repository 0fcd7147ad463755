"""vocoder-only time vs T (device buffers, graph replay): how much do longer launches amortise the fixed latencies?"""
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
for T in [int(x) for x in (sys.argv[1:] or ["512", "1024", "2048", "4096", "8192", "16384"])]:
    mel = np.random.default_rng(1).standard_normal((T, 80)).astype(np.float32)
    d_mel = m.device_alloc(mel.nbytes); d_wav = m.device_alloc(T * 300 * 4)
    m.h2d(d_mel, mel)
    m.reserve(1, T)
    m.set_graph_mode(True)
    for _ in range(3): m.vocode_device(d_mel, T, d_wav)
    m.synchronize()
    R = 20
    t0 = time.perf_counter()
    for _ in range(R): m.vocode_device(d_mel, T, d_wav)
    m.synchronize()
    dt = (time.perf_counter() - t0) / R
    m.set_graph_mode(False)
    print(f"T={T}: {dt*1e3:.3f} ms  {dt/T*1e6:.3f} us/frame  {T*300/22050/dt:.0f} xRT")
    m.device_free(d_mel); m.device_free(d_wav)
