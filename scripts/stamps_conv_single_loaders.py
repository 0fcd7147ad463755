"""diagnostic: timeline of the LOADER waves of the single-utterance conv form (variant build -DZV_STAMPS -DZV_DIAG -DZV_STAMPS_LOADER)
usage: ZV_STAMP_CONV=9 ZEROVOX_AMD_LIB=zerovox.cpp_amd/_ab/libzv_stamps.so python scripts/stamps_conv_single_loaders.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from zerovox_cpp_amd import capi, synth
g = synth.MEDIUM
ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), "zerovox_medium_seed1234.gguf")
if not os.path.exists(ckpt): synth.write_checkpoint(ckpt, g, 1234)
m = capi.Model(ckpt, 0)
ids, puncts, style = synth.encoder_inputs(g, 5, 128)
for _ in range(5): m.synthesize(ids, puncts, style, 512)
m.synchronize()
NW, NS = 1 << 17, 12
buf = np.zeros(NW * NS, np.uint64)
lib = C.CDLL(capi.LIB_PATH)
lib.zv_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.zv_debug_read_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(NW, NS); s = s[s[:, 11] > 0].astype(np.int64)
print("workgroups:", len(s))
t = (s - s[:, 0:1]) * 10.0
names = ["ld t0,t1 + st t0 + ld t2", "barrier 0", "store t1", "load t3", "barrier 1", "store t2", "load t4", "barrier 2", "store t3", "load -", "barrier 3"]
d = np.diff(t, axis=1)
for i, n in enumerate(names): print(f"  {n:26s} mean {d[:, i].mean():8.0f} ns  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
