"""diagnostic: phase timeline of the fused pair kernel's workgroups (variant build with -DZV_STAMPS)
usage: ZV_STAMP_CP=64 ZEROVOX_AMD_LIB=variants/libzv_stamps.so python scripts/stamps.py"""
import ctypes as C, os, sys
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
call = m.prepare_batch(utts)
call.run(); call.run()
m.synchronize()
NW, NS = 1 << 17, 12
buf = np.zeros(NW * NS, np.uint64)
lib = C.CDLL(capi.LIB_PATH)
lib.zv_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.zv_debug_read_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(NW, NS)
live = s[:, 7] > 0
s = s[live].astype(np.int64)
print("workgroups stamped:", len(s))
t = s[:, :8] - s[:, 0:1]
names = ["start", "staging issued+cvt", "barrier1", "conv1 done", "barrier2", "xt packed+barrier3", "conv2 done", "stores drained"]
d = np.diff(t, axis=1) * 10.0      # memrealtime: 100 MHz -> ns
for i, n in enumerate(names[1:]):
    print(f"  {n:22s} mean {d[:, i].mean():9.0f} ns  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
print("  total                  mean %9.0f ns" % (t[:, 7].mean() * 10))
# co-residency: per (xcc, cu) how many workgroups overlap in time and in which phases
hw, xcc = s[:, 8], s[:, 9] & 0xF
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = xcc * 4096 + se * 256 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs seen:", len(u), "workgroups per CU: mean", cnt.mean())
k0 = u[0]
sel = s[key == k0]
sel = sel[np.argsort(sel[:, 0])][:16]
base = sel[0, 0]
for r in sel:
    print("   ", " ".join("%7.1f" % ((x - base) / 100.0) for x in r[:8]), "us")
