"""diagnostic: phase timeline of the generic conv kernel's workgroups on the batch's wide decoder convs (variant build -DZV_STAMPS)
usage: ZV_STAMP_CONV=5 ZEROVOX_AMD_LIB=variants/libzv_stamps.so python scripts/stamps_conv.py"""
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
s = s[s[:, 11] > 0].astype(np.int64)
print("workgroups stamped:", len(s))
t = (s - s[:, 0:1]) * 10.0
names = ["stage0+barrier", "mfma0", "stage1+barrier", "mfma1", "stage2+barrier", "mfma2", "stage3+barrier", "mfma3", "stage4+barrier", "mfma4", "epilogue+drain"]
nch = int(((s[:, 1:11] > 0).sum(axis=1) // 2).max())          # chunks this conv has: unused slots repeat the last stamp
for k in range(2 * nch + 1, 11):
    s[:, k] = s[:, 2 * nch]
t = (s - s[:, 0:1]) * 10.0
d = np.diff(t, axis=1)
for i, n in enumerate(names):
    print(f"  {n:16s} mean {d[:, i].mean():8.0f} ns  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
print("  total            mean %8.0f ns" % t[:, 11].mean())
print("  start spread of the launch: first %.1f us, last %.1f us" % (0.0, (s[:, 0].max() - s[:, 0].min()) / 100.0))
