"""diagnostic: phase timeline of the whole-block kernel's workgroups (variant build -DZV_STAMPS)
usage: ZV_STAMP_CP=32 ZV_TAIL_GROUPS=0 ZEROVOX_AMD_LIB=variants/libzv_stamps.so python scripts/stamps_block32.py"""
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
S = buf.reshape(NW, NS)
names = ["load + X write 0", "conv1 0", "pack 0", "conv2+update+X write 1", "conv1 1", "pack 1", "conv2+update+X write 2", "conv1 2", "pack 2", "-", "conv2 2 + store + drain"]
wg = np.arange(NW)
# the batch launch interleaves the three MRF branches (3, 7, 11 taps) over the workgroup index: job = (blockIdx.x >> 3) % 3
for jb, taps in ((None, "all"), (0, 3), (1, 7), (2, 11)):
    sel = S[:, 11] > 0
    if jb is not None: sel &= ((wg >> 3) % 3) == jb
    s = S[sel].astype(np.int64)
    print(f"branch {taps}: workgroups stamped: {len(s)}")
    if not len(s): continue
    s[:, 10] = s[:, 9]              # slot 10 is not stamped (three dilation pairs)
    t = (s - s[:, 0:1]) * 10.0
    d = np.diff(t, axis=1)
    for i, n in enumerate(names):
        if n == "-": continue
        print(f"  {n:26s} mean {d[:, i].mean():8.0f} ns  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
    print("  total                      mean %8.0f ns" % t[:, 11].mean())
