"""diagnostic: where conv_gemm_kernel's wave 0 spends its K loop (variant build with -DZV_STAMPS): counted waits / barriers / rest
usage: ZEROVOX_AMD_LIB=zerovox.cpp_amd/_ab/libzv_stamps.so python scripts/stamps_gemm.py"""
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
T = 32 * 1024
hid = synth.decoder_hidden(g, 11, 1024)
_, _, style = synth.encoder_inputs(g, 5, 64)
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
s = buf.reshape(NW, NS)[:4096]
s = s[s[:, 7] == 1].astype(np.float64)
print("workgroups stamped (the last conv_gemm launch of the pass):", len(s), "halves", np.unique(s[:, 3]))
for ex in (0, 1):
    z = s[s[:, 4] == ex]
    if not len(z): continue
    loop, wait, bar, nh = z[:, 0].mean(), z[:, 1].mean(), z[:, 2].mean(), z[:, 3].mean()
    print(f"ninth tile {ex}: loop {loop:.0f} cycles = {loop / nh:.0f} per half unit; counted waits {wait / loop:.1%}, barriers {bar / loop:.1%}, "
          f"reads + MFMAs + DMA issue {(loop - wait - bar) / loop:.1%} = {(loop - wait - bar) / nh:.0f} cycles per half (16-18 MFMAs = 512-576 cycles of the matrix pipe per wave, two waves per SIMD)")
