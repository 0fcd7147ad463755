"""debug: how often does a multi-lane batch differ from stand-alone runs? (env toggles select kernel paths)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from zerovox_cpp_amd import capi, synth, sharding
g = synth.SMALL
path = "/tmp/small_stress.gguf"
synth.write_checkpoint(path, g, 1234)
m = capi.Model(path, 0)
lens = [min(n, 96) for n in sharding.mixed_length_batch(11, 7)]
utts = []
for u, n in enumerate(lens):
    ids, puncts, style = synth.encoder_inputs(g, 100 + u, n)
    utts.append((ids, puncts, style, 64 + 32 * (u % 3)))
refs = [m.synthesize(*u) for u in utts]
bad = 0
for it in range(int(sys.argv[1])):
    got = m.synthesize_batch(utts)
    for k, ((wav, nf), (ref, nfr)) in enumerate(zip(got, refs)):
        if nf != nfr or not np.array_equal(wav, ref):
            bad += 1
            d = np.nonzero(wav != ref)[0]
            print(f"it {it} utt {k} (lane {k % 4}) nf {nf}/{nfr} first diff sample {d[0] if d.size else -1} of {wav.size} ndiff {d.size}", flush=True)
print("mismatches:", bad, "of", int(sys.argv[1]) * len(utts), flush=True)
