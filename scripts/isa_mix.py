"""Static instruction mix per kernel of a hipcc -S listing (diagnostic): python scripts/isa_mix.py conv.s [name-substring]
   -> per kernel: instructions by class (mfma / valu / salu / lds / vmem / waitcnt+barrier / branch), and per basic block with MFMAs."""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = len(sys.argv) > 3
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop") or op.startswith("s_setprio") or op.startswith("s_sleep"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"
name = None; cnt = None; bb = None; bbs = None
def flush():
    if name and pat in name and cnt:
        import subprocess
        print(name[:150]); print("   ", dict(cnt))
        if blocks:
            for lbl, c in bbs:
                if c.get("mfma", 0) >= 4 or c.get("valu", 0) >= 16: print("      %-14s %s" % (lbl, dict(c)))
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        flush(); name = m.group(1); cnt = collections.Counter(); bbs = []; bb = collections.Counter(); bbs.append(("entry", bb)); continue
    if name is None: continue
    m = re.match(r"^(\.LBB\w+):", ln)
    if m: bb = collections.Counter(); bbs.append((m.group(1), bb)); continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", ln + " ")
    if m and not ln.strip().startswith("."):
        c = cls(m.group(1)); cnt[c] += 1; bb[c] += 1
    if ln.startswith(".Lfunc_end"): flush(); name = None
