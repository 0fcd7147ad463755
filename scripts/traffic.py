#!/usr/bin/env python3
"""HBM-side bytes per launch of the HiFi-GAN ResBlock kernel family from the PMC passes (FETCH_SIZE doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads, + WRITE_SIZE), weighted over the
family's launches, with the launch durations of the kernel trace of the same build.
usage: traffic.py <pmc.txt> <kernel_trace_summary.txt> <tag>"""
import hashlib, json, os, re, sys
pmc, trace, tag = sys.argv[1:4]
fam = ("resblock_pair_kernel", "resblock_pair64_kernel", "resblock_triple_kernel", "resblock_block32_kernel", "resblock_tail_kernel",
       "resblock_block64_kernel")
SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zerovox.cpp_amd", "csrc", "conv1d_mfma.hip")


def stage_of(key):
    """vocoder stage (s0 .. s3 = 256 / 128 / 64 / 32 channels) of a kernel configuration"""
    if key.startswith("resblock_pair_kernel<256"): return "s0"
    if key.startswith("resblock_pair_kernel<128"): return "s1"
    if key.startswith(("resblock_pair64_kernel", "resblock_pair_kernel<64", "resblock_block64_kernel")): return "s2"
    return "s3"
def norm(ln):
    """kernel names as the profiler prints them: templates as `void zv::name<...>`, plain kernels as `zv::name` — one form here"""
    return re.sub(r"^(void )?(zv::)?", "", ln)


fetch, write, n = {}, {}, {}
sect = None
for ln in open(pmc):
    if ln.startswith("#"):
        sect = ln
        continue
    ln = norm(ln)
    if not ln.startswith(fam):
        continue
    key = ln.split(" grid ")[0] + " grid " + ln.split(" grid ")[1].split()[0]
    m = re.search(r"FETCH_SIZE=([0-9.e+]+)", ln)
    if m:
        fetch[key] = float(m.group(1)) * 1024 * 2          # KB -> bytes, x2: gfx950 tallies 128-B requests at 64 B
        n[key] = int(re.search(r"n=(\d+)", ln).group(1))
    m = re.search(r"WRITE_SIZE=([0-9.e+]+)", ln)
    if m:
        write[key] = float(m.group(1)) * 1024
dur = {}
for ln in open(trace):
    ln = norm(ln)
    if not ln.startswith(fam):
        continue
    name = ln.split("grid=")[0].strip()
    g = re.search(r"grid=\((\d+),(\d+),(\d+)\)", ln)
    key = "%s grid %d" % (name, int(g.group(1)) * int(g.group(2)) * int(g.group(3)))
    dur[key] = (int(re.search(r"calls=(\d+)", ln).group(1)), float(re.search(r"avg_us=([0-9.]+)", ln).group(1)))
# the evidence chain must close: every family configuration of the PMC passes has a FETCH row, a WRITE row and a duration in the
# kernel trace, and a family kernel of the trace without PMC rows is only the same kernel on another grid (the timed region's
# utterance groups; the PMC passes run whole-batch launches).  Anything else is a parsing hole: stop, do not average around it.
problems = []
for k in sorted(set(fetch) | set(write)):
    if k not in fetch: problems.append("no FETCH_SIZE row for " + k)
    if k not in write: problems.append("no WRITE_SIZE row for " + k)
    if k not in dur: problems.append("no kernel-trace duration for " + k)
pmc_names = {k.split(" grid ")[0] for k in fetch}
for k in dur:
    if k not in fetch and k.split(" grid ")[0] not in pmc_names:
        problems.append("family kernel of the trace has no PMC row on any grid: " + k)
if not fetch: problems.append("no ResBlock family rows found in " + pmc)
if problems:
    sys.exit("traffic.py: " + "; ".join(problems))
tot_b = tot_n = tot_us = 0.0
per = {}
for k in fetch:
    calls, us = dur[k]
    b = fetch[k] + write[k]
    per[k] = {"hbm_bytes_per_launch": b, "avg_launch_us": us, "GBps": round(b / (us * 1e-6) / 1e9, 1) if us else None}
    tot_b += b * n[k]
    tot_n += n[k]
    tot_us += us * n[k]
passes = min(n.values()) if n else 1          # a configuration that runs once per pass exists in every stage's schedule
per_stage = {}
for k in per:
    st = per_stage.setdefault(stage_of(k), {"hbm_bytes_per_pass": 0.0, "launches_per_pass": 0.0, "us_per_pass": 0.0})
    st["hbm_bytes_per_pass"] += per[k]["hbm_bytes_per_launch"] * n[k] / passes
    st["launches_per_pass"] += n[k] / passes
    st["us_per_pass"] += (per[k]["avg_launch_us"] or 0.0) * n[k] / passes
out = {"kernel_rev": tag, "kernel_src_sha256": hashlib.sha256(open(SRC, "rb").read()).hexdigest(),
       "kernel_src": "zerovox.cpp_amd/csrc/conv1d_mfma.hip", "passes": passes, "per_stage": per_stage, "workload": "bench.py batch: 32 utterances x 1024 frames per launch",
       "hbm_bytes_per_launch": tot_b / tot_n if tot_n else None, "avg_launch_us": tot_us / tot_n if tot_n else None,
       "method": "per kernel configuration: 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, separate --pmc passes; "
                 "averaged over the family's dispatches; durations from the kernel trace of the same build",
       "per_kernel": per}
print(json.dumps(out, indent=1))
