#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: audio-seconds per wall-second (xRT), end-to-end
phoneme ids -> 22.05 kHz waveform.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3]; with N = 8 it is configs[4]): per GPU a batch of 32 mixed-length utterances
(32..256 phonemes, seeded), T = 1024 frames each, medium geometry (emb 512+16, 4 encoder layers, 512 -> 32 vocoder
channels, hop 300), synthetic seeded weights.  One step = one zv_synthesize_batch call over the rank's utterances:
HOST phoneme / punctuation ids and style vectors in, HOST waveforms out — the upload of the inputs, the three stages
(FastSpeech2 encoder -> StyleTTS decoder -> HiFi-GAN, reference src/zerovox.cpp:326-334; all T frames are vocoded, as
the reference does) and the download of the waveforms are all inside the timed region (SURVEY.md §8d "Wall").  The
whole batch is one launch per kernel (segment tables in HBM) replayed as one hipGraph.

value = steps x (audio seconds of all utterances of all ranks) / wall  — whole-job audio seconds per second.
The global list has 32 x N utterances, rank r owns sharding.shard_utterances(32 N, N, r): weak scaling, no collective
on the data path; torch.distributed only provides the barrier and the reductions of (audio, wall).

Extra objects on the same JSON line:
  roofline     — dominant kernel family (the HiFi-GAN ResBlock Conv1d launches) of the SAME batch workload, timed live
                 with HIP events on the model's stream (zv_profile_begin/_end, eager launches, one event pair per stage's
                 run of ResBlock launches).  Every stage is priced against the roof that binds IT: time at the dense f16
                 MFMA peak for its algorithmic flops vs time at the HBM peak for its real traffic (PMC-derived bytes of
                 the committed profile named in `traffic_source`, used only while the kernel source's hash still matches
                 the one the profile was taken on); `stages[i].frac_binding` = that time / measured time, the family's
                 `frac_binding` the same over the sums.  `bound`, `achieved`, `peak`, `frac` quote the family against the
                 roof that binds most of its time (algorithmic TFLOP/s for "mfma", real GB/s for "hbm"): a number that
                 cannot exceed 1.  The SURVEY.md §8d figure (algorithmic bytes of the UNFUSED 72-conv formulation / time)
                 is kept beside it as `algo_GBps_unfused_equiv`; the fused kernels move far fewer real bytes.
  roofline_configs1 — the same family at BASELINE.json configs[1] (one 512-frame utterance): 1 909.4 MB algorithmic / sum of
                 the ResBlock launch times / 8 TB/s (the north_star wording of the 60 % target).
  cpu_baseline — the compiled reference (oracle/_ref/zvref: ggml CPU backend + the reference's own stage classes) timed
                 end to end (encoder + decoder + vocoder back to back) on ONE utterance of the batch on this host's cores;
                 falls back to our CPU port (oracle/zv_oracle.c) where the reference binary is absent.
  extra        — BASELINE.json configs[1] (vocoder only, 512 frames, mel resident in HBM) and configs[2] (one 128-phoneme
                 utterance end to end), the per-kernel-family table of the batch, PCIe-inclusive notes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_PEAK_TF = 2500.0          # dense f16 MFMA peak (same guide)
UTTS_PER_GPU = 32
FRAMES = 1024
SEED_W, SEED_BATCH = 1234, 3
TRAFFIC_PROFILE = "profiles/r04_resblock_traffic.json"
KERNEL_SRC = "zerovox.cpp_amd/csrc/conv1d_mfma.hip"


def _kernel_src_sha():
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, KERNEL_SRC), "rb").read()).hexdigest()
    except OSError:
        return None


def _load_traffic():
    """PMC-derived HBM bytes per ResBlock stage and pass of the committed profile, or (None, reason) when the kernel source
    has changed since that profile was taken (a stale number must not be quoted)"""
    tpath = os.path.join(ROOT, TRAFFIC_PROFILE)
    if not os.path.exists(tpath):
        return None, "no committed traffic profile (%s)" % TRAFFIC_PROFILE
    try:
        tj = json.load(open(tpath))
    except Exception as e:
        return None, "unreadable %s: %s" % (TRAFFIC_PROFILE, e)
    if tj.get("kernel_src_sha256") != _kernel_src_sha():
        return None, "stale: %s was measured on another revision of %s (sha256 %s...)" % (
            TRAFFIC_PROFILE, KERNEL_SRC, str(tj.get("kernel_src_sha256"))[:12])
    return tj, "static: %s (%s) — PMC passes (2 x FETCH_SIZE + WRITE_SIZE) of a committed profile of this kernel source, not measured in this run" % (
        TRAFFIC_PROFILE, tj.get("kernel_rev", "?"))


def resblock_roofline(stats, psteps, timing_note):
    """the ResBlock family priced stage by stage against the roof that binds each stage (see the module docstring)"""
    tj, tsrc = _load_traffic()
    stages, fam = [], dict(ms=0.0, flops=0.0, abytes=0.0, traffic=0.0, t_roof=0.0, launches=0, t_mfma=0.0, t_hbm=0.0)
    have_traffic = tj is not None
    for st in sorted((x for x in stats if x["name"].startswith("voc_resblock_s")), key=lambda x: x["name"]):
        key = st["name"][len("voc_resblock_"):]
        ms = st["total_ms"] / psteps
        flops, abytes = st["algo_flops"] / psteps, st["algo_bytes"] / psteps
        traffic = None
        if have_traffic and key in tj.get("per_stage", {}):
            # the profile's bytes belong to ITS launches: a stage whose launch count differs from what was just timed is a
            # different schedule (or a hole in the profile's parsing) — not quoted
            if int(round(tj["per_stage"][key].get("launches_per_pass", -1))) == st["launches"] // psteps:
                traffic = tj["per_stage"][key]["hbm_bytes_per_pass"]
            else:
                tsrc = "refused: %s lists %s launches per pass for stage %s, this run timed %d" % (
                    TRAFFIC_PROFILE, tj["per_stage"][key].get("launches_per_pass"), key, st["launches"] // psteps)
        t_mfma = flops / (MFMA_PEAK_TF * 1e12) * 1e3
        t_hbm = traffic / (HBM_PEAK_GBS * 1e9) * 1e3 if traffic else None
        bound = "hbm" if (t_hbm is not None and t_hbm > t_mfma) else "mfma"
        t_roof = max(t_mfma, t_hbm or 0.0)
        stages.append({"stage": key, "launches": st["launches"] // psteps, "ms": round(ms, 4),
                       "algo_TFLOPs": round(flops / (ms * 1e-3) / 1e12, 1),
                       "traffic_GB": round(traffic / 1e9, 3) if traffic else None,
                       "traffic_GBps": round(traffic / (ms * 1e-3) / 1e9, 1) if traffic else None,
                       "bound": bound, "frac_binding": round(t_roof / ms, 4)})
        fam["ms"] += ms
        fam["flops"] += flops
        fam["abytes"] += abytes
        fam["t_roof"] += t_roof
        fam["t_mfma"] += t_mfma
        fam["launches"] += st["launches"] // psteps
        if traffic:
            fam["traffic"] += traffic
            fam["t_hbm"] += t_hbm
        else:
            have_traffic = False
    if not stages:
        return None
    traffic = fam["traffic"] if have_traffic else None
    # the family's bound: the roof under which most of its at-the-roof time sits
    t_by = {"mfma": sum(s_["frac_binding"] * s_["ms"] for s_ in stages if s_["bound"] == "mfma"),
            "hbm": sum(s_["frac_binding"] * s_["ms"] for s_ in stages if s_["bound"] == "hbm")}
    bound = "hbm" if t_by["hbm"] > t_by["mfma"] else "mfma"
    tf = fam["flops"] / (fam["ms"] * 1e-3) / 1e12
    if bound == "mfma":
        achieved, peak, unit = tf, MFMA_PEAK_TF, "TFLOP/s"
    else:
        achieved, peak, unit = traffic / (fam["ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
    return {"bound": bound,
            "kernel": "HiFi-GAN ResBlock Conv1d launches (reference src/hifigan.cpp:74-185), every launch covers all utterances of the "
                      "batch; one entry per vocoder stage in `stages` (256 / 128 / 64 / 32 channels)",
            "achieved": round(achieved, 1), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4),
            "frac_is": "algorithmic TFLOP/s (true MACs of the 72 convs, no halo recompute counted) / 2.5 PFLOP/s dense f16 peak"
                       if bound == "mfma" else "real HBM bytes (PMC) / time / 8 TB/s",
            "frac_binding": round(fam["t_roof"] / fam["ms"], 4),
            "frac_binding_is": "sum over stages of max(algorithmic flops / 2.5 PFLOP/s, real traffic / 8 TB/s) / measured time: what a "
                               "kernel family sitting exactly on each stage's binding roof would score 1.0 on",
            "traffic": (traffic / fam["launches"]) if traffic else None,
            "traffic_per_pass": traffic, "traffic_source": tsrc,
            "traffic_GBps": round(traffic / (fam["ms"] * 1e-3) / 1e9, 1) if traffic else None,
            "mfma_TFLOPs": round(tf, 1), "mfma_frac_of_dense_f16_peak": round(tf / MFMA_PEAK_TF, 4),
            "algo_GBps_unfused_equiv": round(fam["abytes"] / (fam["ms"] * 1e-3) / 1e9, 1),
            "algo_GBps_unfused_equiv_is": "SURVEY.md §8d: algorithmic bytes of the UNFUSED 72-conv formulation (3.686 MB per mel frame "
                                          "+ weights) / time — can exceed the HBM peak because the fused kernels never move those bytes",
            "algo_bytes_per_launch": fam["abytes"] / fam["launches"], "ms_per_pass": round(fam["ms"], 4),
            "avg_launch_us": round(1e3 * fam["ms"] / fam["launches"], 2), "launches_per_step": fam["launches"],
            "stages": stages,
            "timing": "hipEvent pair around each stage's run of consecutive ResBlock launches on the model's stream (eager); " + timing_note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts-per-gpu", type=int, default=UTTS_PER_GPU)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[1]/[2] extras (used under rocprofv3)")
    ap.add_argument("--no-pipeline", action="store_true", help="one synchronous zv_synthesize_batch per step instead of two batches in flight")
    ap.add_argument("--dump-dir", default=None, help="every rank saves its utterances' waveforms there (tests: union of the shards)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible — the HIP path has no CPU fallback")
    # rehearsal hook: ZV_BENCH_ONE_GPU=1 puts every rank on cuda:0 with the gloo backend, so the N > 1 code path
    # (sharding, barrier, reductions, rank-0 JSON) can be exercised on a one-GPU box; never set by the driver
    one_gpu = os.environ.get("ZV_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d (LOCAL_RANK %d of WORLD_SIZE %d) has no GPU: %d device(s) visible — one process per GPU, "
                 "launch at most that many ranks" % (rank, local_rank, world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    load_package()
    from zerovox_cpp_amd import capi, gguf, sharding, synth

    g = synth.MEDIUM
    T = args.frames
    tmpdir = os.environ.get("TMPDIR", tempfile.gettempdir())
    ckpt = os.path.join(tmpdir, f"zerovox_medium_seed{SEED_W}.gguf")
    if rank == 0 and not os.path.exists(ckpt):
        synth.write_checkpoint(ckpt + ".tmp", g, SEED_W)
        os.replace(ckpt + ".tmp", ckpt)
    if world > 1:
        dist.barrier()

    # the global utterance list (32 per GPU) and this rank's contiguous shard of it
    n_global = args.utts_per_gpu * world
    lens = sharding.mixed_length_batch(SEED_BATCH, n_global)
    lo, hi = sharding.shard_utterances(n_global, world, rank)
    utts = []
    for u in range(lo, hi):
        ids, puncts, style = synth.encoder_inputs(g, 200 + u, lens[u])
        utts.append((ids, puncts, style, T))

    model = capi.Model(ckpt, device=local_rank)
    hop, sr = model.hp.audio_hop_size, model.hp.audio_sampling_rate
    model.set_graph_mode(not args.no_graph)
    call = model.prepare_batch(utts)            # host buffers allocated once; every run() is one zv_synthesize_batch
    # batches in flight (A/B hook ZV_BENCH_LANES; 2 is what is reported: with two the GPU never waits for the host —
    # extra.gpu_idle_ms_per_step, measured below without a profiler — and three or four are 0.5-1.5 % slower)
    depth = max(1, min(4, int(os.environ.get("ZV_BENCH_LANES", "2"))))
    lanes = [call] + [model.prepare_batch(utts) for _ in range(depth - 1)]   # one set of output buffers per lane
    local_audio_per_step = sum(t * hop / sr for (_, _, _, t) in utts)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        model.synchronize()

    # A serving loop keeps a batch in flight per lane (zv_synthesize_batch_begin / _end): step k is enqueued — input block,
    # upload, kernels, waveform downloads — before step k - (depth - 1) is waited for, so a step's last downloads and
    # copy-out run under the next steps' kernels and the GPU always has a batch queued.  Every step is a whole batch, host
    # ids in -> host waveforms out, and all K steps are complete when the timed region ends.  --no-pipeline: one synchronous
    # zv_synthesize_batch per step.
    def run_steps(k_steps):
        if args.no_pipeline or depth == 1:
            for _ in range(k_steps):
                call.run()
            return
        for k in range(k_steps):
            lanes[k % depth].begin(k % depth)
            if k >= depth - 1:
                j = k - (depth - 1)
                lanes[j % depth].end(j % depth)
        for j in range(max(0, k_steps - (depth - 1)), k_steps):
            lanes[j % depth].end(j % depth)

    run_steps(max(args.warmup, depth))          # every lane's arena, staging block and graph exist before the clock starts
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    # where the GPU had no batch to work on inside the timed region: gaps of the union of the batches' [first operation,
    # last kernel] intervals (HIP events the library records on the lanes' streams — no profiler attached)
    gpu_idle = None
    if not (args.no_pipeline or depth == 1):
        tl = model.batch_timeline(min(args.steps, 64))
        if len(tl) >= 2:
            span0, span1, cover, cur_s, cur_e = tl[0][0], max(e for _, e in tl), 0.0, None, None
            for s_, e_ in sorted(tl):
                if cur_e is None or s_ > cur_e:
                    if cur_e is not None:
                        cover += cur_e - cur_s
                    cur_s, cur_e = s_, e_
                else:
                    cur_e = max(cur_e, e_)
            cover += cur_e - cur_s
            gpu_idle = {"batches": len(tl), "span_ms": round(span1 - span0, 3), "idle_ms": round((span1 - span0) - cover, 3),
                        "idle_ms_per_step": round(((span1 - span0) - cover) / len(tl), 4),
                        "is": "time inside the span of the last batches' kernels at which no batch had an operation running or queued "
                              "behind a running one on its stream (zv_batch_timeline: HIP events on the lanes' streams, un-profiled run)"}
    # whole-job rate: sum of audio over ranks / max of wall over ranks
    value = sharding.aggregate_throughput(args.steps * local_audio_per_step, dt)
    rank_ms = {"min": 1e3 * dt / args.steps, "max": 1e3 * dt / args.steps}
    if world > 1:
        # the slowest and the fastest rank's own ms per step: a straggler shows the first time the scaling run happens
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        tmin = tmax.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        rank_ms = {"min": 1e3 * float(tmin.item()) / args.steps, "max": 1e3 * float(tmax.item()) / args.steps}
        dt = float(tmax.item())

    res = call.results()
    if not all(np.isfinite(w).all() and 0 < nf <= T for (w, nf) in res):
        sys.exit("bench.py: non-finite waveform or bad frame count")
    live_frames = int(sum(nf for _, nf in res))
    if args.dump_dir:
        np.savez(os.path.join(args.dump_dir, "rank%d.npz" % rank), index=np.arange(lo, hi), n_frames=np.array([nf for _, nf in res]),
                 **{"wav%d" % u: w for u, (w, _) in zip(range(lo, hi), res)})

    # ---- roofline of the dominant kernel family, same batch workload: live HIP-event timing, eager launches ----
    roofline, roofline_c1, kernels = None, None, []
    if rank == 0:
        model.set_graph_mode(False)
        call.run()
        model.profile_begin()
        psteps = 3
        for _ in range(psteps):
            call.run()
        stats = model.profile_end()
        tot_ms = sum(s["total_ms"] for s in stats)
        for s in stats:
            kernels.append({"name": s["name"], "launches_per_step": s["launches"] // psteps,
                            "avg_us": round(1e3 * s["total_ms"] / s["launches"], 2), "ms_per_step": round(s["total_ms"] / psteps, 4),
                            "share": round(s["total_ms"] / tot_ms, 4),
                            "algo_GBps": round(s["algo_bytes"] / (s["total_ms"] * 1e-3) / 1e9, 1),
                            "algo_TFLOPs": round(s["algo_flops"] / (s["total_ms"] * 1e-3) / 1e12, 1)})
        roofline = resblock_roofline(stats, psteps, "batch of %d utterances x %d frames, %d eager passes" % (len(utts), T, psteps))

    # ---- the other single-GPU configs of BASELINE.json, reported for reference (never `value`) ----
    extra = {}
    if rank == 0:
        extra["kernels"] = kernels
        extra["timed_region_s"] = round(dt, 4)
        extra["gpu_idle"] = gpu_idle
        extra["gpu_idle_ms_per_step"] = gpu_idle["idle_ms_per_step"] if gpu_idle else None
        try:
            if args.no_extras:
                raise RuntimeError("skipped (--no-extras)")
            # configs[1]: HiFi-GAN vocoder only, 512 frames, batch 1, mel resident in HBM, hipGraph replay
            _, tensors = gguf.read_gguf(ckpt)
            T1 = 512
            mel = synth.vocoder_mel(g, tensors, 7, T1)
            d_mel = model.device_alloc(mel.nbytes)
            d_wav = model.device_alloc(T1 * hop * 4)
            model.h2d(d_mel, mel)
            model.set_graph_mode(True)
            for _ in range(5):
                model.vocode_device(d_mel, T1, d_wav)
            model.synchronize()
            reps = 200
            t1 = time.perf_counter()
            for _ in range(reps):
                model.vocode_device(d_mel, T1, d_wav)
            model.synchronize()
            dt1 = (time.perf_counter() - t1) / reps
            extra["configs1_vocoder_only_512f"] = {"xrt": round(T1 * hop / sr / dt1, 1), "ms": round(1e3 * dt1, 4),
                                                   "note": "mel resident in HBM, wav left in HBM, hipGraph replay"}
            # the ResBlock family at configs[1]: the north_star's ">= 60 % of the HBM roofline for the ResBlock Conv1d" is worded on
            # this config; same formula as SURVEY.md §8d (algorithmic bytes of the 72 convs / sum of the launch times / 8 TB/s)
            model.set_graph_mode(False)
            model.vocode_device(d_mel, T1, d_wav)
            model.profile_begin()
            p1 = 20
            for _ in range(p1):
                model.vocode_device(d_mel, T1, d_wav)
            st1 = [x for x in model.profile_end() if x["name"].startswith("voc_resblock_s")]
            if st1:
                ms1 = sum(x["total_ms"] for x in st1) / p1
                ab1 = sum(x["algo_bytes"] for x in st1) / p1
                fl1 = sum(x["algo_flops"] for x in st1) / p1
                roofline_c1 = {"workload": "BASELINE.json configs[1]: HiFi-GAN vocoder, one utterance, 512 frames",
                               "bound": "hbm", "algo_MB": round(ab1 / 1e6, 1), "ms": round(ms1, 4),
                               "launches": sum(x["launches"] for x in st1) // p1,
                               "achieved": round(ab1 / (ms1 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ab1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "frac_is": "ALGORITHMIC bytes of the unfused 72-conv formulation (SURVEY.md §8d) / launch time / 8 TB/s "
                                          "(the north_star's 60 % target); at this size a launch is one round of workgroups",
                               "mfma_TFLOPs": round(fl1 / (ms1 * 1e-3) / 1e12, 1),
                               "timing": "hipEvent pair around each stage's ResBlock launches, eager, %d passes" % p1}
            model.set_graph_mode(True)
            for _ in range(3):
                model.vocode(mel)
            t1 = time.perf_counter()
            for _ in range(20):
                model.vocode(mel)
            extra["configs1_pcie_inclusive_xrt"] = round(20 * T1 * hop / sr / (time.perf_counter() - t1), 1)
            model.device_free(d_mel)
            model.device_free(d_wav)
            # configs[2]: one 128-phoneme utterance end to end, T = 512, host buffers in / out
            ids, puncts, style = synth.encoder_inputs(g, 5, 128)
            for _ in range(3):
                model.synthesize(ids, puncts, style, T1)
            t1 = time.perf_counter()
            for _ in range(20):
                model.synthesize(ids, puncts, style, T1)
            dt2 = (time.perf_counter() - t1) / 20
            extra["configs2_full_chain_128ph_512f"] = {"xrt": round(T1 * hop / sr / dt2, 1), "ms": round(1e3 * dt2, 4)}
            # the same 32 utterances one call each (what the batch path is measured against)
            model.set_graph_mode(False)
            t1 = time.perf_counter()
            for (i_, p_, s_, t_) in utts:
                model.synthesize(i_, p_, s_, t_)
            extra["one_by_one_xrt"] = round(local_audio_per_step / (time.perf_counter() - t1), 1)
        except Exception as e:      # these extras must never take the headline measurement down
            extra["extras_error"] = str(e)

    # ---- CPU baseline on this host's cores (rank 0, N = 1 only): the compiled reference end to end on one utterance ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import zvoracle
        threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        ids0, pun0, sty0, T0 = utts[0]
        audio0 = T0 * hop / sr
        if zvoracle.have_reference():
            runs = [zvoracle.run_reference_chain(ckpt, ids0, pun0, sty0, T=T0, threads=threads) for _ in range(3)]
            tot = sorted((x["timing"]["enc_s"] + x["timing"]["dec_s"] + x["timing"]["voc_s"], i) for i, x in enumerate(runs))
            t_cpu, r = tot[1][0], runs[tot[1][1]]                 # the median run
            err = float(np.sqrt(np.mean((res[0][0].astype(np.float64) - r["wav"]) ** 2)))
            cpu = {"value": round(audio0 / t_cpu, 3), "unit": "x_realtime", "cores": threads, "kind": "reference",
                   "sample": "compiled reference (ggml CPU backend, x86-64-v3 build), utterance 0 of the batch (%d phonemes, "
                             "T = %d frames = %.2f s of audio, of which the length regulator fills %d frames = %.2f s), encoder + decoder + "
                             "vocoder back to back, median of 3 runs" % (len(ids0), T0, audio0, r["n_frames"], r["n_frames"] * hop / sr),
                   "seconds": round(t_cpu, 3), "seconds_all_runs": [round(t, 3) for t, _ in tot],
                   "stage_seconds": {k: round(r["timing"][k], 3) for k in ("enc_s", "dec_s", "voc_s")},
                   "live_frames": r["n_frames"], "live_audio_xrt": round(r["n_frames"] * hop / sr / t_cpu, 3),
                   # NOT a parity figure: on these synthetic weights 10-20 % of the pitch / energy buckets sit within summation-order
                   # noise of a boundary, so an un-forced end-to-end run differs from the reference by whole bucket embeddings; parity
                   # is gated stage by stage (teacher-forced) and on a decision-robust utterance in tests/test_gpu_round3.py
                   "wav_rms_vs_reference_unforced_bucket_flips_included": err,
                   "n_frames": {"reference": r["n_frames"], "gpu": res[0][1]}}
        else:
            _, tensors = gguf.read_gguf(ckpt)
            lib = zvoracle.build(native=True, out_dir=tmpdir)
            orc = zvoracle.Oracle(tensors, lib_path=lib, threads=threads)
            Tc = 128
            t1 = time.perf_counter()
            e = orc.encoder(g, ids0, pun0, sty0, Tc)
            m_ = orc.decoder(e["hidden"], sty0)
            orc.vocoder(m_)
            t_cpu = time.perf_counter() - t1
            cpu = {"value": round(Tc * hop / sr / t_cpu, 3), "unit": "x_realtime", "cores": threads, "kind": "port",
                   "sample": "CPU port (oracle/zv_oracle.c, -march=native), utterance 0 at T = %d frames, encoder + decoder + "
                             "vocoder, 1 run" % Tc, "seconds": round(t_cpu, 3)}

    if rank == 0:
        out = {
            "metric": "audio-seconds/wall-second (xRT) end-to-end phoneme->22.05 kHz wav",
            "value": round(value, 1), "unit": "x_realtime", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "ms_per_step_rank_min": round(rank_ms["min"], 4),
            "ms_per_step_rank_max": round(rank_ms["max"], 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16*f16->f32 (MFMA) convs, f32 (MFMA) attention/linear, f32 activations", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[3]%s: per GPU a batch of %d mixed-length utterances (32..256 phonemes), "
                                   "T = %d frames each, full fs2encoder -> stylettsdec -> hifigan, host ids in / host wav out "
                                   "(H2D + D2H inside the timed region), one launch per kernel for the whole batch (the last vocoder stage's residual "
                                   "blocks + output conv in 8 utterance groups, each group's waveform download under the next group's kernels), %s" %
                                   (" x %d GPUs = configs[4]" % world if world > 1 else "", len(utts), T,
                                    ("eager launches" if args.no_graph else "hipGraph replay") +
                                    ("" if args.no_pipeline or depth == 1 else "; %d batches in flight (step k enqueued before step k - %d is waited for)" % (depth, depth - 1))),
                       "utterances_per_gpu": len(utts), "utterances_total": n_global, "frames": T,
                       "audio_seconds_per_step": round(local_audio_per_step * world, 3),
                       "live_frames_rank0": live_frames, "live_audio_seconds_per_step_rank0": round(live_frames * hop / sr, 3),
                       "phonemes_rank0": int(sum(len(u[0]) for u in utts)),
                       "parallelism": "independent utterances, contiguous shards, one process per GPU, no collective"},
            # `value` counts every vocoded frame as audio (the reference vocodes all T frames of every utterance, src/zerovox.cpp:326-334,
            # SURVEY.md §8d); the frames the length regulator actually filled are about half of them on this batch
            "live_audio_xrt": round(live_frames * hop / sr * args.steps / dt, 1) if world == 1 else None,
            "roofline": roofline, "roofline_configs1": roofline_c1, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(out))
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
