#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: audio-seconds per wall-second (xRT).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): HiFi-GAN vocoder only, 80-ch mel, 512 frames, batch = 1 per GPU,
medium geometry (512 -> 32 channels, hop 300), synthetic seeded weights and mel.  One step = one pass
of the vocoder schedule over one utterance whose mel is already resident in HBM; the waveform stays in
HBM.  value = (steps x 512 frames x 300 / 22050 s) x n_gpus / wall  — whole-job audio seconds per
second.  Ranks share nothing on the data path (utterances are independent): weak scaling, no RCCL
collective inside the timed region; torch.distributed only provides the barrier and the max-over-ranks.

Extra objects on the same JSON line:
  roofline     — dominant kernel family (the ResBlock Conv1d launches): algorithmic bytes per launch
                 (3.686 MB per mel frame for the 72 ResBlock convs + their weights) divided
                 by the average launch duration measured live with HIP events on the model's stream
                 (zv_profile_begin/_end: eager launches, one event pair per launch, same K steps).
  cpu_baseline — the compiled reference (oracle/_ref/zvref, ggml CPU backend, x86-64-v3 build) timed on this
                 host on the same 512-frame workload; falls back to our CPU port (oracle/zv_oracle.c).
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
FRAMES = 512
SEED_W, SEED_MEL = 1234, 7


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[2]/[3] extras (used under rocprofv3)")
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
    # (barrier, max-over-ranks, rank-0 JSON) can be exercised on a one-GPU box; never set by the driver
    one_gpu = os.environ.get("ZV_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    load_package()
    from zerovox_cpp_amd import capi, gguf, synth

    g = synth.MEDIUM
    T = args.frames
    tmpdir = os.environ.get("TMPDIR", tempfile.gettempdir())
    ckpt = os.path.join(tmpdir, f"zerovox_medium_seed{SEED_W}.gguf")
    if rank == 0 and not os.path.exists(ckpt):
        synth.write_checkpoint(ckpt + ".tmp", g, SEED_W)
        os.replace(ckpt + ".tmp", ckpt)
    if world > 1:
        dist.barrier()
    _, tensors = gguf.read_gguf(ckpt)
    mel = synth.vocoder_mel(g, tensors, SEED_MEL + rank, T)

    model = capi.Model(ckpt, device=local_rank)
    model.reserve(1, T)
    hop, sr = model.hp.audio_hop_size, model.hp.audio_sampling_rate
    d_mel = model.device_alloc(mel.nbytes)
    d_wav = model.device_alloc(T * hop * 4)
    model.h2d(d_mel, mel)
    model.set_graph_mode(not args.no_graph)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        model.synchronize()

    for _ in range(args.warmup):
        model.vocode_device(d_mel, T, d_wav)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.vocode_device(d_mel, T, d_wav)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    wav = np.empty(T * hop, np.float32)
    model.d2h(wav, d_wav)
    if not np.isfinite(wav).all():
        sys.exit("bench.py: non-finite waveform")

    audio_s = T * hop / sr
    value = args.steps * audio_s * world / dt

    # ---- roofline of the dominant kernel family: live HIP-event timing, eager launches, same K steps ----
    roofline = None
    kernels = []
    if rank == 0:
        model.set_graph_mode(False)
        for _ in range(3):
            model.vocode_device(d_mel, T, d_wav)
        model.synchronize()
        model.profile_begin()
        psteps = min(args.steps, 50)
        for _ in range(psteps):
            model.vocode_device(d_mel, T, d_wav)
        stats = model.profile_end()
        tot_ms = sum(s["total_ms"] for s in stats)
        for s in stats:
            kernels.append({"name": s["name"], "launches_per_step": s["launches"] // psteps,
                            "avg_us": 1e3 * s["total_ms"] / s["launches"], "share": s["total_ms"] / tot_ms,
                            "algo_GBps": s["algo_bytes"] / (s["total_ms"] * 1e-3) / 1e9,
                            "algo_TFLOPs": s["algo_flops"] / (s["total_ms"] * 1e-3) / 1e12})
        rb = next(s for s in stats if s["name"] == "voc_resblock_conv")
        achieved = rb["algo_bytes"] / (rb["total_ms"] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_resblock_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "HiFi-GAN ResBlock Conv1d launches (conv1d_mfma_kernel<1,4> x6, resblock_pair_kernel<128|64,2> x3 each, "
                              "resblock_triple_kernel<32,2,256> x1 at 512 frames)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algo_bytes_per_launch": rb["algo_bytes"] / rb["launches"],
                    "avg_launch_us": round(1e3 * rb["total_ms"] / rb["launches"], 2),
                    "launches_per_step": rb["launches"] // psteps,
                    "mfma_TFLOPs": round(rb["algo_flops"] / (rb["total_ms"] * 1e-3) / 1e12, 1),
                    "timing": "hipEvent pair around each stage's run of consecutive ResBlock launches on the model's stream "
                              "(eager, 3-6 launches per pair), %d steps" % psteps}

    # ---- PCIe-inclusive rate (host mel in, host wav out) — reported, never `value` ----
    extra = {}
    if rank == 0:
        model.set_graph_mode(not args.no_graph)
        for _ in range(3):
            model.vocode(mel)
        t1 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            model.vocode(mel)
        extra["pcie_inclusive_xrt"] = round(reps * audio_s / (time.perf_counter() - t1), 1)
        extra["kernels"] = kernels
        # the other single-GPU configs of BASELINE.json, reported for reference (never `value`):
        #   configs[2] full chain phoneme -> wav, 128 phonemes, T = 512, batch 1 (host buffers in/out)
        #   configs[3] batch of 32 mixed-length utterances (32..256 phonemes), T = 1024 each, 4 in-flight lanes
        try:
            if args.no_extras:
                raise RuntimeError("skipped (--no-extras)")
            from zerovox_cpp_amd import sharding
            model.set_graph_mode(False)
            ids, puncts, style = synth.encoder_inputs(g, 5, 128)
            for _ in range(3):
                model.synthesize(ids, puncts, style, T)
            t1 = time.perf_counter()
            for _ in range(10):
                model.synthesize(ids, puncts, style, T)
            extra["full_chain_128ph_T%d_xrt" % T] = round(10 * audio_s / (time.perf_counter() - t1), 1)
            utts = []
            for u, n in enumerate(sharding.mixed_length_batch(3, 32)):
                i_, p_, s_ = synth.encoder_inputs(g, 200 + u, n)
                utts.append((i_, p_, s_, 1024))
            model.synthesize_batch(utts[:4])
            t1 = time.perf_counter()
            model.synthesize_batch(utts)
            extra["batch32_mixed_T1024_xrt"] = round(32 * 1024 * hop / sr / (time.perf_counter() - t1), 1)
            # the same ResBlock kernels once a launch has many rounds of workgroups (8 192 frames in one call): the
            # 512-frame single utterance of configs[1] is one round per launch and pays every launch's fixed
            # latencies (dispatch, first loads, store drain) in full
            TL = 8192
            mel_l = np.tile(mel, (TL // T + 1, 1))[:TL]
            d_ml, d_wl = model.device_alloc(mel_l.nbytes), model.device_alloc(TL * hop * 4)
            model.h2d(d_ml, mel_l)
            model.reserve(1, TL)
            for _ in range(2):
                model.vocode_device(d_ml, TL, d_wl)
            model.synchronize()
            model.profile_begin()
            for _ in range(5):
                model.vocode_device(d_ml, TL, d_wl)
            st_l = model.profile_end()
            rbl = next(s_ for s_ in st_l if s_["name"] == "voc_resblock_conv")
            tot_l = sum(s_["total_ms"] for s_ in st_l) / 5
            extra["long_utterance_T%d" % TL] = {
                "resblock_algo_GBps": round(rbl["algo_bytes"] / (rbl["total_ms"] * 1e-3) / 1e9, 1),
                "resblock_frac_of_hbm_peak": round(rbl["algo_bytes"] / (rbl["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "resblock_mfma_TFLOPs": round(rbl["algo_flops"] / (rbl["total_ms"] * 1e-3) / 1e12, 1),
                "kernel_ms_per_pass": round(tot_l, 3), "xrt_kernel_time": round(TL * hop / sr / (tot_l * 1e-3), 1)}
            model.device_free(d_ml)
            model.device_free(d_wl)
        except Exception as e:      # these extras must never take the headline measurement down
            extra["extras_error"] = str(e)

    # ---- CPU baseline on this host's cores (rank 0, N = 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import zvoracle
        threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        if zvoracle.have_reference():
            r = zvoracle.run_reference(ckpt, T=T, threads=threads, reps=3, voc=mel)
            t_cpu = r["timing"]["voc_s"]
            err = float(np.sqrt(np.mean((wav.astype(np.float64) - r["wav"]) ** 2)))
            cpu = {"value": round(audio_s / t_cpu, 3), "unit": "x_realtime", "cores": threads, "kind": "reference",
                   "sample": "compiled reference (ggml CPU, x86-64-v3 build), HiFi-GAN %d frames, best of 3 runs" % T,
                   "seconds": round(t_cpu, 3), "gpu_vs_reference_wav_rms": err}
        else:
            lib = zvoracle.build(native=True, out_dir=tmpdir)
            orc = zvoracle.Oracle(tensors, lib_path=lib, threads=threads)
            Tc = min(T, 128)
            t1 = time.perf_counter()
            ref = orc.vocoder(mel[:Tc])
            t_cpu = time.perf_counter() - t1
            cpu = {"value": round(Tc * hop / sr / t_cpu, 3), "unit": "x_realtime", "cores": threads, "kind": "port",
                   "sample": "CPU port (oracle/zv_oracle.c, -march=native), HiFi-GAN first %d frames, 1 run" % Tc,
                   "seconds": round(t_cpu, 3)}

    if rank == 0:
        out = {
            "metric": "audio-seconds/wall-second (xRT), HiFi-GAN vocoding 80-ch mel -> 22.05 kHz wav",
            "value": round(value, 1), "unit": "x_realtime", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16*f16->f32 (MFMA), f32 activations", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: HiFi-GAN vocoder only, 80-ch mel, %d frames, batch=1 per GPU, "
                                   "medium geometry (512ch, x300), mel resident in HBM, %s" %
                                   (T, "eager launches" if args.no_graph else "hipGraph replay"),
                       "frames": T, "audio_seconds_per_step": round(audio_s, 4), "utterances_per_gpu": 1,
                       "parallelism": "independent utterances, one process per GPU, no collective"},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(out))
    model.device_free(d_mel)
    model.device_free(d_wav)
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
