"""Utterance sharding across GPUs: one utterance = one independent unit, utterance u of a
batch goes to rank u // ceil(B / world) — contiguous blocks, e.g. 256 utterances -> 8 x 32.  There is no
data-path collective; torch.distributed is used only to agree on the wall time (max over ranks) and to
sum the audio seconds each rank produced."""
from __future__ import annotations

from typing import List, Tuple


def shard_utterances(n_utterances: int, world: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of the utterances owned by `rank`."""
    per = (n_utterances + world - 1) // world
    b = min(n_utterances, rank * per)
    return b, min(n_utterances, b + per)


def aggregate_throughput(local_audio_seconds: float, local_wall_seconds: float, group=None) -> float:
    """Whole-job audio-seconds per wall-second: sum of audio over ranks / max of wall over ranks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_audio_seconds / local_wall_seconds
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    a = torch.tensor([local_audio_seconds], dtype=torch.float64, device=dev)
    w = torch.tensor([local_wall_seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
    return float(a.item() / w.item())


def mixed_length_batch(seed: int, n_utterances: int, lo: int = 32, hi: int = 256) -> List[int]:
    """phoneme counts N_u ~ U{lo..hi} of BASELINE.json configs[3]/[4] (seeded, machine independent)"""
    from .synth import u01
    return [int(lo + x * (hi - lo + 1)) for x in u01(seed, "batch_lengths", n_utterances)]
