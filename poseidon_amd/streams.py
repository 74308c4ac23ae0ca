"""HIP streams that really run beside each other.

The ROCm runtime maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues, and torch hands out its 32 pooled streams round-robin: in
a long-lived process (a test session, a notebook, a trainer that builds several models) two streams that are meant to overlap — the engine's
main chain and its weight-gradient stream, or either of them and the gradient exchange's comm stream — sooner or later share a queue and
serialise: the step runs 1.5-3x slower with no error (`test_overlapped_gradient_exchange_under_a_one_rank_rccl_group` caught it).
HIP does not say which queue a stream got, so this module MEASURES it: a candidate stream is accepted when a marker on it retires while a
spin kernel is still running on each stream it must overlap with."""
from __future__ import annotations

import time
from typing import Iterable

import torch

_SPIN_CYCLES = 4_000_000        # ~2 ms of torch.cuda._sleep: long against a marker's ~20 us, short against anything a user would notice


def overlaps(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """True when a kernel enqueued on `b` retires while a kernel is still running on `a` (different hardware queues).  `b` is used once
    before the measurement (the runtime attaches a stream to its hardware queue at its first submission), and what is timed on it is a real
    launch — an event recorded on an idle stream is complete without ever visiting the queue."""
    dev = a.device
    with torch.cuda.stream(b):
        cell = torch.zeros(8, device=dev)
        cell.add_(1.0)
    torch.cuda.synchronize(dev)
    done_a, done_b = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(_SPIN_CYCLES)
        done_a.record()
    with torch.cuda.stream(b):
        cell.add_(1.0)
        done_b.record()
    t0 = time.perf_counter()
    ok = False
    while time.perf_counter() - t0 < 0.2:
        if done_b.query():
            ok = not done_a.query()
            break
        if done_a.query():
            break
    torch.cuda.synchronize(dev)
    return ok


def independent_stream(device, beside: Iterable["torch.cuda.Stream"], tries: int = 24, priority: int = 0) -> "torch.cuda.Stream":
    """A stream on `device` that overlaps every stream of `beside`; the first candidate when none of `tries` does (one hardware queue
    configured, or a capture in progress: nothing can be measured then)."""
    beside = [s for s in beside if s is not None]
    first = None
    for _ in range(max(1, tries)):
        s = torch.cuda.Stream(device=device, priority=priority)
        if first is None:
            first = s
        if torch.cuda.is_current_stream_capturing():
            return s
        if all(overlaps(o, s) and overlaps(s, o) for o in beside):
            return s
    return first
