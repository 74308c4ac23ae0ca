"""HIP streams that really run beside each other.

The ROCm runtime maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues round-robin (measured on MI355X / ROCm 7.2 with 8 queues:
the default stream owns one, every 7th stream created after it lands on that same queue — tools/probe_stream_pingpong.py, profiles/round6).  Two
streams that share a queue still run INDEPENDENT kernels concurrently (nothing in the queue orders them), but as soon as they exchange events —
the engine's fork / join pattern: ~130 event edges per step — the runtime orders the shared queue with barrier packets and the pair
serialises: the step runs 1.5-3x slower with no error (Poseidon-B at batch 16: 33.1 vs 11.3 ms for the 4th engine a process creates).
HIP does not say which queue a stream got, so this module MEASURES the pattern that matters: K1 on `a`, an event `b` waits for, then K2 on
`b` beside K3 on `a`, joined by a second event — accepted when the three spin kernels take about two kernel times, not three.

(Round 5 tested "a marker on b retires while a spin kernel runs on a": true for two streams of ONE queue as well — no event edge, no
barrier — which is how a queue-sharing side stream got through: round 6, `test_overlapped_gradient_exchange_under_a_one_rank_rccl_group` run
alone, and bench.py's other_configs, 46 vs 30 ms.)

**Communicators (round 6, second finding).**  An RCCL communicator brings internal streams of its own, and every collective exchanges events
between them and the stream it is enqueued on.  A weight-gradient stream that was picked BEFORE the communicator existed can then collide with
one of those although every pair this module can see still measures as overlapping: Poseidon-B batch 16, communicator created after the first
steps: 21.6 ms per step with the exchange attached against 11.7 ms when the communicator came first — torch's process group and the C ABI's own
communicator alike (`tools/probe_rccl_alone.py`, profiles/round6/rccl_late_init_cliff.txt).  Re-picking the weight-gradient stream after the
communicator exists removes it (11.65 ms), so `OverlappedGradAllReducer.attach()` drops the cached picks (`forget`) and takes fresh ones for
the engine it attaches to: the state of a process that created its communicator first — what bench.py and the Trainer do anyway.

A verified stream is cached per (device, streams it must run beside): every engine of a process shares ONE weight-gradient stream (engines
step one at a time), so a long-lived process — a test session, a notebook, bench.py's five configurations — neither accumulates streams nor
re-measures, and a capture in progress reuses what was measured before it began."""
from __future__ import annotations

import warnings
from typing import Dict, Iterable, Tuple

import torch

_SPIN_CYCLES = 1_000_000        # ~0.5 ms of torch.cuda._sleep per kernel: long against event / launch latencies, short against anything a user would notice
_cache: Dict[Tuple, "torch.cuda.Stream"] = {}


def _fork_join_ms(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> float:
    """GPU time on `a` of: K1 on a; b waits for it; K2 on b beside K3 on a; a waits for K2."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fork, join = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        e0.record(a)
        torch.cuda._sleep(_SPIN_CYCLES)
        fork.record(a)
    b.wait_event(fork)
    with torch.cuda.stream(b):
        torch.cuda._sleep(_SPIN_CYCLES)
        join.record(b)
    with torch.cuda.stream(a):
        torch.cuda._sleep(_SPIN_CYCLES)
        a.wait_event(join)
        e1.record(a)
    torch.cuda.synchronize(a.device)
    return e0.elapsed_time(e1)


def overlaps(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """True when `b` runs beside `a` under the fork / join pattern of the engine (different hardware queues): three spin kernels — one before the
    fork on `a`, one on each stream after it — take about two kernel times.  Both streams are used once before the measurement (the runtime
    attaches a stream to its hardware queue at its first submission)."""
    dev = a.device
    if a.cuda_stream == b.cuda_stream:
        return False
    for s in (a, b):
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):          # one kernel time, on this clock
        e0.record(a)
        torch.cuda._sleep(_SPIN_CYCLES)
        e1.record(a)
    torch.cuda.synchronize(dev)
    one = e0.elapsed_time(e1)
    best = min(_fork_join_ms(a, b) for _ in range(2))       # (a foreign kernel on a shared GPU can stretch one trial)
    return best < 2.5 * one


def forget(device=None):
    """Drop the cached picks (of `device`, or all): the next `independent_stream` measures fresh candidates.  Streams already handed out stay
    valid for their holders."""
    if device is None:
        _cache.clear()
        return
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    for k in [k for k in _cache if k[0] == idx]:
        del _cache[k]


def independent_stream(device, beside: Iterable["torch.cuda.Stream"], tries: int = 24, priority: int = 0) -> "torch.cuda.Stream":
    """A stream on `device` that runs beside every stream of `beside` under event hand-offs (see the module text); cached per (device, beside).
    When none of `tries` candidates passes (one hardware queue configured, a shared GPU too busy to measure) the first candidate is returned
    with a warning: the step is then correct but may serialise."""
    beside = [s for s in beside if s is not None]
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), priority, tuple(sorted(s.cuda_stream for s in beside)))
    hit = _cache.get(key)
    if hit is not None:
        return hit
    if torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=device, priority=priority)      # nothing can be measured inside a capture (and nothing is cached)
    first = None
    for _ in range(max(1, tries)):
        s = torch.cuda.Stream(device=device, priority=priority)
        if first is None:
            first = s
        if all(overlaps(o, s) for o in beside):
            _cache[key] = s
            return s
    warnings.warn(f"poseidon_amd.streams: none of {tries} candidate streams was measured to run beside {len(beside)} other stream(s) "
                  "(GPU_MAX_HW_QUEUES=1, or a GPU too busy to measure): the weight-gradient / gradient-exchange overlap may serialise")
    _cache[key] = first
    return first
