"""Fork/join of independent launch chains onto side HIP streams.

Inside one refinement iteration several chains of small kernels do not depend on each other (the two halves of the
motion encoder and the coarse GRUs; the two stems of the gradient predictor; its two heads; the context branch and the
candidate branch of DispRefine).  Each of those kernels fills only part of the 256 CUs, so running the chains
side by side raises occupancy.  Under HIP-graph capture the fork/join becomes parallel branches of the graph.

ON by default (TCS_MI355_STREAMS=0 disables): with the loop's kernels at 150-600 workgroups, overlapping the motion encoder
with the hidden-state update and the coarse GRUs is worth ~1.5 ms of a 32 ms frame.  On ROCm 7.2 `hipStreamEndCapture`
segfaults when a SIDE branch of a captured fork forks again (tools/repro_fork_capture.py), so a fork requested from
inside a side branch runs its branches serially on that side stream; forks nested on the ORIGIN stream are captured as
parallel graph branches.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence

import torch

def _hw_queues_allow_forks() -> bool:
    """A captured frame with parallel branches needs >= 3 hardware queues: with GPU_MAX_HW_QUEUES=1 or 2 no frame ever completed
    (profiles/r03_ab_logs.txt; the stall is in replay and needs parallel launch lists).  Most likely cause (inferred — the executor's source
    is not available here): the lists of a replayed graph are fed to their queues in batches of consecutive nodes, and a
    batch of the main list can contain a wait for a side-list node that is submitted after it: a pending barrier packet on separate hardware
    queues, a deadlock when the lists share one in-order queue.  ROCclr's default is 4; when
    the environment asks for fewer, forking is switched off (one launch list: correct on any queue count, ~2 ms per frame slower)."""
    v = os.environ.get("GPU_MAX_HW_QUEUES", "").strip()
    if not v:
        return True
    try:
        return int(v) >= 3
    except ValueError:
        return True


ENABLED = os.environ.get("TCS_MI355_STREAMS", "1") == "1"
if ENABLED and not _hw_queues_allow_forks():
    import warnings
    warnings.warn("tcs_mi355: GPU_MAX_HW_QUEUES < 3 — parallel graph branches would deadlock on this few hardware queues; running every frame as "
                  "one launch list (TCS_MI355_STREAMS=0 behaviour)")
    ENABLED = False
SITES = os.environ.get("TCS_MI355_FORK_SITES", "all").split(",")       # diagnostic: restrict forking to named call sites
OFF = set(t for t in os.environ.get("TCS_MI355_FORK_OFF", "").split(",") if t)      # diagnostic (A/B runs): call sites that run serially
# Capture order at a fork.  ROCm's graph executor cuts a captured graph into launch lists by a depth-first walk that follows a node's
# FIRST captured child: that child stays in its parent's list (same hardware queue, ~1.5 us boundary), every later child starts a
# new list whose first node waits for the parent across queues (~10 us) and whose last node the join waits for across queues
# again.  The branch given first to fork_join — by convention the one on the iteration's critical chain —
# is enqueued BEFORE the side branches (which wait for an event recorded at the fork point), so the critical chain never
# leaves its queue and the side branches, which have slack, absorb the cross-queue latencies.  (Rounds 2-3 enqueued the side branches
# first; measured against each other in profiles/r04_ab_logs.txt.)
_POOL: dict = {}
_DEPTH = 0          # nesting level of fork_join: each level owns its own side streams (a nested fork must never
                    # pick the stream it is already running on)
_IN_SIDE = 0        # > 0 while a side branch is being enqueued


def _side_streams(device, depth: int, n: int, origin: int = 0) -> List[torch.cuda.Stream]:
    """Side streams of one origin stream and nesting depth: two launch sequences enqueued on different origin streams (the extract
    stage of the next frame beside the refinement of the current one, tcs_mi355/graph.py) never share a side stream."""
    pool = _POOL.setdefault((device, depth, origin), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _keep_alive(obj, stream, seen=None):
    """Eager launches only: tensors a side chain allocated (on its side stream) and hands to the origin stream must not be
    given to another allocation while the origin stream still reads them — `record_stream` tells the caching allocator.
    (Under graph capture every allocation is static: nothing to do.)"""
    seen = set() if seen is None else seen
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _keep_alive(o, stream, seen)
    elif isinstance(obj, dict):
        for o in obj.values():
            _keep_alive(o, stream, seen)
    elif hasattr(obj, "data") and torch.is_tensor(getattr(obj, "data")):          # s16.S16
        _keep_alive(obj.data, stream, seen)


class Spawned:
    """Handle of `spawn`: the result of the function and the side stream it was enqueued on (None = ran inline)."""

    def __init__(self, result, stream):
        self.result, self.stream = result, stream


def mark():
    """An event at the current point of the current stream (a fork point for `spawn(..., after=)`); None when forking is off."""
    if not ENABLED or _IN_SIDE > 0 or not torch.cuda.is_available():
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    return ev


def spawn(fn: Callable[[], object], site: str = "", slot: int = 0, after=None) -> Spawned:
    """Enqueue `fn` on a side stream behind everything already on the current stream and return at once; `join` makes the
    current stream wait for it.  Used for work whose result is needed much later (the next iteration's gru32).  Spawns that
    are in flight at the same time take different `slot`s (one stream each).  `after` (a `mark()`): the side chain starts
    behind THAT point of the current stream instead — work enqueued on the current stream since then neither delays it nor, under
    capture, loses its place as the first child of the fork point (see "Capture order at a fork" above)."""
    global _IN_SIDE, _DEPTH
    if not ENABLED or _IN_SIDE > 0 or not torch.cuda.is_available() or ("all" not in SITES and site not in SITES) or site in OFF:
        return Spawned(fn(), None)
    cur = torch.cuda.current_stream()
    pool = _POOL.setdefault((cur.device, "spawn", cur.cuda_stream), [])
    while len(pool) <= slot:
        pool.append(torch.cuda.Stream(device=cur.device))
    st = pool[slot]
    if after is not None:
        st.wait_event(after)
    else:
        st.wait_stream(cur)
    _IN_SIDE += 1
    try:
        with torch.cuda.stream(st):
            res = fn()
    finally:
        _IN_SIDE -= 1
    return Spawned(res, st)


def join(h: Spawned):
    if h is not None and h.stream is not None:
        cur = torch.cuda.current_stream()
        cur.wait_stream(h.stream)
        if not torch.cuda.is_current_stream_capturing():
            _keep_alive(h.result, cur)
    return None if h is None else h.result


def fork_join(fns: Sequence[Callable[[], object]], site: str = "") -> list:
    """Run fns[0] on the current stream and fns[1:] on side streams; returns their results after joining.
    Every side chain starts after everything already enqueued on the current stream and the current stream waits for
    every side chain before continuing, so memory handed between the chains is ordered."""
    if not ENABLED or len(fns) <= 1 or not torch.cuda.is_available() or ("all" not in SITES and site not in SITES) or site in OFF:
        return [f() for f in fns]
    global _DEPTH, _IN_SIDE
    if _IN_SIDE > 0:
        return [f() for f in fns]          # a side branch never forks again (ROCm 7.2 hipStreamEndCapture crash, see above)
    cur = torch.cuda.current_stream()
    sides = _side_streams(cur.device, _DEPTH, len(fns) - 1, cur.cuda_stream)
    results = [None] * len(fns)
    _DEPTH += 1
    try:
        here = torch.cuda.Event()
        here.record(cur)
        results[0] = fns[0]()
        for i, st in enumerate(sides, start=1):
            st.wait_event(here)
            _IN_SIDE += 1
            try:
                with torch.cuda.stream(st):
                    results[i] = fns[i]()
            finally:
                _IN_SIDE -= 1
    finally:
        _DEPTH -= 1
    for st in sides:
        cur.wait_stream(st)
    if not torch.cuda.is_current_stream_capturing():
        for r in results[1:]:
            _keep_alive(r, cur)
    return results
