"""A TC-Stereo frame as a two-stage pipeline, replayed from HIP graphs.

A frame has two parts.  EXTRACT — image normalisation, feature / context networks, correlation pyramid, context convolutions
(core/tc_stereo.py:101-116,147-149 of the reference) — depends on nothing but the two images.  REFINE — the temporal warp or
arg-max prior, disparity completion, hidden-state warp and the refinement loop (tc_stereo.py:119-229) — needs EXTRACT's
features and the previous frame's outputs.  Frames of a sequence are serial through REFINE only, so the EXTRACT of frame t+1
can be enqueued while the REFINE of frame t still runs: `prefetch(image1, image2)` launches it into the free one
of two feature slots, and the following `__call__` with the same image tensors finds it there.  What that buys, as measured
(profiles/r03_frame_phases.txt): the HOST-side launch work of the next EXTRACT (0.35-0.6 ms per frame) leaves
the frame's serial path; on the GPU the extraction does NOT run beside the loop (its first kernel starts 0.4-1.3 ms after the
frame's last one — the loop's graph occupies the hardware queues).  Without a prefetch the two stages simply run back to back, as
the reference does.

Each stage is ~100 / ~1,700 kernel launches with static shapes and no host decision inside, so it is captured once per
(shape, branch[, iteration count]) and slot into a HIP graph and replayed.  Inputs are copied into the graph's static
buffers, outputs are cloned out, so callers keep ordinary tensor semantics (the temporal state they pass back next frame
is theirs).  The library kernels are capture-safe by construction (no allocation, no sync — include/tcs_mi355.h).
Two slots also give two executables per stage, used in turn: on ROCm 7.2 launching an executable graph WITH parallel
branches again blocks the host until its previous launch has nearly finished (tools/capture_variance.py), and alternating
lets the host enqueue frame t+1 while frame t runs.
"""
from __future__ import annotations

import gc
import os
import warnings
from typing import Callable, Dict, List, Optional

import torch

_CAPTURING = 0          # > 0 while any FrameGraphs capture is recording (FrameGraphs.drop refuses then)
# EXTRACT is enqueued on the CALLER's stream.  Rounds 3's second stream could not make it overlap the loop on this stack (module docstring) and
# measured the same (profiles/r04_ab_logs.txt, r4_k); on the caller's stream there are no cross-stream waits to get wrong, and a prefetch
# only moves the host's launch work ahead.


def _flatten(temporal):
    if temporal is None:
        return []
    K, T, Tp, base, last_disp, nets, fmap1 = temporal
    return [K, T, Tp, base, last_disp, *nets, fmap1]


def _unflatten(flat):
    if not flat:
        return None
    return (flat[0], flat[1], flat[2], flat[3], flat[4], list(flat[5:-1]), flat[-1])


def _tensors(obj, seen=None):
    """Every torch tensor reachable from a feature bundle (lists, tuples, dicts, objects with __dict__)."""
    seen = set() if seen is None else seen
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o, seen)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o, seen)
    elif hasattr(obj, "__dict__"):
        for o in vars(obj).values():
            yield from _tensors(o, seen)


class _no_gc:
    """No cyclic garbage collection while a stream is capturing.  torch.cuda.graph() collects before it starts capturing; a collection that
    fires DURING the capture and finds an unreachable HIP graph / event of an earlier capture destroys it mid-capture, and the runtime aborts
    the process (seen once: `Fatal Python error: Aborted ... Garbage-collecting` inside `_capture_extract` after a weight change had dropped
    the previous captures)."""

    def __enter__(self):
        global _CAPTURING
        self.was = gc.isenabled()
        gc.disable()
        _CAPTURING += 1

    def __exit__(self, *exc):
        global _CAPTURING
        _CAPTURING -= 1
        if self.was:
            gc.enable()
        return False


class _Slot:
    """One of the two feature slots: what the last EXTRACT into it produced, and the events that order its reuse."""

    def __init__(self):
        self.feats = None
        self.token = None                      # identifies the images (and mode) the features belong to
        self.fresh = False                     # features nobody has consumed yet
        self.by_prefetch = False
        self.age = 0                           # calls since the features were produced without anybody consuming them
        self.ex_key = None
        self.ready = torch.cuda.Event()        # EXTRACT done (recorded on the extract stream)
        self.free = torch.cuda.Event()         # the REFINE that read the features is done (recorded on the caller's stream)


class _Entry:
    def __init__(self, graph, static_in, static_out, graph2=None):
        self.graph, self.static_in, self.static_out, self.graph2 = graph, static_in, static_out, graph2


class FrameGraphs:
    """`extract_fn(image1, image2, first) -> feats`, `head_fn(feats, temporal) -> start`, `loop_fn(feats, start, iters) -> dict`.
    REFINE is captured as two graphs, the short state-dependent head and the loop, with an event between their launches: a prefetch
    made right after a frame's call waits for THAT event, so the next frame's EXTRACT runs beside this frame's loop and not beside
    its head (a chain of small launches that the extractor's 2 400-workgroup launches would starve).

    `epoch_fn` returns a value that changes whenever a model parameter is replaced or written in place (`load_state_dict`, an
    optimiser step): a captured graph holds the packed weight images of the moment of capture, so every entry is dropped and
    re-captured when it changes.  `fell_back` counts the frames that ran eagerly because a capture failed; `strict=True` turns
    such a failure into an error instead (bench.py, tests).  `captures` counts captured REFINE keys."""

    def __init__(self, extract_fn: Callable, head_fn: Callable, loop_fn: Callable, warmup: int = 2,
                 epoch_fn: Optional[Callable[[], object]] = None, strict: bool = False):
        self.extract_fn, self.head_fn, self.loop_fn = extract_fn, head_fn, loop_fn
        self.warmup, self.epoch_fn, self.strict = warmup, epoch_fn, strict
        self.loop_start = torch.cuda.Event()   # recorded on the caller's stream between a frame's head and its loop
        self.slots = [_Slot(), _Slot()]
        self.turn = 0                          # the slot the next EXTRACT goes to (unless it still holds unconsumed features)
        self.ex: Dict[tuple, Optional[List[_Entry]]] = {}
        self.rf: Dict[tuple, Optional[List[_Entry]]] = {}
        self.epoch = epoch_fn() if epoch_fn is not None else None
        self.fell_back = 0
        self.captures = 0
        self.prefetched = 0                    # frames whose EXTRACT had been launched by prefetch()
        self._sx: Dict[object, torch.cuda.Stream] = {}

    # ---- bookkeeping ---------------------------------------------------------------------------------------------------------
    @property
    def cache(self):
        return {**{("extract", *k): v for k, v in self.ex.items()}, **{("refine", *k): v for k, v in self.rf.items()}}

    def _check_epoch(self):
        if self.epoch_fn is not None:
            now = self.epoch_fn()
            if now != self.epoch:              # weights changed: the captured packed-weight images are stale
                torch.cuda.synchronize()
                self.ex.clear()
                self.rf.clear()
                self.slots = [_Slot(), _Slot()]
                self.turn, self.epoch = 0, now
                gc.collect()                   # the dropped captures die here, not inside the next capture (see _no_gc)

    def drop(self):
        """Release every captured graph NOW (e.g. between two benchmark legs).  Destroying a HIP graph while another stream is recording a
        capture aborts the process on ROCm 7.2 (the round-3 abort came through the garbage collector; `del` / `= None` of a FrameGraphs
        or of its entries is the same thing by hand), so this refuses while any capture is active, waits for the device, drops the
        entries and collects at once."""
        if _CAPTURING:
            raise RuntimeError("FrameGraphs.drop() while a HIP graph capture is recording")
        torch.cuda.synchronize()
        self.ex.clear()
        self.rf.clear()
        self.slots = [_Slot(), _Slot()]
        self.turn = 0
        gc.collect()

    def _stream(self, device) -> torch.cuda.Stream:
        st = self._sx.get(device)
        if st is None:
            st = self._sx[device] = torch.cuda.Stream(device=device)
        return st

    def _pick_slot(self) -> int:
        """Where the next EXTRACT goes: the slot whose features have been consumed (a prefetch made for the NEXT frame must survive
        the call for the current one); with two unconsumed sets, the older.  A prefetch nobody consumed ages out (`__call__`): after
        two calls that did not match it the slot is free again and its image tensors are released — one mispredicted prefetch must not
        pin a slot (and with it the two-executable alternation) for the rest of the process."""
        for k in (self.turn, self.turn ^ 1):
            if not self.slots[k].fresh:
                return k
        return self.turn

    @staticmethod
    def _token(image1, image2, first, use_graph):
        """Identifies the frame a slot's features belong to: the image tensor OBJECTS (the slot keeps them alive, so their storage
        cannot be handed to other tensors meanwhile — a data_ptr alone can come back with other content), their versions (no
        in-place write since), the branch and the launch mode."""
        return (image1, image1._version, image2, image2._version, bool(first), bool(use_graph))

    @staticmethod
    def _same(a, b):
        return a is not None and b is not None and a[0] is b[0] and a[2] is b[2] and a[1] == b[1] and a[3] == b[3] and a[4:] == b[4:]

    @staticmethod
    def _ex_key(image1, first):
        return (tuple(image1.shape), image1.device.index, bool(first))

    @staticmethod
    def _rf_key(image1, iters, flat):
        return (tuple(image1.shape), image1.device.index, int(iters), tuple(tuple(t.shape) for t in flat))

    # ---- EXTRACT -------------------------------------------------------------------------------------------------------------
    def _capture_extract(self, key, image1, image2, first) -> Optional[List[_Entry]]:
        static_in = [image1.clone(), image2.clone()]
        run = lambda: self.extract_fn(static_in[0], static_in[1], first)
        # The warm-up and the capture write the model's pool buffers, which every extract key of this shape shares: an EXTRACT still in
        # flight on the extract stream (a prefetch) must be over first, and whatever a slot holds unconsumed is no longer trustworthy.
        for sl in self.slots:
            sl.fresh, sl.token = False, None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            side.wait_stream(self._stream(image1.device))
            with torch.cuda.stream(side):
                for _ in range(self.warmup):        # packs weights, sizes the pools
                    run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            entries = []
            for _ in range(2):                      # one executable (and one set of output tensors) per slot
                g = torch.cuda.CUDAGraph()
                with _no_gc(), torch.cuda.graph(g):
                    feats = run()
                g.replay()          # the first launch of an executable graph uploads it (~30 ms): pay that here, not in a timed frame
                entries.append(_Entry(g, static_in, feats))
            torch.cuda.synchronize()
            return entries
        except Exception as e:  # capture is an optimisation: the eager HIP path computes the same thing
            if self.strict:
                raise
            warnings.warn(f"HIP graph capture of the extract stage failed ({type(e).__name__}: {e}); running this shape with eager launches")
            torch.cuda.synchronize()
            return None

    def _launch_extract(self, si: int, image1, image2, first: bool, use_graph: bool, inputs_ready: bool = False):
        """EXTRACT into slot `si` on the extract stream: after the slot's last reader, and after everything queued on the caller's
        stream so far (the images are ready) — or, with `inputs_ready` (the images were complete before the latest frame was
        called), only after that frame's head: the extraction then overlaps its loop."""
        slot = self.slots[si]
        main = torch.cuda.current_stream()
        sx = main
        key = self._ex_key(image1, first)
        entries = None
        if use_graph:
            if key not in self.ex:
                self.ex[key] = self._capture_extract(key, image1, image2, first)
            entries = self.ex[key]
        with torch.cuda.stream(sx):
            if entries is not None:
                e = entries[si]
                e.static_in[0].copy_(image1)
                e.static_in[1].copy_(image2)
                e.graph.replay()
                slot.feats = e.static_out
            else:
                slot.feats = self.extract_fn(image1, image2, first)
                for t in _tensors(slot.feats):       # allocated on the extract stream, read on the caller's: keep the blocks alive for it
                    t.record_stream(main)
            slot.ready.record(sx)
        slot.token, slot.ex_key, slot.fresh, slot.by_prefetch, slot.age = self._token(image1, image2, first, use_graph), key, True, False, 0
        self.turn = si ^ 1

    def prefetch(self, image1, image2, first: bool = False, use_graph: bool = True, inputs_ready: bool = False) -> int:
        """Launch the EXTRACT stage of a coming frame; the next `__call__` with the same image tensor objects (unmodified) uses it.
        Call it right after the `__call__` it is to overlap with, with `inputs_ready=True` when the images were on the device before
        that call (see `_launch_extract`).  Returns the slot."""
        self._check_epoch()
        token = self._token(image1, image2, first, use_graph)
        for k in (0, 1):
            if self.slots[k].fresh and self._same(self.slots[k].token, token):
                return k                                    # already there
        si = self._pick_slot()
        self._launch_extract(si, image1, image2, first, use_graph, inputs_ready)
        self.slots[si].by_prefetch = True
        return si

    # ---- REFINE --------------------------------------------------------------------------------------------------------------
    def _capture_refine(self, key, image1, image2, iters, flat, first) -> Optional[List[_Entry]]:
        ex_key = self._ex_key(image1, first)
        try:
            for si in (0, 1):                        # both slots need features of this shape / branch to capture against
                if self.slots[si].feats is None or self.slots[si].ex_key != ex_key or self.ex.get(ex_key) is None \
                        or self.slots[si].feats is not self.ex[ex_key][si].static_out:
                    self._launch_extract(si, image1, image2, first, True)
            if self.ex.get(ex_key) is None:
                return None                          # the extract stage runs eagerly: so does this one
            torch.cuda.synchronize()
            static_in = [t.detach().float().clone() for t in flat]
            entries = []
            for si in (0, 1):
                feats = self.ex[ex_key][si].static_out
                run = lambda: self.loop_fn(feats, self.head_fn(feats, _unflatten(static_in)), iters)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(self.warmup if si == 0 else 1):
                        run()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with _no_gc(), torch.cuda.graph(g1):
                    start = self.head_fn(feats, _unflatten(static_in))
                g1.replay()                               # (the loop graph is captured against the head graph's output tensors)
                with _no_gc(), torch.cuda.graph(g2):
                    out = self.loop_fn(feats, start, iters)
                g2.replay()
                entries.append(_Entry(g1, static_in, out, graph2=g2))
                entries[-1].start = start                 # (the loop graph holds raw pointers to the head graph's outputs)
            torch.cuda.synchronize()
            self.captures += 1
            for sl in self.slots:                       # the capture used both slots: whatever was prefetched into them is gone
                sl.fresh, sl.token = False, None
            return entries
        except Exception as e:
            if self.strict:
                raise
            warnings.warn(f"HIP graph capture failed ({type(e).__name__}: {e}); running this shape with eager launches")
            torch.cuda.synchronize()
            return None

    def __call__(self, image1, image2, iters, temporal, use_graph: bool = True):
        self._check_epoch()
        flat = _flatten(temporal)
        first = temporal is None
        token = self._token(image1, image2, first, use_graph)
        entries = None
        if use_graph:
            key = self._rf_key(image1, iters, flat)
            if key not in self.rf:
                self.rf[key] = self._capture_refine(key, image1, image2, iters, flat, first)
            entries = self.rf[key]
            if entries is None:
                self.fell_back += 1
        si = next((k for k in (0, 1) if self.slots[k].fresh and self._same(self.slots[k].token, token)), None)
        for k in (0, 1):                                # unconsumed features this call does not match: one call older
            sl = self.slots[k]
            if sl.fresh and k != si:
                sl.age += 1
                if sl.age >= 2:
                    sl.fresh, sl.token = False, None
        if si is not None:
            self.prefetched += int(self.slots[si].by_prefetch)
        else:
            si = self._pick_slot()
            self._launch_extract(si, image1, image2, first, use_graph and entries is not None)
        slot = self.slots[si]
        slot.fresh, slot.token = False, None            # consumed (and the image tensors are no longer held)
        main = torch.cuda.current_stream()
        main.wait_event(slot.ready)
        if entries is not None and self.ex.get(slot.ex_key) is not None and slot.feats is self.ex[slot.ex_key][si].static_out:
            e = entries[si]
            for dst, src in zip(e.static_in, flat):
                dst.copy_(src)
            e.graph.replay()                            # head
            self.loop_start.record(main)
            e.graph2.replay()                           # loop
            o = e.static_out
            out = {"flow": o["flow"].clone(), "flow_q": o["flow_q"].clone(), "net_list": [t.clone() for t in o["net_list"]],
                   "fmap1": o["fmap1"].clone()}
        else:
            start = self.head_fn(slot.feats, temporal)
            self.loop_start.record(main)
            out = self.loop_fn(slot.feats, start, iters)
            out = dict(out, fmap1=out["fmap1"].clone())      # the slot's tensor is overwritten two frames from now
        slot.free.record(main)
        return out
