"""A second HIP stream for the index builds of a step.

A training step of the pair path builds ~40 small index arrays from the batch's structure and its 0 / 1 filter gates (the
union graph's CSR, degree coefficients, edge selectors, row masks, kept-row lists, the kept edges' class tiles, their CSR
and incidence CSR, the pooling indexes): launches of 5-20 us each that read no parameter and no feature row -- ~0.25 ms
of a 2.9 ms step when they queue up between the large kernels.  ``fork()`` runs such builds on a side stream, ordered
after everything the current stream holds so far, while the current stream goes on with the embedding and first-layer
kernels; ``mark(name)`` records a stage on the side stream, ``wait(name)`` makes the current stream wait for that stage
(and drops every earlier one: the side stream runs in order) right before its first consumer, ``join()`` waits for the
rest.  Inside a HIP-graph recording (``dp.StepGraph``) the same calls become a forked branch of the graph.

Discipline (what keeps the caching allocator's stream-local reuse safe without ``record_stream``): every fork waits for
the forking stream first, every product is consumed only after its ``wait``, and ``join()`` runs before the forward pass
returns -- so memory freed on either stream is only ever reused behind work that was ordered after its last reader.
"""
import contextlib

import torch

USE_SIDE_STREAM = True      # module attribute (tests flip it): off = everything on the caller's stream, as before

_streams = {}
_pending = []               # [(name, event, device index)] in issue order
_depth = 0


def _stream(device):
    s = _streams.get(device.index)
    if s is None:
        s = _streams[device.index] = torch.cuda.Stream(device=device)
    return s


def active():
    """True while the enclosing code runs inside ``fork()``."""
    return _depth > 0


@contextlib.contextmanager
def fork(enabled=True, device=None):
    """Run the enclosed launches on the side stream (after what the current stream holds so far).  A no-op context when the
    side stream is switched off (or ``enabled`` is false), on a CPU build, or when already inside a fork.  Several forks in a
    row queue up on the one side stream; the caller keeps every tensor the forked launches read alive until its ``join()``."""
    global _depth
    if not enabled or not USE_SIDE_STREAM or _depth > 0 or not torch.cuda.is_available():
        yield False
        return
    cur = torch.cuda.current_stream(device)
    s = _stream(cur.device)
    s.wait_stream(cur)
    _depth += 1
    try:
        with torch.cuda.stream(s):
            yield True
            mark("_end")     # (so that ``join()`` after a fork without stages of its own has an event to wait for)
    finally:
        _depth -= 1


def mark(name):
    """Record stage ``name`` at this point of the side stream (inside ``fork()``)."""
    if _depth == 0:
        return
    ev = torch.cuda.Event()
    s = torch.cuda.current_stream()
    ev.record(s)
    _pending.append((name, ev, s.device.index))


def wait(name):
    """The current stream waits for stage ``name`` (no-op if it is not pending); earlier stages are complete by then too."""
    if not _pending or _depth > 0:
        return
    for i, (n, ev, _) in enumerate(_pending):
        if n == name:
            torch.cuda.current_stream().wait_event(ev)
            del _pending[:i + 1]
            return


def join():
    """The current stream waits for everything the side stream still holds."""
    if not _pending or _depth > 0:
        return
    torch.cuda.current_stream().wait_event(_pending[-1][1])
    del _pending[:]
