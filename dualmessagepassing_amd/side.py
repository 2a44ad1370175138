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
import threading

import torch

USE_SIDE_STREAM = True      # module attribute (tests flip it): off = everything on the caller's stream, as before


class _State(threading.local):
    """Per host thread (a thread drives one stream of launches; two threads stepping two models must not see each other's
    stages): the side streams by device, the pending stages in issue order, the fork depth."""

    def __init__(self):
        self.streams = {}
        self.pending = []       # [(name, event, device index)]
        self.depth = 0


_st = _State()


def _stream(device):
    s = _st.streams.get(device.index)
    if s is None:
        s = _st.streams[device.index] = torch.cuda.Stream(device=device)
    return s


def active():
    """True while the enclosing code runs inside ``fork()``."""
    return _st.depth > 0


def pending():
    """Names of the stages recorded and not yet waited for (this thread)."""
    return [n for n, _, _ in _st.pending]


@contextlib.contextmanager
def fork(enabled=True, device=None):
    """Run the enclosed launches on the side stream (after what the current stream holds so far).  A no-op context when the
    side stream is switched off (or ``enabled`` is false), on a CPU build, or when already inside a fork.  Several forks in a
    row queue up on the one side stream; the caller keeps every tensor the forked launches read alive until its ``join()``."""
    if not enabled or not USE_SIDE_STREAM or _st.depth > 0 or not torch.cuda.is_available():
        yield False
        return
    cur = torch.cuda.current_stream(device)
    s = _stream(cur.device)
    s.wait_stream(cur)
    _st.depth += 1
    first = len(_st.pending)
    try:
        with torch.cuda.stream(s):
            yield True
            mark("_end")     # (so that ``join()`` after a fork without stages of its own has an event to wait for)
    except BaseException:
        # a fork that failed half way: its stages name products that were never (all) made -- nobody may wait for them by name;
        # the caller's stream is ordered after whatever the side stream did get to, so nothing of it is left dangling either
        del _st.pending[first:]
        cur.wait_stream(s)
        raise
    finally:
        _st.depth -= 1


def mark(name):
    """Record stage ``name`` at this point of the side stream (inside ``fork()``)."""
    if _st.depth == 0:
        return
    ev = torch.cuda.Event()
    s = torch.cuda.current_stream()
    ev.record(s)
    _st.pending.append((name, ev, s.device.index))


def wait(name):
    """The current stream waits for stage ``name`` (no-op if it is not pending); earlier stages are complete by then too."""
    _pending = _st.pending
    if not _pending or _st.depth > 0:
        return
    for i, (n, ev, _) in enumerate(_pending):
        if n == name:
            torch.cuda.current_stream().wait_event(ev)
            del _pending[:i + 1]
            return


def join():
    """The current stream waits for everything the side stream still holds."""
    _pending = _st.pending
    if not _pending or _st.depth > 0:
        return
    torch.cuda.current_stream().wait_event(_pending[-1][1])
    del _pending[:]
