"""Data parallelism over (pattern, graph) pairs: one process per GPU, gradients summed
with ONE all-reduce of a flat fp32 buffer per step (RCCL over xGMI on the GPU box;
``gloo`` in the CPU tests).

The reference is single-device (SubgraphCountingMatching/train.py:1080-1083); pairs are
independent in forward and backward as long as BatchNorm is off (the default,
config.py:201-207), and the losses are means over the local batch (train.py:463-480), so
averaging the per-rank gradients reproduces the global-batch gradient for equal shards.

The payload is small (~0.6 M parameters = 2.4 MB at H=128): latency-bound, so it is sent
as one message instead of per-parameter buckets.
"""
from collections import OrderedDict

import os

import torch
import torch.distributed as dist


def shard_range(num_items, rank, world):
    """Contiguous shard [lo, hi) of ``num_items`` pairs for ``rank``; sizes differ by <= 1."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradSync:
    """Makes every trainable parameter's ``.grad`` a view into one contiguous buffer.

    Parameters that get no gradient in a step (the reference has such: ``out_weight``
    without REVFLAG, UNC ``nfc``/``efc``, model.py:137-138) simply keep zeros there.
    Use ``zero()`` instead of ``optimizer.zero_grad()`` (which would drop the views).
    """

    def __init__(self, module, group=None, average=True, force_collective=False):
        """``force_collective``: issue the all-reduce / broadcast on a ONE-rank process group too (a one-rank RCCL group on
        a single GPU runs the code path of the N > 1 step: ``bench.py --force-collective``)."""
        self.group = group
        self.average = average
        self.force_collective = bool(force_collective)
        seen, params = set(), []
        for p in module.parameters():  # shared modules (share_rep_net) appear once
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("no trainable parameters")
        dev, dt = params[0].device, params[0].dtype
        for p in params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all parameters must share device and dtype")
        self.params = params
        # every slice starts on a 16-byte boundary (the HIP kernels take 16-byte-aligned parameter pointers when the
        # parameters themselves are moved into such a buffer, flatten_parameters): sizes rounded up to 4 elements,
        # the padding stays zero
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        # the payload of the all-reduce: the flat gradient, then one 0 / 1 indicator per parameter tensor ("this rank produced
        # a gradient for it this step").  Summed with the gradient, the indicators tell every rank the UNION of the live
        # parameters, so that all replicas step the same set (a tensor that is dead on this rank's batch but live on another's
        # holds a non-zero average after the sum: an optimizer that skipped it here would let the replicas drift apart)
        P = len(params)
        self._buf = torch.zeros(off + (P + 3) // 4 * 4, device=dev, dtype=dt)
        self.flat = self._buf[:off]
        self.live = self._buf[off:off + P]
        self.live.fill_(1.0)
        for p, off in zip(params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
        _warn_batch_norm(module, self)

    def flatten_parameters(self):
        """Move every parameter's storage into one contiguous buffer (each ``p.data`` becomes a view of
        it) and return ONE ``nn.Parameter`` over that buffer whose ``.grad`` is the flat gradient:
        an elementwise optimizer (Adam / AdamW / SGD with one parameter group) stepped on it performs
        exactly the per-parameter updates, in one launch instead of one multi-tensor chunk list."""
        if getattr(self, "master", None) is None:
            store = torch.zeros_like(self.flat)
            for p, off in zip(self.params, self.offsets):
                v = store[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
            self.master = torch.nn.Parameter(store, requires_grad=True)
            self.master.grad = self.flat
            # where the parameter tensors sit in the buffer: FlatAdamW keeps one step count per tensor, as torch.optim.AdamW does
            self.master._dmp_seg = (torch.tensor(self.offsets + [int(self.flat.numel())], dtype=torch.int64, device=self.flat.device),
                                    len(self.params))
            self.master._dmp_live_params = None
            self.master._dmp_live_dev = self.live if self.collective else None
        return self.master

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def collective(self):
        """Does ``sync()`` issue a collective (several ranks, or a forced one-rank group)?"""
        return self.world > 1 or (self.force_collective and dist.is_available() and dist.is_initialized())

    def zero(self):
        self.flat.zero_()
        self._zeroed_for = None

    # ---- alternative to accumulating into the views: let autograd produce fresh gradient tensors
    # (no per-parameter ``grad += new`` launches) and pack them with one multi-tensor copy
    def detach_grads(self):
        """Call before backward instead of ``zero()``: parameters start without a ``.grad``."""
        for p in self.params:
            p.grad = None

    def pack(self):
        """After backward: copy the fresh gradients into the flat buffer (zeros where a parameter
        got none) and point every ``.grad`` at its slice again."""
        views = getattr(self, "_views", None)
        if views is None:                                    # the slices never change: built once
            views = [self.flat[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self.offsets)]
            self._views = views
        dst, src, which, missing, live, absent = [], [], [], False, [], []
        for i, (p, v) in enumerate(zip(self.params, views)):
            g = p.grad
            if g is None:
                missing = True
                absent.append(i)
            else:
                live.append(i)
                if g is not v:
                    dst.append(v)
                    src.append(g)
                    which.append(i)
            p.grad = v
        hip = self.flat.is_cuda and self.flat.dtype == torch.float32 and all(g.dtype == torch.float32 for g in src)
        if missing and hip and dst:
            # ... cleared by the packing launch itself (segments without a source): no launch of its own, nothing to remember
            self._zeroed_for = None
            self._note_live(live)
            self._pack_hip(src, which, absent)
            return
        if missing:                                          # parameters without a gradient this step keep zeros:
            key = tuple(live)                                # their slices were zero before and nothing writes them, so the
            # buffer is cleared only when the set of live parameters changes.  Inside a stream capture the clearing is
            # ALWAYS recorded (whether it is needed at a replay depends on what ran in between, which the recording cannot
            # know: an eager step with a larger live set would leave its values in the slices this recording treats as
            # missing), and the host-side note is dropped so that the next eager step clears too.
            capturing = self.flat.is_cuda and torch.cuda.is_current_stream_capturing()
            if capturing or getattr(self, "_zeroed_for", None) != key:
                self.flat.zero_()
            self._zeroed_for = None if capturing else key
        else:
            self._zeroed_for = None
        self._note_live(live if missing else None)
        if not dst:
            return
        if hip:
            self._pack_hip(src, which)                       # one launch (csrc/dmp_fused.hip::pack_segments_kernel)
        else:
            torch._foreach_copy_(dst, src)

    def _note_live(self, live):
        """torch.optim.AdamW (the reference's optimizer, train.py:1231) skips parameters whose ``.grad`` is None: no weight
        decay, no moment update.  The flat optimizer sees one parameter; it is told which contiguous runs of the buffer
        belong to parameters that received a gradient this step (None = all of it).

        With a collective (``world > 1``) the set must be the same on every rank: this rank's 0 / 1 indicators go into the
        tail of the all-reduce's payload (``self.live``, a device-to-device copy of a cached table: no host sync, replays),
        the optimizer reads the summed indicators from the device (``dmp_adamw_step_segments(live_dev=...)``) and the
        host-side set is not used.  An optimizer without per-tensor segments (more than 1024 tensors) treats every tensor as
        live then -- the same on every rank, which is what keeps the replicas together."""
        if self.collective:
            P = len(self.params)
            key = None if live is None else tuple(live)
            tabs = self.__dict__.setdefault("_live_tabs", {})
            tab = tabs.get(key)
            if tab is None:
                host = torch.ones(P) if key is None else torch.zeros(P).index_fill_(0, torch.tensor(list(key), dtype=torch.int64), 1.0)
                tab = tabs[key] = host.to(self.live.device, self.live.dtype)
            self.live.copy_(tab)
            live = None
        master = getattr(self, "master", None)
        if master is None:
            return
        master._dmp_live_dev = self.live if self.collective else None
        master._dmp_live_params = None if live is None else tuple(live)
        if live is None:
            master._dmp_live_runs = None
            return
        key = tuple(live)
        cached = getattr(self, "_live_cache", None)
        if cached is None or cached[0] != key:
            runs = []
            for i in live:
                off, n = self.offsets[i], (self.params[i].numel() + 3) // 4 * 4
                if runs and runs[-1][0] + runs[-1][1] == off:
                    runs[-1][1] += n
                else:
                    runs.append([off, n])
            cached = self._live_cache = (key, [(a, b) for a, b in runs])
        master._dmp_live_runs = cached[1]

    def _pack_hip(self, src, which, absent=()):
        """``absent``: parameters without a gradient -- their slices are cleared by the same launch (sources NULL)."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        n = len(src) + len(absent)
        key = (tuple(which), tuple(absent))
        cached = getattr(self, "_pack_tables", None)
        if cached is None or cached[0] != key:               # the slice table only changes with the set of gradients
            offs = (ctypes.c_int64 * n)(*[self.offsets[i] for i in list(which) + list(absent)])
            lens = (ctypes.c_int64 * n)(*[self.params[i].numel() for i in list(which) + list(absent)])
            cached = self._pack_tables = (key, offs, lens)
        src = [g if g.is_contiguous() else g.contiguous() for g in src]
        ptrs = (ctypes.c_void_p * n)(*([g.data_ptr() for g in src] + [None] * len(absent)))
        _lib.check(lib.dmp_pack_segments(ptrs, cached[1], cached[2], n, 1, self.flat.data_ptr(), _lib.stream_ptr()),
                   "dmp_pack_segments")

    def broadcast_parameters(self, src=0):
        if self.collective:
            if getattr(self, "master", None) is not None:
                dist.broadcast(self.master.data, src=src, group=self.group)
                return
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.group)

    def sync(self, async_op=False):
        """Sum the flat gradient over ranks (and divide by the world size if ``average``).  ``async_op``: returns the
        work handle at once; ``finish(handle)`` makes the current stream wait for the sum (no host block with RCCL) and
        applies the average -- whatever is enqueued in between (the next batch's collate and index build, which do not
        depend on the parameters) overlaps the collective."""
        w = self.world
        if not self.collective:
            return None
        # (gradient + live indicators: one message; the indicators stay sums -- any value above 0 means "live somewhere")
        if async_op:
            return dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            self.flat.div_(w)
        return None

    def finish(self, work):
        if work is not None:
            work.wait()
            if self.average:
                self.flat.div_(self.world)


def _warn_batch_norm(module, sync):
    """BatchNorm couples the pairs of a batch (``rep_dmpnn_batch_norm True``, config.py:201-207): under data
    parallelism every rank normalises with the statistics of ITS shard (the running statistics are not exchanged either), so
    a multi-rank run is not the single-device run of the global batch.  Said once per process, where the collective is set up."""
    if not sync.collective or sync.world < 2:
        return
    bn = [n for n, m in module.named_modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    if bn:
        import warnings
        warnings.warn("FlatGradSync over %d ranks with BatchNorm modules (%s%s): statistics are PER SHARD -- each rank normalises "
                      "with its own pairs and keeps its own running statistics; the result differs from the single-device run of "
                      "the global batch" % (sync.world, ", ".join(bn[:3]), ", ..." if len(bn) > 3 else ""), RuntimeWarning, stacklevel=3)


USE_SEGMENT_STEPS = True   # one AdamW step count per parameter tensor of a flat buffer


class FlatAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` (optionally ``amsgrad``: the reference's optimizer, train.py:1231) whose step is
    ONE HIP launch per parameter tensor (``dmp_adamw_step``) -- meant for the single flat parameter of
    ``FlatGradSync.flatten_parameters()``.  Learning-rate schedulers work as usual (``param_groups``).
    fp32 contiguous CUDA parameters only; no CPU path.

    On the flat parameter of ``FlatGradSync.flatten_parameters()`` every parameter TENSOR of the buffer keeps its own step
    count (``dmp_adamw_step_segments``: a tensor that first receives a gradient at step k starts its bias corrections at
    1, one without a gradient is left alone -- ``torch.optim.AdamW``'s per-tensor state); the counts and the learning rate
    live in device memory, so such a step also replays from a HIP graph.

    ``capturable=True``: the step count and the learning rate live in device memory (``dmp_adamw_step_dev``), so a step
    recorded in a HIP graph (``StepGraph``) replays correctly; outside a capture ``step()`` writes the group's current
    learning rate to the device itself, around a replay ``StepGraph`` calls ``sync_hyper()``."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self.capturable = bool(capturable)
        self.veto = None      # (int32 [4] device tensor, mask): ``set_veto``

    def set_veto(self, word, mask=1):
        """Drop a step when a device-side flag is up (``dmp_adamw_step_guarded``), as a loss-scaling optimizer drops a step
        whose gradients overflowed: ``word`` = int32 [4] on the device, ``word[0]`` = flags raised since the last step
        (``GraphAdjModelV2.set_gate_capacity``: a batch that kept more edges than the capacity ORs bit 0 -- its gradients
        are wrong).  A dropped step leaves parameters, moments and the step count alone; ``word[3]`` counts them.  No
        host sync, so it replays; the step count then lives on the device whether or not ``capturable`` was asked for."""
        if word is not None and sum(len(g["params"]) for g in self.param_groups) != 1:
            # the veto word is consumed (flags cleared, drop latched) by the ONE launch of the one parameter tensor: with
            # several tensors the first launch would clear it and the others would apply the flagged gradients
            raise ValueError("FlatAdamW.set_veto: the optimizer must hold exactly one (flat) parameter tensor, "
                             "FlatGradSync.flatten_parameters()")
        self.veto = None if word is None else (word, int(mask))
        return self

    def _device_state(self, p, group):
        st = self.state[p]
        if "dev" not in st:
            # called after this step's host-side increment: the device count starts one below (the launch adds 1 first)
            st["dev"] = torch.tensor([float(st.get("step", 1) - 1), float(group["lr"])], dtype=torch.float64, device=p.device)
            st["dev_lr"] = float(group["lr"])
        return st["dev"]

    @torch.no_grad()
    def sync_state(self):
        """Read the device-side step counts back into ``state[p]["step"]`` (and ``["seg_steps"]``: one count per parameter
        tensor of a flat buffer) -- one host sync: graph replays advance only the device counts.  ``state_dict()`` calls it."""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st and "dev_seg" in st:
                    host = st["dev_seg"].tolist()
                    st["step"] = int(round(host[0]))
                    st["seg_steps"] = [int(round(x)) for x in host[2:]]
                elif st and "dev" in st:
                    st["step"] = int(round(float(st["dev"][0].item())))

    def _segment_state(self, p, group, P):
        """Device state of the per-tensor form: doubles [steps taken, lr, step of tensor 0, ..., step of tensor P - 1]."""
        st = self.state[p]
        if "dev_seg" not in st:
            seg = st.get("seg_steps")
            if seg is None or len(seg) != P:
                seg = [st.get("step", 1) - 1] * P               # an older state: every tensor at the buffer's count
            st["dev_seg"] = torch.tensor([float(st.get("step", 1) - 1), float(group["lr"])] + [float(x) for x in seg],
                                         dtype=torch.float64, device=p.device)
            st["seg_tab"] = torch.empty(2 * P, dtype=torch.float32, device=p.device)
            st["dev_lr"] = float(group["lr"])
        return st["dev_seg"], st["seg_tab"]

    def state_dict(self):
        """``torch.optim.Optimizer.state_dict`` with the step counts current (``sync_state``) and without the device-side
        scalars (``dev`` / ``dev_lr``: rebuilt from ``step`` and the group's rate on the next step), so the saved state
        loads into a capturable or a plain ``FlatAdamW`` alike."""
        self.sync_state()
        sd = super().state_dict()
        sd["state"] = {k: {n: v for n, v in st.items() if n not in ("dev", "dev_lr", "dev_seg", "seg_tab")} for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():                       # the device scalars follow the loaded step count
            st.pop("dev", None)
            st.pop("dev_lr", None)
            st.pop("dev_seg", None)
            st.pop("seg_tab", None)
            if torch.is_tensor(st.get("step")):
                st["step"] = int(st["step"].item())

    @torch.no_grad()
    def sync_hyper(self):
        """Write every group's current learning rate to its parameters' device state (one tiny launch per parameter whose
        rate changed; nothing when it did not).  Not capturable: call it before a graph replay."""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st and ("dev" in st or "dev_seg" in st) and st["dev_lr"] != float(group["lr"]):
                    for key in ("dev", "dev_seg"):
                        if key in st:
                            st[key][1:2].fill_(float(group["lr"]))
                    st["dev_lr"] = float(group["lr"])

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib
        loss = closure() if closure is not None else None
        lib = _lib.load()
        if self.veto is not None and sum(len(g["params"]) for g in self.param_groups) != 1:      # (add_param_group after set_veto)
            raise ValueError("FlatAdamW: a veto word needs exactly one (flat) parameter tensor")
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_gpu(p, p.grad)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise ValueError("FlatAdamW needs contiguous fp32 parameters and gradients")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                    if group["amsgrad"]:
                        st["max_exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                seg = getattr(p, "_dmp_seg", None)
                if seg is not None and USE_SEGMENT_STEPS and seg[1] <= 1024 and seg[0].device == p.device:
                    # a flat buffer of parameter tensors: one step count per tensor (torch.optim.AdamW's state), on the device
                    import ctypes
                    dev, tab = self._segment_state(p, group, seg[1])
                    if not torch.cuda.is_current_stream_capturing():
                        self.sync_hyper()
                    live = getattr(p, "_dmp_live_params", None)
                    live_dev = getattr(p, "_dmp_live_dev", None)      # a data-parallel run: the union over the ranks, on the device
                    if live_dev is not None:
                        _lib.require_gpu(live_dev)
                        live = None
                    bits = None
                    if live is not None:
                        words = [0] * ((seg[1] + 63) // 64)
                        for i in live:
                            words[i >> 6] |= 1 << (i & 63)
                        bits = (ctypes.c_uint64 * len(words))(*words)
                    veto = self.veto
                    if veto is not None:
                        _lib.require_gpu(veto[0])
                    _lib.check(lib.dmp_adamw_step_segments(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                                           st["exp_avg_sq"].data_ptr(), _lib.ptr(st.get("max_exp_avg_sq")), p.numel(),
                                                           dev.data_ptr(), seg[0].data_ptr(), seg[1], bits, _lib.ptr(live_dev),
                                                           tab.data_ptr(), b1, b2,
                                                           group["eps"], group["weight_decay"],
                                                           None if veto is None else veto[0].data_ptr(), 0 if veto is None else veto[1],
                                                           _lib.stream_ptr()), "dmp_adamw_step_segments")
                    continue
                state = None
                if self.capturable or self.veto is not None:
                    state = self._device_state(p, group)
                    if not torch.cuda.is_current_stream_capturing():
                        self.sync_hyper()
                # runs of the buffer whose parameters received a gradient (FlatGradSync.pack); parameters without one
                # are left untouched like torch.optim.AdamW does (they share the buffer's step count, though)
                runs = getattr(p, "_dmp_live_runs", None) or [(0, p.numel())]
                mx = st.get("max_exp_avg_sq")
                gaps, pos = [], 0                            # the complement of the live runs: what the launch leaves alone
                for off, n in runs:
                    if off > pos:
                        gaps.append((pos, off))
                    pos = off + n
                if pos < p.numel():
                    gaps.append((pos, p.numel() // 4 * 4))
                if len(gaps) <= 16:                          # DMP_ADAMW_MAX_SKIP: one launch
                    import ctypes
                    lo = (ctypes.c_int64 * max(len(gaps), 1))(*[g[0] for g in gaps])
                    hi = (ctypes.c_int64 * max(len(gaps), 1))(*[g[1] for g in gaps])
                    if state is not None and self.veto is not None:
                        _lib.require_gpu(self.veto[0])
                        _lib.check(lib.dmp_adamw_step_guarded(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                                              st["exp_avg_sq"].data_ptr(), _lib.ptr(mx), p.numel(), state.data_ptr(),
                                                              b1, b2, group["eps"], group["weight_decay"], lo, hi, len(gaps),
                                                              self.veto[0].data_ptr(), self.veto[1], _lib.stream_ptr()),
                                   "dmp_adamw_step_guarded")
                        continue
                    if state is not None:
                        _lib.check(lib.dmp_adamw_step_dev(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                                          st["exp_avg_sq"].data_ptr(), _lib.ptr(mx), p.numel(), state.data_ptr(),
                                                          b1, b2, group["eps"], group["weight_decay"], lo, hi, len(gaps),
                                                          _lib.stream_ptr()), "dmp_adamw_step_dev")
                        continue
                    _lib.check(lib.dmp_adamw_step_skip(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                                       st["exp_avg_sq"].data_ptr(), _lib.ptr(mx), p.numel(), float(group["lr"]), b1, b2,
                                                       group["eps"], group["weight_decay"], st["step"], lo, hi, len(gaps),
                                                       _lib.stream_ptr()), "dmp_adamw_step_skip")
                    continue
                if state is not None:
                    raise ValueError("FlatAdamW(capturable=True / set_veto): more than 16 gaps between the parameters with gradients")
                for off, n in runs:
                    b = 4 * off
                    _lib.check(lib.dmp_adamw_step(p.data_ptr() + b, p.grad.data_ptr() + b, st["exp_avg"].data_ptr() + b,
                                                  st["exp_avg_sq"].data_ptr() + b, None if mx is None else mx.data_ptr() + b, n,
                                                  float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], st["step"],
                                                  _lib.stream_ptr()), "dmp_adamw_step")
        return loss


def _drain_collectives():
    """Before a stream capture: make sure no process group's watchdog still holds a work object.  RCCL's watchdog thread
    polls the completion events of the collectives on its list; a poll that lands inside a stream capture fails in the
    runtime and aborts the process (found by the one-rank RCCL run of round 5).  After ``torch.cuda.synchronize()`` every
    collective is complete but stays listed until the watchdog's next pass: ``ProcessGroup._wait_for_pending_works`` blocks
    until that list is EMPTY -- a condition, not a delay -- and with nothing listed the watchdog has nothing to poll while
    the recording runs (the recording itself enqueues no collective: ``StepGraph`` functions are collective-free by contract,
    the all-reduce sits between the replayed front and the optimizer)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    from torch.distributed import distributed_c10d as c10d
    for pg in list(getattr(c10d._world, "pg_map", {}).keys()):
        if "nccl" not in str(dist.get_backend(pg)):          # (gloo has no watchdog list: nothing polls events)
            continue
        wait = getattr(pg, "_wait_for_pending_works", None)
        if wait is None:
            raise RuntimeError("StepGraph: this torch build cannot drain a process group's pending work before a stream capture "
                               "(ProcessGroup._wait_for_pending_works is missing); record the step before the first collective "
                               "of the run, or run it eagerly")
        wait()


class StepGraph:
    """A training (or evaluation) step recorded once per input shape as a HIP graph and replayed: the hot path's step is
    ~140 launches behind ~3 ms of Python; a replay is one ``hipGraphLaunch``.  ``fn(*tensors)`` must be a pure function of
    its tensor arguments and of state it updates in place on the device (parameters, optimizer moments: use
    ``FlatAdamW(capturable=True)``), free of host syncs, and must return a tensor or a tuple of tensors.

    Per distinct ``(shape, dtype)`` signature of the arguments: the first call runs ``fn`` eagerly (lazy initialisation,
    library workspaces), the second copies the arguments into static buffers, records ``fn`` on them and replays, later
    calls copy and replay.  The returned tensors are the recording's own outputs: consume them (stream-ordered) before
    the next call with the same signature.  At most ``max_shapes`` recordings are kept (each owns the memory pool of
    its intermediates); further signatures run eagerly.  ``optimizer.sync_hyper()`` is called before every replay.

    Eager calls and recordings run on ONE side stream owned by this object (ordered after / before the caller's stream):
    autograd's gradient accumulators remember the stream they were created on, and an accumulator made on the default
    stream pulls that stream into a later recording, which the runtime cannot finish (segfault in hipStreamEndCapture on
    this stack) -- with everything on the side stream the recording sees one stream only.  Accumulators made elsewhere
    stay alive as long as something holds the autograd graph of an earlier step: do not keep a step's loss (undetached)
    or other tensors with history across calls, and run every step of the run through this object (the layers of this
    package leave only detached values on the graph objects they are given, ``graph.leave_detached``).

    Second hazard of this stack (ROCm 7.2, measured): a blocking copy of 64 KB or more between host and device on the LEGACY
    DEFAULT stream (``tensor.cpu()``, ``torch.save`` of device tensors, ``tensor.to(device)``) between two recordings makes
    the later recording fault at replay (GPU memory access fault); the same copies on any other stream, pinned
    non-blocking copies, and copies after the last recording are harmless.  Run the loop that owns this object under
    ``with step_graph.on_stream():`` -- everything the loop does then happens on this object's side stream
    (``harness.fit`` does)."""

    MAX_SEEN = 1024      # signatures remembered as "seen once" (oldest forgotten first)

    def __init__(self, fn, optimizer=None, max_shapes=4):
        self.fn, self.optimizer, self.max_shapes = fn, optimizer, int(max_shapes)
        self._seen, self._graphs = OrderedDict(), {}
        self._stream = None
        self.replays = self.eager_calls = 0

    @staticmethod
    def _key(meta, tensors):
        return (meta,) + tuple((tuple(t.shape), t.dtype, t.device.index) for t in tensors)

    def _side_stream(self):
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        return self._stream

    @property
    def stream(self):
        """The side stream eager calls and recordings run on."""
        return self._side_stream()

    def on_stream(self):
        """Context manager: the enclosed code runs with this object's side stream as the current stream (ordered after the
        work already queued on the caller's stream, and the caller's stream waits for it on exit)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            side, cur = self._side_stream(), torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                yield side
            cur.wait_stream(side)
        return ctx()

    def _eager(self, meta, tensors):
        side, cur = self._side_stream(), torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = self.fn(*meta, *tensors)
        cur.wait_stream(side)
        return out

    def __call__(self, *args):
        """``args``: the tensors; a leading non-tensor (hashable: host-side sizes the step depends on) is part of the
        signature and passed through to ``fn`` as its first argument."""
        meta = ()
        if args and not torch.is_tensor(args[0]):
            meta, args = (args[0],), args[1:]
        tensors = args
        key = self._key(meta, tensors)
        rec = self._graphs.get(key)
        if rec is None:
            if key not in self._seen or len(self._graphs) >= self.max_shapes:
                self._seen[key] = True
                while len(self._seen) > self.MAX_SEEN:          # ragged data: every batch a new signature -- bounded memory
                    self._seen.popitem(last=False)
                self.eager_calls += 1
                return self._eager(meta, tensors)
            static = [torch.empty_like(t).copy_(t) for t in tensors]
            torch.cuda.synchronize()
            _drain_collectives()
            graph = torch.cuda.CUDAGraph()
            # with a process group alive its watchdog thread polls events while we record: only THIS thread's calls are
            # policed then (torch's "thread_local" capture mode); a single process keeps the strict default
            mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
            from . import _lib
            with _lib.capture_pins() as pins, torch.cuda.graph(graph, stream=self._side_stream(), capture_error_mode=mode):
                out = self.fn(*meta, *static)
            rec = self._graphs[key] = (graph, static, out, pins)     # memoised index arrays the recording reads: kept with it
        else:
            for s, t in zip(rec[1], tensors):
                if s.data_ptr() != t.data_ptr():
                    s.copy_(t)
        if self.optimizer is not None and hasattr(self.optimizer, "sync_hyper"):
            self.optimizer.sync_hyper()
        rec[0].replay()
        self.replays += 1
        return rec[2]
