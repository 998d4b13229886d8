"""clip_grad_norm_ + optimiser steps of a training step as two kernel launches.

The reference (src/engine/trainer.py:163-166,258-268) clips the encoder's and the decoder's gradient norms and
steps ``torch.optim.SGD`` / ``torch.optim.Adam`` objects its caller made (src/utils/solvers.py:6-52).  Callers
keep doing exactly that: ``NativeStep.build(groups)`` looks at the optimiser OBJECTS it is handed and, when
they are plain SGD / Adam with the options the reference uses (momentum, weight decay; no nesterov, dampening,
amsgrad, maximize, capturable, fused, differentiable; fp32 dense parameters on one HIP device), runs their
arithmetic in ``nasseg_optim_step`` (csrc/optim.hip) on the optimisers' OWN state - ``momentum_buffer``,
``exp_avg``, ``exp_avg_sq``, ``step`` entries with torch's names, shapes and devices, so ``state_dict()``,
``load_state_dict()`` and a later ``optim.step()`` by torch itself see what they expect.  Anything else
(another optimiser class, unusual options, sparse or non-fp32 parameters) makes ``build`` return None and the
caller uses torch's own implementation - same device, more launches.

Adam's step count lives in torch's per-parameter CPU ``state["step"]`` tensors AND in a device array the kernels
advance and read: a whole training step including the optimisers can be replayed from a hipGraph although the
optimisers are not ``capturable``.  The two are kept equal (``bump_host_steps`` / ``sync_steps``).

NASSEG_NATIVE_OPTIM=0 switches this off (A/B measurements).
"""
import ctypes
import os
import weakref

import torch

from .. import functional as F

ENABLED = os.environ.get("NASSEG_NATIVE_OPTIM", "1") != "0"
_COLS = 8


def _plain(value):
    return isinstance(value, (int, float)) and not isinstance(value, bool)


def _sgd_ok(group):
    return (_plain(group["lr"]) and _plain(group["momentum"]) and _plain(group["weight_decay"])
            and group.get("dampening", 0) == 0 and not group.get("nesterov", False)
            and not group.get("maximize", False) and not group.get("differentiable", False)
            and not group.get("fused", False))


def _adam_ok(group):
    return (_plain(group["lr"]) and _plain(group["weight_decay"]) and _plain(group["eps"])
            and all(_plain(b) for b in group["betas"]) and not group.get("amsgrad", False)
            and not group.get("maximize", False) and not group.get("capturable", False)
            and not group.get("differentiable", False) and not group.get("fused", False)
            and not group.get("decoupled_weight_decay", False))


def _hyper(kind, group):
    if kind == 0:
        return (0.0, float(group["lr"]), float(group["weight_decay"]), float(group["momentum"]), 0.0, 0.0)
    return (1.0, float(group["lr"]), float(group["weight_decay"]), float(group["betas"][0]),
            float(group["betas"][1]), float(group["eps"]))


class NativeStep(object):
    """groups: [(parameters, max_norm, optimiser)] as ``clip_and_step`` takes them (an optimiser may be None:
    its parameters are clipped only - torch's clip_grad_norm_ does that, so ``build`` declines)."""

    @classmethod
    def build(cls, groups):
        if not ENABLED:
            return None
        try:
            return cls(groups)
        except _Unsupported:
            return None

    def __init__(self, groups):
        self.chunk = int(F.lib.query("nasseg_optim_chunk"))
        # (param, weak reference to its optimiser, param_group, kind, clip set or -1, hyper group index).  The
        # optimisers are referenced WEAKLY: this object lives in a cache keyed by one of them (_CACHE), and a cached
        # value that owned its key would keep optimiser, parameters, gradients, state and the pinned tables below
        # alive for the life of the process - the search makes new optimisers per candidate and task
        # (src/main_search.py:575)
        self.entries = []
        self.clip_norms = []  # max_norm per clip set
        self.hyper_src = []  # (kind, param_group) per hyper group
        seen = set()
        device = None
        for params, max_norm, optim in groups:
            params = list(params)
            if optim is None or type(optim) not in (torch.optim.SGD, torch.optim.Adam):
                raise _Unsupported()
            kind = 0 if type(optim) is torch.optim.SGD else 1
            oref = weakref.ref(optim)
            clip = -1
            if max_norm > 0:
                clip = len(self.clip_norms)
                self.clip_norms.append(float(max_norm))
            owner = {}
            for gi, g in enumerate(optim.param_groups):
                if not (_sgd_ok(g) if kind == 0 else _adam_ok(g)):
                    raise _Unsupported()
                for p in g["params"]:
                    owner[id(p)] = g
            if set(owner) != set(id(p) for p in params):
                # (clipped and stepped parameter sets differ: torch's own functions keep their exact meaning)
                raise _Unsupported()
            hyper_of = {}
            for p in params:
                if id(p) in seen:
                    raise _Unsupported()
                seen.add(id(p))
                if not p.requires_grad:
                    continue
                if (p.dtype != torch.float32 or not p.is_cuda or p.is_sparse or not p.is_contiguous()
                        or (device is not None and p.device != device)):
                    raise _Unsupported()
                device = p.device
                g = owner[id(p)]
                if id(g) not in hyper_of:
                    hyper_of[id(g)] = len(self.hyper_src)
                    self.hyper_src.append((kind, g))
                self.entries.append((p, oref, g, kind, clip, hyper_of[id(g)]))
        if not self.entries or len(self.hyper_src) > 8 or len(self.clip_norms) > 8:
            raise _Unsupported()
        self.device = device
        self._slot = dict((id(e[0]), i) for i, e in enumerate(self.entries))
        self._optims = []  # (weak references, one per optimiser)
        for e in self.entries:
            if not any(e[1]() is o() for o in self._optims):  # (referents, not weakref objects)
                self._optims.append(e[1])
        n = len(self.entries)
        max_chunks = sum((e[0].numel() + self.chunk - 1) // self.chunk for e in self.entries)
        self.dstep = torch.zeros(n, device=device, dtype=torch.float32)
        # tables: pinned host sources + device twins, allocated ONCE (nothing is allocated when a step is recorded
        # into a hipGraph; the recorded copy nodes read the pinned sources again at every replay, so an object
        # whose step was captured is used by that graph only - engine/graphed.py builds its own)
        # (two pinned tensor tables, used in turn: a step that only moved its gradients fills the other one
        #  while the previous upload may still be in flight)
        self._host_tt = [torch.zeros(n, _COLS, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._tt_events = [None, None]
        self._cur = 0
        self._host_c = torch.zeros(max_chunks, 2, dtype=torch.int32).pin_memory()
        self._dev_t = torch.zeros(n, _COLS, dtype=torch.int64, device=device)
        self._dev_c = torch.zeros(max_chunks, 2, dtype=torch.int32, device=device)
        self._partial = torch.zeros(max_chunks, device=device, dtype=torch.float64)
        self._norms = torch.zeros(max(1, len(self.clip_norms)), device=device, dtype=torch.float32)
        self._hyper = (ctypes.c_double * (6 * len(self.hyper_src)))()
        self._clips = (ctypes.c_double * (3 * max(1, len(self.clip_norms))))()
        self._uploaded = None   # event behind the last upload of the chunk list
        self._key = None        # gradient addresses the tables were built for
        self._sig = None        # identity of the optimisers' state mappings at that time
        self._n_chunks = 0
        self._stepped = []
        self._host_steps_seen = [0.0] * n
        self._dirty_steps = True
        self.rebuilds = 0
        self.moves = 0

    # -- optimiser state, with torch's names ------------------------------------------------------
    def _state(self, entry):
        p, oref, group, kind = entry[:4]
        if kind == 0 and group["momentum"] == 0:
            return None, None  # (torch keeps no state for it either: optim.state stays without an entry)
        st = _alive(oref).state[p]
        if kind == 0:
            if st.get("momentum_buffer") is None:
                # torch clones d_p here on the first step; zeros + "buf = momentum * buf + d_p" is the same value
                st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            return st["momentum_buffer"], None
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self._dirty_steps = True
        if st["step"].is_cuda:
            raise _Unsupported()
        return st["exp_avg"], st["exp_avg_sq"]

    def _signature(self):
        """changes when an optimiser's state mapping was replaced (load_state_dict), emptied or extended, or its
        param_groups edited: the tables hold addresses of state tensors"""
        return tuple((id(o.state), len(o.state), len(o.param_groups), sum(len(g["params"]) for g in o.param_groups))
                     for o in (_alive(r) for r in self._optims))

    def sync_steps(self):
        """device step counters := the optimisers' ``state["step"]`` (after load_state_dict, after the state was
        put back behind a graph capture's warm-up, after torch itself stepped in between ...)"""
        host = []
        for e in self.entries:
            st = _alive(e[1]).state.get(e[0], {})
            host.append(float(st["step"]) if e[3] == 1 and "step" in st else 0.0)
        self.dstep.copy_(torch.tensor(host, dtype=torch.float32))
        self._host_steps_seen = host
        self._dirty_steps = False

    def host_steps_match(self):
        """one Adam parameter's CPU counter against what this object last left there (a load_state_dict, a foreign
        optim.step() or another NativeStep on the same optimisers moves it)"""
        for i, e in enumerate(self._stepped):
            if e[3] == 1:
                st = _alive(e[1]).state.get(e[0])
                return (st is not None and "step" in st
                        and float(st["step"]) == self._host_steps_seen[self._slot[id(e[0])]])
        return True

    def bump_host_steps(self):
        """the CPU step counters follow the device's (after a host-launched step, after a hipGraph replay)"""
        steps = [_alive(e[1]).state[e[0]]["step"] for e in self._stepped if e[3] == 1]
        if steps:
            torch._foreach_add_(steps, 1)
            for e in self._stepped:
                if e[3] == 1:
                    self._host_steps_seen[self._slot[id(e[0])]] += 1.0

    def hyper_values(self):
        """what a recorded step has baked in by value: every hyper group's numbers and the clip norms"""
        return (tuple(_hyper(kind, g) for kind, g in self.hyper_src), tuple(self.clip_norms))

    def prepare_capture(self):
        """before this object's step is recorded into a hipGraph: nothing of an earlier upload is in flight"""
        for ev in [self._uploaded] + self._tt_events:
            if ev is not None:
                ev.synchronize()
        self._uploaded = None
        self._tt_events = [None, None]

    def _next_table(self):
        """the pinned tensor table to fill now (numpy view): the one not used by the last upload, once its own
        last upload is through"""
        self._cur ^= 1
        ev = self._tt_events[self._cur]
        if ev is not None:
            ev.synchronize()
            self._tt_events[self._cur] = None
        return self._host_tt[self._cur].numpy()

    def _upload_table(self):
        if torch.cuda.is_current_stream_capturing():
            # no copy nodes in a recorded step (a graph of kernels only replays from pre-built packets, and the
            # tables of a recorded step never change: its gradients are static tensors): finish_capture uploads
            self._upload_pending = True
            return
        self._dev_t.copy_(self._host_tt[self._cur], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._tt_events[self._cur] = ev

    def finish_capture(self):
        """after this object's step was recorded into a hipGraph: the tables the recorded kernels read (tensor
        addresses, chunk list) go to the device now, once - the graph's owner uses this object for that graph only"""
        if getattr(self, "_upload_pending", False):
            self._upload_pending = False
            self._dev_t.copy_(self._host_tt[self._cur], non_blocking=True)
            self._dev_c.copy_(self._host_c, non_blocking=True)
            torch.cuda.current_stream().synchronize()

    def _move_gradients(self, key):
        """the same tensors got their gradients at other addresses (the usual host-launched step: backward's
        allocations do not repeat exactly): column 1 and the alignment flags of the table, nothing else"""
        import numpy as np

        for e in self._stepped:
            if not e[0].grad.is_contiguous():
                raise _Unsupported()
        prev = self._host_tt[self._cur].numpy()
        table = self._next_table()
        np.copyto(table, prev)
        ptrs = np.array(key, dtype=np.int64)
        table[:, 1] = ptrs
        table[:, 7] = self._static_aligned & (ptrs % 16 == 0)
        self._upload_table()
        self.moves += 1

    # -- tables -------------------------------------------------------------------------------------
    def _build_tables(self):
        live = []
        for e in self.entries:
            g = e[0].grad
            if g is None:
                continue
            if g.dtype != torch.float32 or g.is_sparse or not g.is_contiguous() or g.device != self.device:
                raise _Unsupported()
            s1, s2 = self._state(e)
            live.append((e, g, s1, s2))
        live.sort(key=lambda l: (l[0][4] < 0, l[0][4]))  # clip sets first, each one's chunks consecutive
        import numpy as np

        if self._uploaded is not None:
            self._uploaded.synchronize()  # (the previous upload may still be reading the pinned chunk list)
        table = self._next_table()
        chunks = self._host_c.numpy()
        table[:] = 0
        self._static_aligned = np.zeros(len(self.entries), dtype=np.int64)
        by_clip = {}
        n_chunks = 0
        for e, g, s1, s2 in live:
            p = e[0]
            ptrs = (p.data_ptr(), g.data_ptr(), s1.data_ptr() if s1 is not None else 0,
                    s2.data_ptr() if s2 is not None else 0)
            t = self._slot[id(p)]
            table[t] = ptrs + (p.numel(), e[4], e[5], 1 if all(a % 16 == 0 for a in ptrs) else 0)
            self._static_aligned[t] = 1 if all(a % 16 == 0 for a in (ptrs[0], ptrs[2], ptrs[3])) else 0
            k = (p.numel() + self.chunk - 1) // self.chunk
            chunks[n_chunks:n_chunks + k, 0] = t
            chunks[n_chunks:n_chunks + k, 1] = range(0, k * self.chunk, self.chunk)
            ent = by_clip.setdefault(e[4], [n_chunks, 0])
            ent[1] += k
            n_chunks += k
        # (a clip set none of whose tensors has a gradient: torch returns norm 0 for it - leave that to torch)
        if any(c not in by_clip for c in range(len(self.clip_norms))):
            raise _Unsupported()
        self._live_clips = [(mx,) + tuple(by_clip[c]) for c, mx in enumerate(self.clip_norms)]
        self._upload_table()
        if not torch.cuda.is_current_stream_capturing():
            self._dev_c.copy_(self._host_c, non_blocking=True)
            self._uploaded = torch.cuda.Event()
            self._uploaded.record()
        self._n_chunks = n_chunks
        self._stepped = [l[0] for l in live]
        # the state tensors stay referenced (their addresses are in the table whatever the caller drops); the
        # gradients do NOT: they are read by the launch that follows, in stream order, and holding them would keep
        # their memory from being handed out again - next step's gradients would land elsewhere, and the tables
        # would be rebuilt every step
        self._held = [l[2:] for l in live]
        self.rebuilds += 1

    def step(self):
        """clip + step every parameter that has a gradient; returns the clip sets' total norms (device tensor)"""
        key = tuple(0 if e[0].grad is None else e[0].grad.data_ptr() for e in self.entries)
        sig = self._signature()
        if sig == self._sig and self._key is not None and key != self._key and len(key) == len(self._key) \
                and all((a == 0) == (b == 0) for a, b in zip(key, self._key)):
            self._move_gradients(key)
            self._key = key
        elif key != self._key or sig != self._sig:
            self._build_tables()
            self._key, self._sig = key, self._signature()
        if not self._stepped:
            return self._norms
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing and (self._dirty_steps or not self.host_steps_match()):
            self.sync_steps()
        for i, (kind, g) in enumerate(self.hyper_src):
            self._hyper[6 * i:6 * i + 6] = _hyper(kind, g)
        for i, c in enumerate(self._live_clips):
            self._clips[3 * i:3 * i + 3] = c
        F.lib.call("nasseg_optim_step", self._dev_t.data_ptr(), len(self.entries), self._dev_c.data_ptr(),
                   self._n_chunks, ctypes.addressof(self._hyper), len(self.hyper_src), ctypes.addressof(self._clips),
                   len(self._live_clips), self.dstep.data_ptr(), self._partial.data_ptr(), self._norms.data_ptr(),
                   F.current_stream())
        if not capturing:
            self.bump_host_steps()
        return self._norms


class _Unsupported(RuntimeError):
    """(a RuntimeError: a hipGraph capture that meets it falls back to host launches, engine/trainer.py)"""


def _alive(oref):
    """the optimiser behind a NativeStep's weak reference (callers hand the optimisers in with every step, so a
    dead one means the stepper outlived its owner: torch's path then)"""
    optim = oref()
    if optim is None:
        raise _Unsupported()
    return optim


# steppers built by clip_and_step, per (first) optimiser object - kept OUTSIDE the optimiser (a weak mapping), so that
# copying or pickling an optimiser never meets device tensors and events of ours
_CACHE = weakref.WeakKeyDictionary()


def cached_stepper(optim):
    """the NativeStep clip_and_step last used with `optim` as its first optimiser (None: torch's implementations)"""
    ent = _CACHE.get(optim)
    return None if ent is None else ent[1]


def _find(groups):
    """the NativeStep cached for these (parameters, max_norm, optimiser) groups, built on first use; None when
    torch's implementations have to run"""
    first = next((o for _, _, o in groups if o is not None), None)
    if first is None or not ENABLED:
        return None
    params = [p if isinstance(p, (list, tuple)) else list(p) for p, _, _ in groups]
    # (what a cached stepper was built for: these optimiser objects, these parameter lists - first and last tensor and
    #  the count -, these clip norms, the optimisers' own groups as they are now)
    key = tuple((id(o), float(m), len(p), id(p[0]) if p else 0, id(p[-1]) if p else 0,
                 0 if o is None else len(o.param_groups),
                 0 if o is None else sum(len(g["params"]) for g in o.param_groups))
                for p, (_, m, o) in zip(params, groups))
    cached = _CACHE.get(first)
    if cached is not None and cached[0] == key:
        return cached[1]
    stepper = NativeStep.build([(p, m, o) for p, (_, m, o) in zip(params, groups)])
    _CACHE[first] = (key, stepper)
    return stepper


def native_clip_and_step(groups):
    """True when the step was taken by nasseg_optim_step"""
    stepper = _find(groups)
    if stepper is None:
        return False
    try:
        stepper.step()
    except _Unsupported:
        first = next(o for _, _, o in groups if o is not None)
        _CACHE[first] = (_CACHE[first][0], None)
        return False
    return True
