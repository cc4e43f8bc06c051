"""HIP-graph replay of the inference hot path for a fixed batch shape.

A step of ``Renderer_TensoIR_train`` is ~25 short kernel launches plus their buffer allocations; issued eagerly the
host needs ~0.8 ms per step, which the primary stage (0.25 ms of GPU work) cannot hide, and the two record-capacity
checks drain the queue.  ``GraphedRenderer`` captures the whole step once -- all launches go to the capture stream
through the same C ABI, all buffers live in the graph's private pool -- and replays it with one launch.  The last node
of the graph writes the two device-side record counters, their running maxima and a sticky overflow flag into pinned
host memory (tir_record_check), so NOTHING is launched between replays: the host waits for the stream and reads them
(or, with ``defer_check``, queues replays back to back and asks ``validate()`` once); an overflow (more w > thres
samples than the captured capacity) re-captures with room for the largest count seen.  Results are identical to the
eager path.

Several renderers of one model may be in flight at once (two batches on two HIP streams fill each other's kernel tails:
+10-14 % whole-job rate, `tools/two_stream_probe.py`): every renderer owns the device-side words its pass re-arms (pair /
record counters, last-workgroup tickets, the jitter RNG state) and installs them in the model while it runs eager calls
or captures.

Constraints (checked): inference only (no autograd, ``is_train=False``), ``sample_method='fixed_envirmap'`` (the
stratified direction jitter is a host-side RNG draw), fixed number of rays per call.
"""
from __future__ import annotations

import gc

import torch

from . import ops, relight  # noqa: F401  (relight: imported for its caches being warmed by the eager call)
from ._lib import TensoirHipError
from .renderer import Renderer_TensoIR_train


class _OwnState:
    """See GraphedRenderer._own_state.  (A plain module-level class: an object created per call must not sit in a reference
    cycle -- cyclic garbage that keeps a renderer, and with it a captured graph, alive until the collector runs could be
    freed in the middle of ANOTHER capture, and destroying a graph while a stream is capturing aborts the process.)"""
    __slots__ = ("r", "saved")

    def __init__(self, renderer):
        self.r, self.saved = renderer, None

    def __enter__(self):
        d = self.r.model.__dict__
        self.saved = {k: d.pop(k) for k in GraphedRenderer._OWN_KEYS if k in d}
        d.update(self.r._own)

    def __exit__(self, *exc):
        d = self.r.model.__dict__
        self.r._own = {k: d.pop(k) for k in GraphedRenderer._OWN_KEYS if k in d}
        d.update(self.saved)
        self.saved = None


def _clone_grouped(out):
    """Fresh copies of a replay's outputs with ONE copy per underlying buffer: the map columns are views of one [B, 20] block, so
    twelve outputs cost two or three copy launches instead of twelve (the boundary call hands out fresh tensors on every replay)."""
    bases, res = {}, {}
    for k, v in out.items():
        if not torch.is_tensor(v):
            res[k] = v
            continue
        st = v.untyped_storage()
        key = st.data_ptr()
        if key not in bases:
            nbytes = st.nbytes()
            if nbytes > 8 * v.numel() * v.element_size() + (1 << 16):     # a small view of a large buffer: copy the view only
                bases[key] = None
            else:
                bases[key] = torch.empty(0, dtype=torch.uint8, device=v.device).set_(st, 0, (nbytes,), (1,)).clone()
        b = bases[key]
        res[k] = v.clone() if b is None else torch.empty(0, dtype=v.dtype, device=v.device).set_(
            b.untyped_storage(), v.storage_offset(), v.shape, v.stride())
    return res


class GraphedRenderer:
    def __init__(self, tensoIR, n_rays, N_samples=-1, white_bg=True, is_relight=True, sample_method="fixed_envirmap",
                 args=None, device="cuda"):
        if sample_method != "fixed_envirmap":
            raise TensoirHipError("GraphedRenderer supports sample_method='fixed_envirmap' only")
        self.model, self.n_rays, self.device = tensoIR, int(n_rays), torch.device(device)
        self.kw = dict(N_samples=N_samples, white_bg=white_bg, is_train=False, is_relight=is_relight,
                       sample_method=sample_method, chunk_size=160000, device=device, args=args)
        self.rays = torch.zeros((self.n_rays, 6), dtype=torch.float32, device=self.device)
        self.lidx = torch.zeros((self.n_rays, 1), dtype=torch.int32, device=self.device)
        self.graph = None
        self.out = None
        self.checks = []
        # written by the last node of the graph (tir_record_check): this replay's record counters [0:4], their running
        # maxima over all replays [4:8], sticky overflow flag [8]; `_state` is the device side of maxima + flag
        self._host = torch.zeros((9,), dtype=torch.int64).pin_memory()             # allocated outside any capture
        self._state = torch.zeros((5,), dtype=torch.int64, device=self.device)
        self._own = {}                  # this renderer's device-side pass state (see _own_state)
        self._deferred = 0              # replays queued with defer_check=True since the last validate()
        self.captures = 0
        self._model_key = None          # what the captured descriptors were built from (see _stale)

    def _key(self):
        """Everything the by-value descriptors inside the graph point into: the field tables, the SGs, the decoder
        blobs and the marching constants.  packed_field() refreshes the field key as a side effect."""
        m = self.model
        m.packed_field()
        decs = [getattr(m, n) for n in ("renderModule", "renderModule_brdf", "renderModule_normal") if hasattr(m, n)]
        for d in decs:
            d.packed()
        # every light parameter: the general multi-light model keeps one SG set per light in a plain list
        lights = m.light_parameters()
        return (m._field_key, tuple((t.data_ptr(), t._version) for t in lights), tuple(d._key for d in decs),
                float(m.march_t_stop), ops.MLP_IMPL, ops.app_contraction(), ops.secondary_mlp_impl(), ops.secondary_app_impl(), ops.fused_indirect(),
                ops.INDIRECT_GUARD and (m.__dict__.get("_indirect_state") or {}).get("verdict"))

    def _stale(self):
        return self.graph is not None and self._key() != self._model_key

    _OWN_KEYS = ("_words", "_pair_counter", "_jit_rng")

    def _own_state(self):
        """Context manager: while it is open the model's per-pass device words are THIS renderer's (created by the
        model's own accessors on first use), so that another renderer's graph can replay concurrently."""
        return _OwnState(self)

    def _eager(self):
        with torch.no_grad(), self._own_state():
            return Renderer_TensoIR_train(self.rays, None, self.lidx, self.model, _no_graph=True, **self.kw)

    def _capture(self):
        self._eager()                                     # learns the capacities, fills every cache, sets kernel attributes
        self._eager()                                     # and exercises the hinted (sync-free) route once
        torch.cuda.synchronize()
        # never capture with less room than an earlier capture of this renderer had: a chunked image render walks
        # through light and heavy chunks, and the capacity should converge to the heaviest instead of following the last
        for (kind, *key), cap in self.__dict__.get("_cap_floor", {}).items():
            hints = self.model._app_cap_hints if kind == "primary" else self.model._rec_cap_hints
            k = tuple(key) if kind == "primary" else key[0]
            if k in hints:
                hints[k] = max(hints[k], cap)
        shrink = self.__dict__.pop("_test_shrink_capacity", None)      # tests: capture with a capacity that is too small
        if shrink:
            for k in list(self.model._app_cap_hints):
                self.model._app_cap_hints[k] = shrink
        self.checks = []
        self.model.__dict__["_capture"] = self.checks
        self._keepalive = []           # tensors outside the graph's pool whose addresses the captured launches bake in (ops.CAPTURE_KEEPALIVE)
        g = torch.cuda.CUDAGraph()
        # No garbage collection while the stream is capturing: the collector may free objects that own device resources
        # (another renderer's graph, an event, pinned memory), and their destructors call into the runtime -- not permitted
        # during a capture (the process aborts).  Collect what is pending now, then hold the collector off until the end.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            ops.CAPTURE_KEEPALIVE = self._keepalive
            with torch.no_grad(), self._own_state(), torch.cuda.graph(g):
                self.out = Renderer_TensoIR_train(self.rays, None, self.lidx, self.model, _no_graph=True, **self.kw)
                if self.checks:
                    # the record counters, their running maxima and a sticky overflow flag (never cleared by a replay)
                    # leave the device as part of the graph: one single-thread kernel writing pinned host memory.
                    # After a replay the host only waits for the stream and reads them; a caller may also queue many
                    # replays and ask later whether ANY of them overflowed, and how much room they need (validate()).
                    ops.record_check([t.reshape(1) for t, _, _ in self.checks], [c for _, c, _ in self.checks],
                                     self._state, self._host)
        finally:
            ops.CAPTURE_KEEPALIVE = None
            self.model.__dict__.pop("_capture", None)
            if gc_was_on:
                gc.enable()
        self.graph = g
        self._model_key = self._key()
        self.captures += 1
        if not shrink:
            floor = self.__dict__.setdefault("_cap_floor", {})
            for _, cap, key in self.checks:
                floor[key] = max(floor.get(key, 0), cap)

    def _overflowed(self):
        if not self.checks:
            return False
        torch.cuda.current_stream(self.device).synchronize()                         # the one host wait per call
        totals = self._host[:len(self.checks)].tolist()
        bad = False
        for total, (_, cap, key) in zip(totals, self.checks):
            if total > cap:
                bad = True
                self._need(key, total)
        return bad

    def _need(self, key, total):
        """A replay produced `total` records for `key`: the next capture gets room for it (and the eager call that
        precedes the capture relearns the hint)."""
        floor = self.__dict__.setdefault("_cap_floor", {})
        need = int(total * 1.25) + 4096
        if key[0] == "primary":
            need = min(need, key[1] * key[2])              # never more than rays x samples
        floor[key] = max(floor.get(key, 0), need)
        if key[0] == "primary":
            self.model._app_cap_hints.pop((key[1], key[2]), None)
        else:
            self.model._rec_cap_hints.pop(key[1], None)

    def validate(self):
        """Wait for the queued replays and report whether all of them stayed within the captured record capacities.
        False: the outputs of the deferred calls since the last validate() are not to be used -- render them again
        (the next call re-captures with room for the largest count seen)."""
        torch.cuda.current_stream(self.device).synchronize()
        self._deferred = 0
        if int(self._host[8]) == 0:
            return True
        maxima = self._host[4:4 + len(self.checks)].tolist()
        for total, (_, cap, key) in zip(maxima, self.checks):
            if total > cap:
                self._need(key, total)
        self._clear_sticky()
        self.graph = None
        return False

    def _clear_sticky(self):
        self._state.zero_()
        torch.cuda.current_stream(self.device).synchronize()
        self._host.zero_()

    def __call__(self, rays=None, light_idx=None, clone_outputs=True, defer_check=False):
        """rays [n_rays, 6], light_idx [n_rays, 1] (any device) -> the 12-key dict.

        The graph reads its inputs from the static buffers ``self.rays`` / ``self.lidx``: pass tensors to have them
        copied in, or fill the buffers yourself and pass nothing.  ``clone_outputs=False`` returns the graph's own
        output tensors (valid until the next call) instead of fresh copies -- a chunked image render that packs each
        chunk's records right away (``dist.render_sharded``) needs no copies.  ``defer_check=True`` does not wait for
        the replay: the capacity check is made by ``validate()`` later (a sticky device-side flag covers every replay
        queued in between), so back-to-back calls keep the GPU busy without a host round trip per step."""
        if rays is not None and rays is not self.rays:
            if rays.shape[0] != self.n_rays:
                raise ValueError(f"GraphedRenderer was built for {self.n_rays} rays, got {rays.shape[0]}")
            self.rays.copy_(rays.to(self.device, torch.float32), non_blocking=True)
        if light_idx is not None and light_idx is not self.lidx:
            self.lidx.copy_(light_idx.to(self.device, torch.int32).view(-1, 1), non_blocking=True)
        if self._stale():            # an optimizer step, upsample, shrink, new mask or decoder mode since the capture: the
            self.graph = None        # graph's descriptors point at freed / outdated tables -> capture again
        for _ in range(3):
            if self.graph is None:
                self._capture()
            self.graph.replay()
            if defer_check:
                self._deferred += 1
                return dict(self.out) if not clone_outputs else _clone_grouped(self.out)
            # the copies are queued BEFORE the host waits for the capacity check: nothing is launched behind the wait (a copy made
            # of an overflowed replay is simply dropped)
            res = _clone_grouped(self.out) if clone_outputs else dict(self.out)
            if not self._overflowed():
                return res
            self._clear_sticky()
            self.graph = None                            # capacity too small for this batch: re-capture with room
        raise TensoirHipError("record capacity kept overflowing while re-capturing the HIP graph")

    def invalidate(self):
        """Drop the captured graph (also done automatically when the model's parameters / grid / mask / decoder mode
        changed since the capture, see _stale)."""
        self.graph = None
