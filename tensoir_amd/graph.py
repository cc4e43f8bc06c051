"""HIP-graph replay of the inference hot path for a fixed batch shape.

A step of ``Renderer_TensoIR_train`` is ~25 short kernel launches plus their buffer allocations; issued eagerly the
host needs ~0.8 ms per step, which the primary stage (0.25 ms of GPU work) cannot hide, and the two record-capacity
checks drain the queue.  ``GraphedRenderer`` captures the whole step once -- all launches go to the capture stream
through the same C ABI, all buffers live in the graph's private pool -- and replays it with one launch; the two
device-side record counters are read after the replay, and an overflow (more w > thres samples than the captured
capacity) re-captures with larger buffers.  Results are identical to the eager path.

Constraints (checked): inference only (no autograd, ``is_train=False``), ``sample_method='fixed_envirmap'`` (the
stratified direction jitter is a host-side RNG draw), fixed number of rays per call.
"""
from __future__ import annotations

import torch

from . import relight  # noqa: F401  (imported for its caches being warmed by the eager call)
from ._lib import TensoirHipError
from .renderer import Renderer_TensoIR_train


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
        self.captures = 0

    def _eager(self):
        with torch.no_grad():
            return Renderer_TensoIR_train(self.rays, None, self.lidx, self.model, **self.kw)

    def _capture(self):
        self._eager()                                     # learns the capacities, fills every cache, sets kernel attributes
        self._eager()                                     # and exercises the hinted (sync-free) route once
        torch.cuda.synchronize()
        shrink = self.__dict__.pop("_test_shrink_capacity", None)      # tests: capture with a capacity that is too small
        if shrink:
            for k in list(self.model._app_cap_hints):
                self.model._app_cap_hints[k] = shrink
        self.checks = []
        self.model.__dict__["_capture"] = self.checks
        g = torch.cuda.CUDAGraph()
        try:
            with torch.no_grad(), torch.cuda.graph(g):
                self.out = Renderer_TensoIR_train(self.rays, None, self.lidx, self.model, **self.kw)
        finally:
            self.model.__dict__.pop("_capture", None)
        self.graph = g
        self.captures += 1

    def _overflowed(self):
        if not self.checks:
            return False
        totals = torch.cat([t.reshape(1) for t, _, _ in self.checks]).tolist()       # the one host read per call
        bad = False
        for total, (_, cap, key) in zip(totals, self.checks):
            if total > cap:
                bad = True
                if key[0] == "primary":
                    self.model._app_cap_hints.pop((key[1], key[2]), None)             # relearnt by the next eager call
                else:
                    self.model._rec_cap_hints.pop(key[1], None)
        return bad

    def __call__(self, rays, light_idx):
        """rays [n_rays, 6], light_idx [n_rays, 1] (any device) -> the 12-key dict (fresh tensors)."""
        if rays.shape[0] != self.n_rays:
            raise ValueError(f"GraphedRenderer was built for {self.n_rays} rays, got {rays.shape[0]}")
        self.rays.copy_(rays.to(self.device, torch.float32), non_blocking=True)
        self.lidx.copy_(light_idx.to(self.device, torch.int32).view(-1, 1), non_blocking=True)
        for _ in range(3):
            if self.graph is None:
                self._capture()
            self.graph.replay()
            if not self._overflowed():
                return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.out.items()}
            self.graph = None                            # capacity too small for this batch: re-capture with room
        raise TensoirHipError("record capacity kept overflowing while re-capturing the HIP graph")

    def invalidate(self):
        """Call after the model's parameters / grid / mask change (the graph holds the packed shadows' addresses)."""
        self.graph = None
