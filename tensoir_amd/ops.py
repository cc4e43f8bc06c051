"""torch-tensor front end of the C ABI: argument checking, output allocation, stream plumbing.

PyTorch is used for device memory and streams only; every arithmetic step of the hot path runs in
libtensoir_hip.so.  All functions require CUDA(HIP) tensors and raise otherwise -- no fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import TirEnvSG, TirField, TirFieldHalf, TirMlp, check, lib

MAP_STRIDE = 20

# Optional instrumentation used by bench.py: when STATS is a dict, the march wrappers accumulate the
# number of gathered density samples per kernel (device-side counter, read back by the caller), and
# when TIMING is a list every C call is bracketed by events on the launch stream.
STATS = None
TIMING = None

# Launch options (performance only; results are the same either way).  The library itself keeps no settable state and reads no
# environment variable: the options travel in the descriptors / arguments of each call, and THIS module is where
# their defaults come from -- the environment of the embedding process, read once at import:
#   TENSOIR_LDS_LINES=0   secondary march without the LDS-staged line factors        -> TirField.tune_lds_lines = 2
#   TIR_XCD=1             contiguous per-XCD work ranges in the gathers / the march   -> TirField.tune_xcd_order = 1
#   TENSOIR_MLP_GRID=n    persistent workgroups of a decoder launch (default 256)     -> TirMlp.tune_grid = n
#   TIR_PAIR_ORDER=m      point-major secondary pair list (default direction-major)   -> tir_shade_setup_compact(pair_order = 2)
TUNE = {
    "lds_lines": 2 if os.environ.get("TENSOIR_LDS_LINES", "1") == "0" else 0,
    "xcd_order": 1 if os.environ.get("TIR_XCD", "0") == "1" else 0,
    "mlp_grid": max(0, int(os.environ.get("TENSOIR_MLP_GRID", "0") or 0)),
    "pair_order": 2 if os.environ.get("TIR_PAIR_ORDER", "d")[:1] == "m" else 0,
    # TENSOIR_WGRAD_BLOCKS=n: workgroups of the fused weight-gradient launch (0 = 256: each fills a CU).  It runs on the leaf
    # stream beside the scatter / decoder-backward chain; measured (tools/train_bench.py, same box): 256 -> 4.1-4.3 ms per step,
    # 128 -> 4.6, 96 -> 4.6, 64 -> 5.2, 48 -> 7.0: with fewer workgroups the scatter kernels get faster (1.14 -> 0.92 ms) but
    # the leaf launch becomes the critical path
    "wgrad_blocks": max(0, int(os.environ.get("TENSOIR_WGRAD_BLOCKS", "0") or 0)),
}
# TENSOIR_MLP_AUXTAB=0: decoders whose aux input comes through an index map (the radiance decoder's view direction: one per ray
# or per light direction) run the full 150-input layer 1 instead of the aux-table variant (tir_mlp_aux_table + 9 k-blocks)
AUX_TABLE = os.environ.get("TENSOIR_MLP_AUXTAB", "1") != "0"
AUX_TABLE_MULTI = os.environ.get("TENSOIR_MLP_AUXTAB_MULTI", "0") == "1"


def mlp_aux_table(m: "PackedMlp", aux):
    """T[a][unit] = b0 + W0[:, aux columns] x(aux_a) for every row of `aux` [n_aux, 3] -> [n_aux, 128] fp32 (tir_mlp_aux_table):
    the start values of the layer-1 accumulators in the aux-table decoder launches."""
    aux = f32(aux, "aux", 3)
    table = torch.empty((aux.shape[0], 128), dtype=torch.float32, device=aux.device)
    _call("tir_mlp_aux_table", C.byref(m.desc), _ptr(aux), aux.shape[0], _ptr(table), _stream())
    return table



# While a HIP graph is being captured (GraphedRenderer._capture sets this to a list it keeps for the graph's lifetime) every
# cached table a captured launch reads is appended here: the graph bakes the table's raw address, so the tensor must outlive
# its cache entry (the cache holds at most four tables per decoder and evicts on the fifth aux tensor).
CAPTURE_KEEPALIVE = None


def _aux_table_cached(m: "PackedMlp", aux):
    """The per-aux-row layer-1 table of (decoder image, aux tensor version), computed once and kept with the packed decoder
    (which is rebuilt whenever the weights change).  An entry remembers the stream that computed it and an event behind that
    launch: a later user on ANOTHER stream waits for the event first (one model driven from several streams -- two batches in
    flight -- must not read a table whose kernel has not run yet)."""
    cache = m.__dict__.setdefault("_aux_tables", {})
    key = (aux.data_ptr(), aux._version, aux.shape[0])
    capturing = torch.cuda.is_current_stream_capturing()
    hit = cache.get(key)
    if hit is None:
        if capturing:
            # a miss inside a capture: the table is computed by a node of the graph itself and lives in the graph's pool
            # (recomputed per replay; not cached -- an eager call must never see pool memory)
            return mlp_aux_table(m, aux)
        if len(cache) >= 4:
            cache.clear()                # evicted tables stay alive wherever a captured graph holds them (CAPTURE_KEEPALIVE)
        table = mlp_aux_table(m, aux)
        ev = torch.cuda.Event()
        ev.record()
        # the aux tensor is held too: its address cannot be recycled for another tensor while the entry lives
        hit = cache[key] = (aux, table, ev, _raw_stream(torch.cuda.current_device()) if _raw_stream else None)
    elif not capturing and (_raw_stream is None or hit[3] != _raw_stream(torch.cuda.current_device())):
        torch.cuda.current_stream().wait_event(hit[2])
    # (capturing: GraphedRenderer drains the device after its eager warm-up passes and before the capture starts, so a cached
    #  table is complete; an event recorded outside the capture must not be waited on inside it)
    if capturing and CAPTURE_KEEPALIVE is not None:
        CAPTURE_KEEPALIVE.append(hit)
    return hit[1]


def mlp_rows_table(m: "PackedMlp", feat, table, n_dev=None, save_hidden=False):
    """The split-bf16 decoder with ONE table row per decoder row (tir_mlp_[train_]fwd_auxtab_bf16x3, aux_map = NULL): `table`
    [n, 128] carries everything of layer 1 that is not a function of the features -- bias, the aux columns and, for the
    residue-prediction normal decoder, the derived-normal columns.  -> out, or (out, h1, h2) with save_hidden."""
    if MLP_IMPL != "bf16x3":
        raise NotImplementedError("per-row layer-1 tables exist for the split-bf16 decoder only (TENSOIR_MLP=bf16x3)")
    feat = f32(feat, "feat")
    n = feat.shape[0]
    table = f32(table, "table", 128)
    if table.shape[0] != n:
        raise ValueError("table must have one row per feature row")
    out = torch.empty((n, m.out_dim), dtype=torch.float32, device=feat.device)
    if not save_hidden:
        _call("tir_mlp_fwd_auxtab_bf16x3", C.byref(m.desc), _ptr(feat), feat.shape[1], _ptr(table), None, 0, _ptr(out), n,
              _ptr(n_dev), _stream())
        return out
    h1 = torch.empty((n, 128), dtype=torch.float32, device=feat.device)
    h2 = torch.empty((n, 128), dtype=torch.float32, device=feat.device)
    _call("tir_mlp_train_fwd_auxtab_bf16x3", C.byref(m.desc), _ptr(feat), feat.shape[1], _ptr(table), None, 0, _ptr(out),
          _ptr(h1), _ptr(h2), n, _ptr(n_dev), _stream())
    return out, h1, h2


def _stats_ptr(name, dev):
    if STATS is None:
        return None
    if name not in STATS:
        STATS[name] = torch.zeros((1,), dtype=torch.int64, device=dev)
    return C.c_void_p(STATS[name].data_ptr())


def _call(name, *args):
    with _timed(name):
        rc = getattr(lib(), name)(*args)
    check(rc, name)


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if TIMING is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if TIMING is not None:
            self.e1.record()
            TIMING.append((self.name, self.e0, self.e1))
        return False


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t=None):
    """The current HIP stream of the current device as the void* the C ABI takes.  The raw accessor skips the Stream object
    torch.cuda.current_stream() builds per call (~20 calls per step: 10 us -> 1 us each on the host-bound eager path)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class AsyncCount:
    """A device-side int32 counter on its way to the host: the copy into pinned memory and an event are queued NOW, on the
    current stream, right behind the kernel that produced the counter.  `get()` waits for that event only -- not for the
    launches queued afterwards -- so a capacity check at the end of a pass does not drain the launch queue (with
    `tensor.item()` it does: the GPU then idles while the host queues the next stage)."""

    def __init__(self, counter):
        self.pin = torch.empty(1, dtype=torch.int32, pin_memory=True)
        self.pin.copy_(counter.view(-1)[:1], non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record(torch.cuda.current_stream(counter.device))      # the counter's device, which need not be current
        self.value = None

    def get(self) -> int:
        if self.value is None:
            self.ev.synchronize()
            self.value = int(self.pin[0])
            self.pin = None
        return self.value


class AsyncFloats:
    """A few device floats on their way to the host (AsyncCount for float tables): copy into pinned memory + event queued NOW.
    Pinned buffers and events are recycled (a training step creates one of these per parameter version)."""
    _free = []

    def __init__(self, t):
        n = t.numel()
        slot = next((k for k, (p, _) in enumerate(AsyncFloats._free) if p.numel() == n), None)
        self.pin, self.ev = AsyncFloats._free.pop(slot) if slot is not None else (torch.empty(n, dtype=torch.float32, pin_memory=True), torch.cuda.Event())
        self.pin.copy_(t.view(-1), non_blocking=True)
        self.ev.record(torch.cuda.current_stream(t.device))
        self.value = None

    def ready(self):
        return self.value is not None or self.ev.query()

    def get(self):
        if self.value is None:
            self.ev.synchronize()
            self.value = self.pin.tolist()
            if len(AsyncFloats._free) < 8:
                AsyncFloats._free.append((self.pin, self.ev))
            self.pin = self.ev = None
        return self.value


class HalfRange:
    """Range guard of one fp16 shadow of the field (tir_pack_half_checked's RANGE CONTRACT): the abs-maxima of the three
    appearance planes, the three lines, the light rows and basis_mat^T travel to the host behind the pack launch; ok() is
    True when no plane * line product, no plane * line * light-row product and no basis_mat element can leave the finite fp16
    range: bound = max_i (max|plane_i| max|line_i|) max(1, max|light row|) < INDIRECT_PROBE["range"]."""

    def __init__(self, absmax):
        self.pending = AsyncFloats(absmax)
        self.maxima = self.bound = self.result = None

    def ready(self):
        return self.result is not None or self.pending.ready()

    @staticmethod
    def judge(m, lim):
        """m = [max|plane_0..2|, max|line_0..2|, max|light rows|, max|basis_mat|] -> (ok, bound).  The packed-fp16 gather forms
        (plane x line) in fp16 BEFORE the light row is applied (h16_chunk_pk), so the bound carries max(1, light).  A NaN maximum
        fails every comparison."""
        bound = max(m[i] * m[3 + i] for i in range(3)) * max(1.0, m[6])
        return bool(bound < lim and m[7] < lim and all(v < lim for v in m[:7])), bound

    def ok(self):
        if self.result is None:
            m = self.pending.get()
            self.maxima = {"plane": m[0:3], "line": m[3:6], "light": m[6], "basis": m[7]}
            self.result, self.bound = HalfRange.judge(m, INDIRECT_PROBE["range"])
        return self.result


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req(t, dtype, name, last=None):
    if not torch.is_tensor(t):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise _lib.TensoirHipError(f"{name}: tensor must live on the GPU (got {t.device}); "
                                   "tensoir_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if last is not None and (t.dim() < 1 or t.shape[-1] != last):
        raise ValueError(f"{name}: last dim must be {last}, got {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def f32(t, name, last=None):
    return _req(t, torch.float32, name, last)


def i32(t, name):
    return _req(t, torch.int32, name)


def to_device(t, device, dtype=None):
    """`t.to(device, dtype)` that does not stall the launch queue: a host tensor goes through pinned memory and a
    non-blocking copy (a pageable-memory copy waits for everything queued on the stream before it returns)."""
    device = torch.device(device)
    if t.device.type == "cpu" and device.type == "cuda":
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous().pin_memory().to(device, non_blocking=True)
    return t.to(device) if dtype is None else t.to(device, dtype)


def record_check(counters, caps, state, host_out):
    """One kernel: device record counters -> pinned host_out (counts [0:4], running maxima [4:8], sticky overflow flag
    [8]); `state` = device int64[5] holding the maxima and the flag across launches."""
    n = len(counters)
    for t in counters:
        if t.dtype != torch.int32 or not t.is_cuda:
            raise _lib.TensoirHipError("record_check: counters must be int32 device tensors")
    if not host_out.is_pinned() or host_out.dtype != torch.int64 or host_out.numel() < 9:
        raise _lib.TensoirHipError("record_check: host_out must be a pinned int64[9] tensor")
    if state.dtype != torch.int64 or state.numel() < 5 or not state.is_cuda:
        raise _lib.TensoirHipError("record_check: state must be a device int64[5] tensor")
    ptrs = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in counters])
    cap_arr = (C.c_int64 * max(n, 1))(*[int(c) for c in caps])
    _call("tir_record_check", ptrs, cap_arr, n, _ptr(state), C.c_void_p(host_out.data_ptr()), _stream())


# ---- packing ------------------------------------------------------------------------------------
def pack_plane(src):
    """[1,C,H,W] (or [C,H,W]) -> channel-last [H,W,C]."""
    src = f32(src.detach(), "plane")
    c, h, w = src.shape[-3:]
    dst = torch.empty((h, w, c), dtype=torch.float32, device=src.device)
    _call("tir_pack_plane", _ptr(src), _ptr(dst), c, h, w, _stream())
    return dst


def pack_half(tables, scan=()):
    """fp16 copies (round to nearest even, SATURATING, same element order) of up to 8 fp32 tensors in ONE launch
    (tir_pack_half_checked) -> (copies, absmax): absmax[i] = max |x| of tables[i], then of every tensor in `scan` (tables whose
    range matters but which the kernels read as fp32: light rows, basis_mat) -- a device tensor, read by the range guard of the
    indirect-light precision policy (relight.HalfRange); a NaN anywhere in a table shows as NaN."""
    tables = [t.detach() for t in tables]
    scan = [t.detach() for t in scan]
    for t in tables + scan:
        if t.dtype != torch.float32 or not t.is_cuda:
            raise ValueError("pack_half: fp32 CUDA tensors expected")
    # same strides as the source (the channel-last planes keep their layout): a raw element-for-element copy of the storage
    outs = [torch.empty_strided(t.shape, t.stride(), dtype=torch.float16, device=t.device) for t in tables]
    k = len(tables) + len(scan)
    srcs = (C.c_void_p * k)(*[t.data_ptr() for t in tables + scan])
    dsts = (C.c_void_p * k)(*([o.data_ptr() for o in outs] + [None] * len(scan)))
    cnts = (C.c_int64 * k)(*[_storage_span(t) for t in tables + scan])
    absmax = torch.zeros((k,), dtype=torch.float32, device=(tables + scan)[0].device)
    _call("tir_pack_half_checked", srcs, dsts, cnts, k, _ptr(absmax), _stream())
    return outs, absmax


def _storage_span(t):
    """Elements between the first and the last element of a dense (possibly permuted) tensor."""
    n = 1 + sum((s - 1) * st for s, st in zip(t.shape, t.stride()))
    if n != t.numel():
        raise ValueError("pack_half: dense tensors expected")
    return n


def pack_occupancy(vol):
    """[.., D, H, W] float 0/1 volume -> (D+1)(H+1)(W+1) neighbourhood bytes."""
    vol = f32(vol.detach(), "alpha_volume")
    D, H, W = vol.shape[-3:]
    nbr = torch.empty(((D + 1) * (H + 1) * (W + 1),), dtype=torch.uint8, device=vol.device)
    _call("tir_pack_occupancy", _ptr(vol), _ptr(nbr), W, H, D, _stream())
    return nbr


def pack_basis(w):
    w = f32(w.detach(), "basis_mat.weight")
    app_dim, n_in = w.shape
    dst = torch.empty((n_in, 32), dtype=torch.float32, device=w.device)
    _call("tir_pack_basis", _ptr(w), _ptr(dst), app_dim, n_in, _stream())
    return dst


def light_mean(ll):
    ll = f32(ll.detach(), "light_line.weight")
    L, n = ll.shape
    out = torch.empty((n,), dtype=torch.float32, device=ll.device)
    _call("tir_light_mean", _ptr(ll), _ptr(out), L, n, _stream())
    return out


def pack_mlp(w0, b0, w1, b1, w2, b2, feat_dim, pe):
    ws = [f32(t.detach(), "mlp weight") for t in (w0, b0, w1, b1, w2, b2)]
    hidden, out_dim = w1.shape[0], w2.shape[0]
    n = lib().tir_mlp_packed_floats(feat_dim, pe, hidden, out_dim)
    if n < 0:
        check(int(n), f"tir_mlp_packed_floats(feat={feat_dim}, pe={pe}, hidden={hidden}, out={out_dim})")
    if tuple(w0.shape) != (hidden, feat_dim + 3 + 2 * pe * feat_dim + 2 * pe * 3):
        raise ValueError(f"mlp.0.weight has shape {tuple(w0.shape)}")
    packed = torch.empty((int(n),), dtype=torch.float32, device=w0.device)
    _call("tir_pack_mlp", *[_ptr(t) for t in ws], feat_dim, pe, hidden, out_dim, _ptr(packed),
                             _stream())
    return packed


class PackedMlp:
    def __init__(self, seq, feat_dim, pe, act, w0=None):
        """seq: the reference's nn.Sequential(Linear, ReLU, Linear, ReLU, Linear); w0: layer-1 weights in the kernels' column
        order [feat, aux, PE(feat), PE(aux)] when the module stores them differently (the residue-prediction normal decoder)."""
        self.packed = pack_mlp(seq[0].weight if w0 is None else w0, seq[0].bias, seq[2].weight, seq[2].bias,
                               seq[4].weight, seq[4].bias, feat_dim, pe)
        self.out_dim = seq[4].weight.shape[0]
        self.desc = TirMlp(self.packed.data_ptr(), feat_dim, pe, seq[2].weight.shape[0],
                           self.out_dim, act, TUNE["mlp_grid"])


# ---- field kernels --------------------------------------------------------------------------------
def vm_density(field: TirField, xyz, want_feat=True, want_sigma=False):
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    feat = torch.empty((n,), dtype=torch.float32, device=xyz.device) if want_feat else None
    sigma = torch.empty((n,), dtype=torch.float32, device=xyz.device) if want_sigma else None
    _call("tir_vm_density_fwd", C.byref(field), _ptr(xyz), _ptr(feat), _ptr(sigma), n, _stream())
    return feat, sigma


def occupancy_query(field: TirField, xyz):
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    hit = torch.empty((n,), dtype=torch.uint8, device=xyz.device)
    _call("tir_occupancy_query", C.byref(field), _ptr(xyz), _ptr(hit), n, _stream())
    return hit


def dense_alpha(field: TirField, grid_size, length):
    """getDenseAlpha: alpha [gx,gy,gz] on the lattice aabb0*(1-s)+aabb1*s, s = torch.linspace(0,1,g) per axis."""
    gx, gy, gz = [int(g) for g in grid_size]
    dev = torch.device("cuda", torch.cuda.current_device())
    lins = [torch.linspace(0, 1, g).to(dev).contiguous() for g in (gx, gy, gz)]
    alpha = torch.empty((gx, gy, gz), dtype=torch.float32, device=dev)
    _call("tir_dense_alpha", C.byref(field), _ptr(lins[0]), _ptr(lins[1]), _ptr(lins[2]), gx, gy, gz, float(length),
          _ptr(alpha), _stream())
    return alpha, lins


def alpha_pool(alpha, thres):
    """updateAlphaMask's pool + threshold: alpha [gx,gy,gz] -> (volume [gz,gy,gx] float 0/1, index bbox int32[6])."""
    alpha = f32(alpha, "alpha")
    gx, gy, gz = alpha.shape
    vol = torch.empty((gz, gy, gx), dtype=torch.float32, device=alpha.device)
    bbox = torch.tensor([2 ** 31 - 1] * 3 + [-1] * 3, dtype=torch.int32, device=alpha.device)
    _call("tir_alpha_pool", _ptr(alpha), gx, gy, gz, float(thres), _ptr(vol), _ptr(bbox), _stream())
    return vol, bbox


def filter_rays(field: TirField, rays, n_samples, bbox_only):
    rays = f32(rays, "rays", 6)
    n = rays.shape[0]
    mask = torch.empty((n,), dtype=torch.uint8, device=rays.device)
    _call("tir_filter_rays", C.byref(field), _ptr(rays), n, int(n_samples), int(bool(bbox_only)), _ptr(mask), _stream())
    return mask.bool()


def density_grad(field: TirField, xyz, want_sigma=False, want_grad=False, want_normal=True, n_dev=None):
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    mk = lambda on, *s: torch.empty(s, dtype=torch.float32, device=xyz.device) if on else None
    sigma, grad, normal = mk(want_sigma, n), mk(want_grad, n, 3), mk(want_normal, n, 3)
    _call("tir_density_grad_fwd", C.byref(field), _ptr(xyz), _ptr(sigma), _ptr(grad), _ptr(normal),
                                     n, _ptr(n_dev), _stream())
    return sigma, grad, normal


FEAT_STRIDE = 32     # feature rows are padded to 128 bytes (aligned float4 stores / loads)
# "mfma" (exact fp32 matrix-core contraction), "bf16x3" (split-bf16 matrix cores, parity grade) or "valu" (cross-check)
APP_ENTRY = {"mfma": "tir_vm_app_fwd", "x3": "tir_vm_app_fwd_x3", "bf16x3": "tir_vm_app_fwd_bf16x3", "valu": "tir_vm_app_fwd_valu"}
APP_IMPL = os.environ.get("TENSOIR_APP", "mfma")


def vm_app(field: TirField, xyz, light_idx=None, idx_map=None, want_rad=True, want_int=False, impl=None,
           idx_div=0, n_dev=None):
    """Returns padded feature buffers [n, FEAT_STRIDE]; columns >= app_dim are zero.
    n_dev (int32 device scalar): only the first min(n, n_dev) rows are computed (rest left uninitialised)."""
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    ad = FEAT_STRIDE
    if want_rad:
        if light_idx is None:
            raise ValueError("light_idx is required for the radiance feature")
        light_idx = i32(light_idx, "light_idx").view(-1)
        if idx_map is not None:
            idx_map = i32(idx_map, "idx_map").view(-1)
            if idx_map.numel() != n:
                raise ValueError("idx_map must have one entry per point")
        elif light_idx.numel() != n and idx_div <= 1:
            raise ValueError("light_idx must have one entry per point")
    rad = torch.empty((n, ad), dtype=torch.float32, device=xyz.device) if want_rad else None
    intr = torch.empty((n, ad), dtype=torch.float32, device=xyz.device) if want_int else None
    if impl is None:       # the default route follows the contraction setting (x3 unless the exact decoders / fp32 contraction are selected)
        impl = "x3" if (APP_IMPL == "mfma" and app_contraction() == "x3" and int(field.n_acomp) in (16, 24, 48)) else APP_IMPL
    _call(APP_ENTRY[impl], C.byref(field), _ptr(xyz),
          _ptr(light_idx) if want_rad else None, _ptr(idx_map) if want_rad else None, _ptr(rad), _ptr(intr),
          ad, int(idx_div), n, _ptr(n_dev), _stream())
    return rad, intr


def vm_app_h16(field: TirField, fh: TirFieldHalf, xyz, light_idx, idx_map=None, idx_div=0, n_dev=None):
    """Radiance features [n, FEAT_STRIDE] from the fp16 shadow of the appearance planes / lines (tir_vm_app_fwd_h16): the
    indirect-light precision policy's gather (48 appearance components per plane only)."""
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    light_idx = i32(light_idx, "light_idx").view(-1)
    if idx_map is not None:
        idx_map = i32(idx_map, "idx_map").view(-1)
        if idx_map.numel() != n:
            raise ValueError("idx_map must have one entry per point")
    elif light_idx.numel() != n and idx_div <= 1:
        raise ValueError("light_idx must have one entry per point")
    rad = torch.empty((n, FEAT_STRIDE), dtype=torch.float32, device=xyz.device)
    _call("tir_vm_app_fwd_h16", C.byref(field), C.byref(fh), _ptr(xyz), _ptr(light_idx), _ptr(idx_map), _ptr(rad), FEAT_STRIDE,
          int(idx_div), n, _ptr(n_dev), _stream())
    return rad


def indirect_fused(field: TirField, fh: TirFieldHalf, m: "PackedMlp", xyz, light_idx, rec_map, idx_div, dirs, n_dirs, n_dev=None):
    """Radiance [n, out_dim] of the secondary-ray records in ONE launch (tir_indirect_fused_fwd): fp16-shadow appearance gather,
    basis_mat contraction and the radiance decoder, the feature rows staying in registers.  rec_map[s] = pair id of record s:
    light index = light_idx[pair // idx_div], view direction = dirs[pair % n_dirs] (through the cached aux table)."""
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    light_idx = i32(light_idx, "light_idx").view(-1)
    rec_map = i32(rec_map, "rec_map").view(-1)
    if rec_map.numel() != n:
        raise ValueError("rec_map must have one entry per record")
    dirs = f32(dirs, "dirs", 3)
    table = _aux_table_cached(m, dirs)
    out = torch.empty((n, m.out_dim), dtype=torch.float32, device=xyz.device)
    _call("tir_indirect_fused_fwd", C.byref(field), C.byref(fh), C.byref(m.desc), _ptr(xyz), _ptr(light_idx), _ptr(rec_map), int(idx_div),
          int(n_dirs), _ptr(table), _ptr(out), n, _ptr(n_dev), _stream())
    return out


def indirect_fused_hp(field: TirField, m: "PackedMlp", xyz, light_idx, rec_map, idx_div, dirs, n_dirs, n_dev=None):
    """indirect_fused at the precision a trained checkpoint needs (tir_indirect_fused_hp_fwd): fp32 taps, fp16 hi + lo contraction,
    decoder weights as fp16 + fp8 residue -- the auto policy's first fallback, one launch instead of gather + decoder with the
    feature rows through HBM."""
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    light_idx = i32(light_idx, "light_idx").view(-1)
    rec_map = i32(rec_map, "rec_map").view(-1)
    if rec_map.numel() != n:
        raise ValueError("rec_map must have one entry per record")
    dirs = f32(dirs, "dirs", 3)
    table = _aux_table_cached(m, dirs)
    out = torch.empty((n, m.out_dim), dtype=torch.float32, device=xyz.device)
    _call("tir_indirect_fused_hp_fwd", C.byref(field), C.byref(m.desc), _ptr(xyz), _ptr(light_idx), _ptr(rec_map), int(idx_div),
          int(n_dirs), _ptr(table), _ptr(out), n, _ptr(n_dev), _stream())
    return out


# decoder implementations: "mfma" = exact fp32 matrix cores, "bf16x3" = split-bf16 matrix cores (parity
# grade, ~5x fewer MFMA cycles), "bf16" = single-product reduced precision, "valu" = cross-check kernel
MLP_ENTRY = {"mfma": "tir_mlp_fwd", "bf16x3": "tir_mlp_fwd_bf16x3", "bf16": "tir_mlp_fwd_bf16",
             "valu": "tir_mlp_fwd_valu"}
MLP_IMPL = os.environ.get("TENSOIR_DECODER", "bf16x3")
if MLP_IMPL not in MLP_ENTRY:
    raise ValueError(f"TENSOIR_DECODER={MLP_IMPL!r}: expected one of {sorted(MLP_ENTRY)}")
# Precision policy for INDIRECT light.  The radiance of the SECONDARY-ray records (models/relight_utils.py:818-832)
# is averaged over a ray's records and over the light directions before it reaches rgb_with_brdf_map; its gather and decoder may
# run at another precision than the launches whose outputs are composited into the maps directly:
#   TENSOIR_INDIRECT_PRECISION = f16: appearance taps from an fp16 shadow of the planes / lines (tir_vm_app_fwd_h16)
#                                  and the single-product fp16 decoder (tir_mlp_fwd_auxtab_f16), fp32 accumulation everywhere;
#                                  measured on rgb_with_brdf_map: profiles/r04_precision_policy.json
#                              = full: the same kernels as the primary stage (fp32 taps, split-bf16 x3 decoder)
#                              = auto (default, round 5): the f16 kernels, but only for field / decoder versions that pass (i) the
#                                  RANGE guard -- max|plane_i| max|line_i| max|light row| and max|basis_mat| from the pack launch
#                                  bound every fp16 product below 6e4, checked for every new parameter version without an extra
#                                  host synchronisation -- and (ii) the SELF-CHECK probe: up to INDIRECT_PROBE["records"] of the
#                                  pass's own records are decoded by both paths and the f16 path is kept only while the signed
#                                  mean / rms / max difference stay inside INDIRECT_PROBE's limits; re-probed whenever parameter
#                                  storage changes (load, upsample, shrink) and every INDIRECT_PROBE["interval"] parameter
#                                  versions otherwise (optimizer steps).  Anything else falls back to `full` for that version
#                                  (relight._indirect_mode; the verdict is kept with the model and written into checkpoints).
# Applies only while MLP_IMPL is the split-bf16 default (the exact / cross-check decoder modes stay exact end to end).
#                              = hp (round 6): the high-precision fused kernel unconditionally (tir_indirect_fused_hp_fwd: fp32 taps,
#                                  decoder weights as fp16 + fp8 residue).  Under `auto` it is the FIRST fallback: a version whose
#                                  self-check rejects the f16 kernels is checked the same way with the hp kernel (both against the
#                                  full kernels) and only goes to `full` when that fails too -- a field trained to 300^3 takes
#                                  this route (profiles/r06_precision_trained_300.json).  TENSOIR_INDIRECT_HP=0 removes the tier.
_IND = os.environ.get("TENSOIR_INDIRECT_PRECISION", "auto")
if _IND not in ("auto", "f16", "hp", "full"):
    raise ValueError(f"TENSOIR_INDIRECT_PRECISION={_IND!r}: expected auto, f16, hp or full")
SECONDARY_MLP_IMPL = {"full": None, "hp": "hp"}.get(_IND, "f16")       # None | "f16" | "hp" | "bf16" (probe only) | "bf16x3"
SECONDARY_APP_IMPL = "h16" if _IND in ("auto", "f16") else None        # None | "h16"
INDIRECT_GUARD = _IND == "auto"        # False: the settings above apply unconditionally (f16 / hp: the caller vouches for range and precision)
INDIRECT_HP = os.environ.get("TENSOIR_INDIRECT_HP", "1") != "0"        # the hp tier of the auto policy
# The self-check.  Through render_with_BRDF / Renderer_TensoIR_train (relight.shade_from_maps) it is a MEASUREMENT of the
# quantity the tolerance is stated on: all secondary-ray records of the pass are decoded by both paths, the integration kernel
# renders rgb_with_brdf_map from both, and the f16 kernels are kept while
#     max over the pass's rays of |rgb_with_brdf_map(f16) - rgb_with_brdf_map(full)|  <=  map_limit  (2.5e-5)
# -- a quarter of the 1e-4 budget; other batches of the same parameters can be worse than the probed one, measured up to 2x
# (profiles/r05_precision_trained.json), which leaves the policy's contribution under half of the budget.
# The bare compute_radiance / compute_secondary_shading_effects entry points have no map to measure: there
# (relight._probe_indirect) an evenly strided subset of the records is decoded by both paths and the map error is ESTIMATED from
# the signed mean ("bias", max over the colour channels), the rms and the max of the difference:
#     max(w_bias * bias + w_rms * rms, w_max * max) <= limit
# calibrated on the scaling sweep and the trained checkpoint of tests/precision_cases.py, where the measured map error was
# 0.45 bias + 0.2 rms within 15 % on the smooth scenes and 0.2 max on the trained one (profiles/r05_precision_sweep.json: 3.3e-6 as
# initialised, 3.0e-5 with the radiance decoder's weights doubled, 2.0e-4 with x4 -- unguarded fp16 leaves the budget there).
# range: the largest |product| the range guard accepts (largest finite fp16 = 65504).
# TENSOIR_INDIRECT_MAP_LIMIT overrides map_limit (a trained analytic scene measured 2.0e-5 ... 3.1e-5 on its own training rays:
# around the default, so such a checkpoint may run either way; both are within the budget).
# train_map_limit: what a TRAINING forward accepts (is_train renders feed the loss only, and the secondary stage is a no_grad
# constant there, models/relight_utils.py:344): the contract's tolerance itself.  A verdict taken with it never serves an
# inference pass (relight._indirect_mode re-probes with the strict limit).
INDIRECT_PROBE = {"map_limit": float(os.environ.get("TENSOIR_INDIRECT_MAP_LIMIT", "2.5e-5")), "train_map_limit": 1.0e-4, "records": 32768, "interval": 64, "w_bias": 0.5, "w_rms": 0.25, "w_max": 0.25, "limit": 2.5e-5,
                  "range": 6.0e4}


# TENSOIR_FUSED_INDIRECT=0: gather and decoder of the secondary-ray records as two launches (tir_vm_app_fwd_h16 +
# tir_mlp_fwd_auxtab_f16, features through HBM) instead of the fused kernel (tir_indirect_fused_fwd).  Only meaningful under the f16 policy.
FUSED_INDIRECT = os.environ.get("TENSOIR_FUSED_INDIRECT", "1") != "0"


def fused_indirect():
    return FUSED_INDIRECT and AUX_TABLE and secondary_app_impl() == "h16" and secondary_mlp_impl() == "f16"


def full_indirect_route():
    """Which launches decode the secondary-ray records when the indirect-light policy says `full` (reported by bench.py)."""
    gather = "tir_vm_app_fwd_x3 (fp32 taps, fp16 hi + lo contraction)" if app_contraction() == "x3" else "tir_vm_app_fwd (fp32 taps, fp32 MFMA contraction)"
    return gather + " + tir_mlp_fwd_auxtab_bf16x3 (split-bf16 x3), feature rows through HBM"


def hp_indirect_route():
    return ("tir_indirect_fused_hp_fwd: fp32 taps, fp16 hi + lo basis contraction, decoder with fp16 activations and fp16 + fp8-residue "
            "weights (residue on the block-scaled fp8 matrix instruction), one launch, feature rows in registers")


def secondary_app_impl():
    """Gather mode of the secondary-record radiance features under the current settings."""
    return SECONDARY_APP_IMPL if (MLP_IMPL == "bf16x3" and SECONDARY_APP_IMPL) else None


def secondary_mlp_impl():
    """Decoder mode of the secondary-record radiance launch under the current settings."""
    return SECONDARY_MLP_IMPL if (MLP_IMPL == "bf16x3" and SECONDARY_MLP_IMPL) else None


def mlp_multi(jobs, n_dev=None, save_hidden=False):
    """Several decoders over the same rows in one launch (tir_mlp_fwd_multi_bf16x3).  jobs: list of up to four
    (PackedMlp, feat [n, FEAT_STRIDE], aux [*, 3], aux_map or None); returns the list of outputs [n, out_dim] -- with
    save_hidden (training forward, tir_mlp_train_fwd_multi_bf16x3) the list of (out, h1 [n,128], h2 [n,128])."""
    n = jobs[0][1].shape[0]
    k = len(jobs)
    feats, auxs, maps, outs = [], [], [], []
    for m, feat, aux, aux_map in jobs:
        feat = f32(feat, "feat")
        if feat.shape[0] != n or feat.dim() != 2 or feat.shape[1] != FEAT_STRIDE:
            raise ValueError(f"mlp_multi: every job needs [n, {FEAT_STRIDE}] feature rows")
        aux = f32(aux, "aux", 3)
        if aux_map is not None:
            aux_map = i32(aux_map, "aux_map").view(-1)
        elif aux.shape[0] != n:
            raise ValueError("aux must have one row per feature row (or pass aux_map)")
        feats.append(feat); auxs.append(aux); maps.append(aux_map)
        outs.append(torch.empty((n, m.out_dim), dtype=torch.float32, device=feat.device))
    arr = lambda ts: (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
    descs = (C.POINTER(TirMlp) * k)(*[C.pointer(m.desc) for m, _, _, _ in jobs])
    # Off by default in the merged primary-stage launch: one job in four would save 12 of its 216 MFMAs per tile (~2 us at
    # 230 k rows) while its per-ray table costs a 9 us launch of its own.  AUX_TABLE_MULTI = True selects it (tests do).
    use_tab = [AUX_TABLE and AUX_TABLE_MULTI and not save_hidden and mp is not None and aux.shape[0] * 8 <= max(n, 1)
               for aux, mp in zip(auxs, maps)]
    if any(use_tab):
        # jobs whose aux rows come through an index map with few distinct rows (same rule as mlp()): aux-table variant of layer 1
        tables = [mlp_aux_table(m, aux) if t else None for (m, _, _, _), aux, t in zip(jobs, auxs, use_tab)]
        _call("tir_mlp_fwd_multi_auxtab_bf16x3", descs, arr(feats), FEAT_STRIDE, arr(auxs), arr(maps), arr(tables), arr(outs), k, n,
              _ptr(n_dev), _stream())
        return outs
    if save_hidden:
        h1s = [torch.empty((n, 128), dtype=torch.float32, device=outs[0].device) for _ in range(k)]
        h2s = [torch.empty((n, 128), dtype=torch.float32, device=outs[0].device) for _ in range(k)]
        _call("tir_mlp_train_fwd_multi_bf16x3", descs, arr(feats), FEAT_STRIDE, arr(auxs), arr(maps), arr(outs), arr(h1s), arr(h2s),
              k, n, _ptr(n_dev), _stream())
        return list(zip(outs, h1s, h2s))
    _call("tir_mlp_fwd_multi_bf16x3", descs, arr(feats), FEAT_STRIDE, arr(auxs), arr(maps), arr(outs), k, n, _ptr(n_dev),
          _stream())
    return outs


def mlp(m: PackedMlp, feat, aux, aux_map=None, impl=None, aux_mod=0, n_dev=None):
    impl = impl or MLP_IMPL
    feat = f32(feat, "feat")
    if feat.dim() != 2 or feat.shape[1] < m.desc.feat_dim:
        raise ValueError(f"feat: expected [n, >= {m.desc.feat_dim}], got {tuple(feat.shape)}")
    aux = f32(aux, "aux", 3)
    n = feat.shape[0]
    if aux_map is not None:
        aux_map = i32(aux_map, "aux_map").view(-1)
        if aux_map.numel() != n:
            raise ValueError("aux_map must have one entry per row")
    elif aux.shape[0] != n and aux_mod <= 0:
        raise ValueError("aux must have one row per feature row")
    out = torch.empty((n, m.out_dim), dtype=torch.float32, device=feat.device)
    f16 = impl == "f16"
    if f16:
        impl = "bf16x3"              # rows that do not qualify for the aux-table launch below take the parity-grade kernel
    if impl == "bf16x3" and AUX_TABLE and (aux_map is not None or aux_mod > 0) and aux.shape[0] * 8 <= max(n, 1):
        # few distinct aux rows (one per ray / light direction) for many decoder rows: their 15 input columns + the bias as a
        # per-aux-row start value of the layer-1 accumulators, 9 instead of 10 k-blocks of matrix work per row
        # the light-direction grid of a scene is a persistent tensor: its table is computed once per (decoder image, aux tensor
        # version) and kept with the packed decoder (which is rebuilt whenever the weights change)
        table = _aux_table_cached(m, aux)
        _call("tir_mlp_fwd_auxtab_f16" if f16 else "tir_mlp_fwd_auxtab_bf16x3", C.byref(m.desc), _ptr(feat), feat.shape[1], _ptr(table),
              _ptr(aux_map), int(aux_mod), _ptr(out), n, _ptr(n_dev), _stream())
        return out
    _call(MLP_ENTRY[impl], C.byref(m.desc), _ptr(feat), feat.shape[1], _ptr(aux),
          _ptr(aux_map), int(aux_mod), _ptr(out), n, _ptr(n_dev), _stream())
    return out


# ---- primary march --------------------------------------------------------------------------------
def march_primary(field: TirField, rays, ray_jitter, n_samples, t_stop):
    rays = f32(rays, "rays", 6)
    B = rays.shape[0]
    dev = rays.device
    if ray_jitter is not None:
        ray_jitter = f32(ray_jitter, "ray_jitter").view(-1)
        if ray_jitter.numel() != B:
            raise ValueError("ray_jitter must be [B] / [B,1]")
    weight = torch.empty((B, n_samples), dtype=torch.float32, device=dev)
    acc = torch.empty((B,), dtype=torch.float32, device=dev)
    depth = torch.empty((B,), dtype=torch.float32, device=dev)
    tend = torch.empty((B,), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    _call("tir_march_primary_fwd", C.byref(field), _ptr(rays), _ptr(ray_jitter), B, n_samples,
                                      float(t_stop), _ptr(weight), _ptr(acc), _ptr(depth), _ptr(tend),
                                      _ptr(cnt), _stats_ptr("tir_march_primary_fwd", dev), _stream())
    return weight, acc, depth, tend, cnt


def march_primary_fused(field: TirField, rays, n_samples, t_stop, cap, words):
    """Inference-only primary march that also emits the view-direction table, re-arms the pass counters and scans the
    record counts (tir_march_primary_fused_fwd).  words: persistent int32[8] of the model -- [1:3] secondary record
    counter (zeroed here), [3] scan ticket, [5] record total.  Returns weight, acc, depth, cnt, offsets, total, viewdirs."""
    rays = f32(rays, "rays", 6)
    B = rays.shape[0]
    dev = rays.device
    weight = torch.empty((B, n_samples), dtype=torch.float32, device=dev)
    acc = torch.empty((B,), dtype=torch.float32, device=dev)
    depth = torch.empty((B,), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    off = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    vd = torch.empty((B, 3), dtype=torch.float32, device=dev)
    total = words[5:6]
    _call("tir_march_primary_fused_fwd", C.byref(field), _ptr(rays), None, B, n_samples, float(t_stop), _ptr(weight),
          _ptr(acc), _ptr(depth), None, _ptr(cnt), _stats_ptr("tir_march_primary_fwd", dev), _ptr(vd),
          C.c_void_p(words.data_ptr() + 4), 2, C.c_void_p(words.data_ptr() + 12), _ptr(off), int(cap), _ptr(total), _stream())
    return weight, acc, depth, cnt, off, total, vd


def composite_primary_fused(rays, offsets, rec_w, rgb, brdf, brdf_jit, pred_n, der_n, acc, depth, white_bg, is_relight,
                            fixed_fresnel, words, rng_state=None, rng_step=0):
    """tir_composite_primary_fused: map rows + the two smoothness means (float32[2]); words[4] is the ticket."""
    rays = f32(rays, "rays", 6)
    B = rays.shape[0]
    out = torch.empty((B, MAP_STRIDE), dtype=torch.float32, device=rays.device)
    smooth = torch.zeros((2,), dtype=torch.float32, device=rays.device) if B == 0 else \
        torch.empty((2,), dtype=torch.float32, device=rays.device)
    _call("tir_composite_primary_fused", _ptr(rays), _ptr(offsets), _ptr(rec_w), _ptr(rgb), _ptr(brdf), _ptr(brdf_jit),
          _ptr(pred_n), _ptr(der_n), _ptr(acc), _ptr(depth), B, int(bool(white_bg)), int(bool(is_relight)),
          float(fixed_fresnel), _ptr(out), C.c_void_p(words.data_ptr() + 16), _ptr(smooth), _ptr(rng_state), int(rng_step),
          _stream())
    return out, smooth


def vm_app_jitter(field: TirField, xyz, scale, seed, offset, rng_state=None, n_dev=None):
    """Intrinsic features of xyz + scale * N(0,1), noise drawn in the kernel (tir_vm_app_jitter_fwd).
    Returns (xyz_jittered [n,3], int_feat [n, FEAT_STRIDE])."""
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    xyz_j = torch.empty_like(xyz)
    intr = torch.empty((n, FEAT_STRIDE), dtype=torch.float32, device=xyz.device)
    _call("tir_vm_app_jitter_fwd", C.byref(field), _ptr(xyz), n, _ptr(n_dev), float(scale), int(seed) & (2 ** 64 - 1),
          int(offset) & (2 ** 64 - 1), _ptr(rng_state), _ptr(xyz_j), _ptr(intr), FEAT_STRIDE, _stream())
    return xyz_j, intr


# Contraction of the primary stage's merged gather: "fp32" = the exact-fp32 matrix instruction (tir_vm_app_primary_fwd, the default),
# "x3" = fp16 hi + lo operands, three matrix products (tir_vm_app_primary_x3_fwd / tir_vm_app_fwd_x3).  Round 6 built x3 expecting
# the exact instruction (vector rate) to bind the launch; measured, the launch went from 99 to 96 us only, and on a checkpoint
# trained to 300^3 the albedo / roughness maps moved to 1.0e-4 / 8.5e-5 from the oracle (2e-6 with fp32): the fp16 RESIDUE of a
# product below 0.125 is a subnormal, so the features are good to ~1e-6 of their scale, not 2^-21, and the BRDF decoder of a trained
# field amplifies that ~100-fold (profiles/r06m_precision_trained_300_x3_gather.json).  Opt-in only.
APP_CONTRACTION = os.environ.get("TENSOIR_APP_CONTRACTION", "fp32")
if APP_CONTRACTION not in ("x3", "fp32"):
    raise ValueError(f"TENSOIR_APP_CONTRACTION={APP_CONTRACTION!r}: expected x3 or fp32")


def app_contraction():
    return APP_CONTRACTION if MLP_IMPL == "bf16x3" else "fp32"


def vm_app_primary(field: TirField, xyz, light_idx, idx_map, scale, rng_state, n_dev=None, exact=False):
    """The primary stage's two appearance gathers in one launch (tir_vm_app_primary_x3_fwd / tir_vm_app_primary_fwd): returns
    (rad [n, FEAT_STRIDE], intr, xyz_jittered [n, 3], intr_jittered).  exact: the fp32 contraction whatever the setting (the
    training forward: its saved features feed the hand-written backward)."""
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    dev = xyz.device
    rad = torch.empty((n, FEAT_STRIDE), dtype=torch.float32, device=dev)
    intr = torch.empty((n, FEAT_STRIDE), dtype=torch.float32, device=dev)
    intr_j = torch.empty((n, FEAT_STRIDE), dtype=torch.float32, device=dev)
    xyz_j = torch.empty_like(xyz)
    _call("tir_vm_app_primary_fwd" if (exact or app_contraction() == "fp32") else "tir_vm_app_primary_x3_fwd", C.byref(field), _ptr(xyz),
          _ptr(i32(light_idx, "light_idx").view(-1)),
          _ptr(None if idx_map is None else i32(idx_map, "idx_map").view(-1)), _ptr(rad), _ptr(intr), FEAT_STRIDE, n, _ptr(n_dev),
          float(scale), 0, 0, _ptr(rng_state), _ptr(xyz_j), _ptr(intr_j), _stream())
    return rad, intr, xyz_j, intr_j


def exclusive_scan(counts):
    counts = i32(counts, "counts").view(-1)
    n = counts.numel()
    off = torch.empty((n + 1,), dtype=torch.int32, device=counts.device)
    _call("tir_exclusive_scan", _ptr(counts), _ptr(off), n, _stream())
    return off


def exclusive_scan_capped(counts, cap):
    """offsets clamped to `cap` records + the true total (device int32 [1])."""
    counts = i32(counts, "counts").view(-1)
    n = counts.numel()
    off = torch.empty((n + 1,), dtype=torch.int32, device=counts.device)
    total = torch.empty((1,), dtype=torch.int32, device=counts.device)
    _call("tir_exclusive_scan_capped", _ptr(counts), _ptr(off), n, int(cap), _ptr(total), _stream())
    return off, total


def compact_primary(field: TirField, rays, ray_jitter, weight, offsets, total):
    rays = f32(rays, "rays", 6)
    B, S = weight.shape
    dev = rays.device
    if ray_jitter is not None:
        ray_jitter = f32(ray_jitter, "ray_jitter").view(-1)
    rec_ray = torch.empty((total,), dtype=torch.int32, device=dev)
    rec_k = torch.empty((total,), dtype=torch.int32, device=dev)
    rec_w = torch.empty((total,), dtype=torch.float32, device=dev)
    rec_xyz = torch.empty((total, 3), dtype=torch.float32, device=dev)
    if total > 0:
        _call("tir_compact_primary", C.byref(field), _ptr(rays), _ptr(ray_jitter), _ptr(weight),
                                        _ptr(offsets), B, S, _ptr(rec_ray), _ptr(rec_k), _ptr(rec_w),
                                        _ptr(rec_xyz), _stream())
    return rec_ray, rec_k, rec_w, rec_xyz


def composite_primary(rays, offsets, rec_w, rgb, brdf, brdf_jit, pred_n, der_n, acc, depth,
                      white_bg, is_relight, fixed_fresnel):
    rays = f32(rays, "rays", 6)
    B = rays.shape[0]
    out = torch.empty((B, MAP_STRIDE), dtype=torch.float32, device=rays.device)
    _call("tir_composite_primary", _ptr(rays), _ptr(offsets), _ptr(rec_w), _ptr(rgb), _ptr(brdf),
                                      _ptr(brdf_jit), _ptr(pred_n), _ptr(der_n), _ptr(acc), _ptr(depth),
                                      B, int(bool(white_bg)), int(bool(is_relight)),
                                      float(fixed_fresnel), _ptr(out), _stream())
    return out


# ---- secondary march ------------------------------------------------------------------------------
def march_secondary(field: TirField, origins, dirs, z_vals, n_rays, org_map=None, dir_map=None,
                    active=None, t_stop=0.0, want_records=False, rec_cap=0, want_nerfactor=True, n_dirs=0,
                    ray_ids=None, n_ids_dev=None, vis=None, rec_cnt=None, counter=None):
    """ray_ids / n_ids_dev: march only the listed pair ids (tir_march_secondary_ids_fwd); vis / rec_cnt: pre-filled
    per-pair buffers (shade_setup_compact wrote the masked pairs' zeros into them)."""
    origins = f32(origins, "origins", 3)
    dirs = f32(dirs, "dirs", 3)
    z_vals = f32(z_vals, "z_vals").view(-1)
    dev = origins.device
    n_sample = z_vals.numel()
    org_map = None if org_map is None else i32(org_map, "org_map")
    dir_map = None if dir_map is None else i32(dir_map, "dir_map")
    if active is not None:
        active = _req(active, torch.uint8, "active")
    if vis is None:
        vis = torch.empty((n_rays,), dtype=torch.float32, device=dev)
    oma = torch.empty((n_rays,), dtype=torch.float32, device=dev) if want_nerfactor else None
    rec = None
    if want_records:
        rec = {
            # [total, written prefix]; `counter`: a caller-owned pair that is already zero on the device
            "counter": counter if counter is not None else torch.zeros((2,), dtype=torch.int32, device=dev),
            "ray": torch.empty((rec_cap,), dtype=torch.int32, device=dev),
            "w": torch.empty((rec_cap,), dtype=torch.float32, device=dev),
            "xyz": torch.empty((rec_cap, 3), dtype=torch.float32, device=dev),
            "off": torch.empty((n_rays,), dtype=torch.int32, device=dev),
            "cnt": rec_cnt if rec_cnt is not None else torch.empty((n_rays,), dtype=torch.int32, device=dev),
            "cap": rec_cap,
        }
    r = rec or {}
    if ray_ids is not None:
        ray_ids = i32(ray_ids, "ray_ids")
    _call("tir_march_secondary_fwd" if ray_ids is None else "tir_march_secondary_ids_fwd",
          C.byref(field), _ptr(origins), _ptr(org_map), _ptr(dirs), _ptr(dir_map), _ptr(active),
          n_rays, int(n_dirs), n_sample, _ptr(z_vals), float(t_stop), _ptr(vis), _ptr(oma),
          _ptr(r.get("counter")), int(rec_cap), _ptr(r.get("ray")), _ptr(r.get("w")), _ptr(r.get("xyz")),
          _ptr(r.get("off")), _ptr(r.get("cnt")), _stats_ptr("tir_march_secondary_fwd", dev),
          *(() if ray_ids is None else (_ptr(ray_ids), _ptr(n_ids_dev))), _stream())
    return vis, oma, rec


def accumulate_records(off, cnt, rec_w, rec_rgb, n_rays):
    out = torch.empty((n_rays, 3), dtype=torch.float32, device=off.device)
    _call("tir_accumulate_records", _ptr(off), _ptr(cnt), _ptr(rec_w), _ptr(rec_rgb), n_rays,
                                       _ptr(out), _stream())
    return out


# ---- shading --------------------------------------------------------------------------------------
def env_sg(lgtSGs, rot, dirs):
    sgs = f32(lgtSGs.detach(), "lgtSGs", 7)
    rot = f32(rot, "light_rotation_matrix").view(-1, 9)
    dirs = f32(dirs, "dirs", 3).view(-1, 3)
    L, D = rot.shape[0], dirs.shape[0]
    out = torch.empty((L, D, 3), dtype=torch.float32, device=dirs.device)
    desc = TirEnvSG(sgs.data_ptr(), rot.data_ptr(), sgs.shape[0], L)
    _call("tir_env_sg_fwd", C.byref(desc), _ptr(dirs), D, _ptr(out), _stream())
    return out


def env_pixel(light_rgbs, H, W, rot, dirs, softplus=True):
    """light_kind == 'pixel': [H*W, 3] raw map parameters -> environment radiance [L, D, 3] (tir_env_pixel_fwd); softplus=False:
    the image as it is (light_kind == 'gt', the data set's probe)."""
    lr = f32(light_rgbs.detach(), "_light_rgbs", 3).view(-1, 3)
    if lr.shape[0] != H * W:
        raise ValueError(f"_light_rgbs: expected {H * W} rows, got {lr.shape[0]}")
    rot = f32(rot, "light_rotation_matrix").view(-1, 9)
    dirs = f32(dirs, "dirs", 3).view(-1, 3)
    L, D = rot.shape[0], dirs.shape[0]
    out = torch.empty((L, D, 3), dtype=torch.float32, device=dirs.device)
    _call("tir_env_pixel_fwd", _ptr(lr), int(H), int(W), _ptr(rot), _ptr(dirs), L, D, int(bool(softplus)), _ptr(out), _stream())
    return out


def env_pixel_bwd(light_rgbs, H, W, rot, dirs, g_env):
    lr = f32(light_rgbs.detach(), "_light_rgbs", 3).view(-1, 3)
    rot = f32(rot, "light_rotation_matrix").view(-1, 9)
    dirs = f32(dirs, "dirs", 3).view(-1, 3)
    g_env = f32(g_env, "g_env", 3)
    L, D = rot.shape[0], dirs.shape[0]
    g = torch.zeros_like(lr)
    _call("tir_env_pixel_bwd", _ptr(lr), int(H), int(W), _ptr(rot), _ptr(dirs), L, D, _ptr(g_env), _ptr(g), _stream())
    return g.view_as(light_rgbs)


def shade_setup(maps, rays, dirs, acc_thres=-1e30):
    maps = f32(maps, "maps", MAP_STRIDE)
    rays = f32(rays, "rays", 6)
    dirs = f32(dirs, "dirs", 3)
    M, D = maps.shape[0], dirs.shape[0]
    surf = torch.empty((M, 3), dtype=torch.float32, device=maps.device)
    active = torch.empty((M, D), dtype=torch.uint8, device=maps.device)
    _call("tir_shade_setup", _ptr(maps), _ptr(rays), _ptr(dirs), M, D, float(acc_thres), _ptr(surf), _ptr(active),
                                _stream())
    return surf, active


def shade_setup_compact(maps, rays, dirs, acc_thres, n_active):
    """shade_setup + compacted active-pair list.  n_active: zeroed int32 device scalar (re-armed by
    shade_integrate_records).  Returns surf, active, pair_ids [M*D], vis [M*D] and rec_cnt [M*D] (zeros at masked pairs)."""
    maps = f32(maps, "maps", MAP_STRIDE)
    rays = f32(rays, "rays", 6)
    dirs = f32(dirs, "dirs", 3)
    M, D = maps.shape[0], dirs.shape[0]
    dev = maps.device
    surf = torch.empty((M, 3), dtype=torch.float32, device=dev)
    active = torch.empty((M, D), dtype=torch.uint8, device=dev)
    pair_ids = torch.empty((M * D,), dtype=torch.int32, device=dev)
    vis = torch.empty((M * D,), dtype=torch.float32, device=dev)
    rec_cnt = torch.empty((M * D,), dtype=torch.int32, device=dev)
    _call("tir_shade_setup_compact", _ptr(maps), _ptr(rays), _ptr(dirs), M, D, float(acc_thres), _ptr(surf), _ptr(active),
          _ptr(pair_ids), _ptr(n_active), _ptr(vis), _ptr(rec_cnt), TUNE["pair_order"], _stream())
    return surf, active, pair_ids, vis, rec_cnt


def shade_integrate(maps, rays, dirs, light_idx, vis, indirect, env, weight_d, equal_area=False,
                    use_srgb=True, acc_thres=-1e30):
    maps = f32(maps, "maps", MAP_STRIDE)
    rays = f32(rays, "rays", 6)
    dirs = f32(dirs, "dirs", 3)
    M, D = maps.shape[0], dirs.shape[0]
    light_idx = i32(light_idx, "light_idx").view(-1)
    vis = f32(vis, "vis")
    env = f32(env, "env", 3)
    if indirect is not None:
        indirect = f32(indirect, "indirect", 3)
    if weight_d is not None:
        weight_d = f32(weight_d, "light_area_weight")
    out = torch.empty((M, 3), dtype=torch.float32, device=maps.device)
    _call("tir_shade_integrate", _ptr(maps), _ptr(rays), _ptr(dirs), _ptr(light_idx), _ptr(vis),
                                    _ptr(indirect), _ptr(env), _ptr(weight_d), M, D, env.shape[0],
                                    int(bool(equal_area)), int(bool(use_srgb)), float(acc_thres), _ptr(out), _stream())
    return out


def shade_integrate_records(maps, rays, dirs, light_idx, vis, rec_off, rec_cnt, rec_w, rec_rgb, env, weight_d,
                            equal_area=False, use_srgb=True, acc_thres=-1e30, reset_counter=None):
    """shade_integrate with the per-ray indirect sum fused in (reads the secondary records directly)."""
    maps = f32(maps, "maps", MAP_STRIDE)
    rays = f32(rays, "rays", 6)
    dirs = f32(dirs, "dirs", 3)
    M, D = maps.shape[0], dirs.shape[0]
    light_idx = i32(light_idx, "light_idx").view(-1)
    env = f32(env, "env", 3)
    if weight_d is not None:
        weight_d = f32(weight_d, "light_area_weight")
    out = torch.empty((M, 3), dtype=torch.float32, device=maps.device)
    _call("tir_shade_integrate_records", _ptr(maps), _ptr(rays), _ptr(dirs), _ptr(light_idx), _ptr(f32(vis, "vis")),
          _ptr(i32(rec_off, "rec_off")), _ptr(i32(rec_cnt, "rec_cnt")), _ptr(f32(rec_w, "rec_w")),
          _ptr(f32(rec_rgb, "rec_rgb", 3)), _ptr(env), _ptr(weight_d), M, D, env.shape[0], int(bool(equal_area)),
          int(bool(use_srgb)), float(acc_thres), _ptr(out), _ptr(reset_counter), _stream())
    return out


def relight_importance(normal, albedo, rough, fresnel, rays_d, light_dir, light_rgb, light_pdf, vis):
    normal, albedo = f32(normal, "normal", 3), f32(albedo, "albedo", 3)
    rough = f32(rough, "roughness").view(-1)
    fresnel, rays_d = f32(fresnel, "fresnel", 3), f32(rays_d, "rays_d", 3)
    light_dir, light_rgb = f32(light_dir, "light_dir", 3), f32(light_rgb, "light_rgb", 3)
    M, Ns = light_dir.shape[0], light_dir.shape[1]
    light_pdf = f32(light_pdf, "light_pdf").view(M, Ns)
    vis = f32(vis, "vis").view(M, Ns)
    out = torch.empty((M, 3), dtype=torch.float32, device=normal.device)
    _call("tir_relight_importance", _ptr(normal), _ptr(albedo), _ptr(rough), _ptr(fresnel), _ptr(rays_d),
                                       _ptr(light_dir), _ptr(light_rgb), _ptr(light_pdf), _ptr(vis), M, Ns,
                                       _ptr(out), _stream())
    return out


def env_sample_setup(row_cdf, col_cdf, env_dir, normal, n_samples, seed, offset):
    """Importance samples of an H x W environment map for every surface point + the cosine mask
    (tir_env_sample_setup).  Returns cell [M, Ns] int32, active [M, Ns] uint8."""
    H, W = col_cdf.shape
    normal = f32(normal, "normal", 3)
    M = normal.shape[0]
    cell = torch.empty((M, n_samples), dtype=torch.int32, device=normal.device)
    active = torch.empty((M, n_samples), dtype=torch.uint8, device=normal.device)
    _call("tir_env_sample_setup", _ptr(f32(row_cdf, "row_cdf")), _ptr(f32(col_cdf, "col_cdf")), H, W,
          _ptr(f32(env_dir, "env_dir", 3)), _ptr(normal), M, int(n_samples), int(seed) & (2 ** 64 - 1),
          int(offset) & (2 ** 64 - 1), _ptr(cell), _ptr(active), _stream())
    return cell, active


def c5_pair_order():
    """How the importance-sampled (point, cell) pairs reach the visibility march: TENSOIR_C5_PAIRS = `binned` (default: the
    compacted list of the unmasked pairs, every 512 consecutive pairs -- one surface point at 512 samples -- ordered by a
    15 x 17 grid of direction bins, tir_env_sample_setup_list), `compact` (the same list without the bins) or `mask` (every
    pair with a uint8 mask, tir_env_sample_setup).  Results do not depend on it (profiles/r04_c5_pair_lists.json).
    Returns (mode, bins, block_pairs); TENSOIR_C5_BINS = `RxC` and TENSOIR_C5_BLOCK_PAIRS override the list's grouping."""
    mode = os.environ.get("TENSOIR_C5_PAIRS", "binned").strip().lower() or "binned"
    if mode not in ("binned", "compact", "mask"):
        raise ValueError(f"TENSOIR_C5_PAIRS={mode!r}: expected binned, compact or mask")
    bins, block = ((15, 17), 512) if mode == "binned" else ((1, 1), 512)
    if os.environ.get("TENSOIR_C5_BINS"):
        bins = tuple(int(v) for v in os.environ["TENSOIR_C5_BINS"].lower().split("x"))
        if len(bins) != 2:
            raise ValueError("TENSOIR_C5_BINS: expected ROWSxCOLS, e.g. 8x8")
    if os.environ.get("TENSOIR_C5_BLOCK_PAIRS"):
        block = int(os.environ["TENSOIR_C5_BLOCK_PAIRS"])
    return mode, bins, block


def cdf_guide_tables(row_cdf, col_cdf):
    """Guide tables of the inverse-CDF search (tir_env_sample_setup_list): for G = the next power of two >= n, entry k of
    a table is the search result for u = k / G -- the first index whose cdf exceeds it, clamped to n - 1.  Built with
    torch.searchsorted on the fp32 tables the kernel searches (k / G is exact in fp32).  Returns (row_guide int32 [Gr + 1],
    col_guide [H, Gc + 2] 16-bit entries packed in pairs into int32 [H, Gc / 2 + 1] (little endian; the last entry pads the
    row to whole words), Gr, Gc), or None when a row has more than 65535 columns or fewer than 2 (Gc + 2 entries would not
    fill whole words; a one-column map needs no search)."""
    H, W = col_cdf.shape
    if W > 65535 or W < 2:
        return None
    gr, gc = 1 << max(0, (H - 1).bit_length()), 1 << max(0, (W - 1).bit_length())
    dev = row_cdf.device
    tr = torch.arange(gr + 1, dtype=torch.float32, device=dev) / gr
    tc = torch.arange(gc + 1, dtype=torch.float32, device=dev) / gc
    rg = torch.searchsorted(row_cdf.contiguous(), tr, right=True).clamp_(max=H - 1).to(torch.int32)
    cg = torch.searchsorted(col_cdf.contiguous(), tc.unsqueeze(0).expand(H, -1).contiguous(), right=True).clamp_(max=W - 1)
    rg[-1], cg[:, -1] = H - 1, W - 1
    cg = torch.cat([cg, cg[:, -1:]], dim=1).to(torch.int32)                      # Gc + 2 entries per row
    packed = (cg[:, 0::2] | (cg[:, 1::2] << 16)).to(torch.int32)                 # two 16-bit entries per word
    return rg.contiguous(), packed.contiguous(), gr, gc


def env_sample_setup_list(row_cdf, col_cdf, env_dir, normal, n_samples, seed, offset, bins=(1, 1), block_pairs=256, guide=None,
                          m_dev=None):
    """tir_env_sample_setup_list[_n]: cells of every (point, sample) + the list of the pairs that pass the cosine mask.
    Returns cell [M, Ns] int32, vis [M, Ns] fp32 (0 where masked, the rest for the march to fill), pair_ids [M*Ns] int32,
    n_active [1] int32 (device).  m_dev (int32 device scalar): only the first min(M, m_dev) points exist."""
    H, W = col_cdf.shape
    normal = f32(normal, "normal", 3)
    M, dev = normal.shape[0], normal.device
    cell = torch.empty((M, n_samples), dtype=torch.int32, device=dev)
    vis = torch.empty((M, n_samples), dtype=torch.float32, device=dev)
    pair_ids = torch.empty((M * n_samples,), dtype=torch.int32, device=dev)
    n_active = torch.zeros((1,), dtype=torch.int32, device=dev)
    stride = int(env_dir.shape[-1])                    # [H*W, 3] directions or the [H*W, 8] records of pack_env_cells
    if stride not in (3, 8):
        raise ValueError(f"env_dir: expected [H*W, 3] directions or [H*W, 8] cell records, got {tuple(env_dir.shape)}")
    _call("tir_env_sample_setup_list_n", _ptr(f32(row_cdf, "row_cdf")), _ptr(f32(col_cdf, "col_cdf")), H, W,
          _ptr(f32(env_dir, "env_dir", stride)), stride, _ptr(normal), M, int(n_samples), int(seed) & (2 ** 64 - 1),
          int(offset) & (2 ** 64 - 1), int(bins[0]), int(bins[1]), int(block_pairs),
          *((None, None, 0, 0) if guide is None else (_ptr(_req(guide[0], torch.int32, "row_guide")),
                                                      _ptr(_req(guide[1], torch.int32, "col_guide")), int(guide[2]), int(guide[3]))),
          _ptr(cell), _ptr(vis), _ptr(pair_ids), _ptr(n_active), _ptr(m_dev), _stream())
    return cell, vis, pair_ids, n_active


def pack_env_cells(env_dir, env_rgb, env_pdf):
    """[H*W, 8] fp32 records {dir.xyz, pdf_return, rgb, 0} of an environment map for tir_relight_importance_cells_packed."""
    d, c, p = f32(env_dir, "env_dir", 3).view(-1, 3), f32(env_rgb, "env_rgb", 3).view(-1, 3), f32(env_pdf, "env_pdf").view(-1, 1)
    return torch.cat([d, p, c, torch.zeros_like(p)], dim=1).contiguous()


def relight_importance_cells(normal, albedo, rough, fresnel, rays_d, cell, env_dir, env_rgb, env_pdf, vis, env_cell=None, m_dev=None):
    """BRDF x radiance x cosine / pdf mean over a point's samples -> sRGB (scripts/relight_importance.py:133-160).  env_cell
    (pack_env_cells of the same three tables): one 32-byte record per sample instead of three gathers; same result.
    m_dev (int32 device scalar, packed form only): only the first min(M, m_dev) points exist (rows beyond: not written)."""
    normal, albedo = f32(normal, "normal", 3), f32(albedo, "albedo", 3)
    rough = f32(rough, "roughness").view(-1)
    fresnel, rays_d = f32(fresnel, "fresnel", 3), f32(rays_d, "rays_d", 3)
    cell = i32(cell, "cell")
    M, Ns = cell.shape
    vis = f32(vis, "vis").view(M, Ns)
    out = torch.empty((M, 3), dtype=torch.float32, device=normal.device)
    if env_cell is not None:
        _call("tir_relight_importance_cells_packed_n", _ptr(normal), _ptr(albedo), _ptr(rough), _ptr(fresnel), _ptr(rays_d), _ptr(cell),
              _ptr(f32(env_cell, "env_cell", 8)), _ptr(vis), M, Ns, _ptr(out), _ptr(m_dev), _stream())
        return out
    if m_dev is not None:
        raise ValueError("relight_importance_cells: a device-side point count needs the packed cell records (env_cell)")
    _call("tir_relight_importance_cells", _ptr(normal), _ptr(albedo), _ptr(rough), _ptr(fresnel), _ptr(rays_d), _ptr(cell),
          _ptr(f32(env_dir, "env_dir", 3)), _ptr(f32(env_rgb, "env_rgb", 3)), _ptr(f32(env_pdf, "env_pdf")), _ptr(vis),
          M, Ns, _ptr(out), _stream())
    return out


def env_lookup(env_rgb, dirs):
    """Bilinear lookup of an [H, W, 3] map at [n, 3] directions (tir_env_lookup)."""
    env_rgb = f32(env_rgb, "env_rgb", 3)
    H, W = env_rgb.shape[0], env_rgb.shape[1]
    dirs = f32(dirs, "dirs", 3).view(-1, 3)
    out = torch.empty_like(dirs)
    _call("tir_env_lookup", _ptr(env_rgb), H, W, _ptr(dirs), dirs.shape[0], _ptr(out), _stream())
    return out


def surface_compact(maps, rays, acc_thres=0.5):
    """tir_surface_compact: the acc > acc_thres rows of a chunk's primary maps [B, 20] as compacted surface-point arrays
    (capacity B, ascending row order) -> dict(surf, normal, albedo, rough, fresnel, rays_d, slot [B] int32, n_hit [1] int32)."""
    maps = f32(maps, "maps", MAP_STRIDE)
    rays = f32(rays, "rays", 6)
    B, dev = rays.shape[0], rays.device
    v3 = lambda: torch.empty((B, 3), dtype=torch.float32, device=dev)
    out = {"surf": v3(), "normal": v3(), "albedo": v3(), "rough": torch.empty((B,), dtype=torch.float32, device=dev), "fresnel": v3(),
           "rays_d": v3(), "slot": torch.empty((B,), dtype=torch.int32, device=dev), "n_hit": torch.empty((1,), dtype=torch.int32, device=dev)}
    _call("tir_surface_compact", _ptr(maps), _ptr(rays), B, float(acc_thres), _ptr(out["surf"]), _ptr(out["normal"]), _ptr(out["albedo"]),
          _ptr(out["rough"]), _ptr(out["fresnel"]), _ptr(out["rays_d"]), _ptr(out["slot"]), _ptr(out["n_hit"]), _stream())
    return out


def env_compose(env_rgb, rays, slot, fg_rgb, out, col=0):
    """tir_env_compose: out[:, col:col+3] = the relit colour fg_rgb[slot[i]] of a foreground row, the background lookup of the
    map at the ray direction rays[i, 3:6] otherwise.  out [B, >= col + 3] fp32, written in place."""
    env_rgb = f32(env_rgb, "env_rgb", 3)
    rays = f32(rays, "rays", 6)
    B = rays.shape[0]
    if out.dtype != torch.float32 or not out.is_contiguous() or out.shape[0] != B or out.shape[1] < col + 3:
        raise ValueError("env_compose: out must be a contiguous fp32 [B, >= col + 3] tensor")
    _call("tir_env_compose", _ptr(env_rgb), env_rgb.shape[0], env_rgb.shape[1], C.c_void_p(rays.data_ptr() + 12), 6, B,
          _ptr(i32(slot, "slot")), _ptr(f32(fg_rgb, "fg_rgb", 3)), C.c_void_p(out.data_ptr() + 4 * col), int(out.shape[1]), _stream())
    return out


def ggx_specular(normal, v, l, rough, fresnel):
    normal, v = f32(normal, "normal", 3), f32(v, "pts2c", 3)
    l = f32(l, "pts2l", 3)
    M, D = l.shape[0], l.shape[1]
    rough = f32(rough.expand(M, 3) if rough.shape[-1] == 1 else rough, "roughness", 3)
    fresnel = f32(fresnel.expand(M, 3) if fresnel.shape[-1] == 1 else fresnel, "fresnel", 3)
    out = torch.empty((M, D, 3), dtype=torch.float32, device=l.device)
    _call("tir_ggx_specular", _ptr(normal), _ptr(v), _ptr(l), _ptr(rough), _ptr(fresnel), M, D,
                                 _ptr(out), _stream())
    return out


# ---- training (backward) kernels: SURVEY.md section 8(f)-1 -------------------------------------------
from ._lib import TirFieldGrad  # noqa: E402


def march_primary_train(field: TirField, rays, ray_jitter, n_samples, t_stop):
    """march_primary that also returns the per-sample density sigma [B,S] (needed by the backward)."""
    rays = f32(rays, "rays", 6)
    B, dev = rays.shape[0], rays.device
    if ray_jitter is not None:
        ray_jitter = f32(ray_jitter, "ray_jitter").view(-1)
    weight = torch.empty((B, n_samples), dtype=torch.float32, device=dev)
    sigma = torch.empty((B, n_samples), dtype=torch.float32, device=dev)
    acc = torch.empty((B,), dtype=torch.float32, device=dev)
    depth = torch.empty((B,), dtype=torch.float32, device=dev)
    tend = torch.empty((B,), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    _call("tir_march_primary_train_fwd", C.byref(field), _ptr(rays), _ptr(ray_jitter), B, n_samples, float(t_stop),
          _ptr(weight), _ptr(sigma), _ptr(acc), _ptr(depth), _ptr(tend), _ptr(cnt), _stream())
    return weight, sigma, acc, depth, tend, cnt


def composite_primary_bwd(rays, offsets, rec_k, rec_w, rgb, brdf, brdf_jit, pred_n, der_n, acc, depth, S,
                          white_bg, is_relight, fixed_fresnel, g_maps):
    rays = f32(rays, "rays", 6)
    g_maps = f32(g_maps, "g_maps", MAP_STRIDE)
    B, dev = rays.shape[0], rays.device
    like = lambda t: None if t is None else torch.empty_like(t)
    g_rgb, g_brdf, g_brdf_jit, g_pred, g_der = like(rgb), like(brdf), like(brdf_jit), like(pred_n), like(der_n)
    g_weight = torch.zeros((B, S), dtype=torch.float32, device=dev)
    g_acc = torch.empty((B,), dtype=torch.float32, device=dev)
    g_depth = torch.empty((B,), dtype=torch.float32, device=dev)
    _call("tir_composite_primary_bwd", _ptr(rays), _ptr(offsets), _ptr(rec_k), _ptr(rec_w), _ptr(rgb), _ptr(brdf),
          _ptr(brdf_jit), _ptr(pred_n), _ptr(der_n), _ptr(acc), _ptr(depth), B, int(S), int(bool(white_bg)),
          int(bool(is_relight)), float(fixed_fresnel), _ptr(g_maps), _ptr(g_rgb), _ptr(g_brdf), _ptr(g_brdf_jit),
          _ptr(g_pred), _ptr(g_der), _ptr(g_weight), _ptr(g_acc), _ptr(g_depth), _stream())
    return g_rgb, g_brdf, g_brdf_jit, g_pred, g_der, g_weight, g_acc, g_depth


def march_primary_bwd(field: TirField, grad: TirFieldGrad, rays, ray_jitter, sigma, weight, g_weight, g_acc, g_depth,
                      want_g_feature=False):
    B, S = weight.shape
    g_feature = torch.empty_like(weight) if want_g_feature else None
    if ray_jitter is not None:
        ray_jitter = f32(ray_jitter, "ray_jitter").view(-1)
    _call("tir_march_primary_bwd", C.byref(field), C.byref(grad), _ptr(f32(rays, "rays", 6)), _ptr(ray_jitter), B, S,
          _ptr(f32(sigma, "sigma")), _ptr(f32(weight, "weight")), _ptr(f32(g_weight, "g_weight")),
          _ptr(f32(g_acc, "g_acc")), _ptr(f32(g_depth, "g_depth")), _ptr(g_feature), _stream())
    return g_feature


def density_grad_bwd(field: TirField, grad: TirFieldGrad, xyz, g_normal):
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    g_normal = f32(g_normal, "g_normal", 3).view(-1, 3)
    _call("tir_density_grad_bwd", C.byref(field), C.byref(grad), _ptr(xyz), _ptr(g_normal), xyz.shape[0], _stream())


def vm_app_bwd(field: TirField, grad: TirFieldGrad, xyz, light_idx, idx_map, g_rad, g_int):
    """Scatter d feat into the appearance planes/lines + light rows; returns (y_rad, y_int) [n, 3*Ca]."""
    xyz = f32(xyz, "xyz", 3).view(-1, 3)
    n = xyz.shape[0]
    nch = 3 * field.n_acomp
    g_rad = None if g_rad is None else f32(g_rad, "g_rad")
    g_int = None if g_int is None else f32(g_int, "g_int")
    stride = (g_rad if g_rad is not None else g_int).shape[1]
    y_rad = torch.empty((n, nch), dtype=torch.float32, device=xyz.device) if g_rad is not None else None
    y_int = torch.empty((n, nch), dtype=torch.float32, device=xyz.device) if g_int is not None else None
    if light_idx is not None:
        light_idx = i32(light_idx, "light_idx").view(-1)
    if idx_map is not None:
        idx_map = i32(idx_map, "idx_map").view(-1)
    _call("tir_vm_app_bwd", C.byref(field), C.byref(grad), _ptr(xyz), _ptr(light_idx), _ptr(idx_map), _ptr(g_rad),
          _ptr(g_int), stride, n, _ptr(y_rad), _ptr(y_int), _stream())
    return y_rad, y_int


def mlp_train(m: "PackedMlp", feat, aux, aux_map=None, aux_mod=0, impl=None, n_dev=None):
    """Decoder forward that also returns the hidden activations (h1, h2) [n,128]: split-bf16 matrix cores when the
    product's decoder mode is bf16x3 (default), exact fp32 MFMA otherwise.  n_dev: device-side row count (rows past
    it are left unwritten)."""
    impl = impl or MLP_IMPL
    feat = f32(feat, "feat")
    aux = f32(aux, "aux", 3)
    n = feat.shape[0]
    if aux_map is not None:
        aux_map = i32(aux_map, "aux_map").view(-1)
    out = torch.empty((n, m.out_dim), dtype=torch.float32, device=feat.device)
    h1 = torch.empty((n, 128), dtype=torch.float32, device=feat.device)
    h2 = torch.empty((n, 128), dtype=torch.float32, device=feat.device)
    _call("tir_mlp_train_fwd_bf16x3" if impl == "bf16x3" else "tir_mlp_train_fwd", C.byref(m.desc), _ptr(feat),
          feat.shape[1], _ptr(aux), _ptr(aux_map), int(aux_mod), _ptr(out), _ptr(h1), _ptr(h2), n, _ptr(n_dev), _stream())
    return out, h1, h2


def mlp_inputs(m: "PackedMlp", feat, aux, aux_map=None, aux_mod=0):
    feat = f32(feat, "feat")
    aux = f32(aux, "aux", 3)
    n = feat.shape[0]
    if aux_map is not None:
        aux_map = i32(aux_map, "aux_map").view(-1)
    x = torch.empty((n, 160), dtype=torch.float32, device=feat.device)
    _call("tir_mlp_inputs", C.byref(m.desc), _ptr(feat), feat.shape[1], _ptr(aux), _ptr(aux_map), int(aux_mod),
          _ptr(x), n, _stream())
    return x


def pack_mlp_bwd(w0, w1, w2, feat_dim, pe):
    ws = [f32(t.detach(), "mlp weight") for t in (w0, w1, w2)]
    hidden, out_dim = w1.shape[0], w2.shape[0]
    n = lib().tir_mlp_bwd_packed_floats(feat_dim, pe, hidden, out_dim)
    if n < 0:
        check(int(n), "tir_mlp_bwd_packed_floats")
    packed = torch.empty((int(n),), dtype=torch.float32, device=w0.device)
    _call("tir_pack_mlp_bwd", *[_ptr(t) for t in ws], feat_dim, pe, hidden, out_dim, _ptr(packed), _stream())
    return packed


def mlp_bwd(m: "PackedMlp", packed_bwd, feat, out, g_out, h1, h2, impl=None):
    """Backward-data of one decoder (g_feat, dz1, dz2, dz3): split-bf16 matrix cores when the product's decoder mode
    is bf16x3 (default), exact fp32 MFMA otherwise."""
    impl = impl or MLP_IMPL
    feat = f32(feat, "feat")
    n = feat.shape[0]
    dev = feat.device
    g_feat = torch.empty((n, FEAT_STRIDE), dtype=torch.float32, device=dev)
    dz1 = torch.empty((n, 128), dtype=torch.float32, device=dev)
    dz2 = torch.empty((n, 128), dtype=torch.float32, device=dev)
    dz3 = torch.empty((n, 4), dtype=torch.float32, device=dev)
    _call("tir_mlp_bwd_bf16x3" if impl == "bf16x3" else "tir_mlp_bwd", C.byref(m.desc), _ptr(packed_bwd), _ptr(feat),
          feat.shape[1], _ptr(f32(out, "out")),
          _ptr(f32(g_out, "g_out")), _ptr(h1), _ptr(h2), n, _ptr(g_feat), _ptr(dz1), _ptr(dz2), _ptr(dz3), _stream())
    return g_feat, dz1, dz2, dz3


def mlp_bwd_multi(jobs):
    """Backward-data of up to four decoder invocations over the same rows in one launch (tir_mlp_bwd_multi_bf16x3).
    jobs: list of (PackedMlp, packed_bwd, feat [n, FEAT_STRIDE], out, g_out, h1, h2); returns [(g_feat, dz1, dz2, dz3)]."""
    k = len(jobs)
    n = jobs[0][2].shape[0]
    dev = jobs[0][2].device
    res, cols = [], [[] for _ in range(11)]
    for m, pb, feat, out, g_out, h1, h2 in jobs:
        feat = f32(feat, "feat")
        if feat.shape[0] != n or feat.shape[1] != FEAT_STRIDE:
            raise ValueError(f"mlp_bwd_multi: every job needs [n, {FEAT_STRIDE}] feature rows")
        g_feat = torch.empty((n, FEAT_STRIDE), dtype=torch.float32, device=dev)
        dz1 = torch.empty((n, 128), dtype=torch.float32, device=dev)
        dz2 = torch.empty((n, 128), dtype=torch.float32, device=dev)
        dz3 = torch.empty((n, 4), dtype=torch.float32, device=dev)
        res.append((g_feat, dz1, dz2, dz3))
        for c, t in zip(cols, (pb, feat, f32(out, "out"), f32(g_out, "g_out"), h1, h2, g_feat, dz1, dz2, dz3)):
            c.append(t)
    arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() for t in ts])
    descs = (C.POINTER(TirMlp) * k)(*[C.pointer(j[0].desc) for j in jobs])
    _call("tir_mlp_bwd_multi_bf16x3", descs, arr(cols[0]), arr(cols[1]), FEAT_STRIDE, arr(cols[2]), arr(cols[3]), arr(cols[4]),
          arr(cols[5]), k, n, arr(cols[6]), arr(cols[7]), arr(cols[8]), arr(cols[9]), _stream())
    _KEEP_ALIVE = cols          # noqa: F841  (operands stay referenced until the launch is queued)
    return res


def gemm_tn_small(pairs, M, N, C_out):
    """C_out[M, N] += sum over the (A, B) pairs of A[:, :M]^T @ B[:, :N], M <= 32, N <= 160, every pair over the same number of
    rows (tir_gemm_tn_small_bf16x3: the basis-matrix gradient of several appearance gathers in one launch)."""
    k = len(pairs)
    As = [f32(a, "A") for a, _ in pairs]
    Bs = [f32(b, "B") for _, b in pairs]
    n = As[0].shape[0]
    if any(a.shape[0] != n or b.shape[0] != n or a.shape[1] != As[0].shape[1] or b.shape[1] != Bs[0].shape[1] for a, b in zip(As, Bs)):
        raise ValueError("gemm_tn_small: every pair needs the same row count and strides")
    arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() for t in ts])
    _call("tir_gemm_tn_small_bf16x3", arr(As), As[0].shape[1], int(M), arr(Bs), Bs[0].shape[1], int(N), k, n, _ptr(C_out),
          C_out.shape[1], _stream())
    _KEEP_ALIVE = (As, Bs)          # noqa: F841
    return C_out


def mlp_wgrad_multi(jobs):
    """Weight gradients of up to four decoder invocations over the same rows in ONE launch (tir_mlp_wgrad_multi): no
    materialised input rows, no per-layer GEMM launches.  jobs: list of (dz1, dz2, dz3, h1, h2, feat [n, stride >= 27; 32 = FEAT_STRIDE takes the LDS-staged kernel], aux,
    aux_map or None, dW0 [128,150], db0 [128], dW1 [128,128], db1 [128], dW2 [4,128], db2 [4]); the six outputs are
    accumulated into (zero them first; two jobs may name the same outputs)."""
    k = len(jobs)
    n = jobs[0][0].shape[0]
    cols = [[] for _ in range(14)]
    null_map = True
    for job in jobs:
        dz1, dz2, dz3, h1, h2, feat, aux, aux_map, dW0, db0, dW1, db1, dW2, db2 = job
        feat = f32(feat, "feat")
        if feat.shape[0] != n or feat.shape[1] < 27 or feat.shape[1] != jobs[0][5].shape[1]:
            raise ValueError("mlp_wgrad_multi: every job needs [n, stride >= 27] feature rows of one common stride")
        for t, shape, name in ((dz1, (n, 128), "dz1"), (dz2, (n, 128), "dz2"), (dz3, (n, 4), "dz3"), (h1, (n, 128), "h1"),
                               (h2, (n, 128), "h2"), (dW0, (128, 150), "dW0"), (db0, (128,), "db0"), (dW1, (128, 128), "dW1"),
                               (db1, (128,), "db1"), (dW2, (4, 128), "dW2"), (db2, (4,), "db2")):
            if tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != torch.float32:
                raise ValueError(f"mlp_wgrad_multi: {name} must be contiguous fp32 {shape}, got {tuple(t.shape)}")
        aux = f32(aux, "aux", 3)
        if aux_map is not None:
            aux_map = i32(aux_map, "aux_map")
            null_map = False
        for c, t in zip(cols, (dz1, dz2, dz3, h1, h2, feat, aux, aux_map, dW0, db0, dW1, db1, dW2, db2)):
            c.append(t)
    arr = lambda ts: (C.c_void_p * k)(*[(0 if t is None else t.data_ptr()) for t in ts])
    _call("tir_mlp_wgrad_multi", arr(cols[0]), arr(cols[1]), arr(cols[2]), arr(cols[3]), arr(cols[4]), arr(cols[5]), int(jobs[0][5].shape[1]),
          arr(cols[6]), (None if null_map else arr(cols[7])), arr(cols[8]), arr(cols[9]), arr(cols[10]), arr(cols[11]),
          arr(cols[12]), arr(cols[13]), k, n, TUNE["wgrad_blocks"], _stream())
    _KEEP_ALIVE = cols          # noqa: F841  (operands stay referenced until the launch is queued)


def gemm_tn(A, M, B, N, C_out, ones_col=False, impl=None, bias_out=None):
    """C_out[M, N(+1)] += A[:, :M]^T @ B[:, :N]  (+ column N = A^T 1, or bias_out[M] += A^T 1 when given).  Split-bf16
    matrix cores when the product's decoder mode is bf16x3 (default), exact fp32 MFMA otherwise."""
    impl = impl or MLP_IMPL
    A, B = f32(A, "A"), f32(B, "B")
    n = A.shape[0]
    if B.shape[0] != n:
        raise ValueError("gemm_tn: row counts differ")
    _call("tir_gemm_tn_bf16x3" if impl == "bf16x3" else "tir_gemm_tn", _ptr(A), A.shape[1], int(M), _ptr(B), B.shape[1], int(N), int(bool(ones_col)), n,
          _ptr(C_out), C_out.shape[1], _ptr(bias_out), _stream())
    return C_out


def adam_step(entries, beta1, beta2, eps):
    """tir_adam_step over entries (p, g, m, v, lr, bias_correction1, bias_correction2); the four tensors of an entry are
    dense with the same element order (tensoir_amd.optim.Adam checks)."""
    n = len(entries)
    PA, FA, LA = C.c_void_p * n, C.c_float * n, C.c_int64 * n
    _call("tir_adam_step", n, PA(*[e[0].data_ptr() for e in entries]), PA(*[e[1].data_ptr() for e in entries]),
          PA(*[e[2].data_ptr() for e in entries]), PA(*[e[3].data_ptr() for e in entries]),
          LA(*[e[0].numel() for e in entries]), FA(*[e[4] for e in entries]), FA(*[e[5] for e in entries]),
          FA(*[e[6] for e in entries]), float(beta1), float(beta2), float(eps), _stream())


def adam_tables(triples):
    """The argument tables of tir_adam_step for a FIXED parameter list [(p, m, v)]: parameter / moment pointers and element counts
    filled in, gradient pointers and the three per-step float columns left for the caller (tensoir_amd.optim.Adam._fast_step).
    -> (p, m, v, numel, g, lr, bias_correction1, bias_correction2) ctypes arrays."""
    n = len(triples)
    PA, FA, LA = C.c_void_p * n, C.c_float * n, C.c_int64 * n
    return (PA(*[t[0].data_ptr() for t in triples]), PA(*[t[1].data_ptr() for t in triples]), PA(*[t[2].data_ptr() for t in triples]),
            LA(*[t[0].numel() for t in triples]), PA(), FA(), FA(), FA())


def adam_step_tables(n, tables, beta1, beta2, eps):
    """tir_adam_step on prepared tables (adam_tables; the library copies them into the launch: they can be refilled at once)."""
    p, m, v, numel, g, lr, b1, b2 = tables
    _call("tir_adam_step", n, p, g, m, v, numel, lr, b1, b2, float(beta1), float(beta2), float(eps), _stream())


def shade_integrate_bwd(maps, rays, dirs, light_idx, vis, indirect, env, weight_d, equal_area, use_srgb, acc_thres,
                        g_out):
    maps = f32(maps, "maps", MAP_STRIDE)
    M, D = maps.shape[0], dirs.shape[0]
    g_maps = torch.empty_like(maps)
    g_env = torch.zeros_like(env)
    _call("tir_shade_integrate_bwd", _ptr(maps), _ptr(f32(rays, "rays", 6)), _ptr(f32(dirs, "dirs", 3)),
          _ptr(i32(light_idx, "light_idx").view(-1)), _ptr(f32(vis, "vis")),
          _ptr(None if indirect is None else f32(indirect, "indirect", 3)), _ptr(f32(env, "env", 3)),
          _ptr(None if weight_d is None else f32(weight_d, "light_area_weight")), M, D, env.shape[0],
          int(bool(equal_area)), int(bool(use_srgb)), float(acc_thres), _ptr(f32(g_out, "g_out", 3)), _ptr(g_maps),
          _ptr(g_env), _stream())
    return g_maps, g_env


def env_sg_bwd(lgtSGs, rot, dirs, g_env):
    sgs = f32(lgtSGs.detach(), "lgtSGs", 7)
    rot = f32(rot, "light_rotation_matrix").view(-1, 9)
    dirs = f32(dirs, "dirs", 3).view(-1, 3)
    g = torch.zeros_like(sgs)
    desc = TirEnvSG(sgs.data_ptr(), rot.data_ptr(), sgs.shape[0], rot.shape[0])
    _call("tir_env_sg_bwd", C.byref(desc), _ptr(dirs), dirs.shape[0], _ptr(f32(g_env, "g_env", 3)), _ptr(g), _stream())
    return g
