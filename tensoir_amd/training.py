"""Training path: autograd bridges whose forward AND backward are HIP launches.

The reference differentiates its ATen op chain with torch.autograd (train_tensoIR.py:315-317).  Here
torch.autograd only carries gradients *between* three fused stages -- each stage's backward is a chain
of hand-written kernels (tir_*_bwd in include/tensoir_hip.h):

  PrimaryRenderFn : rays -> [B,20] map rows   (TensorBase.forward, models/tensorBase_rotated_lights.py:868-1036)
  EnvSGFn         : lgtSGs -> environment radiance [L,D,3]   (:577-588, :70-86)
  ShadeFn         : map rows + environment -> rgb_with_brdf  (render_with_BRDF, models/relight_utils.py:403-483;
                    visibility / indirect light are constants there: compute_secondary_shading_effects is no_grad)

There is no eager-PyTorch fallback: every derivative below is produced by libtensoir_hip.so.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from . import ops
from ._lib import TensoirHipError, TirFieldGrad


from .field_model import NORMAL_LOSS_KINDS  # noqa: E402


def field_param_list(model):
    """Parameters PrimaryRenderFn differentiates, in the order its backward returns their gradients."""
    ps = list(model.density_plane) + list(model.density_line) + list(model.app_plane) + list(model.app_line)
    ps += [model.basis_mat.weight, model.light_line.weight]
    for dec in _decoders(model):
        m = dec.mlp
        ps += [m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias]
    return ps


def _decoders(model):
    decs = [model.renderModule, model.renderModule_brdf]
    if hasattr(model, "renderModule_normal"):
        decs.append(model.renderModule_normal)
    return decs


def _packed_bwd(dec):
    """Transposed decoder weights in MFMA operand order for tir_mlp_bwd (cached per parameter version)."""
    ps = [dec.mlp[0].weight, dec.mlp[2].weight, dec.mlp[4].weight]
    key = tuple((p.data_ptr(), p._version) for p in ps)
    cache = dec.__dict__.get("_bwd_cache")
    if cache is None or cache[0] != key:
        cache = (key, ops.pack_mlp_bwd(dec.w0_std(), ps[1], ps[2], dec.in_chanel, dec.feape))
        dec.__dict__["_bwd_cache"] = cache
    return cache[1]


def _grad_buffers(model, field):
    """Zero-filled gradient buffers in the packed channel-last layout + their TirFieldGrad descriptor.  One allocation and
    ONE fill launch for all fourteen: each buffer is a view of the flat block with the strides of what the kernels gather
    from (the channel-last parameter itself or its packed shadow)."""
    keep = model._field_cache
    names = [f"{name}{i}" for i in range(3) for name in ("dp", "dl", "ap", "al")] + ["ll", "lm"]
    srcs = [keep[n] for n in names]
    offs, total = [], 0
    for t in srcs:
        offs.append(total)
        total += (t.numel() + 3) // 4 * 4                 # 16-byte aligned starts
    flat = torch.zeros((total,), dtype=torch.float32, device=srcs[0].device)
    bufs = {n: torch.as_strided(flat, t.shape, t.stride(), o) for n, t, o in zip(names, srcs, offs)}
    g = TirFieldGrad()
    for i in range(3):
        for name, dst in (("dp", g.dplane), ("dl", g.dline), ("ap", g.aplane), ("al", g.aline)):
            dst[i] = bufs[f"{name}{i}"].data_ptr()
    g.light_line, g.light_mean = bufs["ll"].data_ptr(), bufs["lm"].data_ptr()
    bufs["desc"] = g
    return bufs


def _to_param_layout(g):
    """Gradient buffer -> parameter shape [1,C,H,W].  Buffers are zeros_like() of what the kernels gather from: the
    channel-last parameter itself (already the right shape and strides: autograd takes it without a copy) or a packed
    [H,W,C] shadow (a permuted view)."""
    return g if g.dim() == 4 else g.permute(2, 0, 1).unsqueeze(0)


class _DecoderCall(SimpleNamespace):
    pass


class _LeafStream:
    """The weight-gradient products (dW = X^T dZ per layer, the basis-matrix gradient) are leaves of the backward: nothing
    in the chain waits for them.  They are HBM-streaming kernels with an atomic epilogue, the chain between them is
    scatter- and matrix-bound: queued on a second HIP stream they fill the chain's tails instead of extending it.
    `run` orders the leaf after everything queued so far; `join` makes the caller's stream wait for all leaves.
    Operands are kept referenced until the join, so the caching allocator cannot hand their memory to a later kernel of
    the main stream while a leaf still reads it.  TENSOIR_BWD_STREAMS=0: everything on the caller's stream."""
    _streams = {}

    def __init__(self, dev):
        self.on = os.environ.get("TENSOIR_BWD_STREAMS", "1") != "0" and not torch.cuda.is_current_stream_capturing()
        self.keep = []
        self.dev = dev              # every stream query below names the device: the model may live on a non-current GPU
        if self.on:
            key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
            if key not in self._streams:
                self._streams[key] = torch.cuda.Stream(device=dev)
            self.side = self._streams[key]

    def run(self, fn, *operands):
        if not self.on:
            return fn()
        self.keep.extend(operands)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            out = fn()
        self.keep.append(out)
        return out

    def join(self):
        if self.on:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
        self.keep.clear()


_DEC_GRAD_SIZES = (128 * 150, 128, 128 * 128, 128, 4 * 128, 4)
_DEC_GRAD_SHAPES = ((128, 150), (128,), (128, 128), (128,), (4, 128), (4,))
_DEC_GRAD_FLOATS = sum(_DEC_GRAD_SIZES)


def _decoder_backward(dec, calls, impl=None, leaf=None, bwd=None, wg=None, keep_dz1=None, flat=None):
    """Backward of every recorded invocation of one decoder.  Returns ([g_feat per call], 6 parameter grads).
    bwd: the (g_feat, dz1, dz2, dz3) of each call when the backward-data kernel has already run (merged launch).
    wg: a list -- the weight-gradient work of every call is appended to it as a job of ops.mlp_wgrad_multi (the caller
    launches all of the stage's jobs at once) instead of being issued here as tir_mlp_inputs + three tir_gemm_tn."""
    pm, pb = dec.packed(), _packed_bwd(dec)
    dev = calls[0].feat.device
    od = pm.out_dim
    # one zeroed buffer (flat: the caller's, zeroed for all of the stage's decoders with one fill launch), carved into the six
    # exact-shape (contiguous) gradients: autograd takes them as they are
    if flat is None:
        flat = torch.zeros((_DEC_GRAD_FLOATS,), dtype=torch.float32, device=dev)
    dW0, db0, dW1, db1, dW2, db2 = (t.view(shape) for t, shape in zip(torch.split(flat, _DEC_GRAD_SIZES), _DEC_GRAD_SHAPES))
    g_feats = []
    for ci, c in enumerate(calls):
        if bwd is None:
            g_feat, dz1, dz2, dz3 = ops.mlp_bwd(pm, pb, c.feat, c.out, c.g_out, c.h1, c.h2, impl=impl)
        else:
            g_feat, dz1, dz2, dz3 = bwd[ci]
        if keep_dz1 is not None:
            keep_dz1.append(dz1)

        def weight_grads(c=c, dz1=dz1, dz2=dz2, dz3=dz3):
            x = ops.mlp_inputs(pm, c.feat, c.aux, c.aux_map)
            ops.gemm_tn(dz1, 128, x, 150, dW0, True, impl=impl, bias_out=db0)
            ops.gemm_tn(dz2, 128, c.h1, 128, dW1, True, impl=impl, bias_out=db1)
            ops.gemm_tn(dz3, 4, c.h2, 128, dW2, True, impl=impl, bias_out=db2)
            return x

        if wg is not None:
            wg.append((dz1, dz2, dz3, c.h1, c.h2, c.feat, c.aux, c.aux_map, dW0, db0, dW1, db1, dW2, db2))
        elif leaf is None:
            weight_grads()
        else:
            leaf.run(weight_grads, dz1, dz2, dz3, c.feat, c.aux, c.aux_map, c.h1, c.h2, flat)
        g_feats.append(g_feat)
    grads = [dW0, db0, dW1, db1, dW2[:od], db2[:od]]
    return g_feats, grads


def _merged_backward_data(model, calls):
    """The backward-data kernel of all recorded decoder invocations of the stage in ONE launch (same rows for every
    call): {call name: (g_feat, dz1, dz2, dz3)}, or None when the merged launch does not apply."""
    if ops.MLP_IMPL != "bf16x3" or len(calls) < 2 or len(calls) > 4:
        return None
    decs = {"rgb": model.renderModule, "brdf": model.renderModule_brdf, "brdf_j": model.renderModule_brdf,
            "normal": getattr(model, "renderModule_normal", None)}
    names = [n for n in ("rgb", "brdf", "brdf_j", "normal") if n in calls]
    if any(calls[n].feat.shape[1] != ops.FEAT_STRIDE for n in names):
        return None
    jobs = [(decs[n].packed(), _packed_bwd(decs[n]), calls[n].feat, calls[n].out, calls[n].g_out, calls[n].h1, calls[n].h2)
            for n in names]
    return dict(zip(names, ops.mlp_bwd_multi(jobs)))


class _CapacityOverflow(RuntimeError):
    """The record buffers sized from the previous step were too small: the forward is re-run with an exact count."""


class _FinishOnce:
    """Deferred end-of-forward bookkeeping of one PrimaryRenderFn pass (runs at most once)."""

    def __init__(self, st):
        self.st, self.result = st, None

    def __call__(self):
        if self.result is None:
            fin, self.st.finish = self.st.finish, None
            self.result = True if fin is None else fin()
        return self.result


def _trim_state(st, total):
    """Capacity-sized saved buffers -> exact [:total] views for the backward."""
    st.A = total
    for name in ("rec_ray", "rec_k", "rec_w", "rec_xyz", "rgb", "brdf", "brdf_j", "pred", "derived", "xyz_j"):
        t = getattr(st, name, None)
        if t is not None:
            setattr(st, name, t[:total])
    for c in st.calls.values():
        for name in ("feat", "aux", "out", "h1", "h2"):
            t = getattr(c, name)
            if t is not None and not (name == "aux" and c.aux_map is not None):
                setattr(c, name, t[:total])
        if c.aux_map is not None:
            c.aux_map = c.aux_map[:total]


class PrimaryRenderFn(torch.autograd.Function):
    """march -> scan -> compact -> appearance gather -> decoders -> analytic normals -> composite, with a
    hand-written backward for every link."""

    @staticmethod
    def forward(ctx, model, rays, lidx, S, white_bg, is_relight, jitter, noise_dense, defer, *params):
        """Record capacity as in the inference forward (field_model.forward): the number A of w > thres samples lives
        on the device.  With a capacity learnt from the previous step the buffers are sized cap rows, every kernel is
        bounded by the device-side count, and A is read once everything has been queued (`finish`; when `defer` the
        caller -- Renderer_TensoIR_train -- calls it after its shading stage is queued too).  The backward runs on
        the exact [:A] views."""
        f = model.packed_field()
        dev = rays.device
        B = rays.shape[0]
        weight, sigma, acc, depth, _tend, cnt = ops.march_primary_train(f, rays, jitter, S, model.march_t_stop)
        hints = model.__dict__.setdefault("_train_cap_hints", {})
        cap = hints.get((B, S)) if noise_dense is None else None
        if cap is None:
            offsets = ops.exclusive_scan(cnt)
            A = int(offsets[-1].item())                    # first step of a batch shape: one host sync mid-pass
            n_dev = total_dev = None
        else:
            offsets, total_dev = ops.exclusive_scan_capped(cnt, cap)
            A, n_dev = cap, offsets[B:]
            total_host = ops.AsyncCount(total_dev)       # on its way to the host while the rest of the pass is queued
        rec_ray, rec_k, rec_w, rec_xyz = ops.compact_primary(f, rays, jitter, weight, offsets, A)
        st = SimpleNamespace(model=model, rays=rays, lidx=lidx, S=S, white_bg=white_bg, is_relight=is_relight,
                             jitter=jitter, weight=weight, sigma=sigma, acc=acc, depth=depth, offsets=offsets, A=A,
                             rec_ray=rec_ray, rec_k=rec_k, rec_w=rec_w, rec_xyz=rec_xyz, calls={}, n_params=len(params),
                             xyz_j=None, valid=True)
        rgb = brdf = brdf_j = pred = derived = None
        if A > 0:
            viewdirs = rays[:, 3:6].contiguous()
            merged = bool(is_relight) and noise_dense is None and f.n_acomp == 48 and ops.APP_IMPL == "mfma"
            xyz_j = intr_j = None
            if merged:
                # records + jittered records in one gather launch; the noise is drawn in the kernel, keyed by (seed, pass
                # counter, record index): the draw does not depend on the record-capacity hint, so a fixed torch seed
                # reproduces the run
                rng_state = model._jitter_rng(dev)
                rad, intr, xyz_j, intr_j = ops.vm_app_primary(f, rec_xyz, lidx, rec_ray, 0.01, rng_state, n_dev, exact=True)
                rng_state[1] += 1
            else:
                rad, intr = ops.vm_app(f, rec_xyz, lidx, rec_ray, True, bool(is_relight), ops.APP_IMPL, 0, n_dev)      # (exact contraction: saved for the backward)
                if is_relight and noise_dense is not None:
                    noise = noise_dense.to(dev, torch.float32)[rec_ray.long(), rec_k.long()]
                    xyz_j = torch.add(rec_xyz, noise, alpha=0.01)
                    intr_j = ops.vm_app(f, xyz_j, None, None, False, True, ops.APP_IMPL, 0, n_dev)[1]
                elif is_relight:
                    rng_state = model._jitter_rng(dev)
                    xyz_j, intr_j = ops.vm_app_jitter(f, rec_xyz, 0.01, 0, 0, rng_state, n_dev)
                    rng_state[1] += 1
            # the decoders of the stage: (name, module, features, aux, aux_map)
            jobs = [("rgb", model.renderModule, rad, viewdirs, rec_ray)]
            if is_relight:
                jobs.append(("brdf", model.renderModule_brdf, intr, rec_xyz, None))
                jobs.append(("brdf_j", model.renderModule_brdf, intr_j, xyz_j, None))
                if model.normals_kind in ("purely_predicted", "derived_plus_predicted"):
                    jobs.append(("normal", model.renderModule_normal, intr, rec_xyz, None))
            if ops.MLP_IMPL == "bf16x3" and ops.FEAT_STRIDE == rad.shape[1]:
                # same records for every decoder: ONE launch, the grid split between them
                res = ops.mlp_multi([(d.packed(), ft, ax, mp) for _, d, ft, ax, mp in jobs], n_dev, save_hidden=True)
            else:
                res = [ops.mlp_train(d.packed(), ft, ax, mp, n_dev=n_dev) for _, d, ft, ax, mp in jobs]
            outs = {}
            for (name, _, ft, ax, mp), (o, h1, h2) in zip(jobs, res):
                outs[name] = o
                st.calls[name] = _DecoderCall(feat=ft, aux=ax, aux_map=mp, out=o, h1=h1, h2=h2)
            rgb, brdf, brdf_j = outs["rgb"], outs.get("brdf"), outs.get("brdf_j")
            if is_relight:
                st.xyz_j = xyz_j
                if model.normals_kind == "purely_derived":
                    pred = ops.density_grad(f, rec_xyz, n_dev=n_dev)[2]
                elif model.normals_kind == "gt_normals":
                    pred = None
                elif model.normals_kind == "residue_prediction":      # :962-968: its decoder also takes the derived normal
                    derived = ops.density_grad(f, rec_xyz, n_dev=n_dev)[2]
                    pred, h1, h2 = model.renderModule_normal.rows(rec_xyz, derived, intr, n_dev, save_hidden=True)
                    st.calls["normal"] = _DecoderCall(feat=intr, aux=rec_xyz, aux_map=None, out=pred, h1=h1, h2=h2)
                else:
                    pred = outs["normal"]
                    if model.normals_kind == "derived_plus_predicted":
                        derived = ops.density_grad(f, rec_xyz, n_dev=n_dev)[2]
        st.rgb, st.brdf, st.brdf_j, st.pred, st.derived = rgb, brdf, brdf_j, pred, derived
        maps = ops.composite_primary(rays, offsets, rec_w, rgb, brdf, brdf_j, pred, derived, acc, depth,
                                     white_bg, is_relight, model.fixed_fresnel)
        if model.normals_kind not in NORMAL_LOSS_KINDS and is_relight:
            maps[:, 16] = 0.0            # only the predict-and-derive branches fill it (tensorBase_rotated_lights.py:953-968)

        def finish():
            """Read the record count; trim the saved rows to it.  False = the capacity overflowed (re-run the pass)."""
            if total_dev is None:
                total = A
            else:
                total = total_host.get()
                if total > cap:
                    hints.pop((B, S), None)            # next call takes the exact (synchronising) route
                    st.valid = False
                    return False
                _trim_state(st, total)
            if len(hints) > 64:
                hints.clear()
            if noise_dense is None:
                hints[(B, S)] = min(max(int(total * 1.25) + 4096, 1 << 14, int(0.97 * hints.get((B, S), 0))), B * S)      # decays slowly: a heavy batch after a light one must not overflow
            return True

        st.finish = None if total_dev is None else finish
        if total_dev is None:
            finish()
        elif defer:
            model.__dict__["_pending_primary"] = _FinishOnce(st)
        elif not _FinishOnce(st)():
            raise _CapacityOverflow()
        ctx.st = st
        return maps

    @staticmethod
    def backward(ctx, g_maps):
        st = ctx.st
        if st is None:
            raise TensoirHipError("backward through the same primary render twice is not supported: the saved activations "
                                  "are released after the first backward (combine the losses and call backward once)")
        model = st.model
        pend = model.__dict__.get("_pending_primary")
        if isinstance(pend, _FinishOnce) and pend.st is st and st.finish is None:
            model.__dict__.pop("_pending_primary", None)       # nothing left to check: do not keep the activations alive
        if st.finish is not None:                          # nobody ran the deferred check: do it now
            pend = model.__dict__.get("_pending_primary")
            ok = pend() if isinstance(pend, _FinishOnce) and pend.st is st else _FinishOnce(st)()
            if not ok:
                raise TensoirHipError("record capacity overflowed in the training forward and the pass was not re-run")
        if not st.valid:
            raise TensoirHipError("backward through a training forward whose record capacity overflowed")
        f = model.packed_field()
        g_maps = g_maps.contiguous().to(torch.float32)
        if model.normals_kind not in NORMAL_LOSS_KINDS and st.is_relight:
            g_maps = g_maps.clone()
            g_maps[:, 16] = 0.0
        bufs = _grad_buffers(model, f)
        gd = bufs["desc"]
        dev = st.rays.device
        (g_rgb, g_brdf, g_brdf_j, g_pred, g_der, g_weight, g_acc, g_depth) = ops.composite_primary_bwd(
            st.rays, st.offsets, st.rec_k, st.rec_w, st.rgb, st.brdf, st.brdf_j, st.pred, st.derived, st.acc, st.depth,
            st.S, st.white_bg, st.is_relight, model.fixed_fresnel, g_maps)
        dec_grads = {}
        leaf = _LeafStream(dev)
        # the basis-matrix gradient and the three decoders' gradient blocks: ONE allocation, one fill launch
        nbm = model.app_dim * 3 * f.n_acomp
        nbm_pad = (nbm + 3) // 4 * 4
        if os.environ.get("TENSOIR_MERGED_ZEROS", "1") != "0":
            small = torch.zeros((nbm_pad + 3 * _DEC_GRAD_FLOATS,), dtype=torch.float32, device=dev)
            d_basis = small[:nbm].view(model.app_dim, 3 * f.n_acomp)
            dflat = [small[nbm_pad + i * _DEC_GRAD_FLOATS: nbm_pad + (i + 1) * _DEC_GRAD_FLOATS] for i in range(3)]
        else:
            d_basis = torch.zeros((model.app_dim, 3 * f.n_acomp), dtype=torch.float32, device=dev)
            dflat = [None, None, None]
        if st.A > 0:
            c = st.calls["rgb"]
            c.g_out = g_rgb
            if st.is_relight:
                st.calls["brdf"].g_out, st.calls["brdf_j"].g_out = g_brdf, g_brdf_j
                if "normal" in st.calls:
                    st.calls["normal"].g_out = g_pred
            bd = _merged_backward_data(model, st.calls)      # one launch for the stage's decoders (None: one per call)
            pick = lambda *names: None if bd is None else [bd[n] for n in names]
            # the stage's weight gradients in ONE launch on the leaf stream (TENSOIR_FUSED_WGRAD=0: per call, per layer)
            wg = [] if (ops.MLP_IMPL == "bf16x3" and os.environ.get("TENSOIR_FUSED_WGRAD", "1") != "0" and
                        all(cc.feat.shape[1] == ops.FEAT_STRIDE for cc in st.calls.values())) else None
            (g_rad,), dec_grads["rgb"] = _decoder_backward(model.renderModule, [c], leaf=leaf, bwd=pick("rgb"), wg=wg, flat=dflat[0])
            g_int = g_int_j = None
            if st.is_relight:
                cb, cj = st.calls["brdf"], st.calls["brdf_j"]
                (g_int, g_int_j), dec_grads["brdf"] = _decoder_backward(model.renderModule_brdf, [cb, cj], leaf=leaf,
                                                                        bwd=pick("brdf", "brdf_j"), wg=wg, flat=dflat[1])
                if "normal" in st.calls:
                    cn = st.calls["normal"]
                    residue = model.normals_kind == "residue_prediction"
                    dz1 = [] if residue else None
                    (g_n,), dec_grads["normal"] = _decoder_backward(model.renderModule_normal, [cn], leaf=leaf, bwd=pick("normal"),
                                                                    wg=wg, keep_dz1=dz1, flat=dflat[2])
                    g_int = g_int + g_n
                    if residue:
                        # the three derived-normal columns of layer 1 ride outside the kernels (they entered through the per-row
                        # table): their weight gradient dz1^T n and the cotangent of the derived normal dz1 W0[:, 3:6]
                        wn = model.renderModule_normal.w0_normal()
                        residue_dw = dz1[0].t() @ st.derived
                        g_der = g_der + dz1[0] @ wn
                    if g_der is not None:
                        ops.density_grad_bwd(f, gd, st.rec_xyz, g_der)
                elif model.normals_kind == "purely_derived":       # the composited normal IS the derived one
                    ops.density_grad_bwd(f, gd, st.rec_xyz, g_pred)
            y_rad, y_int = ops.vm_app_bwd(f, gd, st.rec_xyz, st.lidx, st.rec_ray, g_rad, g_int)
            nb = 3 * f.n_acomp
            small = ops.MLP_IMPL == "bf16x3" and model.app_dim <= 32 and nb <= 160      # d basis_mat: one launch per gather pass
            if g_int is None:
                pairs = [(g_rad, y_rad)]
            else:
                pairs = [(g_rad, y_rad), (g_int, y_int)]
            if small:
                leaf.run(lambda: ops.gemm_tn_small(pairs, model.app_dim, nb, d_basis), *[t for pr in pairs for t in pr], d_basis)
            else:
                for ga, ya in pairs:
                    leaf.run(lambda ga=ga, ya=ya: ops.gemm_tn(ga, model.app_dim, ya, nb, d_basis), ga, ya, d_basis)
            if g_int is not None:
                _, y_j = ops.vm_app_bwd(f, gd, st.xyz_j, None, None, None, g_int_j)
                if small:
                    leaf.run(lambda: ops.gemm_tn_small([(g_int_j, y_j)], model.app_dim, nb, d_basis), g_int_j, y_j)
                else:
                    leaf.run(lambda: ops.gemm_tn(g_int_j, model.app_dim, y_j, nb, d_basis), g_int_j, y_j)
            if wg:
                # the stage's weight gradients, queued on the leaf stream only NOW: a weight-gradient workgroup owns its CU's
                # whole register file, so launched before the appearance scatter (as until round 3) it kept the scatter's
                # projection kernels waiting for CUs -- 0.80 -> 0.55 ms for the scatter passes, 4.0 -> 3.5-3.9 ms per step;
                # here it overlaps the density backward and the framework's gradient bookkeeping instead
                leaf.run(lambda: ops.mlp_wgrad_multi(wg), *[t for job in wg for t in job if t is not None])
        ops.march_primary_bwd(f, gd, st.rays, st.jitter, st.sigma, st.weight, g_weight, g_acc, g_depth)
        leaf.join()
        grads = []
        for name in ("dp", "dl", "ap", "al"):
            grads += [_to_param_layout(bufs[f"{name}{i}"]) for i in range(3)]
        grads.append(d_basis)
        grads.append(torch.add(bufs["ll"], bufs["lm"][None, :], alpha=1.0 / float(model.light_num)))
        if st.A > 0 and model.normals_kind == "residue_prediction" and "normal" in dec_grads:
            # [128,150] in the kernels' column order + the three normal columns -> the module's [128,153] layout
            dec = model.renderModule_normal
            full = torch.empty_like(dec.mlp[0].weight)                 # every column is written below: no fill launch
            full.index_copy_(1, dec.std_cols_index(full.device), dec_grads["normal"][0])
            full[:, 3:6] = residue_dw
            dec_grads["normal"][0] = full
        for key, dec in zip(("rgb", "brdf", "normal"), _decoders(model)):
            grads += dec_grads.get(key, [None] * 6)
        assert len(grads) == st.n_params
        ctx.st = None
        return (None,) * 9 + tuple(grads)


class EnvSGFn(torch.autograd.Function):
    """get_light_rgbs for spherical Gaussians (models/tensorBase_rotated_lights.py:577-588, :70-86)."""

    @staticmethod
    def forward(ctx, lgtSGs, rot, dirs):
        ctx.save_for_backward(lgtSGs, rot, dirs)
        return ops.env_sg(lgtSGs, rot, dirs)

    @staticmethod
    def backward(ctx, g_env):
        lgtSGs, rot, dirs = ctx.saved_tensors
        return ops.env_sg_bwd(lgtSGs, rot, dirs, g_env.contiguous()), None, None


class EnvPixelFn(torch.autograd.Function):
    """get_light_rgbs for light_kind == 'pixel' (models/tensorBase_rotated_lights.py:585-605)."""

    @staticmethod
    def forward(ctx, light_rgbs, rot, dirs, H, W):
        ctx.save_for_backward(light_rgbs, rot, dirs)
        ctx.hw = (H, W)
        return ops.env_pixel(light_rgbs, H, W, rot, dirs)

    @staticmethod
    def backward(ctx, g_env):
        light_rgbs, rot, dirs = ctx.saved_tensors
        return ops.env_pixel_bwd(light_rgbs, ctx.hw[0], ctx.hw[1], rot, dirs, g_env.contiguous()), None, None, None, None


class ShadeFn(torch.autograd.Function):
    """GGX x (visibility * environment + indirect) x cosine integration + tone map
    (models/relight_utils.py:452-480), differentiable w.r.t. the map rows and the environment radiance."""

    @staticmethod
    def forward(ctx, maps, env, rays, dirs, light_idx, vis, indirect, area, equal_area, use_srgb, acc_thres):
        ctx.save_for_backward(maps, env, rays, dirs, light_idx, vis, indirect, area)
        ctx.flags = (equal_area, use_srgb, acc_thres)
        return ops.shade_integrate(maps, rays, dirs, light_idx, vis, indirect, env, area, equal_area, use_srgb,
                                   acc_thres)

    @staticmethod
    def backward(ctx, g_out):
        maps, env, rays, dirs, light_idx, vis, indirect, area = ctx.saved_tensors
        equal_area, use_srgb, acc_thres = ctx.flags
        g_maps, g_env = ops.shade_integrate_bwd(maps, rays, dirs, light_idx, vis, indirect, env, area, equal_area,
                                                use_srgb, acc_thres, g_out.contiguous())
        return (g_maps, g_env) + (None,) * 9


def wants_grad(model) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
