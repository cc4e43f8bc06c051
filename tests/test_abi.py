"""C-ABI checks that need no GPU: the library loads, exports every symbol include/tensoir_hip.h
declares, the ctypes mirror matches the C struct layout, and argument validation works."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tensoir_hip.h")


@pytest.fixture(scope="module")
def lib():
    from tensoir_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.lib()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tir_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in tensoir_hip.h but not exported"


def test_binding_table_matches_header(lib):
    from tensoir_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    # arity of every prototype
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), f"{name}: header has {n} parameters, binding {len(args)}"


def test_struct_layout_matches_c(tmp_path, lib):
    from tensoir_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "tensoir_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(TirField), offsetof(TirField, grid), offsetof(TirField, dplane),
         offsetof(TirField, basis_t), offsetof(TirField, occ_nbr), offsetof(TirField, occ_dim), offsetof(TirField, occ_inv),
         offsetof(TirField, occ_hi));
  printf("%zu %zu %zu\n", sizeof(TirMlp), offsetof(TirMlp, feat_dim), offsetof(TirMlp, act));
  printf("%zu %zu\n", sizeof(TirEnvSG), offsetof(TirEnvSG, n_sg));
  return 0; }''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    F, M, E = _lib.TirField, _lib.TirMlp, _lib.TirEnvSG
    assert [int(x) for x in out[0].split()] == [C.sizeof(F), F.grid.offset, F.dplane.offset, F.basis_t.offset,
                                                F.occ_nbr.offset, F.occ_dim.offset, F.occ_inv.offset, F.occ_hi.offset]
    assert [int(x) for x in out[1].split()] == [C.sizeof(M), M.feat_dim.offset, M.act.offset]
    assert [int(x) for x in out[2].split()] == [C.sizeof(E), E.n_sg.offset]


def test_version_and_errors(lib):
    assert lib.tir_version() == 100
    assert b"invalid argument" in lib.tir_error_string(-1001)
    assert b"not supported" in lib.tir_error_string(-1002)
    # argument validation happens before any device work, so it is testable without a GPU
    assert lib.tir_pack_plane(None, None, 16, 8, 8, None) == -1001
    assert lib.tir_vm_density_fwd(None, None, None, None, 10, None) == -1001
    assert lib.tir_exclusive_scan(None, None, 4, None) == -1001
    assert lib.tir_exclusive_scan(None, None, 0, None) == -1001   # offsets is always required
    assert lib.tir_mlp_packed_floats(27, 2, 128, 4) > 36000
    assert lib.tir_mlp_packed_floats(27, 6, 128, 4) == -1002      # only pe=2 / 27 / 128 are built
    assert lib.tir_mlp_packed_floats(27, 2, 64, 3) == -1002


def test_round5_entry_points_validate_before_any_device_work(lib):
    """The entries added in round 5 (and the guards ADVICE r4 asked for) reject bad arguments on the host: plane element
    offsets of the fp16 gathers (32-bit, 24-bit index multiplies), tir_pack_half_checked's report buffer, the device-bounded C5
    entries and the compaction / compose kernels."""
    from tensoir_amd import _lib
    import torch
    # a field whose plane would need > 2^31 element offsets / a row pitch beyond the 24-bit multiplier: UNSUPPORTED, not garbage
    keep = torch.zeros(64, dtype=torch.float32)                       # any non-null, 16-byte aligned host address: never dereferenced
    ptr = keep.data_ptr()
    assert ptr % 16 == 0
    m = _lib.TirMlp(ptr, 27, 2, 128, 3, 0, 0)
    for grid, want in (((70000, 70000, 8), -1002), ((400000, 8, 8), -1002)):
        f = _lib.TirField()
        f.grid[:] = grid
        f.n_dcomp, f.n_acomp, f.app_dim, f.n_lights = 16, 48, 27, 1
        f.basis_t = f.light_line = f.light_mean = ptr
        fh = _lib.TirFieldHalf()
        for i in range(3):
            fh.aplane[i] = fh.aline[i] = ptr
        assert lib.tir_vm_app_fwd_h16(C.byref(f), C.byref(fh), ptr, ptr, None, ptr, 32, 0, 10, None, None) == want, grid
        assert lib.tir_indirect_fused_fwd(C.byref(f), C.byref(fh), C.byref(m), ptr, ptr, None, 1, 1, ptr, ptr, 10, None, None) == want, grid
    # tir_pack_half_checked: the report buffer is mandatory, a scan-only table needs it, at most TIR_HALF_MAX_JOBS tables
    one = (C.c_void_p * 1)(ptr)
    none = (C.c_void_p * 1)(None)
    cnt = (C.c_int64 * 1)(8)
    assert lib.tir_pack_half_checked(one, one, cnt, 1, None, None) == -1001
    assert lib.tir_pack_half(one, none, cnt, 1, None) == -1001                    # nothing to write, nowhere to report
    nine = (C.c_void_p * 9)(*([ptr] * 9))
    assert lib.tir_pack_half_checked(nine, nine, (C.c_int64 * 9)(*([8] * 9)), 9, ptr, None) == -1001
    assert lib.tir_pack_half_checked(one, one, (C.c_int64 * 1)(0), 1, ptr, None) == 0          # empty tables: nothing to do
    # C5 chunk entries
    assert lib.tir_surface_compact(None, None, 4, 0.5, None, None, None, None, None, None, None, None, None) == -1001
    assert lib.tir_surface_compact(ptr, ptr, -1, 0.5, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, None) == -1001
    assert lib.tir_env_compose(ptr, 8, 16, ptr, 2, 4, ptr, ptr, ptr, 3, None) == -1001        # dir_stride < 3
    assert lib.tir_env_compose(ptr, 8, 16, ptr, 6, 4, None, ptr, ptr, 3, None) == -1001       # no slot map
    assert lib.tir_env_compose(ptr, 8, 16, ptr, 6, 0, ptr, ptr, ptr, 3, None) == 0            # n = 0
    assert lib.tir_relight_importance_cells_packed_n(ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, 0, 512, ptr, None, None) == 0
    assert lib.tir_relight_importance_cells_packed_n(ptr, ptr, ptr, ptr, ptr, ptr, None, ptr, 4, 512, ptr, None, None) == -1001
    assert lib.tir_env_sample_setup_list_n(ptr, ptr, 8, 16, ptr, 8, ptr, 4, 512, 1, 1, 16, 17, 512, None, None, 0, 0, ptr, ptr, ptr, ptr,
                                           None, None) == -1002                                # 16 x 17 = 272 bins > 255


def test_no_gpu_no_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tensoir_amd import _lib, ops
    assert lib.tir_device_check() == -1003
    with pytest.raises(_lib.TensoirHipError):
        ops.pack_plane(torch.zeros(1, 16, 8, 8))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under tensoir_amd/ (or bench's product leg) may import it."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "tensoir_amd")):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "tensoir_oracle" in txt:
                    bad.append(f)
    assert not bad, bad


def test_hot_kernels_keep_their_register_budget(lib):
    """DESIGN 4.1's occupancy statements as a build check (no GPU): the kernels of the bench step hold no scratch object and fit the
    VGPR budget their waves-per-SIMD figure needs (tools/kernel_resources.py reads the gfx950 code object's metadata).  A PE array
    that became a scratch object cost round 4 0.4 GB of scratch writes per launch; the 1024-thread march spilled one register in
    the first tap-record version of round 6."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from tensoir_amd import _lib
    ks = {k["name"]: k for k in kernel_resources.kernels(_lib.LIB_PATH)}
    budget = {"k_march_secondary_lds<4, 3, 512, true>": 128, "k_march_secondary_lds<4, 3, 1024, true>": 128,      # 4 waves per SIMD
              "k_march_secondary_lds<4, 3, 512, false>": 128, "k_march_secondary_lds<4, 3, 1024, false>": 128,  # (visibility only)
              "k_indirect_fused<12, true>": 168,                                                         # 3 waves per SIMD
              "k_indirect_fused_hp<8>": 256, "k_mlp_bf16_multi<3, false>": 256, "k_vm_app_primary<12, false>": 256,
              "k_march_primary": 256, "k_shade_integrate": 256}
    for name, vgpr in budget.items():
        assert name in ks, (name, sorted(ks)[:5])
        k = ks[name]
        assert k["scratch"] == 0 and k["vgpr_spill"] == 0, k
        assert k["vgpr"] + k["agpr"] <= vgpr, k
    # the only kernels allowed a scratch object: the fp32 VALU fallback decoder, the backward decoders (4 spilled registers), the
    # rarely used hidden-saving aux-table decoder and the 16-lane scatter of the wide (24 / 96-channel) appearance grids
    allowed = ("k_mlp_valu", "k_mlp_bwd_bf16", "k_mlp_bf16_auxt<false, true>", "k_mlp_bf16_auxt<true, true>", "k_vm_app_bwd<")
    bad = [k for k in ks.values() if k["scratch"] and not k["name"].startswith(allowed)]
    assert not bad, bad
