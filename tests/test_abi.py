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
