"""The C ABI from a plain C program (tests/c/abi_demo.c, built by __graft_entry__.build()): no Python or torch in the
process that calls libtensoir_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c", "abi_demo")


def test_c_client_builds_against_the_header():
    """CPU side: the demo is compiled by build(); the header must stay valid C99."""
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "tensoir_hip.h")])
    assert os.path.exists(os.path.join(ROOT, "tests", "c", "abi_demo.c"))


@pytest.mark.gpu
def test_c_client_runs_on_the_gpu():
    lib = os.path.join(ROOT, "tensoir_amd", "libtensoir_hip.so")
    assert os.path.exists(lib), "libtensoir_hip.so is not built: run __graft_entry__.build()"
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(EXE + ".c"):     # same image on the GPU box: gcc is there
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                               EXE + ".c", "-o", EXE, "-L", os.path.join(ROOT, "tensoir_amd"), "-ltensoir_hip",
                               "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath,$ORIGIN/../../tensoir_amd",
                               "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c-abi demo ok" in out.stdout
