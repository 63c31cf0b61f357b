"""The boundary is a C ABI: include/aomhip.h must be valid C99 on its own (the reference is C, and cgo / JNI / ctypes
bindings parse it as C), and a plain-C host program must build against nothing but the header and libaomhip.so.  On the
GPU box the program runs: batched SAD + the device-side NSTEP search from C, checked inside the program."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_valid_c99(tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "aomhip.h"\nint main(void) { return (int)sizeof(aomhip_planes) == 0; }\n')
    for std in ("c99", "c11", "gnu89"):
        r = subprocess.run(["gcc", "-std=" + std, "-pedantic" if std != "gnu89" else "-Wall", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                            "-I" + os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
        assert r.returncode == 0, (std, r.stderr[:2000])
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]


def test_c_host_program_builds(tmp_path):
    out = tmp_path / "c_host_demo"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_host_demo.c"), "-L" + os.path.join(ROOT, "aom-av1-psy_amd", "lib"), "-laomhip",
                        "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]


@pytest.mark.gpu
def test_c_host_program_runs(tmp_path):
    out = tmp_path / "c_host_demo"
    lib = os.path.join(ROOT, "aom-av1-psy_amd", "lib")
    r = subprocess.run(["gcc", "-std=c99", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_host_demo.c"), "-L" + lib,
                        "-laomhip", "-Wl,-rpath," + lib, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]
    r = subprocess.run([str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "0 SAD mismatches" in r.stdout


def test_c_host_helpers_run_without_a_gpu(tmp_path):
    """The plain-C host helpers the demo checks first (tile columns, exchange plan, temporal-filter block list and parameters) need no
    device: the program gets past them (exit code 3 = a helper check failed) and then stops at the device probe (2) or runs (0)."""
    out = tmp_path / "c_host_demo"
    lib = os.path.join(ROOT, "aom-av1-psy_amd", "lib")
    r = subprocess.run(["gcc", "-std=c99", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_host_demo.c"), "-L" + lib,
                        "-laomhip", "-Wl,-rpath," + lib, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]
    r = subprocess.run([str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 2), (r.returncode, r.stdout, r.stderr)
    assert "host helper check" not in r.stderr
