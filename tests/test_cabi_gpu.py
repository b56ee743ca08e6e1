"""The C ABI from a foreign host program (examples/cabi_smoke.cpp: no Python, no torch in the process): built with hipcc
against include/pmoe_hip.h + libpmoe_hip.so and run as a child process."""
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def _build(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "cabi_smoke"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", str(REPO / "examples" / "cabi_smoke.cpp"), "-I", str(REPO / "include"),
                    "-L", str(REPO / "pmoe_amd"), "-lpmoe_hip", f"-Wl,-rpath,{REPO / 'pmoe_amd'}", "-o", str(exe)],
                   check=True, capture_output=True, timeout=600)
    return exe


def test_header_is_plain_c_and_example_links(tmp_path):
    """CPU: the header compiles as C99 and as C++, and the example builds and links against the in-tree library."""
    for lang, std in (("c", "-std=c99"), ("c++", "-std=c++11")):
        subprocess.run(["gcc", "-fsyntax-only", "-x", lang, std, str(REPO / "include" / "pmoe_hip.h")], check=True)
    assert _build(tmp_path).exists()


@pytest.mark.gpu
def test_foreign_host_program_runs_a_grouped_conv(tmp_path):
    out = subprocess.run([str(_build(tmp_path))], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CABI OK" in out.stdout, out.stdout + out.stderr
