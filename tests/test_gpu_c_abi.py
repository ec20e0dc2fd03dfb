"""The C ABI from plain C: tests/c_abi/abi_example.cpp (no Python, no torch in the process) is built with hipcc against
include/mcm.h + libmcm_hip.so, runs one pass of the hot path — parameters by HF name, prompt bank once, a batch of
images -> MCM scores, device AUROC / AUPR / FPR95 — and checks the scores against the C oracle in the same process."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    out = str(tmp_path_factory.mktemp("cabi") / "abi_example")
    cmd = [hipcc, "-O2", os.path.join(ROOT, "tests", "c_abi", "abi_example.cpp"), "-I", os.path.join(ROOT, "include"),
           "-L", os.path.join(ROOT, "mcm_amd"), "-lmcm_hip", "-L", os.path.join(ROOT, "oracle"), "-lmcm_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "mcm_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


@pytest.mark.parametrize("precision", [1, 2, 0], ids=["fp32", "fp16", "bf16"])
def test_c_program_scores_match_the_oracle(exe, precision):
    r = subprocess.run([exe, str(precision)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr[-1000:])
    assert r.returncode == 0 and "abi_example OK" in r.stdout, (r.stdout, r.stderr[-2000:])
