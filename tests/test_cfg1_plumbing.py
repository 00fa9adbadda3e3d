"""BASELINE.json configs[0] ("secp256r1 prj_pt_mul batch = 1024 ... plumbing") as a named test:
1024 scalar multiplications through the LIBRARY API in libecc's own types -- prj_pt_mul_batch of libsign_amd.so against a loop of
libecc's scalar prj_pt_mul on the same nn / prj_pt structures, item by item (libecc_amd/compat/compat_check.c `cfg1`).
  * CPU leg: the host logic of the layer with the GPU entry points replaced by the oracle-backed stand-in (tests/mock_ecamd.c) --
    what the reference's "ec_self_tests on CPU (plumbing, no GPU)" asks for;
  * GPU leg: the product library itself."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cfg1_plumbing_cpu():
    if not os.path.exists("/root/reference/src/libsig.h") and not os.path.exists(os.path.join(ROOT, "tests", "_build", "compat_check_mock")):
        pytest.skip("needs the reference tree (authoring container) to build the libecc-typed layer")
    import test_compat_host as T
    T._build()
    r = T._run(["cfg1"], ECAMD_COMPAT_THREADS="4")
    assert r.returncode == 0 and "cfg1: all ok" in r.stdout, r.stdout[-2000:]
    assert "1024 items" in r.stdout


@pytest.mark.gpu
def test_cfg1_plumbing_gpu():
    exe = os.path.join(ROOT, "libecc_amd", "lib", "compat_check")
    if not os.path.exists(exe):
        pytest.skip("libecc_amd/lib/compat_check not built (needs the reference tree at build time)")
    r = subprocess.run([exe, "cfg1"], capture_output=True, text=True, timeout=600)
    if r.returncode == 3 and "no GPU path" in r.stdout:
        pytest.skip("no HIP device on this box (a plain `pytest` run on the authoring container)")
    assert r.returncode == 0 and "cfg1: all ok" in r.stdout, r.stdout[-2000:]
    assert "1024 items" in r.stdout
