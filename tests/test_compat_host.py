"""CPU tests of the HOST side of the libecc-typed boundary (libecc_amd/compat/libecc_amd_compat.c): the marshalling of
nn / prj_pt / ec_key_pair / ec_pub_key, the persistent thread pool, the pack | GPU | unpack pipeline, the grouping by
ec_params, nonce handling (rand hook order, RFC 6979 through libecc's hmac_*) and the error paths.

The GPU entry points are replaced by tests/mock_ecamd.c (a stand-in built on the oracle -- test infrastructure, never part of
the product), and libecc_amd/compat/compat_check.c compares every batch result with libecc's own scalar function, exactly
as it does on the GPU box against the real library (tests/test_gpu_parity.py::test_libecc_typed_boundary_vs_scalar_api)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "compat_check_mock")
HAVE_LIBECC = os.path.exists("/root/reference/src/libsig.h")


def _build():
    if not HAVE_LIBECC:
        pytest.skip("the libecc sources are not here (authoring container only)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "libecc_amd", "compat"), "-j8", "_build/compat.o"] +
                          [], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # the libecc objects of the compat build are reused; build them if this is a fresh tree
    if not os.path.isdir(os.path.join(ROOT, "libecc_amd", "compat", "_build", "obj", "sig")):
        subprocess.call(["make", "-C", os.path.join(ROOT, "libecc_amd", "compat"), "-j8", "-k"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests"), "-f", "Makefile.compat_mock"], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)


def _run(args, **env):
    e = dict(os.environ, COMPAT_CHECK_MOCK="1", **env)
    return subprocess.run([EXE] + args, capture_output=True, text=True, timeout=900, env=e)


def test_every_typed_entry_point_matches_libecc_scalar_functions():
    """all rows of compat_check (edge families included) at a small size: inline path, one thread"""
    _build()
    r = _run(["24"], ECAMD_COMPAT_THREADS="1")
    assert r.returncode == 0, r.stdout[-4000:]
    assert "all ok" in r.stdout
    for row in ("prj_pt_mul_batch", "prj_pt_mul_blind_batch", "ecccdh_derive_secret_batch", "ec_verify_batch ECDSA", "ec_verify_batch EDDSA448",
                "ec_sign_batch ECDSA", "ec_sign_batch DECDSA", "ec_sign_batch EDDSA25519", "ec_key_pair_{gen,import}_batch", "x25519_batch",
                "x448_batch", "foreign generator", "ec_verify_batch BIP0340/SECP256K1", "ec_verify_batch ECFSDSA/SECP256R1", "ECKCDSA (no batch form)"):
        assert row in r.stdout, row


def test_thread_pool_and_pipeline_chunks():
    """one case per family at a size that takes the pool threads and three pipeline chunks of 256 items (pack c+1 | GPU c | unpack c-1); the
    verification calls are streamed (round 4): ONE device call, started at once, that asks the pool through the producer hook for
    each of three ranges of the arrays being packed (the stand-in asks 200 items at a time); nonces and private scalars leave as raw
    random bytes and are reduced behind the C ABI"""
    _build()
    r = _run(["quick", "530"], ECAMD_COMPAT_CHUNK="256", ECAMD_COMPAT_READY_ITEMS="256", ECAMD_COMPAT_THREADS="6")
    assert r.returncode == 0, r.stdout[-4000:]
    assert "all ok" in r.stdout


def test_host_side_fallbacks_of_round_4():
    """the same program with every round-4 path switched back to its host-side predecessor: libecc's hashes, chunked verification
    calls, the two-pass EdDSA verification, nn_get_random_mod on the pool threads, projective keys although Z = 1"""
    _build()
    r = _run(["quick", "300"], ECAMD_COMPAT_HOST_HASH="1", ECAMD_COMPAT_NO_STREAM="1", ECAMD_COMPAT_ED_TWO_PASS="1", ECAMD_COMPAT_HOST_RANDMOD="1",
             ECAMD_COMPAT_PRJ_KEYS="1", ECAMD_COMPAT_THREADS="3", ECAMD_COMPAT_CHUNK="128")
    assert r.returncode == 0, r.stdout[-4000:]
    assert "all ok" in r.stdout


def test_schnorr_batches_try_the_multi_scalar_form_first():
    """BIP0340 / ECFSDSA behind ec_verify_batch with the threshold of the multi-scalar form lowered to one item: every batch whose items
    pass their pre-checks asks ecamd_multi_schnorr_verify_all_batch first (the stand-in answers with the exact conjunction of the item
    form); accepted batches end there, spoiled ones go on to the item-by-item pass and ec_verify_batch_results still matches ec_verify"""
    _build()
    r = _run(["quick", "60"], ECAMD_COMPAT_SCHNORR_MSM_MIN="1", ECAMD_COMPAT_ED_MSM_MIN="1", ECAMD_COMPAT_THREADS="2")
    assert r.returncode == 0, r.stdout[-4000:]
    assert "all ok" in r.stdout and "ec_verify_batch BIP0340/SECP256K1" in r.stdout and "ec_verify_batch ECFSDSA/SECP256R1" in r.stdout
    assert int(r.stdout.split("schnorr multi-scalar calls:")[1].split()[0]) >= 6, r.stdout[-600:]
    # round 6: ec_verify_batch of Ed25519 / Ed25519ctx batches asks for the whole-batch bit first as well (valid batches end there)
    assert int(r.stdout.split("ed25519 whole-batch calls:")[1].split()[0]) >= 4, r.stdout[-600:]


def test_no_device_is_an_error_not_a_fallback():
    _build()
    r = _run(["8"], MOCK_ECAMD_NO_DEVICE="1")
    assert r.returncode == 3 and "no GPU path" in r.stdout
