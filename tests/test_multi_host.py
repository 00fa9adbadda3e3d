"""CPU test of libecc_amd/csrc/ecamd_multi.cpp ITSELF (the multi-GPU sharding layer of the C ABI): the file is
compiled for the host by tests/multi_host_shim.cpp with every single-device entry point replaced by a recorder, and every
ecamd_multi_* batch function is called with fake base addresses for {1, 2, 3, 8} ranks, n below the rank count, uneven and even n, on
curves of every length class (WEI448: 56-octet coordinates but 57 / 114-octet EdDSA encodings; SECP224K1: 29-octet order on a 28-octet
field; SECP521R1: 66).  Each array must advance by exactly its item size (restated here from include/libecc_amd.h's comments), the
shards must tile [0, n) in rank order, and NULL stays NULL.  Round 4 shipped OFF(sigs, 64) for 114-octet Ed448 signatures because the
CPU stand-in tests/mock_ecamd.c replaces this whole file; this test is the one that would have caught it."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")

CURVES = {  # name -> (coordinate octets, order octets)
    "SECP192R1": (24, 24), "SECP224K1": (28, 29), "SECP256R1": (32, 32), "SECP384R1": (48, 48), "SECP521R1": (66, 66),
    "WEI25519": (32, 32), "WEI448": (56, 56), "BRAINPOOLP512R1": (64, 64),
}
EDDSA_CURVES = ("WEI25519", "WEI448")


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "multi_host.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                           "-o", so, os.path.join(ROOT, "tests", "multi_host_shim.cpp"), "-lpthread"])
    L = C.CDLL(so)
    L.mh_record.restype = C.c_char_p
    L.mh_ptr.restype = C.c_uint64
    L.mh_int.restype = C.c_int64
    L.ecamd_last_error.restype = C.c_char_p
    return L


class Arr:
    """a fake array: base address + the item size its shard offset must follow (0 = a NULL pointer)"""
    _next = [1 << 40]

    def __init__(self, item):
        self.item = item
        if item:
            self.base = Arr._next[0]
            Arr._next[0] += 1 << 36
        else:
            self.base = 0

    @property
    def arg(self):
        return C.c_void_p(self.base or None)


def u32(v):
    return C.c_uint32(v)


def cases(cl, ql, eddsa):
    """(multi function, argument list after (m, curve, [op,] n), the single-device function it must call).  An Arr marks a sharded array"""
    kl = 57 if cl == 56 else cl  # EdDSA encoding octets (sig/eddsa.c: EDDSA_R_LEN)
    out = []
    A = Arr
    for slen in (ql, ql + 9):
        out.append(("ecamd_multi_prj_pt_mul_batch", [A(slen), u32(slen), A(2 * cl), A(2 * cl), A(1)], "ec_prj_pt_mul_batch", None))
    for fi in (0, 1):
        for fo in (0, 1):
            il, ol = (3 if fi else 2) * cl, (3 if fo else 2) * cl
            out.append(("ecamd_multi_prj_pt_mul_batch_fmt", [A(ql), u32(ql), A(il), C.c_int(fi), A(ol), C.c_int(fo), A(1)], "ec_prj_pt_mul_batch_fmt", None))
            out.append(("ecamd_multi_prj_pt_unique_batch", [A(il), C.c_int(fi), A(ol), C.c_int(fo), A(1)], "ec_prj_pt_unique_batch", None))
            out.append(("ecamd_multi_prj_pt_op_batch_fmt", [A(il), A(il), C.c_int(fi), A(ol), C.c_int(fo), A(1)], "ec_prj_pt_op_batch_fmt", 0))
            out.append(("ecamd_multi_prj_pt_op_batch_fmt", [A(il), A(0), C.c_int(fi), A(0), C.c_int(fo), A(1)], "ec_prj_pt_op_batch_fmt", 2))
            out.append(("ecamd_multi_prj_pt_op_batch_fmt", [A(il), A(0), C.c_int(fi), A(ol), C.c_int(fo), A(1)], "ec_prj_pt_op_batch_fmt", 3))
            # prj_pt_cmp / prj_pt_eq_or_opp: the output advances by ONE byte per item whatever out_fmt says
            out.append(("ecamd_multi_prj_pt_op_batch_fmt", [A(il), A(il), C.c_int(fi), A(1), C.c_int(fo), A(1)], "ec_prj_pt_op_batch_fmt", 4))
            out.append(("ecamd_multi_prj_pt_op_batch_fmt", [A(il), A(il), C.c_int(fi), A(1), C.c_int(fo), A(1)], "ec_prj_pt_op_batch_fmt", 5))
            out.append(("ecamd_multi_prj_pt_unprotected_mult_batch", [A(ql + 8), u32(ql + 3), u32(ql + 8), A(il), C.c_int(fi), A(ol), C.c_int(fo), A(1)],
                        "ec_prj_pt_unprotected_mult_batch", None))
            out.append(("ecamd_multi_ecdsa_verify_batch_fmt", [A(il), C.c_int(fi), A(2 * ql), A(20 + fo), u32(20 + fo), A(1)], "ec_ecdsa_verify_batch_fmt", None))
            out.append(("ecamd_multi_ecdsa_verify_msg_batch_fmt", [A(il), C.c_int(fi), A(2 * ql), C.c_int(2), A(100 + 4 * fo), u32(100 + 4 * fo), A(1)],
                        "ec_ecdsa_verify_msg_batch_fmt", None))
    out.append(("ecamd_multi_prj_pt_add_batch", [A(2 * cl), A(2 * cl), A(2 * cl), A(1)], "ec_prj_pt_add_batch", None))
    for dl in (28, 32, 64):
        out.append(("ecamd_multi_ecdsa_verify_batch", [A(2 * cl), A(2 * ql), A(dl), u32(dl), A(1)], "ec_ecdsa_verify_batch", None))
        out.append(("ecamd_multi_ecdsa_sign_batch", [A(ql), A(ql), A(dl), u32(dl), A(2 * ql), A(1)], "ec_ecdsa_sign_batch", None))
    out.append(("ecamd_multi_ecdsa_sign_msg_batch", [A(ql), A(2 * ql), C.c_int(2), A(72), u32(72), A(2 * ql), A(1)], "ec_ecdsa_sign_msg_batch", None))
    out.append(("ecamd_multi_key_pair_gen_raw_batch", [A(2 * ql), A(ql), A(2 * cl), A(1)], "ec_key_pair_gen_raw_batch", None))
    out.append(("ecamd_multi_ecccdh_derive_batch", [A(ql), A(2 * cl), A(cl), A(1)], "ec_ecccdh_derive_batch", None))
    if eddsa:
        out.append(("ecamd_multi_xdh_batch", [A(cl), A(cl), A(cl), A(1)], "ec_xdh_batch", None))
        out.append(("ecamd_multi_eddsa_verify_batch", [A(kl), A(2 * kl), A(2 * kl), u32(2 * kl), A(1)], "ec_eddsa_verify_batch", None))
        out.append(("ecamd_multi_eddsa_verify_msg_batch", [A(kl), A(2 * kl), A(200), u32(200), A(1)], "ec_eddsa_verify_msg_batch", None))
        out.append(("ecamd_multi_eddsa_verify_msg_prj_batch", [A(3 * cl), A(2 * kl), A(208), u32(208), u32(kl), A(1)], "ec_eddsa_verify_msg_prj_batch", None))
        out.append(("ecamd_multi_eddsa_verify_ph_prj_batch", [A(3 * cl), A(2 * kl), A(216), u32(216), u32(kl), A(132), u32(132), A(1)],
                    "ec_eddsa_verify_ph_prj_batch", None))
        out.append(("ecamd_multi_eddsa_encode_point_batch", [A(3 * cl), A(kl), A(1)], "ec_eddsa_encode_point_batch", None))
        out.append(("ecamd_multi_eddsa_sign_R_batch", [A(2 * kl), A(kl), A(1)], "ec_eddsa_sign_R_batch", None))
        out.append(("ecamd_multi_eddsa_sign_S_batch", [A(2 * kl), A(2 * kl), A(kl), A(kl)], "ec_eddsa_sign_S_batch", None))
    return out


def records(lib):
    out = []
    for i in range(lib.mh_count()):
        rank, n, nptr, nint = C.c_int(), C.c_uint32(), C.c_int(), C.c_int()
        fn = lib.mh_record(i, C.byref(rank), C.byref(n), C.byref(nptr), C.byref(nint)).decode()
        out.append((fn, rank.value, n.value, [lib.mh_ptr(i, k) for k in range(nptr.value)], [lib.mh_int(i, k) for k in range(nint.value)]))
    return out


def make_multi(lib, nranks, curve):
    lib.mh_reset()
    m = C.c_void_p()
    devs = (C.c_int * nranks)(*([0] * nranks))  # one device listed nranks times: one context and one shard each
    assert lib.ecamd_multi_create(C.byref(m), devs, nranks) == 0
    assert lib.ecamd_multi_size(m) == nranks
    mc = C.c_void_p()
    assert lib.ecamd_multi_curve_by_name(m, curve.encode(), C.byref(mc)) == 0
    return m, mc


def shard_bounds(n, N):
    return [(n * r // N, n * (r + 1) // N) for r in range(N)]


@pytest.mark.parametrize("curve", sorted(CURVES))
@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_every_array_advances_by_its_item_size(lib, curve, nranks):
    cl, ql = CURVES[curve]
    m, mc = make_multi(lib, nranks, curve)
    assert lib.ecamd_multi_curve_coord_len(mc) == cl and lib.ecamd_multi_curve_order_len(mc) == ql
    checked = 0
    for n in (1, 2, 5, 7, 8, 64, 1000, (1 << 20) + 3):
        bounds = shard_bounds(n, nranks)
        assert bounds[0][0] == 0 and bounds[-1][1] == n and all(bounds[r][1] == bounds[r + 1][0] for r in range(nranks - 1))
        for fn, args, single, op in cases(cl, ql, curve in EDDSA_CURVES):
            before = lib.mh_count()
            call = [m, mc] + ([C.c_int(op)] if op is not None else []) + [u32(n)] + [a.arg if isinstance(a, Arr) else a for a in args]
            assert getattr(lib, fn)(*call) == 0, (fn, lib.ecamd_last_error())
            recs = records(lib)[before:]
            arrs = [a for a in args if isinstance(a, Arr)]
            live = [r for r in range(nranks) if bounds[r][1] > bounds[r][0]]
            assert sorted(r[1] for r in recs) == live, (fn, n, nranks)  # one call per rank that owns items, none for an empty shard
            for rfn, rank, rn, ptrs, ints in recs:
                lo, hi = bounds[rank]
                assert rfn == single and rn == hi - lo, (fn, rank)
                assert len(ptrs) == len(arrs), fn
                for a, p in zip(arrs, ptrs):
                    want = a.base + lo * a.item if a.item else 0
                    assert p == want, "%s on %s, %d ranks, n = %d: rank %d array with %d-octet items at +%d, expected +%d" % (
                        fn, curve, nranks, n, rank, a.item, p - a.base, want - a.base)
                # integer arguments travel unchanged (lengths, strides, formats, the op code)
                want_ints = ([op] if op is not None else []) + [v.value for v in args if not isinstance(v, Arr)]
                assert ints == want_ints, (fn, ints, want_ints)
                checked += 1
    assert checked > 0
    lib.ecamd_multi_curve_free(mc)
    lib.ecamd_multi_destroy(m)


@pytest.mark.parametrize("curve", EDDSA_CURVES)
@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_verify_all_combines_shard_verdicts(lib, curve, nranks):
    """ecamd_multi_eddsa_verify_all_batch: valid only when every shard is; first_rejected = the lowest global index"""
    cl, _ = CURVES[curve]
    kl = 57 if cl == 56 else cl
    m, mc = make_multi(lib, nranks, curve)
    n = 1003
    bounds = shard_bounds(n, nranks)
    for mask, item in [(0, 0), (1, 5), (1 << (nranks - 1), 0), ((1 << nranks) - 1, 7), (1 << (nranks // 2), 11)]:
        lib.mh_set_bad(mask, item)
        pk, sg, hr = Arr(kl), Arr(2 * kl), Arr(2 * kl)
        ok, first = C.c_int(-7), C.c_uint32(0)
        before = lib.mh_count()
        assert lib.ecamd_multi_eddsa_verify_all_batch(m, mc, u32(n), pk.arg, sg.arg, hr.arg, u32(2 * kl), C.byref(ok), C.byref(first)) == 0
        for rfn, rank, rn, ptrs, ints in records(lib)[before:]:
            lo, hi = bounds[rank]
            assert rn == hi - lo and ptrs == [pk.base + lo * kl, sg.base + lo * 2 * kl, hr.base + lo * 2 * kl] and ints == [2 * kl]
        bad_ranks = [r for r in range(nranks) if (mask >> r) & 1]
        assert ok.value == (0 if bad_ranks else 1)
        assert first.value == (bounds[bad_ranks[0]][0] + item if bad_ranks else n)
    ok = C.c_int(1)
    assert lib.ecamd_multi_eddsa_verify_all_batch(m, mc, u32(0), None, None, None, u32(64), C.byref(ok), None) == -1  # the reference rejects num = 0
    lib.mh_set_bad(0, 0)
    lib.ecamd_multi_curve_free(mc)
    lib.ecamd_multi_destroy(m)


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP224K1", "SECP521R1"])
@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_schnorr_verify_all_shards_and_combines(lib, curve, nranks):
    """ecamd_multi_schnorr_verify_all_batch: s, q - e by the order's length, keys by 2 clen, r by 2 clen (points) or clen (abscissae);
    valid only when every shard is"""
    cl, ql = CURVES[curve]
    m, mc = make_multi(lib, nranks, curve)
    n = 1001
    bounds = shard_bounds(n, nranks)
    for r_fmt in (0, 1):
        for mask in (0, 1, 1 << (nranks - 1)):
            lib.mh_set_bad(mask, 0)
            sv, ne, ky, rr = Arr(ql), Arr(ql), Arr(2 * cl), Arr(cl if r_fmt else 2 * cl)
            ok = C.c_int(-7)
            before = lib.mh_count()
            assert lib.ecamd_multi_schnorr_verify_all_batch(m, mc, u32(n), sv.arg, ne.arg, ky.arg, rr.arg, C.c_int(r_fmt), C.byref(ok)) == 0
            recs = records(lib)[before:]
            assert sorted(r[1] for r in recs) == list(range(nranks))
            for rfn, rank, rn, ptrs, ints in recs:
                lo, hi = bounds[rank]
                assert rfn == "ec_schnorr_verify_all_batch" and rn == hi - lo and ints == [r_fmt]
                assert ptrs == [a.base + lo * a.item for a in (sv, ne, ky, rr)]
            assert ok.value == (0 if mask else 1)
    ok = C.c_int(1)
    assert lib.ecamd_multi_schnorr_verify_all_batch(m, mc, u32(0), None, None, None, None, C.c_int(0), C.byref(ok)) == -1
    lib.mh_set_bad(0, 0)


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP224K1", "SECP521R1"])
@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_schnorr_verify_msg_all_shards_and_combines(lib, curve, nranks):
    """ecamd_multi_schnorr_verify_msg_all_batch (round 6): keys by 2 or 3 clen (affine / projective), signatures by rlen + qlen (rlen = clen for
    abscissae, 2 clen for points), hash-input slots by their stride; valid only when every shard is"""
    cl, ql = CURVES[curve]
    m, mc = make_multi(lib, nranks, curve)
    n = 777
    bounds = shard_bounds(n, nranks)
    for key_fmt in (0, 1):
        for r_fmt in (0, 1):
            for mask in (0, 1 << (nranks - 1)):
                lib.mh_set_bad(mask, 0)
                stride = 132
                ky, sg, sl = Arr((3 if key_fmt else 2) * cl), Arr((cl if r_fmt else 2 * cl) + ql), Arr(stride)
                ok = C.c_int(-7)
                before = lib.mh_count()
                assert lib.ecamd_multi_schnorr_verify_msg_all_batch(m, mc, u32(n), ky.arg, C.c_int(key_fmt), sg.arg, C.c_int(r_fmt), C.c_int(2), sl.arg,
                                                                    u32(stride), u32(2 * 32 + cl), C.byref(ok)) == 0
                recs = [r for r in records(lib)[before:] if r[0] != "ecamd_ctx_discard_msm_seed"]
                assert sorted(r[1] for r in recs) == list(range(nranks))
                for rfn, rank, rn, ptrs, ints in recs:
                    lo, hi = bounds[rank]
                    assert rfn == "ec_schnorr_verify_msg_all_batch" and rn == hi - lo and ints == [key_fmt, r_fmt, 2, stride, 2 * 32 + cl]
                    assert ptrs == [a.base + lo * a.item for a in (ky, sg, sl)]
                assert ok.value == (0 if mask else 1)
    ok = C.c_int(1)
    assert lib.ecamd_multi_schnorr_verify_msg_all_batch(m, mc, u32(0), None, C.c_int(0), None, C.c_int(0), C.c_int(2), None, u32(4), u32(0), C.byref(ok)) == -1
    lib.mh_set_bad(0, 0)


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_eddsa_verify_msg_prj_all_shards_and_combines(lib, nranks):
    """ecamd_multi_eddsa_verify_msg_prj_all_batch (round 6): projective keys by 3 clen, signatures by two encoding lengths, hash-input slots by
    their stride; valid only when every shard is"""
    curve = "WEI25519"
    cl, _ = CURVES[curve]
    m, mc = make_multi(lib, nranks, curve)
    n = 900
    bounds = shard_bounds(n, nranks)
    for mask in (0, 1, 1 << (nranks - 1)):
        lib.mh_set_bad(mask, 0)
        stride = 100
        ky, sg, sl = Arr(3 * cl), Arr(64), Arr(stride)
        ok = C.c_int(-7)
        before = lib.mh_count()
        assert lib.ecamd_multi_eddsa_verify_msg_prj_all_batch(m, mc, u32(n), ky.arg, sg.arg, sl.arg, u32(stride), u32(32), C.byref(ok)) == 0
        recs = [r for r in records(lib)[before:] if r[0] != "ecamd_ctx_discard_msm_seed"]
        assert sorted(r[1] for r in recs) == list(range(nranks))
        for rfn, rank, rn, ptrs, ints in recs:
            lo, hi = bounds[rank]
            assert rfn == "ec_eddsa_verify_msg_prj_all_batch" and rn == hi - lo and ints == [stride, 32]
            assert ptrs == [a.base + lo * a.item for a in (ky, sg, sl)]
        assert ok.value == (0 if mask else 1)
    lib.mh_set_bad(0, 0)


def test_error_of_one_rank_fails_the_call_and_names_the_rank(lib):
    m, mc = make_multi(lib, 3, "SECP256R1")
    lib.mh_set_fail_rank(1)
    a, b, c, d = Arr(32), Arr(64), Arr(64), Arr(1)
    assert lib.ecamd_multi_prj_pt_mul_batch(m, mc, u32(30), a.arg, u32(32), b.arg, c.arg, d.arg) == -1
    assert b"rank 1" in lib.ecamd_last_error() and b"stub failure" in lib.ecamd_last_error()
    lib.mh_set_fail_rank(-1)
    # a curve handle of another multi-context is refused
    m2, mc2 = make_multi(lib, 2, "SECP256R1")
    assert lib.ecamd_multi_prj_pt_mul_batch(m, mc2, u32(30), a.arg, u32(32), b.arg, c.arg, d.arg) == -1
    assert lib.ecamd_multi_prj_pt_mul_batch(m2, mc2, u32(0), a.arg, u32(32), b.arg, c.arg, d.arg) == 0  # n = 0: nothing to do
    assert lib.mh_count() == 0


READY = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_ready_hook_reports_global_item_ranges(lib, nranks):
    """ecamd_multi_set_host_ready_hook: a rank's chunk [first, first + count) of its shard reaches the caller as global indices"""
    m, mc = make_multi(lib, nranks, "WEI448")
    seen = []
    hook = READY(lambda arg, first, count: seen.append((first, count)))
    assert lib.ecamd_multi_set_host_ready_hook(m, hook, None) == 0
    n = 1001
    pk, sg, hs, rs = Arr(168), Arr(114), Arr(200), Arr(1)
    assert lib.ecamd_multi_eddsa_verify_msg_prj_batch(m, mc, u32(n), pk.arg, sg.arg, hs.arg, u32(200), u32(57), rs.arg) == 0
    assert sorted(seen) == [(lo, hi - lo) for lo, hi in shard_bounds(n, nranks)]
    assert lib.ecamd_multi_set_host_ready_hook(m, None, None) == 0
    seen.clear()
    assert lib.ecamd_multi_eddsa_verify_msg_prj_batch(m, mc, u32(n), pk.arg, sg.arg, hs.arg, u32(200), u32(57), rs.arg) == 0
    assert seen == []


def test_settings_reach_every_rank_and_seeds_differ(lib):
    m, mc = make_multi(lib, 8, "WEI25519")
    assert lib.ecamd_multi_set_secret_scalars(m, 1) == 0
    seed = bytes(range(1, 33))
    assert lib.ecamd_multi_set_msm_seed(m, seed) == 0
    assert lib.ecamd_multi_wipe_scratch(m) == 0
    recs = records(lib)
    assert [(r[0], r[1]) for r in recs[:8]] == [("ecamd_ctx_set_secret_scalars", r) for r in range(8)] and all(r[4] == [1] for r in recs[:8])
    seeds = [tuple(r[4]) for r in recs[8:16]]
    assert seeds == [(seed[0] ^ r, seed[1]) for r in range(8)] and len(set(seeds)) == 8
    assert [(r[0], r[1]) for r in recs[16:]] == [("ecamd_ctx_wipe_scratch", r) for r in range(8)]


def test_shard_range_matches_the_python_side(lib):
    """ecamd_multi_shard_range (C) and libecc_amd.shard.shard_range (the bench's ranks) cut the batch at the same places"""
    from libecc_amd.shard import shard_range
    for n in (0, 1, 7, 8, 9, 1000, 1 << 20, (1 << 32) - 1):
        for N in (1, 2, 3, 4, 8):
            for r in range(N):
                lo, hi = C.c_uint32(), C.c_uint32()
                lib.ecamd_multi_shard_range(u32(n), r, N, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == tuple(shard_range(n, r, N)), (n, r, N)
