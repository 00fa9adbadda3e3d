"""GPU tests of what sits between the kernels and the callers: several contexts on one device (process-global
constant slots), calls on different streams of one context (shared scratch), the C-level multi-device layer
(ecamd_multi: one context + host thread per rank, contiguous shards), projective public keys."""
import threading

import numpy as np
import pytest

import libecc_amd
from oracles import CURVES, Oracle
from test_gpu_parity import make_sigs, rand_bytes

pytestmark = pytest.mark.gpu


def test_two_contexts_hold_different_curves_of_one_width(gpu_ctx):
    """The __constant__ curve tables are per device, not per context: two live contexts that load DIFFERENT curves of the
    same word count must not overwrite each other's p / q constants (each handle gets its own slot of the device's table),
    also when they compute at the same time from two host threads."""
    rng = np.random.default_rng(71)
    names_a, names_b = ["SECP256R1", "BRAINPOOLP256R1"], ["SECP256K1", "WEI25519", "FRP256V1"]
    ctx_a, ctx_b = libecc_amd.Context(0), libecc_amd.Context(0)
    try:
        work = []
        for ctx, names in ((ctx_a, names_a), (ctx_b, names_b)):
            for nm in names:
                o, pubs, sigs, dg, hl, _ = make_sigs(nm, "SHA256", 64, rng)
                sigs = bytearray(sigs)
                sigs[3 * 2 * o.qlen + 1] ^= 2
                sc = rand_bytes(rng, 64 * o.qlen)
                work.append(dict(ctx=ctx, name=nm, o=o, pubs=pubs, sigs=bytes(sigs), dg=dg, hl=hl, sc=sc))
        # handles are created interleaved (A, B, A, B, ...) so that every upload happens while the other context is live
        order = sorted(range(len(work)), key=lambda i: (i % 3, i))
        for i in order:
            work[i]["cv"] = work[i]["ctx"].curve(work[i]["name"])
        errs = []

        def run(ctx):
            try:
                for _ in range(3):
                    for w in work:
                        if w["ctx"] is not ctx:
                            continue
                        o = w["o"]
                        assert w["cv"].ecdsa_verify(w["pubs"], w["sigs"], w["dg"], w["hl"]) == o.ecdsa_verify(w["pubs"], w["sigs"], w["dg"], w["hl"]), w["name"]
                        assert w["cv"].scalar_mult(w["sc"], w["pubs"]) == o.scalar_mult(w["sc"], w["pubs"]), w["name"]
                        assert w["cv"].scalar_mult(w["sc"]) == o.scalar_mult(w["sc"]), w["name"]
            except BaseException as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=run, args=(c,)) for c in (ctx_a, ctx_b)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        # freeing one context's handles leaves the other's constants alone
        for w in work:
            if w["ctx"] is ctx_b:
                w["cv"].free()
        extra = ctx_b.curve("GOST256")
        for w in work:
            if w["ctx"] is ctx_a:
                assert w["cv"].scalar_mult(w["sc"], w["pubs"]) == w["o"].scalar_mult(w["sc"], w["pubs"]), w["name"]
        extra.free()
        # the same curve in two contexts shares its slot; 40 handles of one curve do not exhaust the table
        many = [ctx_b.curve("SECP256R1") for _ in range(40)]
        w = work[0]
        assert many[-1].scalar_mult(w["sc"]) == w["o"].scalar_mult(w["sc"])
        for cv in many:
            cv.free()
        for w in work:
            if w["ctx"] is ctx_a:
                w["cv"].free()
    finally:
        ctx_a.close()
        ctx_b.close()


def test_calls_on_two_streams_of_one_context(gpu_ctx):
    """A context has one set of scratch buffers: back-to-back *_dev calls on DIFFERENT streams (no host synchronisation in
    between) must still each see their own tables -- the second call waits on the device for the first."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(72)
    n = 1 << 14
    o = Oracle("SECP256R1")
    cv = gpu_ctx.curve("SECP256R1")
    try:
        base, st = cv.scalar_mult(rand_bytes(rng, 32 * n))
        assert set(st) == {0}
        scs = [rand_bytes(rng, 32 * n) for _ in range(4)]
        exp = [cv.scalar_mult(s, base) for s in scs]
        streams = [torch.cuda.Stream(device=dev) for _ in range(4)]

        def t(b):
            return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

        dbase = t(base)
        dsc = [t(s) for s in scs]
        dout = [torch.zeros(64 * n, dtype=torch.uint8, device=dev) for _ in range(4)]
        dst = [torch.full((n,), 0xAA, dtype=torch.uint8, device=dev) for _ in range(4)]
        torch.cuda.synchronize()
        for rep in range(2):
            for k in range(4):
                cv.scalar_mult_dev(n, dsc[k].data_ptr(), 32, dbase.data_ptr(), dout[k].data_ptr(), dst[k].data_ptr(),
                                   streams[k].cuda_stream)
        torch.cuda.synchronize()
        for k in range(4):
            assert (bytes(dout[k].cpu().numpy()), bytes(dst[k].cpu().numpy())) == exp[k], k
        # a sample against the oracle too
        assert exp[0][0][:64 * 64] == o.scalar_mult(scs[0][:32 * 64], base[:64 * 64])[0]
    finally:
        cv.free()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_device_layer_shards_like_one_device(gpu_ctx, devices):
    """ecamd_multi over a device list (the 1-GPU box lists device 0 several times: one context, one host thread and one
    contiguous shard per entry): outputs and status bytes equal the single-context call's, for uneven shard sizes, fewer
    items than ranks and empty batches"""
    rng = np.random.default_rng(73)
    m = libecc_amd.Multi(devices)
    try:
        assert m.size == len(devices)
        # shard ranges tile [0, n)
        for n in (0, 1, 2, 5, 1000, 1001, (1 << 20) + 3):
            prev = 0
            for r in range(m.size):
                lo, hi = m.shard_range(n, r)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == n
        for curve in ("SECP256R1", "SECP384R1"):
            o = Oracle(curve)
            mc = m.curve(curve)
            cv = gpu_ctx.curve(curve)
            try:
                for n in (0, 1, 2, 3001):
                    sc = rand_bytes(rng, o.qlen * n)
                    pts, st = cv.scalar_mult(sc)
                    assert mc.scalar_mult(sc) == (pts, st)
                    sc2 = rand_bytes(rng, o.qlen * n)
                    if n:
                        bad = bytearray(pts)
                        bad[-1] ^= 1                     # the last item (last rank's shard) is off the curve
                        bad = bytes(bad)
                        got = mc.scalar_mult(sc2, bad)
                        assert got == cv.scalar_mult(sc2, bad) and got[1][-1] == 1
                        k = min(n, 64)
                        assert got[0][:k * 2 * o.clen] == o.scalar_mult(sc2[:k * o.qlen], bad[:k * 2 * o.clen])[0]
                n = 777
                oo, pubs, sigs, dg, hl, _ = make_sigs(curve, "SHA256", n, rng)
                sigs = bytearray(sigs)
                for i in range(0, n, 13):
                    sigs[i * 2 * oo.qlen + 4] ^= 0x20
                sigs = bytes(sigs)
                exp = cv.ecdsa_verify(pubs, sigs, dg, hl)
                assert mc.ecdsa_verify(pubs, sigs, dg, hl) == exp == oo.ecdsa_verify(pubs, sigs, dg, hl)
                privs = rand_bytes(rng, n * oo.qlen)
                assert mc.ecccdh(privs, pubs) == cv.ecccdh(privs, pubs)
            finally:
                mc.free()
                cv.free()
        # X25519 and Ed25519 verification shard the same way
        import oracles as O
        mc, cv = m.curve("WEI25519"), gpu_ctx.curve("WEI25519")
        try:
            n = 515
            k, u = rand_bytes(rng, 32 * n), bytearray(rand_bytes(rng, 32 * n))
            for i in range(n):
                u[32 * i + 31] &= 0x7f
            assert mc.xdh(k, bytes(u)) == cv.xdh(k, bytes(u))
            pubs, sigs, msgs = b"", b"", []
            for i in range(96):
                a_enc, sig, _ = O.ed25519_sign(rand_bytes(rng, 32), b"m%d" % i)
                if i % 5 == 0:
                    sig = sig[:40] + bytes([sig[40] ^ 1]) + sig[41:]
                pubs += a_enc
                sigs += sig
                msgs.append(b"m%d" % i)
            import hashlib
            hram = b"".join(hashlib.sha512(sigs[64 * i:64 * i + 32] + pubs[32 * i:32 * i + 32] + msgs[i]).digest() for i in range(96))
            exp = cv.eddsa_verify(pubs, sigs, hram)
            assert mc.eddsa_verify(pubs, sigs, hram) == exp and exp[0] == 1 and exp[1] == 0
            # the whole-batch bit, sharded: first rejected index of the whole batch; a valid subset is accepted
            assert mc.eddsa_verify_all(pubs, sigs, hram) == (False, 0)
            good = [i for i in range(96) if exp[i] == 0]
            sub = lambda b, w, idx: b"".join(b[w * i:w * (i + 1)] for i in idx)
            assert mc.eddsa_verify_all(sub(pubs, 32, good), sub(sigs, 64, good), sub(hram, 64, good)) == (True, len(good))
            late = good[:60] + [5] + good[60:]
            assert mc.eddsa_verify_all(sub(pubs, 32, late), sub(sigs, 64, late), sub(hram, 64, late)) == (False, 60)
        finally:
            mc.free()
            cv.free()
    finally:
        m.close()


def test_multi_allgather_single_rank(gpu_ctx):
    """the RCCL all-gather degenerates to a device copy with one rank; with a device listed twice it refuses (RCCL needs
    distinct devices) instead of hanging"""
    import ctypes as C
    import torch
    L = libecc_amd.load_library()
    dev = torch.device("cuda:0")
    m = libecc_amd.Multi([0])
    try:
        src = torch.arange(4096, dtype=torch.uint8, device=dev)
        dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        send = (C.c_void_p * 1)(src.data_ptr())
        recv = (C.c_void_p * 1)(dst.data_ptr())
        assert L.ecamd_multi_allgather(m.h, send, recv, 4096) == 0, L.ecamd_last_error()
        assert torch.equal(src, dst)
    finally:
        m.close()
    m = libecc_amd.Multi([0, 0])
    try:
        send = (C.c_void_p * 2)(src.data_ptr(), src.data_ptr())
        recv = (C.c_void_p * 2)(dst.data_ptr(), dst.data_ptr())
        assert L.ecamd_multi_allgather(m.h, send, recv, 16) == -1
        assert b"distinct devices" in L.ecamd_last_error()
    finally:
        m.close()


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP384R1", "WEI25519"])
def test_ecdsa_verify_with_projective_keys(gpu_ctx, curve):
    """ec_ecdsa_verify_batch_fmt: keys as X || Y || Z (what an ec_pub_key holds).  Scaled representatives verify like their
    affine form; a broken triple is rejected; the point at infinity (0 : 1 : 0) is a KEY for libecc and verification against
    it is W' = [u1]G (sig/ecdsa_common.c:788-800) -- checked here against integers, and against libecc itself by
    libecc_amd/compat/compat_check.c"""
    rng = np.random.default_rng(74)
    n = 48
    o, pubs, sigs, dg, hl, _ = make_sigs(curve, "SHA256", n, rng)
    c = CURVES[curve]
    p, q, cl, ql = c["p"], c["q"], o.clen, o.qlen
    sigs = bytearray(sigs)
    sigs[7 * 2 * ql + 3] ^= 1
    sigs = bytes(sigs)
    exp = bytearray(o.ecdsa_verify(pubs, sigs, dg, hl))
    prj = bytearray()
    for i in range(n):
        x = int.from_bytes(pubs[i * 2 * cl:i * 2 * cl + cl], "big")
        y = int.from_bytes(pubs[i * 2 * cl + cl:(i + 1) * 2 * cl], "big")
        lam = 1 if i % 3 == 0 else (int.from_bytes(rand_bytes(rng, cl + 8), "big") % (p - 1)) + 1
        prj += (x * lam % p).to_bytes(cl, "big") + (y * lam % p).to_bytes(cl, "big") + lam.to_bytes(cl, "big")
    # item 5: Y + 1 (not on the curve); item 6: Z >= p; item 9: (0 : 0 : 0)
    off = 5 * 3 * cl
    prj[off + 2 * cl - 1] ^= 1
    exp[5] = 1
    prj[6 * 3 * cl + 2 * cl:7 * 3 * cl] = p.to_bytes(cl, "big")
    exp[6] = 1
    prj[9 * 3 * cl:10 * 3 * cl] = bytes(3 * cl)
    exp[9] = 1
    # items 10..13: the key is the point at infinity; 10/11 carry signatures built so that r = x([e/s]G) mod q
    qbits = q.bit_length()
    for j, i in enumerate(range(10, 14)):
        prj[i * 3 * cl:(i + 1) * 3 * cl] = bytes(cl) + (1 + j).to_bytes(cl, "big") + bytes(cl)     # (0 : y : 0), any y != 0
        e = int.from_bytes(dg[i * hl:(i + 1) * hl], "big")
        if 8 * hl > qbits:
            e >>= 8 * hl - qbits
        e %= q
        s = (int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1
        u1 = e * pow(s, q - 2, q) % q
        W, st = o.scalar_mult(u1.to_bytes(ql, "big"))
        assert st == b"\x00"
        r = int.from_bytes(W[:cl], "big") % q
        if i >= 12:
            r = (r % (q - 1)) + 1 if r + 1 >= q else r + 1      # some other r: rejected
        sigs = sigs[:i * 2 * ql] + r.to_bytes(ql, "big") + s.to_bytes(ql, "big") + sigs[(i + 1) * 2 * ql:]
        exp[i] = 0 if i < 12 else 1
    cv = gpu_ctx.curve(curve)
    try:
        got = cv.ecdsa_verify_fmt(bytes(prj), 1, sigs, dg, hl)
        assert got == bytes(exp), (list(got), list(exp))
        assert cv.ecdsa_verify_fmt(pubs, 0, sigs, dg, hl) == cv.ecdsa_verify(pubs, sigs, dg, hl)
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP384R1", "BRAINPOOLP256R1"])
def test_ecdsa_sign_rejects_private_key_not_below_q(gpu_ctx, curve):
    """__ecdsa_sign_init fails on x >= q (sig/ecdsa_common.c:367-371): status 1, not a signature under x mod q"""
    from oracles import RefLib, have_ref
    import hashlib
    rng = np.random.default_rng(75)
    o = Oracle(curve)
    q, ql = CURVES[curve]["q"], o.qlen
    top = (1 << (8 * ql)) - 1
    xs = [1, 2, q - 1, q, q + 1, min(top, 2 * q - 1), top] + [int.from_bytes(rand_bytes(rng, ql), "big") % (q - 1) + 1 for _ in range(9)]
    privs = b"".join(x.to_bytes(ql, "big") for x in xs)
    n = len(xs)
    ks = b"".join((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1) + 1).to_bytes(ql, "big") for _ in range(n))
    msgs = rand_bytes(rng, 20 * n)
    dg = b"".join(hashlib.sha256(msgs[20 * i:20 * i + 20]).digest() for i in range(n))
    cv = gpu_ctx.curve(curve)
    try:
        got = cv.ecdsa_sign(privs, ks, dg, 32)
        assert got == o.ecdsa_sign(privs, ks, dg, 32)
        assert list(got[1][:7]) == [0, 0, 0, 1, 1, 1 if 2 * q - 1 <= top else 1, 1]
        if have_ref():
            rs, _, rst = RefLib(curve).ecdsa_sign("SHA256", privs, ks, msgs, 20)
            assert rst == got[1] and rs == got[0]
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP384R1", "WEI25519", "SECP521R1", "SECP256K1", "WEI448", "BRAINPOOLP512R1"])
def test_blinded_scalar_mult(gpu_ctx, curve):
    """ec_prj_pt_mul_blind_batch multiplies by m + b #E (prj_pt_mul_blind, curves/prj_pt.c:1782-1822): same points as the plain
    multiplication for every b in [1, #E); b = 0 and b >= #E are refused; on secp256r1 the ~2|q|-bit scalar stays on the
    radix-2^29 window kernel (k_p256_loop<17>), whose long-scalar recoding is also exercised directly with edge scalars"""
    rng = np.random.default_rng(76)
    o = Oracle(curve)
    c = CURVES[curve]
    q, order, ql, cl = c["q"], c["order"], o.qlen, o.clen
    ol = (order.bit_length() + 7) // 8
    n = 200
    sc = rand_bytes(rng, ql * n)
    cv = gpu_ctx.curve(curve)
    try:
        base, st = cv.scalar_mult(rand_bytes(rng, ql * n))
        assert set(st) <= {0, 2}
        base = b"".join(base[2 * cl * i:2 * cl * (i + 1)] if st[i] == 0 else base[:2 * cl] for i in range(n))
        bl = [int.from_bytes(rand_bytes(rng, ol + 8), "big") % (order - 1) + 1 for _ in range(n)]
        bl[:6] = [1, 2, order - 1, order - 2, 1 << (8 * ol - 9), 3]
        blinds = b"".join(b.to_bytes(ol, "big") for b in bl)
        exp_v, exp_f = cv.scalar_mult(sc, base), cv.scalar_mult(sc)
        assert cv.scalar_mult_blind(sc, blinds, base, ql, ol) == exp_v == o.scalar_mult(sc, base)
        assert cv.scalar_mult_blind(sc, blinds, None, ql, ol) == exp_f
        bad = bytearray(blinds)
        bad[0:ol] = bytes(ol)                                   # b = 0
        bad[ol:2 * ol] = order.to_bytes(ol, "big")              # b = #E
        bad[2 * ol:3 * ol] = b"\xff" * ol                       # b > #E
        got = cv.scalar_mult_blind(sc, bytes(bad), base, ql, ol)
        assert list(got[1][:3]) == [1, 1, 1] and got[0][:3 * 2 * cl] == bytes(3 * 2 * cl)
        assert got[0][3 * 2 * cl:] == exp_v[0][3 * 2 * cl:] and got[1][3:] == exp_v[1][3:]
        # long scalars given directly (what the blinding produces), incl. edge values, against the oracle
        for slen in (ql + 1, ql + ol + 1, 2 * ql + 3):
            if slen > 68 and curve == "SECP256R1":
                continue
            vals = [0, 1, q, q + 1, (1 << (8 * slen)) - 1, 5 * q, order * 7 + 3, 1 << (8 * slen - 1)]
            vals += [int.from_bytes(rand_bytes(rng, slen), "big") for _ in range(24)]
            ss = b"".join((v & ((1 << (8 * slen)) - 1)).to_bytes(slen, "big") for v in vals)
            k = len(vals)
            assert cv.scalar_mult(ss, base[:2 * cl * k], slen) == o.scalar_mult(ss, base[:2 * cl * k], slen), slen
            assert cv.scalar_mult(ss, None, slen) == o.scalar_mult(ss, None, slen), slen
    finally:
        cv.free()


def test_secret_scalar_mode_gives_identical_results():
    """ecamd_ctx_set_secret_scalars: constant-address (masked full-scan) table look-ups on the complete-formula kernel for
    every scalar multiplication of the context -- scalar mult (fixed and variable base, edge scalars), ECDSA signing, ECC-CDH,
    Ed25519 signing step R, key-pair import -- same bytes as the default mode"""
    rng = np.random.default_rng(77)
    ctx_s, ctx_d = libecc_amd.Context(0), libecc_amd.Context(0)
    ctx_s.set_secret_scalars(True)
    try:
        for curve in ("SECP256R1", "BRAINPOOLP256R1", "SECP384R1", "WEI25519", "SECP256K1", "SECP521R1", "WEI448", "SECP224R1"):
            o = Oracle(curve)
            q, ql, cl = CURVES[curve]["q"], o.qlen, o.clen
            n = 96
            cs, cd = ctx_s.curve(curve), ctx_d.curve(curve)
            try:
                vals = [0, 1, 2, q - 1, q, q + 1, (1 << (8 * ql)) - 1] + [int.from_bytes(rand_bytes(rng, ql), "big") for _ in range(n - 7)]
                sc = b"".join(v.to_bytes(ql, "big") for v in vals)
                assert cs.scalar_mult(sc) == cd.scalar_mult(sc) == o.scalar_mult(sc)
                base, st = cd.scalar_mult(rand_bytes(rng, ql * n))
                assert cs.scalar_mult(sc, base) == cd.scalar_mult(sc, base)
                privs = b"".join((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1) + 1).to_bytes(ql, "big") for _ in range(n))
                ks = b"".join((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1) + 1).to_bytes(ql, "big") for _ in range(n))
                dg = rand_bytes(rng, 32 * n)
                assert cs.ecdsa_sign(privs, ks, dg, 32) == cd.ecdsa_sign(privs, ks, dg, 32) == o.ecdsa_sign(privs, ks, dg, 32)
                assert cs.ecccdh(privs, base) == cd.ecccdh(privs, base)
                if curve == "WEI25519":
                    rh = rand_bytes(rng, 64 * n)
                    assert cs.eddsa_sign_R(rh) == cd.eddsa_sign_R(rh)
                    k, u = rand_bytes(rng, 32 * n), (9).to_bytes(32, "little") * n
                    assert cs.xdh(k, u) == cd.xdh(k, u)
            finally:
                cs.free()
                cd.free()
    finally:
        ctx_s.close()
        ctx_d.close()


def _both(L, name, ctx, cv, m, mc, args, op=None):
    """call ec_<name> on one context and ecamd_multi_<name> on the multi-context with the same inputs; args: bytes = input array,
    ("out", size) = output array, anything else = a ctypes scalar.  Returns the two lists of output bytes."""
    import ctypes as C
    outs = []
    for h1, h2, fn in ((ctx.h, cv.h, getattr(L, "ec_" + name)), (m.h, mc.h, getattr(L, "ecamd_multi_" + name))):
        call, bufs = [h1, h2] + ([C.c_int(op)] if op is not None else []), []
        for a in args:
            if isinstance(a, tuple):
                b = C.create_string_buffer(bytes([0xA5]) * max(1, a[1]), max(1, a[1]))
                bufs.append((b, a[1]))
                call.append(b)
            else:
                call.append(a)
        rc = fn(*call)
        assert rc == 0, (name, L.ecamd_last_error())
        outs.append([b.raw[:k] for b, k in bufs])
    return outs


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP521R1", "WEI25519", "WEI448"])
def test_every_multi_entry_point_with_eight_ranks(gpu_ctx, curve):
    """The driver's 8-GPU shape without the hardware: EVERY batch entry point of ecamd_multi.cpp with eight ranks on device 0 (uneven
    shards; and fewer items than ranks) gives the bytes of the one-context call -- one curve per length class (32 / 66 octets, and
    the 57 / 114-octet EdDSA encodings of Ed448 whose signature offsets round 4 got wrong).  Inputs are made valid with the library
    itself (key pairs, signatures), so that accepted items exist and a shard reading at a wrong offset shows as a rejection."""
    import ctypes as C
    import hashlib
    rng = np.random.default_rng(86)
    L = libecc_amd.load_library()
    m = libecc_amd.Multi([0] * 8)
    cv, mc = gpu_ctx.curve(curve), m.curve(curve)
    cl, ql = cv.clen, cv.qlen
    q = CURVES[curve]["q"]
    u32, ci = C.c_uint32, C.c_int
    try:
        for n in (203, 5):
            def same(name, args, op=None, expect_ok=None):
                a, b = _both(L, name, gpu_ctx, cv, m, mc, [u32(n)] + args if op is None else [u32(n)] + args, op)
                assert a == b, (name, n)
                if expect_ok is not None:       # the status / result array: accepted items exist (a wrong offset would reject them)
                    assert a[expect_ok].count(0) >= (3 * n) // 4, (name, n, a[expect_ok])
                return a
            sc = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1).to_bytes(ql, "big") for _ in range(n))
            pts = same("prj_pt_mul_batch", [sc, u32(ql), None, ("out", 2 * cl * n), ("out", n)], expect_ok=1)[0]
            sc2 = rand_bytes(rng, (ql + 5) * n)
            same("prj_pt_mul_batch", [sc2, u32(ql + 5), pts, ("out", 2 * cl * n), ("out", n)], expect_ok=1)
            prj = same("prj_pt_mul_batch_fmt", [sc2, u32(ql), pts, ci(0), ("out", 3 * cl * n), ci(1), ("out", n)], expect_ok=1)[0]
            same("prj_pt_mul_batch_fmt", [sc, u32(ql), prj, ci(1), ("out", 2 * cl * n), ci(0), ("out", n)], expect_ok=1)
            same("prj_pt_unique_batch", [prj, ci(1), ("out", 2 * cl * n), ci(0), ("out", n)], expect_ok=1)
            same("prj_pt_unique_batch", [pts, ci(0), ("out", 3 * cl * n), ci(1), ("out", n)], expect_ok=1)
            same("prj_pt_add_batch", [pts, pts[2 * cl:] + pts[:2 * cl], ("out", 2 * cl * n), ("out", n)], expect_ok=1)
            same("prj_pt_op_batch_fmt", [prj, prj[3 * cl:] + prj[:3 * cl], ci(1), ("out", 3 * cl * n), ci(1), ("out", n)], op=0, expect_ok=1)
            same("prj_pt_op_batch_fmt", [pts, None, ci(0), ("out", 2 * cl * n), ci(0), ("out", n)], op=1, expect_ok=1)
            same("prj_pt_unprotected_mult_batch", [sc, u32(ql), u32(ql), prj, ci(1), ("out", 2 * cl * n), ci(0), ("out", n)], expect_ok=1)
            raw = rand_bytes(rng, 2 * ql * n)
            privs, pubs, _ = same("key_pair_gen_raw_batch", [raw, ("out", ql * n), ("out", 2 * cl * n), ("out", n)], expect_ok=2)
            same("ecccdh_derive_batch", [privs, pts, ("out", cl * n), ("out", n)], expect_ok=1)
            if curve.startswith("SECP"):
                dl = 32
                dg = rand_bytes(rng, dl * n)
                nonces = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1).to_bytes(ql, "big") for _ in range(n))
                sigs = same("ecdsa_sign_batch", [privs, nonces, dg, u32(dl), ("out", 2 * ql * n), ("out", n)], expect_ok=1)[0]
                same("ecdsa_verify_batch", [pubs, sigs, dg, u32(dl), ("out", n)], expect_ok=0)
                pubs_prj = cv.unique(pubs, 0, 1)[0]
                same("ecdsa_verify_batch_fmt", [pubs_prj, ci(1), sigs, dg, u32(dl), ("out", n)], expect_ok=0)
                msgs = [rand_bytes(rng, 1 + (7 * i) % 60) for i in range(n)]
                slots, stride = cv.msg_slots(msgs)
                sg2 = same("ecdsa_sign_msg_batch", [privs, rand_bytes(rng, 2 * ql * n), ci(2), slots, u32(stride), ("out", 2 * ql * n), ("out", n)], expect_ok=1)[0]
                same("ecdsa_verify_msg_batch_fmt", [pubs_prj, ci(1), sg2, ci(2), slots, u32(stride), ("out", n)], expect_ok=0)
                bad = bytearray(sg2)
                bad[-1] ^= 1                                      # the last item of the last rank's shard
                got = same("ecdsa_verify_msg_batch_fmt", [pubs, ci(0), bytes(bad), ci(2), slots, u32(stride), ("out", n)])[0]
                assert got == bytes(n - 1) + b"\x01"
                continue
            # ---- EdDSA on the WEI25519 / WEI448 handles: valid signatures made with the signing entry points themselves ----
            ed448 = cl == 56
            kl, hl = (57, 114) if ed448 else (32, 64)
            H = (lambda x: hashlib.shake_256(b"SigEd448\x00\x00" + x).digest(114)) if ed448 else (lambda x: hashlib.sha512(x).digest())
            kle = lambda x: x.to_bytes(kl, "little")
            a = [(int.from_bytes(rand_bytes(rng, 64), "big") % (q - 1)) + 1 for _ in range(n)]
            inv4 = pow(4, -1, q) if ed448 else 1                  # eddsa_derive_priv_key: the Ed448 key lives on the 4-isogenous curve
            Aw = cv.scalar_mult(b"".join(((x * inv4) % q).to_bytes(ql, "big") for x in a))[0]
            keys_prj = cv.unique(Aw, 0, 1)[0]
            Aenc = same("eddsa_encode_point_batch", [keys_prj, ("out", kl * n), ("out", n)], expect_ok=1)[0]
            r_hash = rand_bytes(rng, hl * n)
            Renc = same("eddsa_sign_R_batch", [r_hash, ("out", kl * n), ("out", n)], expect_ok=1)[0]
            msgs = [rand_bytes(rng, 1 + (5 * i) % 70) for i in range(n)]
            inputs = [Renc[kl * i:kl * (i + 1)] + Aenc[kl * i:kl * (i + 1)] + msgs[i] for i in range(n)]
            hram = b"".join(H(x) for x in inputs)
            S = same("eddsa_sign_S_batch", [r_hash, hram, b"".join(kle(x) for x in a), ("out", kl * n)])[0]
            sigs = b"".join(Renc[kl * i:kl * (i + 1)] + S[kl * i:kl * (i + 1)] for i in range(n))
            same("eddsa_verify_batch", [Aenc, sigs, hram, u32(hl), ("out", n)], expect_ok=0)
            dom = b"SigEd448\x00\x00" if ed448 else b""
            slots, stride = cv.msg_slots([dom + x for x in inputs])
            if not ed448:   # (the encoded-key form hashes with SHA-512 only; Ed448 has the two projective-key forms below)
                same("eddsa_verify_msg_batch", [Aenc, sigs, slots, u32(stride), ("out", n)], expect_ok=0)
            blank = [dom + x[:kl] + bytes(kl) + x[2 * kl:] for x in inputs]
            slots_b, stride_b = cv.msg_slots(blank)
            same("eddsa_verify_msg_prj_batch", [keys_prj, sigs, slots_b, u32(stride_b), u32(len(dom) + kl), ("out", n)], expect_ok=0)
            bad = bytearray(sigs)
            bad[-3] ^= 1                                          # S of the last item of the last shard
            got = same("eddsa_verify_msg_prj_batch", [keys_prj, bytes(bad), slots_b, u32(stride_b), u32(len(dom) + kl), ("out", n)])[0]
            assert got == bytes(n - 1) + b"\x01"
            # the pre-hashed variant: dom(1, "") || R || blank A || blank PH(M), the messages in slots of their own
            domph = b"SigEd448\x01\x00" if ed448 else b"SigEd25519 no Ed25519 collisions\x01\x00"
            PH = (lambda x: hashlib.shake_256(x).digest(64)) if ed448 else (lambda x: hashlib.sha512(x).digest())
            Hph = (lambda x: hashlib.shake_256(x).digest(114)) if ed448 else (lambda x: hashlib.sha512(x).digest())
            hram_ph = b"".join(Hph(domph + Renc[kl * i:kl * (i + 1)] + Aenc[kl * i:kl * (i + 1)] + PH(msgs[i])) for i in range(n))
            S_ph = cv.eddsa_sign_S(r_hash, hram_ph, b"".join(kle(x) for x in a))
            sigs_ph = b"".join(Renc[kl * i:kl * (i + 1)] + S_ph[kl * i:kl * (i + 1)] for i in range(n))
            slots_p, stride_p = cv.msg_slots([domph + Renc[kl * i:kl * (i + 1)] + bytes(kl + 64) for i in range(n)])
            mslots, mstride = cv.msg_slots(msgs)
            same("eddsa_verify_ph_prj_batch", [keys_prj, sigs_ph, slots_p, u32(stride_p), u32(len(domph) + kl), mslots, u32(mstride), ("out", n)], expect_ok=0)
            k = rand_bytes(rng, cl * n)
            u = bytearray(rand_bytes(rng, cl * n))
            if not ed448:
                for i in range(n):
                    u[32 * i + 31] &= 0x7f
            xs = same("xdh_batch", [k, bytes(u), ("out", cl * n), ("out", n)])[1]
            assert xs.count(0) >= n // 4         # (a random u is on the curve -- not its twist -- half of the time: libecc rejects the others)
            # the whole-batch bit: valid, then one bad item in the last shard
            assert mc.eddsa_verify_all(Aenc, sigs, hram) == cv.eddsa_verify_all(Aenc, sigs, hram) == (True, n)
            assert mc.eddsa_verify_all(Aenc, bytes(bad), hram) == cv.eddsa_verify_all(Aenc, bytes(bad), hram) == (False, n - 1)
    finally:
        mc.free()
        cv.free()
        m.close()
