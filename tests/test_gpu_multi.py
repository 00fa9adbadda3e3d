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
