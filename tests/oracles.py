"""ctypes front-ends for the two CPU checkers (TEST INFRASTRUCTURE):

  Oracle  -- oracle/liboracle.so, our plain-C restatement (oracle/ecc_oracle.c)
  RefLib  -- oracle/_ref/libecc_ref.so, the unmodified reference + oracle/ref_driver.c
             (present when it was built in the authoring container; it travels to the GPU box)

Nothing under libecc_amd/ imports this module.
"""
import ctypes as C
import hashlib
import hmac
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
u8p = C.c_char_p


def curves():
    with open(os.path.join(GOLDEN, "curves.json")) as f:
        out = {}
        for c in json.load(f):
            out[c["name"]] = {k: (int(v, 16) if k not in ("name", "type") else v) for k, v in c.items()}
        return out


CURVES = curves()


def curve_type(curve):
    """libecc's ec_curve_type of a built-in curve (the byte structured keys / signatures carry)"""
    return int(CURVES[curve]["type"])


def clen(curve):
    return (CURVES[curve]["p"].bit_length() + 7) // 8


def qlen(curve):
    return (CURVES[curve]["q"].bit_length() + 7) // 8


def _be(x, n=None):
    n = n or max(1, (x.bit_length() + 7) // 8)
    return x.to_bytes(n, "big")


def build_oracle():
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "ecc_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


class Oracle:
    """Restatement oracle bound to one curve."""

    def __init__(self, curve):
        self.L = C.CDLL(build_oracle())
        self.curve = curve
        c = CURVES[curve]
        self.clen, self.qlen = clen(curve), qlen(curve)
        self.nl = (c["p"].bit_length() + 63) // 64
        self.ctx = C.create_string_buffer(self.L.orc_sizeof_curve())
        args = []
        for k in ("p", "a", "b", "order", "gx", "gy", "q"):
            b = _be(c[k])
            args += [b, len(b)]
        assert self.L.orc_curve_init(self.ctx, *args) == 0

    def scalar_mult(self, scalars, points=None, slen=None):
        slen = slen or self.qlen
        n = len(scalars) // slen
        out = C.create_string_buffer(2 * self.clen * n)
        st = C.create_string_buffer(n)
        assert self.L.orc_scalar_mult_batch(self.ctx, n, scalars, slen, points, out, st) == 0
        return out.raw, st.raw

    def pt_add(self, p1, p2=None):
        n = len(p1) // (2 * self.clen)
        out = C.create_string_buffer(2 * self.clen * n)
        st = C.create_string_buffer(n)
        assert self.L.orc_pt_add_batch(self.ctx, n, p1, p2, out, st, 1 if p2 is None else 0) == 0
        return out.raw, st.raw

    def pt_op_fmt(self, op, p1, p2, in_fmt, out_fmt):
        """op 0 add / 1 dbl / 2 on-curve test / 3 neg / 4 cmp / 5 eq_or_opp (4, 5: one predicate byte per item); formats 0 affine X || Y, 1 projective X || Y || Z; returns (out, status)"""
        iw, ow = (3 if in_fmt else 2) * self.clen, (1 if op >= 4 else (3 if out_fmt else 2) * self.clen)
        n = len(p1) // iw
        out, st = C.create_string_buffer(max(1, ow * n)), C.create_string_buffer(max(1, n))
        assert self.L.orc_pt_op_batch_fmt(self.ctx, op, n, p1, p2, in_fmt, out, out_fmt, st) == 0
        return (b"" if op == 2 else out.raw[:ow * n]), st.raw[:n]

    def unprotected_mult(self, scalars, slen, points, in_fmt, out_fmt, broadcast=False):
        """_prj_pt_unprotected_mult per item; broadcast: one scalar (slen bytes) for every point"""
        iw, ow = (3 if in_fmt else 2) * self.clen, (3 if out_fmt else 2) * self.clen
        n = len(points) // iw
        out, st = C.create_string_buffer(max(1, ow * n)), C.create_string_buffer(max(1, n))
        assert self.L.orc_unprotected_mult_batch(self.ctx, n, scalars, slen, 0 if broadcast else slen, points, in_fmt, out, out_fmt, st) == 0
        return out.raw[:ow * n], st.raw[:n]

    def random_mod(self, raw):
        """nn_get_random_mod given its 2 * qlen random bytes per item"""
        n = len(raw) // (2 * self.qlen)
        out = C.create_string_buffer(max(1, self.qlen * n))
        assert self.L.orc_random_mod_batch(self.ctx, n, raw, out) == 0
        return out.raw[:self.qlen * n]

    def fp_op(self, op, a, b):
        """a, b: lists of ints < p; returns list of ints."""
        n = len(a)
        A = (C.c_uint64 * (n * self.nl))(*[(x >> (64 * k)) & (2**64 - 1) for x in a for k in range(self.nl)])
        B = (C.c_uint64 * (n * self.nl))(*[(x >> (64 * k)) & (2**64 - 1) for x in b for k in range(self.nl)])
        O = (C.c_uint64 * (n * self.nl))()
        assert self.L.orc_fp_op_batch(self.ctx, op, n, A, B, O) == 0
        return [sum(O[i * self.nl + k] << (64 * k) for k in range(self.nl)) for i in range(n)]

    def ecdsa_verify(self, pubs, sigs, digests, hsize):
        n = len(pubs) // (2 * self.clen)
        res = C.create_string_buffer(n)
        assert self.L.orc_ecdsa_verify_batch(self.ctx, n, pubs, sigs, digests, hsize, res) == 0
        return res.raw

    def ecdsa_sign(self, privs, nonces, digests, hsize):
        n = len(privs) // self.qlen
        sigs = C.create_string_buffer(2 * self.qlen * n)
        st = C.create_string_buffer(n)
        assert self.L.orc_ecdsa_sign_batch(self.ctx, n, privs, nonces, digests, hsize, sigs, st) == 0
        return sigs.raw, st.raw

    def ecccdh(self, privs, peers):
        n = len(privs) // self.qlen
        sec = C.create_string_buffer(self.clen * n)
        st = C.create_string_buffer(n)
        assert self.L.orc_ecccdh_batch(self.ctx, n, privs, peers, sec, st) == 0
        return sec.raw, st.raw


    def eddsa_sign_R(self, r_hash):
        n = len(r_hash) // 64
        out, st = C.create_string_buffer(max(1, 32 * n)), C.create_string_buffer(max(1, n))
        assert self.L.orc_eddsa25519_sign_R_batch(self.ctx, n, r_hash, out, st) == 0
        return out.raw[:32 * n], st.raw[:n]

    def eddsa_sign_S(self, r_hash, hram, a_scalars):
        n = len(r_hash) // 64
        out = C.create_string_buffer(max(1, 32 * n))
        assert self.L.orc_eddsa25519_sign_S_batch(self.ctx, n, r_hash, hram, a_scalars, out) == 0
        return out.raw[:32 * n]

    def xdh(self, k, u):
        n = len(k) // self.clen
        out = C.create_string_buffer(self.clen * n)
        st = C.create_string_buffer(n)
        assert self.L.orc_xdh_batch(self.ctx, self.clen, n, k, u, out, st) == 0
        return out.raw, st.raw

    def prj(self, points, scalars=None, slen=None):
        """projective X || Y || Z in and out (scalars None: normalisation only)"""
        n = len(points) // (3 * self.clen)
        slen = slen or self.qlen
        out = C.create_string_buffer(max(1, 3 * self.clen * n))
        st = C.create_string_buffer(max(1, n))
        assert self.L.orc_prj_batch(self.ctx, n, scalars, slen, points, out, st) == 0
        return out.raw[:3 * self.clen * n], st.raw[:n]

    def y_from_x(self, xs):
        n = len(xs) // self.clen
        y1, y2, st = C.create_string_buffer(max(1, self.clen * n)), C.create_string_buffer(max(1, self.clen * n)), C.create_string_buffer(max(1, n))
        assert self.L.orc_y_from_x_batch(self.ctx, n, xs, y1, y2, st) == 0
        return y1.raw[:self.clen * n], y2.raw[:self.clen * n], st.raw[:n]

    def eddsa_verify(self, pubs, sigs, hram, hlen=None):
        """Ed25519 on WEI25519 (hram = SHA-512(dom2 || R || A || PH(M)), 64 bytes) or Ed448 on WEI448
        (hram = SHAKE256(dom4 || R || A || PH(M), 114)), one hash per item"""
        if self.curve == "WEI448":
            n = len(pubs) // 57
            res = C.create_string_buffer(max(1, n))
            assert self.L.orc_eddsa448_verify_batch(self.ctx, n, pubs, sigs, hram, hlen or 114, res) == 0
            return res.raw[:n]
        n = len(pubs) // 32
        res = C.create_string_buffer(max(1, n))
        assert self.L.orc_eddsa25519_verify_batch(self.ctx, n, pubs, sigs, hram, hlen or 64, res) == 0
        return res.raw[:n]


REF_SO = os.path.join(ROOT, "oracle", "_ref", "libecc_ref.so")
HASH_IDS = {"SHA224": 1, "SHA256": 2, "SHA384": 3, "SHA512": 4, "SHA3_224": 5, "SHA3_256": 6,
            "SHA3_384": 7, "SHA3_512": 8}
HASHLIB = {"SHA224": "sha224", "SHA256": "sha256", "SHA384": "sha384", "SHA512": "sha512",
           "SHA3_224": "sha3_224", "SHA3_256": "sha3_256", "SHA3_384": "sha3_384",
           "SHA3_512": "sha3_512"}


def have_ref():
    return os.path.exists(REF_SO)


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def in_slices(fn, n, threads=None):
    """Run fn(lo, hi) over contiguous slices of range(n) on `threads` Python threads and return the results in order.
    The checkers are ctypes calls into re-entrant C (the oracle, the unmodified reference): they release the GIL, so
    large reference samples (2^16 items) finish in seconds on the GPU box's host cores."""
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(threads or host_threads(), n or 1))
    cuts = [(n * t // threads, n * (t + 1) // threads) for t in range(threads)]
    cuts = [c for c in cuts if c[1] > c[0]]
    if len(cuts) <= 1:
        return [fn(0, n)]
    with ThreadPoolExecutor(max_workers=len(cuts)) as ex:
        return list(ex.map(lambda c: fn(*c), cuts))


def join_slices(parts):
    """concatenate per-slice results: bytes, or tuples of bytes"""
    if isinstance(parts[0], tuple):
        return tuple(b"".join(p[k] for p in parts) for k in range(len(parts[0])))
    return b"".join(parts)


class RefLib:
    """The unmodified reference, through oracle/ref_driver.c."""

    def __init__(self, curve):
        self.L = C.CDLL(REF_SO)
        self.curve = curve
        self.name = curve.encode()
        self.clen, self.qlen = clen(curve), qlen(curve)
        self.nl = (CURVES[curve]["p"].bit_length() + 63) // 64
        assert self.L.refdrv_coord_len(self.name) == self.clen

    def scalar_mult(self, scalars, points=None, slen=None, nthreads=1, timing=False):
        slen = slen or self.qlen
        n = len(scalars) // slen
        out = C.create_string_buffer(2 * self.clen * n)
        st = C.create_string_buffer(n)
        el, ms = C.c_double(), C.c_double()
        r = self.L.refdrv_scalar_mult_batch(self.name, C.c_uint32(n), scalars, C.c_uint32(slen), points,
                                            out, st, nthreads, C.byref(el), C.byref(ms))
        assert r == 0
        if timing:
            return out.raw, st.raw, el.value, ms.value
        return out.raw, st.raw

    def pt_add(self, p1, p2=None):
        n = len(p1) // (2 * self.clen)
        out = C.create_string_buffer(2 * self.clen * n)
        st = C.create_string_buffer(n)
        assert self.L.refdrv_pt_add_batch(self.name, n, p1, p2, out, st, 1 if p2 is None else 0) == 0
        return out.raw, st.raw

    def pt_op_fmt(self, op, p1, p2, in_fmt, out_fmt):
        iw, ow = (3 if in_fmt else 2) * self.clen, (1 if op >= 4 else (3 if out_fmt else 2) * self.clen)
        n = len(p1) // iw
        out, st = C.create_string_buffer(max(1, ow * n)), C.create_string_buffer(max(1, n))
        assert self.L.refdrv_pt_op_batch_fmt(self.name, op, n, p1, p2, in_fmt, out, out_fmt, st) == 0
        return (b"" if op == 2 else out.raw[:ow * n]), st.raw[:n]

    def unprotected_mult(self, scalars, slen, points, in_fmt, out_fmt, broadcast=False):
        iw, ow = (3 if in_fmt else 2) * self.clen, (3 if out_fmt else 2) * self.clen
        n = len(points) // iw
        out, st = C.create_string_buffer(max(1, ow * n)), C.create_string_buffer(max(1, n))
        assert self.L.refdrv_unprotected_mult_batch(self.name, n, scalars, slen, 0 if broadcast else slen, points, in_fmt, out, out_fmt, st) == 0
        return out.raw[:ow * n], st.raw[:n]

    def random_mod(self, raw):
        """the reference's nn_get_random_mod with get_random replaying `raw` (2 * qlen bytes per item)"""
        n = len(raw) // (2 * self.qlen)
        out = C.create_string_buffer(max(1, self.qlen * n))
        assert self.L.refdrv_random_mod_batch(self.name, n, raw, out) == 0
        return out.raw[:self.qlen * n]

    def fp_op(self, op, a, b):
        n = len(a)
        A = (C.c_uint64 * (n * self.nl))(*[(x >> (64 * k)) & (2**64 - 1) for x in a for k in range(self.nl)])
        B = (C.c_uint64 * (n * self.nl))(*[(x >> (64 * k)) & (2**64 - 1) for x in b for k in range(self.nl)])
        O = (C.c_uint64 * (n * self.nl))()
        assert self.L.refdrv_fp_op_batch(self.name, op, n, self.nl, A, B, O) == 0
        return [sum(O[i * self.nl + k] << (64 * k) for k in range(self.nl)) for i in range(n)]

    def ecdsa_verify(self, hash_name, pubs, sigs, msgs, msg_len):
        n = len(pubs) // (2 * self.clen)
        res = C.create_string_buffer(n)
        assert self.L.refdrv_ecdsa_verify_batch(self.name, HASH_IDS[hash_name], n, pubs, sigs, msgs,
                                                msg_len, res) == 0
        return res.raw

    def ecdsa_sign(self, hash_name, privs, nonces, msgs, msg_len):
        n = len(privs) // self.qlen
        sigs = C.create_string_buffer(2 * self.qlen * n)
        pubs = C.create_string_buffer(2 * self.clen * n)
        st = C.create_string_buffer(n)
        assert self.L.refdrv_ecdsa_sign_batch(self.name, HASH_IDS[hash_name], n, privs, nonces, msgs,
                                              msg_len, sigs, pubs, st) == 0
        return sigs.raw, pubs.raw, st.raw

    def ecccdh(self, privs, peers):
        n = len(privs) // self.qlen
        sec = C.create_string_buffer(self.clen * n)
        st = C.create_string_buffer(n)
        assert self.L.refdrv_ecccdh_batch(self.name, n, privs, peers, sec, st) == 0
        return sec.raw, st.raw


def ref_y_from_x(curve, xs):
    """aff_pt_y_from_x of the unmodified reference: (y1, y2, status)"""
    L = C.CDLL(REF_SO)
    cl = clen(curve)
    n = len(xs) // cl
    y1, y2, st = C.create_string_buffer(max(1, cl * n)), C.create_string_buffer(max(1, cl * n)), C.create_string_buffer(max(1, n))
    assert L.refdrv_y_from_x_batch(curve.encode(), n, xs, y1, y2, st) == 0
    return y1.raw[:cl * n], y2.raw[:cl * n], st.raw[:n]


def ref_structured_key_pairs(curve, keys, klen, alg):
    """ec_structured_key_pair_import_from_priv_key_buf: (private scalars n x qlen, affine public keys, status 0 / 1 / 2)"""
    L = C.CDLL(REF_SO)
    cl, ql = clen(curve), qlen(curve)
    n = len(keys) // klen
    pv, pb, st = C.create_string_buffer(max(1, ql * n)), C.create_string_buffer(max(1, 2 * cl * n)), C.create_string_buffer(max(1, n))
    assert L.refdrv_structured_key_pair_batch(curve.encode(), alg, n, keys, klen, pv, pb, st) == 0
    return pv.raw[:ql * n], pb.raw[:2 * cl * n], st.raw[:n]


def ref_structured_sigs(sigs, slen):
    """ec_structured_sig_import_from_buf: (raw signatures, the three header bytes it hands back per item, status)"""
    L = C.CDLL(REF_SO)
    n = len(sigs) // slen
    raw, hdr, st = C.create_string_buffer(max(1, (slen - 3) * n)), C.create_string_buffer(max(1, 3 * n)), C.create_string_buffer(max(1, n))
    assert L.refdrv_structured_sig_batch(n, sigs, slen, raw, hdr, st) == 0
    return raw.raw[:(slen - 3) * n], hdr.raw[:3 * n], st.raw[:n]


def ref_xdh(length, k, u):
    """x25519() / x448() of the unmodified reference"""
    L = C.CDLL(REF_SO)
    n = len(k) // length
    out = C.create_string_buffer(length * n)
    st = C.create_string_buffer(n)
    assert L.refdrv_xdh_batch(length, n, k, u, out, st) == 0
    return out.raw, st.raw


def ref_prj(curve, points, scalars=None, slen=None):
    """the unmodified reference on the projective wire format (oracle/ref_driver.c:refdrv_prj_batch)"""
    L = C.CDLL(REF_SO)
    cl = clen(curve)
    n = len(points) // (3 * cl)
    out, st = C.create_string_buffer(max(1, 3 * cl * n)), C.create_string_buffer(max(1, n))
    assert L.refdrv_prj_batch(curve.encode(), n, scalars, slen or qlen(curve), points, out, st) == 0
    return out.raw[:3 * cl * n], st.raw[:n]


def ref_ed25519_sign(seeds, msgs, msg_len):
    """keys from 32-byte seeds and pure-Ed25519 signatures by the unmodified reference"""
    L = C.CDLL(REF_SO)
    n = len(seeds) // 32
    pubs, sigs, st = C.create_string_buffer(32 * n), C.create_string_buffer(64 * n), C.create_string_buffer(n)
    assert L.refdrv_eddsa25519_sign_batch(n, seeds, msgs, msg_len, pubs, sigs, st) == 0
    return pubs.raw, sigs.raw, st.raw


def ref_ed25519_verify(pubs, sigs, msgs, msg_len):
    L = C.CDLL(REF_SO)
    n = len(pubs) // 32
    res = C.create_string_buffer(max(1, n))
    assert L.refdrv_eddsa25519_verify_batch(n, pubs, sigs, msgs, msg_len, res) == 0
    return res.raw[:n]


def ref_eddsa_verify_all(pubs, sigs, msgs, msg_len, ed448=False, scratch=False):
    """the reference's ec_verify_batch on pure Ed25519 / Ed448 signatures: True iff it accepts the whole batch"""
    L = C.CDLL(REF_SO)
    n = len(pubs) // (57 if ed448 else 32)
    ok = C.c_int(0)
    assert L.refdrv_eddsa_verify_batch_all(int(ed448), int(scratch), n, pubs, sigs, msgs, msg_len, C.byref(ok)) == 0
    return bool(ok.value)


def ref_eddsa_export_pub_key(points_prj, ed448=False):
    """libecc's eddsa_export_pub_key on keys given as projective Weierstrass points: (encodings, per-item 0 / -1)"""
    L = C.CDLL(REF_SO)
    cl, kl = (56, 57) if ed448 else (32, 32)
    n = len(points_prj) // (3 * cl)
    enc = C.create_string_buffer(max(1, kl * n))
    ret = (C.c_int * max(1, n))()
    assert L.refdrv_eddsa_export_pub_key_batch(int(ed448), n, points_prj, enc, ret) == 0
    return enc.raw[:kl * n], [ret[i] for i in range(n)]


def digest(hash_name, msg):
    return hashlib.new(HASHLIB[hash_name], msg).digest()


def rfc6979_nonce(curve, hash_name, priv, msg):
    """RFC 6979 section 3.2 (the reference's __ecdsa_rfc6979_nonce, sig/ecdsa_common.c:48-169)."""
    q = CURVES[curve]["q"]
    qbits, rlen = q.bit_length(), (q.bit_length() + 7) // 8
    hname = HASHLIB[hash_name]
    h1 = hashlib.new(hname, msg).digest()
    hl = len(h1)

    def bits2int(b):
        x = int.from_bytes(b, "big")
        return x >> (len(b) * 8 - qbits) if len(b) * 8 > qbits else x

    x = int.from_bytes(priv, "big")
    bx = x.to_bytes(rlen, "big") + (bits2int(h1) % q).to_bytes(rlen, "big")
    V, K = b"\x01" * hl, b"\x00" * hl
    K = hmac.new(K, V + b"\x00" + bx, hname).digest()
    V = hmac.new(K, V, hname).digest()
    K = hmac.new(K, V + b"\x01" + bx, hname).digest()
    V = hmac.new(K, V, hname).digest()
    while True:
        T = b""
        while len(T) < rlen:
            V = hmac.new(K, V, hname).digest()
            T += V
        k = bits2int(T[:rlen])
        if 1 <= k < q:
            return k
        K = hmac.new(K, V + b"\x00", hname).digest()
        V = hmac.new(K, V, hname).digest()


# ---- independent Python affine arithmetic (third opinion, tiny cases only) ----
def py_add(P, Q, a, p):
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        lam = (3 * P[0] * P[0] + a) * pow(2 * P[1], p - 2, p) % p
    else:
        lam = (Q[1] - P[1]) * pow(Q[0] - P[0], p - 2, p) % p
    x = (lam * lam - P[0] - Q[0]) % p
    return (x, (lam * (P[0] - x) - P[1]) % p)


def py_mul(k, P, a, p):
    R = None
    while k:
        if k & 1:
            R = py_add(R, P, a, p)
        P = py_add(P, P, a, p)
        k >>= 1
    return R


def py_smul_bytes(curve, k, P=None):
    c = CURVES[curve]
    P = P or (c["gx"], c["gy"])
    R = py_mul(k % c["order"], P, c["a"], c["p"])
    n = clen(curve)
    return None if R is None else R[0].to_bytes(n, "big") + R[1].to_bytes(n, "big")


def rand_points(curve, rng, n):
    """n affine points [t]G (bytes) via Python ints -- slow, keep n small -- or via Oracle."""
    o = Oracle(curve)
    sc = b"".join(int(rng.integers(1, 2**62)).to_bytes(o.qlen, "big") for _ in range(n))
    pts, st = o.scalar_mult(sc)
    assert set(st) == {0}
    return pts


def ed25519_hram(pubs, sigs, msgs, msg_len):
    """SHA-512(R || A || M) per item (pure Ed25519: empty dom2), the hash the batch verifiers take"""
    import hashlib
    n = len(pubs) // 32
    out = bytearray()
    for i in range(n):
        out += hashlib.sha512(sigs[64 * i:64 * i + 32] + pubs[32 * i:32 * i + 32] +
                              msgs[msg_len * i:msg_len * (i + 1)]).digest()
    return bytes(out)


# ---- a small RFC 8032 Ed25519 signer (test input generator; python ints, extended coordinates) ----
ED_P = 2**255 - 19
ED_Q = 2**252 + 27742317777372353535851937790883648493
ED_D = (-121665 * pow(121666, ED_P - 2, ED_P)) % ED_P
ED_I = pow(2, (ED_P - 1) // 4, ED_P)


def ed_add(P, Q):
    x1, y1, z1, t1 = P
    x2, y2, z2, t2 = Q
    a = (y1 - x1) * (y2 - x2) % ED_P
    b = (y1 + x1) * (y2 + x2) % ED_P
    c = 2 * t1 * t2 * ED_D % ED_P
    d = 2 * z1 * z2 % ED_P
    e, f, g, h = b - a, d - c, d + c, b + a
    return (e * f % ED_P, g * h % ED_P, f * g % ED_P, e * h % ED_P)


def ed_mul(k, P):
    R = (0, 1, 1, 0)
    while k:
        if k & 1:
            R = ed_add(R, P)
        P = ed_add(P, P)
        k >>= 1
    return R


def ed_encode(P):
    zi = pow(P[2], ED_P - 2, ED_P)
    x, y = P[0] * zi % ED_P, P[1] * zi % ED_P
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def ed_decode(b):
    y = int.from_bytes(b, "little")
    sign, y = y >> 255, y & ((1 << 255) - 1)
    if y >= ED_P:
        return None
    x2 = (y * y - 1) * pow(ED_D * y * y + 1, ED_P - 2, ED_P) % ED_P
    x = pow(x2, (ED_P + 3) // 8, ED_P)
    if (x * x - x2) % ED_P:
        x = x * ED_I % ED_P
    if (x * x - x2) % ED_P or (x == 0 and sign):
        return None
    if (x & 1) != sign:
        x = ED_P - x
    return (x, y, 1, x * y % ED_P)


ED_B = ed_decode((4 * pow(5, ED_P - 2, ED_P) % ED_P).to_bytes(32, "little"))
ED_TORSION8 = bytes.fromhex("26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05")


def ed_dom2(flag, ctx):
    return b"SigEd25519 no Ed25519 collisions" + bytes([flag, len(ctx)]) + ctx


def ed25519_sign(seed, msg, dom=b"", prehash=False, add_R=None, add_A=None, r_enc=None):
    """RFC 8032 5.1.6; add_R / add_A (points) shift R or the public key by a torsion point, giving
    signatures that only the cofactored verification equation accepts.  r_enc: sign with r = 0 and put these
    bytes where R goes (an encoding of the neutral element: valid for a verifier that decodes it to infinity).
    Returns (A, R || S, hram)."""
    hk = hashlib.sha512(seed).digest()
    a = int.from_bytes(hk[:32], "little")
    a = (a & ((1 << 254) - 8)) | (1 << 254)
    m = hashlib.sha512(msg).digest() if prehash else msg
    A = ed_mul(a, ED_B)
    if add_A is not None:
        A = ed_add(A, add_A)
    Aenc = ed_encode(A)
    r = int.from_bytes(hashlib.sha512(dom + hk[32:] + m).digest(), "little") % ED_Q
    R = ed_mul(r, ED_B)
    if add_R is not None:
        R = ed_add(R, add_R)
    Renc = ed_encode(R)
    if r_enc is not None:
        r, Renc = 0, bytes(r_enc)
    hram = hashlib.sha512(dom + Renc + Aenc + m).digest()
    S = (r + int.from_bytes(hram, "little") * a) % ED_Q
    return Aenc, Renc + S.to_bytes(32, "little"), hram


# ---- a small RFC 8032 Ed448 signer (test input generator; python ints, projective coordinates) ----
E4_P = 2**448 - 2**224 - 1
E4_Q = 2**446 - 13818066809895115352007386748515426880336692474882178609894547503885
E4_D = (-39081) % E4_P


def e4_add(P, Q):
    # add-2008-bbjlp, a = 1
    x1, y1, z1 = P
    x2, y2, z2 = Q
    a = z1 * z2 % E4_P
    b = a * a % E4_P
    c = x1 * x2 % E4_P
    d = y1 * y2 % E4_P
    e = E4_D * c * d % E4_P
    f, g = (b - e) % E4_P, (b + e) % E4_P
    x3 = a * f * ((x1 + y1) * (x2 + y2) - c - d) % E4_P
    y3 = a * g * (d - c) % E4_P
    return (x3, y3, f * g % E4_P)


def e4_mul(k, P):
    R = (0, 1, 1)
    while k:
        if k & 1:
            R = e4_add(R, P)
        P = e4_add(P, P)
        k >>= 1
    return R


def e4_encode(P):
    zi = pow(P[2], E4_P - 2, E4_P)
    x, y = P[0] * zi % E4_P, P[1] * zi % E4_P
    return (y | ((x & 1) << 455)).to_bytes(57, "little")


def e4_decode(b):
    y = int.from_bytes(b, "little")
    sign, y = y >> 455, y & ((1 << 455) - 1)
    if y >= E4_P:
        return None
    u, v = (y * y - 1) % E4_P, (E4_D * y * y - 1) % E4_P
    x2 = u * pow(v, E4_P - 2, E4_P) % E4_P
    x = pow(x2, (E4_P + 1) // 4, E4_P)
    if (x * x - x2) % E4_P or (x == 0 and sign):
        return None
    if (x & 1) != sign:
        x = E4_P - x
    return (x, y, 1)


E4_B = e4_decode(bytes.fromhex("14fa30f25b790898adc8d74e2c13bdfdc4397ce61cffd33ad7c2a0051e9c78874098a36c7373ea4b"
                               "62c7c9563720768824bcb66e71463f6900"))


def ed_dom4(flag, ctx):
    return b"SigEd448" + bytes([flag, len(ctx)]) + ctx


def ed448_sign(seed, msg, ctx=b"", prehash=False, add_R=None, add_A=None, r_enc=None):
    """RFC 8032 5.2.6; add_R / add_A shift R or the public key by a torsion point; r_enc as in ed25519_sign.
    Returns (A, R || S, hram) with hram = SHAKE256(dom4 || R || A || PH(M), 114)."""
    def H(x):
        return hashlib.shake_256(x).digest(114)
    hk = H(seed)
    ab = bytearray(hk[:57])
    ab[0] &= 0xFC
    ab[55] |= 0x80
    ab[56] = 0
    a = int.from_bytes(ab, "little")
    m = hashlib.shake_256(msg).digest(64) if prehash else msg
    dom = ed_dom4(1 if prehash else 0, ctx)
    A = e4_mul(a, E4_B)
    if add_A is not None:
        A = e4_add(A, add_A)
    Aenc = e4_encode(A)
    r = int.from_bytes(H(dom + hk[57:] + m), "little") % E4_Q
    R = e4_mul(r, E4_B)
    if add_R is not None:
        R = e4_add(R, add_R)
    Renc = e4_encode(R)
    if r_enc is not None:
        r, Renc = 0, bytes(r_enc)
    hram = H(dom + Renc + Aenc + m)
    S = (r + int.from_bytes(hram, "little") * a) % E4_Q
    return Aenc, Renc + S.to_bytes(57, "little"), hram


def ref_ed448_sign(seeds, msgs, msg_len):
    L = C.CDLL(REF_SO)
    n = len(seeds) // 57
    pubs, sigs, st = C.create_string_buffer(57 * n), C.create_string_buffer(114 * n), C.create_string_buffer(n)
    assert L.refdrv_eddsa448_sign_batch(n, seeds, msgs, msg_len, pubs, sigs, st) == 0
    return pubs.raw, sigs.raw, st.raw


def ref_ed448_verify(pubs, sigs, msgs, msg_len):
    L = C.CDLL(REF_SO)
    n = len(pubs) // 57
    res = C.create_string_buffer(max(1, n))
    assert L.refdrv_eddsa448_verify_batch(n, pubs, sigs, msgs, msg_len, res) == 0
    return res.raw[:n]


def structured_pub_expect(curve, keys, alg):
    """what ec_structured_pub_key_import_from_buf (sig/ec_key.c:312) does with each key, from the oracle's pieces:
    header EC_PUBKEY (0) / alg / ec_curve_type, prj_pt_import_from_buf + normalisation, subgroup check on cofactor
    curves.  Returns (affine bytes, status)."""
    c = CURVES[curve]
    cl = clen(curve)
    klen = 3 + 3 * cl
    n = len(keys) // klen
    o = Oracle(curve)
    pts = b"".join(keys[klen * i + 3:klen * (i + 1)] for i in range(n))
    uni, st = o.prj(pts)
    out, status = bytearray(), bytearray()
    for i in range(n):
        k = keys[klen * i:klen * (i + 1)]
        s = st[i]
        aff = uni[3 * cl * i:3 * cl * i + 2 * cl]
        if s == 0 and c["order"] != c["q"]:
            _, s2 = o.scalar_mult(c["q"].to_bytes(o.qlen, "big"), aff)
            if s2[0] != 2:
                s = 1
        if s == 2 and c["order"] != c["q"] and k[3:] == bytes(3 * cl):
            s = 1   # check_prj_pt_order on the degenerate (0 : 0 : 0) fails in its additions; (0 : Y : 0) passes
        if k[0] != 0 or k[1] != alg or k[2] != c["type"]:
            s = 1
        status.append(s)
        out += aff if s == 0 else bytes(2 * cl)
    return bytes(out), bytes(status)


def ref_structured_pub_import(curve, keys, alg):
    L = C.CDLL(REF_SO)
    cl = clen(curve)
    klen = 3 + 3 * cl
    n = len(keys) // klen
    out, st = C.create_string_buffer(max(1, 2 * cl * n)), C.create_string_buffer(max(1, n))
    assert L.refdrv_structured_pub_import_batch(curve.encode(), alg, n, keys, klen, out, st) == 0
    return out.raw[:2 * cl * n], st.raw[:n]


# ---- the reference's own batch verifier (ec_verify_batch, sig/sig_algs.c:675) on raw keys / signatures / messages ----
def ref_sig_verify_all(curve, alg, hash_name, pubs_aff, sigs, sig_len, msgs, msg_len, scratch=False, per_item=False, sub_batches=1):
    """ec_verify_batch of the unmodified reference (BIP0340 -> bip0340_verify_batch sig/bip0340.c:1296, ECFSDSA -> ecfsdsa_verify_batch
    sig/ecfsdsa.c:1057) on n items.  sub_batches = 1: ONE call over the whole batch, as an application makes it.  sub_batches > 1: the batch
    is cut into that many contiguous pieces, the reference's batch function runs on each piece on its own host thread, and the verdict is
    the conjunction (libecc's verifier is single-threaded: a 2^20-item call takes a quarter of an hour; the conjunction of its verdicts
    on the pieces is the same predicate up to the 2^-128 of each random combination).  Returns all_valid, or (all_valid, per-item bytes of
    a loop of ec_verify) with per_item."""
    L = C.CDLL(REF_SO)
    cl = clen(curve)
    n = len(pubs_aff) // (2 * cl)
    alg_id = L.refdrv_alg_id(alg.encode())
    assert alg_id >= 0

    def piece(lo, hi):
        ok = C.c_int(0)
        res = C.create_string_buffer(max(1, hi - lo)) if per_item else None
        assert L.refdrv_sig_verify_batch_all(curve.encode(), alg_id, HASH_IDS[hash_name], int(scratch), hi - lo, pubs_aff[2 * cl * lo:2 * cl * hi],
                                             sigs[sig_len * lo:sig_len * hi], sig_len, msgs[msg_len * lo:msg_len * hi], msg_len,
                                             C.byref(ok), res) == 0
        return bool(ok.value), (res.raw[:hi - lo] if per_item else b"")
    if sub_batches <= 1:
        parts = [piece(0, n)]
    else:
        parts = in_slices(piece, n, threads=min(sub_batches, n))
    all_ok = all(p[0] for p in parts)
    return (all_ok, b"".join(p[1] for p in parts)) if per_item else all_ok


def make_bip0340_batch(fixed_base_mult, curve, n, rng, msg_len=32, hash_name="SHA256"):
    import numpy as np
    """n valid BIP0340 signatures made from Python integers (sig/bip0340.c:135-330 restated: d = x or q - x so that [d]G has an even y,
    R = [k]G with k negated when R.y is odd, e = H(H(tag) || H(tag) || R.x || P.x || m) mod q, s = k + e d): the two fixed-base
    multiplications per item come from `fixed_base_mult(scalars) -> (affine points, status)` (the GPU library in the GPU tests and in
    bench.py -- the signatures are then CHECKED by the unmodified reference, so nothing trusts that path -- or the CPU oracle).
    Returns the application's view (pubs = affine [x]G as generated, sigs = r || s, msgs) and the multi-scalar form's inputs
    (s, ne = q - e, keys = the even-y representative, rx)."""
    c = CURVES[curve]
    q, p = c["q"], c["p"]
    cl, ql = clen(curve), qlen(curve)
    raw = rng.integers(0, 256, size=(2, n, ql + 8), dtype=np.uint8)
    x = [(int.from_bytes(raw[0, i].tobytes(), "big") % (q - 1)) + 1 for i in range(n)]
    k = [(int.from_bytes(raw[1, i].tobytes(), "big") % (q - 1)) + 1 for i in range(n)]
    msgs = rng.integers(0, 256, size=n * msg_len, dtype=np.uint8).tobytes()
    P, st = fixed_base_mult(b"".join(v.to_bytes(ql, "big") for v in x))
    R, st2 = fixed_base_mult(b"".join(v.to_bytes(ql, "big") for v in k))
    assert set(st) == {0} and set(st2) == {0}
    hf = getattr(hashlib, HASHLIB[hash_name])
    tagd = hf(b"BIP0340/challenge").digest()
    pre = hf(tagd + tagd)
    sigs, s_l, ne_l, keys, rx = bytearray(), bytearray(), bytearray(), bytearray(), bytearray()
    for i in range(n):
        Px, Py = P[2 * cl * i:2 * cl * i + cl], P[2 * cl * i + cl:2 * cl * (i + 1)]
        Rx, Ry = R[2 * cl * i:2 * cl * i + cl], R[2 * cl * (i + 1) - 1]
        d = x[i] if not (Py[-1] & 1) else q - x[i]
        kk = k[i] if not (Ry & 1) else q - k[i]
        h = pre.copy()
        h.update(Rx + Px + msgs[msg_len * i:msg_len * (i + 1)])
        e = int.from_bytes(h.digest(), "big") % q
        s = ((kk + e * d) % q).to_bytes(ql, "big")
        sigs += Rx + s
        s_l += s
        ne_l += ((q - e) % q).to_bytes(ql, "big")
        keys += Px + (Py if not (Py[-1] & 1) else (p - int.from_bytes(Py, "big")).to_bytes(cl, "big"))
        rx += Rx
    return {"pubs": P, "sigs": bytes(sigs), "sig_len": cl + ql, "msgs": msgs, "msg_len": msg_len, "s": bytes(s_l), "ne": bytes(ne_l),
            "keys": bytes(keys), "rx": bytes(rx), "n": n, "cl": cl, "ql": ql, "q": q, "p": p}
