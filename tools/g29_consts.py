#!/usr/bin/env python3
"""Python mirror of the CurveG<NL> constant image of libecc_amd/csrc/ecamd_u29g.h (what
ecamd_host.cpp uploads per curve).  Used by tests/test_u29g_host.py to drive the host build of the
generic radix-2^29 code; tests/test_gpu_parity.py cross-checks the C++ builder through the GPU."""
W = 29
MASK = (1 << W) - 1
BIAS_STEP = [2, 4, 6, 8, 10, 12, 14, 16] * 2
BIAS_S = [1] * 8 + [2] * 8


def nl_for(pbits):
    return (pbits + 16 + W - 1) // W


def width(flavour):
    """limb width of a flavour: the Goldilocks unit (5) runs on 28-bit limbs (2^224 on a limb boundary), all others on 29"""
    return 28 if flavour == 5 else W


def bias_steps(flavour):
    """the no-headroom flavours (1: secp521r1 on plain residues, 5: Goldilocks) use 2p, 4p, 8p ...; the others 4p, 16p, 64p ..."""
    return ([1, 2, 3, 4, 5, 6, 7, 8] if flavour in (1, 5) else [2, 4, 6, 8, 10, 12, 14, 16]) * 2


def digits(x, nl, w=W):
    d = [(x >> (w * i)) & ((1 << w) - 1) for i in range(nl - 1)]
    d.append(x >> (w * (nl - 1)))
    return d


def image(p, a, b, flavour=0, iso_u=None):
    """flavour 0: dense Montgomery; 1: p = 2^521 - 1 on plain residues, 18 limbs (2^522 = 2); 2: p = 2^255 - 19, nine limbs
    and plain residues (R = 1); 4: p = 2^256 - 2^32 - 977 (secp256k1), the same shape; 5: p = 2^448 - 2^224 - 1, plain residues on 16 limbs
    of 28 bits"""
    pbits = p.bit_length()
    plain = flavour in (1, 2, 4, 5)
    w = width(flavour)
    nl = 9 if flavour in (2, 4) else (16 if flavour == 5 else (18 if flavour == 1 else nl_for(pbits)))
    R = 1 if plain else 1 << (w * nl)
    topsh = pbits - w * (nl - 1)
    off = max(0, 1 - topsh)
    out = []
    out += digits(p, nl, w)
    out += digits(R * R % p, nl, w)
    out += digits(R % p, nl, w)
    if iso_u is not None:
        # the unit computes on the isomorphic curve (x, y) -> (u^2 x, u^3 y): a u^4, b u^6 (ecamd_host.cpp:upload_g29 does this when a u^4 = -3)
        a, b = a * pow(iso_u, 4, p) % p, b * pow(iso_u, 6, p) % p
    out += digits(a * R % p, nl, w)
    out += digits(b * R % p, nl, w)
    out += digits(p - 2, nl, w)
    for step, s in zip(bias_steps(flavour), BIAS_S):
        c = p << (step + off)
        if c.bit_length() > w * (nl - 1) + 32:
            out += [0] * nl   # does not fit the limbs (only without a headroom limb); never selected
            continue
        d = digits(c, nl, w)
        M, BW = 1 << (w + s), 1 << s
        l = [d[0] + M] + [d[j] + M - BW for j in range(1, nl - 1)] + [d[nl - 1] - BW]
        if l[-1] < 0:
            out += [0] * nl   # the top digit cannot lend the borrow (2p on the Goldilocks unit with S = 2); never selected
            continue
        assert sum(v << (w * j) for j, v in enumerate(l)) == c
        assert all(v < 2**32 for v in l)
        out += l
    # coordinate import / export factors (no isomorphism here: R^2, R^2, 1, 1)
    if iso_u is None:
        out += digits(R * R % p, nl, w) * 2 + digits(1, nl, w) * 2
    else:
        ui = pow(iso_u, -1, p)
        out += digits(iso_u**2 * R * R % p, nl, w) + digits(iso_u**3 * R * R % p, nl, w) + digits(ui**2 % p, nl, w) + digits(ui**3 % p, nl, w)
    mpinv = (-pow(p, -1, 1 << w)) % (1 << w)
    out += [mpinv, pbits, 1 if a == p - 3 else 0, 1 if a == 0 else 0]
    return out, nl
