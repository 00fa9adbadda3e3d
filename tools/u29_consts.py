#!/usr/bin/env python3
"""Derive (and print as C initialisers) the radix-2^29 constants of libecc_amd/csrc/ecamd_u29*.h
from the P-256 domain parameters.  tests/test_u29_host.py re-derives them and compares with what the
header holds, so nothing here needs to be trusted."""
W, NL = 29, 9
MASK = (1 << W) - 1
p = 2**256 - 2**224 + 2**192 + 2**96 - 1
b = 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B
R = 1 << (W * NL)


def digits(x, n=NL):
    d = [(x >> (W * i)) & MASK for i in range(n - 1)]
    d.append(x >> (W * (n - 1)))
    return d


def fmt(name, x):
    return f"static constexpr u32 {name}[9] = {{" + ", ".join(f"0x{v:08x}" for v in digits(x)) + "};"


def inv_chain():
    """addition chain for x^(p-2); returns list of ops and checks the exponent"""
    e = {}
    e["x"] = 1
    ops = []

    def sqn(src, n):
        return e[src] << n

    e["e2"] = (e["x"] << 1) + e["x"]
    e["e4"] = (e["e2"] << 2) + e["e2"]
    e["e8"] = (e["e4"] << 4) + e["e4"]
    e["e16"] = (e["e8"] << 8) + e["e8"]
    e["e32"] = (e["e16"] << 16) + e["e16"]
    r = (e["e32"] << 32) + e["x"]
    r = (r << 128) + e["e32"]
    r = (r << 32) + e["e32"]
    r = (r << 16) + e["e16"]
    r = (r << 8) + e["e8"]
    r = (r << 4) + e["e4"]
    r = (r << 2) + e["e2"]
    r = (r << 2) + e["x"]
    assert r == p - 2, hex(r ^ (p - 2))
    return r


if __name__ == "__main__":
    print("p   ", [hex(v) for v in digits(p)])
    print("p+1 ", [hex(v) for v in digits(p + 1)])
    print("2^256 mod p", [hex(v) for v in digits((1 << 256) % p)])
    print(fmt("R2", R * R % p))
    print(fmt("ONE", R % p))
    print(fmt("BM", b * R % p))
    print(fmt("THREE_M", 3 * R % p))
    inv_chain()
    print("inversion chain ok")
