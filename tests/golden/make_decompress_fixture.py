#!/usr/bin/env python3
"""Record the UNMODIFIED reference's aff_pt_y_from_x answers (both roots in fp_sqrt's order, and the failures) so that
this pin travels without oracle/_ref:  python tests/golden/make_decompress_fixture.py  -> tests/golden/decompress_fixture.json"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracles as O  # noqa: E402

CURVES = ["SECP192R1", "SECP224R1", "SECP256R1", "SECP256K1", "BRAINPOOLP256R1", "SECP384R1", "SECP521R1", "WEI25519", "WEI448"]


def inputs(curve, rng, n=24):
    c = O.CURVES[curve]
    p, cl = c["p"], O.clen(curve)
    xs = [int.from_bytes(rng.integers(0, 256, size=cl + 8, dtype=np.uint8).tobytes(), "big") % p for _ in range(n)]
    xs += [0, 1, 2, p - 1, p, p + 1 if (p + 1) >> (8 * cl) == 0 else p, (1 << (8 * cl)) - 1, c["gx"]]
    return b"".join((x & ((1 << (8 * cl)) - 1)).to_bytes(cl, "big") for x in xs)


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    out = {}
    rng = np.random.default_rng(3001)
    for curve in CURVES:
        xs = inputs(curve, rng)
        y1, y2, st = O.ref_y_from_x(curve, xs)
        out[curve] = {"x": xs.hex(), "y1": y1.hex(), "y2": y2.hex(), "status": st.hex()}
    json.dump(out, open(os.path.join(HERE, "decompress_fixture.json"), "w"), indent=1)
    print("wrote", list(out))


if __name__ == "__main__":
    main()
