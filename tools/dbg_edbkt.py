import os, sys, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import libecc_amd
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
rng = np.random.default_rng(3)
ctx = libecc_amd.Context(0)
cv = ctx.curve("WEI25519")
rb = lambda k: rng.integers(0, 256, size=k, dtype=np.uint8).tobytes()
seeds, msgs = rb(32 * n), rb(32 * n)
hk = [hashlib.sha512(seeds[32 * i:32 * i + 32]).digest() for i in range(n)]
a_np = np.frombuffer(b"".join(h[:32] for h in hk), dtype=np.uint8).reshape(n, 32).copy()
a_np[:, 0] &= 248; a_np[:, 31] &= 127; a_np[:, 31] |= 64
wide = np.zeros((n, 64), dtype=np.uint8); wide[:, :32] = a_np
pubs, st = cv.eddsa_sign_R(wide.tobytes())
r_hash = b"".join(hashlib.sha512(hk[i][32:] + msgs[32 * i:32 * i + 32]).digest() for i in range(n))
Renc, st = cv.eddsa_sign_R(r_hash)
hram = b"".join(hashlib.sha512(Renc[32 * i:32 * i + 32] + pubs[32 * i:32 * i + 32] + msgs[32 * i:32 * i + 32]).digest() for i in range(n))
Sb = cv.eddsa_sign_S(r_hash, hram, a_np.tobytes())
sg = np.empty((n, 64), dtype=np.uint8); sg[:, :32] = np.frombuffer(Renc, dtype=np.uint8).reshape(n, 32); sg[:, 32:] = np.frombuffer(Sb, dtype=np.uint8).reshape(n, 32)
sigs = sg.tobytes()
ctx.set_eddsa_msm(2, 0, 0)
dev = torch.device("cuda:0")
for algo in ("bucket",):
    os.environ["ECAMD_ED_MSM_ALGO"] = algo
    print(algo, "host form:", cv.eddsa_verify_all(pubs, sigs, hram))
    for use_stream in (False, True):
        stream = torch.cuda.Stream(device=dev) if use_stream else None
        t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        dp, ds, dh = t(pubs), t(sigs), t(hram)
        verdict = torch.full((1,), 7, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for rep in range(3):
            cv.eddsa_verify_all_dev(n, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), verdict.data_ptr(), stream.cuda_stream if stream else None)
            torch.cuda.synchronize()
            print(algo, "dev form, own stream" if use_stream else "dev form, ctx stream", "rep", rep, int(verdict.item()))
P = 2**255 - 19
seed = bytes(range(32))
for m in (65536, 90000, 98304, 131072, 200000, 262144):
    if m > n: break
    res = {}
    for algo in ("straus", "bucket"):
        os.environ["ECAMD_ED_MSM_ALGO"] = algo
        acc, z, T = cv.debug_eddsa_msm(pubs[:32 * m], sigs[:64 * m], hram[:64 * m], seed)
        X, Y, Z, _ = T
        zi = pow(Z, P - 2, P)
        res[algo] = (acc, X * zi % P, Y * zi % P)
    print(m, "accept", res["straus"][0], res["bucket"][0], "same point", res["straus"][1:] == res["bucket"][1:])
