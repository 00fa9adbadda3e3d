#!/bin/bash
# round 2, GPU pass p: X448 front end on the Goldilocks unit
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2p
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "xdh or x448 or x25519 or 448" 2>&1 | tail -n 30 > $O/pytest.log
tail -n 4 $O/pytest.log
python - <<'PY' > $O/x448.json
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import libecc_amd
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(device=dev)
rng = np.random.default_rng(3); n = 1 << 20
out = {}
for mode in ("goldilocks_front_end", "saturated_front_end"):
    if mode == "saturated_front_end":
        os.environ["ECAMD_NO_G448_DECODE"] = "1"
    ctx = libecc_amd.Context(0); cv = ctx.curve("WEI448")
    k = torch.frombuffer(bytearray(rng.integers(0, 256, size=56 * n, dtype=np.uint8).tobytes()), dtype=torch.uint8).to(dev)
    u = torch.frombuffer(bytearray((5).to_bytes(56, "little") * n), dtype=torch.uint8).to(dev)
    o = torch.empty(56 * n, dtype=torch.uint8, device=dev); st = torch.empty(n, dtype=torch.uint8, device=dev)
    f = lambda: cv.xdh_dev(n, k.data_ptr(), u.data_ptr(), o.data_ptr(), st.data_ptr(), stream.cuda_stream)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(3): f()
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    out[mode] = {"ms_per_2^20": ms, "x448_per_s": n / (ms * 1e-3), "rejected": int(st.sum().item())}
    cv.free(); ctx.close()
print(json.dumps(out, indent=1))
PY
cat $O/x448.json
