#!/bin/bash
# round 2, GPU pass r: X448 x-only ladder on the Goldilocks unit
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py -x -q -m gpu -k "xdh or x448 or x25519 or 448 or secret" 2>&1 | tail -n 30 > $O/pytest.log
tail -n 4 $O/pytest.log
cat > /tmp/x448_time.py <<'PY'
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import libecc_amd
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(device=dev)
rng = np.random.default_rng(3); n = 1 << 20
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
import hashlib
print(json.dumps({"ms_per_2^20": ms, "x448_per_s": n / (ms * 1e-3), "rejected": int(st.sum().item()), "sha256_of_outputs": hashlib.sha256(bytes(o.cpu().numpy())).hexdigest()}))
PY
echo "ladder_2waves $(python /tmp/x448_time.py)" | tee $O/x448_ladder.txt
echo "ladder_1wave $(ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_x448occ1.so python /tmp/x448_time.py)" | tee -a $O/x448_ladder.txt
echo "window_path $(ECAMD_NO_X448_LADDER=1 python /tmp/x448_time.py)" | tee -a $O/x448_ladder.txt
