#!/bin/bash
# the first multi-GPU lease: correctness of both multi-GPU layers and the 1/2/4/8 curve in one pass (tools/scale_check.py)
cd "$(dirname "$0")/.." && exec python tools/scale_check.py "$@"
