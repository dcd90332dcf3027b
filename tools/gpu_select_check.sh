#!/bin/bash
# the encoder tests with and without the device-side cursor selection, and how often the first pass is final.  usage: bash tools/gpu_select_check.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/guess_rate_probe.py | tr -d "\n "; echo
MP3S_NO_SELECT=1 python tools/guess_rate_probe.py | tr -d "\n "; echo
