#!/bin/bash
# quick kernel iteration on the GPU box: resident steps only, per-kernel table + parity flag.  usage: bash tools/kb.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --resident-only --no-cpu-baseline --steps 100 "$@" > /tmp/kb.json 2> /tmp/kb.err || { tail -20 /tmp/kb.err; }
python - <<'PY'
import json
b = json.load(open('/tmp/kb.json'))
print("value", b["value"], "ms/step", b["ms_per_step"], "parity", b["parity_checked"])
print({k: v for k, v in b["kernels_ms_per_step"].items()})
PY
