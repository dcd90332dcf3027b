#!/bin/bash
# GPU box: the bench's host-fed regions under GPU_MAX_HW_QUEUES = default (4) / 8 / 16 (the ROCm runtime maps streams onto that many
# hardware queues per device; read when the runtime starts), two runs each.  usage: bash tools/hw_queues_probe.sh > profiles/r04_hw_queues.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "{"
first=1
for q in default 8 16 default 8 16; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-single-file-100k --sustained-seconds 0 --no-live-pmc > /tmp/hq_$q.json 2> /tmp/hq_$q.err
  [ $first = 1 ] || echo ","
  first=0
  python - "$q" <<'PY'
import json, sys, time
q = sys.argv[1]
try:
    d = json.load(open(f"/tmp/hq_{q}.json"))
    r = d["regions"]
    print(json.dumps(q + "_" + str(int(time.time()) % 100000)) + ": " + json.dumps({"value_ms_per_step": d["ms_per_step"], "value_four_streams_ms": d["value_four_streams"]["ms_per_step"],
          "e2e_steady_ms_per_batch": d["e2e_steady"]["ms_per_batch"], "single_file_10k_ms": r["single_file_10k"]["ms_per_batch"],
          "single_file_10k_ms_on_the_resident_steps_context": r["single_file_10k"]["first_call"]["ms_per_call_on_the_context_of_the_resident_steps"],
          "first_call": {k: r["single_file_10k"]["first_call"][k] for k in ("ms", "rehearsal_ms", "rehearsals", "lanes", "queue_shared")},
          "facade_ms_per_file": r["single_file_10k"]["facade_ms_per_file"], "h2d_kernels_d2h_ms": r["h2d_kernels_d2h"]["ms_per_batch"],
          "decode_steady_frames_per_s": r["decode_steady"]["frames_per_s"], "long_message_steady_ms": d["long_message"]["steady"]["ms_per_batch"],
          "parity_checked": d["parity_checked"]}), end="")
except Exception as e:
    print(json.dumps(q) + ": " + json.dumps({"error": str(e), "stderr": open(f"/tmp/hq_{q}.err").read()[-400:]}), end="")
PY
done
echo
echo "}"
