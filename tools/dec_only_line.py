import json,sys
b=json.load(open(sys.argv[1]))
d=b["decode_only"]; print("decode_only exact ms", d["ms_per_step"], "serial", d.get("serial_ms_per_step"), "fast", d["float_fast"]["ms_per_step"], "serial", d["float_fast"].get("serial_ms_per_step"), "parity", b["parity_checked"], "value", b["value"], b["ms_per_step"])
