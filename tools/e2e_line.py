import json,sys
b=json.load(open(sys.argv[1]))
print("value", b["value"], "e2e", b["e2e_steady"]["ms_per_batch"], b["e2e_steady"]["frames_per_s"], "sustained", b["sustained"]["frames_per_s"] if b["sustained"] else None, "single10k", b["regions"]["single_file_10k"]["ms_per_batch"], "parity", b["parity_checked"])
