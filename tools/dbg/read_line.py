import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("verified_vs_oracle"), d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"].get("other_kernels_ms"))
print(d.get("stage_ms") or d.get("detail"))
