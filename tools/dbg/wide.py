import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, humid_amd
from humid_amd.synth import synth_wide_words
os.environ["HUMID_TRACE_COUNT"] = "1"
n = 10_000_000
w, f = synth_wide_words(n, 1007, 48)
dev = torch.device("cuda:0")
d_w = torch.from_numpy(w.view(np.int64)).to(dev); d_f = torch.from_numpy(f).to(dev)
d_c = torch.zeros(n, dtype=torch.int32, device=dev); d_k = torch.zeros(n, dtype=torch.uint8, device=dev)
dd = humid_amd.Dedup(device=0)
for i in range(6):
    s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n, 48, 1, 0)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.items() if k in ("unique", "records8", "count_mode_used", "ms_total", "ms_count", "ms_k_insert", "ms_k_part")})
