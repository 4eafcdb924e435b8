import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import humid_amd
from humid_amd.synth import synth_words
dev = torch.device("cuda:0")
n_reads = 30_000_000
w, f = synth_words(n_reads, 1005, 24, mode="genome")
d_w = torch.from_numpy(w.view(np.int64)).to(dev); d_f = torch.from_numpy(f).to(dev)
d_c = torch.zeros(n_reads, dtype=torch.int32, device=dev); d_k = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
dd = humid_amd.Dedup(device=0)
res = {}
for edit in (0, 1):
    dd.set_option("edit_distance", edit)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n_reads, 24, 2, 0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cid = d_c.cpu().numpy().view(np.uint32); keep = d_k.cpu().numpy()
    ok = int(keep.sum()) == s["clusters"] == int(cid.max()) and bool(np.array_equal(cid == 0, f == 1))
    res[edit] = s
    print("%s d=2, 30 M genome-mode reads: %.1f ms; unique %d edges %d clusters %d; properties ok %s" % (
        "edit" if edit else "hamming", 1e3 * dt, s["unique"], s["edges"], s["clusters"], ok), flush=True)
assert res[1]["edges"] >= res[0]["edges"] and res[1]["clusters"] <= res[0]["clusters"]
print("edit finds a superset of the Hamming pairs: ok")
