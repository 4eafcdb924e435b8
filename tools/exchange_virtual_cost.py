#!/usr/bin/env python3
"""Per-rank GPU work of the multi-GPU exchange mode as ranks are added, measured on ONE GPU: P thread
ranks (tests/fake_dist.py, collectives are device copies) with 10 M reads each share the device, so
(wall time per pass) / P approximates one rank's device work at world size P -- it must stay flat for
weak scaling (the replicated compact-graph step is the part that grows)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from fake_dist import FakeDist, FakeWorld  # noqa: E402
from humid_amd.sharded import HipStageOps, ShardedDedup  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402


def main():
    n_local = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda:0")
    shards = [synth_words(n_local, 1002 + 7919 * r, 24) for r in range(8)]
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ("exchange", "allgather")
    worlds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (1, 2, 4, 8)
    for mode in modes:
        for P in worlds:
            world = FakeWorld(P)
            times = [0.0] * P
            errs = []

            def rank_main(r):
                try:
                    torch.cuda.set_device(0)
                    ops = HipStageOps(0)
                    sd = ShardedDedup(device=0, word_nt=24, distance=1, ops=ops, dist=FakeDist(world, r), mode=mode)
                    w = torch.from_numpy(shards[r][0].view(np.int64)).to(dev)
                    f = torch.from_numpy(shards[r][1]).to(dev)
                    c = torch.zeros(n_local, dtype=torch.int32, device=dev)
                    k = torch.zeros(n_local, dtype=torch.uint8, device=dev)
                    for _ in range(2):
                        sd.run(w, f, c, k)
                    torch.cuda.synchronize()
                    world.barrier.wait()
                    t0 = time.perf_counter()
                    for _ in range(4):
                        sd.run(w, f, c, k)
                    torch.cuda.synchronize()
                    world.barrier.wait()
                    times[r] = (time.perf_counter() - t0) / 4
                    ops.close()
                except Exception:
                    import traceback
                    errs.append(traceback.format_exc())
                    world.barrier.abort()

            th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                print(errs[0])
                sys.exit(1)
            wall = max(times)
            print("%-9s P=%d: %.2f ms per pass for %d M reads on one GPU = %.2f ms per rank" %
                  (mode, P, 1e3 * wall, P * n_local // 1_000_000, 1e3 * wall / P), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
