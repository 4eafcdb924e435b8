#!/usr/bin/env python3
"""Phase times of ONE rank of the multi-GPU exchange pass at world size P, measured on one GPU: P thread
ranks (tests/fake_dist.py) with N reads each TAKE TURNS on the device -- a rank computes only while it holds
a global lock and gives it up while it waits in a collective -- and the library prints the host time of every
phase (HUMID_XTRACE=1: each mark waits for the stream).  So every phase time is that of an otherwise idle GPU,
with the inputs a rank really has at that world size (its share of the interior pairs, ALL crossing pairs, the
flagged components of the others): what owner-local clustering costs as ranks are added.  The collectives
themselves are device copies here and cost nothing like xGMI transfers: only the compute phases mean anything.

usage: exchange_phase_cost.py [reads_per_rank] [world sizes, comma separated]"""
import os
import sys
import threading

os.environ["HUMID_XTRACE"] = "1"

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from fake_dist import FakeDist, FakeWorld  # noqa: E402
from humid_amd.sharded import HipStageOps, ShardedDedup  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402

GPU = threading.Lock()


class TurnDist(FakeDist):
    """FakeDist whose ranks hold the GPU lock except while they wait for each other"""

    def _exchange(self, obj):
        if obj is not None and torch.is_tensor(obj) and obj.is_cuda:
            torch.cuda.synchronize()
        self.w.slots[self.rank] = obj
        GPU.release()
        self.w.barrier.wait()
        allv = list(self.w.slots)
        self.w.barrier.wait()
        GPU.acquire()
        return allv


def main():
    n_local = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)
    dev = torch.device("cuda:0")
    for P in worlds:
        # ONE shuffled read set of P x n_local reads sliced by input order, as bench.py's weak scaling does
        words, filt = synth_words(n_local * P, 1002, 24)
        world = FakeWorld(P)
        errs = []

        def rank_main(r):
            try:
                torch.cuda.set_device(0)
                GPU.acquire()
                ops = HipStageOps(0)
                sd = ShardedDedup(device=0, word_nt=24, distance=1, ops=ops, dist=TurnDist(world, r), mode="exchange")
                w = torch.from_numpy(words[r * n_local:(r + 1) * n_local].view(np.int64)).to(dev)
                f = torch.from_numpy(filt[r * n_local:(r + 1) * n_local]).to(dev)
                c = torch.zeros(n_local, dtype=torch.int32, device=dev)
                k = torch.zeros(n_local, dtype=torch.uint8, device=dev)
                for i in range(3):
                    if r == 0:
                        print("--- P=%d pass %d%s" % (P, i, " (warm-up)" if i < 2 else ""), file=sys.stderr, flush=True)
                    sd.run(w, f, c, k)
                torch.cuda.synchronize()
                ops.close()
                GPU.release()
            except Exception:
                import traceback
                errs.append(traceback.format_exc())
                world.barrier.abort()
                try:
                    GPU.release()
                except RuntimeError:
                    pass

        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            print(errs[0])
            sys.exit(1)
        del words, filt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
