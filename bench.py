#!/usr/bin/env python3
"""bench.py -- reads/sec deduplicated on the BASELINE.json metric workload.

A "step" is one pass of the whole hot path (exact counts -> neighbours -> clusters ->
per-read cluster id + duplicate flag) over one batch of synthetic packed words that is already
resident in HBM when the timed region starts.  One process per GPU; for N > 1 the driver
launches this file under torch.distributed.run and the read set is sharded over the ranks
(humid_amd.sharded).  Rank 0 prints ONE JSON line.

The oracle (oracle/) is used here ONLY for the cpu_baseline leg; it is never the thing timed
as `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL peer buffers); the driver exports it,
# a bare shell may not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_READ = 13         # 8 B word read + 4 B cluster id + 1 B keep written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--word-nt", type=int, default=24)
    ap.add_argument("--distance", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="reads of the same workload the CPU oracle is run on, outside the timed region "
                         "(-1 = the whole workload: the GPU result is then compared with it bit for bit "
                         "-> verified_vs_oracle; 0 = skip).  One rank only.")
    ap.add_argument("--verify", action="store_true", default=None,
                    help="several ranks: after the timed passes gather every shard's results on rank 0 and "
                         "compare them, bit for bit, with one single-GPU pass over the whole read set "
                         "(default: on)")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--shard-mode", default=None, choices=["exchange", "allgather", "both"],
                    help="multi-GPU orchestration to time (default: HUMID_SHARD_MODE or exchange; 'both' "
                         "times the second one as well and reports it under other_mode)")
    ap.add_argument("--e2e-reads", type=int, default=10_000_000,
                    help="one rank only: read PAIRS of the end-to-end leg -- the `humid` command line timed on "
                         "synthetic PE150 FastQ files with the UMI in the header (FastQ in -> _dedup FastQ out), "
                         "reported as t_e2e_s / e2e_reads_per_s next to the core value (0 = skip)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path even with one rank (overhead measurement)")
    ap.add_argument("--traffic-json", default=None,
                    help="optional JSON with PMC-derived HBM bytes per launch of the dominant kernel")
    return ap.parse_args()


def e2e_leg(n_reads, word_nt, distance):
    """T_e2e: wall time of the `humid` command line, FastQ in -> _dedup FastQ out, on
    the metric's own shape (PE150, UMI = 8 in the header of R1), page cache hot, best of two runs.
    A separate leg like cpu_baseline: never part of `value`."""
    import shutil
    import subprocess
    import tempfile
    from humid_amd.synth import fast_fastq
    exe = os.path.join(ROOT, "humid_amd", "humid")
    if not os.path.exists(exe):
        return {"error": "humid_amd/humid is not built"}
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="humid_e2e_", dir=base)
    try:
        r1, r2 = os.path.join(d, "R1.fastq"), os.path.join(d, "R2.fastq")
        t0 = time.perf_counter()
        size = fast_fastq(r1, n_reads, 1002, mate=0) + fast_fastq(r2, n_reads, 1002, mate=1)
        gen_s = time.perf_counter() - t0
        best = None
        for rep in range(2):
            out = os.path.join(d, "out%d" % rep)
            env = dict(os.environ)
            env["HUMID_TIMING"] = "1"
            # timed from a small helper process: forking this one (torch, the read set: several GB of
            # page tables) would add its own ~0.2 s to the child's wall time
            helper = ("import subprocess, sys, time\n"
                      "t0 = time.perf_counter()\n"
                      "p = subprocess.run(sys.argv[1:], stderr=subprocess.PIPE)\n"
                      "dt = time.perf_counter() - t0\n"
                      "sys.stderr.buffer.write(p.stderr)\n"
                      "print(dt)\n"
                      "sys.exit(p.returncode)\n")
            p = subprocess.run([sys.executable, "-c", helper, exe, "-n", str(word_nt), "-m", str(distance), "-d", out,
                                "-l", os.path.join(d, "log.txt"), r1, r2], env=env, stderr=subprocess.PIPE,
                               stdout=subprocess.PIPE)
            try:
                dt = float(p.stdout.decode().strip().splitlines()[-1])
            except (ValueError, IndexError):
                return {"error": "e2e helper failed: %s" % p.stderr.decode()[-300:]}
            if p.returncode != 0:
                return {"error": "humid exited with %d: %s" % (p.returncode, p.stderr.decode()[-300:])}
            phases = {}
            for line in p.stderr.decode().splitlines():
                if line.startswith("[humid]   of which"):
                    phases["device detail"] = line[7:].strip()
                elif line.startswith("[humid]   init thread"):
                    phases["init thread"] = line[7:].strip()
                elif line.startswith("[humid]") and "epoch" not in line:
                    try:
                        name, val = line[7:].rsplit(None, 2)[0].strip(), line.split()[-2]
                        phases[name] = float(val)
                    except (ValueError, IndexError):      # a line of another shape: not a phase time
                        pass
            kept = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
            if best is None or dt < best["t_e2e_s"]:
                best = {"t_e2e_s": round(dt, 4), "e2e_reads_per_s": round(n_reads / dt, 1),
                        "phases_s": phases, "output_bytes": kept}
            shutil.rmtree(out, ignore_errors=True)
        best.update({"shape": "%d read pairs, PE150, UMI=8 in the R1 header, plain FastQ in (%.2f GB) -> "
                              "R1/R2 _dedup FastQ out; -n %d -m %d; files in %s, page cache hot, best of 2"
                              % (n_reads, size / 1e9, word_nt, distance, base or "the temp dir"),
                     "fastq_bytes": size, "generate_s": round(gen_s, 1),
                     "host_cores_available": os.cpu_count()})
        return best
    finally:
        shutil.rmtree(d, ignore_errors=True)


def spawn_ranks(a):
    """`python bench.py --gpus N` from a bare shell (no launcher: WORLD_SIZE/RANK unset): this process
    becomes the launcher.  It has not imported torch and never touches the GPU; it starts N child
    ranks of this same file (fresh interpreters, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, one GPU
    each), relays rank 0's single JSON line and exits non-zero if any rank does."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    line_file = tempfile.NamedTemporaryFile(prefix="humid_bench_rank0_", suffix=".json", delete=False)
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HUMID_BENCH_LAUNCHER="bench.py")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=line_file if r == 0 else sys.stderr))
    rc = 0
    live = set(range(a.gpus))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, c), file=sys.stderr)
                for o in live:                       # exact PIDs of the children started above
                    procs[o].terminate()
        time.sleep(0.05)
    line_file.close()
    text = open(line_file.name).read().strip()
    os.unlink(line_file.name)
    if rc != 0 and not text:
        raise SystemExit(rc)
    try:
        got = json.loads(text.splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit("bench.py: rank 0 printed no JSON line")
    if got.get("n_gpus") != a.gpus:
        raise SystemExit("bench.py: rank 0 reports n_gpus %r, asked for %d: line withheld" % (got.get("n_gpus"), a.gpus))
    print(json.dumps(got), flush=True)
    raise SystemExit(rc)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        spawn_ranks(a)            # never returns
    # The contract is ONE JSON line on stdout.  Libraries print banners there (RCCL: version, host
    # name, library path; per rank), so file descriptor 1 is pointed at stderr for the whole run
    # and the JSON line goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import humid_amd
    from humid_amd.synth import synth_words

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HUMID_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = 0           # rehearsal: all ranks share the one GPU of the box (collectives over gloo)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        # a line whose n_gpus is not what was asked for would be filed under the wrong point of the curve
        raise SystemExit("bench.py: WORLD_SIZE %d != --gpus %d (start it as `python bench.py --gpus N`, or "
                         "under torch.distributed.run with --nproc-per-node N)" % (world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if os.environ.get("HUMID_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d wants GPU %d, %d visible (HUMID_BENCH_BACKEND=gloo rehearses "
                         "several ranks on one GPU)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or a.force_sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("HUMID_BENCH_BACKEND", "nccl")      # "gloo": multi-process rehearsal on ONE GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n_local = a.reads
    seed = 1002                                   # metric config (1000 + config#)
    # weak scaling: ONE shuffled read set of world*n_local reads, sliced by input order (rank r holds
    # reads [r*n_local, (r+1)*n_local)): families span the ranks, as a real sharded FastQ does.
    # Every rank draws the same set from the same seed and keeps its slice.
    t_synth = time.perf_counter()
    if world == 1:
        words, filt = synth_words(n_local, seed, a.word_nt)
    else:
        all_w, all_f = synth_words(n_local * world, seed, a.word_nt)
        words = np.ascontiguousarray(all_w[rank * n_local:(rank + 1) * n_local])
        filt = np.ascontiguousarray(all_f[rank * n_local:(rank + 1) * n_local])
        if rank != 0:
            del all_w, all_f
    synth_s = time.perf_counter() - t_synth
    d_w = torch.from_numpy(words.view(np.int64)).to(dev)
    d_f = torch.from_numpy(filt).to(dev)
    d_cid = torch.zeros(n_local, dtype=torch.int32, device=dev)
    d_keep = torch.zeros(n_local, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    world_sharded = not (world == 1 and not a.force_sharded)
    KS = ("ms_k_insert", "ms_k_pairs", "ms_k_cluster", "ms_k_map", "ms_k_part", "ms_k_unperm",
          "ms_count", "ms_neighbours", "ms_cluster", "ms_map", "ms_total")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, trace=None):
        """W untimed passes, then exactly K passes between barrier + synchronize; max over ranks"""
        for _ in range(a.warmup):
            step()
        if trace is not None:
            trace.clear()
        barrier()
        t0 = time.perf_counter()
        ks = dict.fromkeys(KS, 0.0)
        last = None
        for _ in range(a.steps):
            last = step()
            for k in ks:
                ks[k] += float(last.get(k, 0.0))
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        for k in ks:
            ks[k] /= max(a.steps, 1)
        return dt, ks, last

    DETAIL = ("ms_k_pairs", "ms_k_cluster", "ms_k_map", "ms_k_part", "ms_k_unperm", "ms_count", "ms_neighbours", "ms_cluster",
              "ms_map")

    def detail_times(step, set_option, ks):
        """the per-kernel and per-stage event times of the library are off in the timed passes (an event record
        between two kernels is a marker the second one waits behind: the 17 of them cost 5-8 % of a pass); three
        more passes, untimed, with them on fill the other kernels' and the stages' columns.  The dominant
        kernel's time (ms_k_insert) and ms_total stay the ones measured inside the timed passes."""
        try:
            set_option("kernel_timing", 1)
        except Exception:
            return ks
        acc = dict.fromkeys(DETAIL, 0.0)
        for _ in range(3):
            s = step()
            for k in acc:
                acc[k] += float(s.get(k, 0.0)) / 3.0
        set_option("kernel_timing", 0)
        ks.update(acc)
        return ks

    other = None
    sd = None
    if not world_sharded:
        dd = humid_amd.Dedup(device=local_rank)

        def step():
            return dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_cid.data_ptr(), d_keep.data_ptr(),
                                 n_local, a.word_nt, a.distance, humid_amd.DIRECTIONAL)
        dt, ks, last = timed(step)
        ks = detail_times(step, dd.set_option, ks)
    else:
        from humid_amd.sharded import ShardedDedup
        first = a.shard_mode if a.shard_mode in ("exchange", "allgather") else None

        def run_mode(mode):
            m = ShardedDedup(device=local_rank, word_nt=a.word_nt, distance=a.distance, mode=mode)

            def step():
                s = dict(m.run(d_w, d_f, d_cid, d_keep))
                if m.mode_used == "exchange":          # HIP-event times of this rank's dominant kernels
                    s.update(m.ops.kernel_ms())
                return s
            r = timed(step, getattr(m, "trace", None))
            r = (r[0], detail_times(step, m.ops.set_option, r[1]), r[2])      # (every rank runs the extra passes: they contain collectives)
            if rank == 0 and getattr(m, "trace", None):
                print("shard trace %s (ms per timed pass): %s" %
                      (m.mode_used, {k: round(v / a.steps, 3) for k, v in sorted(m.trace.items())}), file=sys.stderr)
            return m, r
        if a.shard_mode == "both":               # the second orchestration first: the verified results
            m2, (dt2, _, _) = run_mode("allgather")   # below are then those of the main one
            other = {"mode": m2.mode_used, "ms_per_step": round(1e3 * dt2 / a.steps, 4),
                     "value": round(n_local * world * a.steps / dt2, 1)}
            m2.ops.close()
            first = "exchange"
        sd, (dt, ks, last) = run_mode(first)

    # ---- parity of what was just timed (outside the timed region) ----
    verified_gpu1 = None
    verify_s = None
    ranks_seen = None
    if dist is not None:
        # what the process group itself says: its size, and the device every rank computed on
        pr = torch.cuda.get_device_properties(dev)
        mine = {"rank": dist.get_rank(), "device": local_rank, "name": pr.name,
                "pci_bus": getattr(pr, "pci_bus_id", None), "pid": os.getpid()}
        ranks_seen = [None] * dist.get_world_size()
        dist.all_gather_object(ranks_seen, mine)
    if world_sharded and (a.verify is None or a.verify):
        t_verify = time.perf_counter()
        # every shard's results, gathered on rank 0, against ONE single-GPU pass over the whole set
        from humid_amd.sharded import _all_gather_flat
        g_cid = torch.empty(world * n_local, dtype=torch.int32, device=dev)
        g_keep = torch.empty(world * n_local, dtype=torch.int32, device=dev)
        _all_gather_flat(dist, g_cid, d_cid, world)
        _all_gather_flat(dist, g_keep, d_keep.to(torch.int32), world)
        if rank == 0:
            aw, af = (words, filt) if world == 1 else (all_w, all_f)
            dd1 = humid_amd.Dedup(device=local_rank)
            cid1, keep1, s1 = dd1.run(aw, af, word_nt=a.word_nt, distance=a.distance)
            dd1.close()
            verified_gpu1 = bool(np.array_equal(g_cid.cpu().numpy().view(np.uint32), cid1) and
                                 np.array_equal(g_keep.cpu().numpy().astype(np.uint8), keep1) and
                                 all(int(last[k]) == int(s1[k]) for k in ("total", "usable", "unique", "clusters", "edges")))
        del g_cid, g_keep
        verify_s = time.perf_counter() - t_verify
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_reads = n_local * world
    ms_per_step = 1e3 * dt / a.steps
    value = total_reads * a.steps / dt

    # ---- roofline of the dominant kernel (HIP events around its single launch, live) ----
    # candidates: the N-proportional single-launch kernels (13 B/read x N is their unit)
    mode_used = int(last.get("count_mode_used", 0))
    lds = mode_used in (0, 2) and (not world_sharded or getattr(sd, "mode_used", "") == "exchange")
    # (names: the word-ordered count runs on 8-byte records since round 3 -- k_dedup_rec, k_p8_scatter2,
    # k_unperm_bins8; hashed buckets and the fall-backs keep k_dedup_lds / k_pt_scatter<2> / k_unperm_bins)
    rec8 = bool(last.get("records8", False))
    kern = {(("k_dedup_rec" if rec8 else "k_dedup_lds") if lds else "k_hash_insert"): ks["ms_k_insert"],
            (("k_unperm_bins8" if rec8 else "k_unperm_bins") if ks["ms_k_unperm"] > 0 else ("k_read_map_bucket" if lds else "k_read_map")): ks["ms_k_map"]}
    if ks["ms_k_part"] > 0:
        kern["k_p8_scatter2" if rec8 else "k_pt_scatter<2>"] = ks["ms_k_part"]
    if ks["ms_k_unperm"] > 0:
        kern["k_unperm_window"] = ks["ms_k_unperm"]
    dom = max(kern, key=lambda k: kern[k])
    dom_ms = kern[dom]
    achieved = (BYTES_PER_READ * n_local / (dom_ms * 1e-3)) / 1e9 if dom_ms > 0 else 0.0
    traffic = None
    tj = a.traffic_json or os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        tr = json.load(open(tj))
        traffic = tr.get(dom)
        if traffic is None:                       # template instantiations: k_dedup_lds<true>
            traffic = next((v for k, v in tr.items() if k.split("<")[0] == dom.split("<")[0]), None)
    roofline = {"bound": "hbm", "kernel": dom, "kernel_ms": round(dom_ms, 4),
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "traffic_source": "profiles/traffic.json (separate rocprofv3 --pmc passes of this command; 2*FETCH_SIZE+WRITE_SIZE)" if traffic else None,
                "alg_bytes_per_read": BYTES_PER_READ, "reads_per_launch": n_local,
                "other_kernels_ms": {k: round(v, 4) for k, v in kern.items() if k != dom},
                "whole_path": {"achieved": round(BYTES_PER_READ * n_local / (ms_per_step * 1e-3) / 1e9, 2),
                               "frac": round(BYTES_PER_READ * n_local / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}}

    # ---- CPU baseline + parity: the oracle (1 thread) on the SAME reads, outside the timed region ----
    cpu = None
    verified_oracle = None
    if a.cpu_sample != 0 and world == 1:
        from oracle import pyoracle as orc
        ns = n_local if a.cpu_sample < 0 else min(a.cpu_sample, n_local)
        t1 = time.perf_counter()
        ocid, okeep, osum, phases = orc.dedup_run(words[:ns], filt[:ns], a.word_nt, a.distance, 0)
        cdt = time.perf_counter() - t1
        cpu = {"value": round(ns / cdt, 1), "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "%s %d reads of the same synthetic workload, oracle/humid_oracle.c "
                         "(trie restatement), 1 thread, %.1f s" % ("all" if ns == n_local else "first", ns, cdt),
               "host_cores_available": os.cpu_count(),
               "phase_seconds": {"read+count": round(phases[0], 3), "neighbours": round(phases[1], 3),
                                 "clusters": round(phases[2], 3), "map": round(phases[3], 3)}}
        if ns == n_local:
            # the results of the LAST timed pass, bit for bit against the oracle's
            cid = d_cid.cpu().numpy().view(np.uint32)
            keep = d_keep.cpu().numpy()
            verified_oracle = bool(np.array_equal(cid, ocid) and np.array_equal(keep, okeep) and
                                   all(int(last[k]) == int(osum[k]) for k in ("total", "usable", "unique", "clusters", "edges")))
            if not verified_oracle:
                print("PARITY FAILURE vs oracle: cid mismatches %d, keep mismatches %d, summary gpu %s oracle %s" %
                      (int((cid != ocid).sum()), int((keep != okeep).sum()),
                       {k: int(last[k]) for k in ("usable", "unique", "clusters", "edges")},
                       {k: int(osum[k]) for k in ("usable", "unique", "clusters", "edges")}), file=sys.stderr)

    e2e = None
    if a.e2e_reads > 0 and world == 1 and not a.force_sharded:
        try:
            e2e = e2e_leg(a.e2e_reads, a.word_nt, a.distance)
        except Exception as ex:                      # the leg is extra information: never lose the line
            e2e = {"error": repr(ex)[:300]}

    out = {
        "metric": "reads/sec deduplicated, 10M PE150 UMI=8 d=1",
        "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%d reads/GPU x %d GPU (one shuffled read set of %d reads, sliced by input "
                               "order), PE150 UMI=8 in header -> %d-nt packed words "
                               "(8+8+8), d=%d, directional clustering; synthetic molecules=N/4, "
                               "family 1+Geom(3), p_sub=1e-3, p_N=1e-4"
                               % (n_local, world, total_reads, a.word_nt, a.distance),
                   "reads_per_gpu": n_local, "word_nt": a.word_nt, "distance": a.distance,
                   "method": "directional",
                   "sharding": "single GPU" if not world_sharded else (
                       "reads sharded by input order; RCCL all-to-all of the packed words to value-range "
                       "owners, keyed exchange of unique words, all-gather of the neighbour pairs"
                       + ("; split sizes and counts through shared memory between the ranks' processes"
                          if getattr(sd, "shm_used", False) else "")
                       if sd.mode_used == "exchange" else
                       "reads sharded by input order; RCCL all-gather of the packed words")},
        "summary": {k: int(last[k]) for k in ("total", "usable", "unique", "clusters", "edges") if k in last},
        "device_ms": {k: round(v, 4) for k, v in ks.items()},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if e2e is not None:
        out["e2e"] = e2e
        if "t_e2e_s" in e2e:
            out["t_e2e_s"] = e2e["t_e2e_s"]
            out["e2e_reads_per_s"] = e2e["e2e_reads_per_s"]
    if world_sharded:
        out["shard_mode"] = sd.mode_used
    if dist is not None:
        out["world_size"] = dist.get_world_size()          # as the process group (RCCL) saw it
        out["backend"] = dist.get_backend()
        out["ranks"] = ranks_seen
        out["launcher"] = os.environ.get("HUMID_BENCH_LAUNCHER", "external (WORLD_SIZE was set)")
    else:                                                  # one process, no process group: the same fields, for one rank
        pr = torch.cuda.get_device_properties(dev)
        out["world_size"] = 1
        out["backend"] = None
        out["ranks"] = [{"rank": 0, "device": local_rank, "name": pr.name, "pci_bus": getattr(pr, "pci_bus_id", None),
                         "pid": os.getpid()}]
        out["launcher"] = os.environ.get("HUMID_BENCH_LAUNCHER", "none (one process, no process group)")
    # outside the timed region: every rank draws the whole world x reads set and keeps its slice;
    # rank 0 re-runs the whole set on ONE GPU for verified_vs_single_gpu
    out["synth_s"] = round(synth_s, 1)
    if verify_s is not None:
        out["verify_s"] = round(verify_s, 1)
    if other is not None:
        out["other_mode"] = other
    if verified_oracle is not None:
        out["verified_vs_oracle"] = verified_oracle
    if verified_gpu1 is not None:
        out["verified_vs_single_gpu"] = verified_gpu1
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    if verified_oracle is False or verified_gpu1 is False:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
