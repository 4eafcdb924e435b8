// sharded.hpp -- `humid -g N`: the hot path over N GPUs of one node, driven from the C++ host.
//
// The reference is a single process on one CPU (/root/reference/src/humid.cc:369-409); this is the
// multi-GPU form BASELINE.json's north_star asks for ("the host stays C++ ... the read set shards
// across the GPUs").  The parsed words are cut into N shards in input order, one rank (a host thread
// with its own humid_ctx, stream and device) per shard; every rank makes ONE library call per run,
// humid_dedup_run_exchange (include/humid_hip.h, DESIGN.md section 4a: histogram -> value ranges -> word
// exchange -> counts -> pairs per combination -> compact graph -> result exchange), and this file
// supplies the humid_comm callbacks that move bytes between the ranks.
//
// Bulk data moves between the ranks' device buffers with RCCL (grouped ncclSend / ncclRecv over
// xGMI; librccl is loaded on demand) when every rank has a GPU of its own, and with peer copies
// (hipMemcpyAsync device to device) otherwise -- which is also how several ranks can share one GPU
// (tests/test_cli_gpu.py runs 2 and 3 ranks on the one GPU of the test box).  The few host-side
// numbers (histograms, split sizes) are exchanged through the process's own memory.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/humid_hip.h"

namespace humid_host {

struct ShardedResult {
  humid_summary sum{};
  // -s: counts.dat / neigh.dat / clusters.dat as (key, value), ascending keys (src/humid.cc:301-349)
  std::vector<std::pair<uint64_t, uint64_t>> hist[3];
  std::string error;
  std::string comm;          // "rccl" or "copy": what moved the bulk data
  double ms_init = 0, ms_run = 0;
};

// One multi-rank run.  The constructor returns at once: a starter thread brings up the HIP runtime,
// picks the transport and starts one thread per rank (rank r on device r % visible devices), each of
// which creates its context, stream and communicator and then waits -- so all of that (RCCL's
// communicator set-up alone takes a second on 8 GPUs) overlaps the host's first pass over the FastQ
// files.  run() hands the parsed words over and returns when every rank is done.
class ShardedSession {
 public:
  explicit ShardedSession(unsigned n_ranks);          // 1 .. 16
  ~ShardedSession();                                  // ranks that never got a job leave quietly
  ShardedSession(const ShardedSession &) = delete;
  ShardedSession &operator=(const ShardedSession &) = delete;
  // words[n_reads] packed (two uint64 per read for word_nt > 32), filtered[n_reads]; cluster_id / keep: host outputs.
  // edit: Levenshtein instead of Hamming neighbours (-e; distances 2..5 all-gather the unique words inside the pass).
  // want_hist: fill ShardedResult::hist.  Returns HUMID_OK or a HUMID_E_* code with
  // ShardedResult::error set.  Once per session.
  int run(const uint64_t *words, const uint8_t *filtered, uint64_t n_reads, uint32_t word_nt, uint32_t distance,
          uint32_t method, bool want_hist, uint32_t *cluster_id, uint8_t *keep, ShardedResult &out, bool edit = false);

 private:
  struct Impl;
  Impl *p_;
};

}  // namespace humid_host
