// host_all_gather through shared memory, for ranks that are processes of one node (include/humid_hip.h:
// humid_shm_open / humid_shm_all_gather / humid_shm_abort / humid_shm_close).  Plain C++, no HIP: a translation
// unit of its own inside libhumid_hip.so (round 3; it was part of humid_hip.hip).
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <new>
#include <string>

#include "../../include/humid_hip.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

extern "C" void humid_set_global_error(const char *text);      // humid_hip.hip: what humid_last_error(NULL) returns

namespace {
int fail(void *, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  humid_set_global_error(buf);
  return code;
}
}  // namespace

struct humid_shm {
  std::string name;
  u32 rank = 0, world = 1;
  u64 slot_bytes = 0, map_bytes = 0, calls = 0;
  u8 *base = nullptr;
  bool owner = false;
  // one 64-byte line per rank: [0] arrival counter, [1] hello (the rank's pid once it has mapped the segment),
  // [2] ack (rank 0 copies hello there: "you are on MY segment"); line `world`: [0] abort flag
  std::atomic<u64> *line(u32 q) const { return (std::atomic<u64> *)(base + 64ull * q); }
  std::atomic<u64> *arrive() const { return (std::atomic<u64> *)base; }                 // [8 * q]
  u8 *slot(u32 bank, u32 q) const { return base + 64ull * (world + 1) + ((u64)bank * world + q) * slot_bytes; }
};

extern "C" {

// Collective: every rank of the group calls it.  Rank 0 creates the segment (after unlinking a stale one of the
// same name); the others attach -- and PROVE that what they mapped is rank 0's segment, not a stale one left
// under the name by a crashed run that they opened before rank 0's unlink: each writes its
// pid into its line and waits for rank 0 to echo it; no echo within 50 ms = the wrong inode: map again.
int humid_shm_open(humid_shm **out, const char *name, uint32_t rank, uint32_t world, uint64_t slot_bytes) {
  if (!out || !name || world == 0 || rank >= world || slot_bytes == 0) return fail(nullptr, HUMID_E_INVALID, "humid_shm_open: bad argument");
  slot_bytes = (slot_bytes + 63) & ~63ull;
  const u64 bytes = 64ull * (world + 1) + 2ull * world * slot_bytes;
  const u64 me = ((u64)getpid() << 20) | (rank + 1);
  const auto t0 = std::chrono::steady_clock::now();
  auto timed_out = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60); };
  void *m = MAP_FAILED;
  if (rank == 0) {
    shm_unlink(name);
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail(nullptr, HUMID_E_COMM, "shm_open(%s): %s", name, strerror(errno));
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return fail(nullptr, HUMID_E_COMM, "ftruncate: %s", strerror(errno)); }
    m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { shm_unlink(name); return fail(nullptr, HUMID_E_COMM, "mmap of %s: %s", name, strerror(errno)); }
    // (a fresh segment is zero: counters start at 0.)  Wait for every other rank's hello and echo it.
    for (u32 q = 1; q < world; q++) {
      std::atomic<u64> *ln = (std::atomic<u64> *)((u8 *)m + 64ull * q);
      u64 h = 0;
      while ((h = ln[1].load(std::memory_order_acquire)) == 0) {
        if (timed_out()) { munmap(m, bytes); shm_unlink(name); return fail(nullptr, HUMID_E_COMM, "rank %u did not attach to %s", q, name); }
        usleep(200);
      }
      ln[2].store(h, std::memory_order_release);
    }
  } else {
    while (true) {
      if (timed_out()) return fail(nullptr, HUMID_E_COMM, "shared segment %s did not appear (or is not rank 0's)", name);
      const int fd = shm_open(name, O_RDWR, 0600);
      if (fd < 0) { usleep(1000); continue; }
      struct stat sb;
      if (fstat(fd, &sb) != 0 || (u64)sb.st_size < bytes) { close(fd); usleep(1000); continue; }    // created AND sized
      m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (m == MAP_FAILED) return fail(nullptr, HUMID_E_COMM, "mmap of %s: %s", name, strerror(errno));
      std::atomic<u64> *ln = (std::atomic<u64> *)((u8 *)m + 64ull * rank);
      ln[1].store(me, std::memory_order_release);
      bool acked = false;
      for (int tries = 0; tries < 250 && !acked; tries++) {       // 50 ms
        acked = ln[2].load(std::memory_order_acquire) == me;
        if (!acked) usleep(200);
      }
      if (acked) break;
      munmap(m, bytes);                                             // a stale segment: rank 0 is on another inode
      m = MAP_FAILED;
    }
  }
  humid_shm *h = new (std::nothrow) humid_shm;
  if (!h) { munmap(m, bytes); return fail(nullptr, HUMID_E_NOMEM, "out of host memory"); }
  h->name = name; h->rank = rank; h->world = world; h->slot_bytes = slot_bytes; h->map_bytes = bytes;
  h->base = (u8 *)m; h->owner = rank == 0;
  *out = h;
  return HUMID_OK;
}

int humid_shm_all_gather(void *shm, const void *mine, uint64_t bytes, void *all) {
  humid_shm *h = (humid_shm *)shm;
  if (!h || !mine || !all || bytes > h->slot_bytes) return -1;
  std::atomic<u64> *abort_flag = h->line(h->world);
  if (abort_flag->load(std::memory_order_acquire)) return -1;
  const u64 seq = ++h->calls;
  const u32 bank = (u32)(seq & 1);
  memcpy(h->slot(bank, h->rank), mine, bytes);
  h->arrive()[8 * h->rank].store(seq, std::memory_order_release);
  const auto t0 = std::chrono::steady_clock::now();
  for (u32 q = 0; q < h->world; q++) {
    u32 spins = 0;
    while (h->arrive()[8 * q].load(std::memory_order_acquire) < seq) {
      if ((++spins & 0xfffu) == 0) {
        if (abort_flag->load(std::memory_order_acquire)) return -1;                      // a rank gave up the group (humid_shm_abort)
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return -1;
        // a gather between ranks that all run completes within the first few thousand spins; later than ~0.3 ms the
        // awaited rank is probably not on a core (more ranks than cores): give ours up instead of spinning it away
        if (spins > (1u << 16)) sched_yield();
      }
    }
    memcpy((u8 *)all + (u64)q * bytes, h->slot(bank, q), bytes);
  }
  // the bank is written again two calls from now; by then every rank has arrived at the call in
  // between, i.e. has finished reading this one
  return 0;
}

// a rank that leaves the group for good (its pass failed outside a gather): every gather of every rank returns -1 from now on
void humid_shm_abort(humid_shm *h) {
  if (h && h->base) h->line(h->world)->store(1, std::memory_order_release);
}

void humid_shm_close(humid_shm *h) {
  if (!h) return;
  if (h->base) munmap(h->base, h->map_bytes);
  if (h->owner) shm_unlink(h->name.c_str());
  delete h;
}

}  // extern "C"
