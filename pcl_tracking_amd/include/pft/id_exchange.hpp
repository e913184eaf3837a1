// pft/id_exchange.hpp -- how the ranks of one node hand the RCCL communicator id from rank 0 to the others through a
// file, safely (examples/dist_tracking_amd.cpp; no RCCL or HIP types here, so the logic is tested on the CPU:
// tests/test_id_exchange.py compiles it with g++).
//
// The file is a record, not the bare id:
//   magic, launch nonce, publisher pid + the publisher's start time (clock ticks since boot, /proc/<pid>/stat field 22),
//   publish time, payload length, payload (the ncclUniqueId bytes)
// Rank 0 removes whatever is at the path before it publishes (write under a temporary name, rename: readers never see
// half a record).  A reader accepts a record only if
//   * magic and payload length are right,
//   * the nonce is its own launch's (launch_nonce(): MASTER_ADDR, MASTER_PORT, TORCHELASTIC_RUN_ID, PFT_RUN_NONCE and the
//     launcher -- parent process id and ITS start time: the ranks of one launch on one node share their parent),
//   * the publisher is still alive with the recorded start time (a crashed or killed earlier run leaves a record whose
//     publisher is gone -- or whose pid now belongs to another process with another start time),
// and otherwise keeps polling until the time-out.  A stale file is therefore never taken for this launch's id.
#pragma once
#include <sys/types.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace pft {

struct IdRecord {
  char magic[8];  // "PFTNCCL1"
  uint64_t nonce;
  int64_t publisher_pid;
  uint64_t publisher_start_ticks;
  int64_t publish_unix_ns;
  uint32_t payload_bytes;
  uint32_t reserved;
  unsigned char payload[256];
};

// start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat); 0 if the process does not exist
inline uint64_t process_start_ticks(long pid) {
  char path[64];
  std::snprintf(path, sizeof(path), "/proc/%ld/stat", pid);
  FILE* f = std::fopen(path, "r");
  if (!f) return 0;
  char buf[2048];
  const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
  std::fclose(f);
  buf[n] = 0;
  const char* p = std::strrchr(buf, ')');  // the command name may contain spaces and parentheses: fields follow the LAST ')'
  if (!p) return 0;
  p++;
  int field = 2;  // the field after ')' is number 3 (state)
  unsigned long long v = 0;
  while (*p) {
    while (*p == ' ') p++;
    field++;
    if (field == 22) {
      v = std::strtoull(p, nullptr, 10);
      break;
    }
    while (*p && *p != ' ') p++;
  }
  return (uint64_t)v;
}

inline uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
  const unsigned char* b = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; i++) {
    h ^= b[i];
    h *= 1099511628211ull;
  }
  return h;
}

// the same value in every rank of one launch on one node, different between launches
inline uint64_t launch_nonce() {
  uint64_t h = 1469598103934665603ull;
  for (const char* name : {"MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "PFT_RUN_NONCE"}) {
    const char* v = std::getenv(name);
    h = fnv1a(h, name, std::strlen(name));
    if (v) h = fnv1a(h, v, std::strlen(v));
  }
  const long ppid = (long)::getppid();
  const uint64_t pst = process_start_ticks(ppid);
  h = fnv1a(h, &ppid, sizeof(ppid));
  h = fnv1a(h, &pst, sizeof(pst));
  return h;
}

inline int64_t unix_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// rank 0: remove a stale record, then publish this launch's
inline bool publish_id(const std::string& path, const void* payload, size_t n, uint64_t nonce) {
  if (n > sizeof(IdRecord::payload)) return false;
  ::unlink(path.c_str());
  IdRecord r;
  std::memset(&r, 0, sizeof(r));
  std::memcpy(r.magic, "PFTNCCL1", 8);
  r.nonce = nonce;
  r.publisher_pid = (int64_t)::getpid();
  r.publisher_start_ticks = process_start_ticks((long)::getpid());
  r.publish_unix_ns = unix_ns();
  r.payload_bytes = (uint32_t)n;
  std::memcpy(r.payload, payload, n);
  const std::string tmp = path + ".tmp." + std::to_string((long)::getpid());
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(&r, sizeof(r), 1, f) == 1;
  std::fclose(f);
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
    ::unlink(tmp.c_str());
    return false;
  }
  return true;
}

enum class IdCheck { ok, unreadable, bad_magic, wrong_nonce, publisher_gone };

inline IdCheck check_record(const IdRecord& r, size_t n, uint64_t nonce) {
  if (std::memcmp(r.magic, "PFTNCCL1", 8) != 0 || r.payload_bytes != n) return IdCheck::bad_magic;
  if (r.nonce != nonce) return IdCheck::wrong_nonce;
  const uint64_t st = process_start_ticks((long)r.publisher_pid);
  if (st == 0 || st != r.publisher_start_ticks) return IdCheck::publisher_gone;
  return IdCheck::ok;
}

inline IdCheck read_id_once(const std::string& path, void* payload, size_t n, uint64_t nonce) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return IdCheck::unreadable;
  IdRecord r;
  const bool got = std::fread(&r, sizeof(r), 1, f) == 1;
  std::fclose(f);
  if (!got) return IdCheck::unreadable;
  const IdCheck c = check_record(r, n, nonce);
  if (c == IdCheck::ok) std::memcpy(payload, r.payload, n);
  return c;
}

// ranks != 0: poll until this launch's record is there; `why` receives the reason of the last rejection on a time-out
inline bool await_id(const std::string& path, void* payload, size_t n, uint64_t nonce, int timeout_ms, IdCheck* why = nullptr) {
  IdCheck last = IdCheck::unreadable;
  for (int waited = 0; waited <= timeout_ms; waited += 10) {
    last = read_id_once(path, payload, n, nonce);
    if (last == IdCheck::ok) return true;
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  if (why) *why = last;
  return false;
}

inline const char* id_check_string(IdCheck c) {
  switch (c) {
    case IdCheck::ok: return "ok";
    case IdCheck::unreadable: return "no record at the path";
    case IdCheck::bad_magic: return "not a communicator-id record";
    case IdCheck::wrong_nonce: return "a record of another launch (nonce differs)";
    case IdCheck::publisher_gone: return "a stale record: its publisher no longer runs";
  }
  return "?";
}

}  // namespace pft
