// ShmComm (lasso_amd/host/shm_comm.hpp) against a STALE segment: a run that died between the ranks' attach and rank 0's unlink leaves a segment under the same
// name with the magic set and attached == world.  A rank of the next run that opens it before rank 0 has re-created the name used to pass both start-up barriers on the
// dead segment; it must notice (attach count >= world), let go and join the fresh one.  Rank 1 starts first and finds only the stale segment; rank 0 follows.
// Modes (argv[2]): "full" = the above; "partial" = the dead run got only one rank attached and its creator process is gone (attach count < world: only the
// recorded creator pid tells); "world" = same segment size, another world / slot split in the header; "size" = a segment of another size.
#include <sys/wait.h>
#include "../../lasso_amd/host/shm_comm.hpp"
#include <cstdio>
#include <thread>
#include <vector>
int main(int argc, char** argv) {
  const std::string name = argc > 1 ? argv[1] : "/lasso_test_stale";
  const std::string mode = argc > 2 ? argv[2] : "full";
  const int world = 2; const size_t slot = (size_t)1 << 20; size_t bytes = 4096 + (size_t)world * (2 * 64 + 2 * slot);
  if (mode == "size") bytes *= 2;
  uint32_t dead_pid = 0;
  if (mode == "partial") { pid_t ch = fork(); if (ch == 0) _exit(0); int st = 0; waitpid(ch, &st, 0); dead_pid = (uint32_t)ch; }   // a pid that no longer exists
  shm_unlink(name.c_str());
  {   // the dead run's segment: header as ShmComm lays it out (magic, world, slot_bytes, attached)
    int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { printf("FAIL cannot create the stale segment\n"); return 1; }
    uint8_t* p = (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); close(fd);
    *(uint32_t*)(p + 4) = (uint32_t)(mode == "world" ? 2 * world : world); *(uint64_t*)(p + 8) = mode == "world" ? slot / 2 : slot;
    *(uint32_t*)(p + 16) = (uint32_t)(mode == "partial" ? 1 : world); *(uint32_t*)(p + 24) = mode == "partial" ? dead_pid : (uint32_t)getpid();
    { struct stat ns; *(uint64_t*)(p + 32) = stat("/proc/self/ns/pid", &ns) == 0 ? (uint64_t)ns.st_ino : 0; }   // creator_pidns: this test's own namespace
    *(uint32_t*)(p + 0) = 0x4c53484du;
    munmap(p, bytes);
  }
  int ok[2] = {0, 0}; uint32_t got[2][2] = {{0, 0}, {0, 0}};
  auto run = [&](int rank, int delay_ms) {
    std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
    try {
      lasso::ShmComm c(name, rank, world);
      const uint32_t mine = 100u + (uint32_t)rank;
      if (c.allgather(&mine, got[rank], 4) == 0) ok[rank] = 1;
    } catch (const std::exception& e) { printf("rank %d: %s\n", rank, e.what()); }
  };
  std::thread t1(run, 1, 0), t0(run, 0, 150);
  t1.join(); t0.join();
  shm_unlink(name.c_str());
  if (!ok[0] || !ok[1] || got[0][0] != 100 || got[0][1] != 101 || got[1][0] != 100 || got[1][1] != 101) { printf("FAIL ok=%d,%d got=%u,%u / %u,%u\n", ok[0], ok[1], got[0][0], got[0][1], got[1][0], got[1][1]); return 1; }
  printf("OK\n");
  return 0;
}
