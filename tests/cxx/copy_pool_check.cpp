// CPU check of noahmp_amd/csrc/nmp_copy_pool.hpp (tests/test_host.py): copies of several sizes are exact, the pool can be stopped and
// restarted, and a process that exits WITHOUT stopping a pool it never destroys does exit (round 6: a pool destroyed by a static destructor
// with its workers parked on the condition variable hung the process in pthread_cond_destroy -- the staging object is therefore never
// destroyed, noahmp_stage.hip).
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "nmp_copy_pool.hpp"

int main() {
  nmp_host::CopyPool& pool = *new nmp_host::CopyPool;         // as noahmp_stage.hip holds it
  const size_t sizes[] = {1, 4095, (4u << 20) - 1, (4u << 20), (4u << 20) + 4097, 33u << 20};
  for (int round = 0; round < 2; round++) {
    for (size_t n : sizes) {
      std::vector<uint8_t> a(n + 64), b(n + 64, 0xEE);
      for (size_t i = 0; i < a.size(); i++) a[i] = (uint8_t)(i * 2654435761u >> 13);
      pool.copy(b.data() + 32, a.data() + 32, n);
      for (size_t i = 0; i < n; i++) if (b[32 + i] != a[32 + i]) { printf("mismatch at %zu of %zu\n", i, n); return 1; }
      for (size_t i = 0; i < 32; i++) if (b[i] != 0xEE || b[32 + n + i] != 0xEE) { printf("wrote outside %zu\n", n); return 1; }
    }
    if (round == 0) pool.stop();                              // finalize ... and the next copy starts the threads again
  }
  printf("copy pool ok\n");
  return 0;                                                   // workers are parked; nothing destroys the pool: must exit
}
