// test driver for pft/id_exchange.hpp (tests/test_id_exchange.py): no GPU, no RCCL
//   id_exchange_tool publish PATH NONCE HOLD_MS   -> publishes 128 payload bytes (0, 1, 2, ...), prints "published", stays alive HOLD_MS
//   id_exchange_tool read PATH NONCE TIMEOUT_MS   -> prints "ok <first payload bytes>" or the reason of the rejection
//   id_exchange_tool nonce                         -> prints launch_nonce()
#include <cinttypes>

#include "pft/id_exchange.hpp"

int main(int argc, char** argv) {
  if (argc >= 2 && !std::strcmp(argv[1], "nonce")) {
    std::printf("%" PRIu64 "\n", pft::launch_nonce());
    return 0;
  }
  if (argc < 5) return 2;
  const std::string path = argv[2];
  const uint64_t nonce = std::strtoull(argv[3], nullptr, 10);
  unsigned char payload[128];
  if (!std::strcmp(argv[1], "publish")) {
    for (int i = 0; i < 128; i++) payload[i] = (unsigned char)i;
    if (!pft::publish_id(path, payload, sizeof(payload), nonce)) return 1;
    std::printf("published\n");
    std::fflush(stdout);
    std::this_thread::sleep_for(std::chrono::milliseconds(std::atoi(argv[4])));
    return 0;
  }
  if (!std::strcmp(argv[1], "read")) {
    pft::IdCheck why = pft::IdCheck::ok;
    if (pft::await_id(path, payload, sizeof(payload), nonce, std::atoi(argv[4]), &why)) {
      std::printf("ok %d %d %d\n", payload[0], payload[1], payload[127]);
      return 0;
    }
    std::printf("%s\n", pft::id_check_string(why));
    return 3;
  }
  return 2;
}
