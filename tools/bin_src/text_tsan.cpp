// ThreadSanitizer driver for sai_write_window_rows (not part of the library): two caller threads share the
// process-wide worker pool and the writer's piece buffers; both files must come out the same.
//   g++ -O1 -g -std=c++17 -fsanitize=thread -Iinclude -Isai_amd/csrc tools/bin_src/text_tsan.cpp \
//       sai_amd/csrc/host_core.cpp sai_amd/csrc/text_out.cpp -lpthread -ldl -o /tmp/text_tsan && /tmp/text_tsan
#include <cstdio>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include "saihip.h"
int main() {
  const int n = 9000;
  std::vector<int64_t> win(2 * n), off(n);
  std::vector<int32_t> nsnps(n), cnt(n), pos;
  std::vector<double> q(n);
  for (int w = 0; w < n; ++w) { win[2*w] = w * 500 + 1; win[2*w+1] = w * 500 + 1000; nsnps[w] = w % 7; cnt[w] = w % 3; off[w] = (int64_t)pos.size(); for (int k = 0; k < cnt[w]; ++k) pos.push_back(w * 10 + k); q[w] = w / 9000.0; }
  auto job = [&](int id) {
    char path[64]; std::snprintf(path, sizeof path, "/tmp/tsan_out_%d", id);
    for (int rep = 0; rep < 20; ++rep) {
      int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_APPEND, 0600), fd2 = open("/dev/null", O_WRONLY);
      sai_text_column cols[2] = {{cnt.data(), 4, SAI_TEXT_I32, 0}, {q.data(), 8, SAI_TEXT_F64, 0}};
      sai_log_rows lg{cnt.data(), 4, off.data(), 1, pos.data(), 4, fd2};
      int64_t nb[2];
      if (sai_write_window_rows("7", "a\tb\tc\tNA", n, win.data(), nsnps.data(), 2, cols, fd, 1, &lg, nb)) { std::fprintf(stderr, "%s\n", sai_last_error()); _exit(1); }
      close(fd); close(fd2);
    }
  };
  std::thread a(job, 0), b(job, 1);
  a.join(); b.join();
  std::puts("ok");
}
