// LD_PRELOAD helper for debugging: prints a backtrace when std::terminate is called
// (g++ -shared -fPIC -o tools/terminate_trace.bin tools/terminate_trace.cc).
#include <execinfo.h>
#include <unistd.h>
#include <cstdlib>
#include <exception>
static void on_terminate() {
  void* frames[64];
  int n = backtrace(frames, 64);
  const char msg[] = "=== std::terminate backtrace ===\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  abort();
}
__attribute__((constructor)) static void install() { std::set_terminate(on_terminate); }
