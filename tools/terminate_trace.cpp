// LD_PRELOAD helper for diagnosing an uncaught C++ exception inside a native library of the process (HIP runtime, libpt_hip.so): std::terminate is called at
// the throw point when no handler exists, so the native backtrace taken here still holds the throwing frames.  Test / diagnosis infrastructure only.
//   g++ -O1 -fPIC -shared -o build/terminate_trace.so tools/terminate_trace.cpp ;  LD_PRELOAD=build/terminate_trace.so python3 ...
#include <execinfo.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cxxabi.h>
#include <exception>
#include <typeinfo>
#include <signal.h>
#include <unistd.h>

static void handler() {
    void* bt[96];
    const int n = backtrace(bt, 96);
    const char msg[] = "\n=== std::terminate: native backtrace of the throwing thread ===\n";
    if (write(2, msg, sizeof msg - 1) < 0) {}
    backtrace_symbols_fd(bt, n, 2);
    if (std::type_info* t = abi::__cxa_current_exception_type()) fprintf(stderr, "exception type: %s\n", t->name());
    FILE* m = fopen("/proc/self/maps", "r");
    if (m) {
        char line[512];
        while (fgets(line, sizeof line, m))
            if (strstr(line, "r-xp") && (strstr(line, "libamdhip64") || strstr(line, "libpt_hip") || strstr(line, "libhsa-runtime"))) fputs(line, stderr);
        fclose(m);
    }
    fflush(stderr);
    abort();
}
static void on_abort(int) {      // glibc's heap checks (and MALLOC_CHECK_) end in abort(): the native backtrace of the thread that noticed
    void* bt[96];
    const int n = backtrace(bt, 96);
    const char msg[] = "\n=== SIGABRT: native backtrace ===\n";
    if (write(2, msg, sizeof msg - 1) < 0) {}
    backtrace_symbols_fd(bt, n, 2);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}
__attribute__((constructor)) static void install() { std::set_terminate(handler); if (getenv("PT_TRACE_ABORT")) { signal(SIGABRT, on_abort); signal(SIGSEGV, on_abort); } }
