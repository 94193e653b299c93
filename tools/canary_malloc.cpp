// LD_PRELOAD heap checker for diagnosing a host heap overflow in a process that cannot run under ASan / valgrind (the HIP runtime refuses both):
// every block gets a 32-byte header (magic, size, caller) and a 16-byte tail canary; free() / realloc() verify both and, when one is smashed, print the
// block's size and WHO ALLOCATED IT (return address + library), then abort.  Test / diagnosis infrastructure only.
//   g++ -O1 -fPIC -shared -o build/canary_malloc.so tools/canary_malloc.cpp -ldl ;  LD_PRELOAD=build/canary_malloc.so python3 ...
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

namespace {
typedef void* (*malloc_t)(size_t);
typedef void (*free_t)(void*);
typedef void* (*realloc_t)(void*, size_t);
malloc_t real_malloc; free_t real_free; realloc_t real_realloc;
char boot[1 << 16]; size_t boot_used;       // dlsym calls calloc before the real functions are known
bool initialising;

const uint64_t MAGIC = 0xC0FFEE5AFEB10C5ull, TAIL = 0xDEADBEEFCAFEF00Dull;
struct Hdr { uint64_t magic; size_t size; void* caller; void* raw; };      // 32 bytes, directly before the user pointer

void init() {
    if (real_malloc || initialising) return;
    initialising = true;
    real_malloc = (malloc_t)dlsym(RTLD_NEXT, "malloc");
    real_free = (free_t)dlsym(RTLD_NEXT, "free");
    real_realloc = (realloc_t)dlsym(RTLD_NEXT, "realloc");
    initialising = false;
}
void* boot_alloc(size_t n) { n = (n + 15) & ~(size_t)15; if (boot_used + n > sizeof boot) _exit(99); void* p = boot + boot_used; boot_used += n; return p; }
bool from_boot(void* p) { return (char*)p >= boot && (char*)p < boot + sizeof boot; }

void* make(size_t size, size_t align, void* caller) {
    init();
    if (!real_malloc) return boot_alloc(size);
    if (align < 16) align = 16;
    char* raw = (char*)real_malloc(size + align + sizeof(Hdr) + 16);
    if (!raw) return nullptr;
    uintptr_t u = ((uintptr_t)raw + sizeof(Hdr) + align - 1) & ~(uintptr_t)(align - 1);
    Hdr* h = (Hdr*)(u - sizeof(Hdr));
    h->magic = MAGIC; h->size = size; h->caller = caller; h->raw = raw;
    memcpy((char*)u + size, &TAIL, 8); memcpy((char*)u + size + 8, &TAIL, 8);
    return (void*)u;
}
void report(const char* what, Hdr* h, void* user) {
    Dl_info a; const char* lib = (dladdr(h->caller, &a) && a.dli_fname) ? a.dli_fname : "?";
    char msg[512];
    int n = snprintf(msg, sizeof msg, "\n=== canary_malloc: %s of the block %p, %zu bytes, allocated from %p (%s+0x%lx %s)\n", what, user, h->size, h->caller, lib,
                     (unsigned long)((char*)h->caller - (char*)a.dli_fbase), a.dli_sname ? a.dli_sname : "");
    if (write(2, msg, n) < 0) {}
    unsigned char* t = (unsigned char*)user + h->size;
    n = snprintf(msg, sizeof msg, "tail bytes:");
    for (int k = 0; k < 16; k++) n += snprintf(msg + n, sizeof msg - n, " %02x", t[k]);
    n += snprintf(msg + n, sizeof msg - n, "\nbacktrace of the free:\n");
    if (write(2, msg, n) < 0) {}
    void* bt[48]; int m = backtrace(bt, 48); backtrace_symbols_fd(bt, m, 2);
    abort();
}
// freed blocks of up to 64 KB are filled with 0xDD and parked; when a parked block leaves the ring its fill is verified: a write AFTER free names the block it hit
const int PARK = 8192;
void* parked[PARK]; int park_at; pthread_mutex_t park_lock = PTHREAD_MUTEX_INITIALIZER;
void verify_parked(void* user) {
    Hdr* h = (Hdr*)((char*)user - sizeof(Hdr));
    unsigned char* b = (unsigned char*)user;
    for (size_t k = 0; k < h->size; k++)
        if (b[k] != 0xDD) {
            char msg[256]; int n = snprintf(msg, sizeof msg, "\n=== canary_malloc: byte %zu of a FREED block was written (now %02x)", k, b[k]);
            if (write(2, msg, n) < 0) {}
            h->magic = MAGIC; report("WRITE AFTER FREE", h, user);
        }
    uint64_t t[2]; memcpy(t, (char*)user + h->size, 16);
    if (t[0] != TAIL || t[1] != TAIL) { h->magic = MAGIC; report("WRITE PAST THE END (after free)", h, user); }
}
Hdr* check(void* p) {      // null: not one of ours
    Hdr* h = (Hdr*)((char*)p - sizeof(Hdr));
    if (h->magic != MAGIC) return nullptr;
    uint64_t t[2]; memcpy(t, (char*)p + h->size, 16);
    if (t[0] != TAIL || t[1] != TAIL) report("WRITE PAST THE END", h, p);
    return h;
}
}  // namespace

extern "C" {
void* malloc(size_t n) { return make(n, 16, __builtin_return_address(0)); }
void* calloc(size_t a, size_t b) { size_t n = a * b; void* p = make(n, 16, __builtin_return_address(0)); if (p) memset(p, 0, n); return p; }
void free(void* p) {
    if (!p || from_boot(p)) return;
    init();
    Hdr* h = check(p);
    if (!h) { if (real_free) real_free(p); return; }
    h->magic = 0;
    if (h->size <= (64u << 10)) {
        memset(p, 0xDD, h->size);
        pthread_mutex_lock(&park_lock);
        void* old = parked[park_at]; parked[park_at] = p; park_at = (park_at + 1) % PARK;
        pthread_mutex_unlock(&park_lock);
        if (!old) return;
        verify_parked(old);
        p = old; h = (Hdr*)((char*)p - sizeof(Hdr));
    }
    real_free(h->raw);
}
void* realloc(void* p, size_t n) {
    if (!p) return make(n, 16, __builtin_return_address(0));
    if (from_boot(p)) { void* q = make(n, 16, __builtin_return_address(0)); if (q) memcpy(q, p, n); return q; }
    Hdr* h = check(p);
    if (!h) return real_realloc(p, n);
    void* q = make(n, 16, __builtin_return_address(0));
    if (!q) return nullptr;
    memcpy(q, p, h->size < n ? h->size : n);
    free(p);
    return q;
}
int posix_memalign(void** out, size_t align, size_t n) { void* p = make(n, align, __builtin_return_address(0)); if (!p) return 12; *out = p; return 0; }
void* aligned_alloc(size_t align, size_t n) { return make(n, align, __builtin_return_address(0)); }
void* memalign(size_t align, size_t n) { return make(n, align, __builtin_return_address(0)); }
void* valloc(size_t n) { return make(n, 4096, __builtin_return_address(0)); }
size_t malloc_usable_size(void* p) { if (!p) return 0; Hdr* h = (Hdr*)((char*)p - sizeof(Hdr)); return h->magic == MAGIC ? h->size : 0; }
}
