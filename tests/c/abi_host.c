/* A host with no Python and no C++: includes the public header as C99, dlopens the library and walks
 * the self-describing packed-weight layout.  No GPU needed (no compute entry point is called).
 *   gcc -std=c99 -Wall -Werror -Iinclude tests/c/abi_host.c -ldl -o abi_host && ./abi_host <lib.so> */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "casync_hip.h"

#define SYM(name) (*(void**)(&p_##name) = dlsym(h, #name), p_##name != NULL)

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 1;
  }
  int (*p_casync_abi_version)(void);
  int (*p_casync_packed_count)(void);
  const char* (*p_casync_packed_name)(int);
  int64_t (*p_casync_packed_offset)(int);
  int64_t (*p_casync_packed_size)(int);
  int64_t (*p_casync_packed_total)(void);
  int64_t (*p_casync_workspace_bytes)(int);
  const char* (*p_casync_last_error)(void);
  if (!SYM(casync_abi_version) || !SYM(casync_packed_count) || !SYM(casync_packed_name) ||
      !SYM(casync_packed_offset) || !SYM(casync_packed_size) || !SYM(casync_packed_total) ||
      !SYM(casync_workspace_bytes) || !SYM(casync_last_error)) {
    fprintf(stderr, "missing symbol\n");
    return 1;
  }
  const int n = p_casync_packed_count();
  int64_t used = 0, end = 0;
  int gamma_seen = 0;
  for (int i = 0; i < n; ++i) {
    const char* name = p_casync_packed_name(i);
    const int64_t off = p_casync_packed_offset(i), size = p_casync_packed_size(i);
    if (!name || off < end || size <= 0) return 1;
    if (strstr(name, ".gamma")) ++gamma_seen;
    end = off + size;
    used += size;
  }
  if (end > p_casync_packed_total() || p_casync_workspace_bytes(0) >= 0) return 1;   /* batch 0 is an error */
  printf("abi %d tensors %d floats %lld total %lld gammas %d ws(1) %lld err \"%s\"\n", p_casync_abi_version(), n,
         (long long)used, (long long)p_casync_packed_total(), gamma_seen, (long long)p_casync_workspace_bytes(1),
         p_casync_last_error());
  dlclose(h);
  return 0;
}
