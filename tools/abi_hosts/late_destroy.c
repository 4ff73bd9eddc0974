/* A plain-C host of libhip_nmf.so that frees its handle from an exit handler registered BEFORE the library (and the HIP
 * runtime under it) is loaded -- exit handlers run last-in-first-out, so this one runs AFTER the library's own exit hook and
 * after handlers the HIP runtime registered at load time: the shape of "a host that destroys its objects from static
 * destructors".  hipnmf_destroy must then release the host side only and return 0 (include/hip_nmf.h, hipnmf_destroy).
 *
 *   gcc -O1 -o late_destroy late_destroy.c -ldl && ./late_destroy <path to libhip_nmf.so> [early]
 *
 * "early": also create + destroy a second handle in the ordinary way first (the normal path must still free everything).
 * Prints "LATE-DESTROY-OK rc=0" from the exit handler.  tools/exit_cases.py runs it (case c_host_late_destroy). */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef int (*create_fn)(int, void**);
typedef int (*destroy_fn)(void*);
typedef int (*stream_fn)(void*, long long, int, int, double*);

static destroy_fn g_destroy;
static void* g_handle;

static void late(void) {
  int rc = g_destroy ? g_destroy(g_handle) : -100;
  char buf[64];
  int n = snprintf(buf, sizeof buf, "LATE-DESTROY-%s rc=%d\n", rc == 0 ? "OK" : "FAILED", rc);
  if (write(1, buf, (size_t)n) < 0) _exit(3);
  if (rc != 0) _exit(4);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  atexit(late); /* first registered = last run */
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 2;
  }
  create_fn create = (create_fn)dlsym(lib, "hipnmf_create");
  g_destroy = (destroy_fn)dlsym(lib, "hipnmf_destroy");
  stream_fn stream = (stream_fn)dlsym(lib, "hipnmf_diag_stream_gbs");
  if (!create || !g_destroy || !stream) return 2;
  if (argc > 2 && !strcmp(argv[2], "early")) {
    void* h2 = NULL;
    if (create(0, &h2) != 0 || g_destroy(h2) != 0) return 5;
  }
  if (create(0, &g_handle) != 0) return 6;
  double gbs = 0;
  if (stream(g_handle, 1 << 20, 64, 3, &gbs) != 0) return 7; /* real device work: stream, events, a kernel */
  printf("MARK stream %.0f GB/s\n", gbs);
  fflush(stdout);
  return 0; /* exit(): the library's hook marks the process as exiting, then late() runs */
}
