/* nogpu_shim.c -- LD_PRELOAD shim for the worker processes of bench.py's ReaxFF CPU baseline (oracle/reax_md.py).
 *
 * TEST INFRASTRUCTURE ONLY.  Those workers are CPU-only, but torch's autograd engine asks every registered device type for its
 * device count when it first runs backward(), which on ROCm initialises the HSA runtime and opens /dev/kfd -- the GPU box counts
 * such a process as a user of its one GPU (six allowed).  With this shim preloaded the device nodes simply do not exist for the
 * process: hipGetDeviceCount reports no device, torch carries on with the CPU.  Nothing else is intercepted. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <string.h>
#include <sys/types.h>

static int is_gpu_node(const char *p) {
  return p && (strcmp(p, "/dev/kfd") == 0 || strncmp(p, "/dev/dri/", 9) == 0);
}
#define FORWARD_MODE(flags, ap, mode) do { mode = 0; if ((flags) & (O_CREAT | O_TMPFILE)) { va_start(ap, flags); mode = va_arg(ap, mode_t); va_end(ap); } } while (0)

int open(const char *path, int flags, ...) {
  static int (*real)(const char *, int, ...) = 0;
  va_list ap; mode_t mode;
  if (is_gpu_node(path)) { errno = ENOENT; return -1; }
  if (!real) real = (int (*)(const char *, int, ...))dlsym(RTLD_NEXT, "open");
  FORWARD_MODE(flags, ap, mode);
  return real(path, flags, mode);
}
int open64(const char *path, int flags, ...) {
  static int (*real)(const char *, int, ...) = 0;
  va_list ap; mode_t mode;
  if (is_gpu_node(path)) { errno = ENOENT; return -1; }
  if (!real) real = (int (*)(const char *, int, ...))dlsym(RTLD_NEXT, "open64");
  FORWARD_MODE(flags, ap, mode);
  return real(path, flags, mode);
}
int openat(int dirfd, const char *path, int flags, ...) {
  static int (*real)(int, const char *, int, ...) = 0;
  va_list ap; mode_t mode;
  if (is_gpu_node(path)) { errno = ENOENT; return -1; }
  if (!real) real = (int (*)(int, const char *, int, ...))dlsym(RTLD_NEXT, "openat");
  FORWARD_MODE(flags, ap, mode);
  return real(dirfd, path, flags, mode);
}
int openat64(int dirfd, const char *path, int flags, ...) {
  static int (*real)(int, const char *, int, ...) = 0;
  va_list ap; mode_t mode;
  if (is_gpu_node(path)) { errno = ENOENT; return -1; }
  if (!real) real = (int (*)(int, const char *, int, ...))dlsym(RTLD_NEXT, "openat64");
  FORWARD_MODE(flags, ap, mode);
  return real(dirfd, path, flags, mode);
}
