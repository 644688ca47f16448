// Test driver around matlab/SimpleCABACMex.cpp compiled against the stub mex.h: lets the Python tests build
// argument arrays, call mexFunction with the reference's command strings and read results / error texts back.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "mex.h"

#include <stdlib.h>

#include <set>

static std::string g_printed;
static void (*g_atexit)(void) = nullptr;
static int g_locks = 0;
static std::set<void *> g_auto;  // mxCalloc'ed blocks of the running command
static size_t g_auto_freed = 0;  // blocks the driver (standing in for MATLAB) had to free after a command

void mexWarnMsgTxt(const char *msg) { g_printed += std::string("Warning: ") + msg + "\n"; }
void mexLock(void) { g_locks++; }
void mexUnlock(void) { g_locks--; }
bool mexIsLocked(void) { return g_locks > 0; }
void *mxCalloc(size_t n, size_t size) {
  void *p = calloc(n ? n : 1, size ? size : 1);
  g_auto.insert(p);
  return p;
}
void *mxMalloc(size_t n) { return mxCalloc(n, 1); }
void mxFree(void *p) {
  g_auto.erase(p);
  free(p);
}
static void free_auto() {
  g_auto_freed += g_auto.size();
  for (void *p : g_auto) free(p);
  g_auto.clear();
}

int mexPrintf(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_printed += buf;
  return n;
}
int mexAtExit(void (*fn)(void)) {
  g_atexit = fn;
  return 0;
}
char *mxArrayToString(const mxArray *a) { return a->cls == mxCHAR_CLASS ? strdup(a->str.c_str()) : nullptr; }
mxArray *mxCreateNumericMatrix(size_t m, size_t n, mxClassID cls, mxComplexity) {
  mxArray *a = new mxArray;
  a->cls = cls;
  a->m = m;
  a->n = n;
  a->bytes.assign(m * n * mx_elsize(cls) + 8, 0);
  return a;
}

extern "C" {
void *drv_str(const char *s) {
  mxArray *a = new mxArray;
  a->cls = mxCHAR_CLASS;
  a->m = 1;
  a->n = strlen(s);
  a->str = s;
  return a;
}
void *drv_numeric(int cls, const void *data, size_t m, size_t n) {
  mxArray *a = mxCreateNumericMatrix(m, n, (mxClassID)cls, mxREAL);
  if (data) memcpy(a->bytes.data(), data, m * n * mx_elsize((mxClassID)cls));
  return a;
}
void drv_free(void *a) { delete (mxArray *)a; }
int drv_class(void *a) { return (int)((mxArray *)a)->cls; }
size_t drv_m(void *a) { return ((mxArray *)a)->m; }
size_t drv_n(void *a) { return ((mxArray *)a)->n; }
void *drv_data(void *a) { return ((mxArray *)a)->bytes.data(); }
// returns 0 on success, 1 if the gateway raised a MATLAB error (text in err)
int drv_mex(int nlhs, void **plhs, int nrhs, void **prhs, char *err, int cap) {
  try {
    mexFunction(nlhs, (mxArray **)plhs, nrhs, (const mxArray **)prhs);
  } catch (const MexError &e) {
    snprintf(err, cap, "%s", e.what());
    free_auto();
    return 1;
  }
  free_auto();
  return 0;
}
int drv_locks() { return g_locks; }
void drv_clear_printed() { g_printed.clear(); }
const char *drv_printed() { return g_printed.c_str(); }
void drv_shutdown() {
  if (g_atexit) g_atexit();
  g_atexit = nullptr;
}
}
