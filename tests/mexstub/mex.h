// Minimal stand-in for MATLAB's mex.h / matrix.h, used ONLY to compile and exercise THIS repo's gateway
// (matlab/SimpleCABACMex.cpp) in the test-suite, where no MATLAB exists.  It is test scaffolding for our own shim:
// it is never used to build anything of the reference.
#ifndef ISSCABAC_TEST_MEX_H
#define ISSCABAC_TEST_MEX_H
#include <stddef.h>
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

typedef enum { mxDOUBLE_CLASS, mxUINT8_CLASS, mxUINT32_CLASS, mxCHAR_CLASS, mxUINT16_CLASS } mxClassID;
typedef enum { mxREAL } mxComplexity;

struct mxArray {
  mxClassID cls;
  size_t m, n;
  std::vector<uint8_t> bytes;  // numeric data
  std::string str;             // char data
};

struct MexError : std::runtime_error {
  explicit MexError(const char *m) : std::runtime_error(m) {}
};

[[noreturn]] inline void mexErrMsgTxt(const char *msg) { throw MexError(msg); }
int mexPrintf(const char *fmt, ...);
int mexAtExit(void (*fn)(void));
inline void mexMakeMemoryPersistent(void *) {}
void mexWarnMsgTxt(const char *msg);
void mexLock(void);
void mexUnlock(void);
bool mexIsLocked(void);
// memory MATLAB frees by itself when the MEX function returns or raises (the driver does the same after every call)
void *mxCalloc(size_t n, size_t size);
void *mxMalloc(size_t n);
void mxFree(void *p);

inline size_t mx_elsize(mxClassID c) { return c == mxDOUBLE_CLASS ? 8 : c == mxUINT32_CLASS ? 4 : c == mxUINT16_CLASS ? 2 : 1; }
inline bool mxIsClass(const mxArray *a, const char *name) {
  const std::string n(name);
  return (n == "double" && a->cls == mxDOUBLE_CLASS) || (n == "uint8" && a->cls == mxUINT8_CLASS) ||
         (n == "uint32" && a->cls == mxUINT32_CLASS) || (n == "char" && a->cls == mxCHAR_CLASS) ||
         (n == "uint16" && a->cls == mxUINT16_CLASS);
}
inline bool mxIsDouble(const mxArray *a) { return a->cls == mxDOUBLE_CLASS; }
inline size_t mxGetM(const mxArray *a) { return a->m; }
inline size_t mxGetN(const mxArray *a) { return a->n; }
inline size_t mxGetNumberOfElements(const mxArray *a) { return a->cls == mxCHAR_CLASS ? a->str.size() : a->m * a->n; }
inline size_t mxGetElementSize(const mxArray *a) { return mx_elsize(a->cls); }
inline double *mxGetPr(const mxArray *a) { return (double *)a->bytes.data(); }
inline void *mxGetData(const mxArray *a) { return (void *)a->bytes.data(); }
inline int mxGetString(const mxArray *a, char *buf, size_t cap) {
  if (a->cls != mxCHAR_CLASS || a->str.size() + 1 > cap) return 1;
  for (size_t i = 0; i <= a->str.size(); i++) buf[i] = a->str.c_str()[i];
  return 0;
}
char *mxArrayToString(const mxArray *a);
mxArray *mxCreateNumericMatrix(size_t m, size_t n, mxClassID cls, mxComplexity);
inline mxArray *mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity c) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, c); }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#endif
