// TEST DOUBLE of MATLAB's mex.h -- test infrastructure only, never shipped.
//
// Neither MATLAB nor its headers exist in the build image or on the GPU box, so matlab/rbpf_mex.cpp (the MEX gateway a
// maintainer of the reference compiles with `mex`) would otherwise never be executed.  This header declares the subset of
// the documented MEX / mx C API the gateway uses, with the documented signatures and semantics; tests/mexdouble/
// mexdouble.cpp implements it over a small in-process mxArray, and tests/mexdouble/gateway_driver.cpp drives
// mexFunction() exactly as MATLAB would (tests/test_gpu_mex_gateway.py).  Function handles are std::function objects the
// driver registers; mexCallMATLAB dispatches "feval" and the build's own helper .m functions to them.
#ifndef RBPF_TEST_MEX_H_
#define RBPF_TEST_MEX_H_

#include <stddef.h>
#include <stdint.h>

typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct mxArray_tag mxArray;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum {
  mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS,
  mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS,
  mxINT64_CLASS, mxUINT64_CLASS, mxFUNCTION_CLASS
} mxClassID;

#ifdef __cplusplus
extern "C" {
#endif

mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double v);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity flag);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag);
mxArray* mxCreateString(const char* s);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names);
mxArray* mxCreateLogicalScalar(bool v);
mxArray* mxDuplicateArray(const mxArray* a);
void mxDestroyArray(mxArray* a);

double* mxGetPr(const mxArray* a);
void* mxGetData(const mxArray* a);
double mxGetScalar(const mxArray* a);
mwSize mxGetM(const mxArray* a);
mwSize mxGetN(const mxArray* a);                       /* product of dimensions 2..end, as MATLAB defines it */
mwSize mxGetNumberOfElements(const mxArray* a);
mwSize mxGetNumberOfDimensions(const mxArray* a);
const mwSize* mxGetDimensions(const mxArray* a);
mxClassID mxGetClassID(const mxArray* a);
bool mxIsChar(const mxArray* a);
bool mxIsDouble(const mxArray* a);
bool mxIsInt32(const mxArray* a);
bool mxIsStruct(const mxArray* a);
bool mxIsEmpty(const mxArray* a);
bool mxIsLogical(const mxArray* a);
bool mxIsClass(const mxArray* a, const char* name);    /* "function_handle" */
int mxGetString(const mxArray* a, char* buf, mwSize buflen);
mxArray* mxGetField(const mxArray* a, mwIndex index, const char* name);
void mxSetField(mxArray* a, mwIndex index, const char* name, mxArray* value);
double mxGetNaN(void);

void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);   /* does not return (throws in the double) */
int mexPrintf(const char* fmt, ...);
int mexCallMATLAB(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* name);
/* returns NULL on success, otherwise an MException object (here: a char array with the message) */
mxArray* mexCallMATLABWithTrap(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* name);

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#ifdef __cplusplus
}
#endif
#endif
