// Implementation of the mex.h test double (tests only; see mex.h in this directory).
#include "mexdouble.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>

struct mxArray_tag {
  mxClassID cls = mxDOUBLE_CLASS;
  std::vector<mwSize> dims{0, 0};
  std::vector<unsigned char> data;                                  // numeric / char / logical payload
  std::vector<std::string> field_names;                             // struct
  std::vector<mxArray*> fields;                                     // struct: [element][field], owned
  mexdouble::Callable fn;                                           // function handle
  std::string fn_text;
};

namespace {
size_t g_live = 0;
std::map<std::string, mexdouble::Callable>& registry() { static std::map<std::string, mexdouble::Callable> r; return r; }

size_t elem_size(mxClassID c) {
  switch (c) {
    case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
    case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxINT16_CLASS: case mxUINT16_CLASS: case mxCHAR_CLASS: return 2;
    case mxINT8_CLASS: case mxUINT8_CLASS: case mxLOGICAL_CLASS: return 1;
    default: return 0;
  }
}
size_t numel(const mxArray* a) { size_t n = 1; for (mwSize d : a->dims) n *= d; return n; }
mxArray* fresh(mxClassID c, const std::vector<mwSize>& dims) {
  mxArray* a = new mxArray_tag();
  a->cls = c; a->dims = dims;
  while (a->dims.size() < 2) a->dims.push_back(1);
  a->data.assign(numel(a) * elem_size(c), 0);
  ++g_live;
  return a;
}
}  // namespace

namespace mexdouble {
mxArray* make_function_handle(Callable f, const std::string& text) {
  mxArray* a = fresh(mxFUNCTION_CLASS, {1, 1});
  a->fn = std::move(f); a->fn_text = text;
  return a;
}
void register_function(const std::string& name, Callable f) { registry()[name] = std::move(f); }
size_t live_arrays() { return g_live; }
}  // namespace mexdouble

extern "C" {

mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return fresh(mxDOUBLE_CLASS, {m, n}); }
mxArray* mxCreateDoubleScalar(double v) { mxArray* a = fresh(mxDOUBLE_CLASS, {1, 1}); *reinterpret_cast<double*>(a->data.data()) = v; return a; }
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity) { return fresh(cls, std::vector<mwSize>(dims, dims + ndim)); }
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity) { return fresh(cls, {m, n}); }
mxArray* mxCreateString(const char* s) {
  const size_t len = std::strlen(s);
  mxArray* a = fresh(mxCHAR_CLASS, {len ? (mwSize)1 : (mwSize)0, len});
  uint16_t* p = reinterpret_cast<uint16_t*>(a->data.data());
  for (size_t i = 0; i < len; ++i) p[i] = (unsigned char)s[i];
  return a;
}
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names) {
  mxArray* a = fresh(mxSTRUCT_CLASS, {m, n});
  for (int f = 0; f < nfields; ++f) a->field_names.push_back(names[f]);
  a->fields.assign(m * n * (size_t)nfields, nullptr);
  return a;
}
mxArray* mxCreateLogicalScalar(bool v) { mxArray* a = fresh(mxLOGICAL_CLASS, {1, 1}); a->data[0] = v ? 1 : 0; return a; }
mxArray* mxDuplicateArray(const mxArray* s) {
  mxArray* a = new mxArray_tag(*s);
  ++g_live;
  for (auto& f : a->fields) if (f) f = mxDuplicateArray(f);
  return a;
}
void mxDestroyArray(mxArray* a) {
  if (!a) return;
  for (auto* f : a->fields) mxDestroyArray(f);
  --g_live;
  delete a;
}

double* mxGetPr(const mxArray* a) { return (a && a->cls == mxDOUBLE_CLASS) ? reinterpret_cast<double*>(const_cast<unsigned char*>(a->data.data())) : nullptr; }
void* mxGetData(const mxArray* a) { return a ? const_cast<unsigned char*>(a->data.data()) : nullptr; }
double mxGetScalar(const mxArray* a) {
  if (!a || numel(a) == 0) return 0.0;
  switch (a->cls) {
    case mxDOUBLE_CLASS: return *reinterpret_cast<const double*>(a->data.data());
    case mxINT32_CLASS: return *reinterpret_cast<const int32_t*>(a->data.data());
    case mxLOGICAL_CLASS: return a->data[0] ? 1.0 : 0.0;
    case mxSINGLE_CLASS: return *reinterpret_cast<const float*>(a->data.data());
    default: return 0.0;
  }
}
mwSize mxGetM(const mxArray* a) { return a->dims[0]; }
mwSize mxGetN(const mxArray* a) { mwSize n = 1; for (size_t i = 1; i < a->dims.size(); ++i) n *= a->dims[i]; return n; }
mwSize mxGetNumberOfElements(const mxArray* a) { return numel(a); }
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->dims.size(); }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims.data(); }
mxClassID mxGetClassID(const mxArray* a) { return a->cls; }
bool mxIsChar(const mxArray* a) { return a && a->cls == mxCHAR_CLASS; }
bool mxIsDouble(const mxArray* a) { return a && a->cls == mxDOUBLE_CLASS; }
bool mxIsInt32(const mxArray* a) { return a && a->cls == mxINT32_CLASS; }
bool mxIsStruct(const mxArray* a) { return a && a->cls == mxSTRUCT_CLASS; }
bool mxIsLogical(const mxArray* a) { return a && a->cls == mxLOGICAL_CLASS; }
bool mxIsEmpty(const mxArray* a) { return !a || numel(a) == 0; }
bool mxIsClass(const mxArray* a, const char* name) {
  if (!a) return false;
  if (!std::strcmp(name, "function_handle")) return a->cls == mxFUNCTION_CLASS;
  if (!std::strcmp(name, "double")) return a->cls == mxDOUBLE_CLASS;
  if (!std::strcmp(name, "struct")) return a->cls == mxSTRUCT_CLASS;
  return false;
}
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
  if (!a || a->cls != mxCHAR_CLASS || buflen == 0) return 1;
  const size_t len = numel(a);
  const uint16_t* p = reinterpret_cast<const uint16_t*>(a->data.data());
  size_t i = 0;
  for (; i < len && i + 1 < buflen; ++i) buf[i] = (char)p[i];
  buf[i] = 0;
  return len + 1 > buflen ? 1 : 0;
}
mxArray* mxGetField(const mxArray* a, mwIndex index, const char* name) {
  if (!a || a->cls != mxSTRUCT_CLASS) return nullptr;
  for (size_t f = 0; f < a->field_names.size(); ++f)
    if (a->field_names[f] == name) return a->fields[index * a->field_names.size() + f];
  return nullptr;
}
void mxSetField(mxArray* a, mwIndex index, const char* name, mxArray* value) {
  for (size_t f = 0; f < a->field_names.size(); ++f)
    if (a->field_names[f] == name) { mxArray*& slot = a->fields[index * a->field_names.size() + f]; if (slot) mxDestroyArray(slot); slot = value; return; }
  // MATLAB's mxSetField requires an existing field; the double adds it (mxAddField + mxSetField)
  const size_t nf = a->field_names.size(), ne = numel(a);
  std::vector<mxArray*> nfields(ne * (nf + 1), nullptr);
  for (size_t e = 0; e < ne; ++e) for (size_t f = 0; f < nf; ++f) nfields[e * (nf + 1) + f] = a->fields[e * nf + f];
  a->field_names.push_back(name);
  a->fields.swap(nfields);
  a->fields[index * (nf + 1) + nf] = value;
}
double mxGetNaN(void) { return std::nan(""); }

void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
  char buf[2048];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  throw mexdouble::MatlabError(id ? id : "", buf);
}
int mexPrintf(const char* fmt, ...) { va_list ap; va_start(ap, fmt); const int r = vfprintf(stdout, fmt, ap); va_end(ap); return r; }

int mexCallMATLAB(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* name) {
  if (!std::strcmp(name, "feval")) {
    if (nrhs < 1 || !prhs[0] || prhs[0]->cls != mxFUNCTION_CLASS) throw mexdouble::MatlabError("MATLAB:feval:argMustBeStringOrHandle", "feval: first argument must be a function handle");
    prhs[0]->fn(nlhs, plhs, nrhs - 1, prhs + 1);
    return 0;
  }
  auto it = registry().find(name);
  if (it == registry().end()) throw mexdouble::MatlabError("MATLAB:UndefinedFunction", std::string("Undefined function '") + name + "'");
  it->second(nlhs, plhs, nrhs, prhs);
  return 0;
}

mxArray* mexCallMATLABWithTrap(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* name) {
  try {
    mexCallMATLAB(nlhs, plhs, nrhs, prhs, name);
    return nullptr;
  } catch (const std::exception& e) {
    return mxCreateString(e.what());
  }
}

}  // extern "C"
