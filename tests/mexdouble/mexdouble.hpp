// Test-side extras of the mex.h double: registering "MATLAB functions" and function-handle objects (tests only).
#pragma once
#include "mex.h"

#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace mexdouble {

// A MATLAB-callable: nlhs, plhs, nrhs, prhs (prhs excludes the handle itself for "feval").  Throws to signal a MATLAB error.
using Callable = std::function<void(int, mxArray**, int, mxArray**)>;

// what mexErrMsgIdAndTxt raises in the double
struct MatlabError : std::runtime_error {
  std::string id;
  MatlabError(const std::string& i, const std::string& m) : std::runtime_error(m), id(i) {}
};

mxArray* make_function_handle(Callable f, const std::string& text);     // a function_handle mxArray
void register_function(const std::string& name, Callable f);             // a named function on the "MATLAB path"
size_t live_arrays();                                                    // leak check

}  // namespace mexdouble
