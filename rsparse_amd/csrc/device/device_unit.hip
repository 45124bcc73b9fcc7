// Single translation unit for all device code + the C ABI (one code object, no relocatable device linking).
#include "tri_engine.hip"
#include "chol_engine.hip"
#include "spmm_kernels.hip"
#include "lu_engine.hip"
#include "../capi.hip"
