// Explicit instantiations of the gemm_nt_kernel variants that stage operands from the 128-byte-row images (include/cvlm.h, ABI 6);
// launched from gemm.hip (cvlm_gemm).  A file of its own so that the two halves of the instantiation list compile side by side.
#include "gemm_kernel.h"
CVLM_GEMM_IL_KERNELS(template)
