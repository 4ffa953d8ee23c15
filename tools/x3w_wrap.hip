// Profiling aid for gemm_x3w.hip: the kernel compiled with a probe macro (see tools/x3w_strip.py).
#include "x3w/gemm_x3w.hip"
