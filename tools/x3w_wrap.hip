// Profiling aid for gemm_x3w.hip: the kernel compiled with a probe macro (see tools/x3w_strip.py).
#include "../score_amd/csrc/gemm_x3w.hip"
