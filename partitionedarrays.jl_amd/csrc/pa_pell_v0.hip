// pa_pell_v0.hip -- k_spmv_pell on the fp64 value stream: this translation unit holds its instantiations (pa_pell_launch.h says why).
// Reference loops: spmv_csr! src/sparse_utils.jl:649-669, mul!(y,A,x,alpha,beta) as called at src/p_sparse_matrix.jl:2088.
#include "pa_pell_launch.h"

void pa_pell_launch_v0(PA_PELL_LAUNCH_ARGS) { pell_launch_vm<0>(A, D, epi, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); }
