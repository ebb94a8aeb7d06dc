// nb_force_sym_bins.hip -- the pair-symmetric grid-mode force kernels instantiated with BINS = true.
//
// Same templates, same control flow as the production instantiations in nb_force_sym.hip (nb_force_sym_kernel.h:
// table-free estimate, one ballot per wave, threshold fallback, uniform / general-mass pair of launches, packed
// and scalar sweeps, every tiling); the only addition is the integer read-out of BinDbg.  nb_quant_bin_sums()
// (nb_api.cpp) runs them on the handle's current positions with the tables of a fresh evaluation, so the claim
// "bit-identical quant-bin assignments" (reference quantization.py:106-123) is checked on the pair loop that
// assigns the bins, not on a separate table walk.  A separate translation unit so that the production objects
// are byte-for-byte what they were and the build parallelises.
#include "nb_force_sym_kernel.h"

hipError_t nb_launch_force_sym_f32_bins(const float *packed, const SymWork *work, int nwork, double *rowslab,
                                        float *colslab, int np, int dim, int r, int uniform, float eps2,
                                        const GridTables *tab, float G, float mass_value, int levels,
                                        unsigned long long *bin_out, int bin_n, hipStream_t st)
{
    const NbKernelEvents ev{};
#define NB_SYMB(DD, RR)                                                                                                  \
    return launch_sym_u<float, DD, RR, HOOK_GRID, true>(packed, work, nwork, rowslab, colslab, np, uniform, eps2, tab, G, st, \
                                                        ev, mass_value, levels, bin_out, bin_n)
    if (dim == 2 && r == 2) { NB_SYMB(2, 2); }
    if (dim == 2 && r == 4) { NB_SYMB(2, 4); }
    if (dim == 3 && r == 2) { NB_SYMB(3, 2); }
    if (dim == 3 && r == 4) { NB_SYMB(3, 4); }
#undef NB_SYMB
    return hipErrorInvalidValue;
}
