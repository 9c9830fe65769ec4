// Launch helper shared by the two translation units of the in-kernel generator
// (hmc_gauss_rng.hip: one-wave chains; hmc_gauss_rng_wide.hip: chains of 2 / 4 / 8
// waves) -- split so that the two sets of template instantiations compile in parallel.
#pragma once
#include "hmc_gauss_kernel.hpp"

namespace binf {

template <int TMAX, bool REGULAR, int RNG, int LW = 0, bool UDT = false>
static hipError_t launch_rng_tr(const GaussNArgs &a, bool unit, bool fma, dim3 grid, hipStream_t st)
{
    if (RNG == GAUSS_RNG_DUMP) {
        hmc_gauss_persist_kernel<TMAX, REGULAR, true, false, LW, RNG, UDT><<<grid, 512, 0, st>>>(a);
    } else if (unit) {
        if (fma) hmc_gauss_persist_kernel<TMAX, REGULAR, true, true, LW, RNG, UDT><<<grid, 512, 0, st>>>(a);
        else     hmc_gauss_persist_kernel<TMAX, REGULAR, true, false, LW, RNG, UDT><<<grid, 512, 0, st>>>(a);
    } else {
        if (fma) hmc_gauss_persist_kernel<TMAX, REGULAR, false, true, LW, RNG, UDT><<<grid, 512, 0, st>>>(a);
        else     hmc_gauss_persist_kernel<TMAX, REGULAR, false, false, LW, RNG, UDT><<<grid, 512, 0, st>>>(a);
    }
    return hipGetLastError();
}

// hmc_gauss_rng_wide.hip
hipError_t launch_gauss_rng_wide(const GaussNArgs &a, const GaussPlan &p, int rng, bool unit,
                                 bool fma, hipStream_t st);

}  // namespace binf
