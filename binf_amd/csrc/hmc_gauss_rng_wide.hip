// In-kernel draws (hmc_gauss_rng.hip) for chains that span 2 / 4 / 8 waves
// (1024 < D <= 8192): the same kernel template, LW = 1..3; the chain's acceptance
// draw is passed from its first wave to the others through LDS.
#include "hmc_gauss_rng_launch.hpp"

namespace binf {

// chains of 2 / 4 / 8 waves: leaves of any length <= 128, so TMAX = 16
template <int RNG, int LW>
static hipError_t launch_rng_wide(const GaussNArgs &a, const GaussPlan &p, bool unit, bool fma,
                                  hipStream_t st)
{
    const int64_t chains_per_block = 8 >> LW;        // 8 waves per workgroup (gauss_wpb)
    const dim3 grid((unsigned)((a.C + chains_per_block - 1) / chains_per_block));
    return (p.regular && p.tneed == 16) ? launch_rng_tr<16, true, RNG, LW>(a, unit, fma, grid, st)
                                        : launch_rng_tr<16, false, RNG, LW>(a, unit, fma, grid, st);
}

hipError_t launch_gauss_rng_wide(const GaussNArgs &a, const GaussPlan &p, int rng, bool unit,
                                 bool fma, hipStream_t st)
{
    if (rng == GAUSS_RNG_DUMP) {
        if (p.LW == 1) return launch_rng_wide<GAUSS_RNG_DUMP, 1>(a, p, unit, fma, st);
        if (p.LW == 2) return launch_rng_wide<GAUSS_RNG_DUMP, 2>(a, p, unit, fma, st);
        return launch_rng_wide<GAUSS_RNG_DUMP, 3>(a, p, unit, fma, st);
    }
    if (p.LW == 1) return launch_rng_wide<GAUSS_RNG_FUSED, 1>(a, p, unit, fma, st);
    if (p.LW == 2) return launch_rng_wide<GAUSS_RNG_FUSED, 2>(a, p, unit, fma, st);
    return launch_rng_wide<GAUSS_RNG_FUSED, 3>(a, p, unit, fma, st);
}

}  // namespace binf
