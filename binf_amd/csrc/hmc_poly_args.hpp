// Argument block shared by the two fused polynomial-transition kernels
// (hmc_poly.hip: one lane per chain; hmc_poly_wave.hip: one wave per chain).
#pragma once
#include "gauss_common.hpp"

namespace binf {

struct PolyHmcArgs {
    const double *q0;          // [C x K]
    const double *p0;          // [C x K]
    const double *u;           // [C]
    double *q_out;             // [C x K]
    uint8_t *accepted;         // [C]
    int64_t *n_accepted;       // [C] or null
    double *e_before;          // [C] or null
    double *e_after;           // [C] or null
    const double *xs;          // [N]
    const double *ys;          // [N]
    const double *tau_chain;   // [C] or null
    double tau;
    const double *prior_means; // [K] or null: Gaussian prior on theta (energy only)
    const double *prior_vars;  // [K]
    const double *lp_pre;      // [C] or null: theta-independent log-prob terms added first
    const double *lp_post;     // [C] or null: ... added last
    double *dt_chain;          // [C] or null
    double timestep;
    double uprate;
    double downrate;
    int64_t C;
    int32_t K;
    int32_t N;
    int32_t nsteps;
    int32_t prior_first;
    int32_t adapt;
};

// hmc_poly_wave.hip
int32_t launch_poly_wave_from(const PolyHmcArgs &a, bool fma, hipStream_t st);

}  // namespace binf
