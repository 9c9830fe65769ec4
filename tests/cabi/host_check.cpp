// TEST INFRASTRUCTURE: a host program that binds the drop-in boundary the way a
// foreign-function interface does -- dlopen + dlsym on libbinf_hip.so, plain
// hipMalloc'ed pointers, its own hipStream_t -- with no Python and no torch in the
// process, and checks the results bit for bit against the C restatement of the
// reference path (oracle/liboracle_c.so, also dlopen'ed: the checker).
//
//   host_check --symbols <libbinf_hip.so>                 resolve only (no GPU call)
//   host_check <libbinf_hip.so> <liboracle_c.so>          the checks below on device 0
//
// What it checks (reference: binf/samplers/hmc.py:92-164, binf/pdf/__init__.py:181-191):
//   1 one fused transition, EXACT and FMA arithmetic, on a stream of the program's own
//   2 the same with per-chain step sizes and adaption (hmc.py:183-191)
//   3 n transitions in one launch == n oracle transitions, recorded states included
//   4 the per-step tier (gradient, kick, drift, energies, accept) composed by the CALLER
//     == the fused transition == the oracle
//   5 the same call captured into a hipGraph and replayed (include/binf_hip.h: "safe
//     inside hipGraph stream capture")
//   6 argument errors come back as codes with a text, nothing is thrown or printed
//   8 the polynomial model (binf/example/likelihood.py:24-26, 54-57): forward bit for bit,
//     log-prob bit for bit where log(precision) is exact, to an ulp of N/2 log(precision)
//     otherwise (the device library's log), rows longer than numpy's 8192-element buffer too
//   7 four host threads, each with a stream of its own, call concurrently (the header's
//     "stateless and re-entrant"): every thread's chain of transitions == the oracle's,
//     and binf_last_error is per thread
// Build: hipcc -O2 tests/cabi/host_check.cpp -o tests/cabi/host_check -ldl -lpthread
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include "../../include/binf_hip.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static int n_checks = 0, n_failed = 0;
static void check(bool ok, const char *what)
{
    ++n_checks;
    if (!ok) { ++n_failed; printf("FAIL %s\n", what); }
}

template <class F> static F sym(void *h, const char *name)
{
    void *p = dlsym(h, name);
    if (!p) { printf("FAIL dlsym %s: %s\n", name, dlerror()); exit(2); }
    return reinterpret_cast<F>(p);
}

// the entry points this program binds, typed from include/binf_hip.h
static decltype(&binf_abi_version) abi_version;
static decltype(&binf_last_error) last_error;
static decltype(&binf_hmc_sample_gauss_f64) sample_gauss;
static decltype(&binf_hmc_sample_n_gauss_f64) sample_n_gauss;
static decltype(&binf_row_sum_f64) row_sum;
static decltype(&binf_hmc_energy_f64) hmc_energy;
static decltype(&binf_leapfrog_kick_f64) kick;
static decltype(&binf_leapfrog_drift_f64) drift;
static decltype(&binf_leapfrog_kick_drift_f64) kick_drift;
static decltype(&binf_gauss_grad_f64) gauss_grad;
static decltype(&binf_clipped_exp_f64) clipped_exp;
static decltype(&binf_accept_select_f64) accept_select;
static decltype(&binf_poly_forward_f64) poly_forward;
static decltype(&binf_poly_gauss_logp_f64) poly_logp;

typedef int (*oracle_polyval_fn)(const double *, const double *, double *, int64_t, int64_t, int64_t);
typedef int (*oracle_poly_logp_fn)(const double *, const double *, const double *, const double *, double *,
                                   double *, int64_t, int64_t, int64_t);
typedef int (*oracle_fn)(const double *, const double *, const double *, double *, uint8_t *,
                         double *, double *, double *, int64_t, int64_t, int32_t, double, double,
                         int32_t, double, double, int32_t);

static void bind(void *h)
{
    abi_version = sym<decltype(abi_version)>(h, "binf_abi_version");
    last_error = sym<decltype(last_error)>(h, "binf_last_error");
    sample_gauss = sym<decltype(sample_gauss)>(h, "binf_hmc_sample_gauss_f64");
    sample_n_gauss = sym<decltype(sample_n_gauss)>(h, "binf_hmc_sample_n_gauss_f64");
    row_sum = sym<decltype(row_sum)>(h, "binf_row_sum_f64");
    hmc_energy = sym<decltype(hmc_energy)>(h, "binf_hmc_energy_f64");
    kick = sym<decltype(kick)>(h, "binf_leapfrog_kick_f64");
    drift = sym<decltype(drift)>(h, "binf_leapfrog_drift_f64");
    kick_drift = sym<decltype(kick_drift)>(h, "binf_leapfrog_kick_drift_f64");
    gauss_grad = sym<decltype(gauss_grad)>(h, "binf_gauss_grad_f64");
    clipped_exp = sym<decltype(clipped_exp)>(h, "binf_clipped_exp_f64");
    accept_select = sym<decltype(accept_select)>(h, "binf_accept_select_f64");
    poly_forward = sym<decltype(poly_forward)>(h, "binf_poly_forward_f64");
    poly_logp = sym<decltype(poly_logp)>(h, "binf_poly_gauss_logp_f64");
}

// splitmix64 -> doubles: the inputs only have to be the same on both sides
static uint64_t rng_state;
static double next_unit()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static void fill(std::vector<double> &v, double lo, double hi)
{
    for (auto &x : v) x = lo + (hi - lo) * next_unit();
}

template <class T> struct Dev {
    T *p = nullptr;
    size_t n = 0;
    explicit Dev(size_t n_) : n(n_) { HIP(hipMalloc(&p, (n ? n : 1) * sizeof(T))); }
    explicit Dev(const std::vector<T> &h) : Dev(h.size()) { put(h); }
    ~Dev() { (void)hipFree(p); }
    void put(const std::vector<T> &h) { HIP(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice)); }
    std::vector<T> get() const
    {
        std::vector<T> h(n);
        HIP(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
        return h;
    }
};
template <class T> static bool same(const std::vector<T> &a, const std::vector<T> &b)
{
    return a.size() == b.size() && memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0;
}

struct Want {
    std::vector<double> q, eb, ea, dt;
    std::vector<uint8_t> acc;
};
static Want oracle_call(oracle_fn fn, const std::vector<double> &q0, const double *p0, const double *u,
                        std::vector<double> dt, int64_t C, int64_t D, int L, double k, double x0, int adapt)
{
    Want w;
    w.q.resize(C * D); w.eb.resize(C); w.ea.resize(C); w.acc.resize(C); w.dt = dt;
    int rc = fn(q0.data(), p0, u, w.q.data(), w.acc.data(), w.eb.data(), w.ea.data(), w.dt.data(), C, D, L,
                k, x0, adapt, 1.05, 0.95, 4);
    if (rc) { printf("FAIL oracle rc=%d\n", rc); exit(2); }
    return w;
}

int main(int argc, char **argv)
{
    if (argc == 3 && !strcmp(argv[1], "--symbols")) {
        void *h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("FAIL dlopen: %s\n", dlerror()); return 2; }
        bind(h);
        printf("OK symbols abi=%d header=%d\n", abi_version(), BINF_ABI_VERSION);
        return abi_version() == BINF_ABI_VERSION ? 0 : 1;
    }
    if (argc != 3) { printf("usage: host_check [--symbols] <libbinf_hip.so> [<liboracle_c.so>]\n"); return 2; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    void *ho = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!h || !ho) { printf("FAIL dlopen: %s\n", dlerror()); return 2; }
    bind(h);
    oracle_fn oracle[2] = {sym<oracle_fn>(ho, "oracle_hmc_sample_gauss"),
                           sym<oracle_fn>(ho, "oracle_hmc_sample_gauss_fma")};
    check(abi_version() == BINF_ABI_VERSION, "abi version == header");
    int ndev = 0;
    HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) { printf("FAIL no device\n"); return 2; }
    HIP(hipSetDevice(0));
    hipStream_t st;
    HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    rng_state = 20240917;

    struct Shape { int64_t C, D; int L; double k, x0, dt; };
    const Shape shapes[] = {{37, 1024, 20, 1.0, 0.0, 0.2}, {5, 300, 7, 2.5, 0.3, 0.11}, {130, 64, 3, 0.7, -1.0, 0.4}};
    for (const Shape &s : shapes) {
        const int64_t C = s.C, D = s.D;
        std::vector<double> q0(C * D), p0(C * D), u(C), dts(C);
        fill(q0, -1.7, 1.7); fill(p0, -1.7, 1.7); fill(u, 0.0, 1.0); fill(dts, 0.5 * s.dt, 1.5 * s.dt);
        Dev<double> dq0(q0), dp0(p0), du(u), dq(C * D), deb(C), dea(C), ddt(C);
        Dev<uint8_t> dacc(C);
        Dev<int64_t> dn(C);
        for (int mode = 0; mode < 2; ++mode) {
            // 1: uniform step size, stream of our own
            HIP(hipMemsetAsync(dn.p, 0, C * 8, st));
            int rc = sample_gauss(dq0.p, dp0.p, du.p, dq.p, dacc.p, dn.p, deb.p, dea.p, s.dt, nullptr, C, D, s.L,
                                  s.k, s.x0, 0, 1.05, 0.95, mode, st);
            HIP(hipStreamSynchronize(st));
            Want w = oracle_call(oracle[mode], q0, p0.data(), u.data(), std::vector<double>(C, s.dt), C, D, s.L,
                                 s.k, s.x0, 0);
            check(rc == 0, "sample_gauss rc");
            check(same(dq.get(), w.q), "1 q_out == oracle");
            check(same(dacc.get(), w.acc), "1 accepted == oracle");
            check(same(deb.get(), w.eb) && same(dea.get(), w.ea), "1 energies == oracle");
            std::vector<int64_t> na = dn.get();
            bool cnt = true;
            for (int64_t c = 0; c < C; ++c) cnt &= na[c] == w.acc[c];
            check(cnt, "1 n_accepted += accepted");
            // 2: per-chain step sizes + adaption
            ddt.put(dts);
            rc = sample_gauss(dq0.p, dp0.p, du.p, dq.p, dacc.p, nullptr, nullptr, nullptr, 0.0, ddt.p, C, D, s.L,
                              s.k, s.x0, 1, 1.05, 0.95, mode, st);
            HIP(hipStreamSynchronize(st));
            Want wa = oracle_call(oracle[mode], q0, p0.data(), u.data(), dts, C, D, s.L, s.k, s.x0, 1);
            check(rc == 0 && same(dq.get(), wa.q) && same(dacc.get(), wa.acc), "2 per-chain dt: state, flags");
            check(same(ddt.get(), wa.dt), "2 adapted step sizes == oracle");
        }
        // 3: n transitions in one launch, every state recorded
        const int n = 6;
        std::vector<double> pn(n * C * D), un(n * C);
        fill(pn, -1.7, 1.7); fill(un, 0.0, 1.0);
        Dev<double> dpn(pn), dun(un), dsamples(n * C * D), debn(n * C), dean(n * C);
        Dev<uint8_t> daccn(n * C);
        int rc = sample_n_gauss(dq0.p, dpn.p, dun.p, dq.p, dsamples.p, daccn.p, nullptr, debn.p, dean.p, s.dt, nullptr,
                                C, D, s.L, n, 1, s.k, s.x0, 0, 1.05, 0.95, BINF_MODE_EXACT, st);
        HIP(hipStreamSynchronize(st));
        std::vector<double> q = q0, rec, ebs, eas;
        std::vector<uint8_t> accs;
        for (int i = 0; i < n; ++i) {
            Want w = oracle_call(oracle[0], q, pn.data() + (size_t)i * C * D, un.data() + (size_t)i * C,
                                 std::vector<double>(C, s.dt), C, D, s.L, s.k, s.x0, 0);
            q = w.q;
            rec.insert(rec.end(), w.q.begin(), w.q.end());
            ebs.insert(ebs.end(), w.eb.begin(), w.eb.end());
            eas.insert(eas.end(), w.ea.begin(), w.ea.end());
            accs.insert(accs.end(), w.acc.begin(), w.acc.end());
        }
        check(rc == 0 && same(dq.get(), q), "3 sample_n final state == n oracle transitions");
        check(same(dsamples.get(), rec), "3 recorded states");
        check(same(daccn.get(), accs) && same(debn.get(), ebs) && same(dean.get(), eas), "3 flags, energies");

        // 4: the per-step tier composed by the caller (hmc.py:92-125, 136-164 statement by statement)
        for (int mode = 0; mode < 2; ++mode) {
            Dev<double> tq(q0), tp(p0), tg(C * D), lp(C), e0(C), e1(C), de(C), ex(C), out(C * D);
            Dev<uint8_t> acc(C);
            auto logp = [&](Dev<double> &x, Dev<double> &o) {     // -0.5*k*np.sum((x-x0)**2)
                return row_sum(x.p, o.p, C, D, BINF_ROW_SUMSQ_SHIFT, s.x0, -0.5 * s.k, st);
            };
            int r = logp(tq, lp);
            r |= hmc_energy(tp.p, lp.p, e0.p, C, D, st);
            r |= gauss_grad(tq.p, tg.p, s.k, s.x0, C, D, st);
            r |= kick(tp.p, tg.p, s.dt, nullptr, 1, C, D, mode, st);
            r |= drift(tq.p, tp.p, s.dt, nullptr, C, D, mode, st);
            for (int i = 0; i < s.L - 1; ++i) {
                r |= gauss_grad(tq.p, tg.p, s.k, s.x0, C, D, st);
                r |= kick_drift(tq.p, tp.p, tg.p, s.dt, nullptr, C, D, mode, st);
            }
            r |= gauss_grad(tq.p, tg.p, s.k, s.x0, C, D, st);
            r |= kick(tp.p, tg.p, s.dt, nullptr, 1, C, D, mode, st);
            r |= logp(tq, lp);
            r |= hmc_energy(tp.p, lp.p, e1.p, C, D, st);
            HIP(hipStreamSynchronize(st));
            Want w = oracle_call(oracle[mode], q0, p0.data(), u.data(), std::vector<double>(C, s.dt), C, D, s.L,
                                 s.k, s.x0, 0);
            check(r == 0, "4 per-step tier rc");
            check(same(e0.get(), w.eb) && same(e1.get(), w.ea), "4 per-step energies == oracle");
            r = accept_select(tq.p, dq0.p, e0.p, e1.p, du.p, out.p, acc.p, nullptr, nullptr, 0, 1.05, 0.95, C, D, st);
            HIP(hipStreamSynchronize(st));
            check(r == 0 && same(out.get(), w.q) && same(acc.get(), w.acc), "4 per-step state, flags == oracle");
        }

        // 5: captured into a graph and replayed twice (the second replay continues from the first)
        {
            Dev<double> ga(q0), gb(C * D);
            hipGraph_t g;
            hipGraphExec_t ge;
            HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int r = sample_gauss(ga.p, dp0.p, du.p, gb.p, dacc.p, nullptr, nullptr, nullptr, s.dt, nullptr, C, D, s.L,
                                 s.k, s.x0, 0, 1.05, 0.95, BINF_MODE_EXACT, st);
            r |= sample_gauss(gb.p, dp0.p, du.p, ga.p, dacc.p, nullptr, nullptr, nullptr, s.dt, nullptr, C, D, s.L,
                              s.k, s.x0, 0, 1.05, 0.95, BINF_MODE_EXACT, st);
            HIP(hipStreamEndCapture(st, &g));
            HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            HIP(hipGraphLaunch(ge, st));
            HIP(hipGraphLaunch(ge, st));
            HIP(hipStreamSynchronize(st));
            std::vector<double> qq = q0;
            for (int i = 0; i < 4; ++i)
                qq = oracle_call(oracle[0], qq, p0.data(), u.data(), std::vector<double>(C, s.dt), C, D, s.L, s.k,
                                 s.x0, 0).q;
            check(r == 0 && same(ga.get(), qq), "5 two replays of a captured pair == four oracle transitions");
            HIP(hipGraphExecDestroy(ge));
            HIP(hipGraphDestroy(g));
        }
    }

    // 6: errors are codes + text
    {
        Dev<double> a(64), b(64);
        Dev<uint8_t> f(1);
        char msg[256] = "";
        int rc = sample_gauss(nullptr, a.p, a.p, b.p, f.p, nullptr, nullptr, nullptr, 0.1, nullptr, 1, 64, 3, 1.0,
                              0.0, 0, 1.05, 0.95, 0, st);
        check(rc == BINF_E_ARG && last_error(msg, sizeof msg) > 0 && msg[0], "6 NULL q0 -> BINF_E_ARG with a text");
        rc = sample_gauss(a.p, b.p, b.p, a.p + 8, f.p, nullptr, nullptr, nullptr, 0.1, nullptr, 1, 32, 3, 1.0, 0.0,
                          0, 1.05, 0.95, 0, st);
        check(rc == BINF_E_ALIAS, "6 partially overlapping q_out -> BINF_E_ALIAS");
        rc = sample_gauss(a.p, b.p, b.p, a.p, f.p, nullptr, nullptr, nullptr, 0.1, nullptr, 1, 64, 3, 1.0, 0.0, 0,
                          1.05, 0.95, 7, st);
        check(rc == BINF_E_ARG, "6 unknown mode -> BINF_E_ARG");
        rc = sample_gauss(a.p, b.p, b.p, a.p, f.p, nullptr, nullptr, nullptr, 0.1, nullptr, 0, 64, 3, 1.0, 0.0, 0,
                          1.05, 0.95, 0, st);
        check(rc == 0, "6 zero chains is a no-op");
        HIP(hipStreamSynchronize(st));
    }
    // 8: the polynomial model
    {
        oracle_polyval_fn o_polyval = sym<oracle_polyval_fn>(ho, "oracle_polyval");
        oracle_poly_logp_fn o_logp = sym<oracle_poly_logp_fn>(ho, "oracle_poly_gauss_logp");
        struct PShape { int64_t C, K, N; };
        const PShape ps[] = {{7, 4, 20}, {3, 33, 700}, {2, 9, 16389}, {5, 1, 64}, {300, 17, 129}};
        for (const PShape &s : ps) {
            std::vector<double> co(s.C * s.K), xs(s.N), ys(s.N), tau(s.C);
            fill(co, -1.5, 1.5); fill(xs, -1.0, 1.0); fill(ys, -2.0, 2.0); fill(tau, 0.3, 6.0);
            Dev<double> dco(co), dxs(xs), dys(ys), dtau(tau), dmock(s.C * s.N), dlp(s.C);
            std::vector<double> want(s.C * s.N);
            o_polyval(xs.data(), co.data(), want.data(), s.C, s.K, s.N);
            int rc = poly_forward(dco.p, dxs.p, dmock.p, s.C, s.K, s.N, st);
            HIP(hipStreamSynchronize(st));
            check(rc == 0 && same(dmock.get(), want), "8 polynomial forward == oracle");
            for (double p : {1.0, 2.0, 0.25}) {
                std::vector<double> pr(s.C, p), lp(s.C);
                o_logp(co.data(), xs.data(), ys.data(), pr.data(), lp.data(), nullptr, s.C, s.K, s.N);
                rc = poly_logp(dco.p, dxs.p, dys.p, p, nullptr, dlp.p, s.C, s.K, s.N, st);
                HIP(hipStreamSynchronize(st));
                check(rc == 0 && same(dlp.get(), lp), "8 log-prob == oracle (log(precision) exact)");
            }
            std::vector<double> lp(s.C);
            o_logp(co.data(), xs.data(), ys.data(), tau.data(), lp.data(), nullptr, s.C, s.K, s.N);
            rc = poly_logp(dco.p, dxs.p, dys.p, 0.0, dtau.p, dlp.p, s.C, s.K, s.N, st);
            HIP(hipStreamSynchronize(st));
            std::vector<double> got = dlp.get();
            bool ok = rc == 0;
            for (int64_t c = 0; c < s.C; ++c) {
                const double logz = fabs((double)s.N * 0.5 * log(tau[c]));
                const double tol = 4e-16 * (logz > fabs(lp[c]) ? logz : fabs(lp[c]));
                ok &= fabs(got[c] - lp[c]) <= tol;
            }
            check(ok, "8 log-prob, one precision per chain: within an ulp of N/2 log(precision)");
        }
    }
    // 7: concurrent callers
    {
        const int T = 4, ROUNDS = 8;
        std::atomic<int> bad(0), err_mixed(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                const int64_t C = 20 + 17 * t, D = (t & 1) ? 1024 : 520;
                const int L = 5 + t;
                uint64_t z = 77 + t;
                auto unit = [&z] {
                    z = z * 6364136223846793005ull + 1442695040888963407ull;
                    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
                };
                std::vector<double> q(C * D), p(C * D), u(C);
                for (auto &x : q) x = 3.0 * unit() - 1.5;
                for (auto &x : p) x = 3.0 * unit() - 1.5;
                for (auto &x : u) x = unit();
                hipStream_t s2;
                HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
                Dev<double> da(q), db(C * D), dp(p), du(u);
                Dev<uint8_t> df(C);
                int rc = 0;
                for (int i = 0; i < ROUNDS; ++i)
                    rc |= sample_gauss(i & 1 ? db.p : da.p, dp.p, du.p, i & 1 ? da.p : db.p, df.p, nullptr, nullptr,
                                       nullptr, 0.13, nullptr, C, D, L, 1.5, 0.25, 0, 1.05, 0.95, t & 1, s2);
                // an error of THIS thread's own, raised while the others are computing
                char msg[128] = "";
                int e = sample_gauss(da.p, dp.p, du.p, db.p, df.p, nullptr, nullptr, nullptr, 0.13, nullptr, C, D, L,
                                     1.5, 0.25, 0, 1.05, 0.95, 40 + t, s2);
                last_error(msg, sizeof msg);
                char want[32];
                snprintf(want, sizeof want, "mode %d", 40 + t);
                if (e != BINF_E_ARG || !strstr(msg, want)) ++err_mixed;
                HIP(hipStreamSynchronize(s2));
                std::vector<double> qq = q;
                for (int i = 0; i < ROUNDS; ++i)
                    qq = oracle_call(oracle[t & 1], qq, p.data(), u.data(), std::vector<double>(C, 0.13), C, D, L, 1.5,
                                     0.25, 0).q;
                if (rc || !same(da.get(), qq)) ++bad;
                HIP(hipStreamDestroy(s2));
            });
        for (auto &x : th) x.join();
        check(bad.load() == 0, "7 concurrent callers == oracle");
        check(err_mixed.load() == 0, "7 binf_last_error is the calling thread's");
    }
    HIP(hipStreamDestroy(st));
    if (n_failed) { printf("FAILED %d of %d checks\n", n_failed, n_checks); return 1; }
    printf("OK %d checks (libbinf_hip.so ABI %d, no Python in the process)\n", n_checks, abi_version());
    return 0;
}
