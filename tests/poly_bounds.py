"""First-order error propagation for HMC on the polynomial (linear-Gaussian)
posterior: the tolerance the GPU trajectory is held to, COMPUTED per case
instead of a flat rtol.

Why a bound and not bits.  Everything elementwise on this path is numpy's
arithmetic and is reproduced bit for bit; the one operation that cannot be is
the force contraction ``J . ((mock - ys) * tau)`` (binf/pdf/likelihoods.py:155:
BLAS dgemv in the reference, an MFMA / FMA dot product here), whose summation
order differs.  BASELINE.json's bar for such results is 1e-10 relative; for a
sum the meaningful scale is ``B = sum_n |J_in| |r_n|`` (any summation order is
within (N-1) u of the exact value on that scale).  So the model is:

    every force evaluation of the kernel equals the reference's up to
    |e_i| <= EPS * B_i(theta)            (EPS = 1e-10)

and a transition is L + 1 force evaluations.  The posterior is linear-Gaussian
(force = H theta - b, H = tau J J^T), so the leapfrog map (hmc.py:116-123) is
linear and an error injected at kick j reaches the end state through the exact
product of the remaining step matrices, S_j.  With w_j the kick weights (dt/2,
dt, ..., dt, dt/2):

    |dz_L| <= |P| [dq_0; 0] + sum_j w_j |S_j[:, p]| (EPS B_j + (dtau/tau) |g_j|)

entrywise, where dq_0 / dtau are bounds on differences already present in the
start state / precision (Gibbs sweeps chain them).  Energies follow with
|dE| <= |grad E| . |dz| plus a few ulps for the evaluation itself.  All
quantities are evaluated along the numpy trajectory inside this module."""
import numpy as np

EPS = 1e-10          # BASELINE.json north_star: 1e-10 relative
U = 2.0 ** -53


def design(xs, K):
    return np.vstack([np.asarray(xs, dtype=np.float64) ** i for i in range(K)])


class PolyBound(object):
    """One chain of the polynomial posterior's coefficient conditional."""

    def __init__(self, xs, ys, K, prior_mu=None, prior_var=None):
        self.J = design(xs, K)
        self.aJ = np.abs(self.J)
        self.y = np.asarray(ys, dtype=np.float64)
        self.K = K
        self.N = len(self.y)
        self.mu = None if prior_mu is None else np.asarray(prior_mu, dtype=np.float64)
        self.var = None if prior_var is None else np.asarray(prior_var, dtype=np.float64)
        self.JJt = self.J.dot(self.J.T)

    # force of the likelihood at precision tau, and its sum-of-magnitudes scale
    def force(self, theta, tau):
        r = (self.J.T.dot(theta) - self.y) * tau
        return self.J.dot(r), self.aJ.dot(np.abs(r))

    def grad_energy_theta(self, theta, tau):
        g, _ = self.force(theta, tau)
        if self.mu is not None:
            g = g + (theta - self.mu) / self.var      # the prior is in the energy (quirk Q4)
        return g

    def energy_terms(self, theta, p, tau):
        chi2 = np.sum((self.J.T.dot(theta) - self.y) ** 2)
        # magnitudes of the summed terms (likelihood, log Z, kinetic, the Gamma
        # prior's tau * rate with rate <= 1, the Gaussian prior)
        t = [0.5 * tau * chi2, 0.5 * self.N * abs(np.log(tau)), 0.5 * np.sum(p ** 2), tau]
        if self.mu is not None:
            t.append(0.5 * np.sum((theta - self.mu) ** 2 / self.var))
        return chi2, t

    def transition(self, q0, p0, tau, dt, L, bq0=None, btau=0.0, eps=EPS):
        """Bounds for one HMC transition started at (q0, p0).

        bq0   entrywise bound on the start-state difference (None = identical)
        btau  bound on |dtau| / tau
        Returns dict(bq, bp, be_before, be_after, q, p): entrywise bounds on the
        end-state difference of the PROPOSAL, energy bounds, and the numpy end
        state they were evaluated on."""
        K = self.K
        q = np.array(q0, dtype=np.float64)
        p = np.array(p0, dtype=np.float64)
        bq0 = np.zeros(K) if bq0 is None else np.asarray(bq0, dtype=np.float64)
        H = tau * self.JJt
        I, Z = np.eye(K), np.zeros((K, K))
        Dm = np.block([[I, dt * I], [Z, I]])
        w = [0.5 * dt] + [dt] * (L - 1) + [0.5 * dt]
        Km = [np.block([[I, Z], [-wj * H, I]]) for wj in w]

        # numpy trajectory: force and its scale at every kick
        inj = []
        chi2_0, terms0 = self.energy_terms(q, p, tau)
        gE0 = self.grad_energy_theta(q, tau)
        for j in range(L + 1):
            g, B = self.force(q, tau)
            inj.append(eps * B + btau * np.abs(g))
            p = p - w[j] * g
            if j < L:
                q = q + p * dt

        # suffix products: S_L = I, S_j = S_{j+1} K_{j+1} D
        S = [None] * (L + 1)
        S[L] = np.eye(2 * K)
        for j in range(L - 1, -1, -1):
            S[j] = S[j + 1].dot(Km[j + 1]).dot(Dm)
        P = S[0].dot(Km[0])
        bz = np.abs(P[:, :K]).dot(bq0)
        for j in range(L + 1):
            bz = bz + w[j] * np.abs(S[j][:, K:]).dot(inj[j])
        bq, bp = bz[:K], bz[K:]

        chi2_L, termsL = self.energy_terms(q, p, tau)
        dE_dtau0 = abs(0.5 * tau * chi2_0 - 0.5 * self.N)
        dE_dtauL = abs(0.5 * tau * chi2_L - 0.5 * self.N)
        slack0 = 16 * U * sum(terms0)
        slackL = 16 * U * sum(termsL)
        be_before = np.abs(gE0).dot(bq0) + btau * dE_dtau0 + slack0
        gEL = self.grad_energy_theta(q, tau)
        be_after = np.abs(gEL).dot(bq) + np.abs(p).dot(bp) + btau * dE_dtauL + slackL
        return dict(bq=bq, bp=bp, be_before=be_before, be_after=be_after, q=q, p=p)

    def gamma_update(self, theta, bq, beta):
        """tau' = g / rate, rate = 0.5 chi^2(theta) + beta
        (binf/example/samplers.py:34-49): relative bound on tau' given |dtheta| <= bq."""
        r = self.J.T.dot(theta) - self.y
        rate = 0.5 * np.sum(r ** 2) + beta
        drate = np.abs(self.J.dot(r)).dot(bq) + 8 * U * rate
        return drate / rate + 4 * U


def gibbs_bounds(pb, q_sweeps, accepted, p0, taus, tau0, q0, dt, L, beta, eps=EPS):
    """Chain the bounds over Gibbs sweeps (HMC on the coefficients, then the
    conjugate precision draw).  q_sweeps[s] / taus[s]: the reference's state
    after sweep s; accepted[s]; p0[s] the injected momentum.  Returns per-sweep
    lists of dicts(bq, btau, be_before, be_after)."""
    out = []
    bq = np.zeros(pb.K)
    btau = 0.0
    q_prev, tau_prev = np.asarray(q0, dtype=np.float64), float(tau0)
    for s in range(len(q_sweeps)):
        t = pb.transition(q_prev, p0[s], tau_prev, dt, L, bq0=bq, btau=btau, eps=eps)
        if accepted[s]:
            bq = t['bq'] + 4 * U * np.abs(q_sweeps[s])
        btau_new = pb.gamma_update(np.asarray(q_sweeps[s]), bq, beta)
        out.append(dict(bq=bq.copy(), btau=btau_new, be_before=t['be_before'],
                        be_after=t['be_after']))
        btau = btau_new
        q_prev, tau_prev = np.asarray(q_sweeps[s], dtype=np.float64), float(taus[s])
    return out
