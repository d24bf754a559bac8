// The scalar side of liblbfgs-1.10's lbfgs() as a PLAIN-DATA state machine that compiles for the host AND for
// gfx950: the same source decides a line search on the host (forces method, More-Thuente A/B runs, self-tests) and
// inside the one-block decision kernel of the device-resident log-weights engine (kernels_devls.hip), so that a round
// needs no host turn-around.  Reference line map (third-party/liblbfgs-1.10/lib/lbfgs.c):
//   parameter checks and error codes        :285-331
//   initial evaluation / "already minimal"  :412-451
//   initial step 1/|d|, later 1.0           :456, :614
//   convergence |g|/max(1,|x|) <= epsilon   :497-508
//   delta test over `past` iterations       :515-530
//   max_iterations                          :532-536
//   backtracking line search                :645-734
//   More-Thuente line search                :812-976, update_trial_interval :1125-1296
//
// Bitwise host == device: every function here is compiled with floating-point contraction OFF (the host build has
// no FMA to contract into; the device would otherwise fuse a*b+c), min / max / fabs are written out, and the only
// library calls are IEEE-exact ones (sqrt, division).  tests: test_device_decisions_match_host_machine (GPU).
#pragma once

#include "../../include/bioen_hip.h"

#if defined(__HIPCC__)
#define BIOEN_HD __host__ __device__
#else
#define BIOEN_HD
#endif

namespace bioen {

// liblbfgs status codes (include/lbfgs.h:76-147)
enum LbfgsCode : int {
    LBFGS_CONVERGED = 0,
    LBFGS_STOPPED = 1,
    LBFGS_ALREADY_MINIMIZED = 2,
    LBFGSERR_UNKNOWN = -1024,
    LBFGSERR_LOGIC = -1023,
    LBFGSERR_OUTOFMEMORY = -1022,
    LBFGSERR_CANCELED = -1021,
    LBFGSERR_INVALID_N = -1020,
    LBFGSERR_INVALID_N_SSE = -1019,
    LBFGSERR_INVALID_X_SSE = -1018,
    LBFGSERR_INVALID_EPSILON = -1017,
    LBFGSERR_INVALID_TESTPERIOD = -1016,
    LBFGSERR_INVALID_DELTA = -1015,
    LBFGSERR_INVALID_LINESEARCH = -1014,
    LBFGSERR_INVALID_MINSTEP = -1013,
    LBFGSERR_INVALID_MAXSTEP = -1012,
    LBFGSERR_INVALID_FTOL = -1011,
    LBFGSERR_INVALID_WOLFE = -1010,
    LBFGSERR_INVALID_GTOL = -1009,
    LBFGSERR_INVALID_XTOL = -1008,
    LBFGSERR_INVALID_MAXLINESEARCH = -1007,
    LBFGSERR_INVALID_ORTHANTWISE = -1006,
    LBFGSERR_INVALID_ORTHANTWISE_START = -1005,
    LBFGSERR_INVALID_ORTHANTWISE_END = -1004,
    LBFGSERR_OUTOFINTERVAL = -1003,
    LBFGSERR_INCORRECT_TMINMAX = -1002,
    LBFGSERR_ROUNDING_ERROR = -1001,
    LBFGSERR_MINIMUMSTEP = -1000,
    LBFGSERR_MAXIMUMSTEP = -999,
    LBFGSERR_MAXIMUMLINESEARCH = -998,
    LBFGSERR_MAXIMUMITERATION = -997,
    LBFGSERR_WIDTHTOOSMALL = -996,
    LBFGSERR_INVALIDPARAMETERS = -995,
    LBFGSERR_INCREASEGRADIENT = -994
};

// liblbfgs defaults BioEn leaves untouched (lbfgs.c:113-118)
constexpr int kLbfgsM = 6;   // history length
constexpr double kMinStep = 1e-20;
constexpr double kMaxStep = 1e20;
constexpr double kXtol = 1e-16;

// Values a backend reports for one evaluated trial point.
struct TrialResult {
    double f;       // objective at the trial point
    double dg;      // gradient(trial) . d
    double gg;      // |gradient(trial)|^2
    double xx;      // |x(trial)|^2
    double dginit;  // gradient(accepted) . d   (constant during a line search)
};

// line-search state (backtracking needs the first five entries, More-Thuente all of them)
struct LsState {
    int count;
    int have_dginit;
    double finit, dginit, dgtest;
    int brackt, stage1, uinfo;
    int not_descent;        // the search was refused before its first evaluation (0 < dginit): see on_trial
    double stx, fx, dgx, sty, fy, dgy;
    double stmin, stmax, width, prev_width;
};

// one L-BFGS problem (lbfgs.c:245-641 without the vector work)
struct LbfgsState {
    LsState ls;
    double fx, stp;
    int k, end;
    int iterations, evaluations;
    int ls_error;
    int npf;          // entries of pf in use (= past, or 0)
    double* pf;       // past function values: host vector storage | a device array, set by the owner
};

enum ActionKind : int { ACT_TRIAL = 0, ACT_ACCEPT = 1, ACT_DONE = 2 };
struct LbfgsAction {
    int kind;
    int end;          // ACCEPT: history slot receiving the new (s, y) pair
    int bound;        // ACCEPT: number of pairs the two-loop recursion uses
    int code;         // DONE: liblbfgs status
    int keep_trial;   // DONE: result is the trial point (else the accepted point)
};

// Contraction is switched off INSIDE every function body (a pragma at file scope would leak into whatever includes
// this header and change the bits of the kernels; clang has no push / pop for it on this target).
#define BIOEN_NO_CONTRACT _Pragma("clang fp contract(off)")

namespace lb {

BIOEN_HD inline double dmin(double a, double b) { BIOEN_NO_CONTRACT return b < a ? b : a; }       // std::min
BIOEN_HD inline double dmax(double a, double b) { BIOEN_NO_CONTRACT return a < b ? b : a; }       // std::max
BIOEN_HD inline double dabs(double a) { BIOEN_NO_CONTRACT return __builtin_fabs(a); }
BIOEN_HD inline double dsqrt(double a) { BIOEN_NO_CONTRACT return __builtin_sqrt(a); }
BIOEN_HD inline double max3(double a, double b, double c) { BIOEN_NO_CONTRACT return dmax(dmax(a, b), c); }

// ---- interpolation helpers of the More-Thuente step selection (lbfgs.c:985-1070) --------------------
// minimiser of the cubic interpolating f, f' at u and v
BIOEN_HD inline double cubic(double u, double fu, double du, double v, double fv, double dv) {
    BIOEN_NO_CONTRACT
    const double d = v - u;
    const double theta = (fu - fv) * 3.0 / d + du + dv;
    const double s = max3(dabs(theta), dabs(du), dabs(dv));
    const double a = theta / s;
    double gamma = s * dsqrt(a * a - (du / s) * (dv / s));
    if (v < u) gamma = -gamma;
    const double p = gamma - du + theta;
    const double q = gamma - du + gamma + dv;
    return u + p / q * d;
}

// same with the safeguards of the "derivative decreases" case (returns lo / hi if the cubic has no minimiser beyond v)
BIOEN_HD inline double cubic_guarded(double u, double fu, double du, double v, double fv, double dv, double lo, double hi) {
    BIOEN_NO_CONTRACT
    const double d = v - u;
    const double theta = (fu - fv) * 3.0 / d + du + dv;
    const double s = max3(dabs(theta), dabs(du), dabs(dv));
    const double a = theta / s;
    double gamma = s * dsqrt(dmax(0.0, a * a - (du / s) * (dv / s)));
    if (u < v) gamma = -gamma;
    const double p = gamma - dv + theta;
    const double q = gamma - dv + gamma + du;
    const double r = p / q;
    if (r < 0.0 && gamma != 0.0) return v - r * d;
    return a < 0.0 ? hi : lo;
}

BIOEN_HD inline double quadratic(double u, double fu, double du, double v, double fv) {
    BIOEN_NO_CONTRACT
    const double a = v - u;
    return u + du / ((fu - fv) / a + du) / 2.0 * a;
}

BIOEN_HD inline double secant(double u, double du, double v, double dv) {
    BIOEN_NO_CONTRACT
    const double a = u - v;
    return v + dv / (dv - du) * a;
}

// lbfgs.c:1125-1296.  (x, fx, dx) best step, (y, fy, dy) other end point, t trial.
BIOEN_HD inline int update_interval(double& x, double& fx, double& dx, double& y, double& fy, double& dy, double& t,
                                    double ft, double dt, double tmin, double tmax, int& brackt) {
    BIOEN_NO_CONTRACT
    const bool opposite = dt * (dx / dabs(dx)) < 0.0;
    bool bound;
    double newt;

    if (brackt) {
        if (t <= dmin(x, y) || dmax(x, y) <= t) return LBFGSERR_OUTOFINTERVAL;
        if (0.0 <= dx * (t - x)) return LBFGSERR_INCREASEGRADIENT;
        if (tmax < tmin) return LBFGSERR_INCORRECT_TMINMAX;
    }

    if (fx < ft) {
        brackt = 1;
        bound = true;
        const double mc = cubic(x, fx, dx, t, ft, dt);
        const double mq = quadratic(x, fx, dx, t, ft);
        newt = (dabs(mc - x) < dabs(mq - x)) ? mc : mc + 0.5 * (mq - mc);
    } else if (opposite) {
        brackt = 1;
        bound = false;
        const double mc = cubic(x, fx, dx, t, ft, dt);
        const double mq = secant(x, dx, t, dt);
        newt = (dabs(mc - t) > dabs(mq - t)) ? mc : mq;
    } else if (dabs(dt) < dabs(dx)) {
        bound = true;
        const double mc = cubic_guarded(x, fx, dx, t, ft, dt, tmin, tmax);
        const double mq = secant(x, dx, t, dt);
        if (brackt)
            newt = (dabs(t - mc) < dabs(t - mq)) ? mc : mq;
        else
            newt = (dabs(t - mc) > dabs(t - mq)) ? mc : mq;
    } else {
        bound = false;
        if (brackt)
            newt = cubic(t, ft, dt, y, fy, dy);
        else
            newt = (x < t) ? tmax : tmin;
    }

    if (fx < ft) {
        y = t; fy = ft; dy = dt;
    } else {
        if (opposite) { y = x; fy = fx; dy = dx; }
        x = t; fx = ft; dx = dt;
    }

    newt = dmin(newt, tmax);
    newt = dmax(newt, tmin);
    if (brackt && bound) {
        const double mq = x + 0.66 * (y - x);
        if (x < y) newt = dmin(newt, mq);
        else       newt = dmax(newt, mq);
    }
    t = newt;
    return 0;
}

// ---- line searches as resumable state machines: begin hands out the first step, report consumes an evaluation ------
// the part of the More-Thuente loop that runs BEFORE an evaluation (lbfgs.c:871-893)
BIOEN_HD inline void mt_prepare(LsState& s, const bioen_lbfgs_config& c, double* stp) {
    BIOEN_NO_CONTRACT
    if (s.brackt) {
        s.stmin = dmin(s.stx, s.sty);
        s.stmax = dmax(s.stx, s.sty);
    } else {
        s.stmin = s.stx;
        s.stmax = *stp + 4.0 * (*stp - s.stx);
    }
    if (*stp < kMinStep) *stp = kMinStep;
    if (kMaxStep < *stp) *stp = kMaxStep;
    if ((s.brackt && ((*stp <= s.stmin || s.stmax <= *stp) || c.max_linesearch <= s.count + 1 || s.uinfo != 0)) ||
        (s.brackt && (s.stmax - s.stmin <= kXtol * s.stmax)))
        *stp = s.stx;
}

// returns < 0 on immediate error, otherwise 0 and sets *stp to the first trial step
BIOEN_HD inline int ls_begin(LsState& s, const bioen_lbfgs_config& c, double finit, double stp0, double* stp) {
    BIOEN_NO_CONTRACT
    s.count = 0;
    s.have_dginit = 0;
    s.not_descent = 0;
    s.finit = finit;
    if (stp0 <= 0.0) return LBFGSERR_INVALIDPARAMETERS;
    *stp = stp0;
    if (c.linesearch == 0) {
        s.brackt = 0; s.stage1 = 1; s.uinfo = 0;
        s.width = kMaxStep - kMinStep;
        s.prev_width = 2.0 * s.width;
        s.stx = s.sty = 0.0;
        s.fx = s.fy = finit;
        mt_prepare(s, c, stp);
    }
    return 0;
}

// lbfgs.c:680-733
BIOEN_HD inline int report_backtracking(LsState& s, const bioen_lbfgs_config& c, const TrialResult& t, double* stp) {
    BIOEN_NO_CONTRACT
    ++s.count;
    double width;
    if (t.f > s.finit + *stp * s.dgtest) {
        width = 0.5;
    } else {
        if (c.linesearch == 1) return s.count;           // Armijo
        if (t.dg < c.wolfe * s.dginit) {
            width = 2.1;
        } else {
            if (c.linesearch == 2) return s.count;       // regular Wolfe
            if (t.dg > -c.wolfe * s.dginit)
                width = 0.5;
            else
                return s.count;                          // strong Wolfe
        }
    }
    if (*stp < kMinStep) return LBFGSERR_MINIMUMSTEP;
    if (*stp > kMaxStep) return LBFGSERR_MAXIMUMSTEP;
    if (c.max_linesearch <= s.count) return LBFGSERR_MAXIMUMLINESEARCH;
    *stp *= width;
    return 0;
}

// the part AFTER an evaluation (lbfgs.c:903-975)
BIOEN_HD inline int report_morethuente(LsState& s, const bioen_lbfgs_config& c, const TrialResult& t, double* stp) {
    BIOEN_NO_CONTRACT
    const double f = t.f;
    double dg = t.dg;
    const double ftest1 = s.finit + *stp * s.dgtest;
    ++s.count;

    if (s.brackt && ((*stp <= s.stmin || s.stmax <= *stp) || s.uinfo != 0)) return LBFGSERR_ROUNDING_ERROR;
    if (*stp == kMaxStep && f <= ftest1 && dg <= s.dgtest) return LBFGSERR_MAXIMUMSTEP;
    if (*stp == kMinStep && (ftest1 < f || s.dgtest <= dg)) return LBFGSERR_MINIMUMSTEP;
    if (s.brackt && (s.stmax - s.stmin) <= kXtol * s.stmax) return LBFGSERR_WIDTHTOOSMALL;
    if (c.max_linesearch <= s.count) return LBFGSERR_MAXIMUMLINESEARCH;
    if (f <= ftest1 && dabs(dg) <= c.gtol * (-s.dginit)) return s.count;

    if (s.stage1 && f <= ftest1 && dmin(c.ftol, c.gtol) * s.dginit <= dg) s.stage1 = 0;

    if (s.stage1 && ftest1 < f && f <= s.fx) {
        // modified function psi(t) = f(t) - t * dgtest until a sufficient decrease is seen
        double fm = f - *stp * s.dgtest, dgm = dg - s.dgtest;
        double fxm = s.fx - s.stx * s.dgtest, dgxm = s.dgx - s.dgtest;
        double fym = s.fy - s.sty * s.dgtest, dgym = s.dgy - s.dgtest;
        s.uinfo = update_interval(s.stx, fxm, dgxm, s.sty, fym, dgym, *stp, fm, dgm, s.stmin, s.stmax, s.brackt);
        s.fx = fxm + s.stx * s.dgtest;
        s.fy = fym + s.sty * s.dgtest;
        s.dgx = dgxm + s.dgtest;
        s.dgy = dgym + s.dgtest;
    } else {
        s.uinfo = update_interval(s.stx, s.fx, s.dgx, s.sty, s.fy, s.dgy, *stp, f, dg, s.stmin, s.stmax, s.brackt);
    }

    if (s.brackt) {
        if (0.66 * s.prev_width <= dabs(s.sty - s.stx)) *stp = s.stx + 0.5 * (s.sty - s.stx);
        s.prev_width = s.width;
        s.width = dabs(s.sty - s.stx);
    }
    mt_prepare(s, c, stp);
    return 0;
}

// Feed the evaluation of the last trial.  > 0 = number of evaluations (done), 0 = continue with *stp updated,
// < 0 = liblbfgs error code.
BIOEN_HD inline int ls_report(LsState& s, const bioen_lbfgs_config& c, const TrialResult& t, double* stp) {
    BIOEN_NO_CONTRACT
    if (!s.have_dginit) {
        s.have_dginit = 1;
        s.dginit = t.dginit;
        // "make sure that s points to a descent direction" (lbfgs.c:671-674, :845-848)
        // NaN counts as "not a descent direction", as in the reference's build (-ffast-math: the test comes out as
        // !(dginit <= 0)): beyond the rounding floor a pair with y.s = 0 turns the two-loop recursion's direction into NaN;
        // the reference's binary ends such a run with -994 and the last accepted point (tools/fuzz_batch.py, seed 23: Armijo
        // search, epsilon below the gradient's noise), IEEE rules would carry the NaN into x and call it converged
        if (!(s.dginit <= 0.0)) {
            s.not_descent = 1;
            return LBFGSERR_INCREASEGRADIENT;
        }
        s.dgtest = c.ftol * s.dginit;
        s.dgx = s.dgy = s.dginit;
    }
    return c.linesearch == 0 ? report_morethuente(s, c, t, stp) : report_backtracking(s, c, t, stp);
}

// ---- the control flow of lbfgs() (lbfgs.c:412-616) ------------------------------------------------------------------
BIOEN_HD inline void begin_linesearch(LbfgsState& m, const bioen_lbfgs_config& c, double step0) {
    BIOEN_NO_CONTRACT
    double stp = 0.0;
    m.ls_error = ls_begin(m.ls, c, m.fx, step0, &stp);
    m.stp = stp;
}

// before the first evaluation; pf: storage for max(past, 0) doubles (may be null when past <= 0)
BIOEN_HD inline void machine_reset(LbfgsState& m, const bioen_lbfgs_config& c, double* pf) {
    BIOEN_NO_CONTRACT
    m.ls = LsState{};
    m.fx = 0.0;
    m.stp = 0.0;
    m.k = 1;
    m.end = 0;
    m.iterations = 0;
    m.evaluations = 0;
    m.ls_error = 0;
    m.npf = c.past > 0 ? c.past : 0;
    m.pf = pf;
}

// Result of the evaluation at the start point (d = -g is built by the owner afterwards unless the answer is DONE).
BIOEN_HD inline LbfgsAction on_initial(LbfgsState& m, const bioen_lbfgs_config& c, double f, double gg, double xx) {
    BIOEN_NO_CONTRACT
    ++m.evaluations;
    m.fx = f;
    for (int i = 0; i < m.npf; ++i) m.pf[i] = 0.0;
    if (m.npf > 0) m.pf[0] = f;
    double xnorm = dsqrt(xx);
    const double gnorm = dsqrt(gg);
    if (xnorm < 1.0) xnorm = 1.0;
    // lbfgs.c:447 writes `gnorm / xnorm <= epsilon`; the reference BUILDS its liblbfgs with -ffast-math (the library's own
    // Makefile), where the test comes out as !(... > epsilon): a non-finite start (NaN / inf in g, G, yTilde,
    // YTilde, theta, forces, w0) ends the run at once with LBFGS_ALREADY_MINIMIZED and the start point -- measured against
    // oracle/_ref (tools/attic/nan_probe.py, tests/test_hip_edgecases.py).  Under IEEE rules the `<=` form would iterate on NaN
    // until max_iterations (5000 lock-step rounds of nothing).
    if (!(gnorm / xnorm > c.epsilon)) return LbfgsAction{ACT_DONE, 0, 0, LBFGS_ALREADY_MINIMIZED, 0};
    m.k = 1;
    m.end = 0;
    begin_linesearch(m, c, 1.0 / gnorm);   // d = -g  =>  |d| = |g|   (lbfgs.c:456)
    if (m.ls_error < 0) return LbfgsAction{ACT_DONE, 0, 0, m.ls_error, 0};
    return LbfgsAction{ACT_TRIAL, 0, 0, 0, 0};
}

// Result of the evaluation of x = xp + stp d.
BIOEN_HD inline LbfgsAction on_trial(LbfgsState& m, const bioen_lbfgs_config& c, const TrialResult& t) {
    BIOEN_NO_CONTRACT
    ++m.evaluations;
    double stp = m.stp;
    const int st = ls_report(m.ls, c, t, &stp);
    if (st == 0) {          // line search wants another point
        m.stp = stp;
        return LbfgsAction{ACT_TRIAL, 0, 0, 0, 0};
    }
    if (st < 0) {
        // liblbfgs reverts to the previous point and returns the code; *ptr_fx keeps the last trial's value
        // (lbfgs.c:476-481, 622-624) -- unless the search was refused BEFORE its first evaluation: "make sure that s points
        // to a descent direction" (lbfgs.c:671-674, :845-848) returns with fx untouched, the accepted point's value.  Here
        // the initial slope arrives with the first trial's results, i.e. one evaluation late: that evaluation is not the
        // reference's (not counted) and its value is not the run's (r04, found by tools/fuzz_parity.py: fmin 139.1 where
        // the reference and the restatement return 81.9, status -994 on all three).
        if (m.ls.not_descent) --m.evaluations;
        else m.fx = t.f;
        return LbfgsAction{ACT_DONE, 0, 0, st, 0};
    }
    // accepted
    m.fx = t.f;
    double xnorm = dsqrt(t.xx);
    const double gnorm = dsqrt(t.gg);
    ++m.iterations;   // progress callback, c_bioen_kernels_logw.c:565-576
    if (xnorm < 1.0) xnorm = 1.0;
    if (!(gnorm / xnorm > c.epsilon)) return LbfgsAction{ACT_DONE, 0, 0, LBFGS_CONVERGED, 1};   // NaN-aware, as on_initial
    if (m.npf > 0) {
        if (c.past <= m.k) {
            const double rate = (m.pf[m.k % c.past] - m.fx) / m.fx;
            if (rate < c.delta) return LbfgsAction{ACT_DONE, 0, 0, LBFGS_STOPPED, 1};
        }
        m.pf[m.k % c.past] = m.fx;
    }
    if (c.max_iterations != 0 && c.max_iterations < m.k + 1)
        return LbfgsAction{ACT_DONE, 0, 0, LBFGSERR_MAXIMUMITERATION, 1};

    const int bound = (kLbfgsM <= m.k) ? kLbfgsM : m.k;
    const int end = m.end;
    ++m.k;
    m.end = (m.end + 1) % kLbfgsM;
    begin_linesearch(m, c, 1.0);           // "now the search direction d is ready. We try step = 1 first."
    if (m.ls_error < 0) return LbfgsAction{ACT_DONE, 0, 0, m.ls_error, 1};
    return LbfgsAction{ACT_ACCEPT, end, bound, 0, 0};
}

// Steps the backtracking searches (linesearch 1..3) can ask for NEXT, should the pending trial be rejected:
// stp * 0.5 (sufficient-decrease or strong-Wolfe failure) and, for the Wolfe variants, stp * 2.1 (curvature failure)
// -- formed exactly as report_backtracking forms them (lbfgs.c:686-727).  More-Thuente steps depend on the trial's
// values and cannot be foreseen (returns 0).
BIOEN_HD inline int speculative_steps(double stp, int linesearch, double out[2]) {
    BIOEN_NO_CONTRACT
    if (linesearch < 1 || linesearch > 3) return 0;
    double dec = stp, inc = stp;
    dec *= 0.5;
    inc *= 2.1;
    out[0] = dec;
    if (linesearch == 1) return 1;
    out[1] = inc;
    return 2;
}

}  // namespace lb

}  // namespace bioen
