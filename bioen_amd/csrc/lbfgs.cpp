// Scalar side of the L-BFGS driver: parameter validation, status strings and the two
// line searches as state machines (see lbfgs.hpp for the reference line map).
#include "lbfgs.hpp"

#include <algorithm>

namespace bioen {

// Text of c_bioen_error.c:23-115 (lbfgs_strerror) -- the Python layer embeds it in the
// RuntimeError the reference raises (c_bioen.pyx:516-520).
const char* lbfgs_code_string(int code) {
    switch (code) {
        case LBFGS_CONVERGED: return "Convergence reached.";
        case LBFGS_STOPPED: return "LBFGS_STOP";
        case LBFGS_ALREADY_MINIMIZED: return "The initial variables already minimize the objective function.";
        case LBFGSERR_UNKNOWN: return "Unknown error.";
        case LBFGSERR_LOGIC: return "Logic error.";
        case LBFGSERR_OUTOFMEMORY: return "Insufficient memory.";
        case LBFGSERR_CANCELED: return "The minimization process has been canceled.";
        case LBFGSERR_INVALID_N: return "Invalid number of variables specified.";
        case LBFGSERR_INVALID_N_SSE: return "Invalid number of variables (for SSE) specified.";
        case LBFGSERR_INVALID_X_SSE: return "The array x must be aligned to 16 (for SSE).";
        case LBFGSERR_INVALID_EPSILON: return "Invalid parameter lbfgs_parameter_t::epsilon specified.";
        case LBFGSERR_INVALID_TESTPERIOD: return "Invalid parameter lbfgs_parameter_t::past specified.";
        case LBFGSERR_INVALID_DELTA: return "Invalid parameter lbfgs_parameter_t::delta specified.";
        case LBFGSERR_INVALID_LINESEARCH: return "Invalid parameter lbfgs_parameter_t::linesearch specified.";
        case LBFGSERR_INVALID_MINSTEP: return "Invalid parameter lbfgs_parameter_t::max_step specified";
        case LBFGSERR_INVALID_MAXSTEP: return "Invalid parameter lbfgs_parameter_t::max_step specified.";
        case LBFGSERR_INVALID_FTOL: return "Invalid parameter lbfgs_parameter_t::ftol specified.";
        case LBFGSERR_INVALID_WOLFE: return "Invalid parameter lbfgs_parameter_t::wolfe specified.";
        case LBFGSERR_INVALID_GTOL: return "Invalid parameter lbfgs_parameter_t::gtol specified.";
        case LBFGSERR_INVALID_XTOL: return "Invalid parameter lbfgs_parameter_t::xtol specified.";
        case LBFGSERR_INVALID_MAXLINESEARCH: return "Invalid parameter lbfgs_parameter_t::max_linesearch specified.";
        case LBFGSERR_INVALID_ORTHANTWISE: return "Invalid parameter lbfgs_parameter_t::orthantwise_c specified.";
        case LBFGSERR_INVALID_ORTHANTWISE_START:
            return "Invalid parameter lbfgs_parameter_t::orthantwise_start specified.";
        case LBFGSERR_INVALID_ORTHANTWISE_END:
            return "Invalid parameter lbfgs_parameter_t::orthantwise_end specified.";
        case LBFGSERR_OUTOFINTERVAL: return "The line-search step went out of the interval of uncertainty.";
        case LBFGSERR_INCORRECT_TMINMAX:
            return "A logic error occurred; alternatively, the interval of uncertainty";
        case LBFGSERR_ROUNDING_ERROR:
            return "A rounding error occurred; alternatively, no line-search step satisfies the sufficient "
                   "decrease and curvature conditions.";
        case LBFGSERR_MINIMUMSTEP: return "The line-search step became smaller than lbfgs_parameter_t::min_step.";
        case LBFGSERR_MAXIMUMSTEP: return "The line-search step became larger than lbfgs_parameter_t::max_step.";
        case LBFGSERR_MAXIMUMLINESEARCH: return "The line-search routine reaches the maximum number of evaluations.";
        case LBFGSERR_MAXIMUMITERATION: return "The algorithm routine reaches the maximum number of iterations.";
        case LBFGSERR_WIDTHTOOSMALL:
            return "Relative width of the interval of uncertainty is at most lbfgs_parameter_t::xtol.";
        case LBFGSERR_INVALIDPARAMETERS: return "A logic error (negative line-search step) occurred.";
        case LBFGSERR_INCREASEGRADIENT: return "The current search direction increases the objective function value.";
        default: return "(unknown)";
    }
}

// lbfgs.c:285-355, in liblbfgs' order (min_step/max_step/xtol/orthantwise stay at defaults)
int validate_lbfgs_config(int n, const bioen_lbfgs_config& c) {
    if (n <= 0) return LBFGSERR_INVALID_N;
    if (c.epsilon < 0.0) return LBFGSERR_INVALID_EPSILON;
    if (c.past < 0) return LBFGSERR_INVALID_TESTPERIOD;
    if (c.delta < 0.0) return LBFGSERR_INVALID_DELTA;
    if (c.ftol < 0.0) return LBFGSERR_INVALID_FTOL;
    if (c.linesearch == 2 || c.linesearch == 3)
        if (c.wolfe <= c.ftol || 1.0 <= c.wolfe) return LBFGSERR_INVALID_WOLFE;
    if (c.gtol < 0.0) return LBFGSERR_INVALID_GTOL;
    if (c.max_linesearch <= 0) return LBFGSERR_INVALID_MAXLINESEARCH;
    if (c.linesearch < 0 || c.linesearch > 3) return LBFGSERR_INVALID_LINESEARCH;
    return 0;
}

// ---------------------------------------------------------------------------------------
// interpolation helpers of the More-Thuente step selection (lbfgs.c:985-1070)
// ---------------------------------------------------------------------------------------
namespace {

inline double max3(double a, double b, double c) { return std::max(std::max(a, b), c); }

// minimiser of the cubic interpolating f,f' at u and v
double cubic(double u, double fu, double du, double v, double fv, double dv) {
    const double d = v - u;
    const double theta = (fu - fv) * 3.0 / d + du + dv;
    const double s = max3(std::fabs(theta), std::fabs(du), std::fabs(dv));
    const double a = theta / s;
    double gamma = s * std::sqrt(a * a - (du / s) * (dv / s));
    if (v < u) gamma = -gamma;
    const double p = gamma - du + theta;
    const double q = gamma - du + gamma + dv;
    return u + p / q * d;
}

// same with the safeguards of the "derivative decreases" case (returns lo/hi if the
// cubic has no minimiser beyond v)
double cubic_guarded(double u, double fu, double du, double v, double fv, double dv, double lo, double hi) {
    const double d = v - u;
    const double theta = (fu - fv) * 3.0 / d + du + dv;
    const double s = max3(std::fabs(theta), std::fabs(du), std::fabs(dv));
    const double a = theta / s;
    double gamma = s * std::sqrt(std::max(0.0, a * a - (du / s) * (dv / s)));
    if (u < v) gamma = -gamma;
    const double p = gamma - dv + theta;
    const double q = gamma - dv + gamma + du;
    const double r = p / q;
    if (r < 0.0 && gamma != 0.0) return v - r * d;
    return a < 0.0 ? hi : lo;
}

double quadratic(double u, double fu, double du, double v, double fv) {
    const double a = v - u;
    return u + du / ((fu - fv) / a + du) / 2.0 * a;
}

double secant(double u, double du, double v, double dv) {
    const double a = u - v;
    return v + dv / (dv - du) * a;
}

// lbfgs.c:1125-1296.  (x,fx,dx) best step, (y,fy,dy) other end point, t trial.
int update_interval(double& x, double& fx, double& dx, double& y, double& fy, double& dy, double& t, double ft,
                    double dt, double tmin, double tmax, int& brackt) {
    const bool opposite = dt * (dx / std::fabs(dx)) < 0.0;
    bool bound;
    double newt;

    if (brackt) {
        if (t <= std::min(x, y) || std::max(x, y) <= t) return LBFGSERR_OUTOFINTERVAL;
        if (0.0 <= dx * (t - x)) return LBFGSERR_INCREASEGRADIENT;
        if (tmax < tmin) return LBFGSERR_INCORRECT_TMINMAX;
    }

    if (fx < ft) {
        brackt = 1;
        bound = true;
        const double mc = cubic(x, fx, dx, t, ft, dt);
        const double mq = quadratic(x, fx, dx, t, ft);
        newt = (std::fabs(mc - x) < std::fabs(mq - x)) ? mc : mc + 0.5 * (mq - mc);
    } else if (opposite) {
        brackt = 1;
        bound = false;
        const double mc = cubic(x, fx, dx, t, ft, dt);
        const double mq = secant(x, dx, t, dt);
        newt = (std::fabs(mc - t) > std::fabs(mq - t)) ? mc : mq;
    } else if (std::fabs(dt) < std::fabs(dx)) {
        bound = true;
        const double mc = cubic_guarded(x, fx, dx, t, ft, dt, tmin, tmax);
        const double mq = secant(x, dx, t, dt);
        if (brackt)
            newt = (std::fabs(t - mc) < std::fabs(t - mq)) ? mc : mq;
        else
            newt = (std::fabs(t - mc) > std::fabs(t - mq)) ? mc : mq;
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

    newt = std::min(newt, tmax);
    newt = std::max(newt, tmin);
    if (brackt && bound) {
        const double mq = x + 0.66 * (y - x);
        if (x < y) newt = std::min(newt, mq);
        else       newt = std::max(newt, mq);
    }
    t = newt;
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------
int LineSearch::begin(double finit, double stp0, double* stp) {
    count_ = 0;
    have_dginit_ = false;
    finit_ = finit;
    if (stp0 <= 0.0) return LBFGSERR_INVALIDPARAMETERS;
    *stp = stp0;
    if (c_.linesearch == 0) {
        brackt_ = 0; stage1_ = 1; uinfo_ = 0;
        width_ = kMaxStep - kMinStep;
        prev_width_ = 2.0 * width_;
        stx_ = sty_ = 0.0;
        fx_ = fy_ = finit;
        mt_prepare(stp);
    }
    return 0;
}

int LineSearch::report(const TrialResult& t, double* stp) {
    if (!have_dginit_) {
        have_dginit_ = true;
        dginit_ = t.dginit;
        // "make sure that s points to a descent direction" (lbfgs.c:671-674, :845-848)
        if (0.0 < dginit_) return LBFGSERR_INCREASEGRADIENT;
        dgtest_ = c_.ftol * dginit_;
        dgx_ = dgy_ = dginit_;
    }
    return c_.linesearch == 0 ? report_morethuente(t, stp) : report_backtracking(t, stp);
}

// lbfgs.c:680-733
int LineSearch::report_backtracking(const TrialResult& t, double* stp) {
    ++count_;
    double width;
    if (t.f > finit_ + *stp * dgtest_) {
        width = 0.5;
    } else {
        if (c_.linesearch == 1) return count_;           // Armijo
        if (t.dg < c_.wolfe * dginit_) {
            width = 2.1;
        } else {
            if (c_.linesearch == 2) return count_;       // regular Wolfe
            if (t.dg > -c_.wolfe * dginit_)
                width = 0.5;
            else
                return count_;                           // strong Wolfe
        }
    }
    if (*stp < kMinStep) return LBFGSERR_MINIMUMSTEP;
    if (*stp > kMaxStep) return LBFGSERR_MAXIMUMSTEP;
    if (c_.max_linesearch <= count_) return LBFGSERR_MAXIMUMLINESEARCH;
    *stp *= width;
    return 0;
}

// the part of the More-Thuente loop that runs BEFORE an evaluation (lbfgs.c:871-893)
void LineSearch::mt_prepare(double* stp) {
    if (brackt_) {
        stmin_ = std::min(stx_, sty_);
        stmax_ = std::max(stx_, sty_);
    } else {
        stmin_ = stx_;
        stmax_ = *stp + 4.0 * (*stp - stx_);
    }
    if (*stp < kMinStep) *stp = kMinStep;
    if (kMaxStep < *stp) *stp = kMaxStep;
    if ((brackt_ && ((*stp <= stmin_ || stmax_ <= *stp) || c_.max_linesearch <= count_ + 1 || uinfo_ != 0)) ||
        (brackt_ && (stmax_ - stmin_ <= kXtol * stmax_)))
        *stp = stx_;
}

// the part AFTER an evaluation (lbfgs.c:903-975)
int LineSearch::report_morethuente(const TrialResult& t, double* stp) {
    const double f = t.f;
    double dg = t.dg;
    const double ftest1 = finit_ + *stp * dgtest_;
    ++count_;

    if (brackt_ && ((*stp <= stmin_ || stmax_ <= *stp) || uinfo_ != 0)) return LBFGSERR_ROUNDING_ERROR;
    if (*stp == kMaxStep && f <= ftest1 && dg <= dgtest_) return LBFGSERR_MAXIMUMSTEP;
    if (*stp == kMinStep && (ftest1 < f || dgtest_ <= dg)) return LBFGSERR_MINIMUMSTEP;
    if (brackt_ && (stmax_ - stmin_) <= kXtol * stmax_) return LBFGSERR_WIDTHTOOSMALL;
    if (c_.max_linesearch <= count_) return LBFGSERR_MAXIMUMLINESEARCH;
    if (f <= ftest1 && std::fabs(dg) <= c_.gtol * (-dginit_)) return count_;

    if (stage1_ && f <= ftest1 && std::min(c_.ftol, c_.gtol) * dginit_ <= dg) stage1_ = 0;

    if (stage1_ && ftest1 < f && f <= fx_) {
        // modified function psi(t) = f(t) - t * dgtest until a sufficient decrease is seen
        double fm = f - *stp * dgtest_, dgm = dg - dgtest_;
        double fxm = fx_ - stx_ * dgtest_, dgxm = dgx_ - dgtest_;
        double fym = fy_ - sty_ * dgtest_, dgym = dgy_ - dgtest_;
        uinfo_ = update_interval(stx_, fxm, dgxm, sty_, fym, dgym, *stp, fm, dgm, stmin_, stmax_, brackt_);
        fx_ = fxm + stx_ * dgtest_;
        fy_ = fym + sty_ * dgtest_;
        dgx_ = dgxm + dgtest_;
        dgy_ = dgym + dgtest_;
    } else {
        uinfo_ = update_interval(stx_, fx_, dgx_, sty_, fy_, dgy_, *stp, f, dg, stmin_, stmax_, brackt_);
    }

    if (brackt_) {
        if (0.66 * prev_width_ <= std::fabs(sty_ - stx_)) *stp = stx_ + 0.5 * (sty_ - stx_);
        prev_width_ = width_;
        width_ = std::fabs(sty_ - stx_);
    }
    mt_prepare(stp);
    return 0;
}

// ---------------------------------------------------------------------------------------
// LbfgsMachine: the control flow of lbfgs() (lbfgs.c:412-616)
// ---------------------------------------------------------------------------------------
void LbfgsMachine::begin_linesearch(double step0) {
    double stp = 0.0;
    ls_error_ = ls_.begin(fx_, step0, &stp);
    stp_ = stp;
}

LbfgsMachine::Action LbfgsMachine::on_initial(double f, double gg, double xx) {
    ++evaluations_;
    fx_ = f;
    pf_.assign(cfg_.past > 0 ? cfg_.past : 0, 0.0);
    if (!pf_.empty()) pf_[0] = f;
    double xnorm = std::sqrt(xx);
    const double gnorm = std::sqrt(gg);
    if (xnorm < 1.0) xnorm = 1.0;
    if (gnorm / xnorm <= cfg_.epsilon) return Action{DONE, 0, 0, LBFGS_ALREADY_MINIMIZED, false};
    k_ = 1;
    end_ = 0;
    begin_linesearch(1.0 / gnorm);   // d = -g  =>  |d| = |g|   (lbfgs.c:456)
    if (ls_error_ < 0) return Action{DONE, 0, 0, ls_error_, false};
    return Action{TRIAL, 0, 0, 0, false};
}

LbfgsMachine::Action LbfgsMachine::on_trial(const TrialResult& t) {
    ++evaluations_;
    double stp = stp_;
    const int st = ls_.report(t, &stp);
    if (st == 0) {          // line search wants another point
        stp_ = stp;
        return Action{TRIAL, 0, 0, 0, false};
    }
    if (st < 0) {
        // liblbfgs reverts to the previous point and returns the code; *ptr_fx keeps the
        // last trial's value (lbfgs.c:476-481,622-624)
        fx_ = t.f;
        return Action{DONE, 0, 0, st, false};
    }
    // accepted
    fx_ = t.f;
    double xnorm = std::sqrt(t.xx);
    const double gnorm = std::sqrt(t.gg);
    ++iterations_;   // progress callback, c_bioen_kernels_logw.c:565-576
    if (xnorm < 1.0) xnorm = 1.0;
    if (gnorm / xnorm <= cfg_.epsilon) return Action{DONE, 0, 0, LBFGS_CONVERGED, true};
    if (!pf_.empty()) {
        if (cfg_.past <= k_) {
            const double rate = (pf_[k_ % cfg_.past] - fx_) / fx_;
            if (rate < cfg_.delta) return Action{DONE, 0, 0, LBFGS_STOPPED, true};
        }
        pf_[k_ % cfg_.past] = fx_;
    }
    if (cfg_.max_iterations != 0 && cfg_.max_iterations < k_ + 1)
        return Action{DONE, 0, 0, LBFGSERR_MAXIMUMITERATION, true};

    const int bound = (kLbfgsM <= k_) ? kLbfgsM : k_;
    const int end = end_;
    ++k_;
    end_ = (end_ + 1) % kLbfgsM;
    begin_linesearch(1.0);           // "now the search direction d is ready. We try step = 1 first."
    if (ls_error_ < 0) return Action{DONE, 0, 0, ls_error_, true};
    return Action{ACCEPT, end, bound, 0, false};
}

}  // namespace bioen
