// L-BFGS driver with liblbfgs-1.10 semantics, written against an abstract
// backend so that the SAME control flow runs
//   * device-resident for the log-weights method (N variables; every vector and
//     every dot product stays in HBM, one PCIe round trip per evaluation), and
//   * host-resident for the forces method (M variables, a few KB).
//
// What is reproduced (third-party/liblbfgs-1.10/lib/lbfgs.c):
//   parameter checks and error codes        :285-331
//   initial evaluation / "already minimal"  :412-451
//   initial step 1/|d|, later 1.0           :456, :614
//   convergence |g|/max(1,|x|) <= epsilon   :497-508
//   delta test over `past` iterations       :515-530
//   max_iterations                          :532-536
//   history update + two-loop recursion     :543-598   (backend)
//   backtracking line search                :645-734
//   More-Thuente line search                :812-976, :1125-1296
// Orthant-wise (OWL-QN) paths are not implemented: BioEn never enables them.
#pragma once

#include <cmath>
#include <vector>

#include "../../include/bioen_hip.h"

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

const char* lbfgs_code_string(int code);

// Values a backend reports for one evaluated trial point.
struct TrialResult {
    double f;       // objective at the trial point
    double dg;      // gradient(trial) . d
    double gg;      // |gradient(trial)|^2
    double xx;      // |x(trial)|^2
    double dginit;  // gradient(accepted) . d   (constant during a line search)
};

// ------------------------------------------------------------------------------------
// Line searches as resumable state machines: `first()` / `next()` hand out the trial
// step, `report()` consumes the evaluation.  Keeping them free of any vector work lets
// the device backend launch x = xp + stp*d and the evaluation without a host copy.
// ------------------------------------------------------------------------------------
class LineSearch {
  public:
    LineSearch(const bioen_lbfgs_config& cfg) : c_(cfg) {}
    // returns <0 on immediate error, otherwise 0 and sets `stp` to the first trial step
    int begin(double finit, double stp0, double* stp);
    // Feed the evaluation of the last trial. Returns: >0 = number of evaluations (done),
    // 0 = continue with *stp updated, <0 = liblbfgs error code.
    int report(const TrialResult& t, double* stp);
    int count() const { return count_; }

  private:
    int report_backtracking(const TrialResult& t, double* stp);
    int report_morethuente(const TrialResult& t, double* stp);
    void mt_prepare(double* stp);

    bioen_lbfgs_config c_;   // by value: machines are stored in containers
    int count_ = 0;
    bool have_dginit_ = false;
    double finit_ = 0, dginit_ = 0, dgtest_ = 0;
    // More-Thuente state
    int brackt_ = 0, stage1_ = 1, uinfo_ = 0;
    double stx_ = 0, fx_ = 0, dgx_ = 0, sty_ = 0, fy_ = 0, dgy_ = 0;
    double stmin_ = 0, stmax_ = 0, width_ = 0, prev_width_ = 0;
};

int validate_lbfgs_config(int n, const bioen_lbfgs_config& c);

// ------------------------------------------------------------------------------------
// One L-BFGS problem as a state machine (lbfgs.c:245-641 without the vector work).
// The owner evaluates points and builds directions; the machine decides.  Used by the
// single-problem loop below AND by the lock-step batch engine (several thetas advancing
// one evaluation per round against the same matrix pass).
// ------------------------------------------------------------------------------------
class LbfgsMachine {
  public:
    enum Kind { TRIAL, ACCEPT, DONE };
    struct Action {
        Kind kind;
        int end;          // ACCEPT: history slot receiving the new (s, y) pair
        int bound;        // ACCEPT: number of pairs the two-loop recursion uses
        int code;         // DONE: liblbfgs status
        bool keep_trial;  // DONE: result is the trial point (else the accepted point)
    };

    LbfgsMachine(int n, const bioen_lbfgs_config& cfg) : n_(n), cfg_(cfg), ls_(cfg_) {}

    // 0 = parameters fine, else the liblbfgs error code (no evaluation happens)
    int validate() const { return validate_lbfgs_config(n_, cfg_); }

    // Result of the evaluation at the start point (d = -g is built by the owner afterwards
    // unless the answer is DONE).  Returns TRIAL (go on) or DONE (already minimal).
    Action on_initial(double f, double gg, double xx);
    // Step length of the next point to evaluate: x = xp + step * d
    double trial_step() const { return stp_; }
    // Steps the backtracking searches (linesearch 1..3) can ask for NEXT, should the pending trial be
    // rejected: stp * 0.5 (sufficient-decrease or strong-Wolfe failure) and, for the Wolfe variants,
    // stp * 2.1 (curvature failure) -- formed exactly as report_backtracking forms them (lbfgs.c:686-727).
    // Lets an owner with idle batch slots evaluate them alongside the trial; More-Thuente steps depend on
    // the trial's values and cannot be foreseen (returns 0).
    int speculative_steps(double out[2]) const {
        if (cfg_.linesearch < 1 || cfg_.linesearch > 3) return 0;
        double dec = stp_, inc = stp_;
        dec *= 0.5;
        inc *= 2.1;
        out[0] = dec;
        if (cfg_.linesearch == 1) return 1;
        out[1] = inc;
        return 2;
    }
    // Result of that evaluation.
    Action on_trial(const TrialResult& t);

    double fx() const { return fx_; }
    int iterations() const { return iterations_; }
    int evaluations() const { return evaluations_; }

  private:
    void begin_linesearch(double step0);

    int n_;
    bioen_lbfgs_config cfg_;
    LineSearch ls_;
    std::vector<double> pf_;
    double fx_ = 0.0, stp_ = 0.0;
    int k_ = 1, end_ = 0;
    int iterations_ = 0, evaluations_ = 0;
    int ls_error_ = 0;
};

// Backend concept (xp/gp = accepted point and gradient, x/g = trial point and gradient):
//   void initial(double* f, double* gg, double* xx); // evaluate at x0 (= xp); d = -gp
//   void trial(double stp, TrialResult*);            // x = xp + stp d ; evaluate at x
//   void accept(int end, int bound);                 // s,y -> slot `end`; trial becomes the
//                                                    // accepted point; d = -H gp over the
//                                                    // `bound` newest pairs (lbfgs.c:543-598)
//   void revert();                                   // result = accepted point (xp)
//   void keep_trial();                               // result = trial point (x)
template <class Backend>
int lbfgs_run(Backend& B, int n, const bioen_lbfgs_config& cfg, double* fx_out, int* iterations_out,
              int* evaluations_out) {
    *iterations_out = 0;
    *evaluations_out = 0;
    *fx_out = 0.0;
    LbfgsMachine m(n, cfg);
    int code = m.validate();
    if (code != 0) return code;

    double f0, gg, xx;
    B.initial(&f0, &gg, &xx);
    LbfgsMachine::Action a = m.on_initial(f0, gg, xx);
    while (a.kind != LbfgsMachine::DONE) {
        if (a.kind == LbfgsMachine::ACCEPT) B.accept(a.end, a.bound);
        TrialResult tr{};
        B.trial(m.trial_step(), &tr);
        a = m.on_trial(tr);
    }
    if (a.keep_trial) B.keep_trial(); else B.revert();
    *fx_out = m.fx();
    *iterations_out = m.iterations();
    *evaluations_out = m.evaluations();
    return a.code;
}

}  // namespace bioen
