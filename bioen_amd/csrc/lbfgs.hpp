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
#include "lbfgs_state.hpp"

namespace bioen {

const char* lbfgs_code_string(int code);

int validate_lbfgs_config(int n, const bioen_lbfgs_config& c);

// ------------------------------------------------------------------------------------
// One L-BFGS problem as a state machine (lbfgs.c:245-641 without the vector work): the host-side owner of an
// LbfgsState (lbfgs_state.hpp -- the decisions themselves are plain functions that also run inside the decision kernel
// of the device-resident engine).  The owner evaluates points and builds directions; the machine decides.  Used by the
// single-problem loop below AND by the lock-step batch engines (several thetas advancing one evaluation per round
// against the same matrix pass).  Line searches are resumable: trial_step() hands out the step, on_trial() consumes the
// evaluation -- no vector work in here, so a device backend launches x = xp + stp d without a host copy.
// ------------------------------------------------------------------------------------
class LbfgsMachine {
  public:
    enum Kind { TRIAL = ACT_TRIAL, ACCEPT = ACT_ACCEPT, DONE = ACT_DONE };
    struct Action {
        Kind kind;
        int end;          // ACCEPT: history slot receiving the new (s, y) pair
        int bound;        // ACCEPT: number of pairs the two-loop recursion uses
        int code;         // DONE: liblbfgs status
        bool keep_trial;  // DONE: result is the trial point (else the accepted point)
    };

    LbfgsMachine(int n, const bioen_lbfgs_config& cfg) : n_(n), cfg_(cfg) { lb::machine_reset(st_, cfg_, nullptr); }

    // 0 = parameters fine, else the liblbfgs error code (no evaluation happens)
    int validate() const { return validate_lbfgs_config(n_, cfg_); }

    // Result of the evaluation at the start point (d = -g is built by the owner afterwards
    // unless the answer is DONE).  Returns TRIAL (go on) or DONE (already minimal).
    Action on_initial(double f, double gg, double xx) {
        pf_.assign(cfg_.past > 0 ? cfg_.past : 0, 0.0);
        lb::machine_reset(st_, cfg_, pf_.data());
        return convert(lb::on_initial(st_, cfg_, f, gg, xx));
    }
    // Step length of the next point to evaluate: x = xp + step * d
    double trial_step() const { return st_.stp; }
    // Steps the backtracking searches can ask for NEXT, should the pending trial be rejected (lbfgs_state.hpp)
    int speculative_steps(double out[2]) const { return lb::speculative_steps(st_.stp, cfg_.linesearch, out); }
    // Result of that evaluation.
    Action on_trial(const TrialResult& t) {
        st_.pf = pf_.data();          // machines live in containers: the storage may have moved with them
        return convert(lb::on_trial(st_, cfg_, t));
    }

    double fx() const { return st_.fx; }
    int iterations() const { return st_.iterations; }
    int evaluations() const { return st_.evaluations; }

  private:
    static Action convert(const LbfgsAction& a) {
        return Action{static_cast<Kind>(a.kind), a.end, a.bound, a.code, a.keep_trial != 0};
    }
    int n_;
    bioen_lbfgs_config cfg_;
    LbfgsState st_;
    std::vector<double> pf_;
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
