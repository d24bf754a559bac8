// Scalar pieces of the GSL-style minimizers (multimin.hpp): Fletcher's interpolation formulas,
// status texts and the analytic objectives of GSL's multimin test programme.
#include "multimin.hpp"

#include <algorithm>

namespace bioen {
namespace multimin {

const char* status_string(int code) {   // gsl-2.5/err/strerror.c
    switch (code) {
        case SUCCESS: return "success";
        case CONTINUE: return "the iteration has not converged yet";
        case EBADTOL: return "specified tolerance is invalid or theoretically unattainable";
        case ENOPROG: return "iteration is not making progress towards solution";
        case EBACKEND: return "failure";
        default: return "unknown error code";
    }
}

const char* algorithm_name(int algorithm) {   // c_bioen_common.h:36-38
    static const char* names[] = {"fdfminimizer_conjugate_fr", "fdfminimizer_conjugate_pr", "fdfminimizer_vector_bfgs2",
                                  "fdfminimizer_vector_bfgs", "fdfminimizer_steepest_descent"};
    return (algorithm >= 0 && algorithm <= 4) ? names[algorithm] : "unknown";
}

int quadratic_roots(double a, double b, double c, double* r0, double* r1) {   // poly/solve_quadratic.c:27-84
    if (a == 0) {
        if (b == 0) return 0;
        *r0 = -c / b;
        return 1;
    }
    const double disc = b * b - 4 * a * c;
    if (disc < 0) return 0;
    if (disc == 0) {
        *r0 = *r1 = -0.5 * b / a;
        return 2;
    }
    if (b == 0) {
        const double r = std::sqrt(-c / a);
        *r0 = -r;
        *r1 = r;
        return 2;
    }
    const double t = -0.5 * (b + (b > 0 ? 1 : -1) * std::sqrt(disc));   // the cancellation-free root first
    const double ra = t / a, rb = c / t;
    *r0 = ra < rb ? ra : rb;
    *r1 = ra < rb ? rb : ra;
    return 2;
}

namespace {

struct Best {           // running minimum over candidate abscissae
    double z, f;
    void offer(double zc, double fc) {
        if (fc < f) { z = zc; f = fc; }
    }
};

// minimum on [zl, zh] of q(z) = f0 + fp0 z + (f1 - f0 - fp0) z^2      (linear_minimize.c:10-33)
double quadratic_min(double f0, double fp0, double f1, double zl, double zh) {
    const double k = f1 - f0 - fp0;
    auto q = [&](double z) { return f0 + z * (fp0 + z * k); };
    Best best{zl, q(zl)};
    best.offer(zh, q(zh));
    const double curv = 2 * k;
    if (curv > 0) {
        const double z = -fp0 / curv;
        if (z > zl && z < zh) best.offer(z, q(z));
    }
    return best.z;
}

// minimum on [zl, zh] of the Hermite cubic through (0,f0,fp0), (1,f1,fp1)   (:45-100)
double cubic_min(double f0, double fp0, double f1, double fp1, double zl, double zh) {
    const double c2 = 3 * (f1 - f0) - 2 * fp0 - fp1;
    const double c3 = fp0 + fp1 - 2 * (f1 - f0);
    auto c = [&](double z) { return f0 + z * (fp0 + z * (c2 + z * c3)); };
    Best best{zl, c(zl)};
    best.offer(zh, c(zh));
    double z0 = 0.0, z1 = 0.0;
    const int nr = quadratic_roots(3 * c3, 2 * c2, fp0, &z0, &z1);
    if (nr >= 1 && z0 > zl && z0 < zh) best.offer(z0, c(z0));
    if (nr == 2 && z1 > zl && z1 < zh) best.offer(z1, c(z1));
    return best.z;
}

}  // namespace

double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax,
                   int order) {   // :103-131
    const double w = b - a;
    double zlo = (xmin - a) / w, zhi = (xmax - a) / w;
    if (zlo > zhi) std::swap(zlo, zhi);
    const double z = (order > 2 && std::isfinite(fpb)) ? cubic_min(fa, fpa * w, fb, fpb * w, zlo, zhi)
                                                       : quadratic_min(fa, fpa * w, fb, zlo, zhi);
    return a + z * w;
}

int test_function_dim(int kind) { return kind == 1 ? 4 : 2; }

void test_function(int kind, const double* x, double* f, double* g) {   // multimin/test_funcs.c
    if (kind == 0) {            // Roth
        const double u = x[0], v = x[1];
        const double a = -13.0 + u + ((5.0 - v) * v - 2.0) * v;
        const double b = -29.0 + u + ((v + 1.0) * v - 14.0) * v;
        if (f) *f = a * a + b * b;
        if (g) {
            g[0] = 2 * a + 2 * b;
            g[1] = 2 * a * (-2 + v * (10 - 3 * v)) + 2 * b * (-14 + v * (2 + 3 * v));
        }
    } else if (kind == 1) {     // Wood
        const double t1 = x[0] * x[0] - x[1], t2 = x[2] * x[2] - x[3];
        const double p0 = 1 - x[0], p1 = 1 - x[1], p2 = 1 - x[2], p3 = 1 - x[3];
        if (f) *f = 100 * t1 * t1 + p0 * p0 + 90 * t2 * t2 + p2 * p2 + 10.1 * (p1 * p1 + p3 * p3) + 19.8 * p1 * p3;
        if (g) {
            g[0] = 400 * x[0] * t1 - 2 * p0;
            g[1] = -200 * t1 - 20.2 * p1 - 19.8 * p3;
            g[2] = 360 * x[2] * t2 - 2 * p2;
            g[3] = -180 * t2 - 20.2 * p3 - 19.8 * p1;
        }
    } else if (kind == 2) {     // Rosenbrock with GSL's factor 10
        const double a = x[0] - 1, b = x[0] * x[0] - x[1];
        if (f) *f = a * a + 10 * b * b;
        if (g) {
            g[0] = 2 * a + 40 * x[0] * b;
            g[1] = -20 * b;
        }
    } else {                    // |u - 1| + |v - 2| with GSL_SIGN(0) = +1
        const double a = x[0] - 1, b = x[1] - 2;
        if (f) *f = std::fabs(a) + std::fabs(b);
        if (g) {
            g[0] = a >= 0.0 ? 1.0 : -1.0;
            g[1] = b >= 0.0 ? 1.0 : -1.0;
        }
    }
}

}  // namespace multimin
}  // namespace bioen
