// Host backend of bioen_hip_selftest_lbfgs (part of api.hip's translation unit: uses its static helpers).

// analytic objectives of bioen_hip_selftest_lbfgs (host only, test hook)
static double selftest_objective(int kind, int n, const double* x, double* g) {
    double f = 0.0;
    for (int i = 0; i < n; ++i) g[i] = 0.0;
    if (kind == 0) {   // extended Rosenbrock over consecutive pairs
        for (int i = 0; i + 1 < n; i += 2) {
            const double t1 = 1.0 - x[i];
            const double t2 = 10.0 * (x[i + 1] - x[i] * x[i]);
            g[i + 1] = 20.0 * t2;
            g[i] = -2.0 * (x[i] * g[i + 1] + t1);
            f += t1 * t1 + t2 * t2;
        }
        if (n & 1) {
            f += x[n - 1] * x[n - 1];
            g[n - 1] = 2.0 * x[n - 1];
        }
    } else {           // sum_i c_i (x_i - 1)^2 + 0.01 (x_i - 1)^4, c_i spread over 4 decades
        for (int i = 0; i < n; ++i) {
            const double c = std::pow(10.0, 4.0 * i / (n > 1 ? n - 1 : 1) - 2.0);
            const double d = x[i] - 1.0;
            f += c * d * d + 0.01 * d * d * d * d;
            g[i] = 2.0 * c * d + 0.04 * d * d * d;
        }
    }
    return f;
}

struct HostSelftestBackend {
    int kind, n;
    std::vector<double> x, xp, g, gp, d;
    std::vector<double> S[kHistory], Y[kHistory];
    double ys[kHistory] = {}, alpha[kHistory] = {};
    bool result_is_trial = false;

    HostSelftestBackend(int k, int nn, const double* x0) : kind(k), n(nn), x(nn), xp(x0, x0 + nn), g(nn), gp(nn), d(nn) {
        for (int i = 0; i < kHistory; ++i) {
            S[i].assign(nn, 0.0);
            Y[i].assign(nn, 0.0);
        }
    }
    static double dot(const std::vector<double>& a, const std::vector<double>& b) {
        double s = 0.0;
        for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
        return s;
    }
    void initial(double* f, double* gg, double* xx) {
        *f = selftest_objective(kind, n, xp.data(), gp.data());
        *gg = dot(gp, gp);
        *xx = dot(xp, xp);
        for (int i = 0; i < n; ++i) d[i] = -gp[i];
    }
    void trial(double stp, TrialResult* t) {
        for (int i = 0; i < n; ++i) x[i] = xp[i] + stp * d[i];
        t->f = selftest_objective(kind, n, x.data(), g.data());
        t->dg = dot(g, d);
        t->gg = dot(g, g);
        t->xx = dot(x, x);
        t->dginit = dot(gp, d);
    }
    void accept(int end, int bound) {
        for (int i = 0; i < n; ++i) {
            S[end][i] = x[i] - xp[i];
            Y[end][i] = g[i] - gp[i];
        }
        const double ys_new = dot(Y[end], S[end]), yy = dot(Y[end], Y[end]);
        ys[end] = ys_new;
        x.swap(xp);
        g.swap(gp);
        for (int i = 0; i < n; ++i) d[i] = -gp[i];
        int j = (end + 1) % kHistory;
        for (int b = 0; b < bound; ++b) {
            j = (j + kHistory - 1) % kHistory;
            alpha[j] = dot(S[j], d) / ys[j];
            for (int i = 0; i < n; ++i) d[i] -= alpha[j] * Y[j][i];
        }
        const double sc = ys_new / yy;
        for (int i = 0; i < n; ++i) d[i] *= sc;
        for (int b = 0; b < bound; ++b) {
            const double beta = dot(Y[j], d) / ys[j];
            const double coef = alpha[j] - beta;
            for (int i = 0; i < n; ++i) d[i] += coef * S[j][i];
            j = (j + 1) % kHistory;
        }
    }
    void revert() { result_is_trial = false; }
    void keep_trial() { result_is_trial = true; }
};

