// stand-in for osqp-eigen (pre-0.7 API, as the reference uses it in src/solver/solver.cpp:48-74): captures the data
// the reference hands to OSQP and solves it with the oracle's OSQP-style ADMM.  TEST INFRASTRUCTURE ONLY.
#ifndef PO_REF_SHIM_OSQPEIGEN
#define PO_REF_SHIM_OSQPEIGEN
#include <Eigen/Dense>
#include <memory>
#include <vector>
extern "C" {
#include "po_oracle.h"
}
namespace OsqpEigen {
const double INFTY = 1e30;  // == OSQP_INFTY
struct Captured {  // last problem handed over, readable by the glue
    int n = 0, m = 0;
    std::vector<int> Pp, Pi, Ap, Ai;
    std::vector<double> Px, Ax, q, l, u, x, y;
    po_info info{};
    bool solved = false;
};
Captured &last_captured();
const po_params &shim_params();
class Settings { public: void setVerbosity(bool) {} void setWarmStart(bool) {} };
class Data {
 public:
    explicit Data(Captured &c) : c_(c) {}
    void setNumberOfVariables(int n) { c_.n = n; }
    void setNumberOfConstraints(int m) { c_.m = m; }
    bool setHessianMatrix(const Eigen::SparseMatrix<double> &h) {  // OsqpEigen keeps the upper triangle
        to_csc(h, true, c_.Pp, c_.Pi, c_.Px);
        return h.rows() == c_.n && h.cols() == c_.n;
    }
    bool setGradient(const Eigen::MatBase &g) { c_.q.assign((size_t)g.size(), 0.0); for (long i = 0; i < g.size(); ++i) c_.q[(size_t)i] = g(i); return g.size() == c_.n; }
    bool setLinearConstraintsMatrix(const Eigen::SparseMatrix<double> &a) { to_csc(a, false, c_.Ap, c_.Ai, c_.Ax); return a.rows() == c_.m && a.cols() == c_.n; }
    bool setLowerBound(const Eigen::MatBase &v) { c_.l.resize((size_t)v.size()); for (long i = 0; i < v.size(); ++i) c_.l[(size_t)i] = v(i); return v.size() == c_.m; }
    bool setUpperBound(const Eigen::MatBase &v) { c_.u.resize((size_t)v.size()); for (long i = 0; i < v.size(); ++i) c_.u[(size_t)i] = v(i); return v.size() == c_.m; }
 private:
    static void to_csc(const Eigen::SparseMatrix<double> &s, bool upper, std::vector<int> &p, std::vector<int> &idx, std::vector<double> &val) {
        p.assign((size_t)s.cols() + 1, 0); idx.clear(); val.clear();
        for (const auto &kv : s.entries()) {  // map is ordered by (col,row)
            const long col = kv.first.first, row = kv.first.second;
            if (upper && row > col) continue;
            idx.push_back((int)row); val.push_back(kv.second); p[(size_t)col + 1]++;
        }
        for (size_t c = 0; c < (size_t)s.cols(); ++c) p[c + 1] += p[c];
    }
    Captured &c_;
};
class Solver {
 public:
    Solver() : data_(last_captured()) {}
    Settings *settings() { return &settings_; }
    Data *data() { return &data_; }
    bool initSolver() { return true; }
    bool solve() {
        Captured &c = last_captured();
        c.x.assign((size_t)c.n, 0.0); c.y.assign((size_t)c.m, 0.0);
        std::vector<double> z((size_t)c.m, 0.0);
        const int rc = po_oracle_qp_solve(c.n, c.m, c.Pp.data(), c.Pi.data(), c.Px.data(), c.q.data(), c.Ap.data(), c.Ai.data(), c.Ax.data(),
                                          c.l.data(), c.u.data(), &shim_params(), nullptr, c.x.data(), c.y.data(), z.data(), &c.info);
        c.solved = (rc == 0 && c.info.status == PO_STATUS_SOLVED);
        if (c.solved) { sol_ = Eigen::VectorXd::Zero(c.n); for (int i = 0; i < c.n; ++i) sol_(i) = c.x[(size_t)i]; }
        return c.solved;  // osqp-eigen: true only for OSQP_SOLVED
    }
    const Eigen::VectorXd &getSolution() { return sol_; }
 private:
    Settings settings_;
    Data data_;
    Eigen::VectorXd sol_;
};
}  // namespace OsqpEigen
#endif
