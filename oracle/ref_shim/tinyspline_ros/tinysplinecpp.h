// Stand-in for tinyspline (ROS package tinyspline_ros, not installed here and not part of /root/reference), so that the reference's
// reference_path_smoother / path_optimizer sources compile WHERE THEY LIE.  Only the members those files name exist.  eval() forwards
// to the oracle's restatement of tinyspline's clamped B-spline (po_oracle_bspline_eval: uniform interior knots on [0, 1], de Boor) —
// the library itself is absent, so this part is "parity unpinned" (po_oracle.h).  TEST INFRASTRUCTURE ONLY.
#ifndef PO_REF_SHIM_TINYSPLINE
#define PO_REF_SHIM_TINYSPLINE
#include <cstdlib>
#include <stdexcept>
#include <vector>
extern "C" {
#include "po_oracle.h"
}
namespace tinyspline {
typedef double real;
class DeBoorNet {
 public:
    DeBoorNet(real x, real y) : r_{x, y} {}
    std::vector<real> result() const { return r_; }
 private:
    std::vector<real> r_;
};
class BSpline {
 public:
    BSpline() : n_(0), dim_(2), deg_(3) {}
    BSpline(size_t n, size_t dim = 2, size_t deg = 3) : n_(n), dim_(dim), deg_(deg), c_(n * dim, 0.0) {
        if (dim != 2 || n <= deg) throw std::runtime_error("tinyspline stand-in: dim must be 2 and n_ctrlp > degree");
    }
    std::vector<real> controlPoints() const { return c_; }
    void setControlPoints(const std::vector<real> &c) { c_ = c; }
    DeBoorNet eval(real u) const {
        std::vector<real> cx(n_), cy(n_);
        for (size_t i = 0; i < n_; ++i) { cx[i] = c_[2 * i]; cy[i] = c_[2 * i + 1]; }
        real x = 0, y = 0;
        if (po_oracle_bspline_eval((int)n_, (int)deg_, cx.data(), cy.data(), u, &x, &y) != 0) std::abort();
        return DeBoorNet(x, y);
    }
    DeBoorNet operator()(real u) const { return eval(u); }
    BSpline derive(size_t = 1) const { std::abort(); }
 private:
    size_t n_, dim_, deg_;
    std::vector<real> c_;
};
}  // namespace tinyspline
#endif
