// Stand-in for tinyspline (ROS package tinyspline_ros, not installed here and not part of /root/reference), so that the reference's
// reference_path_smoother sources compile WHERE THEY LIE.  Only the members those files name exist; none of the pinned paths (the
// osqpSmooth / postSmooth assemblies) evaluates a B-spline, so eval() aborts.  TEST INFRASTRUCTURE ONLY.
#ifndef PO_REF_SHIM_TINYSPLINE
#define PO_REF_SHIM_TINYSPLINE
#include <cstdlib>
#include <vector>
namespace tinyspline {
typedef double real;
class DeBoorNet {
 public:
    std::vector<real> result() const { std::abort(); }
};
class BSpline {
 public:
    BSpline() {}
    BSpline(size_t n, size_t dim = 2, size_t deg = 3) : c_(n * dim, 0.0) { (void)deg; }
    std::vector<real> controlPoints() const { return c_; }
    void setControlPoints(const std::vector<real> &c) { c_ = c; }
    DeBoorNet eval(real) const { std::abort(); }
    DeBoorNet operator()(real) const { std::abort(); }
    BSpline derive(size_t = 1) const { std::abort(); }
 private:
    std::vector<real> c_;
};
}  // namespace tinyspline
#endif
