// Stand-in for CppAD (not installed, not part of /root/reference): just enough surface for the reference's IPOPT branches
// (FgEval* functors, ipoptSmooth) to COMPILE where they lie.  Those branches are not the shipped default (FLAGS_tension_solver =
// "OSQP") and are never executed by the glue: CppAD::ipopt::solve() aborts.  TEST INFRASTRUCTURE ONLY.
#ifndef PO_REF_SHIM_CPPAD
#define PO_REF_SHIM_CPPAD
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>
#define CPPAD_TESTVECTOR(T) std::vector<T>
namespace CppAD {
template <typename T> class AD {
 public:
    AD() : v_(0) {}
    AD(T v) : v_(v) {}
    AD &operator+=(const AD &o) { v_ += o.v_; return *this; }
    AD &operator-=(const AD &o) { v_ -= o.v_; return *this; }
    AD &operator*=(const AD &o) { v_ *= o.v_; return *this; }
    T v_;
};
template <typename T> AD<T> operator+(const AD<T> &a, const AD<T> &b) { return AD<T>(a.v_ + b.v_); }
template <typename T> AD<T> operator-(const AD<T> &a, const AD<T> &b) { return AD<T>(a.v_ - b.v_); }
template <typename T> AD<T> operator*(const AD<T> &a, const AD<T> &b) { return AD<T>(a.v_ * b.v_); }
template <typename T> AD<T> operator/(const AD<T> &a, const AD<T> &b) { return AD<T>(a.v_ / b.v_); }
template <typename T> AD<T> operator-(const AD<T> &a) { return AD<T>(-a.v_); }
template <typename T> AD<T> operator+(const AD<T> &a, double b) { return AD<T>(a.v_ + b); }
template <typename T> AD<T> operator-(const AD<T> &a, double b) { return AD<T>(a.v_ - b); }
template <typename T> AD<T> operator*(const AD<T> &a, double b) { return AD<T>(a.v_ * b); }
template <typename T> AD<T> operator/(const AD<T> &a, double b) { return AD<T>(a.v_ / b); }
template <typename T> AD<T> operator+(double a, const AD<T> &b) { return AD<T>(a + b.v_); }
template <typename T> AD<T> operator-(double a, const AD<T> &b) { return AD<T>(a - b.v_); }
template <typename T> AD<T> operator*(double a, const AD<T> &b) { return AD<T>(a * b.v_); }
template <typename T> AD<T> operator/(double a, const AD<T> &b) { return AD<T>(a / b.v_); }
template <typename T> AD<T> operator*(int a, const AD<T> &b) { return AD<T>(a * b.v_); }
template <typename T> AD<T> operator*(const AD<T> &a, int b) { return AD<T>(a.v_ * b); }
template <typename T> AD<T> pow(const AD<T> &a, int e) { return AD<T>(std::pow(a.v_, e)); }
template <typename T> AD<T> pow(const AD<T> &a, double e) { return AD<T>(std::pow(a.v_, e)); }
template <typename T> AD<T> pow(const AD<T> &a, const AD<T> &e) { return AD<T>(std::pow(a.v_, e.v_)); }
template <typename T> AD<T> sin(const AD<T> &a) { return AD<T>(std::sin(a.v_)); }
template <typename T> AD<T> cos(const AD<T> &a) { return AD<T>(std::cos(a.v_)); }
template <typename T> AD<T> tan(const AD<T> &a) { return AD<T>(std::tan(a.v_)); }
template <typename T> AD<T> atan(const AD<T> &a) { return AD<T>(std::atan(a.v_)); }
template <typename T> AD<T> atan2(const AD<T> &a, const AD<T> &b) { return AD<T>(std::atan2(a.v_, b.v_)); }
template <typename T> AD<T> sqrt(const AD<T> &a) { return AD<T>(std::sqrt(a.v_)); }
template <typename T> AD<T> fabs(const AD<T> &a) { return AD<T>(std::fabs(a.v_)); }
}  // namespace CppAD
#endif
