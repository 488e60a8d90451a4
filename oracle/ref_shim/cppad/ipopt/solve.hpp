// Stand-in for <cppad/ipopt/solve.hpp>: see ../cppad.hpp.  solve() aborts: IPOPT is absent and its branch is not the default.
#ifndef PO_REF_SHIM_CPPAD_IPOPT
#define PO_REF_SHIM_CPPAD_IPOPT
#include "../cppad.hpp"
namespace CppAD { namespace ipopt {
template <typename Dvector> class solve_result {
 public:
    enum status_type { not_defined, success, unknown };
    status_type status = not_defined;
    Dvector x;
    double obj_value = 0;
};
template <typename Dvector, typename FG>
void solve(const std::string &, const Dvector &, const Dvector &, const Dvector &, const Dvector &, const Dvector &, FG &, solve_result<Dvector> &) {
    std::abort();
}
}}  // namespace CppAD::ipopt
#endif
