// Host-side data surface the solver reads: the same types, member names and semantics as the reference's
//   State                      include/path_optimizer/data_struct/data_struct.hpp:13-30
//   CoveringCircleBounds       include/path_optimizer/data_struct/data_struct.hpp:72-91
//   ReferencePath (getters)    include/path_optimizer/data_struct/reference_path.hpp:34-37
//   VehicleState               include/path_optimizer/data_struct/vehicle_state_frenet.hpp:11-34
// so that PathOptimizer::optimizePath (src/path_optimizer/path_optimizer.cpp:182-183) compiles unchanged against
// this header.  Only what the hot path touches is declared; the producers (spline re-sampling, corridor bounds from
// the distance map, limits) stay on the reference side.
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

namespace PathOptimizationNS {

struct State {
    State() = default;
    State(double x, double y, double z = 0, double k = 0, double s = 0, double v = 0, double a = 0) : x(x), y(y), z(z), k(k), s(s), v(v), a(a) {}
    double x{}, y{}, z{}, k{}, s{}, v{}, a{};
};

struct CoveringCircleBounds {
    struct SingleCircleBounds {
        SingleCircleBounds &operator=(const std::vector<double> &bounds) {
            ub = bounds[0];
            lb = bounds[1];
            return *this;  // the reference forgets this return (UB hidden by -w); fixed deliberately
        }
        void set(const std::vector<double> &bounds, const State &center) {
            ub = bounds[0]; lb = bounds[1]; x = center.x; y = center.y; heading = center.z;
        }
        double ub{};  // left
        double lb{};  // right
        double x{}, y{}, heading{};
    } c0, c1, c2, c3;
};

class ReferencePath {
 public:
    std::size_t getSize() const { return states_.size(); }
    const std::vector<State> &getReferenceStates() const { return states_; }
    const std::vector<CoveringCircleBounds> &getBounds() const { return bounds_; }
    const std::vector<double> &getMaxKList() const { return max_k_; }
    const std::vector<double> &getMaxKpList() const { return max_kp_; }
    void setReference(const std::vector<State> &reference) { states_ = reference; }
    void setReference(std::vector<State> &&reference) { states_ = std::move(reference); }
    // the reference fills these from the distance map / speed profile (updateBounds, updateLimits); here they are set directly
    void setBounds(std::vector<CoveringCircleBounds> b) { bounds_ = std::move(b); }
    void setLimits(std::vector<double> max_k, std::vector<double> max_kp) { max_k_ = std::move(max_k); max_kp_ = std::move(max_kp); }
    void clear() { states_.clear(); bounds_.clear(); max_k_.clear(); max_kp_.clear(); }
 private:
    std::vector<State> states_;
    std::vector<CoveringCircleBounds> bounds_;
    std::vector<double> max_k_, max_kp_;
};

class VehicleState {
 public:
    VehicleState() = default;
    VehicleState(const State &start_state, const State &end_state, double offset = 0, double heading_error = 0)
        : start_(start_state), end_(end_state), offset_(offset), heading_error_(heading_error) {}
    const State &getStartState() const { return start_; }
    const State &getEndState() const { return end_; }
    void setStartState(const State &s) { start_ = s; }
    void setEndState(const State &s) { end_ = s; }
    std::vector<double> getInitError() const { return {offset_, heading_error_}; }
    void setInitError(double init_offset, double init_heading_error) { offset_ = init_offset; heading_error_ = init_heading_error; }
 private:
    State start_, end_;
    double offset_{}, heading_error_{};
};

}  // namespace PathOptimizationNS
