// Stand-in for <grid_map_core/grid_map_core.hpp> (ROS package grid_map_core is not installed here and is not part of
// /root/reference), so that the reference's src/tools/{Map,collision_checker}.cpp and src/data_struct/reference_path_impl.cpp
// compile WHERE THEY LIE.  Only the three members those files call exist.  The sampling arithmetic is NOT re-implemented
// here: it forwards to the oracle's restatement (oracle/po_oracle.c), so what this shim pins is the reference's in-tree
// logic around the map, not grid_map itself ("parity unpinned" for the library, see po_oracle.h).  TEST INFRASTRUCTURE ONLY.
#ifndef PO_REF_SHIM_GRID_MAP_CORE
#define PO_REF_SHIM_GRID_MAP_CORE
#include <string>

#include "Eigen/Core"
#include "po_oracle.h"

namespace grid_map {
typedef Eigen::Vector2d Position;
enum class InterpolationMethods { INTER_NEAREST, INTER_LINEAR };
class GridMap {
 public:
    GridMap() : m_{} {}
    explicit GridMap(const po_map &m) : m_(m) {}
    bool exists(const std::string &layer) const { return layer == "distance"; }
    bool isInside(const Position &p) const { return po_oracle_map_inside(&m_, p(0), p(1)) != 0; }
    float atPosition(const std::string &, const Position &p, InterpolationMethods = InterpolationMethods::INTER_LINEAR) const {
        return po_oracle_map_at_linear(&m_, p(0), p(1));
    }
 private:
    po_map m_;
};
}  // namespace grid_map
#endif
