// Drop-in replacement of the reference's include/path_optimizer/solver/solver.hpp.
// Put this directory BEFORE the reference's include/ on the include path: src/path_optimizer/path_optimizer.cpp then compiles unchanged —
// `OsqpSolver::create(FLAGS_optimization_method, *reference_path_, *vehicle_state_, size_)` and `solver->solve(final_path)` (path_optimizer.cpp:182-183)
// resolve to the class below, which packs the reference's own ReferencePath / VehicleState objects into po_batch_in and calls libpo_hip.so.
// src/solver/*.cpp and OsqpEigen are no longer needed by the path QP.
#ifndef PATH_OPTIMIZER_SOLVER_HPP
#define PATH_OPTIMIZER_SOLVER_HPP
#define PO_USE_REFERENCE_TYPES
#include "path_optimizer_amd/solver.hpp"
#endif  // PATH_OPTIMIZER_SOLVER_HPP
