// stand-in for gflags: FLAGS_* become plain globals, so the reference's own src/config/planning_flags.cpp
// (compiled where it lies) provides the default values.  TEST INFRASTRUCTURE ONLY.
#ifndef PO_REF_SHIM_GFLAGS
#define PO_REF_SHIM_GFLAGS
#include <string>
#define DECLARE_double(name) extern double FLAGS_##name
#define DECLARE_bool(name) extern bool FLAGS_##name
#define DECLARE_int32(name) extern int FLAGS_##name
#define DECLARE_string(name) extern std::string FLAGS_##name
#define DEFINE_double(name, val, txt) double FLAGS_##name = (val)
#define DEFINE_bool(name, val, txt) bool FLAGS_##name = (val)
#define DEFINE_int32(name, val, txt) int FLAGS_##name = (val)
#define DEFINE_string(name, val, txt) std::string FLAGS_##name = (val)
namespace google { template <typename T, typename F> bool RegisterFlagValidator(const T *, F) { return true; } }
namespace gflags = google;
#endif
