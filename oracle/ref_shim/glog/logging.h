// stand-in for glog: swallow log streams; CHECK_* keep their abort semantics.  TEST INFRASTRUCTURE ONLY.
#ifndef PO_REF_SHIM_GLOG
#define PO_REF_SHIM_GLOG
#include <cstdlib>
#include <iostream>
struct PoNullStream { template <typename T> PoNullStream &operator<<(const T &) { return *this; } PoNullStream &operator<<(std::ostream &(*)(std::ostream &)) { return *this; } };
#define LOG(sev) PoNullStream()
#define DLOG(sev) PoNullStream()
#define LOG_EVERY_N(sev, n) PoNullStream()
#define LOG_IF(sev, c) PoNullStream()
#define CHECK_EQ(a, b) if (!((a) == (b))) std::abort(); else PoNullStream()
#define CHECK_LE(a, b) if (!((a) <= (b))) std::abort(); else PoNullStream()
#define CHECK_NOTNULL(p) (p)
#define CHECK(c) if (!(c)) std::abort(); else PoNullStream()
#endif
