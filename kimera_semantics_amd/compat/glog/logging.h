// Minimal stand-in for the glog macros used by the Kimera-Semantics hot path.  CHECK failures
// and LOG(FATAL) abort the process like glog does; other severities go to stderr.
#pragma once
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sstream>

namespace ks_minilog {
struct Message {
  std::ostringstream ss;
  bool fatal;
  Message(const char* file, int line, const char* sev, bool f) : fatal(f) { ss << sev << " " << file << ":" << line << "] "; }
  ~Message() {
    std::cerr << ss.str() << std::endl;
    if (fatal) std::abort();
  }
  std::ostream& stream() { return ss; }
};
struct Voidify {
  void operator&(std::ostream&) {}
};
template <typename T>
T& CheckNotNull(const char* file, int line, const char* expr, T& t) {
  if (t == nullptr) Message(file, line, "F", true).stream() << "'" << expr << "' Must be non NULL";
  return t;
}
template <typename T>
T CheckNotNull(const char* file, int line, const char* expr, T&& t) {
  if (t == nullptr) Message(file, line, "F", true).stream() << "'" << expr << "' Must be non NULL";
  return static_cast<T&&>(t);
}
}  // namespace ks_minilog

#define KS_LOG_INFO ::ks_minilog::Message(__FILE__, __LINE__, "I", false).stream()
#define KS_LOG_WARNING ::ks_minilog::Message(__FILE__, __LINE__, "W", false).stream()
#define KS_LOG_ERROR ::ks_minilog::Message(__FILE__, __LINE__, "E", false).stream()
#define KS_LOG_FATAL ::ks_minilog::Message(__FILE__, __LINE__, "F", true).stream()
#define LOG(sev) KS_LOG_##sev
#define VLOG(n) if (true) {} else KS_LOG_INFO
#define LOG_IF(sev, cond) !(cond) ? (void)0 : ::ks_minilog::Voidify() & LOG(sev)

#define CHECK(cond) (cond) ? (void)0 : ::ks_minilog::Voidify() & KS_LOG_FATAL << "Check failed: " #cond " "
#define KS_CHECK_OP(a, b, op) CHECK((a)op(b))
#define CHECK_EQ(a, b) KS_CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) KS_CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) KS_CHECK_OP(a, b, <)
#define CHECK_LE(a, b) KS_CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) KS_CHECK_OP(a, b, >)
#define CHECK_GE(a, b) KS_CHECK_OP(a, b, >=)
#define CHECK_NEAR(a, b, m) CHECK(std::abs((a) - (b)) <= (m))
#define CHECK_NOTNULL(p) ::ks_minilog::CheckNotNull(__FILE__, __LINE__, #p, (p))
// debug checks compile to nothing (release build of the reference)
#define DCHECK(cond) if (true) {} else KS_LOG_INFO
#define DCHECK_EQ(a, b) DCHECK((a) == (b))
#define DCHECK_NE(a, b) DCHECK((a) != (b))
#define DCHECK_LT(a, b) DCHECK((a) < (b))
#define DCHECK_LE(a, b) DCHECK((a) <= (b))
#define DCHECK_GT(a, b) DCHECK((a) > (b))
#define DCHECK_GE(a, b) DCHECK((a) >= (b))
