// counting scalar: every arithmetic operation on `real` bumps a global counter (test tooling: tests/tools/count_flops.py)
#pragma once
#include <cmath>
#include <cstdint>
struct FlopCounters { uint64_t add, mul, div, sqrt_, trans, cmp; };
extern FlopCounters g_fc;
struct real {
  double v;
  real() : v(0) {}
  real(double x) : v(x) {}
  real(float x) : v(x) {}
  real(int x) : v(x) {}
  real(long x) : v((double)x) {}
  real(unsigned x) : v(x) {}
  explicit operator double() const { return v; }
  explicit operator float() const { return (float)v; }
  explicit operator int() const { return (int)v; }
  explicit operator bool() const { return v != 0; }
  real operator-() const { return real(-v); }
  real& operator+=(real o) { g_fc.add++; v += o.v; return *this; }
  real& operator-=(real o) { g_fc.add++; v -= o.v; return *this; }
  real& operator*=(real o) { g_fc.mul++; v *= o.v; return *this; }
  real& operator/=(real o) { g_fc.div++; v /= o.v; return *this; }
};
#define BINOP(op, ctr) \
  inline real operator op(real a, real b) { g_fc.ctr++; return real(a.v op b.v); } \
  inline real operator op(real a, double b) { g_fc.ctr++; return real(a.v op b); } \
  inline real operator op(double a, real b) { g_fc.ctr++; return real(a op b.v); } \
  inline real operator op(real a, int b) { g_fc.ctr++; return real(a.v op b); } \
  inline real operator op(int a, real b) { g_fc.ctr++; return real(a op b.v); }
BINOP(+, add) BINOP(-, add) BINOP(*, mul) BINOP(/, div)
#define CMPOP(op) \
  inline bool operator op(real a, real b) { g_fc.cmp++; return a.v op b.v; } \
  inline bool operator op(real a, double b) { g_fc.cmp++; return a.v op b; } \
  inline bool operator op(double a, real b) { g_fc.cmp++; return a op b.v; } \
  inline bool operator op(real a, int b) { g_fc.cmp++; return a.v op b; } \
  inline bool operator op(int a, real b) { g_fc.cmp++; return a op b.v; }
CMPOP(<) CMPOP(>) CMPOP(<=) CMPOP(>=) CMPOP(==) CMPOP(!=)
inline bool operator!(real a) { return a.v == 0; }
inline real sqrt(real a) { g_fc.sqrt_++; return real(std::sqrt(a.v)); }
inline real fabs(real a) { return real(std::fabs(a.v)); }
#define F2(name) \
  inline real name(real a, real b) { g_fc.cmp++; return real(std::name(a.v, b.v)); } \
  inline real name(real a, double b) { g_fc.cmp++; return real(std::name(a.v, b)); } \
  inline real name(double a, real b) { g_fc.cmp++; return real(std::name(a, b.v)); }
F2(fmax) F2(fmin)
#define T1(name) inline real name(real a) { g_fc.trans++; return real(std::name(a.v)); }
T1(sin) T1(cos) T1(acos) T1(asin) T1(exp) T1(log) T1(tan) T1(atan)
inline real atan2(real a, real b) { g_fc.trans++; return real(std::atan2(a.v, b.v)); }
inline real pow(real a, real b) { g_fc.trans++; return real(std::pow(a.v, b.v)); }
inline real pow(real a, double b) { g_fc.trans++; return real(std::pow(a.v, b)); }
inline real floor(real a) { return real(std::floor(a.v)); }
inline bool isfinite(real a) { return std::isfinite(a.v); }
inline long lround(real a) { return std::lround(a.v); }
inline long lrint(real a) { return std::lrint(a.v); }
