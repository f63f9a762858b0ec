// Scalar that computes in double and -- inside the stages selected by g_round_mask -- rounds the result of EVERY operation to
// fp32: an emulation of a plain fp32 implementation of the oracle's algorithm, switchable per pipeline stage
// (tests/tools/precision_study.py).  Test tooling only.
#pragma once
#include <cmath>
#include <cstdint>
extern int g_round_on;          // current stage rounds
extern unsigned g_round_mask;   // bit k: stage k (MMO_ST_*) rounds to fp32
#define MMO_STAGE(k) (g_round_on = (int)((g_round_mask >> (k)) & 1u))
static inline double rnd_(double x) { return g_round_on ? (double)(float)x : x; }
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
  real& operator+=(real o) { v = rnd_(v + o.v); return *this; }
  real& operator-=(real o) { v = rnd_(v - o.v); return *this; }
  real& operator*=(real o) { v = rnd_(v * o.v); return *this; }
  real& operator/=(real o) { v = rnd_(v / o.v); return *this; }
};
#define BINOP(op) \
  inline real operator op(real a, real b) { return real(rnd_(a.v op b.v)); } \
  inline real operator op(real a, double b) { return real(rnd_(a.v op b)); } \
  inline real operator op(double a, real b) { return real(rnd_(a op b.v)); } \
  inline real operator op(real a, int b) { return real(rnd_(a.v op b)); } \
  inline real operator op(int a, real b) { return real(rnd_(a op b.v)); }
BINOP(+) BINOP(-) BINOP(*) BINOP(/)
#define CMPOP(op) \
  inline bool operator op(real a, real b) { return a.v op b.v; } \
  inline bool operator op(real a, double b) { return a.v op b; } \
  inline bool operator op(double a, real b) { return a op b.v; } \
  inline bool operator op(real a, int b) { return a.v op b; } \
  inline bool operator op(int a, real b) { return a op b.v; }
CMPOP(<) CMPOP(>) CMPOP(<=) CMPOP(>=) CMPOP(==) CMPOP(!=)
inline bool operator!(real a) { return a.v == 0; }
inline real sqrt(real a) { return real(rnd_(std::sqrt(a.v))); }
inline real fabs(real a) { return real(std::fabs(a.v)); }
#define F2(name) \
  inline real name(real a, real b) { return real(std::name(a.v, b.v)); } \
  inline real name(real a, double b) { return real(std::name(a.v, b)); } \
  inline real name(double a, real b) { return real(std::name(a, b.v)); }
F2(fmax) F2(fmin)
#define T1(name) inline real name(real a) { return real(rnd_(std::name(a.v))); }
T1(sin) T1(cos) T1(acos) T1(asin) T1(exp) T1(log) T1(tan) T1(atan)
inline real atan2(real a, real b) { return real(rnd_(std::atan2(a.v, b.v))); }
inline real pow(real a, real b) { return real(rnd_(std::pow(a.v, b.v))); }
inline real pow(real a, double b) { return real(rnd_(std::pow(a.v, b))); }
inline real floor(real a) { return real(std::floor(a.v)); }
inline bool isfinite(real a) { return std::isfinite(a.v); }
inline long lround(real a) { return std::lround(a.v); }
inline long lrint(real a) { return std::lrint(a.v); }
